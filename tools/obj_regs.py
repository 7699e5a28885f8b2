"""Registers / spills / scratch of the gfx950 kernels inside a hipcc object or shared library (reads the offload bundle directly):
python tools/obj_regs.py <file.o | lib.so> [name filter]"""
import re, struct, subprocess, sys, tempfile, os
b = open(sys.argv[1], "rb").read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for mm in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", b):
  i = mm.start()
  p = i + 24
  nb = struct.unpack("<Q", b[p:p + 8])[0]
  p += 8
  for _ in range(nb):
    off, size, tl = struct.unpack("<QQQ", b[p:p + 24])
    p += 24
    tr = b[p:p + tl].decode()
    p += tl
    if "gfx950" not in tr:
      continue
    with tempfile.NamedTemporaryFile(suffix=".elf", delete=False) as f:
      f.write(b[i + off:i + off + size])
    t = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
    os.unlink(f.name)
    for blk in t.split("- .agpr_count:")[1:]:
      g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
      if flt in g("name"):
        print(f"{g('name')[:80]:80s} vgpr {g('vgpr_count'):>4s} agpr {blk.split()[0]:>4s} sgpr {g('sgpr_count'):>4s} spill {g('vgpr_spill_count'):>4s} scratch {g('private_segment_fixed_size'):>5s}")
