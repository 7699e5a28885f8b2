"""Disassembly of ONE gfx950 kernel inside a hipcc object / shared library, with an instruction-class histogram:
python tools/kernel_disasm.py <file.o | lib.so> <kernel name substring> [--dump]"""
import re, struct, subprocess, sys, tempfile, os, collections
b = open(sys.argv[1], "rb").read()
name = sys.argv[2]
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
    t = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout
    os.unlink(f.name)
    for blk in re.split(r"\n(?=[0-9a-f]+ <)", t):
      head = blk.split("\n", 1)[0]
      if name in head and "<" in head:
        ins = [l.split()[0] for l in blk.split("\n")[1:] if l.strip() and not l.strip().startswith(("<", ";")) and not l.strip().endswith(":")]
        cls = collections.Counter()
        for x in ins:
          k = ("ds_" if x.startswith("ds_") else "global/flat/scratch" if x.startswith(("global_", "flat_", "scratch_", "buffer_")) else "s_waitcnt" if x.startswith("s_waitcnt") else
               "s_nop" if x.startswith("s_nop") else "salu/branch" if x.startswith("s_") else "mfma" if "mfma" in x else "valu")
          cls[k] += 1
        print(head.strip(), len(ins), dict(cls))
        if "--dump" in sys.argv:
          print(blk)
