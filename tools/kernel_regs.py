"""Register / scratch figures of every kernel in a `hipcc -S --cuda-device-only` assembly file (amdhsa metadata):
python tools/kernel_regs.py build/t/x.s [name filter]"""
import re, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("  - .agpr_count:")[1:]:
  g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
  name = g("name")
  if flt in name:
    print(f"{name[:70]:70s} vgpr {g('vgpr_count'):>4s} agpr {blk.split()[0]:>4s} spill {g('vgpr_spill_count'):>4s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size')}")
