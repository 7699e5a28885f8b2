"""Static instruction mix of selected kernels from `hipcc -S --cuda-device-only ... csrc/unity.hip` output (build/mjhip.s)."""
import collections, re, sys
path = sys.argv[1] if len(sys.argv) > 1 else "build/mjhip.s"
pats = sys.argv[2:] or ["k_solve_plusILi7ELi2ELb0", "k_solve_plusILi7ELi2ELb1", "k_midILi32", "k_fwd_pos_plusILi32", "k_integrate_plusILi32"]
name, lines, out = None, [], {}
for l in open(path):
  m = re.match(r"^(_Z\w+):", l)
  if m:
    if name:
      out[name] = lines
    name, lines = m.group(1), []
  elif name and l.startswith("\t") and not l.strip().startswith((".", ";")):
    lines.append(l.strip())
  if l.startswith("\t.end_amdhsa_kernel") or l.startswith(".Lfunc_end"):
    if name:
      out[name] = lines
    name = None
for n, ls in out.items():
  if not any(p in n for p in pats):
    continue
  c = collections.Counter(x.split()[0] for x in ls)
  valu = sum(v for k, v in c.items() if k.startswith("v_"))
  print(n[:60], "instrs", len(ls), "valu", valu, "pk", sum(v for k, v in c.items() if k.startswith("v_pk")),
        "dpp", sum(1 for x in ls if "row_" in x or "quad_perm" in x), "readlane", c.get("v_readlane_b32", 0), "s_nop", c.get("s_nop", 0),
        "waitcnt", c.get("s_waitcnt", 0), "ds", sum(v for k, v in c.items() if k.startswith("ds_")),
        "global", sum(v for k, v in c.items() if k.startswith("global_")))
  print("    ", c.most_common(16))
