"""Print a one-line digest of bench.py's JSON (reads stdin)."""
import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
for line in sys.stdin:
  line = line.strip()
  if line.startswith("{"):
    o = json.loads(line)
    pk = o.get("per_kernel_us", {})
    print(tag, o["config"]["workload"].split(",")[2].strip(), "env-steps/s", round(o["value"]), "ms/step", round(o["ms_per_step"], 4),
          {k: round(v, 1) for k, v in pk.items() if k != "other"}, "niter", round(o["solver_niter_mean"], 2))
