"""profiles/round1_pmc_summary.json from a tools/profile.sh summary: HBM traffic per launch of the dominant kernel.

usage: python tools/make_pmc_summary.py gpurun_out/prof_<tag>/summary.json profiles/roundN_pmc_<solver>.json <solver>
FETCH_SIZE / WRITE_SIZE are collected in their own rocprofv3 --pmc passes (tools/profile.sh); FETCH_SIZE is doubled on
gfx950 as MI355X_MICROARCH.md prescribes (it under-reports wide coalesced reads by 2x), both are KiB units.
"""
import json, sys
src, dst = sys.argv[1], sys.argv[2]
s = json.load(open(src))
out = {"source": src, "solver": sys.argv[3] if len(sys.argv) > 3 else None, "kernels": {}}
for k, c in s.get("pmc_per_dispatch", {}).items():
  if "fetch_bytes_gfx950_corrected" in c:
    out["kernels"][k] = {"fetch_bytes_per_launch": c["fetch_bytes_gfx950_corrected"], "write_bytes_per_launch": c.get("write_bytes"),
                         "hbm_bytes_per_launch": c["fetch_bytes_gfx950_corrected"] + c.get("write_bytes", 0.0),
                         "valu_insts_per_launch": c.get("SQ_INSTS_VALU"), "busy_cycles_sum": c.get("SQ_BUSY_CYCLES")}
dom = [k for k in out["kernels"] if k.startswith("k_solve<") or k == "k_solve_newton"]
dom.sort(key=lambda k: -(out["kernels"][k].get("busy_cycles_sum") or 0.0))
if dom:
  out["k_solve_hbm_bytes_per_launch"] = out["kernels"][dom[0]]["hbm_bytes_per_launch"]
  out["k_solve_kernel"] = dom[0]
for t in s.get("kernel_trace", []):
  if t["kernel"] in out["kernels"]:
    out["kernels"][t["kernel"]]["mean_us"] = t["mean_us"]
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out)[:600])
