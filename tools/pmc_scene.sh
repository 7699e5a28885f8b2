#!/bin/bash
# GPU box: SQ counters (issue / wait / instruction-cache) per kernel of one registry benchmark -> gpurun_out/<tag>_pmc_<scene>.json
# usage: tools/pmc_scene.sh <tag, e.g. round4> <registry name, e.g. aloha_pot> [nstep]
set -u
TAG=${1:-round4}; SCENE=${2:-aloha_pot}; N=${3:-200}
OUT=$PWD/gpurun_out/pmc_scene_$SCENE; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python $PWD/benchmarks/run.py -f ^$SCENE\$ --nstep $N"
(cd /tmp && rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $OUT/p1 -o pmc -- $CMD > $OUT/p1.log 2>&1)
(cd /tmp && rocprofv3 --pmc SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d $OUT/p2 -o pmc -- $CMD > $OUT/p2.log 2>&1)
python - $OUT "$CMD" gpurun_out/${TAG}_pmc_$SCENE.json <<'PY'
import sys, glob, os, sqlite3, collections, json
out, cmd, dst = sys.argv[1:4]
acc = collections.defaultdict(dict)
for sub in ("p1", "p2"):
  for f in glob.glob(os.path.join(out, sub, "**", "*.db"), recursive=True):
    con = sqlite3.connect(f)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    pmc = [t for t in tabs if "pmc_event" in t][0]; info = [t for t in tabs if "info_pmc" in t][0]
    disp = [t for t in tabs if "kernel_dispatch" in t][0]; sym = [t for t in tabs if "kernel_symbol" in t][0]
    q = f"select s.kernel_name, i.name, sum(p.value), count(distinct d.id) from {pmc} p join {info} i on p.pmc_id = i.id join {disp} d on p.event_id = d.event_id join {sym} s on d.kernel_id = s.id group by s.kernel_name, i.name"
    for k, n, v, c in con.execute(q):
      acc[k.replace(".kd", "")][n] = v / max(c, 1)
res = {"command": cmd, "note": "per launch; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_ANY tick once per 4 cycles, SQ_BUSY_CYCLES is summed over 32 shader engines", "kernels": {}}
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0)):
  if v.get("SQ_BUSY_CYCLES", 0) < 1e5:
    continue
  dur = v["SQ_BUSY_CYCLES"] / 32.0
  e = {n: round(x) for n, x in v.items()}
  e["duration_cycles"] = round(dur)
  if "SQ_INSTS_VALU" in v: e["valu_issue_frac"] = round(v["SQ_INSTS_VALU"] * 2.0 / (1024.0 * dur), 4)
  if v.get("SQ_WAVE_CYCLES"):
    e["waves_per_simd"] = round(4.0 * v["SQ_WAVE_CYCLES"] / (1024.0 * dur), 3)
    e["wait_frac"] = round(v.get("SQ_WAIT_ANY", 0) / v["SQ_WAVE_CYCLES"], 4)
    e["issue_stall_frac"] = round(v.get("SQ_WAIT_INST_ANY", 0) / v["SQ_WAVE_CYCLES"], 4)
  if "SQ_LDS_IDX_ACTIVE" in v: e["lds_busy_frac"] = round(v["SQ_LDS_IDX_ACTIVE"] / (256.0 * dur), 4)
  if v.get("SQC_ICACHE_REQ"): e["icache_miss_frac"] = round(v.get("SQC_ICACHE_MISSES", 0) / v["SQC_ICACHE_REQ"], 6)
  res["kernels"][k[:60]] = e
json.dump(res, open(dst, "w"), indent=1)
for k, e in list(res["kernels"].items())[:10]:
  print(f"{k[:44]:44s} cycles {e['duration_cycles']:8d} valu {e.get('valu_issue_frac')} waves/SIMD {e.get('waves_per_simd')} wait {e.get('wait_frac')} lds {e.get('lds_busy_frac')} icache miss {e.get('icache_miss_frac')}")
PY
rm -rf $OUT
