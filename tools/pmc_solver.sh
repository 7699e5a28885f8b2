#!/bin/bash
# GPU box: kernel trace + the two SQ counter passes of the headline bench, solver kernel only (quick A/B of solver variants).
# usage: [ENV=...] tools/pmc_solver.sh <tag> [bench args...]
set -u
TAG=${1:-x}; shift || true
OUT=$PWD/gpurun_out/pmcs_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-steady --no-configs $*"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $OUT/pmc_sq -o pmc -- $BENCH > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/pmc_sq2 -o pmc -- $BENCH > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_IFETCH SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM -d $OUT/pmc_sq3 -o pmc -- $BENCH > $OUT/pmc_sq3.log 2>&1
cd - >/dev/null
python - $OUT <<'PY'
import sys, glob, os, json
sys.path.insert(0, "tools")
import summarize_profile as sp
out = sys.argv[1]
res = {}
for f in glob.glob(os.path.join(out, "trace", "*.db")):
  for k in sp.kernel_trace(f):
    if "solve" in k["kernel"]:
      res[k["kernel"]] = {"mean_us": k["mean_us"]}
for tag in ("pmc_sq", "pmc_sq2", "pmc_sq3"):
  for f in glob.glob(os.path.join(out, tag, "*.db")):
    for k, cs in sp.pmc(f).items():
      if k in res:
        res[k].update(cs)
print(json.dumps(res, indent=1))
open(os.path.join(out, "solver.json"), "w").write(json.dumps(res, indent=1))
PY
rm -rf $OUT/trace $OUT/pmc_sq $OUT/pmc_sq2 $OUT/pmc_sq3
