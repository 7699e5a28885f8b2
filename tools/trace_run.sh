#!/bin/bash
# GPU box: kernel trace of one registry benchmark (benchmarks/run.py -f <name>): which kernels take the time.  usage: tools/trace_run.sh <filter> [nstep]
OUT=$PWD/gpurun_out/prof_run_$1
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $OLDPWD/benchmarks/run.py -f "$1" --nstep ${2:-300} > $OUT/run.log 2>&1)
grep "steps_per_second\|converged\|step " $OUT/run.log | head -5
python - $OUT <<'PY'
import sys, glob, os
sys.path.insert(0, "tools")
import summarize_profile as sp
for f in glob.glob(os.path.join(sys.argv[1], "trace", "*.db")):
  for k in sp.kernel_trace(f)[:18]:
    print(f"{k['kernel'][:40]:40s} calls {k['calls']:5d} mean {k['mean_us']:8.1f} us  {k['pct']:5.1f} %")
PY
rm -rf $OUT/trace
