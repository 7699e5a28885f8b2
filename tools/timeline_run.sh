#!/bin/bash
# GPU box: dispatch timeline (start / duration / queue of the last dispatches) of one registry benchmark run through benchmarks/run.py (graph replay).
# usage: tools/timeline_run.sh <filter> [nstep] [ndispatch]
OUT=$PWD/gpurun_out/prof_tl_$(echo $1 | tr -cd 'a-z0-9_')
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace -d $OUT/trace -o trace -- python $OLDPWD/benchmarks/run.py -f "$1" --no-trace --nstep ${2:-150} > $OUT/run.log 2>&1)
python - $OUT ${3:-60} <<'PY'
import sys, glob, os
sys.path.insert(0, "tools")
import summarize_profile as sp
for f in glob.glob(os.path.join(sys.argv[1], "trace", "*.db")):
  for r in sp.timeline(f, int(sys.argv[2])):
    print(f"{r['kernel'][:34]:34s} q{r['queue']}  start {r['start_us']:9.1f}  dur {r['dur_us']:7.1f}  end {r['end_us']:9.1f}")
PY
rm -rf $OUT/trace
