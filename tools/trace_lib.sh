#!/bin/bash
# GPU box: kernel trace of one registry benchmark with a given library variant: tools/trace_lib.sh <lib or ""> <filter> [nstep]
OUT=$PWD/gpurun_out/prof_lib
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
[ -n "$1" ] && export MJH_LIB=$PWD/$1
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $OLDPWD/benchmarks/run.py -f "$2" --nstep ${3:-100} > $OUT/run.log 2>&1)
python - $OUT <<'PY'
import sys, glob, os
sys.path.insert(0, "tools")
import summarize_profile as sp
for f in glob.glob(os.path.join(sys.argv[1], "trace", "*.db")):
  for k in sp.kernel_trace(f)[:9]:
    print(f"{k['kernel'][:40]:40s} calls {k['calls']:5d} mean {k['mean_us']:8.1f} us  {k['pct']:5.1f} %")
PY
rm -rf $OUT/trace
