#!/bin/bash
# GPU box: kernel trace of the chunked stepping experiment (tools/chunk_overlap.py): do launches of different streams overlap?
OUT=$PWD/gpurun_out/prof_chunks
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp CHUNKS=${CHUNKS:-2} K=${K:-20} WARM=${WARM:-40} NOSYNC=1
(cd /tmp && rocprofv3 --kernel-trace -d $OUT/trace -o trace -- python $OLDPWD/tools/chunk_overlap.py cg > $OUT/trace.log 2>&1)
python - $OUT <<'PY'
import sys, glob, os
sys.path.insert(0, "tools")
import summarize_profile as sp
for f in glob.glob(os.path.join(sys.argv[1], "trace", "*.db")):
  for r in sp.timeline(f, 48):
    print(f"{r['kernel'][:28]:28s} q{r['queue']}  start {r['start_us']:9.1f}  dur {r['dur_us']:7.1f}  end {r['end_us']:9.1f}")
PY
tail -3 $OUT/trace.log
rm -rf $OUT/trace
