#!/bin/bash
# dispatch timeline of the fused step for any model: tools/timeline_model.sh <mjcf> nworld nconmax njmax [solver]
OUT=$PWD/gpurun_out/prof_timeline
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $PWD/tools/bench_model.py $PWD/$1 $2 $3 $4 $5"
(cd /tmp && rocprofv3 --kernel-trace -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1)
python - <<PY
import glob, sqlite3, sys
sys.path.insert(0, "tools")
import summarize_profile as sp
f = glob.glob("$OUT/trace/*.db")[0]
rows = sp.timeline(f, 3000)
# the fused 200-step loop is the longest run of 5-kernel periods: print three periods from the middle of the trace
mid = len(rows) // 4
for r in rows[mid:mid + 16]:
  print(f"{r['kernel'][:40]:40s} start {r['start_us']:10.1f} dur {r['dur_us']:7.1f}")
PY
rm -rf $OUT/trace
