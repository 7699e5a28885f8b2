#!/bin/bash
# GPU box: kernel-trace only; prints the dispatch timeline of the last steps (overlap, gaps).  usage: tools/timeline.sh [bench args]
OUT=$PWD/gpurun_out/prof_timeline
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline $*"
(cd /tmp && rocprofv3 --kernel-trace -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1)
python tools/summarize_profile.py $OUT $OUT/summary.json > /dev/null 2>&1
rm -rf $OUT/trace
python - <<'PY'
import json
s = json.load(open("gpurun_out/prof_timeline/summary.json"))
for r in s["timeline"]:
  print(f"{r['kernel'][:28]:28s} q{r['queue']}  start {r['start_us']:9.1f}  dur {r['dur_us']:7.1f}  end {r['end_us']:9.1f}")
PY
