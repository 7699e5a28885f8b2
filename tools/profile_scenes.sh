#!/bin/bash
# GPU box: rocprofv3 kernel trace of the feature scenes (tools/bench_scenes.py: meshes, height field, sleeping, sensors) -> small summary.
# usage: tools/profile_scenes.sh <tag, e.g. round3> [nworld]
set -u
TAG=${1:-round3}; NW=${2:-2048}
OUT=$PWD/gpurun_out/prof_scenes_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $PWD/tools/bench_scenes.py $NW"
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1)
grep -v amdgpu.ids $OUT/trace.log | tail -12
python tools/summarize_profile.py $OUT $OUT/summary.json > /dev/null 2>&1
python - $OUT/summary.json "$CMD" gpurun_out/${TAG}_feature_scenes_summary.json <<'PY'
import json, sys
s = json.load(open(sys.argv[1]))
s["command"] = sys.argv[2]
s.pop("timeline", None)
json.dump(s, open(sys.argv[3], "w"), indent=1)
for k in s.get("kernel_trace", [])[:10]:
  print(f"{k['kernel'][:40]:40s} calls {k['calls']:5d} mean {k['mean_us']:8.1f} us  {k['pct']:5.1f} %")
PY
rm -rf $OUT/trace
