#!/bin/bash
# GPU box: kernel-trace + PMC summaries of the headline bench for both solvers on two windows of the rollout -- the driver's
# (--warmup 5: steps 5-55, the humanoids still falling, nefc ~ 15) and the steady one (--warmup 300, nefc ~ 45) --, plus the bench lines.
# usage: tools/profile_round.sh <round tag, e.g. round4> [solvers, default "cg newton"]
set -u
R=${1:-round4}
export PROFILE_LAST=200
for sv in ${2:-cg newton}; do
  for win in steady early; do
    W=300; LABEL="steps 300-350 of the rollout (steady state)"
    if [ $win = early ]; then W=5; LABEL="steps 5-55 of the rollout (the driver's --warmup 5 window)"; fi
    bash tools/profile.sh ${R}_${sv}_$win --solver $sv --warmup $W > /dev/null 2>&1
    python tools/make_pmc_summary.py gpurun_out/prof_${R}_${sv}_$win/summary.json gpurun_out/${R}_pmc_${sv}_$win.json $sv "$LABEL" > /dev/null
    cp gpurun_out/prof_${R}_${sv}_$win/summary.json gpurun_out/${R}_${sv}_${win}_summary.json
  done
  cp gpurun_out/${R}_pmc_${sv}_steady.json gpurun_out/${R}_pmc_$sv.json
  python bench.py --solver $sv --pmc-profile gpurun_out/${R}_pmc_$sv.json > gpurun_out/${R}_bench_$sv.json 2> gpurun_out/${R}_bench_$sv.err
  tail -c 1500 gpurun_out/${R}_bench_$sv.json
done
