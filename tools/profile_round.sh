#!/bin/bash
# GPU box: kernel-trace + PMC summaries of the headline bench for both solvers, plus the bench lines themselves.
# usage: tools/profile_round.sh <round tag, e.g. round2>
set -u
R=${1:-round2}
for sv in newton cg; do
  bash tools/profile.sh ${R}_$sv --solver $sv > /dev/null 2>&1
  python tools/make_pmc_summary.py gpurun_out/prof_${R}_$sv/summary.json gpurun_out/${R}_pmc_$sv.json $sv > /dev/null
  cp gpurun_out/prof_${R}_$sv/summary.json gpurun_out/${R}_${sv}_summary.json
  python bench.py --solver $sv --pmc-profile gpurun_out/${R}_pmc_$sv.json > gpurun_out/${R}_bench_$sv.json 2> gpurun_out/${R}_bench_$sv.err
  tail -c 1500 gpurun_out/${R}_bench_$sv.json
done
