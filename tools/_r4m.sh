#!/bin/bash
# GPU session: the round's final measurements (tests, profiles of both solvers on both windows, bench lines, feature scenes, scene traces, parity report)
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/round4_gpu_tests.log 2>&1
tail -n 3 gpurun_out/round4_gpu_tests.log
timeout 1500 bash tools/profile_round.sh round4 "cg newton" > gpurun_out/round4_profile_round.log 2>&1
timeout 300 bash tools/profile_scenes.sh round4 4096 > gpurun_out/round4_feature_scenes.txt 2>&1
for f in aloha_pot clutter_synth; do
  echo "== python benchmarks/run.py -f $f (kernel trace, full replay / 1000 steps)" >> gpurun_out/round4_scene_traces.txt
  n=1000; [ $f = clutter_synth ] && n=300
  timeout 300 bash tools/trace_lib.sh "" $f $n >> gpurun_out/round4_scene_traces.txt 2>&1
  grep -E "steps_per_second|nefc_mean|ncon_mean|solver_niter_mean" gpurun_out/prof_lib/run.log >> gpurun_out/round4_scene_traces.txt
done
timeout 600 python tools/parity_report.py > gpurun_out/round4_parity_report.txt 2>&1
tail -c 600 gpurun_out/round4_bench_cg.json; echo; cat gpurun_out/round4_scene_traces.txt | head -n 40
