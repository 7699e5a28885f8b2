#!/bin/bash
# GPU session: final bench lines of both solvers (PMC summaries: the committed round4 passes of the unchanged headline kernels), scene traces, feature scenes, tests
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/round4_gpu_tests.log 2>&1
tail -n 2 gpurun_out/round4_gpu_tests.log
for sv in cg newton; do
  python bench.py --solver $sv --pmc-profile profiles/round4_pmc_$sv.json > gpurun_out/round4_bench_$sv.json 2> gpurun_out/round4_bench_$sv.err
  tail -c 300 gpurun_out/round4_bench_$sv.json; echo
done
python bench.py --steps 20 --warmup 5 > gpurun_out/round4_bench_driver_flags.json 2> gpurun_out/round4_bench_driver_flags.err
rm -f gpurun_out/round4_scene_traces.txt
for f in aloha_pot clutter_synth; do
  echo "== python benchmarks/run.py -f $f (kernel trace)" >> gpurun_out/round4_scene_traces.txt
  n=1000; [ $f = clutter_synth ] && n=300
  timeout 300 bash tools/trace_lib.sh "" $f $n >> gpurun_out/round4_scene_traces.txt 2>&1
  grep -E "steps_per_second|nefc_mean|ncon_mean|solver_niter_mean" gpurun_out/prof_lib/run.log >> gpurun_out/round4_scene_traces.txt
done
timeout 300 bash tools/profile_scenes.sh round4 4096 > gpurun_out/round4_feature_scenes_trace.txt 2>&1
timeout 300 python tools/bench_scenes.py 4096 > gpurun_out/round4_feature_scenes_plain.txt 2>&1
cat gpurun_out/round4_scene_traces.txt | head -n 16
