#!/bin/bash
# GPU session: bench lines and scene traces of the final build
mkdir -p gpurun_out
for sv in cg newton; do
  python bench.py --solver $sv --pmc-profile profiles/round4_pmc_$sv.json > gpurun_out/round4_bench_$sv.json 2> gpurun_out/round4_bench_$sv.err
done
rm -f gpurun_out/round4_scene_traces.txt
for f in aloha_pot clutter_synth; do
  echo "== python benchmarks/run.py -f $f (kernel trace)" >> gpurun_out/round4_scene_traces.txt
  n=1000; [ $f = clutter_synth ] && n=300
  timeout 300 bash tools/trace_lib.sh "" $f $n >> gpurun_out/round4_scene_traces.txt 2>&1
  grep -E "steps_per_second|nefc_mean|ncon_mean|solver_niter_mean" gpurun_out/prof_lib/run.log >> gpurun_out/round4_scene_traces.txt
done
timeout 300 python tools/bench_scenes.py 4096 > gpurun_out/round4_feature_scenes_plain.txt 2>&1
head -n 14 gpurun_out/round4_scene_traces.txt
