#!/bin/bash
# scratch driver for one GPU session (round 4): tests, then broadphase / hill-climb ablations on aloha_pot
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r4m_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r4m_tests.log
for v in "" nogpos noa nob skip; do
  lib=""; [ -n "$v" ] && lib=mujoco_warp_amd/libmjhip_$v.so
  echo "== variant '$v'" >> gpurun_out/r4m_trace.log
  timeout 300 bash tools/trace_lib.sh "$lib" aloha_pot 60 >> gpurun_out/r4m_trace.log 2>&1
  tail -n 3 gpurun_out/prof_lib/run.log >> gpurun_out/r4m_trace.log
done
timeout 300 bash tools/trace_lib.sh "" clutter_synth 60 >> gpurun_out/r4m_trace.log 2>&1
timeout 300 python tools/bench_scenes.py 4096 > gpurun_out/r4m_scenes.log 2>&1
tail -n 30 gpurun_out/r4m_tests.log; cat gpurun_out/r4m_trace.log; tail -n 12 gpurun_out/r4m_scenes.log
