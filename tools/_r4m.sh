#!/bin/bash
# scratch driver for one GPU session (round 4)
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r4m_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r4m_tests.log
rm -f gpurun_out/r4m_trace.log
for v in 0; do
  echo "== aloha_pot 600 steps, MJH_GJK_LANES=$v" >> gpurun_out/r4m_trace.log
  [ $v != 0 ] && export MJH_GJK_LANES=$v
  timeout 300 bash tools/trace_lib.sh "" aloha_pot 600 >> gpurun_out/r4m_trace.log 2>&1
  grep steps_per_second gpurun_out/prof_lib/run.log >> gpurun_out/r4m_trace.log
done
unset MJH_GJK_LANES
echo "== clutter_synth" >> gpurun_out/r4m_trace.log
timeout 300 bash tools/trace_lib.sh "" clutter_synth 100 >> gpurun_out/r4m_trace.log 2>&1
grep steps_per_second gpurun_out/prof_lib/run.log >> gpurun_out/r4m_trace.log
timeout 300 python tools/bench_scenes.py 4096 > gpurun_out/r4m_scenes.log 2>&1
tail -n 4 gpurun_out/r4m_tests.log; cat gpurun_out/r4m_trace.log; tail -n 8 gpurun_out/r4m_scenes.log
