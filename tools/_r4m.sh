#!/bin/bash
mkdir -p gpurun_out
for r1 in 1 0 1 0; do
  MJH_SOLVE_R1=$r1 python benchmarks/run.py -f "aloha_pot" 2>&1 | grep steps_per_second | sed "s/^/R1=$r1 /"
done
for r1 in 1 0; do
MJH_SOLVE_R1=$r1 python benchmarks/run.py -f "clutter_synth$" 2>&1 | grep steps_per_second | sed "s/^/R1=$r1 /"
done
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -n 3
