#!/bin/bash
mkdir -p gpurun_out
for r in 0 1 0 1; do
  MJH_SOLVE64_R1=$r python benchmarks/run.py -f "unitree_g1_flat" 2>&1 | grep -E "steps_per_second" | sed "s|^|R1_64=$r |"
done
MJH_SOLVE64_R1=1 python tools/diag_state_hash.py unitree_g1_flat 256 120 2>&1 | tail -n 1
python tools/diag_state_hash.py unitree_g1_flat 256 120 2>&1 | tail -n 1
