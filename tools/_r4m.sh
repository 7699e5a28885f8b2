#!/bin/bash
mkdir -p gpurun_out
for n in unitree_g1_flat aloha_pot clutter_synth three_humanoids; do
  for lib in "" mujoco_warp_amd/libmjhip_prev.so; do
    [ -n "$lib" ] && export MJH_LIB=$PWD/$lib || unset MJH_LIB
    python tools/diag_state_hash.py $n 256 120 2>&1 | tail -n 1
  done
done
unset MJH_LIB
for r in 1 2; do
for lib in "" mujoco_warp_amd/libmjhip_prev.so; do
  [ -n "$lib" ] && export MJH_LIB=$PWD/$lib || unset MJH_LIB
  python benchmarks/run.py -f "unitree_g1_flat|aloha_pot|clutter_synth$|three_humanoids" 2>&1 | grep steps_per_second | sed "s|^|lib=$lib |"
done
done
unset MJH_LIB
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -n 3
