#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/phase_clock.py --solver newton --lib mujoco_warp_amd/libmjhip_clkn64.so --xml benchmarks/unitree_g1/scene_flat.xml --nworld 4096 --nconmax 48 --njmax 192 2>&1 | grep -v amdgpu | grep -A18 -E "^newton:|^solve:" | grep -v ": 0 ticks" | grep -v "    0    0.0%" | tail -n 16
for r in 1 2; do
for lib in "" mujoco_warp_amd/libmjhip_prev.so; do
  [ -n "$lib" ] && export MJH_LIB=$PWD/$lib || unset MJH_LIB
  python benchmarks/run.py -f "unitree_g1_flat|three_humanoids|clutter_synth$" 2>&1 | grep steps_per_second | sed "s|^|lib=$lib |"
done
done
unset MJH_LIB
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -n 3
