#!/bin/bash
mkdir -p gpurun_out
(echo "# python tools/phase_clock.py --solver cg --lib mujoco_warp_amd/libmjhip_clk32.so   (solve_cg32.hip compiled with -DMJH_PHASE_CLOCK; humanoid, 8192 worlds,"
 echo "# 50 steps after 100; shader-clock ticks of lane 0 of every world between the marks of solve_body, summed and divided by worlds x steps)"
 timeout 300 python tools/phase_clock.py --solver cg --lib mujoco_warp_amd/libmjhip_clk32.so 2>&1 | grep -v amdgpu | grep -A18 -E "^cg:|^solve:") > gpurun_out/round4_phase_cg.txt
(echo "# python tools/phase_clock.py --solver newton --lib mujoco_warp_amd/libmjhip_clkn32.so   (solve_newton32.hip with -DMJH_PHASE_CLOCK)"
 timeout 300 python tools/phase_clock.py --solver newton --lib mujoco_warp_amd/libmjhip_clkn32.so 2>&1 | grep -v amdgpu | grep -A18 -E "^newton:|^solve:") > gpurun_out/round4_phase_newton.txt
cat gpurun_out/round4_phase_cg.txt gpurun_out/round4_phase_newton.txt
