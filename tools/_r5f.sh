#!/bin/bash
# round 5, GPU session F: make_constraint by windows (prepass + basis velocities) -- the GPU suite's constraint / parity tests, A/B against the previous library
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5f; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu.py tests/test_cgp.py tests/test_elliptic.py tests/test_reference_trajectory.py -q -x > $O/tests_a.log 2>&1; tail -8 $O/tests_a.log
timeout 600 python tools/solve_ab.py --at 5,300 --json $O/ab_main.json "" > $O/ab_main.log 2>&1; grep "^at" $O/ab_main.log
MJH_LIB=$PWD/mujoco_warp_amd/libmjhip_prev.so timeout 600 python tools/solve_ab.py --at 5,300 --json $O/ab_prev.json "" > $O/ab_prev.log 2>&1; grep "^at" $O/ab_prev.log
timeout 600 python tools/solve_ab.py --at 5,300 --json $O/ab_main2.json "" > $O/ab_main2.log 2>&1; grep "^at" $O/ab_main2.log
MJH_SCHED_IN_MID=1 timeout 600 python tools/solve_ab.py --at 5,300 --json $O/ab_schedmid.json "" > $O/ab_schedmid.log 2>&1; grep "^at" $O/ab_schedmid.log
