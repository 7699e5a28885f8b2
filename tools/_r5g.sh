#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5g; mkdir -p $O
timeout 600 python tools/phase_clock.py --solver cg --lib mujoco_warp_amd/libmjhip_clkm.so > $O/phase_mid.txt 2>&1; grep -A7 "^make_constraint" $O/phase_mid.txt
timeout 600 python tools/solve_ab.py --at 5,300 --json $O/ab_main.json "" > $O/ab_main.log 2>&1; grep "^at" $O/ab_main.log
MJH_LIB=$PWD/mujoco_warp_amd/libmjhip_prev.so timeout 600 python tools/solve_ab.py --at 5,300 --json $O/ab_prev.json "" > $O/ab_prev.log 2>&1; grep "^at" $O/ab_prev.log
timeout 600 python tools/solve_ab.py --at 5,300 --json $O/ab_main2.json "" > $O/ab_main2.log 2>&1; grep "^at" $O/ab_main2.log
timeout 900 python -m pytest tests/test_gpu.py -q -x -k "forward or stage or parity or constraint" > $O/tests_a.log 2>&1; tail -4 $O/tests_a.log
