#!/bin/bash
# scratch GPU session (overwritten per use): $1 = tag
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/$1; mkdir -p $O
timeout 600 python tools/config_ab.py unitree_g1_flat 2>&1 | grep -v amdgpu.ids | tee $O/g1.log
timeout 600 python tools/solve_ab.py --at 5,300 --json $O/ab_main.json "MJH_CG_KERNEL=pair" "" > $O/ab_main.log 2>&1; grep "^at" $O/ab_main.log
timeout 3000 python -m pytest tests/ -q -m gpu -x > $O/tests_all.log 2>&1; tail -3 $O/tests_all.log
