#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5i; mkdir -p $O
timeout 900 python -m pytest tests/test_cgp.py tests/test_gpu.py -q -x -k "cgp or cg or CG or parity" > $O/t.log 2>&1; tail -3 $O/t.log
timeout 600 python tools/solve_ab.py --at 5,300 --json $O/ab_main.json "MJH_CG_KERNEL=pair" "" > $O/ab_main.log 2>&1; grep "^at" $O/ab_main.log
