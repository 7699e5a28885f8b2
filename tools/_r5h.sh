#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5h; mkdir -p $O
timeout 600 python tools/solve_ab.py --at 5,300 --json $O/ab_main.json "" > $O/ab_main.log 2>&1; grep "^at" $O/ab_main.log
MJH_SCHED_IN_POS=1 timeout 600 python tools/solve_ab.py --at 5,300 --json $O/ab_schedpos.json "" > $O/ab_schedpos.log 2>&1; grep "^at" $O/ab_schedpos.log
timeout 600 python tools/solve_ab.py --at 5,300 --json $O/ab_main2.json "" > $O/ab_main2.log 2>&1; grep "^at" $O/ab_main2.log
timeout 600 python -m pytest tests/test_cgp.py -q -k "schedule" > $O/t.log 2>&1; tail -2 $O/t.log
