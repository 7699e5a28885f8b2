#!/bin/bash
# round 5, GPU session B: the rewritten pooled CG kernel (fmac_dpp, one-round-trip J^T f) -- tests, A/B, phase clock, PMC of both kernels
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5c; mkdir -p $O
timeout 900 python -m pytest tests/test_cgp.py tests/test_aloha_pot.py -q -k "cgp or schedule or no_actuation or ccd_flags or per_step_parity_along" -s > $O/tests_cgp.log 2>&1; tail -12 $O/tests_cgp.log
timeout 600 python tools/solve_ab.py --at 5,300 --json $O/ab_main.json "MJH_CG_KERNEL=pair" "MJH_CG_KERNEL=cgp" "MJH_CG_KERNEL=cgp MJH_CGP_THREADS=128" "MJH_CG_KERNEL=cgp MJH_CGP_THREADS=256" > $O/ab_main.log 2>&1; grep "^at" $O/ab_main.log
timeout 600 python tools/phase_clock.py --solver cg --lib mujoco_warp_amd/libmjhip_clkp.so > $O/phase_cgp.txt 2>&1; grep -A11 "^solve" $O/phase_cgp.txt
MJH_CG_KERNEL=cgp timeout 900 bash tools/pmc_solver.sh r5c_cgp --warmup 300 > $O/pmc_cgp.log 2>&1; tail -40 $O/pmc_cgp.log
MJH_CG_KERNEL=pair timeout 900 bash tools/pmc_solver.sh r5c_pair --warmup 300 > $O/pmc_pair.log 2>&1; tail -40 $O/pmc_pair.log
