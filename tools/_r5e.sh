#!/bin/bash
# round 5, GPU session D: cgp with batched J^T f loads -- tests, A/B, phase clock
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5e; mkdir -p $O
timeout 900 python -m pytest tests/test_cgp.py tests/test_aloha_pot.py -q -k "cgp or schedule or no_actuation or ccd_flags or per_step_parity_along" -s > $O/tests_cgp.log 2>&1; tail -12 $O/tests_cgp.log
timeout 600 python tools/solve_ab.py --at 5,300 --json $O/ab_main.json "MJH_CG_KERNEL=pair" "MJH_CG_KERNEL=cgp" "MJH_CG_KERNEL=cgp MJH_CGP_THREADS=64" > $O/ab_main.log 2>&1; grep "^at" $O/ab_main.log
timeout 600 python tools/phase_clock.py --solver cg --lib mujoco_warp_amd/libmjhip_clkp.so > $O/phase_cgp.txt 2>&1; grep -A11 "^solve" $O/phase_cgp.txt
MJH_CG_KERNEL=cgp timeout 900 bash tools/pmc_solver.sh r5e_cgp --warmup 300 > $O/pmc_cgp.log 2>&1; python - <<'PY'
import json
a=json.load(open('gpurun_out/pmcs_r5e_cgp/solver.json'))
for k,v in a.items(): print(k, {q:round(x) for q,x in v.items() if q in ('mean_us','SQ_INSTS_VALU','SQ_INSTS_LDS','SQ_INSTS_SALU','SQ_WAIT_ANY','SQ_WAIT_INST_ANY','SQ_WAVE_CYCLES','SQ_BUSY_CYCLES','SQ_LDS_IDX_ACTIVE','SQ_LDS_BANK_CONFLICT')})
PY
