#!/bin/bash
# round 5, GPU session A: the pooled CG kernel -- parity tests, A/B against k_solve<cg>, workgroup sizes, 3 vs 4 wavefronts per SIMD, phase clock
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5a; mkdir -p $O
python -c "import mujoco_warp_amd._abi as a; print('needs_build', a.needs_build())" > $O/build.log 2>&1
timeout 900 python -m pytest tests/test_cgp.py -x -q > $O/tests_cgp.log 2>&1; tail -5 $O/tests_cgp.log
timeout 600 python tools/solve_ab.py --at 5,300 --json $O/ab_main.json "MJH_CG_KERNEL=pair" "MJH_CG_KERNEL=cgp" "MJH_CG_KERNEL=cgp MJH_CGP_THREADS=384" "MJH_CG_KERNEL=cgp MJH_CGP_THREADS=768" "MJH_CG_KERNEL=cgp MJH_CGP_THREADS=64" > $O/ab_main.log 2>&1; cat $O/ab_main.log | grep "^at"
MJH_CGP_NO_FALLBACK=1 timeout 600 python tools/solve_ab.py --at 5,300 --json $O/ab_nofb.json "MJH_CG_KERNEL=cgp" > $O/ab_nofb.log 2>&1; grep "^at" $O/ab_nofb.log
MJH_SCHED_IN_MID=1 timeout 600 python tools/solve_ab.py --at 5,300 --json $O/ab_schedmid.json "MJH_CG_KERNEL=cgp" "MJH_CG_KERNEL=pair" > $O/ab_schedmid.log 2>&1; grep "^at" $O/ab_schedmid.log
MJH_LIB=$PWD/mujoco_warp_amd/libmjhip_w4.so timeout 600 python tools/solve_ab.py --at 5,300 --json $O/ab_w4.json "MJH_CG_KERNEL=cgp MJH_CGP_THREADS=256" "MJH_CG_KERNEL=cgp MJH_CGP_THREADS=512" "MJH_CG_KERNEL=cgp MJH_CGP_THREADS=1024" > $O/ab_w4.log 2>&1; grep "^at" $O/ab_w4.log
timeout 600 python tools/phase_clock.py --solver cg --lib mujoco_warp_amd/libmjhip_clkp.so > $O/phase_cgp.txt 2>&1; cat $O/phase_cgp.txt | head -14
timeout 900 python -m pytest tests/test_gpu.py -x -q -k "cg or CG or graph or shard" > $O/tests_gpu_cg.log 2>&1; tail -3 $O/tests_gpu_cg.log
