#!/bin/bash
# run the per-step parity test and a short bench for several builds: tools/abtest.sh lib1.so lib2.so ...
for L in "$@"; do
  echo "== $L"
  MJH_LIB=mujoco_warp_amd/$L python -m pytest tests/test_gpu.py -q -k "per_step_parity" 2>&1 | tail -1
  for S in cg newton; do
    MJH_LIB=mujoco_warp_amd/$L python bench.py --steps 100 --warmup 20 --solver $S --no-cpu-baseline 2>/dev/null | python tools/bench_line.py "[$L]"
  done
done
