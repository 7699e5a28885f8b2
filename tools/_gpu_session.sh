#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/$1; mkdir -p $O
for L in "" mid5 mid4w8 ""; do
  if [ -n "$L" ]; then export MJH_LIB=$PWD/mujoco_warp_amd/libmjhip_$L.so; else unset MJH_LIB; fi
  echo "== lib $L"; MJH_DEBUG_OCC=1 timeout 600 python tools/solve_ab.py --at 5,300 "" 2>&1 | grep "^at\|k_mid " | sort | uniq | head -4
done
