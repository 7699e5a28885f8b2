#!/bin/bash
mkdir -p gpurun_out
for n in aloha_pot clutter_synth; do python tools/diag_state_hash.py $n 256 120 2>&1 | tail -n 1; done
for v in "" epaw2 "" epaw2; do
  lib=""; [ -n "$v" ] && lib=mujoco_warp_amd/libmjhip_$v.so
  [ -n "$lib" ] && export MJH_LIB=$PWD/$lib || unset MJH_LIB
  python benchmarks/run.py -f "aloha_pot|clutter_synth$" 2>&1 | grep steps_per_second | sed "s|^|lib=$v |"
done
unset MJH_LIB
timeout 300 bash tools/trace_lib.sh "" aloha_pot 600 2>&1 | grep -E "epa|gjk"
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -n 3
