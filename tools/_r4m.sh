#!/bin/bash
for n in aloha_pot clutter_synth; do python tools/diag_state_hash.py $n 256 120 2>&1 | tail -n 1; done
python benchmarks/run.py -f "aloha_pot|clutter_synth$" 2>&1 | grep steps_per_second
python tools/bench_scenes.py 4096 2>&1 | grep -v amdgpu | head -n 4
timeout 300 bash tools/trace_lib.sh "" aloha_pot 600 2>&1 | grep -E "k_mid|k_solve"
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -n 2
