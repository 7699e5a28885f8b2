#!/bin/bash
mkdir -p gpurun_out
for n in aloha_pot clutter_synth; do python tools/diag_state_hash.py $n 256 120 2>&1 | tail -n 1; done
python benchmarks/run.py -f "aloha_pot|clutter_synth$" 2>&1 | grep steps_per_second
timeout 300 bash tools/trace_lib.sh "" aloha_pot 600 2>&1 | head -n 10
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -n 3
