#!/bin/bash
# scratch driver for one GPU session (round 4)
mkdir -p gpurun_out
python benchmarks/run.py -f "aloha_pot" > gpurun_out/r4m_aloha.log 2>&1
cat gpurun_out/r4m_aloha.log | cut -c1-120
