#!/bin/bash
timeout 300 python tools/phase_clock.py --solver cg --lib mujoco_warp_amd/libmjhip_clkm.so 2>&1 | grep -v amdgpu | grep -A7 -E "^cg:|^make_constraint:"
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -n 5
python bench.py --no-configs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', j['value'], 'steady', j['steady_1000']['value'], 'fused', j['fused_launch_us'], 'box', j['box'].get('slow_box'))"
