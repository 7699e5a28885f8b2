#!/bin/bash
# GPU box: interleaved A/B of the default library against mujoco_warp_amd/libmjhip_var.so (a variant linked from the object cache with one unit
# recompiled) on the headline workload: solver launch (us), ms per step and the longest solve per window.  usage: tools/ab_lib.sh [nworld] [windows]
NW=${1:-8192}; AT=${2:-5,300,600}
for rep in 1 2 3; do for lib in "" "MJH_LIB=mujoco_warp_amd/libmjhip_var.so"; do
  env $lib timeout 300 python tools/solve_ab.py --nworld $NW --at $AT --steps 20 --reps 2 "" < /dev/null 2>&1 | grep "^at" | sed "s#^#[${lib:-base}] #"
done; done | python -c "
import re,json,sys
rows={}
for line in sys.stdin:
    m=re.match(r'\[(.*?)\] at\s+(\d+) \[.*?\] (\{.*\})',line)
    if m:
        d=json.loads(m.group(3)); rows.setdefault((m.group(1)[:8],int(m.group(2))),[]).append((d['solve'],d['ms_per_step'],d['niter_max']))
for k,v in sorted(rows.items(), key=lambda kv:(kv[0][1],kv[0][0])): print(k, v)
"
