#!/bin/bash
# A/B of env knobs within one GPU session: tools/ab.sh "VAR=1" ...   (each arg is an env assignment; "" = baseline)
for rep in 1 2; do for e in "" "$@"; do
  env $e python bench.py --steps 100 --warmup 20 --solver ${MJH_AB_SOLVER:-cg} --no-cpu-baseline --no-roofline 2>/dev/null | python tools/bench_line.py "[$e]"
done; done
