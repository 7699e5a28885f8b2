#!/bin/bash
# like ab.sh but with the per-kernel (serial) timing pass enabled
for rep in 1 2; do for e in "" "$@"; do
  env $e python bench.py --steps 100 --warmup 20 --solver cg --no-cpu-baseline 2>/dev/null | python tools/bench_line.py "[$e]"
done; done
