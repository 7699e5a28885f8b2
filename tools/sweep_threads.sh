#!/bin/bash
for t in 64 128 256; do for s in cg newton; do
  MJH_SOLVE_THREADS=$t python bench.py --steps 60 --warmup 20 --solver $s --no-cpu-baseline 2>/dev/null | python tools/bench_line.py "threads=$t"
done; done
