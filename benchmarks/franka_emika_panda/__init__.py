"""Benchmark registry entry (consumed by benchmarks/run.py)."""

BENCHMARKS = [
  {"name": "franka_emika_panda", "mjcf": "scene.xml", "nworld": 8192, "nconmax": 1, "njmax": 5,
   "note": "BASELINE.json configs[3] shards 8192 worlds over 8 GPUs; this entry is the single-GPU run"},
]
