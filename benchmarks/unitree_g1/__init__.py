"""Benchmark registry entry (consumed by benchmarks/run.py)."""

BENCHMARKS = [
  {"name": "unitree_g1_flat", "mjcf": "scene_flat.xml", "nworld": 4096, "nconmax": 48, "njmax": 192, "replay": "shuffle_dance.npz",
   "note": "BASELINE.json configs[2] (the reference's own default is 8192 worlds)"},
]
