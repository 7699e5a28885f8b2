"""Benchmark registry entry (same keys as the reference's benchmarks/*/__init__.py; consumed by benchmarks/run.py)."""

BENCHMARKS = [
  {"name": "humanoid", "mjcf": "humanoid.xml", "nworld": 8192, "nconmax": 24, "njmax": 64},
  {"name": "humanoid_cg", "mjcf": "humanoid.xml", "nworld": 8192, "nconmax": 24, "njmax": 64, "override": ["opt.solver=cg"],
   "note": "BASELINE.json configs[1]"},
  # nv = 81: the generic (LDS) solver path; the model is built with <replicate>/<frame>/<attach> of humanoid.xml
  {"name": "three_humanoids", "mjcf": "three_humanoids.xml", "nworld": 8192, "nconmax": 100, "njmax": 192},
]
