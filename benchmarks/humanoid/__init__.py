"""Benchmark registry entry (same keys as the reference's benchmarks/*/__init__.py; consumed by benchmarks/run.py)."""

BENCHMARKS = [
  {"name": "humanoid", "mjcf": "humanoid.xml", "nworld": 8192, "nconmax": 24, "njmax": 64},
  {"name": "humanoid_cg", "mjcf": "humanoid.xml", "nworld": 8192, "nconmax": 24, "njmax": 64, "override": ["opt.solver=cg"],
   "note": "BASELINE.json configs[1]"},
]
