"""Benchmark registry entry (same keys as the reference's benchmarks/aloha/__init__.py:15-25 `aloha_pot`; consumed by benchmarks/run.py).

The reference's `scene_pot.xml` needs the menagerie ALOHA meshes it fetches over the network (24 of its 133 files are not in the
reference tree).  `scene.xml` here is the same model as the reference holds it for its own tests -- mujoco_warp/test_data/aloha_pot/,
all 134 assets present -- and `lift_pot.npz` is the reference's replay file for the benchmark (benchmarks/aloha/lift_pot.npz).
"""

BENCHMARKS = [
  {"name": "aloha_pot", "mjcf": "scene.xml", "nworld": 8192, "nconmax": 24, "nccdmax": 1, "njmax": 128, "replay": "lift_pot.npz"},
]
