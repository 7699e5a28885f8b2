"""Timing of the nv = 81 three-humanoid model (generic solver, csrc/solver_big.hpp): python tools/bench_three_humanoids.py [nworld]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import conftest
import mujoco_warp_amd as mjw

nworld = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
mjm = mjw.mjcf.from_xml_string(conftest.multi_humanoid_xml(3), assets_dir=os.path.join(ROOT, "benchmarks", "humanoid"))
for solver in ("newton", "cg"):
  mjw.override_model(mjm, {"opt.solver": solver})
  m = mjw.put_model(mjm)
  d = mjw.make_data(mjm, nworld=nworld, nconmax=100, njmax=192)
  mjw.timed_steps(m, d, 60)
  ms, pk = mjw.timed_steps(m, d, 50, step0=60, per_kernel=True)
  print(f"three humanoids (nv 81) {solver}: {nworld * 50 / ms * 1e3:,.0f} env-steps/s", {k: round(v * 20, 1) for k, v in zip(mjw.KERNEL_NAMES, pk) if v > 0},
        "niter", d.solver_niter.numpy().mean(), "nefc", d.nefc.numpy().mean(), bool(np.isfinite(d.qpos.numpy()).all()))
