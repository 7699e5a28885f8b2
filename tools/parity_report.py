"""Per-step parity report (GPU box): HIP float32 step against the float64 oracle from the SAME state, along the oracle's trajectory.

Prints, per model and solver, the worst error over the steps in three metrics:
  max-norm        max|a - b| / max|b|                       (what round 1 asserted)
  per-element     max_i |a_i - b_i| / max(|b_i|, floor)     for floor = 1e-1, 1e-2, 1e-3 (absolute floor in the field's unit)
Run:  python tools/parity_report.py > profiles/round2_parity_report.txt
"""
import os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mujoco_warp_amd as mjw
from oracle import ref


def elem(a, b, floor):
  a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
  return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def mx(a, b):
  a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
  return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-12))


def run(name, xml, solver, nconmax, njmax, nstep=150, iterations=None):
  mjm = mjw.mjcf.load_xml(xml)
  mjm.opt.solver = solver
  if iterations:
    mjm.opt.iterations = iterations
  s = ref.RefSim(mjm, nconmax=nconmax, njmax=njmax, tolerance=max(mjm.opt.tolerance, 1e-6))
  s.reset(key=0 if mjm.nkey else None)
  m = mjw.put_model(mjm)
  d = mjw.put_data(mjm, mjw.MjData(mjm), nworld=2, nconmax=nconmax, njmax=njmax)
  worst = {}
  for i in range(nstep):
    if mjm.nu:
      s.ctrl_noise(i, 0)
    for f in ("qpos", "qvel", "act", "ctrl", "qacc_warmstart"):
      dst = getattr(d, f)
      if dst.size:
        dst.assign(np.tile(getattr(s, f).astype(np.float32), (2, 1)))
    mjw.step(m, d)
    s.step()
    for f in ("qpos", "qvel", "qacc"):
      g, o = getattr(d, f).numpy()[1], getattr(s, f)
      for key, val in (("max", mx(g, o)), ("e-1", elem(g, o, 1e-1)), ("e-2", elem(g, o, 1e-2)), ("e-3", elem(g, o, 1e-3))):
        worst[(f, key)] = max(worst.get((f, key), 0.0), val)
  print(f"{name:28s} solver {['PGS','CG','NEWTON'][solver]:6s}" + "".join(
    f" | {f}: max-norm {worst[(f,'max')]:.1e} elem@1e-1 {worst[(f,'e-1')]:.1e} @1e-2 {worst[(f,'e-2')]:.1e} @1e-3 {worst[(f,'e-3')]:.1e}" for f in ("qpos", "qvel", "qacc")))


if __name__ == "__main__":
  B = os.path.join(ROOT, "benchmarks")
  for solver in (2, 1):
    run("humanoid", os.path.join(B, "humanoid", "humanoid.xml"), solver, 24, 64)
  for solver in (2, 1):
    run("unitree_g1_flat", os.path.join(B, "unitree_g1", "scene_flat.xml"), solver, 48, 192, nstep=60)
  run("franka_emika_panda", os.path.join(B, "franka_emika_panda", "scene.xml"), 2, 8, 16, nstep=60)
