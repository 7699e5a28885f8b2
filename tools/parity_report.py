"""Per-step parity report (GPU box): HIP float32 step against the float64 oracle from the SAME state, along the oracle's trajectory --
and, beside it, the FLOAT32 FLOOR: the same restatement compiled in float32 (oracle/libmjref32.so), restarted from the oracle's state
every step like the engine.  An engine error at or below the twin's error is float32 resolution of the reference's algorithm, not an
implementation difference.

Prints, per model and solver, the worst error over the steps in three metrics:
  max-norm        max|a - b| / max|b|                       (what round 1 asserted)
  per-element     max_i |a_i - b_i| / max(|b_i|, floor)     for floor = 1e-1, 1e-2, 1e-3 (absolute floor in the field's unit)
Steps on which the contact / row counts differ from the oracle's (a contact at float32 resolution of its detection boundary) are
counted, not compared.
Run:  python tools/parity_report.py > profiles/round4_parity_report.txt        (--cpu: the float32 twin alone, no GPU needed)
"""
import os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mujoco_warp_amd as mjw
from oracle import ref

CPU_ONLY = "--cpu" in sys.argv


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
  tol = max(mjm.opt.tolerance, 1e-6)
  s = ref.RefSim(mjm, nconmax=nconmax, njmax=njmax, tolerance=tol)
  s32 = ref.RefSim(mjm, nconmax=nconmax, njmax=njmax, tolerance=tol, real="f32")
  s.reset(key=0 if mjm.nkey else None)
  s32.reset(key=0 if mjm.nkey else None)
  if not CPU_ONLY:
    m = mjw.put_model(mjm)
    d = mjw.put_data(mjm, mjw.MjData(mjm), nworld=2, nconmax=nconmax, njmax=njmax)
  worst, skipped = {"gpu": {}, "f32": {}}, {"gpu": 0, "f32": 0}
  for i in range(nstep):
    if mjm.nu:
      s.ctrl_noise(i, 0)
    for f in ("qpos", "qvel", "act", "ctrl", "qacc_warmstart"):
      if getattr(s, f).size:
        getattr(s32, f)[:] = getattr(s, f)
        if not CPU_ONLY:
          getattr(d, f).assign(np.tile(getattr(s, f).astype(np.float32), (2, 1)))
    if not CPU_ONLY:
      mjw.step(m, d)
    s.step()
    s32.step()
    who = [("f32", lambda f: getattr(s32, f), (s32.ncon, s32.nefc))]
    if not CPU_ONLY:
      who.append(("gpu", lambda f: getattr(d, f).numpy()[1], (int(d.ws_ncon.numpy()[1]), int(d.nefc.numpy()[1]))))
    for tag, get, counts in who:
      if counts != (s.ncon, s.nefc):
        skipped[tag] += 1
        continue
      for f in ("qpos", "qvel", "qacc"):
        g, o = get(f), getattr(s, f)
        for key, val in (("max", mx(g, o)), ("e-1", elem(g, o, 1e-1)), ("e-2", elem(g, o, 1e-2)), ("e-3", elem(g, o, 1e-3))):
          worst[tag][(f, key)] = max(worst[tag].get((f, key), 0.0), val)
  for tag, label in (("gpu", "HIP engine"), ("f32", "float32 twin (CPU floor)")):
    w = worst[tag]
    if not w:
      continue
    print(f"{name:20s} {['PGS','CG','NEWTON'][solver]:6s} {label:24s} skipped {skipped[tag]:3d}/{nstep}" + "".join(
      f" | {f}: max-norm {w[(f,'max')]:.1e} elem@1e-1 {w[(f,'e-1')]:.1e} @1e-2 {w[(f,'e-2')]:.1e} @1e-3 {w[(f,'e-3')]:.1e}" for f in ("qpos", "qvel", "qacc")))


if __name__ == "__main__":
  B = os.path.join(ROOT, "benchmarks")
  for solver in (2, 1):
    run("humanoid", os.path.join(B, "humanoid", "humanoid.xml"), solver, 24, 64)
  for solver in (2, 1):
    run("unitree_g1_flat", os.path.join(B, "unitree_g1", "scene_flat.xml"), solver, 48, 192, nstep=60)
  run("franka_emika_panda", os.path.join(B, "franka_emika_panda", "scene.xml"), 2, 8, 16, nstep=60)
