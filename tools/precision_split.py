"""Where does the float32 per-step qvel error come from?  CPU-only experiment with the oracle (float64) and its float32 twin
(oracle/libmjref32.so): one step each from the same state along the oracle's trajectory, as tools/parity_report.py, in variants that
remove one float32 error source at a time:

  A  twin as it is (state copied from the float64 oracle: the twin rounds it to float32, the oracle does not)
  B  the ORACLE's state rounded to float32 storage before its step as well (same inputs bit for bit)
  C  B + the twin's contact distances replaced by the oracle's (float64 FK -> narrowphase, rounded once)
  D  C + the twin's contact positions / frames replaced as well
  E  B + the twin's whole constraint set (J, D, aref, ...) and smooth terms replaced: the float32 SOLVER + integrator alone

usage: python tools/precision_split.py [--solver newton|cg] [--nstep 150]
"""
import argparse, os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mujoco_warp_amd as mjw
from oracle import ref

p = argparse.ArgumentParser()
p.add_argument("--solver", default="newton")
p.add_argument("--nstep", type=int, default=150)
p.add_argument("--variants", default="ABCDE")
p.add_argument("--xml", default=os.path.join(ROOT, "benchmarks", "humanoid", "humanoid.xml"))
a = p.parse_args()

mjm = mjw.mjcf.load_xml(a.xml)
mjm.opt.solver = int({"cg": mjw.SolverType.CG, "newton": mjw.SolverType.NEWTON}[a.solver])
tol = max(mjm.opt.tolerance, 1e-6)
STATE = ("qpos", "qvel", "act", "ctrl", "qacc_warmstart")


def mx(x, y):
  return float(np.max(np.abs(np.asarray(x, np.float64) - y)) / (np.max(np.abs(y)) + 1e-12))


def twin_step(t, s, variant):
  """the float32 twin's step, stage by stage (ref_step of oracle/mjref.c for Euler without sleeping), with the oracle's intermediate
  results spliced in as the variant says; `s` has run ref_forward already"""
  for st in ("kinematics", "com_pos", "crb", "factor_m", "collision"):
    t.stage(st)
  same = t.ncon == s.ncon and (t.con_geom[: t.ncon] == s.con_geom[: s.ncon]).all()
  if variant in "CD" and same:
    t.con_dist[: t.ncon] = s.con_dist[: s.ncon]
    if variant == "D":
      t.con_pos[: t.ncon] = s.con_pos[: s.ncon]
      t.con_frame[: t.ncon] = s.con_frame[: s.ncon]
  if variant in "STUV":  # make_constraint in float64 ARITHMETIC on the twin's float32 inputs (every array upstream copied into a float64 sim)
    if variant in "UV":  # ... and FK -> contacts in float64 as well (from the same float32 state)
      for f in STATE:
        if getattr(t, f).size:
          getattr(s2, f)[:] = getattr(t, f)
      for st in ("kinematics", "com_pos", "crb", "factor_m", "collision"):
        s2.stage(st)
    else:
      for name in t.arr:
        s2.arr[name][:] = t.arr[name]
      s2.cd.ncon = t.cd.ncon
    s2.stage("make_constraint")
  t.stage("make_constraint")
  if variant in "STUV" and s2.nefc == t.nefc:
    for f in (("efc_aref", "efc_D") if variant in "SU" else ("efc_aref", "efc_D", "efc_J")):
      getattr(t, f).reshape(-1)[:] = getattr(s2, f).reshape(-1)
  t.stage("transmission")
  for st in ("fwd_velocity", "fwd_actuation", "fwd_acceleration"):
    t.stage(st)
  if variant in SPLICE and t.nefc == s.nefc:
    for f in SPLICE[variant]:
      getattr(t, f).reshape(-1)[:] = getattr(s, f).reshape(-1)
  t.stage("solve")
  t.stage("euler")
  return same


EFC = ("efc_J", "efc_D", "efc_aref", "efc_pos", "efc_margin", "efc_vel", "efc_frictionloss")
SMOOTH = ("qfrc_smooth", "qacc_smooth", "M", "qLD", "qLDiagInv")
SPLICE = {"E": EFC + SMOOTH, "F": EFC, "G": SMOOTH, "H": ("efc_aref",), "I": ("efc_J",), "J": ("efc_D",), "K": ("qfrc_smooth", "qacc_smooth"), "L": ("M", "qLD", "qLDiagInv"),
          "M": ("efc_aref", "qfrc_smooth", "qacc_smooth"), "N": ("efc_aref", "efc_D", "qfrc_smooth", "qacc_smooth"),
          "O": ("efc_aref", "efc_D", "efc_J"), "P": ("efc_aref", "efc_J"), "Q": ("efc_D", "efc_J"), "R": ("efc_aref", "efc_D")}
for variant in (sys.argv[sys.argv.index("--variants") + 1] if "--variants" in sys.argv else "ABCDE"):
  s = ref.RefSim(mjm, nconmax=24, njmax=64, tolerance=tol)
  t = ref.RefSim(mjm, nconmax=24, njmax=64, tolerance=tol, real="f32")
  s2 = ref.RefSim(mjm, nconmax=24, njmax=64, tolerance=tol)
  s2.reset(key=0)
  s.reset(key=0)
  t.reset(key=0)
  worst = {"qpos": 0.0, "qvel": 0.0, "qacc": 0.0}
  skipped = 0
  hist = []
  for i in range(a.nstep):
    s.ctrl_noise(i, 0)
    if variant != "A":
      for f in STATE:
        if getattr(s, f).size:
          getattr(s, f)[:] = getattr(s, f).astype(np.float32)
    for f in STATE:
      if getattr(s, f).size:
        getattr(t, f)[:] = getattr(s, f)
    s.forward()
    same = twin_step(t, s, variant)
    s.stage("euler")
    if (t.ncon, t.nefc) != (s.ncon, s.nefc):
      skipped += 1
      continue
    for f in worst:
      worst[f] = max(worst[f], mx(getattr(t, f), getattr(s, f)))
    hist.append(mx(t.qvel, s.qvel))
  print(f"{a.solver:6s} variant {variant}: skipped {skipped:3d}/{a.nstep} | " + " | ".join(f"{f} {v:.1e}" for f, v in worst.items())
        + f" | qvel p50 {np.percentile(hist, 50):.1e} p90 {np.percentile(hist, 90):.1e} argmax {int(np.argmax(hist))} n>1e-5 {int((np.array(hist) > 1e-5).sum())}", flush=True)
