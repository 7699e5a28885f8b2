"""GPU box: state hash after N steps of a registry scene with its replay / control noise (bitwise comparison of two builds: MJH_LIB=...):
python tools/diag_state_hash.py <registry name> [nworld] [nstep]"""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
import numpy as np
import mujoco_warp_amd as mjw
import run as brun

name = sys.argv[1]
folder, b = [(f, e) for f, e in brun.discover() if e["name"] == name][0]
nworld = int(sys.argv[2]) if len(sys.argv) > 2 else 256
nstep = int(sys.argv[3]) if len(sys.argv) > 3 else 100
mjm = mjw.mjcf.load_xml(os.path.join(folder, b["mjcf"]))
for o in b.get("override", []):
  k, v = o.split("=")
  if k == "opt.solver":
    mjm.opt.solver = int(getattr(mjw.SolverType, v.upper()))
  elif k == "opt.enableflags":
    mjm.opt.enableflags = int(getattr(mjw.EnableBit, v)) if not v.isdigit() else int(v)
m = mjw.put_model(mjm)
mjd = mjw.MjData(mjm)
if mjm.nkey:
  mjw.mj_resetDataKeyframe(mjm, mjd, 0)
ctrls = None
if b.get("replay"):
  ctrls = np.asarray(mjw.load_trajectory(os.path.join(folder, b["replay"]), mjm, mjd), dtype=np.float32)
d = mjw.put_data(mjm, mjd, nworld=nworld, nconmax=b["nconmax"], njmax=b["njmax"])
rng = np.random.RandomState(0)
for i in range(nstep):
  if mjm.nu:
    c = ctrls[min(i, len(ctrls) - 1)] if ctrls is not None else np.asarray(mjd.ctrl, np.float32)
    d.ctrl.assign((c[None, :] + 0.05 * rng.randn(nworld, mjm.nu)).astype(np.float32))
  mjw.step(m, d)
q = d.qpos.numpy()
print(name, os.path.basename(os.environ.get("MJH_LIB", "libmjhip.so")), "qpos sha", hashlib.sha256(q.tobytes()).hexdigest()[:16], "qacc sha", hashlib.sha256(d.qacc.numpy().tobytes()).hexdigest()[:16],
      "nefc mean", float(d.nefc.numpy().mean()), "niter mean", float(d.solver_niter.numpy().mean()), "finite", bool(np.isfinite(q).all()))
