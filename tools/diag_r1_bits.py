"""GPU box: state hash after N steps of a registry scene (bitwise comparison of solver instantiations: MJH_SOLVE_R1=0 / 1):
python tools/diag_r1_bits.py humanoid|aloha_pot [nworld] [nstep]"""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mujoco_warp_amd as mjw

name = sys.argv[1]
nworld = int(sys.argv[2]) if len(sys.argv) > 2 else 512
nstep = int(sys.argv[3]) if len(sys.argv) > 3 else 100
if name == "aloha_pot":
  mjm = mjw.mjcf.load_xml(os.path.join(ROOT, "benchmarks", "aloha_pot", "scene.xml"))
  kw = dict(nconmax=24, njmax=128)
else:
  mjm = mjw.mjcf.load_xml(os.path.join(ROOT, "benchmarks", "humanoid", "humanoid.xml"))
  mjm.opt.solver = int(mjw.SolverType.CG)
  kw = dict(nconmax=24, njmax=64)
m = mjw.put_model(mjm)
mjd = mjw.MjData(mjm)
if mjm.nkey:
  mjw.mj_resetDataKeyframe(mjm, mjd, 0)
d = mjw.put_data(mjm, mjd, nworld=nworld, **kw)
rng = np.random.RandomState(0)
for i in range(nstep):
  if mjm.nu:
    d.ctrl.assign((0.3 * rng.randn(nworld, mjm.nu)).astype(np.float32) + (np.asarray(mjd.ctrl, np.float32) if name == "aloha_pot" else 0))
  mjw.step(m, d)
q = d.qpos.numpy()
print(name, "R1 =", os.environ.get("MJH_SOLVE_R1", "default"), "qpos sha", hashlib.sha256(q.tobytes()).hexdigest()[:16], "nefc mean", d.nefc.numpy().mean(), "finite", bool(np.isfinite(q).all()))
