"""Solver iteration counts, float32 engine vs float64 oracle, over free-running humanoid worlds (CG by default):
python tools/niter_dist.py [cg|newton] [nworld] [nstep]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mujoco_warp_amd as mjw
from oracle import ref
solver = sys.argv[1] if len(sys.argv) > 1 else "cg"
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 128
nstep = int(sys.argv[3]) if len(sys.argv) > 3 else 300
mjm = mjw.mjcf.load_xml(os.path.join(ROOT, "benchmarks", "humanoid", "humanoid.xml"))
mjw.override_model(mjm, [f"opt.solver={solver}"])
m = mjw.put_model(mjm)
d = mjw.make_data(mjm, nworld=nw, nconmax=24, njmax=64)
mjw.reset_data_keyframe(m, d, 0)
g = []
for i in range(nstep):
  mjw.ctrl_noise(m, d, i)
  mjw.step(m, d)
  if i >= 100:
    g.append(d.solver_niter.numpy().copy())
g = np.concatenate(g)
s = ref.RefSim(mjm, nconmax=24, njmax=64, tolerance=1e-6)
c = []
for w in range(min(nw, 48)):
  s.reset(key=0)
  for i in range(nstep):
    s.ctrl_noise(i, w)
    s.step()
    if i >= 100:
      c.append(s.solver_niter)
c = np.array(c)
for name, a in (("float32 engine", g), ("float64 oracle", c)):
  print(f"{name}: mean {a.mean():.2f} p50 {np.percentile(a, 50):.0f} p95 {np.percentile(a, 95):.0f} p99 {np.percentile(a, 99):.0f} max {a.max()}  (n = {a.size})")
