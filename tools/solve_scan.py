"""Developer tool (GPU): k_solve time vs iteration cap, and the solver_niter distribution (straggler analysis)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mujoco_warp_amd as mjw

xml = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "benchmarks", "humanoid", "humanoid.xml")
solver = sys.argv[1] if len(sys.argv) > 1 else "cg"
mjm = mjw.mjcf.load_xml(xml)
mjw.override_model(mjm, {"opt.solver": solver})
m = mjw.put_model(mjm)
d = mjw.make_data(mjm, nworld=8192, nconmax=24, njmax=64)
mjw.reset_data_keyframe(m, d, 0)
mjw.timed_steps(m, d, 120)
snap = {k: getattr(d, k).numpy().copy() for k in ("qpos", "qvel", "ctrl", "qacc_warmstart", "time")}
ni = d.solver_niter.numpy()
print("niter mean %.1f p50 %d p90 %d p95 %d p99 %d max %d" % (ni.mean(), *np.percentile(ni, [50, 90, 95, 99]), ni.max()))
print("hist", np.bincount(np.minimum(ni, 100) // 10))
for cap in (1, 2, 5, 10, 20, 40, 100):
  m.opt.iterations = cap
  for k, v in snap.items():
    getattr(d, k).assign(v)
  ms, pk = mjw.timed_steps(m, d, 20, step0=120, per_kernel=True)
  print("cap %3d: solve %.1f us/step  (niter mean %.1f)" % (cap, pk[5] / 20 * 1e3, d.solver_niter.numpy().mean()))
for nw in (256, 1024, 4096):
  m.opt.iterations = 100
  d2 = mjw.make_data(mjm, nworld=nw, nconmax=24, njmax=64)
  for k, v in snap.items():
    getattr(d2, k).assign(v[:nw])
  ms, pk = mjw.timed_steps(m, d2, 20, step0=120, per_kernel=True)
  print("nworld %d: solve %.1f us/step niter max %d" % (nw, pk[5] / 20 * 1e3, d2.solver_niter.numpy().max()))
print("--- single-wave latency (256 worlds = 128 waves on 256 CUs) vs iteration cap")
d2 = mjw.make_data(mjm, nworld=256, nconmax=24, njmax=64)
for cap in (0, 1, 2, 5, 10, 20, 40):
  m.opt.iterations = cap
  for k, v in snap.items():
    getattr(d2, k).assign(v[:256])
  ms, pk = mjw.timed_steps(m, d2, 20, step0=120, per_kernel=True)
  print("cap %3d: solve %.1f us  all kernels %s" % (cap, pk[5] / 20 * 1e3, [round(x / 20 * 1e3, 1) for x in pk[1:7]]))
