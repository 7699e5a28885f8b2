"""GPU box: how much of the solver launch is its longest worlds?  Humanoid, 8192 worlds, CG, state at step `at`: the launch is timed as it is, and
with the worlds whose last solve ran more than `cut` iterations replaced by copies of a median world (same batch size, no stragglers)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, importlib
import mujoco_warp_amd as mjw
fw = importlib.import_module("mujoco_warp_amd.forward")
mjm = mjw.mjcf.load_xml(os.path.join(ROOT, "benchmarks", "humanoid", "humanoid.xml"))
mjm.opt.solver = int(mjw.SolverType.CG)
m = mjw.put_model(mjm)
d = mjw.make_data(mjm, nworld=8192, nconmax=24, njmax=64)
mjw.reset_data_keyframe(m, d, 0)
STATE = ("qpos", "qvel", "ctrl", "qacc_warmstart", "time", "solver_niter")
step = 0
for at in (100, 300, 600):
  fw.timed_steps(m, d, at - step, step0=step)
  step = at
  torch.cuda.synchronize()
  keep = {k: getattr(d, k).t.clone() for k in STATE}
  def timed(label):
    res = []
    for rep in range(3):
      for k, t in keep2.items():
        getattr(d, k).t.copy_(t)
      mjw.step(m, d)  # (one untimed step so that solver_niter -- the schedule's predictor -- belongs to this state)
      for k in ("qpos", "qvel", "ctrl", "qacc_warmstart", "time"):
        getattr(d, k).t.copy_(keep2[k])
      _, pk = fw.timed_steps(m, d, 1, step0=step, per_kernel=True)
      res.append(pk[5] * 1e3)
    ni = d.solver_niter.numpy()
    print(f"at {at} {label:28s} solver launch {np.median(res):6.1f} us  niter mean {ni.mean():5.1f} p99 {np.percentile(ni, 99):4.0f} max {ni.max()}", flush=True)
  keep2 = keep
  timed("as it is")
  mjw.step(m, d)
  ni = d.solver_niter.numpy()
  for cut in (40, 30, 24):
    long_w = np.nonzero(ni > cut)[0]
    med = int(np.argsort(ni)[len(ni) // 2])
    keep2 = {k: t.clone() for k, t in keep.items()}
    for k in STATE:
      a = keep2[k]
      a[torch.as_tensor(long_w, device=a.device)] = a[med].clone()
    timed(f"niter > {cut} replaced ({len(long_w)})")
  for k, t in keep.items():
    getattr(d, k).t.copy_(t)
