"""Per-step-synchronised step time, eager launches against a hipGraph replay of the step (what the reference's benchmark loop does, cli.py:262-292):
humanoid, 8192 worlds, CG; windows of 20 steps from step 5 and from step 300, three interleaved repetitions."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import mujoco_warp_amd as mjw
import importlib
fw = importlib.import_module("mujoco_warp_amd.forward")

mjm = mjw.mjcf.load_xml(os.path.join(ROOT, "benchmarks", "humanoid", "humanoid.xml"))
mjm.opt.solver = int(mjw.SolverType.CG)
m = mjw.put_model(mjm)
d = mjw.make_data(mjm, nworld=8192, nconmax=24, njmax=64)
mjw.reset_data_keyframe(m, d, 0)
mjw.step(m, d)
mjw.reset_data_keyframe(m, d, 0)
g = fw.StepGraph(m, d)
STATE = ("qpos", "qvel", "ctrl", "qacc_warmstart", "time", "solver_niter")
step = 0
for at in (5, 300):
  fw.timed_steps(m, d, at - step, step0=step)
  step = at
  torch.cuda.synchronize()
  keep = {k: getattr(d, k).t.clone() for k in STATE}
  res = {"eager": [], "graph": []}
  for rep in range(4):
    for mode in ("eager", "graph"):
      for k, t in keep.items():
        getattr(d, k).t.copy_(t)
      torch.cuda.synchronize()
      total = 0.0
      for i in range(20):
        mjw.ctrl_noise(m, d, at + i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode == "eager":
          mjw.step(m, d)
        else:
          g.launch()
        torch.cuda.synchronize()
        total += time.perf_counter() - t0
      if rep:
        res[mode].append(total / 20 * 1e6)
  for k, t in keep.items():
    getattr(d, k).t.copy_(t)
  print(f"from step {at}: eager {np.median(res['eager']):.1f} us per synced step, graph {np.median(res['graph']):.1f}")
