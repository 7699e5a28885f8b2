"""Same-session comparison of eager launches (mjh_timed_steps) and hipGraph replays of one step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mujoco_warp_amd as mjw

solver = sys.argv[1] if len(sys.argv) > 1 else "cg"
mjm = mjw.mjcf.load_xml(os.path.join(ROOT, "benchmarks", "humanoid", "humanoid.xml"))
mjw.override_model(mjm, [f"opt.solver={solver}"])
m = mjw.put_model(mjm)
mjd = mjw.MjData(mjm)
mjw.mj_resetDataKeyframe(mjm, mjd, 0)
d = mjw.put_data(mjm, mjd, nworld=8192, nconmax=24, njmax=64)
mjw.timed_steps(m, d, 100, step0=0)
g = mjw.StepGraph(m, d)
N = 200
for rep in range(2):
  ms, _ = mjw.timed_steps(m, d, N, step0=100)
  print(f"eager  {ms / N * 1e3:8.1f} us/step")
  torch.cuda.synchronize()
  t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
  t0.record()
  for i in range(N):
    mjw.ctrl_noise(m, d, 300 + i)
    g.launch()
  t1.record()
  torch.cuda.synchronize()
  print(f"graph  {t0.elapsed_time(t1) / N * 1e3:8.1f} us/step (python loop: ctrl_noise + graph launch)")
