"""The reference's measurement loop (per-step device sync, control untimed) with eager launches and with a hipGraph replay per step.
python tools/sync_loop.py <mjcf> nworld nconmax njmax"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import mujoco_warp_amd as mjw
xml, nworld, nconmax, njmax = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
mjm = mjw.mjcf.load_xml(xml)
m = mjw.put_model(mjm)
d = mjw.make_data(mjm, nworld=nworld, nconmax=nconmax, njmax=njmax)
if mjm.nkey:
  mjw.reset_data_keyframe(m, d, 0)
mjw.timed_steps(m, d, 100)
g = mjw.StepGraph(m, d)
for mode in ("eager", "graph", "eager", "graph"):
  total = 0.0
  for i in range(200):
    mjw.ctrl_noise(m, d, 100 + i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if mode == "eager":
      mjw.step(m, d)
    else:
      g.launch()
    torch.cuda.synchronize()
    total += time.perf_counter() - t0
  print(f"{os.path.basename(xml)} {mode}: {nworld * 200 / total / 1e6:.2f} M env-steps/s  {total / 200 * 1e6:.1f} us/step", flush=True)
