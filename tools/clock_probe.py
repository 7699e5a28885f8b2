"""GPU box: does the step time depend on how long the GPU has been busy (clock ramp) or on the state of the rollout?
Consecutive windows of 20 fused steps from a cold process; per-launch times of the state-independent `k_fwd_pos` launch tell
the two apart.  python tools/clock_probe.py [nwindows]"""
import os, subprocess, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import mujoco_warp_amd as mjw

NW = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def clocks():
  if os.environ.get("NOSMI"):
    return "(rocm-smi skipped)"
  try:
    o = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
    return " ".join(l.split(":")[-1].strip() for l in o.splitlines() if "sclk" in l or "mclk" in l)
  except Exception as e:
    return str(e)


mjm = mjw.mjcf.load_xml(os.path.join(ROOT, "benchmarks", "humanoid", "humanoid.xml"))
mjw.override_model(mjm, {"opt.solver": "cg"})
m = mjw.put_model(mjm)
d = mjw.make_data(mjm, nworld=8192, nconmax=24, njmax=64)
mjw.reset_data_keyframe(m, d, 0)
print("idle clocks:", clocks())
mjw.timed_steps(m, d, 5, step0=0)
torch.cuda.synchronize()
step = 5
for w in range(NW):
  ms, pk = mjw.timed_steps(m, d, 20, step0=step, per_kernel=True)
  step += 20
  names = mjw.KERNEL_NAMES
  us = {n: round(1e3 * t / 20, 1) for n, t in zip(names, pk) if t > 0}
  print(f"window {w:2d} steps {step - 20:4d}.. ms/step {ms / 20:.4f}", us, "nefc", round(float(np.minimum(d.nefc.numpy(), 64).mean()), 1))
  if w == NW // 2:
    print("  (sleeping 3 s idle)")
    time.sleep(3.0)
print("clocks after:", clocks())
# same state, repeated: reset to the keyframe and run the first 25 steps again (hot GPU)
mjw.reset_data_keyframe(m, d, 0)
mjw.timed_steps(m, d, 5, step0=0)
ms, pk = mjw.timed_steps(m, d, 20, step0=5, per_kernel=True)
print("keyframe again, hot: ms/step", round(ms / 20, 4), {n: round(1e3 * t / 20, 1) for n, t in zip(mjw.KERNEL_NAMES, pk) if t > 0})
