"""aloha_pot, 8192 worlds, the bench line's replay (lift_pot.npz as the control centre, per-step sync): run N steps -- meant to be run under
`rocprofv3 --kernel-trace`, whose last dispatches tools/summarize_profile.py timeline() then lists (which launches a step consists of, how long
each takes, where the gaps are)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import mujoco_warp_amd as mjw

B = os.path.join(ROOT, "benchmarks", "aloha_pot")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
mjm = mjw.mjcf.load_xml(os.path.join(B, "scene.xml"))
m = mjw.put_model(mjm)
mjd = mjw.MjData(mjm)
mjw.mj_resetDataKeyframe(mjm, mjd, 0)
ctrl = mjw.load_trajectory(os.path.join(B, "lift_pot.npz"), mjm, mjd)
center = [mjw.DeviceArray.from_numpy(np.asarray(c, dtype=np.float32)) for c in ctrl[:n]]
d = mjw.put_data(mjm, mjd, nworld=8192, nconmax=24, njmax=128)
import time
total = 0.0
for i in range(n):
  mjw.ctrl_noise(m, d, i, center=center[min(i, len(center) - 1)])
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  mjw.step(m, d)
  torch.cuda.synchronize()
  if i >= n // 2:
    total += time.perf_counter() - t0
print("%.1f us per step (second half of %d steps, per-step sync) = %.2f M env-steps/s" % (1e6 * total / (n - n // 2), n, 8192 * (n - n // 2) / total / 1e6))
nefc = np.minimum(d.nefc.numpy(), d.njmax)
print("nefc mean %.1f  >32: %d  >64: %d of %d worlds; niter %.2f" % (nefc.mean(), (nefc > 32).sum(), (nefc > 64).sum(), d.nworld, d.solver_niter.numpy().mean()))
