import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mujoco_warp_amd as mjw
mjm = mjw.mjcf.load_xml(os.path.join(ROOT, "benchmarks", "humanoid", "humanoid.xml"))
mjw.override_model(mjm, ["opt.solver=cg"])
m = mjw.put_model(mjm)
d = mjw.make_data(mjm, nworld=8192, nconmax=24, njmax=64)
mjw.reset_data_keyframe(m, d, 0)
mjw.timed_steps(m, d, 120)
H = []
for i in range(60):
  mjw.timed_steps(m, d, 1, step0=120 + i)
  H.append(d.solver_niter.numpy().copy())
H = np.array(H, dtype=np.float64)
def emax(pred, cur):
  o = np.argsort(-pred, kind="stable")
  return np.maximum(cur[o[0::2]], cur[o[1::2]]).mean()
print("mean", H[20:].mean(), "ideal", np.mean([emax(H[t], H[t]) for t in range(20, 60)]))
for a in (1.0, 0.7, 0.5, 0.35, 0.25, 0.15):
  ema = H[0].copy(); vals = []
  for t in range(1, 60):
    if t >= 20: vals.append(emax(ema, H[t]))
    ema = a * H[t] + (1 - a) * ema
  print(f"ema alpha {a}: E[max] {np.mean(vals):.3f}")
# two-step max and mean of last 2/3
for k in (2, 3, 4):
  vals = [emax(H[t-k:t].mean(axis=0), H[t]) for t in range(20, 60)]
  vals2 = [emax(H[t-k:t].max(axis=0), H[t]) for t in range(20, 60)]
  print(f"mean of last {k}: {np.mean(vals):.3f}   max of last {k}: {np.mean(vals2):.3f}")
