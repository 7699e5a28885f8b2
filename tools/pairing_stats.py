"""Wave pairing quality of the solver schedule: a wavefront hosts two worlds and runs max(niter) iterations."""
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
mjw.timed_steps(m, d, 150)
prev = d.solver_niter.numpy().copy()
prev_nefc = d.nefc.numpy().copy()
rows = []
for i in range(20):
  mjw.timed_steps(m, d, 1, step0=150 + i)
  cur = d.solver_niter.numpy().copy()
  order = np.argsort(-prev, kind="stable")          # what k_schedule_worlds produces (up to ties)
  pm = lambda o: np.maximum(cur[o[0::2]], cur[o[1::2]]).mean()
  ideal = np.argsort(-cur, kind="stable")
  rnd = np.random.default_rng(i).permutation(len(cur))
  rows.append((cur.mean(), pm(order), pm(ideal), pm(rnd), np.corrcoef(prev, cur)[0, 1], np.corrcoef(d.nefc.numpy(), cur)[0, 1]))
  prev = cur
r = np.array(rows).mean(axis=0)
print(f"mean niter {r[0]:.2f} | E[max of a wave]: scheduled {r[1]:.2f}, ideal {r[2]:.2f}, random {r[3]:.2f} | corr(prev,cur) {r[4]:.2f} corr(nefc,cur) {r[5]:.2f}")
