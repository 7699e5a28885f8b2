import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import mujoco_warp_amd as mjw
from oracle import ref
mjm = mjw.mjcf.load_xml("tests/models/clutter_synth.xml")
s = ref.RefSim(mjm, nconmax=256, njmax=384)
s.reset(key=0)
m = mjw.put_model(mjm)
d = mjw.make_data(mjm, nworld=2, nconmax=256, njmax=384)
mjw.reset_data_keyframe(m, d, 0)
rows = []
for i in range(260):
  for name in ("qpos", "qvel", "qacc_warmstart", "ctrl"):
    getattr(d, name).assign(np.tile(getattr(s, name).astype(np.float32), (2, 1)))
  d.tree_asleep.assign(np.tile(s.tree_asleep, (2, 1)))
  mjw.update_sleep(m, d)
  mjw.step(m, d)
  s.step()
  if int(d.ws_ncon.numpy()[1]) != s.ncon or int(d.nefc.numpy()[1]) != s.nefc:
    continue
  eq = np.abs(d.qpos.numpy()[1] - s.qpos); ev = np.abs(d.qvel.numpy()[1] - s.qvel)
  rows.append((i, eq.max(), int(eq.argmax()), ev.max(), int(ev.argmax()), np.abs(s.qvel).max(), s.ncon, s.nefc, int(d.solver_niter.numpy()[1]), s.solver_niter, int(s.ntree_awake)))
rows.sort(key=lambda r: -r[3])
for r in rows[:12]:
  print("step %3d qpos err %.3g (q%d) qvel err %.3g (dof %d) max|qvel| %.3g ncon %d nefc %d niter %d/%d awake %d" % r)
