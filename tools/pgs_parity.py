"""PGS per-step parity vs the oracle (developer tool; knobs: MJH_PGS_REFRESH, MJH_PGS_NOREG): python tools/pgs_parity.py [nstep]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mujoco_warp_amd as mjw
from conftest import HUMANOID_XML, relerr
from oracle import ref

nstep = int(sys.argv[1]) if len(sys.argv) > 1 else 120
mjm = mjw.mjcf.load_xml(HUMANOID_XML)
mjm.opt.solver = 0
s = ref.RefSim(mjm, nconmax=24, njmax=64, tolerance=1e-6)
s.reset(key=0)
m = mjw.put_model(mjm)
d = mjw.put_data(mjm, mjw.MjData(mjm), nworld=2, nconmax=24, njmax=64)
wq = wv = wa = 0.0
dn = []
rows = []
for i in range(nstep):
  s.ctrl_noise(i, 0)
  for name in ("qpos", "qvel", "ctrl", "qacc_warmstart"):
    getattr(d, name).assign(np.tile(getattr(s, name).astype(np.float32), (2, 1)))
  mjw.step(m, d)
  s.step()
  wq = max(wq, relerr(d.qpos.numpy()[1], s.qpos))
  wv = max(wv, relerr(d.qvel.numpy()[1], s.qvel))
  wa = max(wa, relerr(d.qacc.numpy()[1], s.qacc))
  dn.append(int(d.solver_niter.numpy()[1]) - s.solver_niter)
  rows.append((relerr(d.qacc.numpy()[1], s.qacc), i, int(d.solver_niter.numpy()[1]), s.solver_niter, s.nefc, int(d.nefc.numpy()[1]),
               relerr(d.efc.force.numpy()[1, :s.nefc], s.efc_force[:s.nefc]), relerr(d.qacc_smooth.numpy()[1], s.qacc_smooth)))
print(f"refresh={os.environ.get('MJH_PGS_REFRESH', '1')} noreg={os.environ.get('MJH_PGS_NOREG')}: worst qpos {wq:.2e} qvel {wv:.2e} qacc {wa:.2e} niter diff min/max {min(dn)}/{max(dn)}")
for r in sorted(rows, reverse=True)[:8]:
  print("  qacc err %.2e step %d niter gpu %d oracle %d nefc %d gpu nefc %d force err %.2e qacc_smooth err %.2e" % r)
