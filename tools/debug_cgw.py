"""Pieces of the one-world-per-wavefront CG kernel against numpy: iterations = 0 leaves qfrc_constraint = J^T f(q), efc_Ma = M q,
efc_force = f(J q - aref) at the warm-start point q."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mujoco_warp_amd as mjw

mjm = mjw.mjcf.load_xml(os.path.join(ROOT, "benchmarks", "humanoid", "humanoid.xml"))
mjw.override_model(mjm, ["opt.solver=cg"])
m = mjw.put_model(mjm)
d = mjw.make_data(mjm, nworld=4, nconmax=24, njmax=64)
mjw.reset_data_keyframe(m, d, 0)
for _ in range(30):
  mjw.step(m, d)
snap = {k: getattr(d, k).numpy().copy() for k in ("qpos", "qvel", "qacc_warmstart")}
mjw.override_model(mjm, ["opt.iterations=0"])
m0 = mjw.put_model(mjm)
mjw.forward(m0, d)
w = 1
nv = mjm.nv
nefc = int(d.nefc.numpy()[w])
J = d.efc.J.numpy()[w].reshape(-1, d.nv_pad)[:nefc, :nv].astype(np.float64)
f = d.efc.force.numpy()[w][:nefc].astype(np.float64)
q = snap["qacc_warmstart"][w].astype(np.float64)
M = np.zeros((nv, nv))
Ms = d.M.numpy()[w]
md = m.M_dense.numpy().reshape(nv, -1)[:, :nv]
for i in range(nv):
  for j in range(nv):
    if md[i, j] >= 0:
      M[i, j] = Ms[md[i, j]]
aref = d.efc.aref.numpy()[w][:nefc]
D = d.efc.D.numpy()[w][:nefc]
print("nefc", nefc, "qacc == warmstart", np.abs(d.qacc.numpy()[w] - q).max())
print("efc_Ma vs M q        ", np.abs(d.efc.Ma.numpy()[w] - M @ q).max(), np.abs(M @ q).max())
ja = J @ q - aref
ne, nf = int(d.ne.numpy()[w]), int(d.nf.numpy()[w])
fexp = np.where((np.arange(nefc) < ne) | (ja < 0), -D * ja, 0.0)
print("efc_force vs f(Jq-a) ", np.abs(f - fexp).max(), np.abs(fexp).max(), "rows", np.nonzero(np.abs(f - fexp) > 1e-3 * np.abs(fexp).max())[0])
qc = d.qfrc_constraint.numpy()[w]
print("qfrc_constraint vs J'f", np.abs(qc - J.T @ f).max(), np.abs(J.T @ f).max(), "dofs", np.nonzero(np.abs(qc - J.T @ f) > 1e-3 * np.abs(J.T @ f).max())[0])
Minv = np.linalg.inv(M)
print("qacc_smooth vs M^-1 fs", np.abs(d.qacc_smooth.numpy()[w] - Minv @ d.qfrc_smooth.numpy()[w]).max(), np.abs(d.qacc_smooth.numpy()[w]).max())
np.set_printoptions(linewidth=250, precision=4, suppress=True)
for r in range(nefc):
  nzc = np.nonzero(J[r])[0]
  print(r, "f", f[r], "exp", fexp[r], "ja", ja[r], "ja_kernel", -f[r] / D[r] if f[r] != 0 else None, "cols", nzc.min(), nzc.max(), len(nzc))
