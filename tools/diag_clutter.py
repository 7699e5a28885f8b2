import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import mujoco_warp_amd as mjw
from oracle import ref
from tests.conftest import relerr
solver = sys.argv[1] if len(sys.argv) > 1 else "newton"
mjm = mjw.mjcf.load_xml("tests/models/clutter_synth.xml")
mjm.opt.enableflags = 0
mjw.override_model(mjm, {"opt.solver": solver})
s = ref.RefSim(mjm, nconmax=256, njmax=384, tolerance=1e-6)
s.reset(key=0)
m = mjw.put_model(mjm)
d = mjw.make_data(mjm, nworld=2, nconmax=256, njmax=384)
for i in range(130):
  s.step()
  if i in (80, 100, 128):
    for name in ("qpos", "qvel", "qacc_warmstart", "ctrl"):
      getattr(d, name).assign(np.tile(getattr(s, name).astype(np.float32), (2, 1)))
    mjw.forward(m, d); s.forward()
    n = s.nefc
    print("step", i, "nefc", int(d.nefc.numpy()[1]), n, "ncon", int(d.ws_ncon.numpy()[1]), s.ncon, "niter", int(d.solver_niter.numpy()[1]), s.solver_niter, "nisland", int(d.ws_nisland.numpy()[1]), "sep", int(d.ws_separable.numpy()[1]))
    J = d.efc.J.numpy()[1][:n, :136]
    print("  J", relerr(J, s.efc_J[:n]), "D", relerr(d.efc.D.numpy()[1][:n], s.efc_D[:n]), "aref", relerr(d.efc.aref.numpy()[1][:n], s.efc_aref[:n]),
          "qacc_smooth", relerr(d.qacc_smooth.numpy()[1], s.qacc_smooth), "qfrc_smooth", relerr(d.qfrc_smooth.numpy()[1], s.qfrc_smooth))
    e = np.abs(d.qacc.numpy()[1] - s.qacc)
    print("  qacc err max", e.max(), "at dof", e.argmax(), "|qacc| max", np.abs(s.qacc).max(), "err by tree:", [round(float(e[a:a+n_].max()),4) for a, n_ in zip(mjm.tree_dofadr, mjm.tree_dofnum)])
    f = d.efc.force.numpy()[1][:n]; fe = np.abs(f - s.efc_force[:n])
    print("  force err max", fe.max(), "at row", fe.argmax(), "type", s.efc_type[fe.argmax()], "|f| max", np.abs(s.efc_force[:n]).max())
    st = d.efc.state.numpy()[1][:n]
    print("  state mismatches", int((st != s.efc_state[:n]).sum()))
    # cost of both solutions under the oracle's objective is not exposed: compare KKT residual instead
    r_g = (np.asarray(s.dense_M()) @ d.qacc.numpy()[1].astype(np.float64)) - s.qfrc_smooth - s.efc_J[:n].T @ f.astype(np.float64)
    r_o = (np.asarray(s.dense_M()) @ s.qacc) - s.qfrc_smooth - s.efc_J[:n].T @ s.efc_force[:n]
    print("  KKT residual gpu", np.abs(r_g).max(), "oracle", np.abs(r_o).max())
