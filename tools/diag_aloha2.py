"""GPU box: field-by-field error of the HIP engine and of the float32 twin against the float64 oracle along the aloha_pot lift."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mujoco_warp_amd as mjw
from oracle import ref
from tests.test_aloha_pot import XML, find_keys, make_trajectory, _sync

mjm = mjw.mjcf.load_xml(XML)
keys = find_keys(mjm, "lift_pot")
traj = make_trajectory(mjm, keys)
s = ref.RefSim(mjm, nconmax=64, njmax=256, broadphase_filter=15)
t = ref.RefSim(mjm, nconmax=64, njmax=256, broadphase_filter=15, real="f32")
s.reset(key=keys[0]); t.reset(key=keys[0])
m = mjw.put_model(mjm)
d = mjw.make_data(mjm, nworld=2, nconmax=64, njmax=256)
mjw.reset_data_keyframe(m, d, keys[0])
fields = ["qfrc_bias", "qfrc_passive", "qfrc_actuator", "qfrc_smooth", "qacc_smooth", "qacc", "qvel", "qpos"]
acc = {f: ([], []) for f in fields}
def err(a, b):
  return float(np.max(np.abs(np.asarray(a, np.float64) - b)) / (np.max(np.abs(b)) + 1e-12))
steps = []
for i, ctrl in enumerate(traj):
  s.ctrl[:] = ctrl
  _sync(d, s, 2)
  for n in ("qpos", "qvel", "qacc_warmstart", "ctrl"): getattr(t, n)[:] = getattr(s, n)
  mjw.step(m, d); s.step(); t.step()
  if (int(d.ws_ncon.numpy()[1]), int(d.nefc.numpy()[1])) != (s.ncon, s.nefc) or (t.ncon, t.nefc) != (s.ncon, s.nefc):
    continue
  steps.append(i)
  for f in fields:
    o = getattr(s, f)
    acc[f][0].append(err(getattr(d, f).numpy()[1], o)); acc[f][1].append(err(getattr(t, f), o))
  if i in (450, 500):
    ne = s.nefc
    a = int(d.ws_conadr.numpy()[1]); nc = s.ncon
    print("step", i, "rows:", " ".join(f"{f} {err(getattr(d.efc, f).numpy()[1][:ne], getattr(s, 'efc_' + f)[:ne]):.1e}" for f in ("D", "aref", "pos", "margin", "vel", "frictionloss")),
          "J", f"{err(d.efc.J.numpy()[1][:ne, :mjm.nv], s.efc_J[:ne]):.1e}", "force", f"{err(d.efc.force.numpy()[1][:ne], s.efc_force[:ne]):.1e}", "twin force", f"{err(t.efc_force[:ne], s.efc_force[:ne]):.1e}")
    print("   types", s.efc_type[:ne], "dims", s.con_dim[:nc], "geoms", s.con_geom[:nc].tolist())
    print("   force gpu   ", np.round(d.efc.force.numpy()[1][:ne], 4))
    print("   force oracle", np.round(s.efc_force[:ne], 4))
    print("   state gpu   ", d.efc.state.numpy()[1][:ne], "oracle", s.efc_state[:ne])
    print("   con friction gpu", d.contact.friction.numpy()[a:a+nc].round(5).tolist(), "oracle", s.con_friction[:nc].round(5).tolist())
    print("   solimp gpu", d.contact.solimp.numpy()[a:a+nc].round(4).tolist(), "oracle", s.con_solimp[:nc].round(4).tolist())
    print("   solref gpu", d.contact.solref.numpy()[a:a+nc].round(4).tolist(), "oracle", s.con_solref[:nc].round(4).tolist())
  if i in (3, 150, 500, 600, 800, 900):
    print("   niter gpu", int(d.solver_niter.numpy()[1]), "oracle", s.solver_niter, "twin", t.solver_niter, "nefc", s.nefc, "ncon", s.ncon)
    k = np.argmax(np.abs(d.qpos.numpy()[1] - s.qpos))
    print("step", i, "worst qpos dof", k, "gpu", d.qpos.numpy()[1][k], "oracle", s.qpos[k], "twin", t.qpos[k], "| qvel gpu", d.qvel.numpy()[1][max(k-1,0)], s.qvel[max(k-1,0)])
    print("   qacc gpu-oracle", np.round((d.qacc.numpy()[1] - s.qacc), 6)[:23])
    print("   qacc twin-oracle", np.round((t.qacc - s.qacc), 6)[:23])
steps = np.array(steps)
for lo, hi in ((0, 400), (400, 700), (700, 1001)):
  sel = (steps >= lo) & (steps < hi)
  print(f"steps {lo}-{hi}: " + " | ".join(f"{f} engine {np.median(np.array(acc[f][0])[sel]):.1e}/{np.max(np.array(acc[f][0])[sel]):.1e} twin {np.median(np.array(acc[f][1])[sel]):.1e}/{np.max(np.array(acc[f][1])[sel]):.1e}" for f in ("qacc", "qvel", "qpos")))
  print("     niter gpu/oracle/twin sample:", )
for f in fields:
  print(f"{f:14s} engine median {np.median(acc[f][0]):.2e} max {np.max(acc[f][0]):.2e} | twin median {np.median(acc[f][1]):.2e} max {np.max(acc[f][1]):.2e}")
