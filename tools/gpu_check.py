"""Developer tool (GPU box): per-field error report of the HIP path against the float64 oracle."""

import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mujoco_warp_amd as mjw  # noqa: E402
from oracle import ref  # noqa: E402

XML = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "benchmarks", "humanoid", "humanoid.xml")


def relerr(a, b):
  a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
  if a.size == 0:
    return 0.0
  return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-12))


def main():
  solver = sys.argv[1] if len(sys.argv) > 1 else "newton"
  mjm = mjw.mjcf.load_xml(XML)
  mjm.opt.solver = {"cg": 1, "newton": 2}[solver]
  s = ref.RefSim(mjm, nconmax=24, njmax=64, tolerance=1e-6)
  s.reset(key=0)
  s.rollout(20, record=False)
  m = mjw.put_model(mjm)
  nworld = 4
  mjd = mjw.MjData(mjm)
  d = mjw.put_data(mjm, mjd, nworld=nworld, nconmax=24, njmax=64)

  def sync_state():
    for name in ("qpos", "qvel", "ctrl", "qacc_warmstart"):
      getattr(d, name).assign(np.tile(getattr(s, name).astype(np.float32), (nworld, 1)))

  sync_state()
  mjw.forward(m, d)
  torch.cuda.synchronize()
  s.forward()
  w = nworld - 1
  print("solver", solver, "oracle: ncon", s.ncon, "nefc", s.nefc, "niter", s.solver_niter)
  print("gpu   : ncon", d.ws_ncon.numpy(), "nefc", d.nefc.numpy(), "niter", d.solver_niter.numpy(), "ovf", d.overflow.numpy())
  for name in ("xpos", "xquat", "xmat", "xipos", "ximat", "xanchor", "xaxis", "geom_xpos", "geom_xmat", "subtree_com", "cinert", "cdof",
               "crb", "M", "qLD", "qLDiagInv", "cvel", "cdof_dot", "qfrc_spring", "qfrc_damper", "qfrc_passive", "qfrc_bias", "cacc",
               "cfrc_int", "actuator_force", "qfrc_actuator", "qfrc_smooth", "qacc_smooth", "qacc", "qfrc_constraint"):
    g = getattr(d, name).numpy()[w].reshape(-1)
    o = getattr(s, name).reshape(-1)
    print(f"  {name:18s} rel {relerr(g, o):.3e}")
  print(f"  {'efc_Ma':18s} rel {relerr(d.efc.Ma.numpy()[w], s.Ma):.3e}")
  ncon = int(d.ws_ncon.numpy()[w])
  adr = int(d.ws_conadr.numpy()[w])
  if ncon == s.ncon:
    for name, oname in (("dist", "con_dist"), ("pos", "con_pos"), ("frame", "con_frame"), ("friction", "con_friction"),
                        ("solref", "con_solref"), ("solimp", "con_solimp"), ("includemargin", "con_includemargin")):
      g = getattr(d.contact, name).numpy()[adr : adr + ncon].reshape(ncon, -1)
      o = getattr(s, oname)[:ncon].reshape(ncon, -1)
      print(f"  contact.{name:12s} rel {relerr(g, o):.3e}")
    print("  contact.geom eq", np.array_equal(d.contact.geom.numpy()[adr : adr + ncon], s.con_geom[:ncon]))
  nefc = int(d.nefc.numpy()[w])
  if nefc == s.nefc:
    nv = mjm.nv
    print(f"  efc.J              rel {relerr(d.efc.J.numpy()[w, :nefc, :nv], s.efc_J[:nefc]):.3e}")
    for name in ("D", "aref", "pos", "vel", "margin", "force"):
      print(f"  efc.{name:14s} rel {relerr(getattr(d.efc, name).numpy()[w, :nefc], getattr(s, 'efc_' + name)[:nefc]):.3e}")
    print("  efc.type eq", np.array_equal(d.efc.type.numpy()[w, :nefc], s.efc_type[:nefc]),
          "state eq", np.array_equal(d.efc.state.numpy()[w, :nefc], s.efc_state[:nefc]))
  # per-step parity with re-synchronisation (north-star: 1e-5 relative per step)
  worst_q, worst_v = 0.0, 0.0
  for i in range(200):
    s.ctrl_noise(i, 0)
    sync_state()
    mjw.step(m, d)
    s.step()
    torch.cuda.synchronize()
    eq = relerr(d.qpos.numpy()[w], s.qpos)
    ev = relerr(d.qvel.numpy()[w], s.qvel)
    worst_q, worst_v = max(worst_q, eq), max(worst_v, ev)
  print(f"per-step parity over 200 re-synced steps: qpos {worst_q:.3e} qvel {worst_v:.3e}")
  # free-running trajectory
  s.reset(key=0)
  mjw.reset_data_keyframe(m, d, 0)
  for i in range(200):
    s.ctrl_noise(i, 0)
    mjw.ctrl_noise(m, d, i)
    mjw.step(m, d)
    s.step()
  torch.cuda.synchronize()
  print("free-running 200 steps: qpos err world0", relerr(d.qpos.numpy()[0], s.qpos), "finite", np.isfinite(d.qpos.numpy()).all())
  # quick timing
  nworld = 8192
  d2 = mjw.put_data(mjm, mjd, nworld=nworld, nconmax=24, njmax=64)
  mjw.reset_data_keyframe(m, d2, 0)
  ms, _ = mjw.timed_steps(m, d2, 20)
  ms, pk = mjw.timed_steps(m, d2, 100, step0=20, per_kernel=True)
  print(f"8192 worlds: {ms / 100 * 1000:.1f} us/step -> {nworld * 100 / (ms / 1000):.3e} env-steps/s")
  print("per kernel (us/step):", {k: round(v * 10, 1) for k, v in zip(mjw.KERNEL_NAMES, pk)})
  print("nefc mean", d2.nefc.numpy().mean(), "niter mean", d2.solver_niter.numpy().mean(), "nan worlds", int(np.isnan(d2.qpos.numpy()).any(axis=1).sum()))


if __name__ == "__main__":
  main()
