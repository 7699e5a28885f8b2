"""Generates tests/golden/*.npz from the float64 oracle (PARITY UNPINNED: see oracle/mjref.h).

The reference holds no golden vectors for the hot path and MuJoCo C is unavailable here, so these files are
regression anchors produced by the oracle itself, not by the reference.  Run from the repo root:
    python tests/golden/make_golden.py
"""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mujoco_warp_amd as mjw  # noqa: E402
from oracle import ref  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
  mjm = mjw.mjcf.load_xml(os.path.join(ROOT, "benchmarks", "humanoid", "humanoid.xml"))
  nstep, stride, tol = 400, 20, 1e-6
  s = ref.RefSim(mjm, nconmax=24, njmax=64, tolerance=tol)
  s.reset(key=0)
  ok, qp, qv = s.rollout(nstep, worldid=0)
  assert ok == nstep
  np.savez(os.path.join(OUT, "humanoid_oracle_rollout.npz"), qpos=qp[::stride], qvel=qv[::stride], nstep=nstep, stride=stride,
           tolerance=tol, worldid=0)
  # one-step fixtures from a generic state: inputs + oracle outputs of forward()
  s.reset(key=0)
  s.rollout(25, worldid=3, record=False)
  s.ctrl_noise(25, 3)
  state = {k: getattr(s, k).copy() for k in ("qpos", "qvel", "ctrl", "qacc_warmstart")}
  s.forward()
  out = {k: getattr(s, k).copy() for k in ("xpos", "xquat", "subtree_com", "cinert", "cdof", "M", "qfrc_bias", "qfrc_passive",
                                            "qfrc_actuator", "qacc_smooth", "qacc", "qfrc_constraint")}
  out["nefc"], out["ncon"] = s.nefc, s.ncon
  out["efc_J"], out["efc_D"], out["efc_aref"] = s.efc_J[: s.nefc].copy(), s.efc_D[: s.nefc].copy(), s.efc_aref[: s.nefc].copy()
  s.step()
  out["qpos_next"], out["qvel_next"] = s.qpos.copy(), s.qvel.copy()
  np.savez(os.path.join(OUT, "humanoid_oracle_forward.npz"), tolerance=tol, **{"in_" + k: v for k, v in state.items()}, **out)
  # BASELINE configs[2] and [3] models: one-step fixtures (G1: nv 35, implicitfast; Panda: joint equality, affine actuators)
  for name, rel, ncm, njm, warm in (("g1", ("unitree_g1", "scene_flat.xml"), 48, 192, 12), ("panda", ("franka_emika_panda", "scene.xml"), 8, 16, 0)):
    mjm = mjw.mjcf.load_xml(os.path.join(ROOT, "benchmarks", *rel))
    s = ref.RefSim(mjm, nconmax=ncm, njmax=njm, tolerance=tol, iterations=100, ls_iterations=50)
    s.reset(key=0 if mjm.nkey else None)
    if name == "panda":
      rng = np.random.default_rng(7)
      s.qpos[:7] = 0.3 * rng.standard_normal(7)
      s.qpos[3] = -1.5
      s.qpos[7:9] = (0.03, 0.01)
      s.qvel[:] = 0.2 * rng.standard_normal(mjm.nv)
      s.ctrl[:] = 0.2 * rng.standard_normal(mjm.nu)
    for i in range(warm):
      s.ctrl_noise(i, 1)
      s.step()
    state = {k: getattr(s, k).copy() for k in ("qpos", "qvel", "ctrl", "qacc_warmstart")}
    s.forward()
    out = {k: getattr(s, k).copy() for k in ("xpos", "xquat", "subtree_com", "cdof", "M", "qfrc_bias", "qfrc_actuator", "qacc_smooth",
                                              "qacc", "qfrc_constraint")}
    out["nefc"], out["ncon"], out["ne"] = s.nefc, s.ncon, s.ne
    out["efc_J"], out["efc_D"], out["efc_aref"] = s.efc_J[: s.nefc].copy(), s.efc_D[: s.nefc].copy(), s.efc_aref[: s.nefc].copy()
    s.step()
    out["qpos_next"], out["qvel_next"] = s.qpos.copy(), s.qvel.copy()
    np.savez(os.path.join(OUT, f"{name}_oracle_forward.npz"), tolerance=tol, nconmax=ncm, njmax=njm,
             **{"in_" + k: v for k, v in state.items()}, **out)
  pgs_fixture()
  scene_fixtures()
  print("wrote golden files to", OUT)


def scene_fixtures():
  """Forward + one step of the collider scenes and of the three-humanoid model (generic nv > 64 solver)."""
  sys.path.insert(0, os.path.join(ROOT, "tests"))
  import conftest

  tol = 1e-6
  for name, mjm, ncm, njm, warm in (
    ("boxes", mjw.mjcf.from_xml_string(conftest.BOX_BOX_XML), 64, 192, 30),
    ("capsule_box", mjw.mjcf.from_xml_string(conftest.CAPSULE_BOX_XML), 48, 160, 15),
    ("three_humanoids", mjw.mjcf.load_xml(os.path.join(ROOT, "benchmarks", "humanoid", "three_humanoids.xml")), 100, 192, 40),
  ):
    s = ref.RefSim(mjm, nconmax=ncm, njmax=njm, tolerance=tol)
    s.reset(key=0 if mjm.nkey else None)
    for i in range(warm):
      if mjm.nu:
        s.ctrl_noise(i, 2)
      s.step()
    state = {k: getattr(s, k).copy() for k in ("qpos", "qvel", "ctrl", "qacc_warmstart")}
    s.forward()
    out = {k: getattr(s, k).copy() for k in ("xpos", "xquat", "M", "qfrc_bias", "qacc_smooth", "qacc", "qfrc_constraint")}
    out["nefc"], out["ncon"] = s.nefc, s.ncon
    out["con_dist"], out["con_pos"], out["con_geom"] = s.con_dist[: s.ncon].copy(), s.con_pos[: s.ncon].copy(), s.con_geom[: s.ncon].copy()
    s.step()
    out["qpos_next"], out["qvel_next"] = s.qpos.copy(), s.qvel.copy()
    np.savez(os.path.join(OUT, f"{name}_oracle_forward.npz"), tolerance=tol, nconmax=ncm, njmax=njm,
             **{"in_" + k: v for k, v in state.items()}, **out)


def pgs_fixture():
  """PGS solve of the humanoid forward fixture's state (same inputs; solver = PGS, 100 sweeps)."""
  mjm = mjw.mjcf.load_xml(os.path.join(ROOT, "benchmarks", "humanoid", "humanoid.xml"))
  f = np.load(os.path.join(OUT, "humanoid_oracle_forward.npz"))
  tol = 1e-6
  s = ref.RefSim(mjm, nconmax=24, njmax=64, tolerance=tol, solver=0)
  s.reset(key=0)
  for k in ("qpos", "qvel", "ctrl", "qacc_warmstart"):
    getattr(s, k)[:] = f["in_" + k]
  s.forward()
  out = {k: getattr(s, k).copy() for k in ("qacc", "qfrc_constraint")}
  out["efc_force"], out["efc_state"] = s.efc_force[: s.nefc].copy(), s.efc_state[: s.nefc].copy()
  out["nefc"], out["solver_niter"] = s.nefc, s.solver_niter
  s.step()
  out["qpos_next"], out["qvel_next"] = s.qpos.copy(), s.qvel.copy()
  np.savez(os.path.join(OUT, "humanoid_pgs_forward.npz"), tolerance=tol, **out)


if __name__ == "__main__":
  main()
