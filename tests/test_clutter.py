"""BASELINE configs[4]-class workload (aloha_clutter: nv 136, elliptic cones, impratio 10, timestep 0.002, sleeping, mesh objects).

The aloha assets are not in the reference tree, so the workload is tests/models/clutter_synth.xml (tools/make_clutter_synth.py): two 8-dof
arms with finger equalities and position actuators, a table, 20 free convex-mesh objects -- nv = 136, 22 kinematic trees.  What it
exercises that no other test does: elliptic cones on the per-island path (islands of <= 32 and 33..64 dofs) and in the generic solver,
sleeping on a model with more than 64 dofs, `init_asleep`, `nvmax`.
"""

import os

import numpy as np
import pytest

import mujoco_warp_amd as mjw
from oracle import ref
from tests import conftest
from tests.conftest import relerr

XML = os.path.join(conftest.ROOT, "tests", "models", "clutter_synth.xml")
NCONMAX, NJMAX = 256, 384


def test_model_is_configs4_class():
  mjm = mjw.mjcf.load_xml(XML)
  assert (mjm.nv, mjm.nq, mjm.nu, mjm.ntree, mjm.neq, mjm.nmesh) == (136, 156, 14, 22, 2, 20)
  assert int(mjm.opt.cone) == int(mjw.ConeType.ELLIPTIC) and float(mjm.opt.impratio) == 10.0 and float(mjm.opt.timestep) == 0.002
  assert int(mjm.opt.enableflags) & int(mjw.EnableBit.SLEEP)
  assert list(mjm.tree_sleep_policy[:2]) == [int(mjw.SleepPolicy.AUTO_NEVER)] * 2  # the actuated arms never sleep
  assert (np.asarray(mjm.mesh_vertnum) >= 4).all() and int(np.max(mjm.mesh_vertnum)) == 12  # both support-function branches


def test_oracle_settles_and_sleeps():
  mjm = mjw.mjcf.load_xml(XML)
  s = ref.RefSim(mjm, nconmax=NCONMAX, njmax=NJMAX)
  s.reset(key=0)
  awake, rows = [], []
  for i in range(400):
    s.step()
    awake.append(int(s.ntree_awake))
    rows.append(int(s.nefc))
    assert s.overflow & 0xFF == 0  # no capacity bit
  assert max(rows) > 150 and np.isfinite(s.qpos).all()
  assert awake[0] == 22 and min(awake) <= 12  # objects resting on the table go to sleep; the arms stay awake
  assert int(s.tree_awake[0]) == 1 and int(s.tree_awake[1]) == 1


def _tables_equal(d, w, s):
  for k in ("tree_asleep", "tree_awake", "body_awake", "tree_island"):
    if not (getattr(d, k).numpy()[w] == getattr(s, k)).all():
      return False
  return int(d.nisland.numpy()[w]) == int(s.nisland) and int(d.nv_awake.numpy()[w]) == int(s.nv_awake)


def _sync(d, s, nworld):
  for name in ("qpos", "qvel", "qacc_warmstart", "ctrl"):
    getattr(d, name).assign(np.tile(getattr(s, name).astype(np.float32), (nworld, 1)))


@pytest.mark.gpu
def test_gpu_newton_sleep_per_step_parity():
  """Newton + elliptic + sleeping at nv 136: per re-synchronised step against the oracle over the drop / settle / sleep phases, sleep
  tables identical whenever the discrete contact sets agree."""
  mjm = mjw.mjcf.load_xml(XML)
  s = ref.RefSim(mjm, nconmax=NCONMAX, njmax=NJMAX)
  s.reset(key=0)
  m = mjw.put_model(mjm)
  assert m.tree_solve == 1 and m.sleep_enabled
  d = mjw.make_data(mjm, nworld=2, nconmax=NCONMAX, njmax=NJMAX)
  mjw.reset_data_keyframe(m, d, 0)
  eq, ev = [], []
  same = tables_ok = 0
  nstep = 260
  for i in range(nstep):
    _sync(d, s, 2)
    d.tree_asleep.assign(np.tile(s.tree_asleep, (2, 1)))  # the sleep state is part of the re-synchronised state
    mjw.update_sleep(m, d)
    mjw.step(m, d)
    s.step()
    assert (d.overflow.numpy() & 0xFF == 0).all()
    if int(d.ws_ncon.numpy()[1]) != s.ncon or int(d.nefc.numpy()[1]) != s.nefc:
      continue  # (a mesh contact within float32 resolution of the detection boundary / a face-alignment decision of multi-contact)
    same += 1
    tables_ok += _tables_equal(d, 1, s)
    eq.append(relerr(d.qpos.numpy()[1], s.qpos))
    ev.append(relerr(d.qvel.numpy()[1], s.qvel))
  eq, ev = np.array(eq), np.array(ev)
  print(f"clutter newton: {same}/{nstep} steps with identical contact sets, tables equal in {tables_ok}; qpos median {np.median(eq):.3g} p90 {np.percentile(eq, 90):.3g} "
        f"worst {eq.max():.3g}; qvel median {np.median(ev):.3g} p90 {np.percentile(ev, 90):.3g} worst {ev.max():.3g}")
  assert same >= 0.8 * nstep and tables_ok >= 0.97 * same
  # measured: qpos median 2e-7 / p90 1e-6, qvel median 5e-5 / p90 6e-4.  The worst steps (qvel 9e-2 on the spin of one 0.2 kg object) are the
  # ones where an arm pins an object against the table: an elliptic contact problem whose Newton iterates stop resolving below float32
  # cost differences (engine 8-10 iterations, float64 oracle 12-15); the forward test below bounds the same effect through stationarity
  assert np.median(eq) <= 2e-6 and np.percentile(eq, 90) <= 2e-5 and eq.max() <= 2e-3, (np.median(eq), np.percentile(eq, 90), eq.max())
  assert np.median(ev) <= 5e-4 and np.percentile(ev, 90) <= 5e-3 and ev.max() <= 0.25, (np.median(ev), np.percentile(ev, 90), ev.max())
  assert (d.qpos.numpy()[0] == d.qpos.numpy()[1]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["cg", "newton"])
def test_gpu_elliptic_islands_without_sleep(solver):
  """The per-island elliptic solves (CG and Newton) without the sleep machinery: forward fields of the constraint solve against the
  oracle on settled states."""
  mjm = mjw.mjcf.load_xml(XML)
  mjm.opt.enableflags = 0
  mjw.override_model(mjm, {"opt.solver": solver})
  if solver == "cg":
    mjm.opt.iterations, mjm.opt.ls_iterations = 200, 50
  s = ref.RefSim(mjm, nconmax=NCONMAX, njmax=NJMAX, tolerance=1e-6)
  s.reset(key=0)
  twin = ref.RefSim(mjm, nconmax=NCONMAX, njmax=NJMAX, tolerance=1e-6, real="f32")  # the same restatement in float32: the floor
  twin.reset(key=0)
  m = mjw.put_model(mjm)
  d = mjw.make_data(mjm, nworld=2, nconmax=NCONMAX, njmax=NJMAX)
  checked = 0
  worst = worst_kkt = twin_kkt = 0.0
  for i in range(150):
    s.step()
    if i < 60 or i % 6:
      continue
    _sync(d, s, 2)
    mjw.forward(m, d)
    for name in ("qpos", "qvel", "qacc_warmstart", "ctrl"):
      getattr(twin, name)[:] = getattr(s, name)
    twin.forward()
    s.forward()
    if int(d.nefc.numpy()[1]) != s.nefc:
      continue
    if twin.nefc == s.nefc:
      ft = twin.efc_force[: s.nefc].astype(np.float64)
      twin_kkt = max(twin_kkt, np.abs(s.dense_M() @ twin.qacc.astype(np.float64) - s.qfrc_smooth - s.efc_J[: s.nefc].T @ ft).max() / max(1.0, np.abs(s.efc_force[: s.nefc]).max()))
    checked += 1
    assert int(d.solver_niter.numpy()[1]) < int(mjm.opt.iterations)
    worst = max(worst, relerr(d.qacc.numpy()[1], s.qacc))
    n = s.nefc
    f = d.efc.force.numpy()[1][:n].astype(np.float64)
    fmax = max(1.0, np.abs(s.efc_force[:n]).max())
    np.testing.assert_allclose(f, s.efc_force[:n], atol=(2e-2 if solver == "cg" else 5e-3) * fmax)
    # stationarity of the engine's solution under the oracle's float64 M, J (KKT: M qacc - qfrc_smooth - J' f = 0)
    kkt = np.abs(s.dense_M() @ d.qacc.numpy()[1].astype(np.float64) - s.qfrc_smooth - s.efc_J[:n].T @ f).max()
    worst_kkt = max(worst_kkt, kkt / fmax)
  print(f"clutter {solver}: {checked} states, qacc {worst:.3g}, KKT residual / max force {worst_kkt:.3g}")
  # measured: Newton qacc 1.2e-4, KKT 6e-4 (the pinned-object states), CG qacc 5e-3, KKT 6.5e-3.  CG at impratio 10 converges slowly (the float64 oracle needs > 100 iterations where the float32
  # engine's improvement test stops after ~60): its iterate is compared through stationarity, like G1's capped CG (tests/test_gpu.py)
  # round 4: the CG bound is calibrated by the float32 twin on the same states (its worst stationarity residual here: 1.2e-2; the engine's
  # moved between 6.5e-3 and 2.1e-2 with last-digit changes of the inputs): no worse than twice the floor
  print(f"   float32 twin KKT {twin_kkt:.3g}")
  assert checked >= 8 and worst <= (2e-2 if solver == "cg" else 5e-3) and worst_kkt <= (max(2e-2, 2 * twin_kkt) if solver == "cg" else 2e-3), (worst, worst_kkt, twin_kkt)


@pytest.mark.gpu
def test_gpu_init_asleep_and_nvmax():
  """`init_asleep` (reference cli.py:167-168: tree_asleep[:] = arange(ntree) before put_data): the objects start asleep at their
  initial pose and stay there, nothing collides with them until an arm reaches them; `nvmax` below the awake dof count raises
  OverflowType.NVMAX and changes nothing else (this engine does not compact)."""
  mjm = mjw.mjcf.load_xml(XML)
  m = mjw.put_model(mjm)
  mjd = mjw.MjData(mjm)
  mjw.mj_resetDataKeyframe(mjm, mjd, 0)
  q0 = mjd.qpos.copy()
  mjd.tree_asleep[:] = np.arange(mjm.ntree, dtype=np.int32)
  d = mjw.put_data(mjm, mjd, nworld=3, nconmax=NCONMAX, njmax=NJMAX, nvmax=56)
  assert (d.tree_awake.numpy() == 0).all() and (d.nv_awake.numpy() == 0).all()
  s = ref.RefSim(mjm, nconmax=NCONMAX, njmax=NJMAX)
  s.reset(key=0)
  s.tree_asleep[:] = np.arange(mjm.ntree)
  s.stage("update_sleep")
  for i in range(40):
    mjw.step(m, d)
    s.step()
    assert _tables_equal(d, 1, s), i
  awake = d.tree_awake.numpy()
  assert (awake[:, :2] == 1).all()  # the actuated arms wake at once (policy AUTO_NEVER) ...
  asleep = np.flatnonzero(awake[1, 2:] == 0)  # ... the objects stay asleep in mid-air where they started until a hand reaches them
  assert len(asleep) >= 12 and (awake[0] == awake[1]).all() and (awake[2] == awake[1]).all()
  for k in asleep:
    np.testing.assert_array_equal(d.qpos.numpy()[1][16 + 7 * k : 23 + 7 * k], q0[16 + 7 * k : 23 + 7 * k].astype(np.float32))
  assert relerr(d.qpos.numpy()[1][:16], s.qpos[:16]) < 2e-3  # (40 free-running steps of the arms, float32 vs float64: measured 4e-4)
  assert (d.overflow.numpy() & int(mjw.OverflowType.NVMAX) == 0).all()  # 16 awake dofs <= nvmax 56
  # fully awake (136 dofs) against nvmax 56: the flag, and the same trajectory as without a limit
  da = mjw.make_data(mjm, nworld=2, nconmax=NCONMAX, njmax=NJMAX, nvmax=56)
  db = mjw.make_data(mjm, nworld=2, nconmax=NCONMAX, njmax=NJMAX)
  for dd in (da, db):
    mjw.reset_data_keyframe(m, dd, 0)
    for _ in range(5):
      mjw.step(m, dd)
  assert (da.overflow.numpy() & int(mjw.OverflowType.NVMAX) != 0).all() and (db.overflow.numpy() & int(mjw.OverflowType.NVMAX) == 0).all()
  assert (da.qpos.numpy() == db.qpos.numpy()).all()
  with pytest.raises(ValueError):
    mjw.make_data(mjm, nworld=1, nvmax=137)
