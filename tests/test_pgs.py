"""PGS solver: oracle self-checks (CPU) and HIP-vs-oracle parity (GPU).

The reference has no PGS (types.py:502), so there is no reference test to mirror: the oracle's restatement of MuJoCo C's
dual projected Gauss-Seidel (oracle/mjref.c:solve_pgs) is pinned against an independent algorithm -- the Newton solution
of the primal problem, which a converged PGS must reproduce -- and the HIP kernel (csrc/pgs.hpp) against the oracle.
"""

import os

import numpy as np
import pytest

import conftest
import mujoco_warp_amd as mjw
from conftest import relerr
from oracle import ref

PGS, NEWTON = int(mjw.SolverType.PGS), int(mjw.SolverType.NEWTON)

FRICTIONLOSS_XML = """
<mujoco>
  <option timestep="0.002"/>
  <worldbody>
    <geom name="floor" type="plane" size="0 0 .05" pos="0 0 -0.9"/>
    <body name="a" pos="0 0 0">
      <joint name="j1" type="hinge" axis="0 1 0" frictionloss="0.4" armature="0.01" range="-50 50" limited="true"/>
      <geom type="capsule" fromto="0 0 0 0.5 0 0" size="0.04"/>
      <body name="b" pos="0.5 0 0">
        <joint name="j2" type="hinge" axis="0 1 0" frictionloss="0.05" armature="0.01"/>
        <geom type="capsule" fromto="0 0 0 0.4 0 0" size="0.04"/>
        <body name="c" pos="0.4 0 0">
          <joint name="j3" type="slide" axis="1 0 0" frictionloss="2.0" range="-0.1 0.1" limited="true"/>
          <geom type="sphere" size="0.06"/>
        </body>
      </body>
    </body>
  </worldbody>
</mujoco>
"""


def _humanoid_state(s, nstep=30):
  s.reset(key=0)
  for i in range(nstep):
    s.ctrl_noise(i, 0)
    s.step()


# ------------------------------------------------------------------------------------------------------- oracle (CPU)
def test_oracle_pgs_fixed_point_is_the_newton_solution(humanoid):
  a = ref.RefSim(humanoid, nconmax=24, njmax=64, solver=NEWTON, tolerance=1e-12, iterations=300)
  b = ref.RefSim(humanoid, nconmax=24, njmax=64, solver=PGS, tolerance=1e-14, iterations=5000)
  _humanoid_state(a)
  for k in ("qpos", "qvel", "ctrl", "qacc_warmstart"):
    getattr(b, k)[:] = getattr(a, k)
  a.forward()
  b.forward()
  n = a.nefc
  assert n == b.nefc and n > 8
  np.testing.assert_allclose(b.qacc, a.qacc, rtol=0, atol=1e-4 * np.abs(a.qacc).max())
  np.testing.assert_allclose(b.efc_force[:n], a.efc_force[:n], rtol=0, atol=1e-5 * np.abs(a.efc_force).max())
  assert (b.efc_state[:n] == a.efc_state[:n]).all()
  # dual feasibility and complementarity of the PGS forces on limit / contact rows
  f = b.efc_force[:n]
  jar = a.efc_J[:n] @ b.qacc - a.efc_aref[:n]
  assert (f >= 0).all()
  assert np.abs(f * (jar + f / a.efc_D[:n])).max() < 1e-6 * np.abs(f).max() ** 2


def test_oracle_pgs_friction_loss_and_equality_rows():
  for xml, kw in ((FRICTIONLOSS_XML, dict(nconmax=8, njmax=16)), (None, dict(nconmax=8, njmax=16))):
    mjm = mjw.mjcf.from_xml_string(xml) if xml else mjw.mjcf.load_xml(conftest.PANDA_XML)
    a = ref.RefSim(mjm, solver=NEWTON, tolerance=1e-12, iterations=300, ls_iterations=100, **kw)
    b = ref.RefSim(mjm, solver=PGS, tolerance=1e-15, iterations=20000, **kw)
    rng = np.random.default_rng(5)
    for s in (a, b):
      s.reset()
    a.qpos[:] = mjm.qpos0 + 0.2 * rng.standard_normal(mjm.nq)
    a.qvel[:] = 0.5 * rng.standard_normal(mjm.nv)
    if xml is None:
      a.qpos[7:9] = (0.03, 0.01)  # fingers apart: the equality row is violated
    for k in ("qpos", "qvel"):
      getattr(b, k)[:] = getattr(a, k)
    a.forward()
    b.forward()
    n = a.nefc
    assert n == b.nefc and (a.nf > 0 if xml else a.ne == 1)
    np.testing.assert_allclose(b.qacc, a.qacc, rtol=0, atol=2e-4 * max(1.0, np.abs(a.qacc).max()))
    if xml:  # friction-loss forces stay inside their box
      fl = a.efc_frictionloss[a.ne : a.ne + a.nf]
      assert (np.abs(b.efc_force[a.ne : a.ne + a.nf]) <= fl + 1e-12).all()


def test_oracle_pgs_cost_decreases_monotonically(humanoid):
  """Every sweep lowers the dual cost; more sweeps never move the solution away from the fixed point."""
  sims = [ref.RefSim(humanoid, nconmax=24, njmax=64, solver=PGS, tolerance=0.0, iterations=k) for k in (1, 5, 25, 125, 2000)]
  _humanoid_state(sims[-1])
  for s in sims[:-1]:
    for k in ("qpos", "qvel", "ctrl", "qacc_warmstart"):
      getattr(s, k)[:] = getattr(sims[-1], k)
  for s in sims:
    s.forward()
  err = [np.abs(s.qacc - sims[-1].qacc).max() for s in sims[:-1]]
  assert all(e1 <= e0 * 1.0001 + 1e-12 for e0, e1 in zip(err, err[1:])), err
  assert [s.solver_niter for s in sims[:-1]] == [1, 5, 25, 125]


def test_oracle_pgs_golden_regression(humanoid):
  g = np.load(os.path.join(conftest.GOLDEN_DIR, "humanoid_pgs_forward.npz"))
  f = np.load(os.path.join(conftest.GOLDEN_DIR, "humanoid_oracle_forward.npz"))
  s = ref.RefSim(humanoid, nconmax=24, njmax=64, tolerance=float(g["tolerance"]), solver=PGS)
  s.reset(key=0)
  for k in ("qpos", "qvel", "ctrl", "qacc_warmstart"):
    getattr(s, k)[:] = f["in_" + k]
  s.forward()
  assert s.nefc == int(g["nefc"]) and s.solver_niter == int(g["solver_niter"])
  np.testing.assert_allclose(s.qacc, g["qacc"], rtol=1e-9, atol=1e-9)
  np.testing.assert_allclose(s.efc_force[: s.nefc], g["efc_force"], rtol=1e-9, atol=1e-9)
  s.step()
  np.testing.assert_allclose(s.qpos, g["qpos_next"], rtol=1e-12, atol=1e-12)


def test_put_model_accepts_pgs_and_rejects_unknown_solver(humanoid):
  import copy

  mjm = copy.deepcopy(humanoid)
  mjm.opt.solver = 7
  with pytest.raises(NotImplementedError):
    mjw.put_model(mjm)


# ------------------------------------------------------------------------------------------------------------ GPU
def _pair(mjm, nworld, nconmax, njmax, warm_steps, **over):
  mjm.opt.solver = PGS
  for k, v in over.items():
    setattr(mjm.opt, k, v)
  s = ref.RefSim(mjm, nconmax=nconmax, njmax=njmax, tolerance=max(mjm.opt.tolerance, 1e-6))
  s.reset(key=0 if mjm.nkey else None)
  for i in range(warm_steps):
    if mjm.nu:
      s.ctrl_noise(i, 0)
    s.step()
  m = mjw.put_model(mjm)
  d = mjw.put_data(mjm, mjw.MjData(mjm), nworld=nworld, nconmax=nconmax, njmax=njmax)
  _sync(s, d)
  return s, m, d


def _sync(s, d):
  for name in ("qpos", "qvel", "act", "ctrl", "qacc_warmstart"):
    dst = getattr(d, name)
    if dst.size:
      dst.assign(np.tile(getattr(s, name).astype(np.float32), (d.nworld, 1)))


def _check_pgs_solution(s, d, tol, w=-1):
  n = s.nefc
  assert int(d.nefc.numpy()[w]) == n
  assert relerr(d.qacc.numpy()[w], s.qacc) <= tol
  assert relerr(d.qfrc_constraint.numpy()[w], s.qfrc_constraint) <= tol
  assert relerr(d.efc.Ma.numpy()[w], s.Ma) <= tol
  assert relerr(d.qacc_smooth.numpy()[w], s.qacc_smooth) <= 1e-4
  if n:
    assert relerr(d.efc.force.numpy()[w, :n], s.efc_force[:n]) <= tol
    # states can only differ on rows whose force sits at a bound within round-off
    diff = d.efc.state.numpy()[w, :n] != s.efc_state[:n]
    assert np.abs(s.efc_force[:n][diff]).max(initial=0.0) <= tol * np.abs(s.efc_force[:n]).max()


@pytest.mark.gpu
@pytest.mark.parametrize("iterations,njmax", [(100, 64), (3, 64), (100, 96)])
def test_pgs_humanoid_forward_matches_oracle(iterations, njmax):
  """njmax <= 64 runs the register-resident sweep, njmax = 96 the general (LDS) one; both against the same oracle."""
  mjm = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  s, m, d = _pair(mjm, 3, 24, njmax, 15, iterations=iterations)
  s.forward()
  mjw.forward(m, d)
  _check_pgs_solution(s, d, 2e-3)
  assert abs(int(d.solver_niter.numpy()[0]) - s.solver_niter) <= 2
  q = d.qacc.numpy()
  assert (q == q[0]).all()
  # cold start
  mjm.opt.disableflags |= int(mjw.DisableBit.WARMSTART)
  s, m, d = _pair(mjm, 2, 24, njmax, 15, iterations=iterations)
  s.forward()
  mjw.forward(m, d)
  _check_pgs_solution(s, d, 2e-3)


@pytest.mark.gpu
def test_pgs_golden_fixture():
  mjm = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  mjm.opt.solver = PGS
  g = np.load(os.path.join(conftest.GOLDEN_DIR, "humanoid_pgs_forward.npz"))
  f = np.load(os.path.join(conftest.GOLDEN_DIR, "humanoid_oracle_forward.npz"))
  m = mjw.put_model(mjm)
  d = mjw.put_data(mjm, mjw.MjData(mjm), nworld=2, nconmax=24, njmax=64)
  for k in ("qpos", "qvel", "ctrl", "qacc_warmstart"):
    getattr(d, k).assign(np.tile(f["in_" + k].astype(np.float32), (2, 1)))
  mjw.step(m, d)
  n = int(g["nefc"])
  assert int(d.nefc.numpy()[1]) == n
  assert relerr(d.qacc.numpy()[1], g["qacc"]) <= 2e-3
  assert relerr(d.efc.force.numpy()[1, :n], g["efc_force"]) <= 2e-3
  assert abs(int(d.solver_niter.numpy()[1]) - int(g["solver_niter"])) <= 2
  assert relerr(d.qpos.numpy()[1], g["qpos_next"]) <= 1e-5
  assert relerr(d.qvel.numpy()[1], g["qvel_next"]) <= 1e-3


@pytest.mark.gpu
def test_pgs_per_step_parity_resynced():
  mjm = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  s, m, d = _pair(mjm, 2, 24, 64, 0)
  worst_q = worst_v = 0.0
  boundary_steps = 0
  for i in range(120):
    s.ctrl_noise(i, 0)
    _sync(s, d)
    mjw.step(m, d)
    s.forward()
    if int(d.nefc.numpy()[1]) != s.nefc:
      # a contact whose distance is at the detection boundary within float32 resolution is seen by one side only (on the
      # PGS trajectory this happens at step 8: dist = -2.4e-8 m); the truncated PGS iterate then differs by O(10 %) for
      # that step.  Such steps must be rare and explained by a boundary contact; the state is re-synchronised afterwards.
      margin_gap = np.abs(s.con_dist[: s.ncon] - s.con_includemargin[: s.ncon]).min()
      assert margin_gap < 1e-6, (i, margin_gap)
      boundary_steps += 1
      s.step()
      continue
    s.step()
    worst_q = max(worst_q, relerr(d.qpos.numpy()[1], s.qpos))
    worst_v = max(worst_v, relerr(d.qvel.numpy()[1], s.qvel))
  assert boundary_steps <= 2
  # PGS converges linearly: both sides stop when a sweep improves the cost by less than the tolerance, where the iterate
  # is still ~1e-4 from the fixed point, and float32 / float64 may stop one sweep apart (measured qacc 4e-4)
  assert worst_q <= 1e-5, worst_q
  assert worst_v <= 1e-3, worst_v


@pytest.mark.gpu
def test_pgs_friction_loss_equality_and_wide_models():
  # friction-loss rows (box constraint) + joint limits
  mjm = mjw.mjcf.from_xml_string(FRICTIONLOSS_XML)
  s, m, d = _pair(mjm, 3, 8, 16, 0, iterations=200)
  rng = np.random.default_rng(11)
  s.qpos[:] = 0.3 * rng.standard_normal(mjm.nq)
  s.qvel[:] = rng.standard_normal(mjm.nv)
  _sync(s, d)
  for _ in range(40):
    _sync(s, d)
    s.forward()
    mjw.forward(m, d)
    assert s.nf == 3
    _check_pgs_solution(s, d, 2e-3)
    s.step()
  # joint equality (Panda, implicitfast)
  mjm = mjw.mjcf.load_xml(conftest.PANDA_XML)
  s, m, d = _pair(mjm, 3, 8, 16, 0, iterations=200)
  s.qpos[7:9] = (0.03, 0.01)
  _sync(s, d)
  s.forward()
  mjw.forward(m, d)
  assert s.ne == 1
  _check_pgs_solution(s, d, 2e-3)
  # 64 lanes per world (G1, nv = 35, up to 192 rows)
  mjm = mjw.mjcf.load_xml(conftest.G1_XML)
  s, m, d = _pair(mjm, 2, 48, 192, 10, iterations=100)
  s.forward()
  mjw.forward(m, d)
  assert s.nefc > 32
  _check_pgs_solution(s, d, 5e-3)


@pytest.mark.gpu
def test_pgs_large_batch_properties():
  """8192 humanoid worlds under PGS: identical worlds stay bitwise identical, noisy worlds stay finite and on the floor."""
  mjm = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  mjm.opt.solver = PGS
  m = mjw.put_model(mjm)
  d = mjw.make_data(mjm, nworld=8192, nconmax=24, njmax=64)
  mjw.reset_data_keyframe(m, d, 0)
  for _ in range(10):
    mjw.step(m, d)
  q = d.qpos.numpy()
  assert np.isfinite(q).all() and (q == q[0]).all()
  for i in range(30):
    mjw.ctrl_noise(m, d, i)
    mjw.step(m, d)
  q = d.qpos.numpy()
  assert np.isfinite(q).all() and (q[:, 2] > 0.2).all()
  assert (d.overflow.numpy() & 0x1FF == 0).all()
  assert d.solver_niter.numpy().max() <= mjm.opt.iterations


# ------------------------------------------------------------------------------------- elliptic cones, more than 64 dofs (round 3)
@pytest.mark.parametrize("condim", [3, 4, 6])
@pytest.mark.parametrize("impratio", [1.0, 10.0])
def test_oracle_pgs_elliptic_fixed_point_is_the_newton_solution(condim, impratio):
  """PGS on elliptic cones (per contact: apex ray / ray update / friction QCQP at fixed normal force, oracle/mjref.c:solve_pgs) solves the
  dual problem; converged, it must reproduce the Newton solution of the primal elliptic problem -- an independent algorithm with
  different zones, costs and cone parametrisation (solver.py:272-517).  Measured: qacc to 2e-6, forces to 5e-6."""
  from tests.test_elliptic import _model

  mjm = _model(condim)
  mjm.opt.impratio = impratio
  n = ref.RefSim(mjm, nconmax=32, njmax=128, solver=NEWTON, tolerance=1e-12, iterations=200, ls_iterations=100)
  p = ref.RefSim(mjm, nconmax=32, njmax=128, solver=PGS, tolerance=1e-14, iterations=20000)
  n.reset(key=0)
  p.reset(key=0)
  cone_rows = 0
  for step in range(40):
    for f in ("qpos", "qvel", "qacc_warmstart"):
      getattr(p, f)[:] = getattr(n, f)
    n.forward()
    p.forward()
    k = n.nefc
    assert k == p.nefc and k > 0
    assert relerr(p.qacc, n.qacc) <= 2e-5
    assert relerr(p.efc_force[:k], n.efc_force[:k]) <= 5e-5
    # every contact force inside its friction cone
    for c in range(p.ncon):
      r0, dim = p.con_efc_address[c][0], p.con_dim[c]
      if r0 >= 0 and dim > 1:
        tt = np.sqrt(np.sum((p.efc_force[r0 + 1 : r0 + dim] / p.con_friction[c][: dim - 1]) ** 2))
        assert p.efc_force[r0] >= 0 and tt <= p.efc_force[r0] * (1 + 1e-9) + 1e-12
    cone_rows += int((p.efc_state[:k] == 4).sum())
    n.step()
  assert cone_rows > 0  # sliding contacts on the cone's surface were part of the comparison


def _check_pgs_cost(s, d, w, rel):
  """The engine's forces under the oracle's float64 dual problem: inside every cone and no worse a minimiser than the oracle's own
  after the same number of sweeps (PGS iterates are compared through the objective when it has not converged)."""
  k = s.nefc
  M, J = s.dense_M(), s.efc_J[:k]
  A = J @ np.linalg.solve(M, J.T) + np.diag(1.0 / s.efc_D[:k])
  b = J @ s.qacc_smooth - s.efc_aref[:k]
  cost = lambda f: 0.5 * f @ A @ f + f @ b
  fg, fo = d.efc.force.numpy()[w, :k].astype(np.float64), s.efc_force[:k]
  assert cost(fg) <= cost(fo) + rel * abs(cost(fo)), (cost(fg), cost(fo))
  for c in range(s.ncon):
    r0, dim = s.con_efc_address[c][0], s.con_dim[c]
    if r0 >= 0 and dim > 1 and s.efc_type[r0] == 7:
      tt = np.sqrt(np.sum((fg[r0 + 1 : r0 + dim] / s.con_friction[c][: dim - 1]) ** 2))
      assert fg[r0] >= 0 and tt <= fg[r0] * (1 + 1e-4) + 1e-6, (c, fg[r0], tt)


@pytest.mark.gpu
@pytest.mark.parametrize("condim", [3, 4, 6])
def test_gpu_pgs_elliptic_matches_oracle(condim):
  """The generic PGS kernel (csrc/pgs_big.hpp) on the elliptic scenes of tests/test_elliptic.py: converged solution vs the oracle's."""
  from tests.test_elliptic import _model

  mjm = _model(condim)
  s, m, d = _pair(mjm, 3, 32, 128, 0, iterations=3000, tolerance=1e-9)
  assert d.ws_pgsB.size > 0
  worst = 0.0
  for step in range(30):
    _sync(s, d)
    s.forward()
    mjw.forward(m, d)
    k = s.nefc
    assert int(d.nefc.numpy()[-1]) == k
    worst = max(worst, relerr(d.qacc.numpy()[-1], s.qacc))
    assert relerr(d.efc.force.numpy()[-1, :k], s.efc_force[:k]) <= 2e-2
    _check_pgs_cost(s, d, -1, 1e-4)
    s.step()
  print(f"pgs elliptic condim {condim}: qacc {worst:.3g}")
  assert worst <= 5e-3, worst


@pytest.mark.gpu
def test_gpu_pgs_clutter_nv136():
  """BASELINE configs[4] names PGS: the generic kernel on the 136-dof elliptic clutter model (sleeping off: it needs Newton, as in the
  reference), per re-synchronised step against the oracle's PGS at the same sweep cap, compared through the dual objective."""
  mjm = mjw.mjcf.load_xml(os.path.join(conftest.ROOT, "tests", "models", "clutter_synth.xml"))
  mjm.opt.enableflags = 0
  s, m, d = _pair(mjm, 2, 256, 384, 0, iterations=60)
  worst_q = worst_v = 0.0
  same = 0
  for i in range(120):
    _sync(s, d)
    mjw.step(m, d)
    s.step()
    if int(d.nefc.numpy()[1]) != s.nefc:
      continue
    same += 1
    if i % 10 == 0 and s.nefc:
      _check_pgs_cost(s, d, 1, 2e-3)
    worst_q = max(worst_q, relerr(d.qpos.numpy()[1], s.qpos))
    worst_v = max(worst_v, relerr(d.qvel.numpy()[1], s.qvel))
  print(f"pgs clutter: {same}/120 steps, qpos {worst_q:.3g} qvel {worst_v:.3g}")
  assert same >= 100 and worst_q <= 1e-4 and worst_v <= 5e-2, (same, worst_q, worst_v)
  assert np.isfinite(d.qpos.numpy()).all() and (d.qpos.numpy()[0] == d.qpos.numpy()[1]).all()


@pytest.mark.gpu
def test_gpu_pgs_island_parallel_sweeps_equal_the_sequential_sweep():
  """The generic PGS kernel sweeps the constraint islands of a world in parallel (eight wavefronts, 16-lane tracks: csrc/pgs_big.hpp) and
  claims every island's iterates are those of the sequential sweep.  Same rollout with one wavefront per world (MJH_PGSB_WAVES=1: all rows
  in turn) and with the default: the launch configuration is read once per process, so each runs in its own interpreter."""
  import subprocess, sys, tempfile

  script = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import mujoco_warp_amd as mjw
mjm = mjw.mjcf.load_xml(os.path.join(sys.argv[1], "tests", "models", "clutter_synth.xml"))
mjw.override_model(mjm, ["opt.solver=pgs", "opt.enableflags=0", "opt.iterations=40"])
m = mjw.put_model(mjm)
mjd = mjw.MjData(mjm)
mjw.mj_resetDataKeyframe(mjm, mjd, 0)
d = mjw.put_data(mjm, mjd, nworld=4, nconmax=256, njmax=384)
out = []
for i in range(150):
  mjw.step(m, d)
  if i % 30 == 29:
    out.append(np.concatenate([d.qacc.numpy().ravel(), d.efc.force.numpy().ravel(), d.solver_niter.numpy().ravel().astype(np.float32)]))
np.save(sys.argv[2], np.stack(out))
'''
  res = {}
  with tempfile.TemporaryDirectory() as tmp:
    for waves in ("1", "8"):
      env = dict(os.environ, MJH_PGSB_WAVES=waves)
      path = os.path.join(tmp, f"w{waves}.npy")
      p = subprocess.run([sys.executable, "-c", script, conftest.ROOT, path], env=env, capture_output=True, text=True, timeout=300)
      assert p.returncode == 0, p.stderr[-2000:]
      res[waves] = np.load(path)
  a, b = res["1"], res["8"]
  assert a.shape == b.shape and np.isfinite(a).all()
  # identical iterates per island; only the order in which the sweep's improvement is summed differs (it decides nothing here: the cap is hit)
  scale = np.abs(a).max(axis=1, keepdims=True)
  assert (np.abs(a - b) <= 1e-5 * scale).all(), float((np.abs(a - b) / scale).max())
