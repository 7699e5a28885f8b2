"""GPU parity tests: the HIP path (through the C ABI) against the float64 oracle and the golden fixtures.

Tolerances (float32 device arithmetic vs float64 oracle; the reference's own tests use 5e-4 for smooth
stages and 0.1 for the solver, smooth_test.py:32-38 / solver_test.py:34-40):
  SMOOTH   2e-5 : FK / CoM / CRBA / RNE / passive / actuation fields, relative to the field's max magnitude
  FACTOR   1e-4 : L'DL factor and M^-1 products
  EFC      5e-4 : constraint rows (D, aref depend on contact distances ~1e-4 m resolved at float32 eps)
  SOLVE    2e-3 : qacc / forces after the iterative solve
  STEP     qpos 1e-5, qvel 1e-3 relative per re-synchronised step
"""

import os

import numpy as np
import pytest
import torch

import conftest
import mujoco_warp_amd as mjw
from conftest import relerr
from oracle import ref

pytestmark = pytest.mark.gpu

SMOOTH, FACTOR, EFC, SOLVE = 2e-5, 1e-4, 5e-4, 2e-3

_SMOOTH_FIELDS = ("xpos", "xquat", "xmat", "xipos", "ximat", "xanchor", "xaxis", "geom_xpos", "geom_xmat", "subtree_com", "cinert",
                  "cdof", "crb", "M", "cvel", "cdof_dot", "qfrc_spring", "qfrc_damper", "qfrc_passive", "qfrc_bias", "cacc",
                  "cfrc_int", "actuator_force", "qfrc_actuator", "qfrc_smooth")


def _pair(mjm, nworld=3, nconmax=32, njmax=96, solver=None, integrator=None, key=0, warm_steps=15, noise=True):
  """(oracle sim, device model, device data) in the same generic state."""
  if solver is not None:
    mjm.opt.solver = solver
  if integrator is not None:
    mjm.opt.integrator = integrator
  s = ref.RefSim(mjm, nconmax=nconmax, njmax=njmax, tolerance=max(mjm.opt.tolerance, 1e-6))
  s.reset(key=key if mjm.nkey else None)
  for i in range(warm_steps):
    if noise and mjm.nu:
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


def _check_fields(s, d, names, tol, w=-1):
  for name in names:
    g = getattr(d, name).numpy()[w].reshape(-1)
    o = getattr(s, name).reshape(-1)
    assert relerr(g, o) <= tol, f"{name}: rel err {relerr(g, o):.3e} > {tol}"


def _check_contacts_and_rows(s, d, mjm, w=-1, dist_atol=2e-7):
  ww = w % d.nworld
  ncon, adr = int(d.ws_ncon.numpy()[ww]), int(d.ws_conadr.numpy()[ww])
  assert ncon == s.ncon
  sl = slice(adr, adr + ncon)
  if ncon:
    np.testing.assert_array_equal(d.contact.geom.numpy()[sl], s.con_geom[:ncon])
    np.testing.assert_array_equal(d.contact.dim.numpy()[sl], s.con_dim[:ncon])
    np.testing.assert_array_equal(d.contact.worldid.numpy()[sl], ww)
    np.testing.assert_allclose(d.contact.dist.numpy()[sl], s.con_dist[:ncon], atol=dist_atol)  # float32 eps at ~1 m
    assert relerr(d.contact.pos.numpy()[sl], s.con_pos[:ncon]) <= SMOOTH
    assert relerr(d.contact.frame.numpy()[sl].reshape(ncon, 9), s.con_frame[:ncon]) <= SMOOTH
    for a, b in (("friction", "con_friction"), ("solref", "con_solref"), ("solimp", "con_solimp"), ("includemargin", "con_includemargin")):
      assert relerr(getattr(d.contact, a).numpy()[sl].reshape(ncon, -1), getattr(s, b)[:ncon].reshape(ncon, -1)) <= 1e-6
    np.testing.assert_array_equal(d.contact.efc_address.numpy()[sl], s.con_efc_address[:ncon, : d.nmaxpyramid])
  nefc = int(d.nefc.numpy()[ww])
  assert (nefc, int(d.nf.numpy()[ww]), int(d.nl.numpy()[ww])) == (s.nefc, s.nf, s.nl)
  n = min(nefc, d.njmax)
  if n:
    assert relerr(d.efc.J.numpy()[ww, :n, : mjm.nv], s.efc_J[:n]) <= SMOOTH
    assert (d.efc.J.numpy()[ww, :n, mjm.nv :] == 0).all()
    np.testing.assert_array_equal(d.efc.type.numpy()[ww, :n], s.efc_type[:n])
    for name in ("D", "aref", "pos", "vel", "margin", "frictionloss"):
      assert relerr(getattr(d.efc, name).numpy()[ww, :n], getattr(s, "efc_" + name)[:n]) <= EFC, name


def _check_solution(s, d, w=-1, tol=SOLVE):
  ww = w % d.nworld
  n = min(int(d.nefc.numpy()[ww]), d.njmax)
  _check_fields(s, d, ("qacc_smooth",), FACTOR, w)
  _check_fields(s, d, ("qacc", "qfrc_constraint"), tol, w)
  assert relerr(d.efc.Ma.numpy()[ww], s.Ma) <= tol
  if n:
    assert relerr(d.efc.force.numpy()[ww, :n], s.efc_force[:n]) <= tol
  assert int(d.overflow.numpy()[ww]) == 0


@pytest.mark.parametrize("model,solver", [("panda", mjw.SolverType.NEWTON), ("panda", mjw.SolverType.CG), ("g1", mjw.SolverType.CG)])
def test_fused_implicitfast_equals_the_staged_integrator(model, solver):
  """`step` solves implicitfast's dense system (M + h D - h dA/dv) x = M qacc in the solver's epilogue from the register-resident row of M
  (csrc/solver.hpp impfast_acc) when the model has no activations; the stage API (`forward` then `implicit`) goes through the integrator
  kernel's sparse L'DL factor.  Same update, two factorisations."""
  mjm = mjw.mjcf.load_xml(conftest.PANDA_XML if model == "panda" else conftest.G1_XML)
  mjm.opt.solver = int(solver)
  assert int(mjm.opt.integrator) == int(mjw.IntegratorType.IMPLICITFAST) and mjm.na == 0
  m = mjw.put_model(mjm)
  nconmax, njmax = (8, 32) if model == "panda" else (48, 192)
  d1 = mjw.make_data(mjm, nworld=5, nconmax=nconmax, njmax=njmax)
  d2 = mjw.make_data(mjm, nworld=5, nconmax=nconmax, njmax=njmax)
  if mjm.nkey:
    for d in (d1, d2):
      mjw.reset_data_keyframe(m, d, 0)
  rng = np.random.default_rng(5)
  for i in range(25):
    ctrl = d1.ctrl.numpy() + rng.normal(scale=0.05, size=d1.ctrl.shape).astype(np.float32)
    for name in ("qpos", "qvel", "qacc_warmstart", "time"):
      getattr(d2, name).assign(getattr(d1, name).numpy())
    d1.ctrl.assign(ctrl)
    d2.ctrl.assign(ctrl)
    mjw.step(m, d1)
    mjw.forward(m, d2)
    mjw.implicit(m, d2)
    np.testing.assert_allclose(d1.qacc.numpy(), d2.qacc.numpy(), rtol=0, atol=1e-5 * max(1.0, np.abs(d2.qacc.numpy()).max()))
    assert relerr(d1.qvel.numpy(), d2.qvel.numpy()) <= 2e-6, i
    assert relerr(d1.qpos.numpy(), d2.qpos.numpy()) <= 1e-6, i
    assert (d1.qacc_warmstart.numpy() == d1.qacc.numpy()).all()


def test_cg_small_and_large_batch_kernels_both_match_oracle():
  """CG on this model has two product kernels (mjhip.hip cg32_choice): one world per wavefront for batches of at most 3072 worlds
  (csrc/solver_cgw.hpp), the pooled contact-basis kernel above (csrc/solver_cgp.hpp; `mjw.solver_kernel` names them).  The same state
  through both: each within the oracle tolerance, and the same answer.  (Diverged worlds at 8192: tests/test_headline_batch.py.)"""
  mjm = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  s, m, d = _pair(mjm, nworld=4, nconmax=24, njmax=64, solver=int(mjw.SolverType.CG))
  big = mjw.put_data(mjm, mjw.MjData(mjm), nworld=3200, nconmax=24, njmax=64)
  assert mjw.solver_kernel(m, d) == "cgw" and mjw.solver_kernel(m, big) == "cgp"
  _sync(s, big)
  s.forward()
  out = []
  for dd in (d, big):
    mjw.forward(m, dd)
    _check_solution(s, dd)
    q = dd.qacc.numpy()
    assert (q == q[0]).all()
    assert abs(int(dd.solver_niter.numpy()[0]) - s.solver_niter) <= 3  # (float32 CG against the float64 oracle at tolerance 1e-6: measured 18 vs 21 through the pooled kernel)
    out.append(q[0].copy())
  assert relerr(out[0], out[1]) <= 2e-4  # (different summation orders: agreement at the solver tolerance)
  for _ in range(5):  # and through the fused step (Euler in the solver's epilogue)
    mjw.step(m, d)
    mjw.step(m, big)
  assert relerr(d.qpos.numpy()[0], big.qpos.numpy()[0]) <= 1e-5 and relerr(d.qvel.numpy()[0], big.qvel.numpy()[0]) <= 1e-3


@pytest.mark.parametrize("solver", [mjw.SolverType.NEWTON, mjw.SolverType.CG])
def test_humanoid_forward_matches_oracle(solver):
  mjm = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  s, m, d = _pair(mjm, nconmax=24, njmax=64, solver=int(solver))
  mjw.forward(m, d)
  s.forward()
  _check_fields(s, d, _SMOOTH_FIELDS, SMOOTH)
  _check_fields(s, d, ("qLD", "qLDiagInv"), FACTOR)
  _check_contacts_and_rows(s, d, mjm)
  _check_solution(s, d)
  # every world got the same inputs -> bitwise identical outputs
  q = d.qacc.numpy()
  assert (q == q[0]).all()
  assert abs(int(d.solver_niter.numpy()[0]) - s.solver_niter) <= 2
  assert int(d.nacon.numpy()[0]) == d.nworld * s.ncon


def test_stage_functions_individually():
  """Each reference stage function is callable on its own and reads its inputs from Data (smooth_test.py style)."""
  mjm = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  s, m, d = _pair(mjm, nconmax=24, njmax=64)
  s.forward()
  for fn, fields, tol in (
    (mjw.kinematics, ("xpos", "xquat", "xmat", "xipos", "ximat", "xanchor", "xaxis", "geom_xpos", "geom_xmat"), SMOOTH),
    (mjw.com_pos, ("subtree_com", "cinert", "cdof"), SMOOTH),
    (mjw.crb, ("crb", "M"), SMOOTH),
    (mjw.factor_m, ("qLD", "qLDiagInv"), FACTOR),
    (mjw.com_vel, ("cvel", "cdof_dot"), SMOOTH),
    (mjw.passive, ("qfrc_spring", "qfrc_damper", "qfrc_passive"), SMOOTH),
    (mjw.rne, ("qfrc_bias", "cacc", "cfrc_int"), SMOOTH),
    (mjw.fwd_actuation, ("actuator_force", "qfrc_actuator"), SMOOTH),
    (mjw.fwd_acceleration, ("qfrc_smooth", "qacc_smooth"), FACTOR),
  ):
    for name in fields:  # poison the outputs like the reference's tests do
      getattr(d, name).fill_(float("inf"))
    fn(m, d)
    _check_fields(s, d, fields, tol)
  mjw.collision(m, d)
  mjw.make_constraint(m, d)
  _check_contacts_and_rows(s, d, mjm)
  d.qacc.fill_(float("inf"))
  mjw.solve(m, d)
  _check_solution(s, d)
  # fwd_position / fwd_velocity composites
  mjw.fwd_position(m, d)
  mjw.fwd_velocity(m, d)
  _check_fields(s, d, ("M", "qLD", "qfrc_bias", "cvel"), FACTOR)
  # solve_m / mul_m
  y = np.random.RandomState(0).randn(d.nworld, mjm.nv).astype(np.float32)
  ya, xa, ra = mjw.DeviceArray.from_numpy(y), mjw.DeviceArray.zeros(y.shape), mjw.DeviceArray.zeros(y.shape)
  mjw.solve_m(m, d, xa, ya)
  mjw.mul_m(m, d, ra, xa)
  assert relerr(xa.numpy()[1], s.solve_m(y[1].astype(np.float64))) <= FACTOR
  assert relerr(ra.numpy(), y) <= FACTOR


@pytest.mark.parametrize("solver", [mjw.SolverType.NEWTON, mjw.SolverType.CG])
def test_per_step_parity_resynced(solver):
  """North-star parity: same state in, one step each, qpos/qvel compared; 150 steps along the oracle trajectory."""
  mjm = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  s, m, d = _pair(mjm, nworld=2, nconmax=24, njmax=64, solver=int(solver), warm_steps=0)
  worst_q = worst_v = 0.0
  for i in range(150):
    s.ctrl_noise(i, 0)
    _sync(s, d)
    mjw.step(m, d)
    s.step()
    worst_q = max(worst_q, relerr(d.qpos.numpy()[1], s.qpos))
    worst_v = max(worst_v, relerr(d.qvel.numpy()[1], s.qvel))
  # measured (profiles/round4_parity_report.txt): Newton 2.1e-7 / 4.9e-5, CG 7.9e-7 / 2.0e-4; the float32 twin 1.8e-7 / 4.2e-5 and
  # 8.3e-7 / 1.9e-4 -- bounds at about 2x the floor
  tol_q, tol_v = (5e-7, 1e-4) if solver == mjw.SolverType.NEWTON else (2e-6, 4e-4)
  assert worst_q <= tol_q, worst_q
  assert worst_v <= tol_v, worst_v
  assert (d.overflow.numpy() == 0).all()


def test_per_step_parity_distribution_newton_bit_identical_inputs():
  """The north star's 1e-5 on qvel, stated as what float32 can meet (profiles/round6_precision_split.txt, tools/precision_split.py): the oracle
  steps from the SAME float32-rounded state as the engine (bit-identical inputs; the test above hands it the unrounded float64 state), and
  the per-step error is asserted as a DISTRIBUTION: the median and 90 % of the 150 steps are inside 1e-5, a handful of steps around contact
  events are not.  The float32 build of the oracle (CPU twin) shows the same figures -- median 1.9e-6, p90 5.1e-6, 8 steps above 1e-5, max
  2.7e-5 -- and loses its outliers only when FK -> contacts -> efc rows (aref, D and J) are computed in float64 (variants U / V of the report):
  efc.J is float32 in the reference's own Data layout, so no engine behind this API gets below it.  CG has no such statement: the float64
  reference CG at its own tolerance is 2.9e-4 away from its converged solution (same report)."""
  mjm = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  s, m, d = _pair(mjm, nworld=2, nconmax=24, njmax=64, solver=int(mjw.SolverType.NEWTON), warm_steps=0)
  ev, eq = [], []
  for i in range(150):
    s.ctrl_noise(i, 0)
    for name in ("qpos", "qvel", "ctrl", "qacc_warmstart"):
      getattr(s, name)[:] = getattr(s, name).astype(np.float32)
    _sync(s, d)
    mjw.step(m, d)
    s.step()
    eq.append(relerr(d.qpos.numpy()[1], s.qpos))
    ev.append(relerr(d.qvel.numpy()[1], s.qvel))
  ev, eq = np.array(ev), np.array(eq)
  print(f"newton, bit-identical inputs: qvel median {np.median(ev):.1e} p90 {np.percentile(ev, 90):.1e} max {ev.max():.1e} n>1e-5 {(ev > 1e-5).sum()}/150; qpos max {eq.max():.1e}")
  assert eq.max() <= 5e-7, eq.max()
  assert np.median(ev) <= 5e-6 and np.percentile(ev, 90) <= 1e-5, (np.median(ev), np.percentile(ev, 90))
  assert (ev > 1e-5).sum() <= 16 and ev.max() <= 6e-5, ((ev > 1e-5).sum(), ev.max())


# measured worst cases over these runs (tools/parity_report.py, profiles/round4_parity_report.txt, where the float32 twin of the oracle
# shows the same figures: they are the float32 floor of the reference's algorithm): the bounds below are about 2x them.
# qpos floor 1e-2 (rad / m), qvel floor 1e-1 (rad/s / m/s).  CG at the float32 tolerance (1e-6) stops on a different iterate than
# the float64 oracle at the same tolerance, which is what its looser bounds measure; Newton lands inside the same basin.
_ELEM_CASES = [
  ("humanoid", conftest.HUMANOID_XML, mjw.SolverType.NEWTON, 24, 64, 150, 8e-5, 2e-3),
  ("humanoid", conftest.HUMANOID_XML, mjw.SolverType.CG, 24, 64, 150, 2.5e-4, 6e-3),
  ("g1", conftest.G1_XML, mjw.SolverType.NEWTON, 48, 192, 60, 1e-5, 4e-4),
  ("panda", conftest.PANDA_XML, mjw.SolverType.NEWTON, 8, 16, 60, 4e-7, 5e-6),
  ("panda", conftest.PANDA_XML, mjw.SolverType.NEWTON, 1, 5, 60, 1e-6, 1e-5),  # BASELINE configs[3]: nconmax 1, njmax 5
]


@pytest.mark.parametrize("name,xml,solver,nconmax,njmax,nstep,tol_q,tol_v", _ELEM_CASES)
def test_per_step_parity_per_element(name, xml, solver, nconmax, njmax, nstep, tol_q, tol_v):
  """One step from the same state, compared ENTRY BY ENTRY (relative to max(|entry|, floor)) along the oracle's trajectory."""
  mjm = mjw.mjcf.load_xml(xml)
  s, m, d = _pair(mjm, nworld=2, nconmax=nconmax, njmax=njmax, solver=int(solver), warm_steps=0)
  worst_q = worst_v = 0.0
  for i in range(nstep):
    if mjm.nu:
      s.ctrl_noise(i, 0)
    _sync(s, d)
    mjw.step(m, d)
    s.step()
    worst_q = max(worst_q, conftest.relerr_elem(d.qpos.numpy()[1], s.qpos, 1e-2))
    worst_v = max(worst_v, conftest.relerr_elem(d.qvel.numpy()[1], s.qvel, 1e-1))
  assert worst_q <= tol_q, worst_q
  assert worst_v <= tol_v, worst_v


@pytest.mark.parametrize("xml,njmax", [(conftest.PENDULA_XML, 32), (conftest.FREE_BODIES_XML, 96), (conftest.PILE_XML, 96),
                                       (conftest.SPHERE_CYLINDER_XML, 96)])
def test_small_models_forward_and_step(xml, njmax):
  """ball/slide/hinge limits, frictionloss, springs, position actuators; box/cylinder/ellipsoid/sphere/capsule contacts."""
  mjm = mjw.mjcf.from_xml_string(xml)
  s, m, d = _pair(mjm, nworld=2, nconmax=32, njmax=njmax, warm_steps=40, noise=False)
  mjw.forward(m, d)
  s.forward()
  _check_fields(s, d, _SMOOTH_FIELDS, SMOOTH)
  _check_contacts_and_rows(s, d, mjm)
  _check_solution(s, d)
  for _ in range(20):
    _sync(s, d)
    mjw.step(m, d)
    s.step()
    assert relerr(d.qpos.numpy()[0], s.qpos) <= 1e-5
    assert relerr(d.qvel.numpy()[0], s.qvel) <= 2e-3


@pytest.mark.parametrize("solver", ["Newton", "CG"])
@pytest.mark.parametrize("xml,njmax", [(conftest.FREE_BODIES_XML, 96), (conftest.PILE_XML, 96), (conftest.SPHERE_CYLINDER_XML, 96),
                                       (conftest.FREE_BODIES_XML, 48)])
def test_small_models_elliptic_cones(xml, njmax, solver):
  """cone="elliptic" impratio="10" on the contact scenes: rows (condim per contact, friction-row parameters), zones, line search
  and cone Hessian of the register-resident solver against the oracle (reference constraint.py:2698-2704, 4277-4294;
  solver.py:272-517, 2466-2564).  njmax 96 runs the six-rows-per-lane instantiation, 48 the two-rows one."""
  import re
  xml = re.sub(r"<option ([^>]*?)(solver=\"\w+\")?/>", lambda mo: f'<option {mo.group(1)} cone="elliptic" impratio="10" solver="{solver}"/>', xml, count=1)
  mjm = mjw.mjcf.from_xml_string(xml)
  assert mjm.opt.cone == 1 and mjm.opt.impratio == 10.0
  s, m, d = _pair(mjm, nworld=2, nconmax=32, njmax=njmax, warm_steps=40, noise=False)
  mjw.forward(m, d)
  s.forward()
  assert (s.efc_type[: s.nefc] == 7).sum() >= 6
  _check_fields(s, d, _SMOOTH_FIELDS, SMOOTH)
  _check_contacts_and_rows(s, d, mjm)
  # (CG stops at the float32 tolerance 1e-6 on a different iterate than the float64 oracle: measured 2.4e-3 on the pile scene)
  _check_solution(s, d, tol=SOLVE if solver == "Newton" else 3 * SOLVE)
  assert (d.efc.state.numpy()[0, : s.nefc] == s.efc_state[: s.nefc]).mean() > 0.9
  for _ in range(20):
    _sync(s, d)
    mjw.step(m, d)
    s.step()
    assert relerr(d.qpos.numpy()[0], s.qpos) <= 1e-5
    assert relerr(d.qvel.numpy()[0], s.qvel) <= 2e-3
  assert (d.overflow.numpy() == 0).all()


@pytest.mark.parametrize("name,xml,solver,nconmax,njmax,nstep", [
  ("humanoid", conftest.HUMANOID_XML, mjw.SolverType.NEWTON, 24, 64, 100), ("humanoid", conftest.HUMANOID_XML, mjw.SolverType.CG, 24, 64, 100),
  ("g1", conftest.G1_XML, mjw.SolverType.NEWTON, 48, 128, 40)])
def test_per_step_parity_elliptic(name, xml, solver, nconmax, njmax, nstep):
  """The benchmark robots with cone="elliptic": one step from the same state along the oracle trajectory (humanoid: 32 lanes per
  world, G1 with 35 dofs: 64 lanes per world)."""
  mjm = mjw.mjcf.load_xml(xml)
  mjm.opt.cone = int(mjw.ConeType.ELLIPTIC)
  mjm.opt.impratio = 4.0
  if name == "g1":
    # its file caps the solver at 10 iterations and the line search at 20: float32 bracketing needs more line-search iterations
    # than float64 (measured 2-4 % of world-steps flag LS_ITERATIONS at 20, pyramidal and elliptic alike, none at 50); CG does not
    # converge on this model with elliptic cones within 50 iterations in the float64 oracle either, so only Newton is compared
    mjm.opt.iterations, mjm.opt.ls_iterations = 50, 50
  s, m, d = _pair(mjm, nworld=2, nconmax=nconmax, njmax=njmax, solver=int(solver), warm_steps=0)
  worst_q = worst_v = 0.0
  nell = 0
  for i in range(nstep):
    if mjm.nu:
      s.ctrl_noise(i, 0)
    _sync(s, d)
    mjw.step(m, d)
    s.step()
    nell += int((s.efc_type[: s.nefc] == 7).sum())
    worst_q = max(worst_q, relerr(d.qpos.numpy()[1], s.qpos))
    worst_v = max(worst_v, relerr(d.qvel.numpy()[1], s.qvel))
  assert nell > 3 * nstep
  assert worst_q <= 1e-5, worst_q
  assert worst_v <= 2e-3, worst_v
  assert (d.overflow.numpy() == 0).all()


@pytest.mark.parametrize("warm", [5, 15, 60])
def test_capsule_box_collider(warm):
  """capsule_box (core:1099) on the kernel instantiation that carries the large colliders: capsules lying / standing on a box,
  overhanging and leaning on its edges, and resting on a small free box; contacts, rows, solution and 20 steps."""
  mjm = mjw.mjcf.from_xml_string(conftest.CAPSULE_BOX_XML)
  s, m, d = _pair(mjm, nworld=2, nconmax=48, njmax=160, warm_steps=warm, noise=False)
  assert m.heavy_colliders == 1
  mjw.forward(m, d)
  s.forward()
  pairs = {tuple(int(x) for x in g) for g in s.con_geom[: s.ncon]}
  assert (2, 1) in pairs and (3, 1) in pairs  # capsule geoms 2.. against the table box (geom 1)
  _check_fields(s, d, _SMOOTH_FIELDS, SMOOTH)
  _check_contacts_and_rows(s, d, mjm, dist_atol=5e-7)
  _check_solution(s, d)
  for _ in range(20):
    _sync(s, d)
    mjw.step(m, d)
    s.step()
    assert relerr(d.qpos.numpy()[0], s.qpos) <= 1e-5
    assert relerr(d.qvel.numpy()[0], s.qvel) <= 2e-3


@pytest.mark.parametrize("warm", [5, 30, 60, 80])
def test_box_box_collider(warm):
  """box_box (core:589): flat / rotated / overhanging stacks, a two-box tower, a box on its edge and one tumbling over the
  table rim; contact records, rows, solution and 15 re-synchronised steps against the oracle."""
  mjm = mjw.mjcf.from_xml_string(conftest.BOX_BOX_XML)
  s, m, d = _pair(mjm, nworld=2, nconmax=64, njmax=192, warm_steps=warm, noise=False)
  assert m.heavy_colliders == 1
  mjw.forward(m, d)
  s.forward()
  assert s.ncon >= 18
  _check_fields(s, d, _SMOOTH_FIELDS, SMOOTH)
  # Boxes resting flat on each other: which corner the clipping starts from depends on the sign of a tilt of ~1e-9 rad, so
  # float32 and float64 may list the SAME contacts of a pair in a different order (measured: identical physics, qpos to
  # 1e-7 over 100 steps).  Contacts are therefore compared per geom pair as sets, and the solution through the
  # order-independent outputs.
  ncon, adr = int(d.ws_ncon.numpy()[-1]), int(d.ws_conadr.numpy()[-1])
  assert ncon == s.ncon and int(d.nefc.numpy()[-1]) == s.nefc
  gg = d.contact.geom.numpy()[adr : adr + ncon]
  np.testing.assert_array_equal(gg, s.con_geom[:ncon])
  gd, gp, gf = (getattr(d.contact, k).numpy()[adr : adr + ncon] for k in ("dist", "pos", "frame"))
  for pair in {tuple(g) for g in gg.tolist()}:
    idx = np.flatnonzero((gg == pair).all(axis=1))
    og = idx[np.lexsort(np.round(gp[idx], 4).T[::-1])]
    oo = idx[np.lexsort(np.round(s.con_pos[idx], 4).T[::-1])]
    np.testing.assert_allclose(gd[og], s.con_dist[oo], atol=1e-6)
    np.testing.assert_allclose(gp[og], s.con_pos[oo], atol=2e-6)
    np.testing.assert_allclose(gf[og].reshape(-1, 9)[:, :3], s.con_frame[oo][:, :3], atol=1e-5)
  _check_fields(s, d, ("qacc_smooth",), FACTOR)
  _check_fields(s, d, ("qacc", "qfrc_constraint"), SOLVE)
  for _ in range(15):
    _sync(s, d)
    mjw.step(m, d)
    s.step()
    assert relerr(d.qpos.numpy()[0], s.qpos) <= 1e-5
    assert relerr(d.qvel.numpy()[0], s.qvel) <= 3e-3


def test_mocap_bodies():
  """Mocap bodies (smooth.py:104-108): static bodies posed per world through Data.mocap_pos / mocap_quat; a tray that rises and
  tilts under a ball and a crate, a pole swept against a hinged arm."""
  mjm = mjw.mjcf.from_xml_string(conftest.MOCAP_XML)
  assert mjm.nmocap == 2
  s, m, d = _pair(mjm, nworld=3, nconmax=32, njmax=96, warm_steps=10, noise=False)
  assert d.mocap_pos.shape == (3, 2, 3) and d.mocap_quat.shape == (3, 2, 4)
  np.testing.assert_allclose(d.mocap_pos.numpy()[1], [[0, 0, 0.3], [0.6, 0, 0.2]], atol=1e-7)
  nm = mjw._npmath
  for i in range(60):
    tilt = 0.002 * i
    s.mocap_pos[0] = [0.0, 0.0, 0.3 + 0.0004 * i]
    s.mocap_quat[0] = nm.axis_angle_to_quat(np.array([0.0, 1.0, 0.0]), tilt)
    s.mocap_pos[1] = [0.6 - 0.002 * i, 0.0, 0.2]
    d.mocap_pos.assign(np.tile(s.mocap_pos.astype(np.float32), (3, 1, 1)))
    d.mocap_quat.assign(np.tile(s.mocap_quat.astype(np.float32), (3, 1, 1)))
    _sync(s, d)
    if i % 20 == 0:
      s.forward()
      mjw.forward(m, d)
      _check_fields(s, d, ("xpos", "xquat", "geom_xpos", "geom_xmat"), SMOOTH)
      _check_contacts_and_rows(s, d, mjm, dist_atol=5e-7)
    mjw.step(m, d)
    s.step()
    assert relerr(d.qpos.numpy()[1], s.qpos) <= 1e-5
    assert relerr(d.qvel.numpy()[1], s.qvel) <= 3e-3
  assert s.ncon >= 5  # ball and crate on the tray (the pole has pushed the arm aside)
  mjw.reset_data(m, d)
  np.testing.assert_allclose(d.mocap_pos.numpy()[2], [[0, 0, 0.3], [0.6, 0, 0.2]], atol=1e-7)
  # keyframe mocap poses (mpos / mquat) reach the device through reset_data_keyframe
  mk = mjw.mjcf.from_xml_string(conftest.MOCAP_XML.replace("</mujoco>", '<keyframe><key mpos="0 0 .5  .7 0 .2"/></keyframe></mujoco>'))
  mm = mjw.put_model(mk)
  dd = mjw.make_data(mk, nworld=2, nconmax=32, njmax=96)
  mjw.reset_data_keyframe(mm, dd, 0)
  np.testing.assert_allclose(dd.mocap_pos.numpy()[1], [[0, 0, 0.5], [0.7, 0, 0.2]], atol=1e-7)
  mjw.kinematics(mm, dd)
  np.testing.assert_allclose(dd.xpos.numpy()[1, 1], [0, 0, 0.5], atol=1e-7)


def test_explicit_contact_pairs():
  """<contact><pair> (io.py:575-590, collision_core.py contact_params with pairid >= 0): a pair the filters would drop, a pair
  overriding condim / margin / gap / friction / solref / solimp, and unspecified attributes falling back to the geom mix."""
  mjm = mjw.mjcf.from_xml_string(conftest.EXPLICIT_PAIR_XML)
  assert mjm.npair == 3
  pairs, pairid = mjw.io.geom_pairs_with_ids(mjm)
  assert sorted(pairid[pairid >= 0].tolist()) == [0, 1, 2] and (pairid == -1).sum() == len(pairs) - 3
  s, m, d = _pair(mjm, nworld=2, nconmax=16, njmax=48, warm_steps=30, noise=False)
  assert m.nexplicit == 3
  mjw.forward(m, d)
  s.forward()
  assert s.ncon >= 4
  _check_fields(s, d, _SMOOTH_FIELDS, SMOOTH)
  _check_contacts_and_rows(s, d, mjm, dist_atol=5e-7)
  _check_solution(s, d)
  dims = {tuple(int(x) for x in g): int(c) for g, c in zip(s.con_geom[: s.ncon], s.con_dim[: s.ncon])}
  assert dims[(0, 1)] == 1 and dims[(0, 2)] == 4  # the explicit pairs' condim, not the geoms' default 3
  adr = int(d.ws_conadr.numpy()[1])
  k = [tuple(int(x) for x in g) for g in s.con_geom[: s.ncon]].index((0, 2))
  np.testing.assert_allclose(d.contact.includemargin.numpy()[adr + k], 0.02, atol=1e-7)  # (collision_core.py:278: the margin; detection at margin + gap)
  np.testing.assert_allclose(d.contact.solref.numpy()[adr + k], [0.03, 0.8], atol=1e-7)
  for _ in range(20):
    _sync(s, d)
    mjw.step(m, d)
    s.step()
    assert relerr(d.qpos.numpy()[0], s.qpos) <= 1e-5
    assert relerr(d.qvel.numpy()[0], s.qvel) <= 2e-3


ANISO_PAIR_XML = """
<mujoco>
  <option timestep="0.003" cone="{cone}" impratio="{impratio}"/>
  <worldbody>
    <geom name="floor" type="plane" size="0 0 .05"/>
    <body name="sled" pos="0 0 .0495"><freejoint/><geom name="sled" type="box" size=".15 .1 .05"/></body>
    <body name="puck" pos=".6 0 .029"><freejoint/><geom name="puck" type="cylinder" size=".08 .03"/></body>
    <body name="ball" pos="-.6 0 .079"><freejoint/><geom name="ball" type="sphere" size=".08"/></body>
  </worldbody>
  <contact>
    <pair geom1="floor" geom2="sled" condim="3" friction="0.9 0.2 0.01 0.001 0.001" solreffriction="0.05 1.2"/>
    <pair geom1="floor" geom2="puck" condim="4" friction="0.3 0.8 0.03 0.001 0.001"/>
    <pair geom1="floor" geom2="ball" condim="6" friction="0.6 0.6 0.02 0.004 0.0008" solreffriction="0.03 1"/>
  </contact>
  <keyframe><key qvel="1.2 0.9 0 0 0 0.4   0.5 -0.7 0 0 0 3   0.8 0.6 0 3 -2 1"/></keyframe>
</mujoco>
"""


@pytest.mark.parametrize("cone,impratio", [("pyramidal", 1), ("elliptic", 1), ("elliptic", 5)])
def test_anisotropic_pair_friction_and_solreffriction(cone, impratio):
  """Explicit pairs with tangent 1 != tangent 2 and roll 1 != roll 2 friction, and `solreffriction` on the friction rows of elliptic
  contacts (constraint.py:4277-4294; round 3 -- put_model used to reject both): rows, published contact parameters and 40 steps
  against the oracle.  A sled with mu (0.9, 0.2) sliding diagonally turns towards its low-friction axis."""
  mjm = mjw.mjcf.from_xml_string(ANISO_PAIR_XML.format(cone=cone, impratio=impratio))
  s, m, d = _pair(mjm, nworld=2, nconmax=16, njmax=64, warm_steps=0, noise=False)
  mjw.forward(m, d)
  s.forward()
  assert s.ncon >= 6
  _check_contacts_and_rows(s, d, mjm, dist_atol=5e-7)
  _check_solution(s, d)
  adr, n = int(d.ws_conadr.numpy()[1]), s.ncon
  np.testing.assert_allclose(d.contact.friction.numpy()[adr : adr + n], s.con_friction[:n], atol=1e-7)
  np.testing.assert_allclose(d.contact.solreffriction.numpy()[adr : adr + n], s.con_solreffriction[:n], atol=1e-7)
  assert {tuple(np.round(f, 4)) for f in s.con_friction[:n]} >= {(0.9, 0.2, 0.01, 0.001, 0.001), (0.3, 0.8, 0.03, 0.001, 0.001)}
  for _ in range(40):
    _sync(s, d)
    mjw.step(m, d)
    s.step()
    assert relerr(d.qpos.numpy()[0], s.qpos) <= 1e-5
    assert relerr(d.qvel.numpy()[0], s.qvel) <= 3e-3
  v = s.qvel[:2]
  # the plane's contact frame has tangent 1 = y, tangent 2 = -x (math.make_frame of the normal z): y (mu 0.9) is braked harder than x (mu 0.2),
  # the sliding direction turns from 0.75 = 0.9 / 1.2 towards the x axis (measured 0.21 pyramidal, 0.19 elliptic)
  assert abs(v[1]) / max(abs(v[0]), 1e-9) < 0.4


def test_sphere_cylinder_rim_regime():
  """10 steps in, the small sphere still rolls over the cylinder's rim (the 40-step state above only has cap and side)."""
  mjm = mjw.mjcf.from_xml_string(conftest.SPHERE_CYLINDER_XML)
  s, m, d = _pair(mjm, nworld=2, nconmax=32, njmax=96, warm_steps=10, noise=False)
  mjw.forward(m, d)
  s.forward()
  pairs = [tuple(int(x) for x in g) for g in s.con_geom[: s.ncon]]
  assert (3, 1) in pairs and (2, 1) in pairs and (5, 4) in pairs, pairs  # rim, cap, side
  _check_contacts_and_rows(s, d, mjm)
  _check_solution(s, d)


def test_implicitfast_integrator():
  mjm = mjw.mjcf.from_xml_string(conftest.PENDULA_XML)
  s, m, d = _pair(mjm, nworld=2, njmax=32, integrator=int(mjw.IntegratorType.IMPLICITFAST), warm_steps=10, noise=False)
  for _ in range(30):
    _sync(s, d)
    mjw.step(m, d)
    s.step()
    assert relerr(d.qpos.numpy()[0], s.qpos) <= 1e-5
    assert relerr(d.qvel.numpy()[0], s.qvel) <= 1e-3
    assert relerr(d.act.numpy()[0], s.act) <= 1e-5 if mjm.na else True


@pytest.mark.parametrize("solver", [mjw.SolverType.NEWTON, mjw.SolverType.CG])
def test_rk4_integrator(solver):
  """forward.rungekutta4 (forward.py:524): four forwards per step; humanoid per-step parity, then a model with ball / slide /
  hinge joints, activations and contacts (every branch of the state perturbation)."""
  mjm = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  s, m, d = _pair(mjm, nworld=2, nconmax=24, njmax=64, solver=int(solver), integrator=int(mjw.IntegratorType.RK4), warm_steps=5)
  worst_q = worst_v = 0.0
  for i in range(40):
    s.ctrl_noise(5 + i, 0)
    _sync(s, d)
    mjw.step(m, d)
    s.step()
    worst_q = max(worst_q, relerr(d.qpos.numpy()[1], s.qpos))
    worst_v = max(worst_v, relerr(d.qvel.numpy()[1], s.qvel))
  assert worst_q <= 1e-5, worst_q
  assert worst_v <= 1e-3, worst_v
  assert relerr(d.qacc_warmstart.numpy()[1], s.qacc_warmstart) <= SOLVE
  np.testing.assert_allclose(d.time.numpy(), 40 * mjm.opt.timestep, rtol=1e-5)  # one timestep per step, not per evaluation
  mjm = mjw.mjcf.from_xml_string(conftest.PENDULA_XML)
  s, m, d = _pair(mjm, nworld=2, nconmax=32, njmax=32, solver=int(solver), integrator=int(mjw.IntegratorType.RK4), warm_steps=20,
                  noise=False)
  for _ in range(20):
    _sync(s, d)
    mjw.step(m, d)
    s.step()
    assert relerr(d.qpos.numpy()[0], s.qpos) <= 1e-5
    assert relerr(d.qvel.numpy()[0], s.qvel) <= 2e-3
    if mjm.na:
      assert relerr(d.act.numpy()[0], s.act) <= 1e-5


@pytest.mark.parametrize("solver", [mjw.SolverType.NEWTON, mjw.SolverType.CG])
def test_constraint_islands_nv81(solver):
  """nv = 81, three humanoids (27 dofs each): constraint islands = kinematic trees joined by coupling rows (reference island.py at
  tree granularity).  World 0: apart -- three islands of 27 dofs (32-lane kernel); world 1: two humanoids interpenetrate -- an island
  of 54 dofs (64-lane kernel) and one of 27; world 2: all three touch -- one island of 81 dofs, too wide for the register kernels:
  the generic solver.  All in one batch, each against its own float64 oracle world."""
  mjm = mjw.mjcf.from_xml_string(conftest.multi_humanoid_xml(3), assets_dir=os.path.dirname(conftest.HUMANOID_XML))
  mjm.opt.solver = int(solver)
  sims = []
  for w in range(3):
    s = ref.RefSim(mjm, nconmax=100, njmax=192, tolerance=1e-6)
    s.reset(key=0 if mjm.nkey else None)
    if w >= 1:  # second humanoid moved onto the first one: arms and torsos interpenetrate
      s.qpos[28 : 28 + 3] = s.qpos[0:3] + np.array([0.12, 0.05, 0.0])
    if w == 2:  # and the third onto both
      s.qpos[56 : 56 + 3] = s.qpos[0:3] + np.array([-0.1, -0.05, 0.0])
    for _ in range(30):
      s.step()
    sims.append(s)
  m = mjw.put_model(mjm)
  assert m.tree_solve == 1 and m.ntree == 3 and m.isl_nv4 == 7
  d = mjw.put_data(mjm, mjw.MjData(mjm), nworld=3, nconmax=100, njmax=192)
  bt = m.body_treeid.numpy()
  worst = [0.0, 0.0]
  seen = set()
  for i in range(15):
    for name in ("qpos", "qvel", "qacc_warmstart"):
      getattr(d, name).assign(np.stack([getattr(s, name) for s in sims]).astype(np.float32))
    mjw.step(m, d)
    sep, nisl = d.ws_separable.numpy(), d.ws_nisland.numpy()
    for w, s in enumerate(sims):
      s.step()
      comp = list(range(3))  # islands from the oracle's contacts
      for c in range(s.ncon):
        t = sorted({int(bt[mjm.geom_bodyid[g]]) for g in s.con_geom[c]} - {-1})
        if len(t) == 2:
          a, b2 = comp[t[0]], comp[t[1]]
          comp = [min(a, b2) if x in (a, b2) else x for x in comp]
      widest = max(comp.count(x) for x in set(comp)) * 27
      if int(d.ws_ncon.numpy()[w]) != s.ncon:
        continue
      assert int(nisl[w]) == len(set(comp)) and int(sep[w]) == (1 if widest <= 64 else 0), (w, comp, nisl[w], sep[w])
      seen.add((w, widest))
      worst[0] = max(worst[0], relerr(d.qpos.numpy()[w], s.qpos))
      worst[1] = max(worst[1], relerr(d.qvel.numpy()[w], s.qvel))
      assert int(d.solver_niter.numpy()[w]) > 0
  assert (0, 27) in seen and (1, 54) in seen and (2, 81) in seen, seen
  assert worst[0] <= 1e-5 and worst[1] <= 5e-3, worst


@pytest.mark.parametrize("solver", [mjw.SolverType.NEWTON, mjw.SolverType.CG])
def test_three_humanoids_nv81(solver):
  """nv = 81 > 64: the generic LDS solver (csrc/solver_big.hpp) -- the structure of the reference's three_humanoids benchmark
  (benchmarks/humanoid/__init__.py: nconmax 100, njmax 192).  Forward fields, then per-step parity."""
  mjm = mjw.mjcf.from_xml_string(conftest.multi_humanoid_xml(3), assets_dir=os.path.dirname(conftest.HUMANOID_XML))
  assert (mjm.nv, mjm.nbody, mjm.nu) == (81, 49, 63)
  s, m, d = _pair(mjm, nworld=3, nconmax=100, njmax=192, solver=int(solver), warm_steps=60)
  s.forward()
  mjw.forward(m, d)
  assert s.nefc > 64
  _check_fields(s, d, _SMOOTH_FIELDS, SMOOTH)
  _check_fields(s, d, ("qLD", "qLDiagInv"), FACTOR)
  _check_contacts_and_rows(s, d, mjm, dist_atol=1e-6)  # bodies up to 3 m from the origin: float32 eps there is 2.4e-7
  # worlds whose rows each touch one humanoid are solved per (world, tree) (solve_body TREE): every tree runs its own line searches
  # and stops on its own test, so CG -- which stops far from the fixed point at tolerance 1e-6 -- lands on a slightly different iterate
  # than the oracle's joint solve (measured 2.4e-3 on the forces); Newton converges to the same point
  assert m.tree_solve == 1 and int(d.ws_nisland.numpy().min()) >= 1
  _check_solution(s, d, tol=SOLVE if solver == mjw.SolverType.NEWTON else 2 * SOLVE)
  worst_q = worst_v = 0.0
  boundary_steps = 0
  for i in range(40):
    s.ctrl_noise(60 + i, 0)
    _sync(s, d)
    mjw.step(m, d)
    s.forward()
    if int(d.ws_ncon.numpy()[1]) != s.ncon:
      # the three humanoids bounce in lock-step, so contacts cross the detection boundary three at a time; a contact whose
      # distance is at the boundary within float32 resolution is seen by one side only (measured: 2 of 40 steps, everything
      # else agrees to 2e-7).  Such a step must be explained by a boundary contact; the state is re-synchronised afterwards.
      adr, n = int(d.ws_conadr.numpy()[1]), int(d.ws_ncon.numpy()[1])
      gap = min(np.abs(s.con_dist[: s.ncon] - s.con_includemargin[: s.ncon]).min(),
                np.abs(d.contact.dist.numpy()[adr : adr + n] - d.contact.includemargin.numpy()[adr : adr + n]).min())
      assert gap < 2e-6, (i, gap)
      boundary_steps += 1
      s.step()
      continue
    s.step()
    worst_q = max(worst_q, relerr(d.qpos.numpy()[1], s.qpos))
    worst_v = max(worst_v, relerr(d.qvel.numpy()[1], s.qvel))
  assert boundary_steps <= 4
  assert worst_q <= 1e-5, worst_q
  assert worst_v <= 1e-3, worst_v
  q = d.qpos.numpy()
  assert (q == q[0]).all()


def test_golden_forward_fixture():
  g = np.load(os.path.join(conftest.GOLDEN_DIR, "humanoid_oracle_forward.npz"))
  mjm = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  m = mjw.put_model(mjm)
  d = mjw.make_data(mjm, nworld=2, nconmax=24, njmax=64)
  for name in ("qpos", "qvel", "ctrl", "qacc_warmstart"):
    getattr(d, name).assign(np.tile(g["in_" + name].astype(np.float32), (2, 1)))
  mjw.step(m, d)
  for name, tol in (("xpos", SMOOTH), ("xquat", SMOOTH), ("subtree_com", SMOOTH), ("cinert", SMOOTH), ("cdof", SMOOTH), ("M", SMOOTH),
                    ("qfrc_bias", SMOOTH), ("qfrc_passive", SMOOTH), ("qfrc_actuator", SMOOTH), ("qacc_smooth", FACTOR), ("qacc", SOLVE),
                    ("qfrc_constraint", SOLVE)):
    assert relerr(getattr(d, name).numpy()[1].reshape(-1), g[name].reshape(-1)) <= tol, name
  assert int(d.nefc.numpy()[1]) == int(g["nefc"]) and int(d.ws_ncon.numpy()[1]) == int(g["ncon"])
  assert relerr(d.efc.J.numpy()[1, : int(g["nefc"]), :27], g["efc_J"]) <= SMOOTH
  assert relerr(d.qpos.numpy()[1], g["qpos_next"]) <= 1e-5
  assert relerr(d.qvel.numpy()[1], g["qvel_next"]) <= 1e-3


def test_golden_rollout_free_running():
  """400 free-running steps from key 0 with the harness' control noise: world 0 tracks the oracle's golden rollout."""
  g = np.load(os.path.join(conftest.GOLDEN_DIR, "humanoid_oracle_rollout.npz"))
  mjm = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  m = mjw.put_model(mjm)
  d = mjw.make_data(mjm, nworld=4, nconmax=24, njmax=64)
  mjw.reset_data_keyframe(m, d, 0)
  stride = int(g["stride"])
  errs = []
  for i in range(int(g["nstep"])):
    mjw.ctrl_noise(m, d, i)
    mjw.step(m, d)
    if i % stride == 0:
      errs.append(relerr(d.qpos.numpy()[0], g["qpos"][i // stride]))
  assert errs[0] <= 1e-5
  assert errs[5] <= 1e-3  # 100 steps of contact-rich falling: drift stays small before chaos amplifies rounding
  assert np.isfinite(d.qpos.numpy()).all()


def _ks(a, b):
  """two-sample Kolmogorov-Smirnov statistic"""
  both = np.sort(np.concatenate([a, b]))
  return float(np.max(np.abs(np.searchsorted(np.sort(a), both, side="right") / a.size - np.searchsorted(np.sort(b), both, side="right") / b.size)))


def test_free_running_1000_steps_statistics():
  """1000 free-running steps (the benchmark's length) of 192 worlds against 192 float64 oracle worlds with the same per-world control
  noise.  Trajectories are chaotic -- a float32 / float64 pair decorrelates after ~100 contact-rich steps -- so the comparison is
  between DISTRIBUTIONS over worlds at steps 250 / 500 / 1000: root height, torso tilt, speed, contact and row counts
  (two-sample KS at alpha = 0.001, means within four standard errors)."""
  nw, marks = 192, (250, 500, 1000)
  mjm = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  m = mjw.put_model(mjm)
  d = mjw.make_data(mjm, nworld=nw, nconmax=24, njmax=64)
  mjw.reset_data_keyframe(m, d, 0)

  def feats(qpos, qvel, ncon, nefc):
    w, x, y, z = qpos[:, 3], qpos[:, 4], qpos[:, 5], qpos[:, 6]
    up = 1.0 - 2.0 * (x * x + y * y)  # z component of the torso's z axis
    return {"height": qpos[:, 2], "tilt": up, "speed": np.sqrt((qvel ** 2).mean(axis=1)), "ncon": ncon.astype(float), "nefc": nefc.astype(float)}

  gpu = {}
  for i in range(marks[-1]):
    mjw.ctrl_noise(m, d, i)
    mjw.step(m, d)
    if i + 1 in marks:
      gpu[i + 1] = feats(d.qpos.numpy(), d.qvel.numpy(), d.ws_ncon.numpy(), np.minimum(d.nefc.numpy(), d.njmax))
  assert np.isfinite(d.qpos.numpy()).all()
  cpu = {k: {f: np.zeros(nw) for f in ("height", "tilt", "speed", "ncon", "nefc")} for k in marks}
  s = ref.RefSim(mjm, nconmax=24, njmax=64, tolerance=1e-6)
  for w in range(nw):
    s.reset(key=0)
    t = 0
    for k in marks:
      # (ref_rollout restarts its noise step index at 0: advance with explicit steps so the noise sequence continues)
      for i in range(t, k):
        s.ctrl_noise(i, w)
        s.step()
      t = k
      f = feats(s.qpos[None], s.qvel[None], np.array([s.ncon]), np.array([s.nefc]))
      for name in f:
        cpu[k][name][w] = f[name][0]
  crit = 1.95 * np.sqrt(2.0 / nw)  # KS critical value at alpha = 0.001
  for k in marks:
    for name in ("height", "tilt", "speed", "ncon", "nefc"):
      a, b = gpu[k][name], cpu[k][name]
      se = np.sqrt(a.var() / nw + b.var() / nw)
      assert abs(a.mean() - b.mean()) <= 4.0 * se + 1e-3 * (1.0 + abs(b.mean())), (k, name, a.mean(), b.mean(), se)
      if name in ("height", "tilt", "speed"):
        assert _ks(a, b) <= crit, (k, name, _ks(a, b), crit)


def _primal_cost(s, qacc):
  """The solver's objective (Gauss term + constraint costs, reference solver.py:115-272) at qacc, from the oracle's rows."""
  n = s.nefc
  J, D, aref = s.efc_J[:n], s.efc_D[:n], s.efc_aref[:n]
  dq = qacc - s.qacc_smooth
  cost = 0.5 * dq @ (s.dense_M() @ dq)
  x = J @ qacc - aref
  for r in range(n):
    if r < s.ne:
      cost += 0.5 * D[r] * x[r] ** 2
    elif r < s.ne + s.nf:
      f = s.efc_frictionloss[r]
      rf = f / D[r]
      cost += 0.5 * D[r] * x[r] ** 2 if abs(x[r]) < rf else f * (abs(x[r]) - 0.5 * rf)
    elif x[r] < 0:
      cost += 0.5 * D[r] * x[r] ** 2
  return cost


def test_g1_cg_at_the_models_own_iteration_cap():
  """unitree_g1_flat caps the solver at 10 iterations (unitree_g1_mjlab.xml:14-15), where CG has not converged: float32 and float64
  iterates then differ visibly (measured 22 % on qacc), so the statement that can be tested is about the OBJECTIVE -- after the same
  10 iterations the engine's qacc is as good a minimiser as the oracle's (cost within 2.5 %, the reference's own style of solver
  check), and both are far below the cost of the warm start."""
  mjm = mjw.mjcf.load_xml(conftest.G1_XML)
  assert mjm.opt.iterations == 10
  s, m, d = _pair(mjm, nworld=2, nconmax=48, njmax=192, solver=int(mjw.SolverType.CG), warm_steps=0)
  # the float32 twin (the oracle's own sources compiled in float32) takes the same 10 iterations from the same state: its distance to
  # the float64 iterate is what float32 alone does to an unconverged CG, and the engine has to stay within it
  s32 = ref.RefSim(mjm, nconmax=48, njmax=192, tolerance=max(mjm.opt.tolerance, 1e-6), real="f32")
  s32.reset(key=0 if mjm.nkey else None)
  worse = 0
  e_gpu, e_twin = [], []
  for i in range(40):
    s.ctrl_noise(i, 0)
    _sync(s, d)
    for f in ("qpos", "qvel", "act", "ctrl", "qacc_warmstart"):
      if getattr(s, f).size:
        getattr(s32, f)[:] = getattr(s, f)
    mjw.forward(m, d)
    s.forward()
    s32.forward()
    if s.nefc == 0 or int(d.nefc.numpy()[1]) != s.nefc:
      s.step()
      continue
    c_gpu, c_ref, c_warm = _primal_cost(s, d.qacc.numpy()[1].astype(np.float64)), _primal_cost(s, s.qacc), _primal_cost(s, s.qacc_warmstart)
    assert c_gpu <= c_ref + 0.025 * abs(c_ref) + 1e-6, (i, c_gpu, c_ref)
    worse += c_gpu > c_warm
    e_gpu.append(relerr(d.qacc.numpy()[1], s.qacc))
    if s32.nefc == s.nefc:
      e_twin.append(relerr(s32.qacc, s.qacc))
    s.step()
  assert worse == 0
  # measured (profiles/round4_parity_report.txt, 60 steps): engine worst 0.40, twin worst 0.79
  assert len(e_twin) >= 20
  assert max(e_gpu) <= 1.5 * max(e_twin), (max(e_gpu), max(e_twin))
  assert np.median(e_gpu) <= 2.0 * np.median(e_twin) + 1e-4, (np.median(e_gpu), np.median(e_twin))


MANY_ROWS_XML = """
<mujoco>
  <option timestep="0.003" solver="{solver}"><flag eulerdamp="disable"/></option>
  <worldbody>
    <geom name="floor" type="plane" size="0 0 .05"/>
    {bodies}
  </worldbody>
</mujoco>
"""


@pytest.mark.parametrize("solver", ["Newton", "CG"])
def test_njmax_beyond_192_small_model(solver):
  """nv = 60 <= 64 with njmax = 384 (BASELINE configs[4] asks for it): worlds with at most 192 rows run the register-resident
  kernels, worlds beyond go to the generic solver -- a batch with both, each against the oracle."""
  bodies = "".join(f'<body pos="{0.21 * (i % 4):.2f} {0.21 * (i // 4):.2f} .0995"><freejoint/><geom type="box" size=".1 .1 .1" condim="4"/></body>'
                   for i in range(10))  # 10 boxes x 4 contacts x 6 pyramid rows = 240 rows
  mjm = mjw.mjcf.from_xml_string(MANY_ROWS_XML.format(solver=solver, bodies=bodies))
  assert mjm.nv == 60
  sims = []
  for w in range(2):
    s = ref.RefSim(mjm, nconmax=128, njmax=384, tolerance=1e-6)
    s.reset()
    if w == 1:  # only three boxes touch the floor, the others hover: few rows
      s.qpos[7 * 3 + 2 :: 7] += 0.5
    sims.append(s)
  m = mjw.put_model(mjm)
  d = mjw.put_data(mjm, mjw.MjData(mjm), nworld=2, nconmax=128, njmax=384)
  worst = [0.0, 0.0]
  rows = set()
  for i in range(12):
    for name in ("qpos", "qvel", "qacc_warmstart"):
      getattr(d, name).assign(np.stack([getattr(s, name) for s in sims]).astype(np.float32))
    mjw.step(m, d)
    for w, s in enumerate(sims):
      s.step()
      assert int(d.nefc.numpy()[w]) == s.nefc
      rows.add((w, s.nefc > 192))
      worst[0] = max(worst[0], relerr(d.qpos.numpy()[w], s.qpos))
      worst[1] = max(worst[1], relerr(d.qvel.numpy()[w], s.qvel))
      assert int(d.solver_niter.numpy()[w]) > 0
  assert (0, True) in rows and (1, False) in rows, rows
  assert worst[0] <= 1e-5 and worst[1] <= 5e-3, worst
  assert (d.overflow.numpy() == 0).all()


@pytest.mark.parametrize("which", ["humanoid", "three"])
def test_qLD_dense_view(which):
  """The reference packs one dense Cholesky factor per kinematic tree into qLD (io.py:173-211); this engine keeps MuJoCo's sparse
  factor there and hands out the reference-layout copy on request: U^T U must reproduce each tree block of the oracle's M."""
  if which == "humanoid":
    mjm = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  else:
    mjm = mjw.mjcf.from_xml_string(conftest.multi_humanoid_xml(3), assets_dir=os.path.dirname(conftest.HUMANOID_XML))
  s, m, d = _pair(mjm, nworld=2, nconmax=100, njmax=192, warm_steps=5)
  mjw.forward(m, d)
  s.forward()
  q, adr = mjw.qLD_dense(m, d)
  M = s.dense_M()
  for a, n in zip(m.tree_dofadr.numpy(), m.tree_dofnum.numpy()):
    U = q.numpy()[1, adr[a] : adr[a] + n * n].reshape(n, n)
    assert (np.tril(U, -1) == 0).all() and (np.diag(U) > 0).all()
    blk = M[a : a + n, a : a + n]
    assert relerr(U.T @ U, blk) <= 2e-5
    assert relerr(U, np.linalg.cholesky(blk).T) <= 2e-4


def test_efc_J_sparse_view():
  """The reference keeps efc.J in CSR form for nv > 32 (types.py:2021-2070); this engine keeps the dense tile and hands out the CSR
  copy on request: rows, addresses and values must reproduce the dense Jacobian (G1, nv 35)."""
  mjm = mjw.mjcf.load_xml(conftest.G1_XML)
  s, m, d = _pair(mjm, nworld=3, nconmax=48, njmax=128, warm_steps=10)
  mjw.forward(m, d)
  rownnz, rowadr, colind, vals = mjw.efc_J_sparse(m, d)
  assert colind.shape == (3, 1, d.njmax_nnz) and rownnz.shape == (3, d.njmax)
  J = d.efc.J.numpy()
  for w in range(3):
    nefc = int(d.nefc.numpy()[w])
    assert nefc > 10
    dense = np.zeros((nefc, mjm.nv), dtype=np.float32)
    for r in range(nefc):
      a, n = int(rowadr.numpy()[w, r]), int(rownnz.numpy()[w, r])
      cols = colind.numpy()[w, 0, a : a + n]
      assert (np.diff(cols) > 0).all()
      dense[r, cols] = vals.numpy()[w, 0, a : a + n]
      assert (vals.numpy()[w, 0, a : a + n] != 0).all()
    np.testing.assert_array_equal(dense, J[w, :nefc, : mjm.nv])
    assert (rownnz.numpy()[w, nefc:] == 0).all()
  # too small a buffer truncates and flags NJMAX_NNZ instead of writing out of bounds
  mjw.efc_J_sparse(m, d, njmax_nnz=16)
  assert (d.overflow.numpy() & int(mjw.OverflowType.NJMAX_NNZ)).all()


def test_ctrl_noise_matches_oracle():
  mjm = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  s, m, d = _pair(mjm, nworld=64, nconmax=24, njmax=64, warm_steps=0)
  d.world_offset = 1000  # global world ids 1000..1063
  for step in range(3):
    mjw.ctrl_noise(m, d, step)
  for w in (0, 17, 63):
    s.ctrl[:] = 0
    for step in range(3):
      s.ctrl_noise(step, 1000 + w)
    np.testing.assert_allclose(d.ctrl.numpy()[w], s.ctrl, atol=2e-7)


def test_large_batch_properties():
  """BASELINE size (8192 worlds): size-independent invariants instead of an oracle comparison."""
  mjm = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  m = mjw.put_model(mjm)
  nworld = 8192
  d = mjw.make_data(mjm, nworld=nworld, nconmax=24, njmax=64)
  mjw.reset_data_keyframe(m, d, 0)
  mjw.forward(m, d)
  q = d.qacc.numpy()
  assert (q == q[0]).all()  # identical worlds -> bitwise identical results regardless of block / XCD placement
  ncon = d.ws_ncon.numpy()
  assert int(d.nacon.numpy()[0]) == int(ncon.sum()) and (ncon == ncon[0]).all()
  wid = d.contact.worldid.numpy()[: int(d.nacon.numpy()[0])]
  assert (np.bincount(wid, minlength=nworld) == ncon).all()
  adr = d.ws_conadr.numpy()
  assert len(set(adr.tolist())) == nworld and (wid[adr] == np.arange(nworld)).all()  # disjoint contiguous blocks
  for i in range(40):
    mjw.ctrl_noise(m, d, i)
    mjw.step(m, d)
  qpos = d.qpos.numpy()
  assert np.isfinite(qpos).all()
  assert len(np.unique(qpos[:, 2])) > nworld // 2  # noise decorrelates the worlds
  quat = qpos[:, 3:7]
  np.testing.assert_allclose(np.linalg.norm(quat, axis=1), 1.0, atol=1e-5)
  nefc = d.nefc.numpy()
  ovf = d.overflow.numpy()
  assert ((nefc <= 64) | ((ovf & int(mjw.OverflowType.NEFC)) != 0)).all()
  assert (d.solver_niter.numpy() <= mjm.opt.iterations).all()
  np.testing.assert_allclose(d.time.numpy(), 40 * 0.005, rtol=1e-5)
  # run-to-run determinism of the whole pipeline
  d2 = mjw.make_data(mjm, nworld=nworld, nconmax=24, njmax=64)
  mjw.reset_data_keyframe(m, d2, 0)
  for i in range(40):
    mjw.ctrl_noise(m, d2, i)
    mjw.step(m, d2)
  assert (d2.qpos.numpy() == qpos).all()


def test_batched_gravity_and_mass():
  mjm = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  m = mjw.put_model(mjm, batch_sizes={"gravity": 2})
  m.opt.gravity.assign(np.array([[0, 0, -9.81], [0, 0, 0]], dtype=np.float32))
  d = mjw.make_data(mjm, nworld=4, nconmax=24, njmax=64)
  mjw.reset_data_keyframe(m, d, 2)  # in the air, no contacts
  for _ in range(20):
    mjw.step(m, d)
  z = d.qpos.numpy()[:, 2]
  assert z[0] < 2.0 - 0.04 and z[2] == z[0]  # worlds 0,2 fall
  np.testing.assert_allclose(z[1], 2.0, atol=1e-4)  # worlds 1,3 float (zero gravity, springs only move joints)
  assert z[3] == z[1]


@pytest.mark.parametrize("override", [[], ["opt.solver=cg"], ["opt.solver=pgs"], ["opt.integrator=RK4"], ["opt.integrator=implicitfast"]])
def test_graph_replay_matches_eager(override):
  """hipGraph capture of one step (side-stream fork/join for Newton, fused Euler for CG, four forwards for RK4) replays bit for bit."""
  mjm = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  mjw.override_model(mjm, override)
  m = mjw.put_model(mjm)
  da = mjw.make_data(mjm, nworld=256, nconmax=24, njmax=64)
  db = mjw.make_data(mjm, nworld=256, nconmax=24, njmax=64)
  for d in (da, db):
    mjw.reset_data_keyframe(m, d, 0)
  q0 = db.qpos.numpy().copy()
  graph = mjw.StepGraph(m, db)  # the capture's warm-up step must not advance the state
  assert (db.qpos.numpy() == q0).all() and (db.time.numpy() == 0).all()
  for i in range(10):
    mjw.ctrl_noise(m, da, i)
    mjw.ctrl_noise(m, db, i)
    mjw.step(m, da)
    graph.launch()
  torch.cuda.synchronize()
  assert (da.qpos.numpy() == db.qpos.numpy()).all()


def test_overflow_flags():
  mjm = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  m = mjw.put_model(mjm)
  d = mjw.make_data(mjm, nworld=2, nconmax=24, njmax=16)  # squat pose needs 32 rows
  mjw.reset_data_keyframe(m, d, 0)
  mjw.step(m, d)
  assert (d.overflow.numpy() & int(mjw.OverflowType.NEFC)).all()
  assert np.isfinite(d.qpos.numpy()).all()
  d = mjw.make_data(mjm, nworld=2, nconmax=2, njmax=64)  # only 4 public contact slots for 16 contacts
  mjw.reset_data_keyframe(m, d, 0)
  mjw.step(m, d)
  assert (d.overflow.numpy() & int(mjw.OverflowType.NARROWPHASE)).any()


def test_timed_steps_reports_kernel_classes():
  mjm = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  m = mjw.put_model(mjm)
  d = mjw.make_data(mjm, nworld=512, nconmax=24, njmax=64)
  mjw.reset_data_keyframe(m, d, 0)
  ms, pk = mjw.timed_steps(m, d, 5, per_kernel=True)  # the four launches of the fused step
  assert ms > 0 and len(pk) == len(mjw.KERNEL_NAMES)
  # humanoid.xml: explicit Euler without activations is integrated by the solver launch itself (no integrator launch)
  assert all(pk[mjw.KERNEL_NAMES.index(k)] > 0 for k in ("ctrl_noise", "fwd_pos", "mid", "solve"))
  assert all(pk[mjw.KERNEL_NAMES.index(k)] == 0 for k in ("collision", "make_constraint", "fwd_vel"))
  ms, pk = mjw.timed_steps(m, d, 5, step0=5, per_kernel=True, plain_kernels=True)  # one plain kernel per stage
  assert all(pk[mjw.KERNEL_NAMES.index(k)] > 0 for k in ("fwd_pos", "collision", "make_constraint", "fwd_vel", "solve", "integrate", "other"))
  np.testing.assert_allclose(d.time.numpy(), 10 * 0.005, rtol=1e-5)


@pytest.mark.parametrize("override", [[], ["opt.solver=cg"], ["opt.integrator=RK4"]])
def test_timed_steps_equals_noise_plus_step(override):
  """The benchmark loop (mjh_timed_steps: the control noise rides with the first launch of each fused step) produces bit for
  bit the trajectory of the API calls ctrl_noise + step, and so does its per-kernel profiling pass (noise as its own kernel)."""
  mjm = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  mjw.override_model(mjm, override)
  m = mjw.put_model(mjm)
  da, db, dc = (mjw.make_data(mjm, nworld=300, nconmax=24, njmax=64) for _ in range(3))
  for d in (da, db, dc):
    mjw.reset_data_keyframe(m, d, 0)
  mjw.timed_steps(m, da, 12, step0=3)
  mjw.timed_steps(m, dc, 12, step0=3, per_kernel=True)
  for i in range(12):
    mjw.ctrl_noise(m, db, 3 + i)
    mjw.step(m, db)
  for name in ("qpos", "qvel", "ctrl", "qacc_warmstart"):
    assert (getattr(da, name).numpy() == getattr(db, name).numpy()).all(), name
    assert (getattr(dc, name).numpy() == getattr(db, name).numpy()).all(), name


def test_long_rollout_contact_records_stay_valid():
  """300 noisy steps at BASELINE size: every published contact record stays well-formed and agrees with the
  per-world bookkeeping (catches count/publish disagreement between the two narrowphase passes), staged and fused."""
  mjm = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  mjm.opt.solver = mjw.SolverType.CG
  m = mjw.put_model(mjm)
  nworld = 8192
  d = mjw.make_data(mjm, nworld=nworld, nconmax=24, njmax=64)
  mjw.reset_data_keyframe(m, d, 0)
  for chunk in range(6):
    mjw.timed_steps(m, d, 50, step0=50 * chunk)
    # one staged step (separate launches, serial) on top of the fused ones
    mjw.kinematics(m, d); mjw.com_pos(m, d); mjw.crb(m, d); mjw.factor_m(m, d)
    mjw.collision(m, d)
    mjw.make_constraint(m, d)
    torch.cuda.synchronize()
    n = int(d.nacon.numpy()[0])
    ncon, adr = d.ws_ncon.numpy(), d.ws_conadr.numpy()
    assert n == int(ncon.sum()) <= d.naconmax
    wid = d.contact.worldid.numpy()[:n]
    assert (np.bincount(wid, minlength=nworld) == ncon).all()
    dim = d.contact.dim.numpy()[:n]
    assert np.isin(dim, (1, 3, 4, 6)).all()
    geom = d.contact.geom.numpy()[:n]
    assert ((geom >= 0) & (geom < mjm.ngeom)).all() and (geom[:, 0] != geom[:, 1]).all()
    dist = d.contact.dist.numpy()[:n]
    assert np.isfinite(dist).all() and (np.abs(dist) < 1.0).all()  # no placeholder (inactive filler) records
    frame = d.contact.frame.numpy()[:n].reshape(n, 3, 3)
    np.testing.assert_allclose(np.einsum("nij,nkj->nik", frame, frame), np.broadcast_to(np.eye(3), (n, 3, 3)), atol=1e-4)
    nefc = d.nefc.numpy()
    assert (nefc >= 0).all() and (nefc <= 10 * 24 + 64).all()
    assert np.isfinite(d.qpos.numpy()).all()


@pytest.mark.parametrize("solver", [mjw.SolverType.NEWTON, mjw.SolverType.CG])
def test_g1_forward_and_step_match_oracle(solver):
  """BASELINE configs[2] model (unitree_g1_flat: nv = 35 -> 64 lanes per world in the solver, njmax = 192, implicitfast,
  position actuators, capsule feet with priority/condim mixing, contact excludes): forward fields and per-step parity."""
  mjm = mjw.mjcf.load_xml(conftest.G1_XML)
  assert (mjm.nv, mjm.nu, mjm.ngeom) == (35, 29, 69)
  if solver == mjw.SolverType.CG:
    # the model's 10 iterations leave CG far from converged, where float32 and float64 iterates legitimately differ by
    # ~10 %; the parity statement is about the converged solution
    mjm.opt.iterations, mjm.opt.ls_iterations = 100, 50
  s, m, d = _pair(mjm, nworld=3, nconmax=48, njmax=192, solver=int(solver), warm_steps=10)
  s.forward()
  mjw.forward(m, d)
  assert s.nefc > 32  # feet on the floor
  _check_fields(s, d, _SMOOTH_FIELDS, SMOOTH)
  _check_contacts_and_rows(s, d, mjm)
  _check_fields(s, d, ("qacc_smooth",), FACTOR)
  _check_fields(s, d, ("qacc", "qfrc_constraint"), 5e-3)
  worst_q = worst_v = 0.0
  for i in range(60):
    s.ctrl_noise(10 + i, 0)
    _sync(s, d)
    mjw.step(m, d)
    s.step()
    worst_q = max(worst_q, relerr(d.qpos.numpy()[1], s.qpos))
    worst_v = max(worst_v, relerr(d.qvel.numpy()[1], s.qvel))
  assert worst_q <= 3e-5, worst_q  # measured 1.1e-5 (CG); 8 capsule contacts with 4 pyramid rows each, tolerance 1e-6
  assert worst_v <= 2e-3, worst_v  # Newton: 10 solver / 20 line-search iterations, both sides stop before convergence


def test_g1_large_batch_runs():
  """4096 G1 worlds (the BASELINE configs[2] size) step without NaNs; identical worlds stay bitwise identical."""
  mjm = mjw.mjcf.load_xml(conftest.G1_XML)
  m = mjw.put_model(mjm)
  d = mjw.make_data(mjm, nworld=4096, nconmax=48, njmax=192)
  mjw.reset_data_keyframe(m, d, 0)
  for i in range(20):
    mjw.step(m, d)
  q = d.qpos.numpy()
  assert np.isfinite(q).all() and (q == q[0]).all()
  assert 0.7 < q[0, 2] < 0.85  # still on its feet
  for i in range(20):
    mjw.ctrl_noise(m, d, i)
    mjw.step(m, d)
  assert np.isfinite(d.qpos.numpy()).all()


@pytest.mark.parametrize("solver", [mjw.SolverType.NEWTON, mjw.SolverType.CG])
def test_panda_equality_implicitfast_match_oracle(solver):
  """BASELINE configs[3] model (franka_emika_panda: nv = 9, implicitfast, position/general actuators with affine bias and
  velocity gain, a joint equality coupling the fingers, joint limits, plane-box finger pads): forward + per-step parity."""
  mjm = mjw.mjcf.load_xml(conftest.PANDA_XML)
  assert (mjm.nv, mjm.nu, mjm.neq) == (9, 8, 1)
  mjm.opt.iterations, mjm.opt.ls_iterations = 100, 50
  s, m, d = _pair(mjm, nworld=3, nconmax=8, njmax=16, solver=int(solver), warm_steps=0)
  rng = np.random.default_rng(3)
  s.qpos[:7] = 0.3 * rng.standard_normal(7)
  s.qpos[3] = -1.5
  s.qpos[7:9] = (0.03, 0.01)  # fingers apart: the equality row is active and violated
  s.qvel[:] = 0.2 * rng.standard_normal(9)
  s.ctrl[:] = 0.2 * rng.standard_normal(8)
  _sync(s, d)
  s.forward()
  mjw.forward(m, d)
  assert s.ne == 1 and int(d.ne.numpy()[0]) == 1
  _check_fields(s, d, _SMOOTH_FIELDS, SMOOTH)
  _check_contacts_and_rows(s, d, mjm)
  _check_solution(s, d)
  worst_q = worst_v = 0.0
  for i in range(100):
    s.ctrl_noise(i, 0)
    _sync(s, d)
    mjw.step(m, d)
    s.step()
    worst_q = max(worst_q, relerr(d.qpos.numpy()[1], s.qpos))
    worst_v = max(worst_v, relerr(d.qvel.numpy()[1], s.qvel))
  assert worst_q <= 1e-5, worst_q
  assert worst_v <= 1e-3, worst_v
  assert abs(s.qpos[7] - s.qpos[8]) < 5e-3  # the coupling pulled the fingers together


_D = mjw.DisableBit


@pytest.mark.parametrize("flag", [_D.CONSTRAINT, _D.LIMIT, _D.CONTACT, _D.GRAVITY, _D.CLAMPCTRL, _D.WARMSTART, _D.ACTUATION,
                                  _D.REFSAFE, _D.DAMPER | _D.SPRING, _D.FILTERPARENT, _D.EULERDAMP])
def test_disable_flags_match_oracle(flag):
  """Every option.disableflags bit the path honours (types.py DisableBit), toggled on the humanoid: forward and 25 steps.
  EULERDAMP is set in humanoid.xml; toggling it ENABLES the implicit-damping Euler branch (forward.py:387-417)."""
  mjm = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  mjm.opt.disableflags = int(mjm.opt.disableflags) ^ int(flag)
  njmax = 96 if flag == _D.FILTERPARENT else 64  # more self-collision pairs without the parent filter
  s, m, d = _pair(mjm, nworld=2, nconmax=32, njmax=njmax, solver=int(mjw.SolverType.NEWTON), warm_steps=10)
  s.forward()
  mjw.forward(m, d)
  _check_fields(s, d, _SMOOTH_FIELDS, SMOOTH)
  _check_contacts_and_rows(s, d, mjm)
  _check_fields(s, d, ("qacc_smooth",), FACTOR)
  _check_fields(s, d, ("qacc", "qfrc_constraint"), SOLVE)
  worst_q = worst_v = 0.0
  for i in range(25):
    s.ctrl_noise(10 + i, 0)
    _sync(s, d)
    mjw.step(m, d)
    s.step()
    worst_q = max(worst_q, relerr(d.qpos.numpy()[1], s.qpos))
    worst_v = max(worst_v, relerr(d.qvel.numpy()[1], s.qvel))
  assert worst_q <= 1e-5, worst_q
  assert worst_v <= 1e-3, worst_v


def test_disable_equality_and_frictionloss():
  """EQUALITY on the Panda (the finger coupling row disappears) and FRICTIONLOSS on the pendula model."""
  for xml, flag, attr in ((conftest.PANDA_XML, _D.EQUALITY, "ne"), (conftest.PENDULA_XML, _D.FRICTIONLOSS, "nf")):
    mjm = mjw.mjcf.load_xml(xml) if os.path.exists(str(xml)) else mjw.mjcf.from_xml_string(xml)
    base = ref.RefSim(mjm, nconmax=8, njmax=32)
    base.qpos[:] = mjm.qpos0
    base.qvel[:] = 0.3
    base.forward()
    assert getattr(base, attr) >= 1
    mjm.opt.disableflags = int(mjm.opt.disableflags) | int(flag)
    s, m, d = _pair(mjm, nworld=2, nconmax=8, njmax=32, solver=int(mjw.SolverType.NEWTON), warm_steps=0)
    s.qvel[:] = 0.3
    _sync(s, d)
    s.forward()
    mjw.forward(m, d)
    assert getattr(s, attr) == 0 and int(getattr(d, attr).numpy()[0]) == 0
    _check_contacts_and_rows(s, d, mjm)
    _check_fields(s, d, ("qacc",), SOLVE)


def test_ragged_world_counts_are_bitwise_consistent():
  """nworld = 1, 5, 13 (partial workgroups and half-empty wavefronts in every role of the composite launches): 20 steps
  from the same state give bitwise the same trajectory in every world as nworld = 8."""
  mjm = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  m = mjw.put_model(mjm)
  ref_q = None
  for nworld in (8, 1, 5, 13):
    d = mjw.make_data(mjm, nworld=nworld, nconmax=24, njmax=64)
    mjw.reset_data_keyframe(m, d, 0)
    d.world_offset = 0
    for i in range(20):
      mjw.step(m, d)  # no control noise: every world follows the same trajectory
    q, v = d.qpos.numpy(), d.qvel.numpy()
    assert np.isfinite(q).all() and (q == q[0]).all() and (v == v[0]).all()
    n = int(d.nacon.numpy()[0])
    assert n == int(d.ws_ncon.numpy().sum()) and (d.contact.worldid.numpy()[:n] == np.repeat(np.arange(nworld), d.ws_ncon.numpy())).all()
    if ref_q is None:
      ref_q = (q[0].copy(), v[0].copy())
    else:
      assert (q[0] == ref_q[0]).all() and (v[0] == ref_q[1]).all()


@pytest.mark.parametrize("name,xml", [("g1", conftest.G1_XML), ("panda", conftest.PANDA_XML)])
def test_golden_forward_fixture_g1_panda(name, xml):
  """Committed oracle fixtures for the BASELINE configs[2] / [3] models (tests/golden/make_golden.py), Newton, converged."""
  g = np.load(os.path.join(conftest.GOLDEN_DIR, f"{name}_oracle_forward.npz"))
  mjm = mjw.mjcf.load_xml(xml)
  mjm.opt.iterations, mjm.opt.ls_iterations = 100, 50
  m = mjw.put_model(mjm)
  d = mjw.make_data(mjm, nworld=2, nconmax=int(g["nconmax"]), njmax=int(g["njmax"]))
  for k in ("qpos", "qvel", "ctrl", "qacc_warmstart"):
    getattr(d, k).assign(np.tile(g["in_" + k].astype(np.float32), (2, 1)))
  mjw.step(m, d)
  for k, tol in (("xpos", SMOOTH), ("xquat", SMOOTH), ("subtree_com", SMOOTH), ("cdof", SMOOTH), ("M", SMOOTH), ("qfrc_bias", SMOOTH),
                 ("qfrc_actuator", SMOOTH), ("qacc_smooth", FACTOR), ("qacc", 5e-3), ("qfrc_constraint", 5e-3)):
    assert relerr(getattr(d, k).numpy()[1].reshape(-1), g[k].reshape(-1)) <= tol, k
  assert (int(d.nefc.numpy()[1]), int(d.ws_ncon.numpy()[1]), int(d.ne.numpy()[1])) == (int(g["nefc"]), int(g["ncon"]), int(g["ne"]))
  assert relerr(d.efc.J.numpy()[1, : int(g["nefc"]), : mjm.nv], g["efc_J"]) <= SMOOTH
  assert relerr(d.qpos.numpy()[1], g["qpos_next"]) <= 1e-5
  assert relerr(d.qvel.numpy()[1], g["qvel_next"]) <= 2e-3


@pytest.mark.parametrize("name", ["boxes", "capsule_box", "three_humanoids"])
def test_golden_scene_fixtures(name):
  """HIP path against the committed fixtures of the collider scenes (heavy instantiation) and of three_humanoids.xml (loader
  expansion of <replicate>/<frame>/<attach> + the generic nv > 64 solver); contacts compared per geom pair as sets."""
  from test_oracle import _scene_model

  g = np.load(os.path.join(conftest.GOLDEN_DIR, f"{name}_oracle_forward.npz"))
  mjm = _scene_model(name)
  m = mjw.put_model(mjm)
  d = mjw.make_data(mjm, nworld=2, nconmax=int(g["nconmax"]), njmax=int(g["njmax"]))
  for k in ("qpos", "qvel", "ctrl", "qacc_warmstart"):
    if getattr(d, k).size:
      getattr(d, k).assign(np.tile(g["in_" + k].astype(np.float32), (2, 1)))
  mjw.step(m, d)
  for k, tol in (("xpos", SMOOTH), ("xquat", SMOOTH), ("M", SMOOTH), ("qfrc_bias", SMOOTH), ("qacc_smooth", FACTOR), ("qacc", 5e-3),
                 ("qfrc_constraint", 5e-3)):
    assert relerr(getattr(d, k).numpy()[1].reshape(-1), g[k].reshape(-1)) <= tol, k
  ncon, adr = int(d.ws_ncon.numpy()[1]), int(d.ws_conadr.numpy()[1])
  assert (int(d.nefc.numpy()[1]), ncon) == (int(g["nefc"]), int(g["ncon"]))
  gg, gd, gp = (getattr(d.contact, k).numpy()[adr : adr + ncon] for k in ("geom", "dist", "pos"))
  np.testing.assert_array_equal(gg, g["con_geom"])
  for pair in {tuple(x) for x in gg.tolist()}:
    idx = np.flatnonzero((gg == pair).all(axis=1))
    og = idx[np.lexsort(np.round(gp[idx], 4).T[::-1])]
    oo = idx[np.lexsort(np.round(g["con_pos"][idx], 4).T[::-1])]
    np.testing.assert_allclose(gd[og], g["con_dist"][oo], atol=2e-6)
    np.testing.assert_allclose(gp[og], g["con_pos"][oo], atol=5e-6)
  assert relerr(d.qpos.numpy()[1], g["qpos_next"]) <= 1e-5
  assert relerr(d.qvel.numpy()[1], g["qvel_next"]) <= 2e-3


@pytest.mark.parametrize("integrator", ["Euler", "implicitfast", "RK4"])
def test_step1_step2_and_rungekutta4_stage(integrator):
  """step1 + step2 (forward.py:1384-1415) reproduce step (RK4 falls back to Euler there, as in the reference); for RK4,
  forward + rungekutta4 reproduces step."""
  mjm = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  mjw.override_model(mjm, [f"opt.integrator={integrator}", "opt.solver=newton"])
  m = mjw.put_model(mjm)
  da, db = (mjw.make_data(mjm, nworld=8, nconmax=24, njmax=64) for _ in range(2))
  for d in (da, db):
    mjw.reset_data_keyframe(m, d, 0)
  for i in range(12):
    for d in (da, db):
      mjw.ctrl_noise(m, d, i)
    mjw.step(m, da)
    if integrator == "RK4":
      mjw.forward(m, db)
      mjw.rungekutta4(m, db)
    else:
      mjw.step1(m, db)
      mjw.step2(m, db)
  np.testing.assert_allclose(db.qpos.numpy(), da.qpos.numpy(), rtol=0, atol=2e-6)
  np.testing.assert_allclose(db.qvel.numpy(), da.qvel.numpy(), rtol=0, atol=2e-4)
  assert hasattr(mjw, "fwd_kinematics") and mjw.ObjType.SITE == 6


def test_graph_replay_with_sleep_sensors_and_energy():
  """hipGraph capture of the staged paths: a sleep-enabled model (bookkeeping kernels, two collision passes, masked solve), a model with
  position / velocity / acceleration / force sensors and subtree momenta, and EnableBit.ENERGY replay bit for bit."""
  from tests import test_sensor, test_sleep

  energy_xml = test_sensor.FORCE_XML.replace('<option timestep="0.002"/>', '<option timestep="0.002"><flag energy="enable"/></option>')
  energy_xml = energy_xml.replace("</sensor>", '<accelerometer site="neck"/><subtreeangmom body="pend"/><gyro site="root"/></sensor>')
  for xml, nconmax, njmax in ((test_sleep.PILE_XML, 48, 160), (energy_xml, 16, 64)):
    mjm = mjw.mjcf.from_xml_string(xml)
    m = mjw.put_model(mjm)
    da = mjw.make_data(mjm, nworld=8, nconmax=nconmax, njmax=njmax)
    db = mjw.make_data(mjm, nworld=8, nconmax=nconmax, njmax=njmax)
    for d in (da, db):
      v = d.qvel.numpy()
      v[:, 0] = np.linspace(0.1, 2.0, 8)
      d.qvel.assign(v)
    graph = mjw.StepGraph(m, db)
    for _ in range(60):
      mjw.step(m, da)
      graph.launch()
    torch.cuda.synchronize()
    for f in ("qpos", "qvel", "sensordata", "energy", "tree_asleep", "body_awake"):
      a, b = getattr(da, f).numpy(), getattr(db, f).numpy()
      assert (a == b).all(), f
    assert mjm.nsensor == 0 or np.abs(da.sensordata.numpy()).max() > 0
