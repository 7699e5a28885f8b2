"""CPU tests of the float64 oracle (oracle/mjref.c).

The oracle is PARITY-UNPINNED against MuJoCo C (no `mujoco` in this environment, no golden vectors in the
reference: SURVEY.md §8c).  These tests pin it with everything that is available instead:
  * MuJoCo-independent physics identities (energy, M = d2T/dv2, gravity torques = -dV/dq, KKT residual),
  * closed-form collision cases with hand-derived answers,
  * the reference's own MuJoCo-independent known-answer helpers (math_test.py-style Halton / quaternion facts),
  * the committed golden rollout (tests/golden/), as a regression anchor for both oracle and HIP path.
"""

import os

import numpy as np
import pytest

import conftest
import mujoco_warp_amd as mjw
from oracle import ref


def _sim(mjm, **kw):
  kw.setdefault("nconmax", 32)
  kw.setdefault("njmax", 96)
  kw.setdefault("tolerance", 1e-6)
  return ref.RefSim(mjm, **kw)


def _generic_state(s, mjm, seed=1, nstep=10):
  rng = np.random.RandomState(seed)
  s.reset(key=0 if mjm.nkey else None)
  s.qvel[:] = 0.3 * rng.randn(mjm.nv)
  s.ctrl[:] = 0.2 * rng.randn(mjm.nu)
  for _ in range(nstep):
    s.step()


def test_halton_known_answers():
  L = ref.lib()
  # van der Corput / Halton sequence (reference util_misc.py:61)
  assert L.ref_halton(1, 2) == 0.5
  assert L.ref_halton(2, 2) == 0.25
  assert L.ref_halton(3, 2) == 0.75
  np.testing.assert_allclose(L.ref_halton(1, 3), 1.0 / 3.0, rtol=1e-6)
  np.testing.assert_allclose(L.ref_halton(5, 3), 2.0 / 3.0 + 1.0 / 9.0, rtol=1e-6)
  assert L.ref_halton(0, 2) == 0.0


def test_mass_matrix_symmetric_pd_and_matches_host(humanoid):
  s = _sim(humanoid)
  _generic_state(s, humanoid)
  s.forward()
  M = s.dense_M()
  assert np.allclose(M, M.T)
  assert np.linalg.eigvalsh(M).min() > 0
  Mh = mjw.mjcf.host_mass_matrix(humanoid, s.qpos.copy())["M"]
  np.testing.assert_allclose(M, Mh, atol=1e-12)


def test_kinetic_energy_identity(humanoid):
  """0.5 v'Mv (CRBA) == sum_b 0.5 cvel_b' I_b cvel_b (per-body spatial inertia)."""
  s = _sim(humanoid)
  _generic_state(s, humanoid)
  s.forward()
  v = s.qvel.copy()
  ke_m = 0.5 * v @ (s.dense_M() - np.diag(humanoid.dof_armature)) @ v
  ke_b = 0.0
  for b in range(1, humanoid.nbody):
    ke_b += 0.5 * s.cvel[b] @ mjw._npmath.inert_vec(s.cinert[b], s.cvel[b])
  np.testing.assert_allclose(ke_m, ke_b, rtol=1e-10)


def test_factor_solve_roundtrip(humanoid):
  s = _sim(humanoid)
  _generic_state(s, humanoid)
  s.forward()
  y = np.random.RandomState(0).randn(humanoid.nv)
  np.testing.assert_allclose(s.mul_m(s.solve_m(y)), y, atol=1e-10)
  np.testing.assert_allclose(s.solve_m(y), np.linalg.solve(s.dense_M(), y), rtol=1e-9)


def test_gravity_bias_equals_potential_gradient(humanoid):
  """At qvel=0, qfrc_bias = dV/dq with V = -sum m g.xipos (finite differences on hinge dofs)."""
  s = _sim(humanoid)
  s.reset(key=1)
  s.qvel[:] = 0
  s.forward()
  bias = s.qfrc_bias.copy()
  g = np.asarray(humanoid.opt.gravity)

  def V():
    s.stage("kinematics")
    return -sum(humanoid.body_mass[b] * g @ s.xipos[b] for b in range(humanoid.nbody))

  eps = 1e-6
  for j in range(1, humanoid.njnt):  # hinge joints
    qa, da = humanoid.jnt_qposadr[j], humanoid.jnt_dofadr[j]
    q0 = s.qpos[qa]
    s.qpos[qa] = q0 + eps
    vp = V()
    s.qpos[qa] = q0 - eps
    vm = V()
    s.qpos[qa] = q0
    np.testing.assert_allclose(bias[da], (vp - vm) / (2 * eps), rtol=2e-5, atol=1e-6)


def test_energy_conservation_free_flight(humanoid):
  """No contacts, no damping/springs/limits/actuation: Euler keeps total energy to O(h)."""
  m = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  m.opt.disableflags |= (1 << 5) | (1 << 6) | (1 << 3) | (1 << 11) | (1 << 4)
  m.opt.timestep = 1e-4
  s = _sim(m)
  s.reset(key=2)
  rng = np.random.RandomState(3)
  s.qvel[:] = 0.5 * rng.randn(m.nv)
  g = np.asarray(m.opt.gravity)

  def energy():
    s.forward()
    v = s.qvel.copy()
    return 0.5 * v @ s.dense_M() @ v - sum(m.body_mass[b] * g @ s.xipos[b] for b in range(m.nbody))

  e0 = energy()
  for _ in range(300):
    s.step()
  e1 = energy()
  assert abs(e1 - e0) < 2e-3 * max(1.0, abs(e0))


@pytest.mark.parametrize("solver", [1, 2])
def test_solver_kkt_and_state_consistency(humanoid, solver):
  s = _sim(humanoid, nconmax=24, njmax=64, solver=solver, tolerance=1e-10, iterations=300)
  s.reset(key=0)
  for i in range(30):
    s.ctrl_noise(i, 0)
    s.step()
  s.forward()
  nefc = s.nefc
  assert nefc > 0 and s.overflow == 0
  J = s.efc_J[:nefc]
  # stationarity of the primal problem: M qacc - qfrc_smooth - J' f = 0
  res = s.dense_M() @ s.qacc - s.qfrc_smooth - J.T @ s.efc_force[:nefc]
  scale = humanoid.stat.meaninertia * humanoid.nv
  assert np.linalg.norm(res) / scale < (1e-6 if solver == 2 else 1e-4)
  # force/state consistency for limit + contact rows
  jaref = J @ s.qacc - s.efc_aref[:nefc]
  for r in range(nefc):
    if jaref[r] >= 0:
      assert s.efc_state[r] == 0 and s.efc_force[r] == 0.0
    else:
      assert s.efc_state[r] == 1
      np.testing.assert_allclose(s.efc_force[r], -s.efc_D[r] * jaref[r], rtol=1e-9)


def test_newton_and_cg_agree(humanoid):
  a, b = _sim(humanoid, nconmax=24, njmax=64, solver=2, tolerance=1e-10), _sim(humanoid, nconmax=24, njmax=64, solver=1, tolerance=1e-10)
  for s in (a, b):
    s.reset(key=0)
    s.forward()
  np.testing.assert_allclose(a.qacc, b.qacc, rtol=1e-4, atol=1e-4)


def _one_geom_model(geom_xml, extra=""):
  return mjw.mjcf.from_xml_string(f"""
<mujoco><worldbody>
  <geom name="floor" type="plane" size="0 0 .05"/>
  <body name="a" pos="0 0 0"><freejoint/>{geom_xml}</body>{extra}
</worldbody></mujoco>""")


def test_collision_plane_sphere_closed_form():
  m = _one_geom_model('<geom type="sphere" size=".1"/>')
  s = _sim(m)
  s.qpos[:] = [0.3, -0.2, 0.08, 1, 0, 0, 0]
  s.stage("kinematics")
  s.stage("collision")
  assert s.ncon == 1
  np.testing.assert_allclose(s.con_dist[0], -0.02, atol=1e-12)
  np.testing.assert_allclose(s.con_pos[0], [0.3, -0.2, -0.01], atol=1e-12)  # midpoint between surfaces
  np.testing.assert_allclose(s.con_frame[0][:3], [0, 0, 1], atol=1e-12)
  s.qpos[2] = 0.11
  s.stage("kinematics")
  s.stage("collision")
  assert s.ncon == 0


def test_collision_plane_capsule_two_contacts():
  m = _one_geom_model('<geom type="capsule" size=".05 .2"/>')
  s = _sim(m)
  q = mjw._npmath.axis_angle_to_quat([0, 1, 0], np.pi / 2)  # lying along x
  s.qpos[:] = [0, 0, 0.04, *q]
  s.stage("kinematics")
  s.stage("collision")
  assert s.ncon == 2
  np.testing.assert_allclose(s.con_dist[:2], [-0.01, -0.01], atol=1e-12)
  np.testing.assert_allclose(sorted(s.con_pos[:2, 0]), [-0.2, 0.2], atol=1e-12)


def test_collision_sphere_sphere_and_capsule_capsule():
  m = mjw.mjcf.from_xml_string("""
<mujoco><worldbody>
  <body name="a"><freejoint/><geom type="sphere" size=".1"/></body>
  <body name="b" pos=".15 0 0"><freejoint/><geom type="sphere" size=".1"/></body>
  <body name="c" pos="0 2 0"><freejoint/><geom type="capsule" size=".05 .3"/></body>
  <body name="d" pos=".08 2 0" euler="90 0 0"><freejoint/><geom type="capsule" size=".05 .3"/></body>
</worldbody></mujoco>""")
  s = _sim(m)
  s.stage("kinematics")
  s.stage("collision")
  assert s.ncon == 2
  np.testing.assert_allclose(s.con_dist[0], 0.15 - 0.2, atol=1e-12)
  np.testing.assert_allclose(s.con_frame[0][:3], [1, 0, 0], atol=1e-12)
  np.testing.assert_allclose(s.con_dist[1], 0.08 - 0.1, atol=1e-9)  # crossed capsules: axis distance - radii
  np.testing.assert_allclose(np.abs(s.con_frame[1][:3]), [1, 0, 0], atol=1e-9)


def test_collision_plane_box_and_sphere_box():
  m = _one_geom_model('<geom type="box" size=".1 .2 .3"/>',
                      '<body name="s" pos="0 0 .69"><freejoint/><geom type="sphere" size=".1"/></body>')
  s = _sim(m)
  s.qpos[2] = 0.295
  s.stage("kinematics")
  s.stage("collision")
  # 4 bottom corners penetrate by 5 mm, the sphere rests 5 mm into the box top (z = .595)
  assert s.ncon == 5
  np.testing.assert_allclose(s.con_dist[:4], -0.005, atol=1e-12)
  np.testing.assert_allclose(s.con_dist[4], 0.69 - 0.1 - 0.595, atol=1e-12)


def test_collision_sphere_cylinder_side_cap_rim():
  """sphere_cylinder (collision_primitive_core.py:388): closed-form distances for the three contact regimes."""
  m = mjw.mjcf.from_xml_string("""
<mujoco><worldbody>
  <body name="c"><freejoint/><geom type="cylinder" size=".2 .3"/></body>
  <body name="s"><freejoint/><geom type="sphere" size=".1"/></body>
</worldbody></mujoco>""")
  s = _sim(m)
  for spos, dist, normal in (
    ((0.28, 0.0, 0.1), 0.28 - 0.3, (-1, 0, 0)),             # side: radial distance - both radii
    ((0.05, 0.0, 0.38), 0.38 - 0.3 - 0.1, (0, 0, -1)),      # cap: height above the top face - sphere radius
    ((0.23, 0.0, 0.34), np.hypot(0.03, 0.04) - 0.1, (-0.6, 0, -0.8)),  # rim: distance to the edge circle - radius
    ((0.05, 0.0, 0.25), -(0.3 - 0.25) - 0.1, (0, 0, -1)),   # centre inside, nearer to the cap
  ):
    s.qpos[:] = [0, 0, 0, 1, 0, 0, 0, *spos, 1, 0, 0, 0]
    s.stage("kinematics")
    s.stage("collision")
    assert s.ncon == 1, spos
    np.testing.assert_allclose(s.con_dist[0], dist, atol=1e-12)
    # frame normal points from geom1 (sphere: the lower type id) to geom2 (cylinder)
    np.testing.assert_allclose(s.con_frame[0][:3], normal, atol=1e-12)
  s.qpos[7:10] = [0.5, 0, 0]
  s.stage("kinematics")
  s.stage("collision")
  assert s.ncon == 0


def test_constraint_row_limit_closed_form():
  """One hinge beyond its upper limit: J=-1, pos = range1 - q, D and aref from the solref/solimp formulas."""
  m = mjw.mjcf.from_xml_string("""
<mujoco><option timestep="0.01"/><worldbody><body><joint name="h" type="hinge" axis="0 1 0" range="-30 30" limited="true"/>
<geom type="capsule" fromto="0 0 0 0 0 -.5" size=".05"/></body></worldbody></mujoco>""")
  s = _sim(m)
  q = np.deg2rad(35.0)
  s.qpos[0] = q
  s.qvel[0] = 0.7
  s.forward()
  assert (s.nefc, s.nl, s.ncon) == (1, 1, 0)
  pos = np.deg2rad(30.0) - q
  np.testing.assert_allclose(s.efc_pos[0], pos, rtol=1e-12)
  np.testing.assert_allclose(s.efc_J[0, 0], -1.0)
  np.testing.assert_allclose(s.efc_vel[0], -0.7)
  # default solref (.02,1), solimp (.9,.95,.001,.5,2): |pos|/width > 1 -> imp = dmax
  imp, tc = 0.95, max(0.02, 2 * 0.01)
  k, b = 1 / (0.95**2 * tc**2), 2 / (0.95 * tc)
  np.testing.assert_allclose(s.efc_aref[0], -k * imp * pos - b * (-0.7), rtol=1e-12)
  np.testing.assert_allclose(s.efc_D[0], 1 / (m.dof_invweight0[0] * (1 - imp) / imp), rtol=1e-12)


def test_golden_rollout_regression(humanoid):
  path = os.path.join(conftest.GOLDEN_DIR, "humanoid_oracle_rollout.npz")
  g = np.load(path)
  s = _sim(humanoid, nconmax=24, njmax=64, tolerance=float(g["tolerance"]))
  s.reset(key=0)
  ok, qp, qv = s.rollout(int(g["nstep"]), worldid=int(g["worldid"]))
  assert ok == int(g["nstep"])
  np.testing.assert_allclose(qp[:: int(g["stride"])], g["qpos"], rtol=0, atol=1e-9)
  np.testing.assert_allclose(qv[:: int(g["stride"])], g["qvel"], rtol=0, atol=1e-8)


def test_pendula_and_pile_models_run():
  for xml, njmax in ((conftest.PENDULA_XML, 32), (conftest.FREE_BODIES_XML, 96), (conftest.PILE_XML, 96)):
    m = mjw.mjcf.from_xml_string(xml)
    s = _sim(m, njmax=njmax)
    s.reset(key=0)
    ok, qp, qv = s.rollout(200, noise_std=-1)
    assert ok == 200 and s.overflow == 0
    assert np.isfinite(qp).all()


def test_g1_model_compiles_and_stands():
  """BASELINE configs[2] model: two <worldbody> sections merge (floor = geom 0), mesh geoms are skipped for collision,
  nv = 35; the oracle keeps the robot on its capsule feet over the first 60 steps from the home keyframe."""
  from mujoco_warp_amd import mjcf
  mjm = mjcf.load_xml(conftest.G1_XML)
  assert (mjm.nq, mjm.nv, mjm.nbody, mjm.ngeom, mjm.nu, mjm.nkey) == (36, 35, 31, 69, 29, 1)
  assert mjm.geom_type[0] == 0 and mjm.opt.integrator == 3 and mjm.opt.iterations == 10
  s = ref.RefSim(mjm, nconmax=48, njmax=192)
  s.reset(key=0)
  s.forward()
  assert s.ncon >= 8 and 32 <= s.nefc <= 192  # both feet (4 capsules each) on the floor
  for i in range(40):
    s.step()
  assert np.isfinite(s.qpos).all() and 0.7 < s.qpos[2] < 0.85 and s.ncon >= 4


def test_panda_joint_equality_couples_the_fingers():
  """BASELINE configs[3] model: <equality><joint> (polycoef 0 1 0 0 0) is an always-active row (ne = 1) that drives
  finger_joint1 - finger_joint2 to zero; its Jacobian row is (+1, -1) on the two finger dofs."""
  from mujoco_warp_amd import mjcf
  mjm = mjcf.load_xml(conftest.PANDA_XML)
  assert (mjm.nq, mjm.nv, mjm.nu, mjm.neq, mjm.opt.integrator) == (9, 9, 8, 1, 3)
  s = ref.RefSim(mjm, nconmax=8, njmax=16)
  s.reset()
  s.qpos[7] = 0.03
  s.forward()
  assert s.ne == 1 and s.efc_type[0] == 0
  np.testing.assert_allclose(s.efc_J[0], [0, 0, 0, 0, 0, 0, 0, 1, -1])
  np.testing.assert_allclose(s.efc_pos[0], 0.03)
  for i in range(150):
    s.step()
  assert np.isfinite(s.qpos).all() and abs(s.qpos[7] - s.qpos[8]) < 1e-4


@pytest.mark.parametrize("name,xml", [("g1", conftest.G1_XML), ("panda", conftest.PANDA_XML)])
def test_golden_forward_regression_g1_panda(name, xml):
  """The committed G1 / Panda fixtures are what the oracle produces today (regression anchor for oracle changes)."""
  from mujoco_warp_amd import mjcf
  g = np.load(os.path.join(conftest.GOLDEN_DIR, f"{name}_oracle_forward.npz"))
  mjm = mjcf.load_xml(xml)
  s = ref.RefSim(mjm, nconmax=int(g["nconmax"]), njmax=int(g["njmax"]), tolerance=float(g["tolerance"]), iterations=100, ls_iterations=50)
  for k in ("qpos", "qvel", "ctrl", "qacc_warmstart"):
    getattr(s, k)[:] = g["in_" + k]
  s.forward()
  assert (s.nefc, s.ncon, s.ne) == (int(g["nefc"]), int(g["ncon"]), int(g["ne"]))
  np.testing.assert_allclose(s.qacc, g["qacc"], rtol=1e-9, atol=1e-9)
  s.step()
  np.testing.assert_allclose(s.qpos, g["qpos_next"], rtol=1e-12, atol=1e-12)


def _scene_model(name):
  if name == "boxes":
    return mjw.mjcf.from_xml_string(conftest.BOX_BOX_XML)
  if name == "capsule_box":
    return mjw.mjcf.from_xml_string(conftest.CAPSULE_BOX_XML)
  return mjw.mjcf.load_xml(os.path.join(os.path.dirname(conftest.HUMANOID_XML), "three_humanoids.xml"))


@pytest.mark.parametrize("name", ["boxes", "capsule_box", "three_humanoids"])
def test_golden_scene_regression(name):
  """Committed fixtures of the box-box / capsule-box scenes and of three_humanoids.xml (regression anchors, make_golden.py)."""
  g = np.load(os.path.join(conftest.GOLDEN_DIR, f"{name}_oracle_forward.npz"))
  s = ref.RefSim(_scene_model(name), nconmax=int(g["nconmax"]), njmax=int(g["njmax"]), tolerance=float(g["tolerance"]))
  for k in ("qpos", "qvel", "ctrl", "qacc_warmstart"):
    getattr(s, k)[:] = g["in_" + k]
  s.forward()
  assert (s.nefc, s.ncon) == (int(g["nefc"]), int(g["ncon"]))
  np.testing.assert_array_equal(s.con_geom[: s.ncon], g["con_geom"])
  np.testing.assert_allclose(s.con_dist[: s.ncon], g["con_dist"], rtol=0, atol=1e-12)
  np.testing.assert_allclose(s.con_pos[: s.ncon], g["con_pos"], rtol=0, atol=1e-12)
  np.testing.assert_allclose(s.qacc, g["qacc"], rtol=1e-9, atol=1e-9)
  s.step()
  np.testing.assert_allclose(s.qpos, g["qpos_next"], rtol=1e-12, atol=1e-12)


_DOUBLE_PENDULUM_XML = """
<mujoco>
  <option timestep="{h}" integrator="{integ}"><flag contact="disable"/></option>
  <worldbody>
    <body name="a" pos="0 0 0">
      <joint name="h1" type="hinge" axis="0 1 0"/>
      <geom type="capsule" fromto="0 0 0 0.4 0 0" size="0.03"/>
      <body name="b" pos="0.4 0 0">
        <joint name="h2" type="hinge" axis="0 1 0"/>
        <geom type="capsule" fromto="0 0 0 0.3 0 0" size="0.03"/>
        <body name="c" pos="0.3 0 0">
          <joint name="b3" type="ball"/>
          <geom type="capsule" fromto="0 0 0 0.2 0.05 0" size="0.02"/>
        </body>
      </body>
    </body>
  </worldbody>
</mujoco>
"""


def test_rk4_is_fourth_order_and_beats_euler():
  """rungekutta4 (forward.py:524): on hinge coordinates halving the timestep cuts the end-state error ~16x; with a ball joint
  the scheme (positions advanced by the weighted mean velocity through the exponential map) is second order, as in MuJoCo,
  and still orders of magnitude better than Euler at the same step."""
  ball = '<body name="c" pos="0.3 0 0">\n          <joint name="b3" type="ball"/>\n          <geom type="capsule" fromto="0 0 0 0.2 0.05 0" size="0.02"/>\n        </body>'
  assert ball in _DOUBLE_PENDULUM_XML

  def end_state(h, integ, xml, qvel, T=0.32):
    mjm = mjw.mjcf.from_xml_string(xml.format(h=h, integ=integ))
    s = _sim(mjm)
    s.qvel[:] = qvel
    for _ in range(int(round(T / h))):
      s.step()
    assert abs(s.time - T) < 1e-9
    return np.concatenate([s.qpos, s.qvel])

  for xml, qvel, order in ((_DOUBLE_PENDULUM_XML.replace(ball, ""), [1.0, -2.0], 4), (_DOUBLE_PENDULUM_XML, [1.0, -2.0, 0.5, 1.5, -1.0], 2)):
    ref_state = end_state(0.0005, "RK4", xml, qvel)
    e1 = np.abs(end_state(0.008, "RK4", xml, qvel) - ref_state).max()
    e2 = np.abs(end_state(0.004, "RK4", xml, qvel) - ref_state).max()
    ee = np.abs(end_state(0.004, "Euler", xml, qvel) - ref_state).max()
    assert 0.7 * 2**order < e1 / e2 < 1.4 * 2**order, (order, e1, e2)
    assert ee > 500.0 * e2, (ee, e2)


def test_rk4_humanoid_step_runs_and_warmstart_is_last_stage(humanoid):
  import copy

  mjm = copy.deepcopy(humanoid)
  mjm.opt.integrator = int(mjw.IntegratorType.RK4)
  s = _sim(mjm, nconmax=24, njmax=64)
  s.reset(key=0)
  for i in range(20):
    s.ctrl_noise(i, 0)
    s.step()
  assert np.isfinite(s.qpos).all() and s.overflow == 0
  np.testing.assert_array_equal(s.qacc_warmstart, s.qacc)  # qacc of the fourth evaluation (forward.py:343)
  assert abs(s.time - 20 * mjm.opt.timestep) < 1e-12


def test_collision_capsule_box_closed_form():
  """capsule_box (collision_primitive_core.py:1099): one sphere at the closest segment point, a second one further along the
  capsule when it lies along a face or an edge.  Box half sizes (.3, .2, .1), capsule radius .05, half length .15."""
  m = mjw.mjcf.from_xml_string("""
<mujoco><worldbody>
  <body name="box"><freejoint/><geom type="box" size=".3 .2 .1"/></body>
  <body name="cap"><freejoint/><geom type="capsule" size=".05 .15"/></body>
</worldbody></mujoco>""")
  s = _sim(m)
  nm = mjw._npmath
  along_x = nm.axis_angle_to_quat(np.array([0, 1.0, 0]), np.pi / 2)
  along_y = nm.axis_angle_to_quat(np.array([1.0, 0, 0]), np.pi / 2)

  def contacts(cpos, cquat):
    s.qpos[:] = [0, 0, 0, 1, 0, 0, 0, *cpos, *cquat]
    s.stage("kinematics")
    s.stage("collision")
    n = s.ncon
    assert (s.con_geom[:n] == [1, 0]).all()  # capsule (type 3) before box (type 6)
    order = np.lexsort(s.con_pos[:n].T[::-1])
    return s.con_dist[:n][order], s.con_pos[:n][order], s.con_frame[:n, :3][order]

  # lying flat on the top face, 1 cm deep: a contact under each end, normal from the capsule into the box
  dist, pos, nrm = contacts([0, 0, 0.14], along_x)
  np.testing.assert_allclose(dist, [-0.01, -0.01], atol=1e-12)
  np.testing.assert_allclose(pos, [[-0.15, 0, 0.095], [0.15, 0, 0.095]], atol=1e-12)
  np.testing.assert_allclose(nrm, [[0, 0, -1]] * 2, atol=1e-12)
  # standing on the top face: one contact under the lower end
  dist, pos, nrm = contacts([0.1, 0.05, 0.29], [1, 0, 0, 0])
  np.testing.assert_allclose(dist, [-0.01], atol=1e-12)
  np.testing.assert_allclose(pos, [[0.1, 0.05, 0.095]], atol=1e-12)
  # overhanging the +x edge: the second contact stops at the edge of the box
  dist, pos, nrm = contacts([0.3, 0, 0.14], along_x)
  np.testing.assert_allclose(pos, [[0.15, 0, 0.095], [0.3, 0, 0.095]], atol=1e-12)
  np.testing.assert_allclose(dist, [-0.01, -0.01], atol=1e-12)
  # lying against the +x face
  dist, pos, nrm = contacts([0.34, 0, 0.05], along_y)
  np.testing.assert_allclose(dist, [-0.01, -0.01], atol=1e-12)
  np.testing.assert_allclose(pos, [[0.295, -0.15, 0.05], [0.295, 0.15, 0.05]], atol=1e-12)
  np.testing.assert_allclose(nrm, [[-1, 0, 0]] * 2, atol=1e-12)
  # parallel to a vertical edge of the box, outside the corner: contacts at the box top and at the capsule's lower end
  dist, pos, nrm = contacts([0.33, 0.23, 0.13], [1, 0, 0, 0])
  np.testing.assert_allclose(dist, [np.hypot(0.03, 0.03) - 0.05] * 2, atol=1e-12)
  np.testing.assert_allclose(pos[:, 2], [-0.02, 0.1], atol=1e-12)
  np.testing.assert_allclose(nrm, [[-np.sqrt(0.5), -np.sqrt(0.5), 0]] * 2, atol=1e-12)
  # separated
  assert len(contacts([1, 1, 1], [1, 0, 0, 0])[0]) == 0


def _sat_penetration(R, p, sa, sb):
  """Minimum overlap over the 15 separating axes of two boxes (A: identity at the origin, B: rotation R at p)."""
  axes = [np.eye(3)[i] for i in range(3)] + [R[:, j] for j in range(3)]
  for i in range(3):
    for j in range(3):
      c = np.cross(np.eye(3)[i], R[:, j])
      if np.linalg.norm(c) > 1e-9:
        axes.append(c / np.linalg.norm(c))
  best = np.inf
  for a in axes:
    ra = np.abs(a) @ sa
    rb = np.abs(R.T @ a) @ sb
    best = min(best, ra + rb - abs(a @ p))
  return best


def test_collision_box_box_closed_form_and_sat_depth():
  """box_box (collision_primitive_core.py:589): stacked / rotated / overhanging / side-by-side / vertex-down cases in closed
  form, then 600 random poses: a contact exists iff the boxes overlap on all 15 separating axes, and the deepest contact
  distance equals minus the minimum overlap (the penetration depth along the axis the algorithm picks)."""
  m = mjw.mjcf.from_xml_string("""
<mujoco><option><flag nativeccd="disable"/></option><worldbody>
  <body name="a"><freejoint/><geom type="box" size=".3 .2 .1"/></body>
  <body name="b"><freejoint/><geom type="box" size=".1 .1 .1"/></body>
</worldbody></mujoco>""")
  s = _sim(m)
  nm = mjw._npmath

  def contacts(bpos, bquat=(1, 0, 0, 0)):
    s.qpos[:] = [0, 0, 0, 1, 0, 0, 0, *bpos, *bquat]
    s.stage("kinematics")
    s.stage("collision")
    n = s.ncon
    order = np.lexsort(s.con_pos[:n].T[::-1])
    return s.con_dist[:n][order], s.con_pos[:n][order], s.con_frame[:n, :3][order]

  dist, pos, nrm = contacts([0, 0, 0.195])  # B resting 5 mm deep on the top face of A: its four bottom corners
  np.testing.assert_allclose(dist, [-0.005] * 4, atol=1e-12)
  np.testing.assert_allclose(pos, [[-0.1, -0.1, 0.0975], [-0.1, 0.1, 0.0975], [0.1, -0.1, 0.0975], [0.1, 0.1, 0.0975]], atol=1e-12)
  np.testing.assert_allclose(nrm, [[0, 0, 1]] * 4, atol=1e-12)  # from A (geom 0) to B (geom 1)
  dist, pos, nrm = contacts([0, 0, 0.195], nm.axis_angle_to_quat(np.array([0, 0, 1.0]), np.pi / 4))
  r = 0.1 * np.sqrt(2)
  np.testing.assert_allclose(pos, [[-r, 0, 0.0975], [0, r, 0.0975], [0, -r, 0.0975], [r, 0, 0.0975]], atol=1e-9)
  dist, pos, nrm = contacts([0.35, 0, 0.195])  # overhanging the +x edge: clipped at x = 0.3
  np.testing.assert_allclose(pos, [[0.25, -0.1, 0.0975], [0.25, 0.1, 0.0975], [0.3, -0.1, 0.0975], [0.3, 0.1, 0.0975]], atol=1e-12)
  dist, pos, nrm = contacts([0.395, 0, 0])  # against the +x face
  np.testing.assert_allclose(dist, [-0.005] * 4, atol=1e-12)
  np.testing.assert_allclose(nrm, [[1, 0, 0]] * 4, atol=1e-12)
  q = nm.quat_mul(nm.axis_angle_to_quat(np.array([0, 1.0, 0]), np.arctan(np.sqrt(2))), nm.axis_angle_to_quat(np.array([0, 0, 1.0]), np.pi / 4))
  dist, pos, nrm = contacts([0, 0, 0.1 + 0.1 * np.sqrt(3) - 0.01], q)  # standing on a vertex, 1 cm deep
  np.testing.assert_allclose(dist, [-0.01], atol=1e-9)
  np.testing.assert_allclose(pos, [[0, 0, 0.095]], atol=1e-9)
  assert len(contacts([1, 1, 1])[0]) == 0

  rng = np.random.default_rng(3)
  sa, sb = np.array([0.3, 0.2, 0.1]), np.array([0.1, 0.1, 0.1])
  nhit = 0
  for _ in range(600):
    q = rng.standard_normal(4)
    q /= np.linalg.norm(q)
    p = np.array([rng.uniform(-0.45, 0.45), rng.uniform(-0.35, 0.35), rng.uniform(-0.25, 0.25)])
    dist, pos, nrm = contacts(p, q)
    pen = _sat_penetration(nm.quat_to_mat(q), p, sa, sb)
    if pen < -1e-9:
      assert len(dist) == 0
    elif len(dist):
      assert pen > -1e-9 and (dist <= 1e-12).all()
      R = nm.quat_to_mat(q)
      if pen > 0.03:
        continue  # deep interpenetration (a box centre inside the other): the clipping heuristics are not meant for it
      for dd, pp, n in zip(dist, pos, nrm):
        assert abs(np.linalg.norm(n) - 1) < 1e-9 and n @ p > -1e-9  # unit normal pointing from A towards B
        # the contact position is the midpoint between the two surfaces along the normal: half the penetration towards B
        # lies in (or on) A, half the penetration towards A lies in (or on) B
        qa, qb = pp - 0.5 * dd * n, R.T @ (pp + 0.5 * dd * n - p)
        # a single contact (vertex-face or a clean edge-edge crossing) is exact; the additional points of a multi-contact
        # edge configuration reuse the edge-edge normal although their distance is a vertex-edge diagonal
        tol = 1e-7 + (0.0 if len(dist) == 1 else 0.3 * abs(dd))
        assert (np.abs(qa) <= sa + tol).all(), (qa, dd)
        assert (np.abs(qb) <= sb + tol).all(), (qb, dd)
      nhit += 1
  assert nhit > 50
