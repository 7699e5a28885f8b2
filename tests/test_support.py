"""Support functions of the reference's public API (support.py): contact_force (:445), jac (:581), get_state / set_state (:674, :829).

contact_force: the decoded 6D forces must reproduce the generalized constraint force, sum_c J_c^T f_c == qfrc_constraint restricted to
contact rows (pyramidal and elliptic), and the normal force of a resting box must carry its weight.  jac: J qvel equals the velocity
of the point (finite difference of the kinematics) and matches the oracle's contact Jacobian construction.  get / set_state: round trip.
"""

import numpy as np
import pytest

import mujoco_warp_amd as mjw
from mujoco_warp_amd.device import DeviceArray
from oracle import ref
from tests import test_elliptic

REST_XML = """
<mujoco>
  <option timestep="0.002" cone="{cone}"/>
  <worldbody>
    <geom type="plane" size="5 5 .1"/>
    <body pos="0 0 .0995" euler="0 0 20"><freejoint/><geom type="box" size=".1 .15 .1" mass="2"/></body>
    <body pos="1 0 .079"><freejoint/><geom type="sphere" size=".08" mass="1"/></body>
  </worldbody>
</mujoco>
"""


@pytest.mark.gpu
@pytest.mark.parametrize("cone", ["pyramidal", "elliptic"])
def test_contact_force_carries_the_weight_and_matches_qfrc(cone):
  mjm = mjw.mjcf.from_xml_string(REST_XML.format(cone=cone))
  m = mjw.put_model(mjm)
  d = mjw.make_data(mjm, nworld=2, nconmax=16, njmax=64)
  v = d.qvel.numpy()
  v[1, 0] = 0.3  # world 1: the box slides, friction components appear
  d.qvel.assign(v)
  for _ in range(200):
    mjw.step(m, d)
  mjw.forward(m, d)
  nacon = int(d.nacon.numpy()[0])
  assert nacon >= 10
  ids = DeviceArray.from_numpy(np.arange(nacon, dtype=np.int32))
  f = DeviceArray.zeros((nacon, 6))
  fw = DeviceArray.zeros((nacon, 6))
  mjw.contact_force(m, d, ids, False, f)
  mjw.contact_force(m, d, ids, True, fw)
  f, fw = f.numpy(), fw.numpy()
  world, geom, frame = d.contact.worldid.numpy()[:nacon], d.contact.geom.numpy()[:nacon], d.contact.frame.numpy()[:nacon]
  for w in range(2):
    for g, mass in ((1, 2.0), (2, 1.0)):
      sel = (world == w) & (geom[:, 1] == g)
      # resting: the world-frame forces of a body's contacts add up to its weight (the frame's normal points from the plane to the body)
      total = fw[sel, :3].sum(axis=0)
      assert abs(total[2] - mass * 9.81) < 0.02 * mass * 9.81, (w, g, total)
      assert (f[sel, 0] >= -1e-6).all()
    assert np.allclose(np.einsum("ci,cij->cj", f[world == w, :3], frame[world == w]), fw[world == w, :3], atol=1e-5)
  # the sliding box feels friction against its motion; |tangential| <= mu * normal (mu = 1)
  sel = (world == 1) & (geom[:, 1] == 1)
  assert fw[sel, 0].sum() < 0 or abs(d.qvel.numpy()[1, 0]) < 1e-3
  assert (np.linalg.norm(f[sel, 1:3], axis=1) <= f[sel, 0] * 1.0 + 1e-4).all()


@pytest.mark.gpu
def test_jac_matches_finite_differences():
  mjm = mjw.mjcf.load_xml(__import__("tests.conftest", fromlist=["x"]).HUMANOID_XML)
  m = mjw.put_model(mjm)
  nworld = 3
  d = mjw.make_data(mjm, nworld=nworld, nconmax=24, njmax=64)
  mjw.reset_data_keyframe(m, d, 0)
  rng = np.random.default_rng(2)
  mjw.forward(m, d)
  bodies = np.array([5, 9, 16], dtype=np.int32)
  local = rng.normal(size=(nworld, 3)) * 0.1
  xpos, xmat = d.xpos.numpy(), d.xmat.numpy()
  point = np.stack([xpos[w, bodies[w]] + xmat[w, bodies[w]] @ local[w] for w in range(nworld)]).astype(np.float32)
  jacp, jacr = DeviceArray.zeros((nworld, 3, m.nv)), DeviceArray.zeros((nworld, 3, m.nv))
  mjw.jac(m, d, jacp, jacr, DeviceArray.from_numpy(point), DeviceArray.from_numpy(bodies))
  jp, jr = jacp.numpy(), jacr.numpy()
  # finite differences of the point's position along random velocities (integrated on the oracle side: quaternion joints)
  for w in range(nworld):
    s = ref.RefSim(mjm)
    s.qpos[:] = d.qpos.numpy()[w]
    s.forward()
    qvel = rng.normal(size=mjm.nv)
    h = 1e-6
    s2 = ref.RefSim(mjm)
    s2.qpos[:] = s.qpos
    s2.qvel[:] = qvel
    s2.qacc[:] = 0
    # position update only: qpos += h * qvel on the manifold
    from mujoco_warp_amd import _npmath as nm

    def quat_integrate(q, w, dt):  # q * exp(w dt / 2): the reference's math.quat_integrate (math.py:189)
      a = np.linalg.norm(w) * dt
      dq = np.r_[np.cos(a / 2), np.sin(a / 2) * w / max(np.linalg.norm(w), 1e-30)]
      return nm.quat_normalize(nm.quat_mul(q, dq))

    q = s.qpos.copy()
    for j in range(mjm.njnt):
      qa, da, t = mjm.jnt_qposadr[j], mjm.jnt_dofadr[j], mjm.jnt_type[j]
      if t == 0:
        q[qa : qa + 3] += h * qvel[da : da + 3]
        q[qa + 3 : qa + 7] = quat_integrate(q[qa + 3 : qa + 7], qvel[da + 3 : da + 6], h)
      elif t == 1:
        q[qa : qa + 4] = quat_integrate(q[qa : qa + 4], qvel[da : da + 3], h)
      else:
        q[qa] += h * qvel[da]
    s2.qpos[:] = q
    s2.forward()
    b = bodies[w]
    p0 = s.xpos[b] + s.xmat[b].reshape(3, 3) @ local[w]
    p1 = s2.xpos[b] + s2.xmat[b].reshape(3, 3) @ local[w]
    assert np.allclose(jp[w] @ qvel, (p1 - p0) / h, atol=2e-4), w
    dR = s2.xmat[b].reshape(3, 3) @ s.xmat[b].reshape(3, 3).T
    omega = np.array([dR[2, 1] - dR[1, 2], dR[0, 2] - dR[2, 0], dR[1, 0] - dR[0, 1]]) / (2 * h)
    assert np.allclose(jr[w] @ qvel, omega, atol=2e-4), w
  mjw.jac(m, d, jacp, None, DeviceArray.from_numpy(point), DeviceArray.from_numpy(bodies))  # either output may be omitted
  assert (jacp.numpy() == jp).all()


@pytest.mark.gpu
def test_get_set_state_round_trip():
  mjm = mjw.mjcf.load_xml(__import__("tests.conftest", fromlist=["x"]).HUMANOID_XML)
  m = mjw.put_model(mjm)
  d = mjw.make_data(mjm, nworld=4, nconmax=24, njmax=64)
  mjw.reset_data_keyframe(m, d, 0)
  for _ in range(5):
    mjw.step(m, d)
  sig = int(mjw.State.INTEGRATION)
  n = mjw.state_size(m, sig)
  assert n == 1 + m.nq + m.nv + m.na + m.nv + m.nu + m.nv + 6 * m.nbody + m.neq + 7 * m.nmocap
  st = DeviceArray.zeros((4, n))
  mjw.get_state(m, d, st, sig)
  ref_q, ref_v, ref_t = d.qpos.numpy().copy(), d.qvel.numpy().copy(), d.time.numpy().copy()
  assert np.allclose(st.numpy()[:, 0], ref_t) and np.allclose(st.numpy()[:, 1 : 1 + m.nq], ref_q)
  for _ in range(7):
    mjw.step(m, d)
  after = d.qpos.numpy().copy()
  mjw.set_state(m, d, st, sig, active=np.array([True, False, True, False]))
  assert (d.qpos.numpy()[[0, 2]] == ref_q[[0, 2]]).all() and (d.qpos.numpy()[[1, 3]] == after[[1, 3]]).all()
  mjw.set_state(m, d, st, sig)
  for _ in range(7):  # deterministic replay from the restored state
    mjw.step(m, d)
  assert (d.qpos.numpy() == after).all()
  with pytest.raises(ValueError):
    mjw.get_state(m, d, st, 1 << 20)


ENERGY_XML = """
<mujoco>
  <option timestep="0.002" integrator="RK4"><flag energy="enable" contact="disable"/></option>
  <worldbody>
    <body pos="0 0 1">
      <joint name="h" type="hinge" axis="0 1 0" stiffness="3" springref="20"/><geom type="capsule" fromto="0 0 0 .4 0 0" size=".03"/>
      <body pos=".4 0 0"><joint type="ball" stiffness="2"/><geom type="capsule" fromto="0 0 0 .3 0 0" size=".025"/></body>
    </body>
    <body pos="1 0 1"><freejoint/><geom type="box" size=".1 .07 .05"/></body>
  </worldbody>
</mujoco>
"""


@pytest.mark.gpu
def test_energy_definitions_and_conservation():
  """EnableBit.ENERGY: potential = -sum m g . xipos + spring terms, kinetic = 0.5 v^T M v (dense M of the oracle), and the total of a
  conservative system (no contacts, no damping, RK4) stays put over 300 steps."""
  mjm = mjw.mjcf.from_xml_string(ENERGY_XML)
  m = mjw.put_model(mjm)
  d = mjw.make_data(mjm, nworld=2, nconmax=4, njmax=8)
  rng = np.random.default_rng(4)
  v = d.qvel.numpy()
  v[:] = rng.normal(size=v.shape) * 0.8
  d.qvel.assign(v)
  mjw.forward(m, d)
  e = d.energy.numpy()
  for w in range(2):
    s = ref.RefSim(mjm)
    s.qpos[:], s.qvel[:] = d.qpos.numpy()[w], d.qvel.numpy()[w]
    s.forward()
    M = s.dense_M()
    assert abs(e[w, 1] - 0.5 * s.qvel @ M @ s.qvel) < 1e-5 * max(1.0, e[w, 1])
    grav = -sum(mjm.body_mass[b] * np.dot(mjm.opt.gravity, s.xipos[b]) for b in range(1, mjm.nbody))
    hinge = 0.5 * 3 * (s.qpos[0] - mjm.qpos_spring[0]) ** 2
    assert abs(e[w, 0] - (grav + hinge)) < 2e-5 * abs(grav)  # (the ball joint sits at its spring reference)
  e0 = e.sum(axis=1)
  for _ in range(300):
    mjw.step(m, d)
  e1 = d.energy.numpy().sum(axis=1)
  assert np.abs(e1 - e0).max() < 2e-3 * np.abs(e0).max(), (e0, e1)
  d.energy.zero_()
  mjw.energy_pos(m, d)
  assert np.abs(d.energy.numpy().sum(axis=1) - e1).max() < 0.05  # (recomputed from the post-step state: close, not equal)
