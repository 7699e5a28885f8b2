"""Sensors (reference sensor.py), the subset csrc/sensor.hpp computes: joint / actuator / ball-joint readings, frame position / axes /
quaternion / velocities (optionally relative to a reference frame), velocimeter, gyro, accelerometer, frame accelerations, subtree centre of mass, clock.

The reference's sensor_test.py compares with MuJoCo C at run time (absent here); the float64 oracle is pinned by what the definitions
imply on states with a closed form: joint sensors equal the state, a site on a link spinning at w about a fixed hinge moves at w x r and
its gyro reads the axis in site coordinates, a relative frame position / velocity of a frame to itself vanishes, frame quaternions
compose, cutoffs clamp.  The GPU path is compared with the oracle on a scene carrying every supported sensor.
"""

import numpy as np
import pytest

import conftest

import mujoco_warp_amd as mjw
from mujoco_warp_amd import _npmath as nm
from oracle import ref

SENSOR_XML = """
<mujoco>
  <option timestep="0.004"/>
  <worldbody>
    <geom name="floor" type="plane" size="5 5 .1"/>
    <site name="origin" pos="0 0 0"/>
    <body name="arm" pos="0 0 1">
      <joint name="shoulder" type="hinge" axis="0 0 1" range="-30 30" limited="true" stiffness="2"/>
      <geom name="upper" type="capsule" fromto="0 0 0 .4 0 0" size=".03"/>
      <site name="tip" pos=".4 0 0" euler="0 0 30"/>
      <body name="fore" pos=".4 0 0">
        <joint name="elbow" type="ball"/>
        <geom name="lower" type="capsule" fromto="0 0 0 .3 0 0" size=".025"/>
        <site name="hand" pos=".3 0 0" euler="10 20 30"/>
      </body>
    </body>
    <body name="ball" pos="1 0 .5"><freejoint name="free"/><geom name="sphere" type="sphere" size=".1"/><site name="imu" pos=".02 .01 0" euler="0 45 0"/></body>
  </worldbody>
  <actuator><motor name="m0" joint="shoulder" gear="2"/><position name="p0" joint="shoulder" kp="10"/></actuator>
  <sensor>
    <jointpos name="jp" joint="shoulder"/><jointvel name="jv" joint="shoulder"/>
    <actuatorpos name="ap" actuator="m0"/><actuatorvel name="av" actuator="m0"/><actuatorfrc name="af" actuator="p0"/>
    <ballquat name="bq" joint="elbow"/><ballangvel name="bw" joint="elbow"/>
    <framepos name="fp_tip" objtype="site" objname="tip"/>
    <framepos name="fp_rel" objtype="site" objname="hand" reftype="site" refname="tip"/>
    <framepos name="fp_self" objtype="body" objname="fore" reftype="body" refname="fore"/>
    <framexaxis name="fx" objtype="site" objname="tip"/><frameyaxis name="fy" objtype="xbody" objname="arm"/>
    <framezaxis name="fz" objtype="geom" objname="lower" reftype="xbody" refname="arm"/>
    <framequat name="fq" objtype="site" objname="hand"/><framequat name="fq_rel" objtype="site" objname="hand" reftype="xbody" refname="fore"/>
    <framelinvel name="lv_tip" objtype="site" objname="tip"/><frameangvel name="wv_tip" objtype="site" objname="tip"/>
    <framelinvel name="lv_rel" objtype="site" objname="hand" reftype="site" refname="tip"/>
    <frameangvel name="wv_rel" objtype="site" objname="hand" reftype="xbody" refname="arm"/>
    <framelinvel name="lv_self" objtype="geom" objname="sphere" reftype="geom" refname="sphere"/>
    <magnetometer name="mag" site="imu"/><velocimeter name="vel" site="imu"/><gyro name="gyro" site="imu"/>
    <gyro name="gyro_cut" site="imu" cutoff="0.5"/>
    <accelerometer name="acc" site="imu"/><framelinacc name="la" objtype="site" objname="imu"/><frameangacc name="aa" objtype="body" objname="ball"/>
    <framelinacc name="la_tip" objtype="site" objname="tip"/>
    <jointactuatorfrc name="jaf" joint="shoulder"/><jointlimitpos name="jlp" joint="shoulder"/><jointlimitvel name="jlv" joint="shoulder"/>
    <jointlimitfrc name="jlf" joint="shoulder"/><e_potential name="ep"/><e_kinetic name="ek"/>
    <subtreelinvel name="slv" body="arm"/><subtreeangmom name="sam" body="arm"/><subtreeangmom name="sam_ball" body="ball"/>
    <subtreecom name="com" body="arm"/><clock name="t"/>
  </sensor>
</mujoco>
"""


def _read(mjm, data, name):
  i = mjm.sensor_names.index(name)
  return np.asarray(data)[mjm.sensor_adr[i] : mjm.sensor_adr[i] + mjm.sensor_dim[i]]


def _state(sim_or_none, mjm):
  rng = np.random.default_rng(11)
  qpos = mjm.qpos0.copy()
  qpos[0] = 0.7
  qpos[1:5] = nm.quat_normalize(rng.normal(size=4))
  qpos[8:12] = nm.quat_normalize(rng.normal(size=4))
  qvel = rng.normal(size=mjm.nv)
  qvel[0] = 1.3
  return qpos, qvel


def test_oracle_sensor_closed_forms():
  mjm = mjw.mjcf.from_xml_string(SENSOR_XML)
  assert mjm.nsensor == 39 and mjm.nsensordata == int(mjm.sensor_dim.sum())
  s = ref.RefSim(mjm)
  s.qpos[:], s.qvel[:] = _state(s, mjm)
  s.ctrl[:] = [0.3, -0.2]
  s.cd.time = 1.25
  s.forward()
  g = lambda n: _read(mjm, s.sensordata, n)
  w, th = 1.3, 0.7
  assert g("jp")[0] == th and g("jv")[0] == w and g("t")[0] == 1.25
  assert g("ap")[0] == pytest.approx(2 * th) and g("av")[0] == pytest.approx(2 * w)  # gear 2
  assert g("af")[0] == pytest.approx(10 * (-0.2 - th))  # position actuator: kp (ctrl - q)
  assert np.allclose(g("bq"), s.qpos[1:5]) and np.allclose(g("bw"), s.qvel[1:4])
  c, sn = np.cos(th), np.sin(th)
  tip = np.array([0.4 * c, 0.4 * sn, 1.0])
  assert np.allclose(g("fp_tip"), tip) and np.allclose(g("fp_self"), 0, atol=1e-15) and np.allclose(g("lv_self"), 0, atol=1e-15)
  assert np.allclose(g("lv_tip"), np.cross([0, 0, w], tip - [0, 0, 1])) and np.allclose(g("wv_tip"), [0, 0, w])
  assert np.allclose(g("fx"), [np.cos(th + np.pi / 6), np.sin(th + np.pi / 6), 0]) and np.allclose(g("fy"), [-sn, c, 0])
  # relative quantities: hand in the tip frame / the forearm frame
  R_tip = nm.quat_to_mat(nm.quat_mul(s.xquat[1], mjm.site_quat[1]))
  hand = s.xpos[2] + nm.quat_to_mat(s.xquat[2]) @ mjm.site_pos[2]
  assert np.allclose(g("fp_rel"), R_tip.T @ (hand - tip))
  assert np.allclose(g("fq_rel"), mjm.site_quat[2]) and np.allclose(g("fq"), nm.quat_mul(s.xquat[2], mjm.site_quat[2]))
  wb = nm.quat_to_mat(s.xquat[2]) @ s.qvel[1:4]  # the ball joint's angular velocity is given in the child frame
  v_hand = np.cross([0, 0, w], hand - [0, 0, 1]) + np.cross(wb, hand - s.xpos[2])
  assert np.allclose(g("lv_rel"), R_tip.T @ (v_hand - np.cross([0, 0, w], tip - [0, 0, 1]) + np.cross(hand - tip, [0, 0, w])) * 0 + R_tip.T @ (v_hand - np.cross([0, 0, w], hand - [0, 0, 1])), atol=1e-12)
  assert np.allclose(g("wv_rel"), nm.quat_to_mat(s.xquat[1]).T @ wb, atol=1e-12)
  # IMU on the free body: v_site = v + w x r (free joint: linear velocity in world, angular velocity in body coordinates)
  Rb = nm.quat_to_mat(s.xquat[3])
  R_imu = Rb @ nm.quat_to_mat(mjm.site_quat[3])
  w_world = Rb @ s.qvel[7:10]
  assert np.allclose(g("gyro"), R_imu.T @ w_world, atol=1e-12)
  assert np.allclose(g("vel"), R_imu.T @ (s.qvel[4:7] + np.cross(w_world, Rb @ mjm.site_pos[3])), atol=1e-12)
  assert np.allclose(g("gyro_cut"), np.clip(g("gyro"), -0.5, 0.5))
  assert np.allclose(g("mag"), R_imu.T @ np.array([0.0, -0.5, 0.0]), atol=1e-12)
  # joint-level readings: actuator force on the joint (motor gear 2 + position servo), the violated upper limit (0.7 rad > 30 deg), energies
  assert g("jaf")[0] == pytest.approx(2 * 0.3 + 10 * (-0.2 - th))
  assert g("jlp")[0] == pytest.approx(np.deg2rad(30) - th) and g("jlv")[0] == pytest.approx(-w) and g("jlf")[0] > 0
  M = s.dense_M()
  assert g("ek")[0] == pytest.approx(0.5 * s.qvel @ M @ s.qvel, rel=1e-12)
  assert g("ep")[0] == pytest.approx(-sum(mjm.body_mass[b] * np.dot(mjm.opt.gravity, s.xipos[b]) for b in range(1, mjm.nbody)) + 0.5 * 2 * th**2, rel=1e-12)
  # acceleration stage: the free sphere is in free fall (isotropic inertia: no angular acceleration), so a point at r from its centre
  # accelerates at g + w x (w x r) and the accelerometer, which subtracts gravity, reads the centripetal term in site coordinates
  r_imu = Rb @ mjm.site_pos[3]
  cent = np.cross(w_world, np.cross(w_world, r_imu))
  assert np.allclose(g("aa"), 0, atol=1e-9) and np.allclose(g("acc"), R_imu.T @ cent, atol=1e-9) and np.allclose(g("la"), cent, atol=1e-9)
  # subtree momenta: velocity of the arm subtree's centre of mass = sum m v / sum m; angular momentum about it = sum (r x m v + R I R^T w)
  def body_vel(b, wb_world):
    c = s.xipos[b]
    return wb_world, c

  m1, m2 = mjm.body_mass[1], mjm.body_mass[2]
  w1 = np.array([0, 0, w])
  v1 = np.cross(w1, s.xipos[1] - [0, 0, 1])
  w2 = w1 + wb
  v2 = np.cross(w1, s.xpos[2] - [0, 0, 1]) + np.cross(w2, s.xipos[2] - s.xpos[2])
  com = (m1 * s.xipos[1] + m2 * s.xipos[2]) / (m1 + m2)
  assert np.allclose(g("slv"), (m1 * v1 + m2 * v2) / (m1 + m2), atol=1e-12)
  L = np.zeros(3)
  for b, mb, vb, wb_ in ((1, m1, v1, w1), (2, m2, v2, w2)):
    R = s.ximat[b].reshape(3, 3)
    L += np.cross(s.xipos[b] - com, mb * vb) + R @ (mjm.body_inertia[b] * (R.T @ wb_))
  assert np.allclose(g("sam"), L, atol=1e-12)
  Rs = s.ximat[3].reshape(3, 3)
  assert np.allclose(g("sam_ball"), Rs @ (mjm.body_inertia[3] * (Rs.T @ w_world)), atol=1e-12)
  m_arm, m_fore = mjm.body_mass[1], mjm.body_mass[2]
  assert np.allclose(g("com"), (m_arm * s.xipos[1] + m_fore * s.xipos[2]) / (m_arm + m_fore))
  # sensors outside the subset keep their slot (the reference's sensordata layout) and read 0; unknown elements raise
  acc = mjw.mjcf.from_xml_string(SENSOR_XML.replace('<clock name="t"/>', '<force name="acc" site="imu"/><clock name="t"/>'))
  assert acc.nsensordata == mjm.nsensordata + 3 and acc.sensor_adr[-1] == mjm.sensor_adr[-1] + 3
  s2 = ref.RefSim(acc)
  s2.cd.time = 2.0
  s2.forward()
  assert (_read(acc, s2.sensordata, "acc") == 0).all() and _read(acc, s2.sensordata, "t")[0] == 2.0
  with pytest.raises(NotImplementedError):
    mjw.mjcf.from_xml_string(SENSOR_XML.replace('<clock name="t"/>', '<camprojection site="imu" camera="c"/>'))


@pytest.mark.gpu
def test_gpu_sensors_vs_oracle():
  mjm = mjw.mjcf.from_xml_string(SENSOR_XML)
  m = mjw.put_model(mjm)
  d = mjw.make_data(mjm, nworld=3, nconmax=16, njmax=64)
  assert d.sensordata.shape == (3, mjm.nsensordata)
  qpos, qvel = _state(None, mjm)
  sims = []
  q, v, c = d.qpos.numpy(), d.qvel.numpy(), d.ctrl.numpy()
  for w in range(3):
    s = ref.RefSim(mjm, nconmax=16, njmax=64)
    s.qpos[:], s.qvel[:] = qpos, qvel * (1 + 0.3 * w)
    s.ctrl[:] = [0.3 - 0.1 * w, -0.2]
    q[w], v[w], c[w] = s.qpos, s.qvel, s.ctrl
    sims.append(s)
  d.qpos.assign(q)
  d.qvel.assign(v)
  d.ctrl.assign(c)
  for step in range(40):  # through step: the sensors must read the step's INPUT state even where the solver epilogue integrates
    for w, s in enumerate(sims):
      s.qpos[:], s.qvel[:], s.qacc_warmstart[:] = d.qpos.numpy()[w], d.qvel.numpy()[w], d.qacc_warmstart.numpy()[w]
      s.cd.time = float(d.time.numpy()[w])
    mjw.step(m, d)
    sd = d.sensordata.numpy()
    for w, s in enumerate(sims):
      s.step()
      err = np.abs(sd[w] - s.sensordata)
      tol = 2e-5 + 2e-5 * np.abs(s.sensordata)
      for name in ("acc", "la", "aa", "la_tip"):  # accelerations are sums of cancelling terms of size w^2 |r|, g (tens of m/s^2)
        k = mjm.sensor_names.index(name)
        tol[mjm.sensor_adr[k] : mjm.sensor_adr[k] + 3] = 1e-3
      tol[mjm.sensor_adr[mjm.sensor_names.index("jlf")]] = 5e-3 * max(1.0, abs(_read(mjm, s.sensordata, "jlf")[0]))  # (a constraint force: solver tolerance)
      assert (err <= tol).all(), (step, w, mjm.sensor_names[int(np.searchsorted(mjm.sensor_adr, int(err.argmax()), side="right")) - 1])
  class Host:
    pass

  res = Host()
  mjw.get_data_into(res, mjm, d, world_id=1)
  assert (res.sensordata == d.sensordata.numpy()[1]).all() and res.energy.shape == (2,)
  # the stage entry point recomputes from the current state
  d.sensordata.zero_()
  mjw.forward(m, d)
  a = d.sensordata.numpy().copy()
  d.sensordata.zero_()
  mjw.sensor(m, d)
  assert (d.sensordata.numpy() == a).all() and np.abs(a).max() > 0


FORCE_XML = """
<mujoco>
  <option timestep="0.002"/>
  <worldbody>
    <geom type="plane" size="5 5 .1"/>
    <body name="pend" pos="0 0 2">
      <joint name="hinge" type="hinge" axis="0 1 0"/><geom type="capsule" fromto="0 0 0 0 0 -.5" size=".03" mass="1.5"/>
      <site name="root" pos="0 0 0" euler="0 0 40"/>
      <body name="bob" pos="0 0 -.5"><geom type="sphere" size=".06" mass="0.7"/><site name="neck" pos="0 0 0"/></body>
    </body>
    <body name="slab" pos="1 0 .0495"><freejoint/><geom type="box" size=".2 .2 .05" mass="3"/><site name="slab_s" pos="0 0 0"/>
      <site name="pad_all" type="box" size=".25 .25 .08" pos="0 0 0"/><site name="pad_half" type="box" size=".12 .25 .08" pos=".13 0 0"/>
      <site name="pad_off" type="sphere" size=".03" pos="0 0 .3"/><site name="pad_cyl" type="cylinder" size=".5 .1" pos="0 0 -.04"/>
      <body name="load" pos="0 .05 .1"><geom type="sphere" size=".05" mass="0.4" contype="0" conaffinity="0"/><site name="load_s" pos="0 0 0" euler="30 0 0"/></body>
    </body>
  </worldbody>
  <sensor>
    <force name="f_root" site="root"/><torque name="t_root" site="root"/><force name="f_neck" site="neck"/>
    <force name="f_slab" site="slab_s"/><force name="f_load" site="load_s"/><torque name="t_load" site="load_s"/>
    <touch name="touch_all" site="pad_all"/><touch name="touch_half" site="pad_half"/><touch name="touch_off" site="pad_off"/><touch name="touch_cyl" site="pad_cyl"/>
  </sensor>
</mujoco>
"""


def _force_expectations(mjm, g):
  g9 = 9.81
  R_root = nm.quat_to_mat(mjm.site_quat[mjm.sensor_objid[mjm.sensor_names.index("f_root")]])
  return {"f_root": R_root.T @ np.array([0, 0, (1.5 + 0.7) * g9]), "t_root": np.zeros(3), "f_neck": np.array([0, 0, 0.7 * g9]), "f_slab": np.zeros(3)}


def test_oracle_force_torque_statics():
  """rne_postconstraint + force / torque sensors on static configurations: a hanging pendulum's joint carries the weight below it, a bob
  fixed to it is held by its own weight, a slab resting on the floor needs nothing from its free joint (contact forces balance its
  weight and its load's), the load fixed on the slab is held by its weight and no torque about its own centre."""
  mjm = mjw.mjcf.from_xml_string(FORCE_XML)
  s = ref.RefSim(mjm, nconmax=16, njmax=64)
  for _ in range(400):
    s.step()
  s.forward()
  g = lambda n: _read(mjm, s.sensordata, n)
  for name, want in _force_expectations(mjm, g).items():
    assert np.allclose(g(name), want, atol=2e-3), (name, g(name), want)
  R_load = nm.quat_to_mat(nm.quat_mul(s.xquat[4], mjm.site_quat[mjm.sensor_objid[mjm.sensor_names.index("f_load")]]))
  assert np.allclose(g("f_load"), R_load.T @ np.array([0, 0, 0.4 * 9.81]), atol=2e-3) and np.allclose(g("t_load"), 0, atol=2e-3)
  # touch: a pad around the whole slab feels the full weight on it, a pad over its +x half feels the two contacts there (half of it),
  # a small pad above the slab none (the contact normals point away from it), a wide cylinder around the slab's base everything
  wgt = (3 + 0.4) * 9.81
  assert abs(g("touch_all")[0] - wgt) < 5e-3 and abs(g("touch_cyl")[0] - wgt) < 5e-3 and g("touch_off")[0] == 0.0
  assert abs(g("touch_half")[0] - 0.5 * wgt) < 0.35  # (the load sits at y = .05: the four corner forces differ in y, not in x)
  # the bodies' external forces: the slab's contacts carry slab + load
  assert abs(s.cfrc_ext[3][5] - (3 + 0.4) * 9.81) < 5e-3


@pytest.mark.gpu
def test_gpu_force_torque_vs_oracle():
  mjm = mjw.mjcf.from_xml_string(FORCE_XML)
  m = mjw.put_model(mjm)
  assert m.nsensor_frc == 6 and m.nsensor_acc == 10
  d = mjw.make_data(mjm, nworld=2, nconmax=16, njmax=64)
  q = d.qpos.numpy()
  q[1, 0] = 0.6  # world 1: the pendulum swings
  d.qpos.assign(q)
  sims = [ref.RefSim(mjm, nconmax=16, njmax=64) for _ in range(2)]
  for step in range(150):
    for w, s in enumerate(sims):
      s.qpos[:], s.qvel[:], s.qacc_warmstart[:] = d.qpos.numpy()[w], d.qvel.numpy()[w], d.qacc_warmstart.numpy()[w]
    mjw.step(m, d)
    sd = d.sensordata.numpy()
    for w, s in enumerate(sims):
      s.step()
      assert np.abs(sd[w] - s.sensordata).max() < 5e-3, (step, w, sd[w], s.sensordata)  # forces of ~20 N from float32 accelerations
  mjw.forward(m, d)
  mjw.rne_postconstraint(m, d)
  for w, s in enumerate(sims):
    s.qpos[:], s.qvel[:], s.qacc_warmstart[:] = d.qpos.numpy()[w], d.qvel.numpy()[w], d.qacc_warmstart.numpy()[w]
    s.forward()
    s._call("rne_postconstraint")
    for f in ("cacc", "cfrc_int", "cfrc_ext"):
      assert np.abs(getattr(d, f).numpy()[w] - getattr(s, f)).max() < 1e-2, f


@pytest.mark.gpu
def test_gpu_subtree_vel_rne_postconstraint_energy_beyond_32_bodies():
  """The lane-group kernels of the sensor path (csrc/support.hpp: k_subtree_vel, k_rne_postconstraint, k_energy: one lane per body, subtree
  quantities as range sums over depth-first body ids) on a model with more bodies than lanes -- three humanoids, 49 bodies, several trees,
  contacts with the floor -- against the oracle's level-by-level recursions."""
  mjm = mjw.mjcf.from_xml_string(conftest.multi_humanoid_xml(3))
  assert mjm.nbody > 32
  mjm.opt.enableflags |= int(mjw.EnableBit.ENERGY)
  m = mjw.put_model(mjm)
  d = mjw.make_data(mjm, nworld=3, nconmax=48, njmax=192)
  rng = np.random.default_rng(3)
  d.qvel.assign(rng.normal(scale=0.5, size=d.qvel.shape).astype(np.float32))
  for _ in range(60):  # let the humanoids drop onto the floor: contact forces enter cfrc_ext
    mjw.step(m, d)
  mjw.forward(m, d)
  mjw.subtree_vel(m, d)
  mjw.rne_postconstraint(m, d)
  mjw.energy_pos(m, d)
  mjw.energy_vel(m, d)
  assert int(d.ws_ncon.numpy().min()) > 0
  for w in range(3):
    s = ref.RefSim(mjm, nconmax=48, njmax=192)
    s.qpos[:], s.qvel[:], s.qacc_warmstart[:] = d.qpos.numpy()[w], d.qvel.numpy()[w], d.qacc_warmstart.numpy()[w]
    s.forward()
    s._call("subtree_vel")
    s._call("rne_postconstraint")
    for f, tol in (("subtree_linvel", 2e-5), ("subtree_angmom", 2e-4), ("cacc", 5e-3), ("cfrc_ext", 5e-3), ("cfrc_int", 5e-3)):
      a, b = getattr(d, f).numpy()[w], np.asarray(getattr(s, f))
      assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max()), (f, w, np.abs(a - b).max(), np.abs(b).max())
    # Data.energy = (potential, kinetic): kinetic against 0.5 v' M v from the oracle's mass matrix, potential against -sum m g . xipos + springs
    e = d.energy.numpy()[w]
    v = np.asarray(s.qvel, dtype=np.float64)
    kin = 0.5 * v @ np.asarray(s.mul_m(v))
    assert abs(e[1] - kin) <= 1e-4 * max(1.0, abs(kin)), (e, kin)
    pot = -float(np.sum(np.asarray(mjm.body_mass)[:, None] * np.asarray(s.xipos) * np.asarray(mjm.opt.gravity)[None, :]))
    stiff = np.asarray(mjm.jnt_stiffness)
    for j in np.nonzero(stiff)[0]:
      assert int(mjm.jnt_type[j]) in (2, 3)  # slide / hinge springs only in this model
      qa = int(mjm.jnt_qposadr[j])
      pot += 0.5 * stiff[j] * (s.qpos[qa] - mjm.qpos_spring[qa]) ** 2
    assert abs(e[0] - pot) <= 1e-4 * max(1.0, abs(pot)), (e, pot)
