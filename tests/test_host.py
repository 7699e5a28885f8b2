"""CPU tests of the host side: MJCF compiler, the C-ABI library (symbols only, no GPU compute), IO boundary."""

import ctypes
import os
import re

import numpy as np
import pytest

import conftest
import mujoco_warp_amd as mjw
from mujoco_warp_amd import _abi
from mujoco_warp_amd import io


# ---------------------------------------------------------------------------------- MJCF compiler
def test_humanoid_sizes_match_survey(humanoid):
  # nq=28, nv=27, nbody=17, njnt=22, ngeom=20, nu=21, nkey=3, nC=243
  m = humanoid
  assert (m.nq, m.nv, m.nbody, m.njnt, m.ngeom, m.nu, m.nkey) == (28, 27, 17, 22, 20, 21, 3)
  assert int(m.M_rownnz.sum()) == 243
  assert m.opt.timestep == 0.005 and m.opt.integrator == 0 and m.opt.solver == 2
  assert m.opt.disableflags & mjw.DisableBit.EULERDAMP
  # joint ranges are in degrees in the XML (default compiler angle)
  np.testing.assert_allclose(m.jnt_range[1], np.deg2rad([-45, 45]))
  assert m.jnt_limited[1:].all() and not m.jnt_limited[0]
  np.testing.assert_allclose(m.actuator_gear[5, 0], 120.0)
  np.testing.assert_allclose(m.key_qpos[0][:7], [0, 0, 0.596, 0.988015, 0, 0.154359, 0])


def test_capsule_and_sphere_inertia_closed_form():
  m = mjw.mjcf.from_xml_string("""
<mujoco><worldbody>
 <body name="s"><freejoint/><geom type="sphere" size=".1"/></body>
 <body name="c" pos="1 0 0"><freejoint/><geom type="capsule" size=".05 .2"/></body>
 <body name="b" pos="2 0 0"><freejoint/><geom type="box" size=".1 .2 .3" density="500"/></body>
</worldbody></mujoco>""")
  rho = 1000.0
  ms = rho * 4 / 3 * np.pi * 0.1**3
  np.testing.assert_allclose(m.body_mass[1], ms)
  np.testing.assert_allclose(m.body_inertia[1], [0.4 * ms * 0.01] * 3)
  r, h = 0.05, 0.4
  mc, msph = rho * np.pi * r * r * h, rho * 4 / 3 * np.pi * r**3
  np.testing.assert_allclose(m.body_mass[2], mc + msph)
  izz = mc * r * r / 2 + 0.4 * msph * r * r
  ixx = mc * (3 * r * r + h * h) / 12 + 0.4 * msph * r * r + msph * h * (3 * r + 2 * h) / 8
  np.testing.assert_allclose(sorted(m.body_inertia[2]), sorted([ixx, ixx, izz]), rtol=1e-12)
  mb = 500 * 8 * 0.1 * 0.2 * 0.3
  np.testing.assert_allclose(m.body_mass[3], mb)
  np.testing.assert_allclose(sorted(m.body_inertia[3]), sorted(mb / 3 * np.array([0.2**2 + 0.3**2, 0.1**2 + 0.3**2, 0.1**2 + 0.2**2])))


def test_defaults_childclass_and_fromto(humanoid):
  m = humanoid
  g = m.geom_names.index("shin_right")
  np.testing.assert_allclose(m.geom_size[g][:2], [0.049, 0.15])  # class "shin": fromto 0 0 0 0 0 -.3, size .049
  np.testing.assert_allclose(m.geom_pos[g], [0, 0, -0.15])
  np.testing.assert_allclose(m.geom_friction[g], [0.7, 0.005, 0.0001])
  np.testing.assert_allclose(m.geom_solimp[g], [0.9, 0.99, 0.003, 0.5, 2])
  assert m.geom_condim[g] == 1 and m.geom_condim[0] == 3
  j = m.jnt_names.index("knee_right")
  np.testing.assert_allclose(m.jnt_axis[j], [0, -1, 0])
  np.testing.assert_allclose(m.jnt_range[j], np.deg2rad([-160, 2]))
  np.testing.assert_allclose([m.jnt_stiffness[j], m.dof_damping[m.jnt_dofadr[j]], m.dof_armature[m.jnt_dofadr[j]]], [1, 0.2, 0.01])
  j = m.jnt_names.index("abdomen_z")  # class joint_big_stiff inherits damping 5 from joint_big
  np.testing.assert_allclose([m.jnt_stiffness[j], m.dof_damping[m.jnt_dofadr[j]]], [20, 5])
  np.testing.assert_allclose(m.jnt_solimp[j], [0, 0.99, 0.01, 0.5, 2])


def test_set_const(humanoid):
  m = humanoid
  np.testing.assert_allclose(m.body_subtreemass[0], m.body_mass.sum())
  M = mjw.mjcf.host_mass_matrix(m, m.qpos0)["M"]
  np.testing.assert_allclose(m.stat.meaninertia, np.mean(np.diag(M)))
  A = np.diag(np.linalg.inv(M))
  np.testing.assert_allclose(m.dof_invweight0[0:3], np.mean(A[0:3]))
  np.testing.assert_allclose(m.dof_invweight0[3:6], np.mean(A[3:6]))
  np.testing.assert_allclose(m.dof_invweight0[6], A[6])
  assert (m.body_invweight0[1:] > 0).all() and (m.body_invweight0[0] == 0).all()


def test_unsupported_features_raise():
  with pytest.raises(NotImplementedError):
    mjw.mjcf.from_xml_string('<mujoco><worldbody><body><joint/><geom size=".1"/></body></worldbody>'
                             '<tendon><fixed><joint joint="x" coef="1"/></fixed></tendon></mujoco>')
  # (round 3: PGS with elliptic cones and with more than 64 dofs is accepted: the generic kernel csrc/pgs_big.hpp, tests/test_pgs.py)
  m = mjw.mjcf.from_xml_string('<mujoco><option cone="elliptic" solver="PGS"/><worldbody><body><joint/><geom size=".1"/></body></worldbody></mujoco>')
  assert mjw.put_model(m).opt.solver == int(mjw.SolverType.PGS)
  # structural elements the subset compiler does not expand are never skipped silently (they would change the model)
  for body in ('<composite type="grid"/>', '<body><joint/><geom size=".1"/><flexcomp name="f"/></body>'):
    with pytest.raises(NotImplementedError):
      mjw.mjcf.from_xml_string(f"<mujoco><worldbody>{body}</worldbody></mujoco>")
  with pytest.raises(NotImplementedError):
    mjw.mjcf.from_xml_string('<mujoco><worldbody><body><joint/><geom size=".1"/></body></worldbody><bogus/></mujoco>')
  m = mjw.mjcf.from_xml_string('<mujoco><statistic meaninertia="2.5"/><worldbody><body><joint/><geom size=".1"/></body></worldbody></mujoco>')
  assert m.stat.meaninertia == 2.5
  for opt in ('density="1.2"', 'viscosity="0.1"'):
    m = mjw.mjcf.from_xml_string(f'<mujoco><option {opt}/><worldbody><body><joint/><geom size=".1"/></body></worldbody></mujoco>')
    with pytest.raises(NotImplementedError):
      mjw.put_model(m)
  m = mjw.mjcf.from_xml_string('<mujoco><option><flag override="enable"/></option><worldbody><body><joint/><geom size=".1"/></body></worldbody></mujoco>')
  with pytest.raises(NotImplementedError):
    mjw.put_model(m)
  # (joint actuatorfrcrange / actuatorgravcomp are implemented since round 4: tests/test_aloha_pot.py)
  m = mjw.mjcf.from_xml_string('<mujoco><worldbody><body><joint actuatorfrcrange="-1 1"/><geom size=".1"/></body></worldbody></mujoco>')
  assert int(mjw.put_model(m).jnt_actfrclimited.numpy()[0]) == 1
  m = mjw.mjcf.from_xml_string('<mujoco><option integrator="implicit"/><worldbody><body><joint/><geom size=".1"/></body></worldbody></mujoco>')
  assert int(mjw.put_model(m).opt.integrator) == int(mjw.IntegratorType.IMPLICIT)  # (round 3: built for nv <= 64, tests/test_implicit.py)


# ---------------------------------------------------------------------------------- C ABI
def test_library_exports_every_declared_symbol():
  L = _abi.lib()
  header = open(_abi.HEADER).read()
  declared = re.findall(r"^\s*(?:const char\*|int)\s+(mjh_\w+)\(", header, flags=re.M)
  assert len(declared) >= 12
  for name in declared:
    assert hasattr(L, name), f"libmjhip.so does not export {name}"
  assert L.mjh_abi_version() == _abi.DEFINES["MJH_ABI_VERSION"] >= 12


def test_ws_ccd_size_mirror_matches_the_library():
  """io._ccd_words (what put_data allocates for Data.ws_ccd) against the library's own ccd_layout, through mjh_ws_ccd_floats: a binding
  that allocates Data itself calls the latter; the Python mirror must never be smaller."""
  from mujoco_warp_amd import io as mio
  L = _abi.lib()
  for nworld, it, hf, P, D, npair, nconmax in ((1, 35, 0, 0, 0, 3, 8), (64, 35, 1, 4, 3, 190, 24), (8192, 35, 0, 76, 43, 9154, 24), (2048, 50, 0, 12, 7, 780, 256),
                                                (3, 64, 1, 8, 8, 1, 1), (512, 16, 0, 0, 0, 100000, 64)):
    concap = mio.contact_cap(nconmax)
    ccap = mio._collide_ccap(npair, concap)
    out, hand = ctypes.c_double(), ctypes.c_int()
    assert L.mjh_ws_ccd_floats(nworld, it, hf, P, D, npair, concap, ctypes.byref(out), ctypes.byref(hand)) == 0
    assert hand.value == mio._ccd_handcap(nworld, ccap)
    words = mio._ccd_words(nworld, it, hf, P, D, ccap, npair)
    assert words * 32 * nworld >= out.value > (words - 1) * 32 * nworld, (nworld, it, hf, P, D, npair, words, out.value)


def test_struct_layout_matches_header():
  assert ctypes.sizeof(_abi.CModel) % 8 == 0 and ctypes.sizeof(_abi.CData) % 8 == 0
  names = [n for n, _, _ in _abi.MODEL_FIELDS]
  assert names[:3] == ["nq", "nv", "nu"] and "nxn_geom_pair" in names and "body_dofmask" in names
  dnames = [n for n, _, _ in _abi.DATA_FIELDS]
  assert dnames[0] == "nworld" and "efc_J" in dnames and dnames[-1] == "ws_contact" and "eq_active" in dnames and "ws_rk" in dnames
  # every batched pointer has its _nb companion right after it
  for i, (n, k, p) in enumerate(_abi.MODEL_FIELDS):
    if n.endswith("_nb"):
      assert _abi.MODEL_FIELDS[i - 1][0] == n[:-3] and _abi.MODEL_FIELDS[i - 1][2]


def test_no_cpu_fallback(humanoid):
  import torch

  if torch.cuda.is_available():
    pytest.skip("GPU present")
  m = mjw.put_model(humanoid)
  d = mjw.make_data(humanoid, nworld=2)
  with pytest.raises(RuntimeError, match="no CPU fallback"):
    mjw.step(m, d)


# ---------------------------------------------------------------------------------- IO boundary
def test_put_model_and_make_data_shapes(humanoid):
  m = mjw.put_model(humanoid)
  assert (m.nq, m.nv, m.nC, m.nv_pad, m.npair) == (28, 27, 243, 28, 161)
  assert m.opt.timestep.shape == (1,) and m.opt.timestep.dtype == np.float32
  np.testing.assert_allclose(m.opt.tolerance.numpy(), [1e-6])  # clamped like io.py:398-401
  assert m.body_pos.shape == (1, 17, 3) and m.geom_size.shape == (1, 20, 3) and m.actuator_gear.shape == (1, 21, 6)
  assert m.body_parentid.dtype == np.int32
  d = mjw.make_data(humanoid, nworld=5, nconmax=24, njmax=64)
  assert (d.nworld, d.naconmax, d.njmax, d.njmax_pad, d.nv_pad) == (5, 120, 64, 64, 28)
  assert d.qpos.shape == (5, 28) and d.xmat.shape == (5, 17, 3, 3) and d.cdof.shape == (5, 27, 6)
  assert d.efc.J.shape == (5, 64, 28) and d.efc.D.shape == (5, 64) and d.contact.frame.shape == (120, 3, 3)
  assert d.contact.efc_address.shape == (120, 4) and d.nacon.shape == (1,) and d.M.shape == (5, 243)
  np.testing.assert_allclose(d.qpos.numpy()[3], humanoid.qpos0.astype(np.float32))
  with pytest.raises(ValueError):
    mjw.make_data(humanoid, nworld=0)
  with pytest.raises(ValueError):
    mjw.make_data(humanoid, nworld=1, njmax=-1)


def test_put_data_get_data_roundtrip(humanoid):
  mjd = mjw.MjData(humanoid)
  mjw.mj_resetDataKeyframe(humanoid, mjd, 1)
  mjd.qvel[:] = np.arange(27) * 0.01
  mjd.ctrl[:] = 0.1
  mjd.time = 1.5
  d = mjw.put_data(humanoid, mjd, nworld=3, nconmax=8, njmax=32)
  out = mjw.MjData(humanoid)
  mjw.get_data_into(out, humanoid, d, world_id=2)
  np.testing.assert_allclose(out.qpos, mjd.qpos.astype(np.float32))
  np.testing.assert_allclose(out.qvel, mjd.qvel.astype(np.float32))
  assert out.time == 1.5 and out.ncon == 0 and out.nefc == 0


def test_reset_and_override(humanoid):
  m = mjw.put_model(humanoid)
  d = mjw.make_data(humanoid, nworld=4)
  mjw.reset_data_keyframe(m, d, 0)
  np.testing.assert_allclose(d.qpos.numpy()[1][:7], [0, 0, 0.596, 0.988015, 0, 0.154359, 0], rtol=1e-6)
  d.qvel.fill_(1.0)
  mjw.reset_data(m, d, reset=np.array([True, False, True, False]))
  q = d.qvel.numpy()
  assert (q[0] == 0).all() and (q[1] == 1).all()
  mjw.override_model(m, {"opt.solver": "cg", "opt.iterations": 7})
  assert m.opt.solver == mjw.SolverType.CG and m.opt.iterations == 7
  assert io.c_model(m).solver == 1 and io.c_model(m).iterations == 7
  host = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  mjw.override_model(host, ["opt.solver=newton", "opt.ls_iterations=12"])
  assert host.opt.solver == 2 and host.opt.ls_iterations == 12
  with pytest.raises(ValueError):
    mjw.override_model(m, {"opt.nonexistent": 1})


def test_batched_model_field(humanoid):
  m = mjw.put_model(humanoid, batch_sizes={"gravity": 4, "body_mass": 4})
  assert m.opt.gravity.shape == (4, 3) and m.body_mass.shape == (4, 17)
  assert io.c_model(m).opt_gravity_nb == 4 and io.c_model(m).body_mass_nb == 4 and io.c_model(m).body_pos_nb == 1
  with pytest.raises(ValueError):
    mjw.put_model(humanoid, batch_sizes={"body_parentid": 4})


def test_geom_pair_filter(humanoid):
  pairs = io.geom_pairs(humanoid)
  from oracle.ref import filtered_geom_pairs

  np.testing.assert_array_equal(pairs, filtered_geom_pairs(humanoid))
  # floor collides with every body geom; parent/child and same-body pairs are filtered
  assert (pairs[:, 0] == 0).sum() == 19
  b = humanoid.geom_bodyid
  assert all(b[i] != b[j] for i, j in pairs)


def test_device_array_surface():
  a = mjw.DeviceArray.zeros((3, 4))
  assert a.shape == (3, 4) and a.dtype == np.float32 and a.size == 12 and a.capacity == 48
  a.fill_(2.0)
  assert (a.numpy() == 2).all()
  a.zero_()
  a[1].assign(np.arange(4))
  np.testing.assert_array_equal(a.numpy()[1], [0, 1, 2, 3])
  b = mjw.DeviceArray.from_numpy(np.array([1, 2], dtype=np.uint32))
  assert b.numpy().dtype == np.uint32


def test_load_trajectory_zero_order_hold(humanoid, tmp_path):
  """io.load_trajectory (reference io.py:3067): controls held on the model timestep; both `times` conventions; errors."""
  nu, dt = humanoid.nu, float(humanoid.opt.timestep)  # dt = 0.005
  ctrl = np.arange(3 * nu, dtype=np.float64).reshape(3, nu)
  mjd = mjw.MjData(humanoid)
  p = str(tmp_path / "t.npz")
  # one timestamp per control, 4 model steps per control; the last control is held for the same interval
  np.savez(p, ctrl=ctrl, times=np.array([0.0, 0.02, 0.04]), qpos=np.full((1, humanoid.nq), 0.25))
  out = mjw.load_trajectory(p, humanoid, mjd)
  assert out.shape == (12, nu)
  np.testing.assert_array_equal(out[:, 0], np.repeat(ctrl[:, 0], 4))
  np.testing.assert_array_equal(mjd.qpos, 0.25)
  # interval boundaries (n + 1 timestamps), uneven intervals: 2, 1 and 3 steps
  np.savez(p, ctrl=ctrl, times=np.array([1.0, 1.0 + 2 * dt, 1.0 + 3 * dt, 1.0 + 6 * dt]))
  out = mjw.load_trajectory(p, humanoid, mjd)
  np.testing.assert_array_equal(out[:, 0], np.repeat(ctrl[:, 0], [2, 1, 3]))
  # a single control is held for one model timestep
  np.savez(p, ctrl=ctrl[:1], times=np.array([0.0]))
  assert mjw.load_trajectory(p, humanoid, mjd).shape == (1, nu)
  for bad in (dict(ctrl=ctrl[:, :-1], times=np.arange(3.0)), dict(ctrl=ctrl, times=np.array([0.0, 0.02])),
              dict(ctrl=ctrl, times=np.array([0.0, 0.02, 0.02])), dict(ctrl=ctrl, times=np.array([0.0, np.inf, 1.0])),
              dict(ctrl=np.zeros((0, nu)), times=np.zeros(0))):
    np.savez(p, **bad)
    with pytest.raises(ValueError):
      mjw.load_trajectory(p, humanoid, mjd)



def test_frame_and_replicate_expansion():
  """<frame> re-expresses its children in the parent; <replicate> applies its transform i times and suffixes the names."""
  m = mjw.mjcf.from_xml_string("""
<mujoco><worldbody>
  <frame pos="1 0 0" euler="0 0 90">
    <geom name="g" type="capsule" fromto="0 0 0 1 0 0" size=".05"/>
    <body name="a" pos="1 0 0"><joint name="ja" type="hinge" axis="1 0 0"/><geom size=".1"/></body>
  </frame>
  <replicate count="3" offset="0 0 .5" euler="0 0 90" sep="-">
    <body name="r" pos="1 0 0"><freejoint/><geom size=".1"/></body>
  </replicate>
</worldbody></mujoco>""")
  names = list(m.body_names)
  assert names == ["world", "a", "r-0", "r-1", "r-2"]
  np.testing.assert_allclose(m.body_pos[1], [1, 1, 0], atol=1e-12)  # (1,0,0) rotated 90 deg about z, then shifted
  np.testing.assert_allclose(nm_rot(m.body_quat[1], [1, 0, 0]), [0, 1, 0], atol=1e-12)
  # copy i: transform applied i times -> rotation i * 90 deg about z and height i * 0.5
  np.testing.assert_allclose(m.body_pos[2:5], [[1, 0, 0], [0, 1, 0.5], [-1, 0, 1.0]], atol=1e-12)
  g = list(m.geom_names).index("g")
  np.testing.assert_allclose(m.geom_pos[g], [1, 0.5, 0], atol=1e-12)  # midpoint of the rotated, shifted segment
  assert m.nv == 1 + 3 * 6


def nm_rot(q, v):
  from mujoco_warp_amd import _npmath as nm

  return nm.rot_vec_quat(np.asarray(v, dtype=np.float64), np.asarray(q))


def test_attach_three_humanoids_matches_manual_clone():
  """benchmarks/humanoid/three_humanoids.xml (<asset><model>, <replicate>, <frame>, <attach>) against the same model built by
  cloning the torso subtree by hand (conftest.multi_humanoid_xml): identical trees, masses, defaults and actuators."""
  import os

  a = mjw.mjcf.load_xml(os.path.join(os.path.dirname(conftest.HUMANOID_XML), "three_humanoids.xml"))
  b = mjw.mjcf.from_xml_string(conftest.multi_humanoid_xml(3), assets_dir=os.path.dirname(conftest.HUMANOID_XML))
  assert (a.nv, a.nbody, a.nu, a.ngeom, a.njnt) == (b.nv, b.nbody, b.nu, b.ngeom, b.njnt) == (81, 49, 63, 58, 66)
  torsos = [i for i, n in enumerate(a.body_names) if n.startswith("_torso")]
  assert [a.body_names[i] for i in torsos] == ["_torso-0", "_torso-1", "_torso-2"]
  ang = np.deg2rad(16.36) * np.arange(3)
  np.testing.assert_allclose(a.body_pos[torsos], np.stack([4 * np.sin(ang), -4 * np.cos(ang), np.full(3, 1.282)], axis=1), atol=1e-9)
  for f in ("body_mass", "body_inertia", "dof_armature", "dof_damping", "jnt_range", "jnt_stiffness", "actuator_gear", "actuator_ctrlrange",
            "geom_friction", "geom_condim", "body_parentid", "M_rownnz", "jnt_type", "actuator_trnid"):
    np.testing.assert_allclose(getattr(a, f), getattr(b, f), atol=1e-12, err_msg=f)
  np.testing.assert_allclose(a.geom_size[1:], b.geom_size[1:], atol=1e-12)  # (the floors differ)
  assert a.stat.meaninertia == pytest.approx(b.stat.meaninertia, rel=1e-12)


def test_mocap_model_and_keyframe():
  m = mjw.mjcf.from_xml_string("""
<mujoco><worldbody>
  <body name="hand" mocap="true" pos="0 0 .3"><geom type="box" size=".1 .1 .02"/></body>
  <body name="ball" pos="0 0 .4"><freejoint/><geom type="sphere" size=".05"/></body>
</worldbody>
<keyframe><key name="k" mpos="1 2 3" mquat="0 1 0 0"/></keyframe></mujoco>""")
  assert m.nmocap == 1 and list(m.body_mocapid) == [-1, 0, -1]
  d = mjw.MjData(m)
  np.testing.assert_array_equal(d.mocap_pos, [[0, 0, 0.3]])
  mjw.mj_resetDataKeyframe(m, d, 0)
  np.testing.assert_array_equal(d.mocap_pos, [[1, 2, 3]])
  np.testing.assert_array_equal(d.mocap_quat, [[0, 1, 0, 0]])
  with pytest.raises(ValueError):
    mjw.mjcf.from_xml_string('<mujoco><worldbody><body><joint/><geom size=".1"/><body mocap="true"><geom size=".1"/></body></body></worldbody></mujoco>')


# ---------------------------------------------------------------------------------- declared schema of Model / Data
def _schema_check(obj, env, nworld):
  import dataclasses

  from mujoco_warp_amd.device import DeviceArray

  declared = {f.name: f for f in dataclasses.fields(obj)}
  public = {k for k in vars(obj) if not k.startswith("_")}
  assert public <= set(declared), f"undeclared attributes on {type(obj).__name__}: {sorted(public - set(declared))}"
  for name, f in declared.items():
    v = getattr(obj, name)
    meta = f.metadata
    if "shape" not in meta:
      continue
    assert v is not None, f"{type(obj).__name__}.{name} is declared but was never set"
    assert isinstance(v, np.ndarray if meta["host"] else DeviceArray), name
    assert np.dtype(v.dtype) == np.dtype(meta["dtype"]), (name, v.dtype, meta["dtype"])
    want = []
    for i, s in enumerate(meta["shape"]):
      if s == "*":
        assert v.shape[i] in (1, nworld), name
        want.append(v.shape[i])
      else:
        want.append(int(eval(s, {}, env)) if isinstance(s, str) else s)
    assert tuple(v.shape) == tuple(want), (type(obj).__name__, name, v.shape, want)


@pytest.mark.parametrize("xml", ["humanoid", "panda", "pendula"])
def test_declared_schema_matches_arrays(xml):
  """types.Model / Data / Option / Contact / Constraint are real dataclasses: every attribute put_model / make_data sets is a
  declared field and every declared array has the declared (symbolic) shape and dtype."""
  import dataclasses

  mjm = {"humanoid": lambda: mjw.mjcf.load_xml(conftest.HUMANOID_XML), "panda": lambda: mjw.mjcf.load_xml(conftest.PANDA_XML),
         "pendula": lambda: mjw.mjcf.from_xml_string(conftest.PENDULA_XML)}[xml]()
  m = mjw.put_model(mjm)
  d = mjw.make_data(mjm, nworld=3, nconmax=7, njmax=21)
  assert dataclasses.is_dataclass(mjw.Model) and dataclasses.is_dataclass(mjw.Data)
  env = {k: getattr(m, k) for k in ("nq", "nv", "nu", "na", "nbody", "njnt", "ngeom", "nsite", "nkey", "nmocap", "neq", "nC", "npair", "ncullgeom", "ncullgroup", "ncullpair",
                                    "nexplicit", "nbodylevel", "ndoflevel", "nmaxpyramid", "ntree", "nmesh", "nmeshvert", "nmeshpoly", "nmeshpolyvert", "nmeshpolymap", "nmeshgraph", "nhfield", "nhfielddata", "nsensor", "nsensordata", "nmat")}
  env.update(nworld=d.nworld, njmax=d.njmax, njmax_pad=d.njmax_pad, nv_pad=d.nv_pad, naconmax=d.naconmax, concap=d.concap, nccdworld=d.nccdworld, nccdword=d.nccdword, ntreeadr=(m.ntree + 1) if m.tree_solve else 0, ntreerow=d.njmax if m.tree_solve else 0, ntreedof=m.nv if m.tree_solve else 0, ntreeworld=d.nworld if m.tree_solve else 0, nsleepworld=d.nsleepworld, npgsworld=d.npgsworld, nimpworld=d.nimpworld)
  for obj in (m.opt, m.stat, m, d.contact, d.efc, d):
    _schema_check(obj, env, d.nworld)


# The attributes of mujoco.MjModel / MjOption / MjStatistic that put_model / make_data may read: every name below is used by the
# reference itself as `mjm.<name>` or declared as a Model field (checked against /root/reference/mujoco_warp/_src when this list
# was written), so an object carrying exactly these -- the real mujoco.MjModel on the day the package is importable -- drops in.
_MJMODEL_NAMES = set("""
M_colind M_rowadr M_rownnz actuator_actadr actuator_actearly actuator_actlimited actuator_actnum actuator_actrange actuator_biasprm
actuator_biastype actuator_ctrllimited actuator_ctrlrange actuator_dynprm actuator_dyntype actuator_forcelimited actuator_forcerange
actuator_gainprm actuator_gaintype actuator_gear actuator_trnid actuator_trntype body_dofadr body_dofnum body_gravcomp body_inertia
body_invweight0 body_ipos body_iquat body_jntadr body_jntnum body_mass body_mocapid body_parentid body_pos body_quat body_rootid
body_subtreemass body_weldid dof_armature dof_bodyid dof_damping dof_dampingpoly dof_frictionloss dof_invweight0 dof_jntid dof_parentid
dof_solimp dof_solref eq_active0 eq_data eq_obj1id eq_obj2id eq_solimp eq_solref eq_type exclude_signature geom_aabb geom_bodyid
geom_conaffinity geom_condim geom_contype geom_friction geom_gap geom_margin geom_pos geom_priority geom_quat geom_rbound geom_size
geom_solimp geom_solmix geom_solref geom_type jnt_actfrclimited jnt_actgravcomp jnt_axis jnt_bodyid jnt_dofadr jnt_limited jnt_margin
jnt_pos jnt_qposadr jnt_range jnt_solimp jnt_solref jnt_stiffness jnt_stiffnesspoly jnt_type key_act key_ctrl key_mpos key_mquat key_qpos
key_qvel key_time na nbody ncam neq nflex ngeom nhfield njnt nkey nlight nmesh nmocap npair nplugin nq nsensor nsite ntendon nu nv opt stat
qpos0 qpos_spring site_bodyid site_pos site_quat pair_dim pair_friction pair_gap pair_geom1 pair_geom2 pair_margin pair_solimp pair_solref
pair_solreffriction nexclude
""".split())
_MJOPTION_NAMES = set("ccd_iterations ccd_tolerance cone density disableflags enableflags gravity impratio integrator iterations ls_iterations "
                      "ls_tolerance noslip_iterations solver timestep tolerance viscosity wind magnetic".split())


class _StrictModel:
  """Exposes only genuine mujoco.MjModel attribute names of the wrapped object; anything else raises AttributeError, exactly as a
  real MjModel would."""

  def __init__(self, obj, names, log):
    object.__setattr__(self, "_obj", obj)
    object.__setattr__(self, "_names", names)
    object.__setattr__(self, "_log", log)

  def __getattr__(self, k):
    if k not in self._names:
      raise AttributeError(f"'mujoco.MjModel' object has no attribute '{k}'")
    self._log.add(k)
    v = getattr(self._obj, k)
    if k == "opt":
      return _StrictModel(v, _MJOPTION_NAMES, self._log)
    if k == "stat":
      return _StrictModel(v, {"meaninertia", "meanmass", "meansize", "extent", "center"}, self._log)
    return v


@pytest.mark.parametrize("xml", ["humanoid", "g1", "panda"])
def test_put_model_accepts_a_duck_typed_mjmodel(xml):
  path = {"humanoid": conftest.HUMANOID_XML, "g1": conftest.G1_XML, "panda": conftest.PANDA_XML}[xml]
  mjm = mjw.mjcf.load_xml(path)
  log = set()
  strict = _StrictModel(mjm, _MJMODEL_NAMES, log)
  m = mjw.put_model(strict)
  ref_m = mjw.put_model(mjm)
  assert (m.nq, m.nv, m.nC, m.npair) == (ref_m.nq, ref_m.nv, ref_m.nC, ref_m.npair)
  np.testing.assert_array_equal(m.body_inertia.numpy(), ref_m.body_inertia.numpy())
  np.testing.assert_array_equal(m.geom_size.numpy(), ref_m.geom_size.numpy())
  d = mjw.make_data(strict, nworld=2)
  assert d.qpos.shape == (2, mjm.nq)
  assert len(log) > 100  # the whole model was read through the strict view


def test_builder_object_cache_keys_follow_sources_and_flags(tmp_path, monkeypatch):
  """The builder caches one object per translation unit under the hash of everything the unit is compiled from (its source, the quoted
  headers it includes transitively, the flags): touching a header must change the key of exactly the units that include it."""
  from mujoco_warp_amd import _abi

  keys = {u: _abi._unit_key(u) for u in _abi.UNITS}
  assert len(set(keys.values())) == len(keys)  # distinct units, distinct keys
  assert keys == {u: _abi._unit_key(u) for u in _abi.UNITS}  # deterministic
  # every unit listed exists, every header it reaches is in the declared source list (needs_build looks at that list)
  import re

  csrc = os.path.join(os.path.dirname(_abi.__file__), "csrc")
  declared = set(_abi.UNITS + _abi.HEADERS)
  for u in _abi.UNITS:
    seen, todo = set(), [u]
    while todo:
      f = todo.pop()
      if f in seen:
        continue
      seen.add(f)
      for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(os.path.join(csrc, f)).read(), flags=re.M):
        if not inc.startswith(".."):
          todo.append(inc)
    assert seen <= declared, (u, seen - declared)
  # flags are part of the key
  monkeypatch.setitem(_abi.UNIT_FLAGS, "mjhip.hip", ["-DSOMETHING"])
  assert _abi._unit_key("mjhip.hip") != keys["mjhip.hip"]
  assert _abi._unit_key("pgs_tu.hip") == keys["pgs_tu.hip"]
  # the per-unit flags only name real units
  assert set(_abi.UNIT_FLAGS) <= set(_abi.UNITS)


def test_positive_velocity_feedback_keeps_implicitfast_out_of_the_solver_epilogue():
  """M + h D - h dA/dv stays positive definite only while no actuator feeds velocity back with a positive sign; the fused epilogue factors
  it by Cholesky, so models that can break that (affine gain on velocity, bias velocity coefficient > 0) carry Model.act_velfeedback and
  keep the integrator launch's L'DL solve."""
  base = '<mujoco><option integrator="implicitfast"/><worldbody><body><joint name="j"/><geom size=".1"/></body></worldbody><actuator>{}</actuator></mujoco>'
  assert mjw.put_model(mjw.mjcf.from_xml_string(base.format('<position joint="j" kp="10" kv="1"/>'))).act_velfeedback == 0   # bias velocity term -kv
  assert mjw.put_model(mjw.mjcf.from_xml_string(base.format('<motor joint="j"/>'))).act_velfeedback == 0
  assert mjw.put_model(mjw.mjcf.from_xml_string(base.format('<general joint="j" biastype="affine" biasprm="0 0 0.5"/>'))).act_velfeedback == 1
  assert mjw.put_model(mjw.mjcf.from_xml_string(base.format('<general joint="j" gaintype="affine" gainprm="1 0 -0.2"/>'))).act_velfeedback == 1
  assert mjw.put_model(mjw.mjcf.load_xml(conftest.PANDA_XML)).act_velfeedback == 0  # (the Panda keeps the fused update)


def test_library_carries_the_id_of_the_sources_it_was_built_from():
  """csrc/build_id.hip bakes the hash of csrc/*, the header and the compiler flags into the library; the loader compares it with the sources
  on disk (no mtime logic: a pushed tree with fresh timestamps must load, a tree whose sources changed must not run old kernels)."""
  from mujoco_warp_amd import _abi

  L = _abi.lib()
  assert L.mjh_build_id().decode() == _abi.source_build_id() == _abi.library_build_id()
  assert not _abi.needs_build()
  os.utime(os.path.join(os.path.dirname(_abi.__file__), "csrc", "solver.hpp"))  # a fresh timestamp alone changes nothing
  assert not _abi.needs_build()


def test_stale_library_is_never_loaded_silently(monkeypatch):
  """Sources that no longer match the library and no way to rebuild: lib() raises instead of falling through to the old file."""
  import subprocess

  from mujoco_warp_amd import _abi

  def no_compiler(*a, **k):
    raise subprocess.CalledProcessError(127, ["hipcc"])

  monkeypatch.setattr(_abi, "_lib", None)
  monkeypatch.setattr(_abi, "source_build_id", lambda: "0" * 24)
  monkeypatch.setattr(_abi, "build", no_compiler)
  monkeypatch.delenv("MJH_LIB", raising=False)
  with pytest.raises(RuntimeError, match="does not match the sources"):
    _abi.lib()


@pytest.mark.parametrize("scene", ["humanoid", "aloha_pot"])
def test_cull_tables_regroup_the_pair_list(scene):
  """io.cull_tables (k_broad_mask's group pre-test): every pair of the filtered list exactly once; the entries of a group pair lie in its
  two groups; a group holds the colliding geoms of a run of one moving body's geoms, every static geom is a group of its own."""
  xml = conftest.HUMANOID_XML if scene == "humanoid" else os.path.join(conftest.ROOT, "benchmarks", "aloha_pot", "scene.xml")
  mjm = mjw.mjcf.load_xml(xml)
  pairs, pairid = io.geom_pairs_with_ids(mjm)
  cgeom, group, cpair, clist = io.cull_tables(mjm.geom_bodyid, pairs, pairid, mjm.geom_pos, mjm.geom_rbound)
  geoms, ggroup = cgeom[:, 0] & 0xffff, cgeom[:, 0] >> 16
  assert sorted(clist[:, 0].tolist()) == list(range(len(pairs)))
  assert (clist[:, 1] == pairs[clist[:, 0], 0] + (pairs[clist[:, 0], 1] << 16)).all()
  assert group[:, 1].sum() == len(geoms) and (np.diff(geoms) > 0).all() and (np.diff(ggroup) >= 0).all()
  assert set(geoms.tolist()) == set(pairs[pairid < 0].ravel().tolist())
  body = np.asarray(mjm.geom_bodyid)
  gid = np.full(mjm.ngeom, -1)
  for k, (centre, n) in enumerate(group):
    gs = geoms[ggroup == k]
    gid[gs] = k
    assert len(gs) == n and centre in gs and (cgeom[ggroup == k, 1] == centre).all()
    assert (body[gs] == body[gs[0]]).all() and (n == 1 or body[gs[0]] != 0)
  start, count = cpair[:, 3] & 0xffffff, cpair[:, 3] >> 24
  assert count.sum() == len(pairs) and count.max() <= 16 and (start == np.concatenate([[0], np.cumsum(count)[:-1]])).all()
  for (g1, g2, cc, _), st, n in zip(cpair, start, count):
    p = clist[st: st + n, 0]
    if g1 < 0:
      assert (pairid[p] >= 0).all()
    else:
      assert (gid[pairs[p, 0]] == g1).all() and (gid[pairs[p, 1]] == g2).all() and (pairid[p] < 0).all()
      assert cc == group[g1, 0] + (group[g2, 0] << 16)
  assert len(cpair) < len(pairs) or len(pairs) <= 1


def test_cull_tables_edge_cases():
  """io.cull_tables: no pairs / too many geoms -> empty tables (the kernel then tests nothing because there is nothing to test); explicit
  pairs only -> rows of group -1 and no group spheres; a long group pair is split into rows of at most 16 pairs that still cover it once."""
  e = io.cull_tables(np.array([0, 1, 1]), np.zeros((0, 2), np.int32), np.zeros(0, np.int32))
  assert [len(a) for a in e] == [0, 0, 0, 0]
  pairs = np.array([[0, 1], [0, 2], [1, 2]], np.int32)
  cgeom, group, cpair, clist = io.cull_tables(np.array([0, 1, 2]), pairs, np.array([0, 1, 2], np.int32))
  assert len(cgeom) == 0 and len(group) == 0 and (cpair[:, 0] == -1).all() and (cpair[:, 3] >> 24).sum() == 3 and sorted(clist[:, 0].tolist()) == [0, 1, 2]
  # two bodies of 7 geoms each: 49 pairs between them = rows of 16, 16, 16, 1
  body = np.array([1] * 7 + [2] * 7)
  pairs = np.array([(a, b) for a in range(7) for b in range(7, 14)], np.int32)
  pos = np.random.default_rng(0).normal(size=(14, 3))
  cgeom, group, cpair, clist = io.cull_tables(body, pairs, np.full(len(pairs), -1, np.int32), pos, np.full(14, 0.1))
  assert len(group) == 2 and group[:, 1].tolist() == [7, 7] and (cpair[:, 3] >> 24).tolist() == [16, 16, 16, 1]
  assert (cpair[:, 0] == 0).all() and (cpair[:, 1] == 1).all() and sorted(clist[:, 0].tolist()) == list(range(49))
  # the centre geom: the one with the smallest enclosing radius over its group
  x = pos[:7]
  want = int(np.argmin((np.linalg.norm(x[:, None] - x[None], axis=2) + 0.1).max(axis=1)))
  assert group[0, 0] == want and (cgeom[:7, 1] == want).all()
