"""The fully implicit-in-velocity integrator (IntegratorType.IMPLICIT; reference forward.py:578-600 + derivative.py:514 deriv_rne_vel).

The reference compares it with MuJoCo C at run time and holds no numbers.  The oracle's restatement is pinned by what defines it:
d(qfrc_bias)/d(qvel) against central finite differences of the oracle's own RNE, the implicit update against the defining linear system
(M - h dF/dv) qacc' = M qacc with dF/dv from finite differences of qfrc_smooth, agreement with Euler to O(h^2), and its reason to
exist -- a fast-spinning body with velocity-dependent forces stays bounded where the explicit update gains energy.  GPU vs oracle then.
"""

import ctypes

import numpy as np
import pytest

import mujoco_warp_amd as mjw
from oracle import ref
from tests import conftest
from tests.conftest import relerr

IMPLICIT = int(mjw.IntegratorType.IMPLICIT)

SPIN_XML = """
<mujoco>
  <option timestep="0.004" gravity="0 0 -9.81" integrator="implicit"/>
  <worldbody>
    <body name="top" pos="0 0 1">
      <joint type="ball" damping="0.002"/>
      <geom type="box" size=".15 .05 .02" pos=".1 .02 0" density="800"/>
      <body name="arm" pos=".25 0 0">
        <joint type="hinge" axis="0 1 0" damping="0.01" armature="0.001"/>
        <geom type="capsule" fromto="0 0 0 .2 0 .1" size=".02" density="600"/>
        <body name="tip" pos=".2 0 .1">
          <joint type="slide" axis="1 0 0" damping="0.5" stiffness="30"/>
          <geom type="sphere" size=".04" density="900"/>
        </body>
      </body>
    </body>
    <body name="free" pos="1 0 1"><freejoint/><geom type="box" size=".2 .05 .1" pos=".05 0 .02"/></body>
  </worldbody>
  <actuator><position joint="{hinge}" kp="5" kv="0.2"/></actuator>
  <keyframe><key qvel="3 -2 8   4   0.5   0.3 0.2 0.1 6 -5 9"/></keyframe>
</mujoco>
""".replace('joint type="hinge" axis="0 1 0"', 'joint name="h" type="hinge" axis="0 1 0"').replace("{hinge}", "h")


def _deriv_rne(s):
  nv = s.mjm.nv
  D = np.zeros(nv * nv)
  s.lib.ref_deriv_rne_vel(ctypes.byref(s.cm), ctypes.byref(s.cd), D.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
  return D.reshape(nv, nv)


@pytest.mark.parametrize("model", ["spin", "pendula", "humanoid"])
def test_oracle_rne_velocity_derivative_matches_finite_differences(model):
  mjm = {"spin": lambda: mjw.mjcf.from_xml_string(SPIN_XML), "pendula": lambda: mjw.mjcf.from_xml_string(conftest.PENDULA_XML),
         "humanoid": lambda: mjw.mjcf.load_xml(conftest.HUMANOID_XML)}[model]()
  s = ref.RefSim(mjm, nconmax=32, njmax=128)
  s.reset(key=0)
  s.qvel[:] = np.random.default_rng(3).normal(size=mjm.nv) * 2.0
  s.forward()
  D = _deriv_rne(s)
  v0, eps, nv = s.qvel.copy(), 1e-6, mjm.nv
  fd = np.zeros((nv, nv))
  for k in range(nv):
    for sg in (1, -1):
      s.qvel[:] = v0
      s.qvel[k] += sg * eps
      s.stage("com_vel")
      s.stage("rne")
      fd[:, k] += sg * s.qfrc_bias / (2 * eps)
  assert np.abs(fd).max() > 0.1 and np.abs(D - D.T).max() > 1e-3  # a real, non-symmetric matrix
  np.testing.assert_allclose(D, fd, atol=2e-7 * max(1.0, np.abs(fd).max()))


def test_oracle_implicit_step_solves_its_defining_system():
  """(M - h dF/dv) qacc' = M qacc with dF/dv by finite differences of qfrc_smooth (passive - bias + actuation)."""
  mjm = mjw.mjcf.from_xml_string(SPIN_XML)
  s = ref.RefSim(mjm, nconmax=8, njmax=16, integrator=IMPLICIT)
  s.reset(key=0)
  s.ctrl[:] = 0.3
  s.forward()
  nv, h = mjm.nv, float(mjm.opt.timestep)
  M, qacc, v0, q0 = s.dense_M(), s.qacc.copy(), s.qvel.copy(), s.qpos.copy()
  dF = np.zeros((nv, nv))
  t = ref.RefSim(mjm, nconmax=8, njmax=16, integrator=IMPLICIT)
  for k in range(nv):
    for sg in (1, -1):
      t.reset(key=0)
      t.ctrl[:] = 0.3
      t.qvel[k] += sg * 1e-6
      t.forward()
      dF[:, k] += sg * t.qfrc_smooth / 2e-6
  want = np.linalg.solve(M - h * dF, M @ qacc)  # (no constraints in this scene: efc.Ma = M qacc)
  s.stage("implicit")
  got = (s.qvel - v0) / h
  np.testing.assert_allclose(got, want, atol=2e-6 * np.abs(want).max())


def test_oracle_implicit_agrees_with_euler_to_second_order_and_is_more_stable():
  def run(integrator, h, n):
    mjm = mjw.mjcf.from_xml_string(SPIN_XML)
    mjm.opt.timestep = h
    mjm.opt.disableflags = int(mjm.opt.disableflags) | int(mjw.DisableBit.EULERDAMP)
    s = ref.RefSim(mjm, nconmax=8, njmax=16, integrator=integrator)
    s.reset(key=0)
    for _ in range(n):
      s.step()
    return s.qvel.copy()

  e1 = np.abs(run(IMPLICIT, 1e-3, 1) - run(0, 1e-3, 1)).max()
  e2 = np.abs(run(IMPLICIT, 5e-4, 1) - run(0, 5e-4, 1)).max()
  assert 3.0 < e1 / e2 < 5.0, (e1, e2)  # one-step difference ~ h^2
  # the spinning free box (w = 6, -5, 9 rad/s) and the ball joint: explicit Euler pumps energy through the gyroscopic term at h = 0.02
  ve, vi = run(0, 0.02, 400), run(IMPLICIT, 0.02, 400)
  w0, we, wi = np.linalg.norm([6, -5, 9]), np.linalg.norm(ve[-3:]), np.linalg.norm(vi[-3:])
  assert wi < 1.05 * w0 and not we < 1.05 * w0, (wi, we)  # (Euler: grown or already nan)


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["spin", "pendula", "humanoid", "g1"])
def test_gpu_implicit_matches_oracle(model):
  mjm = {"spin": lambda: mjw.mjcf.from_xml_string(SPIN_XML), "pendula": lambda: mjw.mjcf.from_xml_string(conftest.PENDULA_XML),
         "humanoid": lambda: mjw.mjcf.load_xml(conftest.HUMANOID_XML), "g1": lambda: mjw.mjcf.load_xml(conftest.G1_XML)}[model]()
  mjm.opt.integrator = IMPLICIT
  nconmax, njmax = (48, 192) if model == "g1" else (32, 96)
  s = ref.RefSim(mjm, nconmax=nconmax, njmax=njmax, tolerance=1e-6)
  s.reset(key=0)
  m = mjw.put_model(mjm)
  d = mjw.make_data(mjm, nworld=3, nconmax=nconmax, njmax=njmax)
  assert d.ws_iacc.shape == (3, mjm.nv)
  worst_q = worst_v = 0.0
  for i in range(60):
    if mjm.nu:
      s.ctrl_noise(i, 0)
    for name in ("qpos", "qvel", "qacc_warmstart", "ctrl", "act"):
      dst = getattr(d, name)
      if dst.size:
        dst.assign(np.tile(getattr(s, name).astype(np.float32), (3, 1)))
    mjw.step(m, d)
    s.step()
    if int(d.nefc.numpy()[1]) != s.nefc:
      continue
    worst_q = max(worst_q, relerr(d.qpos.numpy()[1], s.qpos))
    worst_v = max(worst_v, relerr(d.qvel.numpy()[1], s.qvel))
  print(f"implicit {model}: qpos {worst_q:.3g} qvel {worst_v:.3g}")
  assert worst_q <= 2e-6 and worst_v <= 3e-4, (worst_q, worst_v)  # (measured 1.7e-7 / 2.9e-5)
  assert (d.qpos.numpy()[0] == d.qpos.numpy()[2]).all()
  # the stage API: implicit(m, d) after forward == step
  d2 = mjw.make_data(mjm, nworld=2, nconmax=nconmax, njmax=njmax)
  for name in ("qpos", "qvel", "qacc_warmstart", "ctrl", "act"):
    if getattr(d2, name).size:
      getattr(d2, name).assign(getattr(d, name).numpy()[:2])
  d3 = mjw.make_data(mjm, nworld=2, nconmax=nconmax, njmax=njmax)
  for name in ("qpos", "qvel", "qacc_warmstart", "ctrl", "act"):
    if getattr(d3, name).size:
      getattr(d3, name).assign(getattr(d, name).numpy()[:2])
  mjw.step(m, d2)
  mjw.forward(m, d3)
  mjw.implicit(m, d3)
  np.testing.assert_allclose(d3.qvel.numpy(), d2.qvel.numpy(), rtol=0, atol=1e-6)


def test_put_model_accepts_implicit_up_to_64_dofs():
  mjm = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  mjm.opt.integrator = IMPLICIT
  assert int(mjw.put_model(mjm).opt.integrator) == IMPLICIT
  big = mjw.mjcf.load_xml(conftest.ROOT + "/tests/models/clutter_synth.xml")
  big.opt.integrator = IMPLICIT
  with pytest.raises(NotImplementedError):
    mjw.put_model(big)
