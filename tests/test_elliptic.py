"""Elliptic friction cones (reference constraint.py:2698-2704, 4277-4294; solver.py:272-517, 957-1015, 2466-2564).

No reference-held numbers exist for the cone solver, so the float64 oracle is pinned by properties an independent derivation
gives: the converged Newton and CG solutions coincide, they are stationary points of the primal problem, every contact force
lies inside its friction cone (on the boundary when sliding), a sliding box decelerates at exactly mu g, and a condim-1 scene
does not depend on the cone type.  The GPU path is then compared with the oracle.
"""

import numpy as np
import pytest

import mujoco_warp_amd as mjw
from oracle import ref
from tests import conftest
from tests.conftest import relerr

ELLIPTIC_XML = """
<mujoco>
  <option timestep="0.004" cone="elliptic" impratio="10"/>
  <worldbody>
    <geom name="floor" type="plane" size="0 0 .05" friction=".7 .01 .001" condim="{condim}"/>
    <body name="box" pos="0 0 .099"><freejoint/><geom type="box" size=".1 .15 .1" friction=".7 .02 .002" condim="{condim}"/></body>
    <body name="ball" pos=".5 0 .079"><freejoint/><geom type="sphere" size=".08" condim="{condim}"/></body>
    <body name="cap" pos="0 -.6 .049" euler="0 80 20"><freejoint/><geom type="capsule" size=".05 .1" condim="{condim}"/></body>
    <body name="pend" pos="-.6 0 .5">
      <joint type="hinge" axis="0 1 0" range="-30 30" limited="true"/><geom type="capsule" fromto="0 0 0 0 0 -.42" size=".04" condim="{condim}"/>
    </body>
  </worldbody>
  <keyframe>
    <key name="slide" qvel="1.5 0.3 0 0 0 0.5   0.8 0 0 0 4 0   0 0.6 0 2 0 0   1.0"/>
  </keyframe>
</mujoco>
"""


def _model(condim=3, cone="elliptic"):
  xml = ELLIPTIC_XML.format(condim=condim)
  if cone != "elliptic":
    xml = xml.replace('cone="elliptic"', 'cone="pyramidal"')
  return mjw.mjcf.from_xml_string(xml)


def _sim(mjm, solver, **kw):
  s = ref.RefSim(mjm, nconmax=32, njmax=128, solver=solver, tolerance=1e-12, iterations=200, ls_iterations=100, **kw)
  s.reset(key=0)
  return s


@pytest.mark.parametrize("condim", [3, 4, 6])
def test_oracle_newton_cg_stationary_inside_cone(condim):
  mjm = _model(condim)
  n, c = _sim(mjm, 2), _sim(mjm, 1)
  nrow = 0
  for step in range(30):
    for f in ("qpos", "qvel", "qacc_warmstart"):
      getattr(c, f)[:] = getattr(n, f)
    n.forward()
    c.forward()
    assert n.overflow == 0 and n.nefc == c.nefc
    nefc = n.nefc
    nrow += nefc
    assert (n.efc_type[:nefc] == 7).sum() == sum(int(n.con_dim[i]) for i in range(n.ncon) if n.con_efc_address[i][0] >= 0)
    np.testing.assert_allclose(n.qacc, c.qacc, rtol=2e-4, atol=2e-4)
    for s in (n, c):
      J = s.efc_J[:nefc]
      res = s.dense_M() @ s.qacc - s.qfrc_smooth - J.T @ s.efc_force[:nefc]
      assert np.linalg.norm(res) / (mjm.stat.meaninertia * mjm.nv) < (2e-6 if s is n else 1e-4)
    # contact forces inside the cone: f_n >= 0, sum_j (f_j / mu_j)^2 <= f_n^2 (dual cone of the reference's scaled primal cone)
    for con in range(n.ncon):
      r0 = n.con_efc_address[con][0]
      if r0 < 0:
        continue
      dim = int(n.con_dim[con])
      assert dim == condim
      f = n.efc_force[r0 : r0 + dim]
      fri = n.con_friction[con]
      assert f[0] >= -1e-9
      tang = np.sqrt(np.sum((f[1:] / fri[: dim - 1]) ** 2))
      assert tang <= f[0] * (1 + 1e-6) + 1e-9, (con, f, fri)
    n.step()
  assert nrow > 30 * condim


def test_oracle_sliding_box_decelerates_at_mu_g():
  """A box sliding flat on the floor: the tangential contact force saturates at mu * normal force (Coulomb), so the centre of
  mass decelerates at mu g -- elliptic cones are exact here where pyramids are anisotropic."""
  mjm = mjw.mjcf.from_xml_string("""
<mujoco><option timestep="0.002" cone="elliptic" impratio="10"/>
<worldbody><geom name="floor" type="plane" size="0 0 .05" friction=".4"/>
  <body name="box" pos="0 0 .0995"><freejoint/><geom type="box" size=".1 .1 .1" friction=".4"/></body></worldbody>
<keyframe><key qvel="1.2 0.9 0 0 0 0"/></keyframe></mujoco>""")
  s = ref.RefSim(mjm, nconmax=16, njmax=64, solver=2, tolerance=1e-12)
  s.reset(key=0)
  for _ in range(60):  # settle the normal direction
    s.step()
  v0 = s.qvel[:2].copy()
  nstep = 50
  for _ in range(nstep):
    s.step()
  v1 = s.qvel[:2]
  dec = (np.linalg.norm(v0) - np.linalg.norm(v1)) / (nstep * 0.002)
  assert abs(dec - 0.4 * 9.81) < 0.02 * 0.4 * 9.81, dec
  np.testing.assert_allclose(v1 / np.linalg.norm(v1), v0 / np.linalg.norm(v0), atol=1e-2)  # isotropic: direction is kept


def test_oracle_condim1_independent_of_cone():
  a = _sim(_model(1, "elliptic"), 2)
  b = _sim(_model(1, "pyramidal"), 2)
  for _ in range(10):
    a.step()
    b.step()
  np.testing.assert_allclose(a.qpos, b.qpos, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("solver", [mjw.SolverType.NEWTON, mjw.SolverType.CG])
@pytest.mark.parametrize("condim", [3, 4, 6])
def test_gpu_elliptic_matches_oracle(condim, solver):
  mjm = _model(condim)
  mjm.opt.solver = int(solver)
  if solver == mjw.SolverType.CG:
    mjm.opt.iterations = 200  # (CG on these cone scenes takes up to ~90 iterations: keep clear of the cap, whose flag the test asserts)
  s = ref.RefSim(mjm, nconmax=32, njmax=128, tolerance=1e-6)
  s.reset(key=0)
  m = mjw.put_model(mjm)
  d = mjw.put_data(mjm, mjw.MjData(mjm), nworld=2, nconmax=32, njmax=128)
  worst_q = worst_v = 0.0
  for step in range(40):
    for name in ("qpos", "qvel", "qacc_warmstart"):
      getattr(d, name).assign(np.tile(getattr(s, name).astype(np.float32), (2, 1)))
    if step == 20:  # fields of one forward pass: rows and solution
      mjw.forward(m, d)
      s.forward()
      nefc = s.nefc
      assert int(d.nefc.numpy()[1]) == nefc
      assert (d.efc.type.numpy()[1, :nefc] == s.efc_type[:nefc]).all()
      assert relerr(d.efc.J.numpy()[1, :nefc, : mjm.nv], s.efc_J[:nefc]) <= 5e-5
      for f in ("D", "aref", "pos", "margin", "vel"):
        assert relerr(getattr(d.efc, f).numpy()[1, :nefc], getattr(s, "efc_" + f)[:nefc]) <= 5e-4, f
      assert relerr(d.qacc.numpy()[1], s.qacc) <= 5e-3
      assert relerr(d.efc.force.numpy()[1, :nefc], s.efc_force[:nefc]) <= 5e-3
    mjw.step(m, d)
    s.step()
    worst_q = max(worst_q, relerr(d.qpos.numpy()[1], s.qpos))
    worst_v = max(worst_v, relerr(d.qvel.numpy()[1], s.qvel))
  assert worst_q <= 1e-5, worst_q
  assert worst_v <= 5e-3, worst_v
  assert (d.overflow.numpy() == 0).all()
