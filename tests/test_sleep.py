"""Sleeping and waking of kinematic trees, tree-level constraint islands (reference sleep.py, island.py:28-310, forward.py:345-349,
652-675, 1273-1278, 1344-1347).

What is held by the reference: the BEHAVIOUR its own sleep_test.py asserts (a resting box sleeps within 15 steps and lands in a
self-cycle with exactly zero velocity; a bullet wakes a sleeping target; an awake tree's trajectory does not depend on a sleeping
neighbour; a damped slider settles to exactly zero; a velocity written by the user wakes the tree; a body woken by a contact sees
all its contacts in the same step -- 5 in that scene; an actuated tree never sleeps; applied forces wake a tree per world; a contact
wakes the sleeping side only, never across two sleeping trees, never through a static geom).  The float64 oracle is pinned by those
scenarios (same XML, same step counts, same expectations); the GPU path is then compared with the oracle table by table.
"""

import numpy as np
import pytest

import mujoco_warp_amd as mjw
from oracle import ref
from tests.conftest import relerr

K_AWAKE = -(1 + mjw.types.MJ_MINAWAKE)

BOX_XML = """
<mujoco>
  <option sleep_tolerance="0.01"><flag sleep="enable" island="enable"/></option>
  <worldbody>
    <geom type="plane" size="10 10 .1"/>
    <body name="box" pos="0 0 0.1"><joint type="free"/><geom type="box" size=".1 .1 .1" mass="1.0"/></body>
  </worldbody>
</mujoco>
"""
BULLET_XML = """
<mujoco>
  <option><flag sleep="enable" island="enable"/></option>
  <worldbody>
    <geom type="plane" size="10 10 .1"/>
    <body name="target" pos="0 0 0.1"><joint type="free"/><geom type="sphere" size=".1" mass="1.0"/></body>
    <body name="bullet" pos="-0.3 0 0.1"><joint type="free"/><geom type="sphere" size=".1" mass="1.0"/></body>
  </worldbody>
</mujoco>
"""
TWO_BOX_XML = """
<mujoco>
  <option sleep_tolerance="0.01"><flag sleep="{sleep}" island="enable"/></option>
  <worldbody>
    <geom type="plane" size="10 10 .1"/>
    <body name="b1" pos="0 0 0.1"><joint type="free"/><geom type="box" size=".1 .1 .1" mass="1.0"/></body>
    <body name="b2" pos="1 0 0.5"><joint type="free"/><geom type="box" size=".1 .1 .1" mass="1.0"/></body>
  </worldbody>
</mujoco>
"""
SLIDER_XML = """
<mujoco>
  <option gravity="0 0 0" sleep_tolerance="0.01"><flag sleep="enable" island="enable"/></option>
  <worldbody>
    <body name="box" pos="0 0 0"><joint type="slide" axis="1 0 0" damping="400.0"/><geom type="box" size=".1 .1 .1" mass="1.0"/></body>
  </worldbody>
</mujoco>
"""
RERUN_XML = """
<mujoco>
  <option><flag sleep="enable" island="enable"/></option>
  <worldbody>
    <geom name="floor" type="plane" size="10 10 .1"/>
    <body name="box" pos="0 0 0.09"><joint type="free"/><geom name="box_geom" type="box" size=".1 .1 .1" mass="1.0"/></body>
    <body name="sphere" pos="0 0 0.23"><joint type="free"/><geom name="sphere_geom" type="sphere" size=".05" mass="1.0"/></body>
  </worldbody>
</mujoco>
"""
NEVER_XML = """
<mujoco>
  <option sleep_tolerance="0.01"><flag sleep="enable"/></option>
  <worldbody>
    <geom type="plane" size="10 10 .1"/>
    <body pos="0 0 0.1"><joint name="hinge" type="hinge" axis="0 0 1"/><geom type="box" size=".1 .1 .1" mass="1.0"/></body>
  </worldbody>
  <actuator><motor joint="hinge"/></actuator>
</mujoco>
"""
ARM_XML = """
<mujoco>
  <worldbody>
    <geom name="floor" type="plane" size="5 5 .1"/>
    <body>
      <joint name="j0" type="hinge"/>
      <geom type="capsule" fromto="0 0 0 0 0 .3" size=".05"/>
      <body pos="0 0 .3"><joint name="j1" type="hinge"/><geom type="capsule" fromto="0 0 0 0 0 .3" size=".05"/></body>
    </body>
    <body pos="1 0 1"><freejoint/><geom name="ball0" type="sphere" size=".1"/></body>
    <body pos="2 0 1"><freejoint/><geom name="ball1" type="sphere" size=".1"/></body>
  </worldbody>
  <actuator><motor joint="j0"/><motor joint="j1"/></actuator>
</mujoco>
"""
# a pile scene for the free-running comparison: boxes and a ball dropped from small heights (they settle, sleep, and a late projectile
# wakes a stack), a limited hinge with friction loss (self-edge island), a joint equality tying two pendula (cross-tree equality)
PILE_XML = """
<mujoco>
  <option timestep="0.004" sleep_tolerance="0.02"><flag sleep="enable" island="enable" nativeccd="disable"/></option>
  <worldbody>
    <geom type="plane" size="10 10 .1"/>
    <body pos="0 0 0.101"><freejoint/><geom type="box" size=".1 .1 .1"/></body>
    <body pos="0 0 0.305"><freejoint/><geom type="box" size=".08 .08 .1"/></body>
    <body pos=".6 0 0.12"><freejoint/><geom type="sphere" size=".1"/></body>
    <body pos="-1.2 0 0.1"><freejoint/><geom type="sphere" size=".1" condim="6" friction="1 .005 .01"/></body>
    <body pos="1.5 0 .6"><joint name="p1" type="hinge" axis="0 1 0" damping=".5" frictionloss=".05" range="-60 60" limited="true"/>
      <geom type="capsule" fromto="0 0 0 0 0 -.3" size=".03"/></body>
    <body pos="2.0 0 .6"><joint name="p2" type="hinge" axis="0 1 0" damping=".5"/><geom type="capsule" fromto="0 0 0 0 0 -.3" size=".03"/></body>
  </worldbody>
  <equality><joint joint1="p1" joint2="p2"/></equality>
</mujoco>
"""


def _sim(xml, **kw):
  return ref.RefSim(mjw.mjcf.from_xml_string(xml), nconmax=kw.pop("nconmax", 32), njmax=kw.pop("njmax", 128), **kw)


# ------------------------------------------------------------------ oracle vs the behaviour the reference's sleep_test.py holds
def test_oracle_sleep_initiation():  # sleep_test.py:37-86
  s = _sim(BOX_XML)
  s.qpos[2] = 0.1
  for _ in range(15):
    s.step()
  assert s.tree_asleep[0] == 0 and s.tree_awake[0] == 0 and s.body_awake[1] == mjw.SleepState.ASLEEP and s.body_awake[0] == mjw.SleepState.STATIC
  assert (s.qvel == 0.0).all() and (s.qacc == 0.0).all()
  assert s.ntree_awake == 0 and s.nv_awake == 0 and s.nbody_awake == 1  # (the world body counts as not asleep)


def test_oracle_collision_waking():  # sleep_test.py:88-137
  s = _sim(BULLET_XML)
  s.qvel[6] = 20.0
  s.tree_asleep[0] = 0
  s.stage("update_sleep")
  assert s.tree_awake[0] == 0
  for _ in range(5):
    s.step()
  assert s.tree_awake[0] == 1 and s.tree_asleep[0] < 0


def test_oracle_waking_unaffected_by_sleeping():  # sleep_test.py:237-299
  a, b = _sim(TWO_BOX_XML.format(sleep="enable")), _sim(TWO_BOX_XML.format(sleep="disable"))
  a.qpos[2] = b.qpos[2] = 0.1
  for _ in range(25):
    a.step()
    b.step()
    np.testing.assert_allclose(a.qpos[7:14], b.qpos[7:14], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(a.qvel[6:12], b.qvel[6:12], rtol=1e-6, atol=1e-6)
  assert a.tree_awake[0] == 0 and a.tree_awake[1] == 1


def test_oracle_settle_zero_velocity():  # sleep_test.py:301-343
  s = _sim(SLIDER_XML)
  s.qvel[0] = 5.0
  for _ in range(25):
    s.step()
  assert s.tree_awake[0] == 0 and s.body_awake[1] == mjw.SleepState.ASLEEP and s.qvel[0] == 0.0 and s.qacc[0] == 0.0


def test_oracle_manual_velocity_wake():  # sleep_test.py:663-705
  s = _sim(BOX_XML.replace(' island="enable"', ""))
  s.tree_asleep[0] = 0
  s.stage("update_sleep")
  assert s.tree_awake[0] == 0
  s.qvel[0] = 0.5
  s.step()
  assert s.tree_awake[0] == 1


def test_oracle_sleep_rerun_collision():  # sleep_test.py:707-759: a body woken by a contact sees all its contacts in the same step
  s = _sim(RERUN_XML)
  s.tree_asleep[0] = 0
  s.stage("update_sleep")
  assert s.tree_awake[0] == 0 and s.tree_awake[1] == 1
  s.qvel[8] = -5.0
  s.step()
  assert s.tree_awake[0] == 1 and s.ncon == 5


def test_oracle_policy_auto_never():  # sleep_test.py:761-792
  mjm = mjw.mjcf.from_xml_string(NEVER_XML)
  assert mjm.tree_sleep_policy[0] == mjw.SleepPolicy.AUTO_NEVER
  s = ref.RefSim(mjm)
  for _ in range(15):
    s.step()
  assert s.tree_awake[0] == 1


def test_oracle_wake_kernels():  # sleep_test.py:948-1046 (ActiveDofTest; nv_awake plays the role of the reference's ncdof)
  mjm = mjw.mjcf.from_xml_string(ARM_XML)
  floor, ball0, ball1 = 0, 3, 4
  assert list(mjm.tree_sleep_policy) == [mjw.SleepPolicy.AUTO_NEVER, mjw.SleepPolicy.AUTO_ALLOWED, mjw.SleepPolicy.AUTO_ALLOWED]
  s = ref.RefSim(mjm)
  # applied force wakes its tree; the actuated arm (AUTO_NEVER) cannot stay asleep either
  s.tree_asleep[:] = [0, 1, 2]
  s.tree_awake[:] = 0
  s.qfrc_applied[2] = 1.0
  s.stage("wake")
  s.stage("update_sleep")
  assert s.nv_awake == 8 and list(s.dof_awake_ind[:8]) == list(range(8))
  s.qfrc_applied[:] = 0
  s.tree_asleep[:] = [0, 1, 2]
  s.tree_awake[:] = 0
  s.stage("wake")
  s.stage("update_sleep")
  assert s.nv_awake == 2

  def contact(g0, g1):
    s.cd.ncon = 1
    s.con_geom[0] = (g0, g1)

  s.tree_asleep[:] = [0, K_AWAKE, 2]  # a contact with an awake body wakes the sleeping side
  s.stage("update_sleep")
  contact(ball0, ball1)
  s.stage("wake_collision")
  s.stage("update_sleep")
  assert s.nv_awake == 12
  s.tree_asleep[:] = [0, 1, 2]  # two sleeping trees in contact stay asleep
  s.stage("update_sleep")
  contact(ball0, ball1)
  s.stage("wake_collision")
  s.stage("update_sleep")
  assert s.nv_awake == 0
  s.tree_asleep[:] = [0, K_AWAKE, 2]  # a static geom wakes nothing
  s.stage("update_sleep")
  contact(floor, ball0)
  s.stage("wake_collision")
  s.stage("update_sleep")
  assert s.nv_awake == 6


def test_oracle_islands_and_cycles():
  """Island labels follow the smallest tree; a sleeping island is one cycle; equality-tied pendula sleep together."""
  s = _sim(PILE_XML)
  s.qvel[18] = 3.0  # the lone ball rolls towards the stack (free body 3: dofs 18..23)
  woke, slept_together = False, False
  prev_awake = None
  for step in range(500):
    s.step()
    isl = s.tree_island.copy()
    lab = [i for i in isl if i >= 0]
    assert sorted(set(lab)) == list(range(s.nisland))  # dense numbering
    first = {}
    for t, i in enumerate(isl):
      if i >= 0:
        first.setdefault(i, t)
    assert list(first.values()) == sorted(first.values())  # islands ordered by their smallest tree
    asleep = s.tree_asleep
    for t in range(len(asleep)):
      if asleep[t] >= 0:  # every sleep cycle closes
        cur, n = int(asleep[t]), 0
        while cur != t:
          cur = int(asleep[cur])
          n += 1
          assert n <= len(asleep)
        assert (s.qvel[s.mjm.tree_dofadr[t] : s.mjm.tree_dofadr[t] + s.mjm.tree_dofnum[t]] == 0).all()
    if asleep[4] >= 0 or asleep[5] >= 0:
      assert asleep[4] >= 0 and asleep[5] >= 0 and {int(asleep[4]), int(asleep[5])} == {4, 5}  # the equality ties their sleep
      slept_together = True
    if asleep[0] >= 0 and asleep[1] >= 0:
      assert int(asleep[0]) == 1 and int(asleep[1]) == 0  # the stack sleeps as one cycle
    if prev_awake is not None and prev_awake[0] == 0 and s.tree_awake[0] == 1:
      woke = True
    prev_awake = s.tree_awake.copy()
  assert slept_together and woke and s.ntree_awake == 0


def test_loader_sleep_tables_and_rejections():
  mjm = mjw.mjcf.from_xml_string(BOX_XML)
  assert mjm.opt.enableflags & int(mjw.EnableBit.SLEEP) and mjm.opt.sleep_tolerance == 0.01
  assert mjm.dof_length[:3].tolist() == [1.0, 1.0, 1.0] and (mjm.dof_length[3:] > 0).all()
  with pytest.raises(NotImplementedError):
    mjw.mjcf.from_xml_string(BOX_XML.replace('<body name="box"', '<body name="box" sleep="never"'))
  cg = mjw.mjcf.from_xml_string(BOX_XML.replace("<option ", '<option solver="CG" '))
  with pytest.raises(ValueError, match="Newton"):  # reference io.py:359 (raised before anything touches the device)
    mjw.put_model(cg)


# ------------------------------------------------------------------ GPU vs oracle
def _gpu(xml, nworld, **kw):
  mjm = mjw.mjcf.from_xml_string(xml)
  m = mjw.put_model(mjm)
  d = mjw.make_data(mjm, nworld=nworld, nconmax=kw.pop("nconmax", 32), njmax=kw.pop("njmax", 128))
  return mjm, m, d


def _tables(d, w):
  return {k: getattr(d, k).numpy()[w] for k in ("tree_asleep", "tree_awake", "body_awake", "tree_island", "nisland", "ntree_awake", "nbody_awake", "nv_awake")}


def _check_tables(d, w, s, what):
  t = _tables(d, w)
  for k in ("tree_asleep", "tree_awake", "body_awake", "tree_island"):
    assert (t[k] == getattr(s, k)).all(), f"{what}: {k} gpu {t[k]} oracle {getattr(s, k)}"
  for k in ("nisland", "ntree_awake", "nbody_awake", "nv_awake"):
    assert int(t[k]) == int(getattr(s, k)), f"{what}: {k}"
  assert sorted(d.body_awake_ind.numpy()[w][: s.nbody_awake]) == sorted(s.body_awake_ind[: s.nbody_awake])
  assert sorted(d.dof_awake_ind.numpy()[w][: s.nv_awake]) == sorted(s.dof_awake_ind[: s.nv_awake])


@pytest.mark.gpu
def test_gpu_sleep_initiation_and_zero_state():
  mjm, m, d = _gpu(BOX_XML, 2)
  q = d.qpos.numpy()
  q[:, 2] = 0.1
  d.qpos.assign(q)
  s = ref.RefSim(mjm, nconmax=32, njmax=128)
  s.qpos[2] = 0.1
  for step in range(15):
    mjw.step(m, d)
    s.step()
    for w in range(2):
      _check_tables(d, w, s, f"step {step} world {w}")
    assert relerr(d.qpos.numpy()[0], s.qpos) < 2e-5
  assert (d.tree_asleep.numpy() == 0).all() and (d.qvel.numpy() == 0.0).all() and (d.qacc.numpy() == 0.0).all()
  # asleep: the pair box-plane is filtered, the tree has no rows, nothing moves
  mjw.step(m, d)
  assert (d.nefc.numpy() == 0).all() and int(d.nacon.numpy()[0]) == 0


@pytest.mark.gpu
def test_gpu_collision_waking_and_rerun():
  mjm, m, d = _gpu(BULLET_XML, 2)
  v = d.qvel.numpy()
  v[1, 6] = 20.0  # world 0: the bullet rests, the target stays asleep; world 1: the bullet wakes it
  d.qvel.assign(v)
  a = d.tree_asleep.numpy()
  a[:, 0] = 0
  d.tree_asleep.assign(a)
  mjw.update_sleep(m, d)
  assert (d.tree_awake.numpy()[:, 0] == 0).all()
  sims = [ref.RefSim(mjm, nconmax=32, njmax=128) for _ in range(2)]
  sims[1].qvel[6] = 20.0
  for s in sims:
    s.tree_asleep[0] = 0
    s.stage("update_sleep")
  for step in range(8):
    mjw.step(m, d)
    for w, s in enumerate(sims):
      s.step()
      _check_tables(d, w, s, f"step {step} world {w}")
      assert relerr(d.qpos.numpy()[w], s.qpos) < 5e-5
  assert d.tree_awake.numpy()[1, 0] == 1 and d.tree_awake.numpy()[0, 0] == 0
  # a body woken by a contact sees all of its contacts in the same step (sleep_test.py:707-759: 5 contacts)
  mjm, m, d = _gpu(RERUN_XML, 2)
  a = d.tree_asleep.numpy()
  a[:, 0] = 0
  d.tree_asleep.assign(a)
  mjw.update_sleep(m, d)
  v = d.qvel.numpy()
  v[:, 8] = -5.0
  d.qvel.assign(v)
  mjw.step(m, d)
  assert (d.tree_awake.numpy()[:, 0] == 1).all() and int(d.nacon.numpy()[0]) == 10


@pytest.mark.gpu
def test_gpu_unaffected_settle_manual_never():
  # an awake tree does not feel its sleeping neighbour (sleep_test.py:237-299, at the reference's own 1e-6)
  _, m1, d1 = _gpu(TWO_BOX_XML.format(sleep="enable"), 2)
  _, m2, d2 = _gpu(TWO_BOX_XML.format(sleep="disable"), 2)
  for d in (d1, d2):
    q = d.qpos.numpy()
    q[:, 2] = 0.1
    d.qpos.assign(q)
  for _ in range(25):
    mjw.step(m1, d1)
    mjw.step(m2, d2)
    np.testing.assert_allclose(d1.qpos.numpy()[:, 7:14], d2.qpos.numpy()[:, 7:14], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(d1.qvel.numpy()[:, 6:12], d2.qvel.numpy()[:, 6:12], rtol=1e-6, atol=1e-6)
  assert (d1.tree_awake.numpy()[:, 0] == 0).all() and (d1.tree_awake.numpy()[:, 1] == 1).all()
  # damped slider settles to exactly zero (sleep_test.py:301-343)
  _, m, d = _gpu(SLIDER_XML, 2)
  v = d.qvel.numpy()
  v[:, 0] = 5.0
  d.qvel.assign(v)
  for _ in range(25):
    mjw.step(m, d)
  assert (d.tree_awake.numpy() == 0).all() and (d.qvel.numpy() == 0.0).all() and (d.qacc.numpy() == 0.0).all()
  # a velocity written by the user wakes the tree, per world (sleep_test.py:663-705)
  _, m, d = _gpu(BOX_XML.replace(' island="enable"', ""), 2)
  a = d.tree_asleep.numpy()
  a[:, 0] = 0
  d.tree_asleep.assign(a)
  mjw.update_sleep(m, d)
  v = d.qvel.numpy()
  v[1, 0] = 0.5
  d.qvel.assign(v)
  mjw.step(m, d)
  assert d.tree_awake.numpy()[:, 0].tolist() == [0, 1]
  # an actuated tree never sleeps (sleep_test.py:761-792)
  _, m, d = _gpu(NEVER_XML, 2)
  assert m.tree_sleep_policy.numpy()[0] == mjw.SleepPolicy.AUTO_NEVER
  for _ in range(15):
    mjw.step(m, d)
  assert (d.tree_awake.numpy() == 1).all()


@pytest.mark.gpu
def test_gpu_wake_stages():  # sleep_test.py:948-1046 through the stage API, two worlds
  mjm, m, d = _gpu(ARM_XML, 2)
  d.tree_asleep.assign(np.array([[0, 1, 2], [0, 1, 2]], np.int32))
  d.tree_awake.assign(np.zeros((2, 3), np.int32))
  f = d.qfrc_applied.numpy()
  f[0, 2] = 1.0
  d.qfrc_applied.assign(f)
  mjw.wake(m, d)
  assert d.nv_awake.numpy().tolist() == [8, 2]
  assert d.dof_awake_ind.numpy()[0][:8].tolist() == list(range(8))


@pytest.mark.gpu
def test_gpu_pile_free_running_tables():
  """Free-running pile scene, 4 worlds with different projectile speeds: the sleep tables are compared with four float64 oracle worlds
  every step while the states agree, and the scenario's milestones (stack asleep as one cycle, projectile wakes it, everything asleep
  at the end, equality-tied pendula sleep together) must happen in every GPU world."""
  mjm, m, d = _gpu(PILE_XML, 4, nconmax=48, njmax=160)
  speeds = [3.0, 2.0, 0.0, 4.0]
  v = d.qvel.numpy()
  v[:, 18] = speeds
  d.qvel.assign(v)
  sims = [ref.RefSim(mjm, nconmax=48, njmax=160) for _ in speeds]
  for s, sp in zip(sims, speeds):
    s.qvel[18] = sp
  nt = mjm.ntree
  slept = np.zeros(4, bool)
  woke = np.zeros(4, bool)
  compared = 0
  tracking = [True] * 4
  prev = d.tree_awake.numpy().copy()
  for step in range(500):
    mjw.step(m, d)
    ta, tas = d.tree_awake.numpy(), d.tree_asleep.numpy()
    qvel = d.qvel.numpy()
    for w, s in enumerate(sims):
      s.step()
      if tracking[w]:
        # (chaotic contact scene in float32 vs float64: table-by-table comparison while the trajectories coincide)
        if relerr(d.qpos.numpy()[w], s.qpos) < 1e-4 and (tas[w] == s.tree_asleep).all():
          _check_tables(d, w, s, f"step {step} world {w}")
          compared += 1
        else:
          tracking[w] = False
      for t in range(nt):
        if tas[w, t] >= 0:
          adr, num = mjm.tree_dofadr[t], mjm.tree_dofnum[t]
          assert (qvel[w, adr : adr + num] == 0).all()
      if tas[w, 0] >= 0 and tas[w, 1] >= 0:
        slept[w] = True
      if prev[w, 0] == 0 and ta[w, 0] == 1:
        woke[w] = True
      if tas[w, 4] >= 0 or tas[w, 5] >= 0:
        assert {int(tas[w, 4]), int(tas[w, 5])} == {4, 5}
    prev = ta.copy()
  assert compared > 200, compared
  assert slept.all(), slept
  assert woke[[0, 1, 3]].all() and not woke[2], woke  # the projectile at rest (world 2) wakes nobody
  assert (d.ntree_awake.numpy() == 0).all() and int(d.nacon.numpy()[0]) == 0 and (d.overflow.numpy() == 0).all()


def _many_boxes_xml(n=12):
  bodies = []
  for i in range(n):
    x, y = 0.5 * (i % 4), 0.5 * (i // 4)
    bodies.append(f'<body pos="{x} {y} 0.101"><freejoint/><geom type="box" size=".1 .1 .1"/></body>')
  bodies.append('<body pos="0 0 0.305"><freejoint/><geom type="box" size=".08 .08 .1"/></body>')  # stacked on box 0
  return ('<mujoco><option timestep="0.004" sleep_tolerance="0.02"><flag sleep="enable" island="enable" nativeccd="disable"/></option>'
          '<worldbody><geom type="plane" size="10 10 .1"/>' + "".join(bodies) + "</worldbody></mujoco>")


@pytest.mark.gpu
def test_gpu_sleep_nv78_tree_solver_and_host_roundtrip():
  """13 free boxes (nv 78 > 64: the per-island solver path, MjhModel.tree_solve) with sleeping: tables vs the oracle while everything
  settles and sleeps; then put_data / get_data_into carry the sleep state."""
  xml = _many_boxes_xml()
  mjm, m, d = _gpu(xml, 2, nconmax=80, njmax=320)
  assert m.tree_solve == 1 and m.sleep_enabled == 1
  s = ref.RefSim(mjm, nconmax=80, njmax=320)
  v = d.qvel.numpy()
  v[1, 6 * 5 + 2] = 1.0  # world 1: box 5 hops
  d.qvel.assign(v)
  sims = [s, ref.RefSim(mjm, nconmax=80, njmax=320)]
  sims[1].qvel[6 * 5 + 2] = 1.0
  for step in range(120):
    mjw.step(m, d)
    for w, sim in enumerate(sims):
      sim.step()
      _check_tables(d, w, sim, f"step {step} world {w}")
      assert relerr(d.qpos.numpy()[w], sim.qpos) < 1e-4
  assert (d.overflow.numpy() == 0).all()
  assert (d.ntree_awake.numpy() == 0).all(), d.tree_asleep.numpy()
  ta = d.tree_asleep.numpy()[0]
  assert {int(ta[0]), int(ta[12])} == {0, 12}  # the stack is one sleep cycle
  # host round trip
  class Host:
    pass

  res = Host()
  mjw.get_data_into(res, mjm, d, world_id=1)
  assert (res.tree_asleep == ta).all() and (res.tree_awake == 0).all() and res.body_awake[0] == mjw.SleepState.STATIC and (res.body_awake[1:] == mjw.SleepState.ASLEEP).all()
  mjd = mjw.mjcf.MjData(mjm)
  mjd.qpos[:] = res.qpos
  mjd.tree_asleep = res.tree_asleep
  mjd.body_awake = res.body_awake
  d2 = mjw.put_data(mjm, mjd, nworld=2, nconmax=80, njmax=320)
  assert (d2.tree_asleep.numpy() == ta).all() and (d2.tree_awake.numpy() == 0).all()
  mjw.step(m, d2)
  assert (d2.tree_awake.numpy() == 0).all() and int(d2.nacon.numpy()[0]) == 0 and (d2.qvel.numpy() == 0).all()
