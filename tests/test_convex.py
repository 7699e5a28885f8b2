"""Convex narrowphase (GJK / EPA): the oracle against the numbers the reference's own tests hold, then the GPU against the oracle.

Reference-held vectors: /root/reference/mujoco_warp/_src/collision_gjk_test.py (test names cited per case; geometry restated
from their inline XML, expected values as asserted there -- assertAlmostEqual = 7 places unless a `places` is given).
"""

import numpy as np
import pytest

from oracle import ref

SPHERE, CAPSULE, ELLIPSOID, CYLINDER, BOX = 2, 3, 4, 5, 6
I3 = np.eye(3)


def _euler_xyz(deg):  # MuJoCo default eulerseq "xyz", intrinsic
  a, b, c = np.radians(deg)
  rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
  ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
  rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
  return rx @ ry @ rz


def test_spheres_distance():  # :307
  dist, _, x1, x2, _ = ref.ccd(SPHERE, [-1.5, 0, 0], I3, [1, 0, 0], SPHERE, [1.5, 0, 0], I3, [1, 0, 0])
  assert dist == 1.0 and x1[0] == -0.5 and x2[0] == 0.5


def test_spheres_touching():  # :325
  dist, _, _, _, _ = ref.ccd(SPHERE, [-1, 0, 0], I3, [1, 0, 0], SPHERE, [1, 0, 0], I3, [1, 0, 0])
  assert dist == 0.0


def test_sphere_sphere_contact():  # :368
  dist, _, _, _, _ = ref.ccd(SPHERE, [-1, 0, 0], I3, [3, 0, 0], SPHERE, [3, 0, 0], I3, [3, 0, 0])
  assert abs(dist + 2) < 5e-8


def test_box_box_contact():  # :384
  dist, _, x1, x2, _ = ref.ccd(BOX, [-1, 0, 0], I3, [2.5, 2.5, 2.5], BOX, [1.5, 0, 0], I3, [1, 1, 1])
  n = (x1 - x2) / np.linalg.norm(x1 - x2)
  assert abs(dist + 1) < 5e-8
  np.testing.assert_allclose(n, [1, 0, 0], atol=5e-8)


def test_cylinder_cylinder_contact():  # :467
  dist, _, _, _, _ = ref.ccd(CYLINDER, [0, 0, 0], I3, [1, 0.5, 0], CYLINDER, [1.999, 0, 0], I3, [1, 0.5, 0])
  assert abs(dist + 0.001) < 5e-8


def test_cylinder_capsule():  # :698
  dist, _, _, _, _ = ref.ccd(CYLINDER, [0, 0, 0], I3, [2, 4, 0], CAPSULE, [0, 0, 5], I3, [1, 1, 0])
  assert abs(dist + 1.0) < 5e-7


def test_cylinder_box():  # :668 (ccd_iterations=50)
  pos = [0.00015228791744448245, -0.00074981129728257656, 0.29839199781417846680]
  rot = [0.99996972084045410156, 0.00776371126994490623, -0.00043433305108919740, -0.00776385562494397163, 0.99996984004974365234,
         -0.00033095158869400620, 0.00043175052269361913, 0.00033431366318836808, 0.99999988079071044922]
  dist, _, _, _, _ = ref.ccd(BOX, [0, 0, 0], I3, [1, 1, 0.1], CYLINDER, pos, rot, [0.1, 0.2, 0.3], iterations=50)
  assert abs(dist + 0.0016624178339902445) < 5e-7


def test_box_box_early():  # :564
  pos1 = [0.07524700462818145752, -0.13524700701236724854, 0.12491077929735183716]
  rot1 = [1.0, 0.00000000006837434091, 0.00000000080494955146, -0.00000000006837435479, 1.0, 0.00000002552030764491,
          -0.00000000080494955146, -0.00000002552030764491, 1.0]
  pos2 = [0.07524700462818145752, -0.13524700701236724854, 0.17094630002975463867]
  rot2 = [1.0, 0.00000000006837435479, -0.00000000018000903546, -0.00000000006837434091, 1.0, 0.00000004174798817758,
          0.00000000018000903546, -0.00000004174798817758, 1.0]
  s = [0.025] * 3
  dist, _, _, _, _ = ref.ccd(BOX, pos1, rot1, s, BOX, pos2, rot2, s)
  assert abs(dist + 0.0039644796979132323) < 5e-7  # ("actual depth" quoted by the reference; its float32 run asserts 6 places)


def test_box_box_early2():  # :606
  pos1 = [0.07122065126895904541, -0.19126638770103454590, 0.29129269719123840332]
  rot1 = [0.99999558925628662109, 0.00258362153545022011, 0.00148368685040622950, -0.00258197076618671417, 0.99999606609344482422,
          -0.00111339206341654062, -0.00148655765224248171, 0.00110955617856234312, 0.99999833106994628906]
  pos2 = [0.07183132320642471313, -0.13260576128959655762, 0.30987158417701721191]
  rot2 = [0.99827724695205688477, 0.02493947930634021759, 0.05311207473278045654, 0.00605074502527713776, 0.85659545660018920898,
          -0.51595354080200195312, -0.05836316198110580444, 0.51538598537445068359, 0.85496860742568969727]
  s = [0.025] * 3
  dist, _, _, _, _ = ref.ccd(BOX, pos1, rot1, s, BOX, pos2, rot2, s)
  assert abs(dist + 2.515764037690309e-06) < 5e-8  # "actual depth" of the reference's comment


def test_box_box_max():  # :900
  rot1 = [0.8378595710, 0.3184406757, -0.4433811009, 0.5328434706, -0.3006005287, 0.7910227776, 0.1186132580, -0.8990187645, -0.4215400815]
  pos1 = [6.0405082703, 21.4734001160, 0.036854844]
  rot2 = [-0.6420212388, -0.0727036372, -0.7632319927, 0.3801730871, -0.8946756721, -0.2345722020, -0.6657907367, -0.4407605529, 0.6020406485]
  pos2 = [6.0641078949, 21.4842395782, 0.0212156791]
  dist, _, _, _, _ = ref.ccd(BOX, pos1, rot1, [0.018, 0.018, 0.01], BOX, pos2, rot2, [0.020, 0.020, 0.04])
  assert abs(dist + 0.03636224) < 5e-7


def test_box_box_max2():  # :952
  rot2 = [0.9999999404, 0.0004342802, -0.0001755831, -0.0004346797, 0.9999973178, -0.0022819033, 0.0001745916, 0.0022819792, 0.9999974370]
  pos2 = [0.0885666460, 0.0911745951, 0.1250119805]
  dist, _, _, _, _ = ref.ccd(BOX, [0, 0, 0], I3, [0.5, 0.5, 0.1], BOX, pos2, rot2, [0.025] * 3)
  assert abs(dist + 4.936969499999555e-05) < 5e-8  # "MJC 64 bit precision" value of the reference's comment


def test_box_box_diagonal_depth():  # :866
  pos2 = [0.135535001754761, -0.195535004138947, 0.124984227120876]
  rot2 = [1.0, 0.000000000048563, -0.000000135524601, -0.000000000048577, 1.0, -0.000000103374248, 0.000000135524601, 0.000000103374248, 1.0]
  dist, _, _, _, _ = ref.ccd(BOX, [0, 0, 0], I3, [0.5, 0.5, 0.1], BOX, pos2, rot2, [0.025] * 3)
  assert abs(dist + 1.5778851595232846e-05) < 5e-8


def test_box_box_shallow_depth():  # :483
  dist, _, _, _, _ = ref.ccd(BOX, [0, 0, 0.19972974], I3, [0.2] * 3, BOX, [0, 0, 0.49947918], I3, [0.1] * 3)
  assert abs(dist + 0.00025054812) < 5e-8


def _quat_mat(q):
  w, x, y, z = np.asarray(q, dtype=float) / np.linalg.norm(q)
  return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                   [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def test_box_box_multicontact_counts():
  """multicontact (collision_gjk.py:2076): contact counts the reference's tests assert for box pairs."""
  # test_box_box_shallow_penetration :483 -- 4 contacts, depth as above
  dist, n, _, _, _ = ref.ccd(BOX, [0, 0, 0.19972974], I3, [0.2] * 3, BOX, [0, 0, 0.49947918], I3, [0.1] * 3, multiccd=True)
  assert n == 4 and abs(dist + 0.00025054812) < 5e-8
  # test_box_edge :499 -- an edge resting on a face: 2 contacts
  _, n, _, _, _ = ref.ccd(BOX, [0, 0, 2], I3, [1, 1, 1], BOX, [0, 0, 4.4], _euler_xyz([0, 90, 40]), [1, 1, 1], multiccd=True)
  assert n == 2
  # test_box_box_ccd :513 -- a box on a slab: 4 contacts
  _, n, _, _, _ = ref.ccd(BOX, [0, 0, 1.9], I3, [1, 1, 1], BOX, [0, 0, 0], I3, [10, 10, 1], multiccd=True)
  assert n == 4
  # test_box_box_ccd2 :548 -- rotated, shifted box on a box: 4 contacts
  _, n, _, _, _ = ref.ccd(BOX, [0, 0, 2], I3, [1, 1, 1], BOX, [0, 1, 3.99], _euler_xyz([0, 0, 40]), [1, 1, 1], multiccd=True)
  assert n == 4
  # test_box_box_diagonal :866 -- 4 contacts
  pos2 = [0.135535001754761, -0.195535004138947, 0.124984227120876]
  rot2 = [1.0, 0.000000000048563, -0.000000135524601, -0.000000000048577, 1.0, -0.000000103374248, 0.000000135524601, 0.000000103374248, 1.0]
  dist, n, _, _, _ = ref.ccd(BOX, [0, 0, 0], I3, [0.5, 0.5, 0.1], BOX, pos2, rot2, [0.025] * 3, multiccd=True)
  assert n == 4 and abs(dist + 1.5778851595232846e-05) < 5e-8


def test_box_edge_flipped_witness_points():  # :986
  dist, n, _, _, wit = ref.ccd(BOX, [1.10164554, -0.11389316, 0.74], _quat_mat([-0.348312918, 0, 0, 0.937378318]), [0.65, 0.48, 0.04], BOX,
                               [1.4, 0, 1.425], I3, [0.1, 1.2, 1.4], multiccd=True)
  assert n == 2
  np.testing.assert_allclose(wit[0, 0], [1.907368, -0.052973, 0.700000], atol=1e-4)
  np.testing.assert_allclose(wit[0, 1], [1.30000, -0.052973, 0.700000], atol=1e-4)


@pytest.mark.parametrize("seed", range(40))
def test_ccd_against_closed_forms(seed):
  """Ellipsoids with equal radii are spheres and capsules against them have a closed form: distance, witness points and the
  separated / shallow / deep branches of gjk_phase against analytic answers."""
  rng = np.random.default_rng(seed)
  r1, r2 = rng.uniform(0.1, 0.5, 2)
  c2 = rng.normal(size=3)
  c2 *= rng.uniform(0.2, 1.2) / np.linalg.norm(c2)
  q = rng.normal(size=(3, 3))
  rot, _ = np.linalg.qr(q)
  d_true = np.linalg.norm(c2) - r1 - r2
  for t1, t2 in ((ELLIPSOID, ELLIPSOID), (SPHERE, ELLIPSOID)):
    dist, n, x1, x2, _ = ref.ccd(t1, [0, 0, 0], I3, [r1] * 3, t2, c2, rot, [r2] * 3)
    assert n == 1
    # separated shapes (and spheres whose centre stays outside the other shape) are answered by GJK to its tolerance; deep
    # penetrations go through EPA, whose polytope approaches a curved surface only slowly: within `iterations` = 35 support
    # points the depth of two overlapping spheres is a few 1e-4 short (the reference's algorithm, not an artefact of this port)
    exact = d_true > 0 or (t1 == SPHERE and np.linalg.norm(c2) > r2)
    assert abs(dist - d_true) < (2e-6 if exact else 1e-3), (t1, dist, d_true)
    assert dist >= d_true - 2e-6  # a polytope inside the Minkowski difference can only under-estimate the depth
    u = c2 / np.linalg.norm(c2)
    np.testing.assert_allclose(x1, u * r1, atol=2e-6 if exact else 3e-2)
    np.testing.assert_allclose(x2, c2 - u * r2, atol=2e-6 if exact else 3e-2)


# ---- GPU vs oracle -------------------------------------------------------------------------------------------------------------
CONVEX_XML = """
<mujoco>
  <option timestep="0.003"/>
  <worldbody>
    <geom name="floor" type="plane" size="0 0 .05"/>
    <body name="ell" pos="0 0 .2"><freejoint/><geom type="ellipsoid" size=".12 .08 .05"/></body>
    <body name="cyl" pos=".4 0 .2"><freejoint/><geom type="cylinder" size=".06 .1"/></body>
    <body name="cap" pos=".8 0 .2"><freejoint/><geom type="capsule" size=".04 .1"/></body>
    <body name="box" pos="1.2 0 .2"><freejoint/><geom type="box" size=".08 .06 .05"/></body>
    <body name="sph" pos="1.6 0 .2"><freejoint/><geom type="sphere" size=".07"/></body>
    <body name="ell2" pos="2.0 0 .2"><freejoint/><geom type="ellipsoid" size=".05 .09 .07"/></body>
    <body name="cyl2" pos="2.4 0 .2"><freejoint/><geom type="cylinder" size=".08 .04"/></body>
  </worldbody>
</mujoco>
"""
# body index (qpos block) of each shape and the convex pairs exercised: (a, b)
_SHAPES = {"ell": 0, "cyl": 1, "cap": 2, "box": 3, "sph": 4, "ell2": 5, "cyl2": 6}
_PAIRS = [("sph", "ell"), ("cap", "ell"), ("cap", "cyl"), ("ell", "ell2"), ("ell", "cyl"), ("ell", "box"), ("cyl", "cyl2"), ("cyl", "box")]


def _pose_pairs(rng, gapscale):
  """qpos with every pair of _PAIRS placed near contact somewhere far from the other pairs (the two shapes of a pair are moved,
  every pair gets its own copy of the state: returned as a list of qpos vectors)."""
  out = []
  for a, b in _PAIRS:
    q = np.zeros(7 * len(_SHAPES))
    for name, i in _SHAPES.items():
      q[7 * i : 7 * i + 3] = [3.0 * i, 5.0, 1.0 + 0.5 * i]  # parked apart, off the floor
      q[7 * i + 3] = 1.0
    for k, name in enumerate((a, b)):
      i = _SHAPES[name]
      quat = rng.normal(size=4)
      q[7 * i + 3 : 7 * i + 7] = quat / np.linalg.norm(quat)
    ia, ib = _SHAPES[a], _SHAPES[b]
    direction = rng.normal(size=3)
    direction /= np.linalg.norm(direction)
    q[7 * ia : 7 * ia + 3] = [0, 0, 2.0]
    q[7 * ib : 7 * ib + 3] = np.array([0, 0, 2.0]) + direction * gapscale * rng.uniform(0.1, 0.2)
    out.append(q)
  return out


@pytest.mark.gpu
@pytest.mark.parametrize("margin", [0.0, 0.03])
@pytest.mark.parametrize("seed", range(6))
def test_gpu_convex_contacts_match_oracle(seed, margin):
  import mujoco_warp_amd as mjw

  # margin > 0: support points are inflated by half of it and contacts appear before the surfaces touch (dist in (0, margin))
  mjm = mjw.mjcf.from_xml_string(CONVEX_XML.replace("<worldbody>", f'<default><geom margin="{margin}"/></default><worldbody>'))
  rng = np.random.default_rng(seed)
  qs = _pose_pairs(rng, 1.0)
  m = mjw.put_model(mjm)
  d = mjw.put_data(mjm, mjw.MjData(mjm), nworld=len(qs), nconmax=16, njmax=64)
  d.qpos.assign(np.asarray(qs, dtype=np.float32))
  mjw.kinematics(m, d)
  mjw.collision(m, d)
  ncon = d.ws_ncon.numpy()
  adr = d.ws_conadr.numpy()
  nhit = 0
  for w, q in enumerate(qs):
    s = ref.RefSim(mjm, nconmax=16, njmax=64)
    s.qpos[:] = q
    s.stage("kinematics")
    s.stage("collision")
    assert int(ncon[w]) == s.ncon, (_PAIRS[w], int(ncon[w]), s.ncon)
    nhit += s.ncon
    for c in range(s.ncon):
      o = int(adr[w]) + c
      assert tuple(d.contact.geom.numpy()[o]) == tuple(s.con_geom[c])
      depth = abs(s.con_dist[c])
      # GJK answers (separated / shallow) agree to float32 rounding; EPA depths of curved shapes are only as converged as 35
      # support points allow (see test_ccd_against_closed_forms), and float32 / float64 may stop on neighbouring faces
      tol = 2e-6 + 5e-3 * depth
      assert abs(d.contact.dist.numpy()[o] - s.con_dist[c]) < tol, (_PAIRS[w], d.contact.dist.numpy()[o], s.con_dist[c])
      np.testing.assert_allclose(d.contact.pos.numpy()[o], s.con_pos[c], atol=5e-5 + 0.25 * depth)  # (flat faces: the witness point is not unique)
      np.testing.assert_allclose(d.contact.frame.numpy()[o].reshape(9)[:3], s.con_frame[c][:3], atol=2e-4 if s.con_dist[c] > s.con_includemargin[c] else 6e-2)  # (EPA normal = a facet of a <= 40-vertex polytope)
  assert nhit >= 3
  assert (d.overflow.numpy() == 0).all()


CONVEX_SCENE_XML = """
<mujoco>
  <option timestep="0.002"/>
  <worldbody>
    <geom name="floor" type="plane" size="0 0 .05"/>
    <geom name="table" type="box" size=".6 .3 .05" pos="0 0 .05"/>
    <geom name="log" type="cylinder" size=".08 .25" pos="0 1 .08" euler="90 0 0"/>
    <geom name="dome" type="ellipsoid" size=".3 .3 .12" pos="1.2 0 .12"/>
    <body name="egg" pos="-.4 0 .149"><freejoint/><geom type="ellipsoid" size=".09 .06 .05"/></body>
    <body name="can" pos="0 0 .179"><freejoint/><geom type="cylinder" size=".05 .08"/></body>
    <body name="roll" pos=".4 0 .149" euler="90 0 0"><freejoint/><geom type="cylinder" size=".05 .1"/></body>
    <body name="cross" pos="0 .9 .209" euler="0 90 0"><freejoint/><geom type="cylinder" size=".05 .15"/></body>
    <body name="pill" pos="0 1.15 .199" euler="0 90 0"><freejoint/><geom type="capsule" size=".04 .12"/></body>
    <body name="ball" pos="1.2 0 .299"><freejoint/><geom type="sphere" size=".06"/></body>
    <body name="pebble" pos="1.0 .1 .27"><freejoint/><geom type="ellipsoid" size=".05 .04 .03"/></body>
  </worldbody>
</mujoco>
"""


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["Newton", "CG"])
def test_gpu_convex_scene_steps_match_oracle(solver):
  """Bodies resting / rolling on convex supports (ellipsoid-box, cylinder-box flat and rolling, crossed cylinders, capsule-cylinder,
  sphere-ellipsoid, ellipsoid-ellipsoid): one step from the same state along the oracle's trajectory."""
  import mujoco_warp_amd as mjw
  from tests.conftest import relerr

  mjm = mjw.mjcf.from_xml_string(CONVEX_SCENE_XML)
  mjm.opt.solver = int(mjw.SolverType.NEWTON if solver == "Newton" else mjw.SolverType.CG)
  s = ref.RefSim(mjm, nconmax=32, njmax=128, tolerance=1e-6)
  s.reset()
  m = mjw.put_model(mjm)
  d = mjw.put_data(mjm, mjw.MjData(mjm), nworld=2, nconmax=32, njmax=128)
  worst_q = worst_v = 0.0
  ncon_total = 0
  for i in range(60):
    for name in ("qpos", "qvel", "qacc_warmstart"):
      getattr(d, name).assign(np.tile(getattr(s, name).astype(np.float32), (2, 1)))
    mjw.step(m, d)
    s.step()
    ncon_total += s.ncon
    assert int(d.ws_ncon.numpy()[1]) == s.ncon
    worst_q = max(worst_q, relerr(d.qpos.numpy()[1], s.qpos))
    worst_v = max(worst_v, relerr(d.qvel.numpy()[1], s.qvel))
  assert ncon_total >= 6 * 60
  assert worst_q <= 1e-5, worst_q
  assert worst_v <= 5e-3, worst_v
  assert (d.overflow.numpy() == 0).all()


# ---- broadphase of models with GJK pairs: k_broad_mask (NXN: a workgroup per world, bit mask over the pair list) against the sweep-and-prune
# launch (k_ccd_broad) and against the oracle, for every filter combination, with margins, gaps and an explicit <contact><pair> ----
CROWD_XML = CONVEX_SCENE_XML.replace("<worldbody>", """<default><geom margin="0.02" gap="0.005"/></default>
  <contact><pair geom1="table" geom2="dome" margin="0.3" gap="0.1" condim="3"/></contact>
  <worldbody>""").replace('name="table"', 'name="table" margin="0.05"').replace('name="dome"', 'name="dome" gap="0.02"')


@pytest.mark.gpu
@pytest.mark.parametrize("bfilter", [1, 2, 3, 5, 7, 10, 11, 15])
def test_gpu_broad_mask_equals_sap_and_oracle(bfilter):
  import mujoco_warp_amd as mjw

  mjm = mjw.mjcf.from_xml_string(CROWD_XML)
  rng = np.random.default_rng(5)
  nworld = 24
  q0 = np.asarray(mjw.MjData(mjm).qpos, dtype=np.float64)
  qs = np.tile(q0, (nworld, 1))
  for w in range(1, nworld):  # world 0: the authored poses; the others: the free bodies thrown together over the table
    for b in range(mjm.nq // 7):
      qs[w, 7 * b: 7 * b + 3] = [rng.uniform(-0.5, 0.5), rng.uniform(-0.3, 0.3), rng.uniform(0.12, 0.35)]
      quat = rng.normal(size=4)
      qs[w, 7 * b + 3: 7 * b + 7] = quat / np.linalg.norm(quat)
  found = {}
  for bp in (mjw.BroadphaseType.NXN, mjw.BroadphaseType.SAP_TILE):
    m = mjw.put_model(mjm)
    m.opt.broadphase = int(bp)
    m.opt.broadphase_filter = int(bfilter)
    d = mjw.put_data(mjm, mjw.MjData(mjm), nworld=nworld, nconmax=48, njmax=192)
    d.qpos.assign(qs.astype(np.float32))
    mjw.kinematics(m, d)
    mjw.collision(m, d)
    assert (d.overflow.numpy() == 0).all()
    ncon, adr, geom = d.ws_ncon.numpy(), d.ws_conadr.numpy(), d.contact.geom.numpy()
    found[int(bp)] = (d.ws_ncollision.numpy().copy(), [tuple(map(tuple, geom[int(adr[w]): int(adr[w]) + int(ncon[w])])) for w in range(nworld)],
                      d.contact.dist.numpy()[: int(ncon.sum())].copy())
  a, b = found[int(mjw.BroadphaseType.NXN)], found[int(mjw.BroadphaseType.SAP_TILE)]
  # (the sweep prunes by its own projection test whatever the filters: its candidate COUNT is its own -- checked against the oracle's sweep
  # below --, the contacts are the same)
  assert (b[0] <= a[0]).all(), (a[0], b[0])
  assert a[1] == b[1]
  assert np.array_equal(a[2], b[2])
  ncoll = 0
  for w in range(0, nworld, 3):
    s = ref.RefSim(mjm, nconmax=48, njmax=192, broadphase=0, broadphase_filter=int(bfilter))
    s.qpos[:] = qs[w]
    s.stage("kinematics")
    s.stage("collision")
    assert int(a[0][w]) == s.ncollision, (w, int(a[0][w]), s.ncollision)
    assert [tuple(g) for g in a[1][w]] == [tuple(int(x) for x in g) for g in s.con_geom[: s.ncon]], w
    ncoll += s.ncollision
    s2 = ref.RefSim(mjm, nconmax=48, njmax=192, broadphase=1, broadphase_filter=int(bfilter))
    s2.qpos[:] = qs[w]
    s2.stage("kinematics")
    s2.stage("collision")
    assert int(b[0][w]) == s2.ncollision, (w, int(b[0][w]), s2.ncollision)
  assert ncoll >= 16


@pytest.mark.gpu
def test_gpu_nccdmax_sizes_the_epa_hand_over_list():
  """make_data(nccdmax=...) / (naccdmax=...) size Data.nccdhand and ws_ccd (reference io.py:1741-1753: CCD contacts per world / in total;
  nccdmax <= nconmax); without them the engine's default stands; models without GJK pairs allocate nothing."""
  import mujoco_warp_amd as mjw
  from mujoco_warp_amd import io as mio

  mjm = mjw.mjcf.from_xml_string(CONVEX_SCENE_XML)
  d0 = mjw.make_data(mjm, nworld=16, nconmax=32, njmax=128)
  assert d0.nccdhand == mio._ccd_handcap(16, mio._collide_ccap(int(mjw.put_model(mjm).npair), d0.concap))
  d1 = mjw.make_data(mjm, nworld=16, nconmax=32, njmax=128, nccdmax=2)
  assert d1.nccdhand == 32 and d1.nccdword < d0.nccdword
  d2 = mjw.make_data(mjm, nworld=16, nconmax=32, njmax=128, naccdmax=7)
  assert d2.nccdhand == 7
  with pytest.raises(ValueError):
    mjw.make_data(mjm, nworld=2, nconmax=8, njmax=64, nccdmax=9)
  with pytest.raises(ValueError):
    mjw.make_data(mjm, nworld=2, nconmax=8, njmax=64, nccdmax=-1)


@pytest.mark.gpu
def test_gpu_small_nccdmax_drops_pairs_and_says_so():
  """An EPA hand-over list smaller than the scene's penetrating convex pairs: the surplus pairs are dropped and every world that lost one
  carries OverflowType.CCD; with room for all of them the contacts are those of the default allocation."""
  import mujoco_warp_amd as mjw

  mjm = mjw.mjcf.from_xml_string(CONVEX_SCENE_XML)
  m = mjw.put_model(mjm)
  out = {}
  for tag, kw in (("default", {}), ("roomy", dict(nccdmax=16)), ("tight", dict(naccdmax=3))):
    d = mjw.put_data(mjm, mjw.MjData(mjm), nworld=4, nconmax=32, njmax=128, **kw)
    for _ in range(40):
      mjw.step(m, d)
    out[tag] = (d.ws_ncon.numpy().copy(), d.overflow.numpy().copy(), d.qpos.numpy().copy())
  assert (out["default"][1] == 0).all() and (out["roomy"][1] == 0).all()
  assert np.array_equal(out["default"][0], out["roomy"][0]) and np.array_equal(out["default"][2], out["roomy"][2])
  assert (out["tight"][1] & int(mjw.OverflowType.CCD)).any()
  assert out["tight"][0].sum() < out["default"][0].sum() or (out["tight"][2] != out["default"][2]).any()
  assert np.isfinite(out["tight"][2]).all()


def _cluster_xml(nbody=7, seed=3):
  """Free bodies of three or four geoms each (sphere, capsule, box, ellipsoid at offsets of the body's size) over a table on legs: the geom
  groups of k_broad_mask's pre-test hold several geoms, the static scenery is a group per geom."""
  r = np.random.default_rng(seed)
  lines = ['<mujoco><option timestep="0.003"/><default><geom margin="0.01" gap="0.002"/></default>',
           '<contact><pair geom1="top" geom2="g0_0" margin="0.2" gap="0.05" condim="3"/></contact><worldbody>',
           '<geom name="floor" type="plane" size="0 0 .05"/><geom name="top" type="box" size=".6 .5 .02" pos="0 0 .3"/>']
  for k, (x, y) in enumerate(((.55, .45), (-.55, .45), (.55, -.45), (-.55, -.45))):
    lines.append(f'<geom name="leg{k}" type="cylinder" size=".03 .14" pos="{x} {y} .14"/>')
  for b in range(nbody):
    lines.append(f'<body name="b{b}" pos="{r.uniform(-.5, .5):.3f} {r.uniform(-.4, .4):.3f} {r.uniform(.4, .6):.3f}"><freejoint/>')
    for g in range(3 + b % 2):
      t = ("sphere", "capsule", "box", "ellipsoid")[(b + g) % 4]
      size = {"sphere": ".03", "capsule": ".02 .04", "box": ".03 .02 .025", "ellipsoid": ".035 .02 .025"}[t]
      lines.append(f'<geom name="g{b}_{g}" type="{t}" size="{size}" pos="{r.uniform(-.08, .08):.3f} {r.uniform(-.08, .08):.3f} {r.uniform(-.08, .08):.3f}"'
                   + (' margin="0.03"' if g == 1 else "") + "/>")
    lines.append("</body>")
  lines.append("</worldbody></mujoco>")
  return "\n".join(lines)


@pytest.mark.gpu
@pytest.mark.parametrize("bfilter", [1, 2, 3, 7, 15])
def test_gpu_broad_mask_group_pretest_keeps_every_candidate(bfilter):
  """k_broad_mask tests one bounding sphere per pair of geom groups before the geom pairs (round 5): with several geoms per body, geom
  margins, an explicit pair of its own margin, a plane and static scenery the candidate count and the contact list of every world are the
  oracle's NXN broadphase's (which tests every pair, collision_driver.py:278-334) and the sweep-and-prune launch's."""
  import mujoco_warp_amd as mjw

  mjm = mjw.mjcf.from_xml_string(_cluster_xml())
  m = mjw.put_model(mjm)
  assert m.ncullgroup == 6 + 7 and 0 < m.ncullpair < m.npair  # floor, top, four legs: a group each; a group per body
  assert int((m.cull_pair.numpy()[:, 3] >> 24).sum()) == m.npair and int((m.cull_pair.numpy()[:, 0] < 0).sum()) == 1 and int((m.cull_pair.numpy()[:, 3] >> 24).max()) <= 16
  rng = np.random.default_rng(11)
  nworld = 40
  q0 = np.asarray(mjw.MjData(mjm).qpos, dtype=np.float64)
  qs = np.tile(q0, (nworld, 1))
  for w in range(1, nworld):  # the bodies thrown together over (and under) the table, closer with the world index
    spread = 0.6 * (1.0 - w / nworld) + 0.08
    for b in range(mjm.nq // 7):
      qs[w, 7 * b: 7 * b + 3] = [rng.uniform(-spread, spread), rng.uniform(-spread, spread), rng.uniform(0.05, 0.5)]
      quat = rng.normal(size=4)
      qs[w, 7 * b + 3: 7 * b + 7] = quat / np.linalg.norm(quat)
  found = {}
  for bp in (mjw.BroadphaseType.NXN, mjw.BroadphaseType.SAP_TILE):
    m = mjw.put_model(mjm)
    m.opt.broadphase = int(bp)
    m.opt.broadphase_filter = int(bfilter)
    d = mjw.put_data(mjm, mjw.MjData(mjm), nworld=nworld, nconmax=96, njmax=384)
    d.qpos.assign(qs.astype(np.float32))
    mjw.kinematics(m, d)
    mjw.collision(m, d)
    assert (d.overflow.numpy() == 0).all()
    ncon, adr, geom = d.ws_ncon.numpy(), d.ws_conadr.numpy(), d.contact.geom.numpy()
    found[int(bp)] = (d.ws_ncollision.numpy().copy(), [tuple(map(tuple, geom[int(adr[w]): int(adr[w]) + int(ncon[w])])) for w in range(nworld)])
  a, b = found[int(mjw.BroadphaseType.NXN)], found[int(mjw.BroadphaseType.SAP_TILE)]
  assert a[1] == b[1]
  ncoll = 0
  for w in range(0, nworld, 4):
    s = ref.RefSim(mjm, nconmax=96, njmax=384, broadphase=0, broadphase_filter=int(bfilter))
    s.qpos[:] = qs[w]
    s.stage("kinematics")
    s.stage("collision")
    assert int(a[0][w]) == s.ncollision, (w, int(a[0][w]), s.ncollision)
    assert [tuple(g) for g in a[1][w]] == [tuple(int(x) for x in g) for g in s.con_geom[: s.ncon]], w
    ncoll += s.ncollision
  assert ncoll >= 40


@pytest.mark.gpu
def test_gpu_epa_groups_shrink_for_a_high_degree_vertex():
  """A cone of 300 sides: its apex has 300 adjacent polygons and its base is a 300-gon, so one EPA lane group's LDS (polytope, feature
  lists of 11 x 300 words, polygon buffers of 18 x 300) is 22 KB -- eight groups do not fit in a CU's LDS and the launch takes fewer groups
  per workgroup.  Base down and apex down on a box: contacts as the oracle's."""
  import mujoco_warp_amd as mjw

  n = 300
  ang = 2 * np.pi * np.arange(n) / n
  pts = np.concatenate([[[0.0, 0.0, 0.12]], np.stack([0.08 * np.cos(ang), 0.08 * np.sin(ang), np.zeros(n)], axis=1)])
  verts = " ".join(f"{x:.6f}" for x in pts.reshape(-1))
  xml = f"""<mujoco><option timestep="0.002"/><asset><mesh name="cone" vertex="{verts}"/></asset>
  <worldbody><geom name="table" type="box" size=".5 .5 .05" pos="0 0 .05"/>
    <body name="down" pos="-.2 0 .0995"><freejoint/><geom type="mesh" mesh="cone"/></body>
    <body name="up" pos=".2 0 .2195" euler="180 0 0"><freejoint/><geom type="mesh" mesh="cone"/></body>
  </worldbody></mujoco>"""
  mjm = mjw.mjcf.from_xml_string(xml)
  m = mjw.put_model(mjm)
  assert m.nmeshdegmax >= 300 and m.npolygonmax >= 300
  s = ref.RefSim(mjm, nconmax=16, njmax=64)
  s.reset()
  s.stage("kinematics")
  s.stage("collision")
  d = mjw.put_data(mjm, mjw.MjData(mjm), nworld=5, nconmax=16, njmax=64)
  mjw.kinematics(m, d)
  mjw.collision(m, d)
  assert (d.overflow.numpy() == 0).all()
  ncon = d.ws_ncon.numpy()
  assert (ncon == s.ncon).all() and s.ncon >= 5, (ncon, s.ncon)  # four points under the base, one at the apex
  adr = int(d.ws_conadr.numpy()[3])
  dist, pos = d.contact.dist.numpy()[adr: adr + s.ncon], d.contact.pos.numpy()[adr: adr + s.ncon]
  assert np.abs(dist - np.asarray(s.con_dist[: s.ncon])).max() <= 1e-5
  assert np.abs(np.sort(pos[:, 2]) - np.sort(np.asarray(s.con_pos[: s.ncon])[:, 2])).max() <= 1e-4


@pytest.mark.gpu
def test_gpu_broad_mask_many_small_bodies():
  """120 free bodies of two geoms each (spheres, capsules, ellipsoids: GJK pairs among them) = 28,800 pairs in 7,260 rows of the pre-test's
  table: the rows go through the workgroup in chunks of BMASK_ROWCHUNK (the survivors' list in LDS stays bounded) and the mask is expanded
  in more than two trips -- candidate counts and contact lists as the oracle's NXN broadphase."""
  import mujoco_warp_amd as mjw

  r = np.random.default_rng(7)
  lines = ['<mujoco><option timestep="0.003"/><worldbody><geom name="floor" type="plane" size="0 0 .05"/>']
  n = 120
  for b in range(n):
    lines.append(f'<body pos="{r.uniform(-.6, .6):.3f} {r.uniform(-.6, .6):.3f} {r.uniform(.02, .5):.3f}"><freejoint/>')
    for g in range(2):
      t = ("sphere", "capsule", "ellipsoid")[(b + g) % 3]
      size = {"sphere": ".03", "capsule": ".02 .03", "ellipsoid": ".035 .02 .025"}[t]
      lines.append(f'<geom type="{t}" size="{size}" pos="{.04 * g} 0 0"/>')
    lines.append("</body>")
  lines.append("</worldbody></mujoco>")
  mjm = mjw.mjcf.from_xml_string("\n".join(lines))
  m = mjw.put_model(mjm)
  assert m.npair == 2 * n * (2 * n - 1) // 2 - n + 2 * n and m.ncullpair > 4096 and (m.npair + 63) // 64 * 2 > 512
  nworld = 3
  q0 = np.asarray(mjw.MjData(mjm).qpos, dtype=np.float64)
  qs = np.tile(q0, (nworld, 1))
  for w in range(1, nworld):
    qs[w, 0::7] *= 0.7 ** w  # (pulled together: more candidates)
    qs[w, 1::7] *= 0.7 ** w
  found = {}
  for bp in (mjw.BroadphaseType.NXN, mjw.BroadphaseType.SAP_TILE):
    mm = mjw.put_model(mjm)
    mm.opt.broadphase = int(bp)
    d = mjw.put_data(mjm, mjw.MjData(mjm), nworld=nworld, nconmax=256, njmax=64)
    d.qpos.assign(qs.astype(np.float32))
    mjw.kinematics(mm, d)
    mjw.collision(mm, d)
    assert (d.overflow.numpy() == 0).all()
    ncon, adr, geom = d.ws_ncon.numpy(), d.ws_conadr.numpy(), d.contact.geom.numpy()
    found[int(bp)] = (d.ws_ncollision.numpy().copy(), [[tuple(int(x) for x in g) for g in geom[int(adr[w]): int(adr[w]) + int(ncon[w])]] for w in range(nworld)])
  nxn, sap = found[int(mjw.BroadphaseType.NXN)], found[int(mjw.BroadphaseType.SAP_TILE)]
  assert nxn[1] == sap[1]  # the same contacts, in the same order, through either broadphase (the narrowphase is shared)
  total = 0
  for w in range(nworld):
    s = ref.RefSim(mjm, nconmax=256, njmax=64, broadphase=0, broadphase_filter=int(m.opt.broadphase_filter))
    s.qpos[:] = qs[w]
    s.stage("kinematics")
    s.stage("collision")
    assert int(nxn[0][w]) == s.ncollision, (w, int(nxn[0][w]), s.ncollision)
    # (random placements interpenetrate deeply: which of a body's two geoms EPA reports in float32 / float64 may differ -- the pair sets agree closely)
    a_, b_ = set(nxn[1][w]), set(tuple(int(x) for x in g) for g in s.con_geom[: s.ncon])
    assert len(a_ & b_) >= 0.9 * max(len(a_), len(b_)), (w, len(a_), len(b_), len(a_ & b_))
    total += s.ncollision
  assert total > 300


@pytest.mark.gpu
def test_gpu_broad_mask_with_explicit_pairs_only():
  """A model whose only pairs are explicit <contact><pair>s (every geom has contype = conaffinity = 0), one of them a GJK pair: the pre-test's
  tables hold no groups, every row is 'always tested' -- contacts as the oracle's."""
  import mujoco_warp_amd as mjw
  from tests.conftest import relerr

  xml = """<mujoco><option timestep="0.003"/><default><geom contype="0" conaffinity="0"/></default>
  <worldbody><geom name="floor" type="plane" size="0 0 .05"/>
    <body name="a" pos="0 0 .06"><freejoint/><geom name="ea" type="ellipsoid" size=".08 .05 .05"/></body>
    <body name="b" pos=".05 .01 .15"><freejoint/><geom name="eb" type="ellipsoid" size=".06 .05 .04"/></body>
  </worldbody>
  <contact><pair geom1="floor" geom2="ea"/><pair geom1="ea" geom2="eb" margin="0.01"/></contact></mujoco>"""
  mjm = mjw.mjcf.from_xml_string(xml)
  m = mjw.put_model(mjm)
  assert m.npair == 2 and m.ncullgeom == 0 and m.ncullgroup == 0 and (m.cull_pair.numpy()[:, 0] == -1).all()
  s = ref.RefSim(mjm, nconmax=16, njmax=64)
  s.reset()
  d = mjw.put_data(mjm, mjw.MjData(mjm), nworld=3, nconmax=16, njmax=64)
  seen = 0
  for i in range(60):
    for name in ("qpos", "qvel", "qacc_warmstart"):
      getattr(d, name).assign(np.tile(getattr(s, name).astype(np.float32), (3, 1)))
    mjw.step(m, d)
    s.step()
    assert int(d.ws_ncon.numpy()[2]) == s.ncon, i
    seen += s.ncon
    assert relerr(d.qpos.numpy()[2], s.qpos) <= 2e-5
  assert seen >= 60 and (d.overflow.numpy() == 0).all()


@pytest.mark.gpu
def test_gpu_broad_mask_group_pretest_with_per_world_bounding_radii():
  """Domain randomisation: Model.geom_rbound and geom_margin batched per world (reference types.py:822-833).  The group spheres of
  k_broad_mask's pre-test are measured in every world from that world's radii and margins: the candidate count of every world equals a
  brute-force count of the plane / sphere tests over the whole pair list (collision_driver.py:278-334) with that world's values."""
  import mujoco_warp_amd as mjw

  nworld = 48
  mjm = mjw.mjcf.from_xml_string(_cluster_xml(nbody=8, seed=5))
  m = mjw.put_model(mjm, batch_sizes={"geom_rbound": nworld, "geom_margin": nworld})
  rng = np.random.default_rng(21)
  rb0, mg0 = m.geom_rbound.numpy()[0].copy(), m.geom_margin.numpy()[0].copy()
  rb = rb0[None, :] * rng.uniform(0.6, 2.5, (nworld, len(rb0)))  # (planes keep radius 0)
  mg = mg0[None, :] * rng.uniform(0.0, 3.0, (nworld, len(mg0)))
  m.geom_rbound.assign(rb.astype(np.float32))
  m.geom_margin.assign(mg.astype(np.float32))
  m.opt.broadphase = int(mjw.BroadphaseType.NXN)
  m.opt.broadphase_filter = 3  # plane + sphere
  d = mjw.put_data(mjm, mjw.MjData(mjm), nworld=nworld, nconmax=128, njmax=512)
  q0 = np.asarray(mjw.MjData(mjm).qpos, dtype=np.float64)
  qs = np.tile(q0, (nworld, 1))
  for w in range(nworld):
    spread = 0.55 * (1.0 - w / nworld) + 0.1
    for b in range(mjm.nq // 7):
      qs[w, 7 * b: 7 * b + 3] = [rng.uniform(-spread, spread), rng.uniform(-spread, spread), rng.uniform(0.05, 0.5)]
      quat = rng.normal(size=4)
      qs[w, 7 * b + 3: 7 * b + 7] = quat / np.linalg.norm(quat)
  d.qpos.assign(qs.astype(np.float32))
  mjw.kinematics(m, d)
  mjw.collision(m, d)
  got = d.ws_ncollision.numpy()
  x, R = d.geom_xpos.numpy().astype(np.float64), d.geom_xmat.numpy().astype(np.float64).reshape(nworld, -1, 3, 3)
  rbf, mgf, gap = m.geom_rbound.numpy().astype(np.float64), m.geom_margin.numpy().astype(np.float64), m.geom_gap.numpy().astype(np.float64)[0]
  pairs, pid = m.nxn_geom_pair.numpy(), m.nxn_pairid.numpy()
  pm, pg = m.pair_margin.numpy().astype(np.float64), m.pair_gap.numpy().astype(np.float64)
  want = np.zeros(nworld, dtype=np.int64)
  edge = 0
  for w in range(nworld):
    for (g1, g2), e in zip(pairs, pid):
      mgn = pm[e] + pg[e] if e >= 0 else mgf[w, g1] + gap[g1] + mgf[w, g2] + gap[g2]
      r1, r2 = rbf[w, g1], rbf[w, g2]
      if r1 == 0.0 or r2 == 0.0:
        pl, ot = (g1, g2) if r1 == 0.0 else (g2, g1)
        val, bound = float(np.dot(x[w, ot] - x[w, pl], R[w, pl][:, 2])), rbf[w, ot] + mgn
      else:
        val, bound = float(np.linalg.norm(x[w, g2] - x[w, g1])), r1 + r2 + mgn
      want[w] += val <= bound
      edge += abs(val - bound) < 1e-5 * max(bound, 1.0)  # (decided inside float32 resolution: either answer is right)
  assert np.abs(got - want).sum() <= edge, (got, want, edge)
  assert want.sum() > 10 * nworld and len(set(want.tolist())) > 5


# ---- box-box through CCD + multi-contact (the reference's default; the primitive mjc_BoxBox collider needs DisableBit.NATIVECCD) ----
BOX_CCD_XML = """
<mujoco>
  <option timestep="0.003"/>
  <worldbody>
    <geom name="floor" type="plane" size="0 0 .05"/>
    <geom name="table" type="box" size=".5 .4 .05" pos="0 0 .05"/>
    <body name="flat" pos="-.25 0 .1495"><freejoint/><geom type="box" size=".08 .06 .05"/></body>
    <body name="turned" pos=".05 .1 .1495" euler="0 0 33"><freejoint/><geom type="box" size=".07 .05 .05"/></body>
    <body name="over" pos=".46 -.2 .1495"><freejoint/><geom type="box" size=".09 .05 .05"/></body>
    <body name="base" pos=".25 .15 .1395"><freejoint/><geom type="box" size=".07 .07 .04"/></body>
    <body name="top" pos=".26 .16 .2085" euler="0 0 20"><freejoint/><geom type="box" size=".03 .035 .03"/></body>
    <body name="edge" pos="-.05 -.22 .17" euler="45 0 10"><freejoint/><geom type="box" size=".06 .05 .05"/></body>
  </worldbody>
</mujoco>
"""


def _sorted_points(p):
  p = np.asarray(p, dtype=np.float64).reshape(-1, 3)
  return p[np.lexsort((np.round(p[:, 2], 4), np.round(p[:, 1], 4), np.round(p[:, 0], 4)))]


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["Newton", "CG"])
def test_gpu_box_ccd_scene_matches_oracle(solver):
  """Boxes resting flat, turned, overhanging the table edge, stacked and balanced on an edge: contact sets per box pair (count,
  distance, positions as a set) and 40 steps.  Every clipped polygon here has at most four vertices: with more, the reference prunes
  with a greedy largest-quadrilateral search (`_polygon_quad`, collision_gjk.py:1463) whose result depends on the start vertex (on
  a 5-gon the selected area varies 300-fold with the rotation of the vertex list), so float32 and float64 legitimately keep different
  corners there."""
  import mujoco_warp_amd as mjw
  from tests.conftest import relerr

  mjm = mjw.mjcf.from_xml_string(BOX_CCD_XML)
  mjm.opt.solver = int(mjw.SolverType.NEWTON if solver == "Newton" else mjw.SolverType.CG)
  s = ref.RefSim(mjm, nconmax=48, njmax=192, tolerance=1e-6)
  s.reset()
  for _ in range(10):
    s.step()
  m = mjw.put_model(mjm)
  assert m.heavy_colliders == 1 and m.epa_iterations == 16
  d = mjw.put_data(mjm, mjw.MjData(mjm), nworld=2, nconmax=48, njmax=192)
  worst_q = worst_v = 0.0
  nbox = mismatched = boundary_steps = 0
  for i in range(40):
    for name in ("qpos", "qvel", "qacc_warmstart"):
      getattr(d, name).assign(np.tile(getattr(s, name).astype(np.float32), (2, 1)))
    if i % 10 == 0:
      mjw.forward(m, d)
      s.forward()
      ncon, adr = int(d.ws_ncon.numpy()[1]), int(d.ws_conadr.numpy()[1])
      gg = d.contact.geom.numpy()[adr : adr + ncon]
      for pair in {tuple(g) for g in s.con_geom[: s.ncon]}:
        io = [c for c in range(s.ncon) if tuple(s.con_geom[c]) == pair]
        ig = [c for c in range(ncon) if tuple(gg[c]) == pair]
        if len(io) != len(ig):
          # two faces count as aligned when their normals agree within 1.6 mrad (FACE_TOL): a box rocking on its support crosses
          # that threshold, and float32 / float64 cross it a step apart (the oracle itself flickers between 2 and 4 contacts here)
          mismatched += 1
          continue
        if mjm.geom_type[pair[0]] == 6 and mjm.geom_type[pair[1]] == 6:
          nbox += len(io)
          np.testing.assert_allclose(d.contact.dist.numpy()[adr + np.array(ig)], s.con_dist[io], atol=2e-6)
          np.testing.assert_allclose(_sorted_points(d.contact.pos.numpy()[adr + np.array(ig)]), _sorted_points(s.con_pos[io]), atol=2e-5)
          np.testing.assert_allclose(d.contact.frame.numpy()[adr + ig[0]].reshape(9)[:3], s.con_frame[io[0]][:3], atol=5e-4)  # (normal = difference of two float32 witness points ~1e-3 apart)
    mjw.step(m, d)
    s.step()
    if int(d.ws_ncon.numpy()[1]) != s.ncon:
      boundary_steps += 1
      continue
    worst_q = max(worst_q, relerr(d.qpos.numpy()[1], s.qpos))
    worst_v = max(worst_v, relerr(d.qvel.numpy()[1], s.qvel))
  assert mismatched <= 2 and boundary_steps <= 8, (mismatched, boundary_steps)
  assert nbox >= 4 * 10
  assert worst_q <= 1e-5, worst_q
  assert worst_v <= 5e-3, worst_v
  assert (d.overflow.numpy() == 0).all()
