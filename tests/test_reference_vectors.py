"""Vectors and counts the REFERENCE's own tests hold (numbers that did not come from this repository).

MuJoCo C cannot be installed here or on the GPU box (profiles/round2_gpubox_probe.txt), so these are the only externally
produced expectations available for the hot path:
  * math_test.py:27-90    seven closest_segment_to_segment_points cases (values computed by the reference authors)
  * math_test.py:92-130   upper_tri_index / upper_trid_index enumerations
  * io_test.py:629-643    humanoid.xml at keyframe 0: MuJoCo C finds ncon = 8, nefc = 32
  * broadphase_test.py:55-188, 190-233  candidate-pair counts of four small scenes (any broadphase, any filter), the margin rule,
                          the parent filter
  * broadphase_test.py:235-337  counts per filter combination (plane / sphere / AABB / OBB) on a plane + two capsules
  * collision_driver_test.py:1336-1372  box-box contact distance and normal (MuJoCo C value -0.02 for the penetrating case; the
                          separated-within-gap case, 0.005, is produced by MuJoCo's convex box-box collider and is held
                          back for the convex path)
Each is checked against the float64 oracle AND the MJCF loader on CPU, and against the HIP path on the GPU box.
"""

import ctypes

import numpy as np
import pytest

import mujoco_warp_amd as mjw
from oracle import ref
from tests import conftest

# ---- math_test.py:27-90 -------------------------------------------------------------------------------------------------
SEGMENT_CASES = [
  # a0, a1, b0, b1, best_a, best_b, places
  ([0.73432405, 0.12372768, 0.20272314], [1.10600128, 0.88555209, 0.65209485], [0.85599262, 0.61736299, 0.9843583],
   [1.84270939, 0.92891793, 1.36343326], [1.09063, 0.85404, 0.63351], [0.99596, 0.66156, 1.03813], 5),
  ([0, 0, -1], [0, 0, 1], [-1, 0, 0], [1, 0, 0], [0, 0, 0], [0, 0, 0], 5),                     # intersecting segments
  ([0.2, 0.2, 0], [1, 1, 0], [0.2, 0.4, 0], [1, 2, 0], [0.3, 0.3, 0], [0.2, 0.4, 0], 2),      # intersecting lines get clipped
  ([0, 0, -1], [0, 0, 1], [1, 0, -1], [1, 0, 1], [0, 0, 0], [1, 0, 0], 5),                     # parallel: midpoints
  ([0, 0, -1], [0, 0, 1], [1, 0, 1], [1, 0, 3], [0, 0, 1], [1, 0, 1], 5),                      # parallel offset: end points
  ([0, 0, -1], [0, 0, -1], [1, 0, 0.1], [1, 0, 0.1], [0, 0, -1], [1, 0, 0.1], 5),              # zero length: no NaNs
  ([0, 0, -1], [0, 0, 1], [0, 0, -1], [0, 0, 1], [0, 0, 0], [0, 0, 0], 5),                     # overlapping: midpoints
]


def _seg(a0, a1, b0, b1):
  L = ref.lib()
  arrs = [np.array(x, dtype=np.float64) for x in (a0, a1, b0, b1)]
  oa, ob = np.zeros(3), np.zeros(3)
  p = lambda x: x.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
  L.ref_closest_segment_to_segment_points(*[p(x) for x in arrs], p(oa), p(ob))
  return oa, ob


@pytest.mark.parametrize("case", range(len(SEGMENT_CASES)))
def test_closest_segment_to_segment_points(case):
  a0, a1, b0, b1, ea, eb, places = SEGMENT_CASES[case]
  ga, gb = _seg(a0, a1, b0, b1)
  assert np.isfinite(ga).all() and np.isfinite(gb).all()
  np.testing.assert_allclose(ga, ea, atol=0.5 * 10.0 ** -places)
  np.testing.assert_allclose(gb, eb, atol=0.5 * 10.0 ** -places)


def test_upper_triangular_indexers():
  L = ref.lib()
  for n in (2, 10):  # math_test.py:92-106
    assert [L.ref_upper_tri_index(n, i, j) for i in range(n) for j in range(i + 1, n)] == list(range(n * (n - 1) // 2))
  for n in (1, 10):  # math_test.py:108-122
    assert [L.ref_upper_trid_index(n, i, j) for i in range(n) for j in range(i, n)] == list(range(n * (n + 1) // 2))
  assert L.ref_upper_trid_index(10, 1, 5) == L.ref_upper_trid_index(10, 5, 1)  # math_test.py:124-126
  # the engine's own pair table uses the same enumeration (io.geom_pairs: np.triu_indices order)
  g1, g2 = np.triu_indices(10, k=1)
  assert [L.ref_upper_tri_index(10, int(a), int(b)) for a, b in zip(g1, g2)] == list(range(45))


# ---- io_test.py:629-643: humanoid at keyframe 0 (MuJoCo C: ncon = 8, nefc = 32) ------------------------------------------
def test_humanoid_key0_counts_oracle():
  mjm = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  s = ref.RefSim(mjm, nconmax=24, njmax=64)
  s.reset(key=0)
  s.forward()
  assert (s.ncon, s.nefc) == (8, 32)
  assert (s.ne, s.nf, s.nl) == (0, 0, 0)


@pytest.mark.gpu
def test_humanoid_key0_counts_gpu():
  mjm = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  m = mjw.put_model(mjm)
  d = mjw.make_data(mjm, nworld=3, nconmax=24, njmax=64)
  mjw.reset_data_keyframe(m, d, 0)
  mjw.forward(m, d)
  assert (d.ws_ncon.numpy() == 8).all() and (d.nefc.numpy() == 32).all()
  assert int(d.nacon.numpy()[0]) == 8 * 3


# ---- broadphase_test.py:55-188 ---------------------------------------------------------------------------------------------
BROADPHASE_XML = """
<mujoco>
  <worldbody>
    <body><freejoint/><geom type="sphere" size="0.1"/></body>
    <body><freejoint/><geom type="sphere" size="0.1"/></body>
    <body><freejoint/><geom type="capsule" size="0.1 0.1"/></body>
    <body><freejoint/><geom type="sphere" size="0.1"/></body>
    <body>
      <freejoint/>
      <geom type="sphere" size="0.1"/>
      <geom type="sphere" size="0.1"/>
      <body><geom type="sphere" size="0.1"/><joint type="hinge"/></body>
    </body>
  </worldbody>
  <keyframe>
    <key qpos='0 0 0 1 0 0 0  1 0 0 1 0 0 0  2 0 0 1 0 0 0  3 0 0 1 0 0 0  4 0 0 1 0 0 0  0'/>
    <key qpos='0 0 0 1 0 0 0  .05 0 0 1 0 0 0  2 0 0 1 0 0 0  3 0 0 1 0 0 0  4 0 0 1 0 0 0  0'/>
    <key qpos='0 0 0 1 0 0 0  .01 0 0 1 0 0 0  .02 0 0 1 0 0 0  3 0 0 1 0 0 0  4 0 0 1 0 0 0  0'/>
    <key qpos='0 0 0 1 0 0 0  1 0 0 1 0 0 0  2 0 0 1 0 0 0  2 0 0 1 0 0 0  4 0 0 1 0 0 0  0'/>
  </keyframe>
</mujoco>
"""
BROADPHASE_COUNTS = {0: 0, 1: 1, 2: 3, 3: 1}  # keyframe -> ncollision (broadphase_test.py:124, 130, 140, 178)

FILTERS = {"plane_sphere": 1 | 2, "plane_aabb": 1 | 4, "plane_obb": 1 | 8, "plane_sphere_aabb": 1 | 2 | 4, "plane_sphere_obb": 1 | 2 | 8,
           "all": 1 | 2 | 4 | 8}


def _oracle_ncollision(mjm, key, **kw):
  s = ref.RefSim(mjm, nconmax=64, njmax=256, **kw)
  s.reset(key=key)
  s.stage("kinematics")
  s.stage("collision")
  return s.ncollision


def _gpu_ncollision(mjm, key, broadphase=None, bfilter=None, nworld=2):
  m = mjw.put_model(mjm)
  if broadphase is not None:
    m.opt.broadphase = int(broadphase)
  if bfilter is not None:
    m.opt.broadphase_filter = int(bfilter)
  d = mjw.make_data(mjm, nworld=nworld, nconmax=64, njmax=256)
  mjw.reset_data_keyframe(m, d, key)
  mjw.kinematics(m, d)
  mjw.collision(m, d)
  n = d.ws_ncollision.numpy()
  assert (n == n[0]).all()
  assert int(d.ncollision.numpy()[0]) == int(n.sum())
  return int(n[0])


@pytest.mark.parametrize("broadphase", [0, 1])
@pytest.mark.parametrize("bfilter", sorted(FILTERS))
@pytest.mark.parametrize("key", sorted(BROADPHASE_COUNTS))
def test_broadphase_scene_counts_oracle(key, bfilter, broadphase):
  mjm = mjw.mjcf.from_xml_string(BROADPHASE_XML)
  kw = dict(broadphase=broadphase, broadphase_filter=FILTERS[bfilter])
  assert _oracle_ncollision(mjm, key, **kw) == BROADPHASE_COUNTS[key]
  if key == 1:  # contype / conaffinity incompatibility: no candidates at all (broadphase_test.py:167-173)
    mjm.geom_contype[:3] = 0
    assert _oracle_ncollision(mjm, key, **kw) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("broadphase", [mjw.BroadphaseType.NXN, mjw.BroadphaseType.SAP_TILE, mjw.BroadphaseType.SAP_SEGMENTED])
@pytest.mark.parametrize("bfilter", sorted(FILTERS))
def test_broadphase_scene_counts_gpu(broadphase, bfilter):
  for key, expect in BROADPHASE_COUNTS.items():
    mjm = mjw.mjcf.from_xml_string(BROADPHASE_XML)
    assert _gpu_ncollision(mjm, key, broadphase, FILTERS[bfilter]) == expect, (key, broadphase, bfilter)
  mjm = mjw.mjcf.from_xml_string(BROADPHASE_XML)
  mjm.geom_contype[:3] = 0
  assert _gpu_ncollision(mjm, 1, broadphase, FILTERS[bfilter]) == 0


MARGIN_CASES = [(0, 0, 0), (0, 0.011, 1), (0.011, 0, 1), (0.00999, 0, 0), (0, 0.00999, 0), (0.00999, 0.00999, 1)]  # broadphase_test.py:181-188


def _margin_xml(m1, m2):
  return f"""
<mujoco>
  <worldbody>
    <body><geom type="sphere" size=".1" margin="{m1}"/><joint type="slide" axis="1 0 0"/></body>
    <body><geom type="sphere" size=".1" margin="{m2}"/><joint type="slide" axis="1 0 0"/></body>
  </worldbody>
  <keyframe><key qpos="0 .21"/></keyframe>
</mujoco>"""


@pytest.mark.parametrize("m1,m2,expect", MARGIN_CASES)
def test_broadphase_margin_oracle(m1, m2, expect):
  assert _oracle_ncollision(mjw.mjcf.from_xml_string(_margin_xml(m1, m2)), 0) == expect


@pytest.mark.gpu
@pytest.mark.parametrize("broadphase", [mjw.BroadphaseType.NXN, mjw.BroadphaseType.SAP_TILE])
@pytest.mark.parametrize("m1,m2,expect", MARGIN_CASES)
def test_broadphase_margin_gpu(m1, m2, expect, broadphase):
  assert _gpu_ncollision(mjw.mjcf.from_xml_string(_margin_xml(m1, m2)), 0, broadphase) == expect


FILTERPARENT_XML = """
<mujoco>
  <worldbody>
    <body><geom type="sphere" size=".1"/><joint type="slide"/>
      <body><geom type="sphere" size=".1"/><joint type="slide"/></body>
    </body>
  </worldbody>
  <keyframe><key qpos="0 0"/></keyframe>
</mujoco>
"""


@pytest.mark.parametrize("disable,expect", [(0, 0), (int(mjw.DisableBit.FILTERPARENT), 1)])  # broadphase_test.py:212-233
def test_broadphase_filterparent_oracle(disable, expect):
  mjm = mjw.mjcf.from_xml_string(FILTERPARENT_XML)
  mjm.opt.disableflags = disable
  assert _oracle_ncollision(mjm, 0) == expect


@pytest.mark.gpu
@pytest.mark.parametrize("disable,expect", [(0, 0), (int(mjw.DisableBit.FILTERPARENT), 1)])
def test_broadphase_filterparent_gpu(disable, expect):
  mjm = mjw.mjcf.from_xml_string(FILTERPARENT_XML)
  mjm.opt.disableflags = disable
  assert _gpu_ncollision(mjm, 0) == expect


# ---- broadphase_test.py:235-337: counts per filter combination --------------------------------------------------------------
PLANE_CAPSULES_XML = """
<mujoco>
  <option gravity="0 0 0"/>
  <worldbody>
    <geom name="floor" size="10 10 .001" type="plane"/>
    <body>
      <geom type="capsule" size=".05 .1"/>
      <joint type="slide" axis="1 0 0"/><joint type="slide" axis="0 0 1"/><joint type="hinge" axis="0 1 0"/>
    </body>
    <body>
      <geom type="capsule" size=".05 .1"/>
      <joint type="slide" axis="1 0 0"/><joint type="slide" axis="0 0 1"/><joint type="hinge" axis="0 1 0"/>
    </body>
  </worldbody>
  <keyframe>
    <key qpos="-.5 .25 0 .5 .25 0"/>
    <key qpos="-.5 .075 1.57 .5 .25 0"/>
    <key qpos="-.075 .25 0 .075 .25 0"/>
    <key qpos="0 .25 .7853 0 .45 .7853"/>
  </keyframe>
</mujoco>
"""
PLANE, SPHERE, AABB, OBB = 1, 2, 4, 8
FILTER_COUNTS = [  # (keyframe, filter, ncollision)
  (0, PLANE | SPHERE, 0), (0, PLANE | AABB, 0), (0, PLANE | OBB, 0),
  (1, PLANE | SPHERE, 1), (1, PLANE, 2), (1, PLANE | OBB, 1),
  (2, PLANE | SPHERE, 1), (2, PLANE | AABB, 0), (2, PLANE | OBB, 0),
  (3, PLANE | SPHERE, 1), (3, PLANE | AABB, 1), (3, PLANE | OBB, 0),
]


@pytest.mark.parametrize("broadphase", [0, 1])
@pytest.mark.parametrize("key,bfilter,expect", FILTER_COUNTS)
def test_broadphase_filter_counts_oracle(key, bfilter, expect, broadphase):
  mjm = mjw.mjcf.from_xml_string(PLANE_CAPSULES_XML)
  assert _oracle_ncollision(mjm, key, broadphase=broadphase, broadphase_filter=bfilter) == expect


@pytest.mark.gpu
@pytest.mark.parametrize("broadphase", [mjw.BroadphaseType.NXN, mjw.BroadphaseType.SAP_TILE])
@pytest.mark.parametrize("key,bfilter,expect", FILTER_COUNTS)
def test_broadphase_filter_counts_gpu(key, bfilter, expect, broadphase):
  mjm = mjw.mjcf.from_xml_string(PLANE_CAPSULES_XML)
  assert _gpu_ncollision(mjm, key, broadphase, bfilter) == expect


# ---- collision_driver_test.py:1336-1372: box-box with a gap; MuJoCo C's distances -------------------------------------------
def _boxes_xml(z2):
  return f"""
<mujoco>
  <worldbody>
    <body pos="0 0 0"><geom type="box" size="0.1 0.1 0.1" gap="0.01"/></body>
    <body pos="0 0 {z2}"><freejoint/><geom type="box" size="0.1 0.1 0.1" gap="0.01"/></body>
  </worldbody>
</mujoco>"""


@pytest.mark.parametrize("z2,expect", [(0.18, -0.02)])
def test_box_box_gap_distance_oracle(z2, expect):
  mjm = mjw.mjcf.from_xml_string(_boxes_xml(z2))
  s = ref.RefSim(mjm, nconmax=16, njmax=64)
  s.reset()
  s.forward()
  assert s.ncon > 0
  np.testing.assert_allclose(s.con_dist[: s.ncon], expect, atol=5e-5)
  n = s.con_frame[: s.ncon].reshape(s.ncon, 9)[:, :3]
  np.testing.assert_allclose(np.abs(n @ np.array([0.0, 0.0, 1.0])), 1.0, atol=1e-4)  # frame normal along z (MuJoCo C: dot = 1)


@pytest.mark.gpu
@pytest.mark.parametrize("z2,expect", [(0.18, -0.02)])
def test_box_box_gap_distance_gpu(z2, expect):
  mjm = mjw.mjcf.from_xml_string(_boxes_xml(z2))
  m = mjw.put_model(mjm)
  d = mjw.make_data(mjm, nworld=2, nconmax=16, njmax=64)
  mjw.forward(m, d)
  n = int(d.nacon.numpy()[0])
  assert n > 0
  np.testing.assert_allclose(d.contact.dist.numpy()[:n], expect, atol=5e-5)
  normal = d.contact.frame.numpy()[:n, 0]
  np.testing.assert_allclose(np.abs(normal @ np.array([0.0, 0.0, 1.0])), 1.0, atol=1e-4)


# ---- SAP against NXN on a crowded scene (no reference numbers: the two broadphases must agree with each other and the oracle) --
def _crowd_xml(n=120, seed=3):
  rng = np.random.default_rng(seed)
  bodies = []
  for i in range(n):
    p = rng.uniform(-1.0, 1.0, 3) * np.array([1.2, 1.2, 0.4]) + np.array([0, 0, 0.6])
    if i % 3 == 0:
      g = f'<geom type="capsule" size="{rng.uniform(.04, .08):.3f} {rng.uniform(.05, .15):.3f}"/>'
    elif i % 3 == 1:
      g = f'<geom type="sphere" size="{rng.uniform(.05, .1):.3f}"/>'
    else:
      g = f'<geom type="box" size="{rng.uniform(.04, .09):.3f} {rng.uniform(.04, .09):.3f} {rng.uniform(.04, .09):.3f}"/>'
    q = rng.standard_normal(4)
    q /= np.linalg.norm(q)
    bodies.append(f'<body pos="{p[0]:.3f} {p[1]:.3f} {p[2]:.3f}" quat="{q[0]:.4f} {q[1]:.4f} {q[2]:.4f} {q[3]:.4f}"><freejoint/>{g}</body>')
  return '<mujoco><worldbody><geom name="floor" type="plane" size="0 0 .05"/>' + "".join(bodies) + "</worldbody></mujoco>"


@pytest.mark.parametrize("bfilter", [3, 15])
def test_sap_equals_nxn_oracle_crowd(bfilter):
  mjm = mjw.mjcf.from_xml_string(_crowd_xml(60))
  out = []
  for bp in (0, 1):
    s = ref.RefSim(mjm, nconmax=400, njmax=1600, broadphase=bp, broadphase_filter=bfilter)
    s.reset()
    s.stage("kinematics")
    s.stage("collision")
    out.append((s.ncollision, s.ncon, s.con_geom[: s.ncon].copy(), s.con_dist[: s.ncon].copy()))
  assert out[0][0] == out[1][0] > 3 and out[0][1] == out[1][1] > 1
  assert (out[0][2] == out[1][2]).all() and (out[0][3] == out[1][3]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("bfilter", [3, 15])
def test_sap_equals_nxn_gpu_crowd(bfilter):
  mjm = mjw.mjcf.from_xml_string(_crowd_xml(120))  # 121 geoms, 7260 pairs: the candidate list is capped, the pair bits are not
  s = ref.RefSim(mjm, nconmax=800, njmax=3200, broadphase=0, broadphase_filter=bfilter)
  s.reset()
  s.stage("kinematics")
  s.stage("collision")
  res = []
  for bp in (mjw.BroadphaseType.NXN, mjw.BroadphaseType.SAP_TILE):
    m = mjw.put_model(mjm)
    m.opt.broadphase, m.opt.broadphase_filter = int(bp), bfilter
    d = mjw.make_data(mjm, nworld=2, nconmax=400, njmax=1600)
    mjw.kinematics(m, d)
    mjw.collision(m, d)
    n = int(d.ws_ncon.numpy()[0])
    res.append((int(d.ws_ncollision.numpy()[0]), n, d.contact.geom.numpy()[:n].copy(), d.contact.dist.numpy()[:n].copy()))
    assert (d.overflow.numpy() == 0).all()
  assert res[0][0] == res[1][0] == s.ncollision and res[0][1] == res[1][1] == s.ncon
  assert (res[0][2] == res[1][2]).all() and (res[0][3] == res[1][3]).all()  # bitwise: same pairs in the same order
  assert (res[0][2] == s.con_geom[: s.ncon]).all()
  np.testing.assert_allclose(res[0][3], s.con_dist[: s.ncon], atol=2e-6)
