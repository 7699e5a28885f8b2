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
