"""The reference-held GJK / EPA / multi-contact vectors of collision_gjk_test.py through the HIP narrowphase (and the oracle's
model-driven collision), not only through the oracle's bare `ccd` call (tests/test_convex.py).

Each case is the reference test's two geoms as two free bodies of a two-geom model posed through `qpos` (position + the quaternion
of the test's rotation matrix), then `kinematics` + `collision` through the C ABI.  Expected values are the numbers the reference's
tests assert (file:line per case); where the reference quotes both its float32 result and the "actual" / MuJoCo-C 64-bit depth the
latter is used.  Spheres are absent: sphere pairs never reach `ccd` in `collision()` (primitive collider, as in the reference).

Tolerances: the reference runs these in float32 and asserts 6-7 places; here the pose additionally goes through float32 qpos ->
quaternion -> rotation matrix on the device, i.e. eps * |pos| + eps * size of rounding on distances (<= 2.5e-6 at |pos| = 21 in
test_box_box_max, 1e-7 elsewhere).
"""

import numpy as np
import pytest

from oracle import ref

I3 = np.eye(3)
_NAMES = {3: "capsule", 4: "ellipsoid", 5: "cylinder", 6: "box"}
CAPSULE, CYLINDER, BOX = 3, 5, 6


def _euler_xyz(deg):
  a, b, c = np.radians(deg)
  rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
  ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
  rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
  return rx @ ry @ rz


def _quat_mat(q):
  w, x, y, z = np.asarray(q, dtype=float) / np.linalg.norm(q)
  return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                   [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _mat_quat(R):
  """Rotation matrix (possibly with 1e-8 noise, as printed by the reference's tests) -> unit quaternion (w, x, y, z)."""
  R = np.asarray(R, dtype=np.float64).reshape(3, 3)
  K = np.array([[R[0, 0] - R[1, 1] - R[2, 2], 0, 0, 0], [R[0, 1] + R[1, 0], R[1, 1] - R[0, 0] - R[2, 2], 0, 0],
                [R[0, 2] + R[2, 0], R[1, 2] + R[2, 1], R[2, 2] - R[0, 0] - R[1, 1], 0],
                [R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1], R[0, 0] + R[1, 1] + R[2, 2]]]) / 3.0
  w, v = np.linalg.eigh(K)  # (lower triangle used)
  q = v[[3, 0, 1, 2], np.argmax(w)]
  return q if q[0] >= 0 else -q


_S25 = [0.025] * 3
# name, reference line, (type, pos, rot, size) x 2, expected dist (None: not asserted), |dist| tolerance, expected ncon (None: >= 1), ccd_iterations
CASES = [
  ("box_box_contact", 384, (BOX, [-1, 0, 0], I3, [2.5, 2.5, 2.5]), (BOX, [1.5, 0, 0], I3, [1, 1, 1]), -1.0, 5e-7, None, 35),
  ("cylinder_cylinder_contact", 467, (CYLINDER, [0, 0, 0], I3, [1, 0.5, 0]), (CYLINDER, [1.999, 0, 0], I3, [1, 0.5, 0]), -0.001, 5e-7, 1, 35),
  ("cylinder_capsule", 698, (CYLINDER, [0, 0, 0], I3, [2, 4, 0]), (CAPSULE, [0, 0, 5], I3, [1, 1, 0]), -1.0, 1e-6, 1, 35),
  ("cylinder_box", 668, (BOX, [0, 0, 0], I3, [1, 1, 0.1]),
   (CYLINDER, [0.00015228791744448245, -0.00074981129728257656, 0.29839199781417846680],
    [0.99996972084045410156, 0.00776371126994490623, -0.00043433305108919740, -0.00776385562494397163, 0.99996984004974365234,
     -0.00033095158869400620, 0.00043175052269361913, 0.00033431366318836808, 0.99999988079071044922], [0.1, 0.2, 0.3]), -0.0016624178339902445, 5e-7, 1, 50),
  ("box_box_shallow_penetration", 483, (BOX, [0, 0, 0.19972974], I3, [0.2] * 3), (BOX, [0, 0, 0.49947918], I3, [0.1] * 3), -0.00025054812, 1e-7, 4, 35),
  ("box_edge", 499, (BOX, [0, 0, 2], I3, [1, 1, 1]), (BOX, [0, 0, 4.4], _euler_xyz([0, 90, 40]), [1, 1, 1]), None, 0, 2, 35),
  ("box_box_ccd", 513, (BOX, [0, 0, 1.9], I3, [1, 1, 1]), (BOX, [0, 0, 0], I3, [10, 10, 1]), None, 0, 4, 35),
  ("box_box_ccd2", 548, (BOX, [0, 0, 2], I3, [1, 1, 1]), (BOX, [0, 1, 3.99], _euler_xyz([0, 0, 40]), [1, 1, 1]), None, 0, 4, 35),
  ("box_box_early", 564,
   (BOX, [0.07524700462818145752, -0.13524700701236724854, 0.12491077929735183716],
    [1.0, 0.00000000006837434091, 0.00000000080494955146, -0.00000000006837435479, 1.0, 0.00000002552030764491, -0.00000000080494955146, -0.00000002552030764491, 1.0], _S25),
   (BOX, [0.07524700462818145752, -0.13524700701236724854, 0.17094630002975463867],
    [1.0, 0.00000000006837435479, -0.00000000018000903546, -0.00000000006837434091, 1.0, 0.00000004174798817758, 0.00000000018000903546, -0.00000004174798817758, 1.0], _S25),
   -0.0039644796979132323, 5e-7, None, 35),
  ("box_box_early2", 606,
   (BOX, [0.07122065126895904541, -0.19126638770103454590, 0.29129269719123840332],
    [0.99999558925628662109, 0.00258362153545022011, 0.00148368685040622950, -0.00258197076618671417, 0.99999606609344482422, -0.00111339206341654062,
     -0.00148655765224248171, 0.00110955617856234312, 0.99999833106994628906], _S25),
   (BOX, [0.07183132320642471313, -0.13260576128959655762, 0.30987158417701721191],
    [0.99827724695205688477, 0.02493947930634021759, 0.05311207473278045654, 0.00605074502527713776, 0.85659545660018920898, -0.51595354080200195312,
     -0.05836316198110580444, 0.51538598537445068359, 0.85496860742568969727], _S25),
   -2.515764037690309e-06, 1.5e-7, None, 35),
  ("box_box_float", 713,
   (BOX, [-0.17624500393867492676, -0.12375499308109283447, 0.12499777972698211670],
    [1.0, -0.00000000184385418045, -0.00000025833372774287, 0.00000000184391857339, 1.0, 0.00000024928382913458, 0.00000025833372774287, -0.00000024928382913458, 1.0], _S25),
   (BOX, [-0.17624500393867492676, -0.12375499308109283447, 0.17499557137489318848],
    [1.0, -0.00000000184292525685, 0.00000012980596864054, 0.00000000184294413064, 1.0, -0.00000014602545661546, -0.00000012980596864054, 0.00000014602545661546, 1.0], _S25),
   None, 0, -1, 35),  # "real depth is ~ 2E-6"; the reference asserts one `ccd` result with dist < 1e-4, which includes dist = 0 (origin on the
  # polytope boundary: what the float64 oracle returns) -- `collision()` keeps a contact only if dist < margin = 0, so 0 or 1 contacts, never deeper than 1e-4
  ("box_box_horizon", 756,
   (BOX, [0.065118454396725, -0.125125020742416, 0.124963559210300],
    [0.996357858181000, 0.085266821086407, -0.000942531623878, -0.085266284644604, 0.996358215808868, 0.000591202871874, 0.000989508931525, -0.000508683384396, 0.999999582767487], _S25),
   (BOX, [0.065104484558105, -0.124979749321938, 0.174992129206657],
    [0.996556758880615, -0.082913912832737, -0.000453041866422, 0.082915119826794, 0.996536433696747, 0.006357696373016, -0.000075668765930, -0.006373368669301, 0.999979794025421], _S25),
   -0.00011579410621457821, 5e-7, None, 35),
  ("box_box_rotation", 816,
   (BOX, [0.015344001352787, -0.195344015955925, 0.174637570977211],
    [1.0, 0.000000000029901, 0.000004057303613, -0.000000000062404, 1.0, 0.000008010840247, -0.000004057303613, -0.000008010840247, 1.0], _S25),
   (BOX, [0.015344001352787, -0.195344015955925, 0.224056228995323],
    [1.0, 0.000000000029692, -0.000003355821491, -0.000000000057016, 1.0, -0.000008142159459, 0.000003355821491, 0.000008142159459, 1.0], _S25),
   None, 0, 4, 35),
  ("box_box_diagonal", 866, (BOX, [0, 0, 0], I3, [0.5, 0.5, 0.1]),
   (BOX, [0.135535001754761, -0.195535004138947, 0.124984227120876],
    [1.0, 0.000000000048563, -0.000000135524601, -0.000000000048577, 1.0, -0.000000103374248, 0.000000135524601, 0.000000103374248, 1.0], _S25),
   -1.5778851595232846e-05, 1e-7, 4, 35),
  ("box_box_max", 900,
   (BOX, [6.0405082703, 21.4734001160, 0.036854844], [0.8378595710, 0.3184406757, -0.4433811009, 0.5328434706, -0.3006005287, 0.7910227776, 0.1186132580, -0.8990187645, -0.4215400815],
    [0.018, 0.018, 0.01]),
   (BOX, [6.0641078949, 21.4842395782, 0.0212156791], [-0.6420212388, -0.0727036372, -0.7632319927, 0.3801730871, -0.8946756721, -0.2345722020, -0.6657907367, -0.4407605529, 0.6020406485],
    [0.020, 0.020, 0.04]),
   -0.03636224, 4e-6, None, 35),
  ("box_box_max2", 952, (BOX, [0, 0, 0], I3, [0.5, 0.5, 0.1]),
   (BOX, [0.0885666460, 0.0911745951, 0.1250119805], [0.9999999404, 0.0004342802, -0.0001755831, -0.0004346797, 0.9999973178, -0.0022819033, 0.0001745916, 0.0022819792, 0.9999974370], _S25),
   -4.936969499999555e-05, 1.5e-7, None, 35),
  ("box_edge_flipped", 986, (BOX, [1.10164554, -0.11389316, 0.74], _quat_mat([-0.348312918, 0, 0, 0.937378318]), [0.65, 0.48, 0.04]), (BOX, [1.4, 0, 1.425], I3, [0.1, 1.2, 1.4]),
   -0.607368, 1e-4, 2, 35),  # witness points (1.907368, -0.052973, 0.7) / (1.3, -0.052973, 0.7): dist = -(1.907368 - 1.3)
]


# what the reference's own float32 run returns where its test quotes it next to the 64-bit depth (:646, :796, :984)
REFERENCE_F32 = {"box_box_early2": -2.515156e-06, "box_box_horizon": -0.00011578822, "box_box_max2": -4.9374998e-05}


def _size_attr(t, size):
  n = {CAPSULE: 2, CYLINDER: 2, BOX: 3}[t]
  return " ".join(repr(float(x)) for x in size[:n])


def _model(case):
  import mujoco_warp_amd as mjw

  _, _, g1, g2, _, _, _, iters = case
  xml = f"""
  <mujoco>
    <option ccd_iterations="{iters}"/>
    <worldbody>
      <geom type="cylinder" size=".01 .01" pos="100 100 100"/>
      <body><freejoint/><geom type="{_NAMES[g1[0]]}" size="{_size_attr(g1[0], g1[3])}"/></body>
      <body><freejoint/><geom type="{_NAMES[g2[0]]}" size="{_size_attr(g2[0], g2[3])}"/></body>
    </worldbody>
  </mujoco>"""
  # (the far-away cylinder keeps the EPA cap at `ccd_iterations`: a model whose convex pairs are all box-box runs EPA with 16
  # iterations, collision_convex.py:1223, while the reference's test harness passes `ccd_iterations` for both GJK and EPA --
  # test_box_box_max needs more than 16: the model-driven depth at 16 is -0.036140 instead of -0.036362)
  mjm = mjw.mjcf.from_xml_string(xml)
  q = np.concatenate([np.asarray(g1[1], float), _mat_quat(g1[2]), np.asarray(g2[1], float), _mat_quat(g2[2])])
  return mjm, q


def _check(name, line, dist, ncon, exp, tol, exp_n):
  if exp_n == -1:
    assert ncon <= 1 and (ncon == 0 or -1e-4 < dist <= 0.0), (name, line, ncon, dist)
    return
  if exp_n is not None:
    assert ncon == exp_n, (name, line, ncon, exp_n)
  else:
    assert ncon >= 1, (name, line, ncon)
  if exp is not None:
    assert abs(dist - exp) <= tol, (name, line, dist, exp)


def test_mat_quat_round_trip():
  rng = np.random.default_rng(0)
  for _ in range(20):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    if q[0] < 0:
      q = -q
    np.testing.assert_allclose(_mat_quat(_quat_mat(q)), q, atol=1e-12)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_oracle_collision_on_reference_poses(case):
  """The oracle's model-driven `collision` (float64) on the same two-body models: the path the GPU is compared with elsewhere."""
  name, line, _, _, exp, tol, exp_n, _ = case
  mjm, q = _model(case)
  s = ref.RefSim(mjm, nconmax=16, njmax=64)
  s.qpos[:] = q
  s.stage("kinematics")
  s.stage("collision")
  _check(name, line, float(s.con_dist[: s.ncon].min()) if s.ncon else np.inf, s.ncon, exp, min(tol, 5e-7) if tol else 0, exp_n)


@pytest.mark.gpu
def test_gpu_narrowphase_on_reference_poses():
  """Every case as its own model (shapes differ), one world each: HIP `kinematics` + `collision`, deepest contact distance and
  contact count against the reference-held values; contact normal / position against the oracle on the same pose."""
  import mujoco_warp_amd as mjw

  report = []
  for case in CASES:
    name, line, g1, g2, exp, tol, exp_n, _ = case
    mjm, q = _model(case)
    m = mjw.put_model(mjm)
    d = mjw.put_data(mjm, mjw.MjData(mjm), nworld=1, nconmax=16, njmax=64)
    d.qpos.assign(q.astype(np.float32)[None])
    mjw.kinematics(m, d)
    mjw.collision(m, d)
    ncon, adr = int(d.ws_ncon.numpy()[0]), int(d.ws_conadr.numpy()[0])
    dist = d.contact.dist.numpy()[adr : adr + ncon]
    s = ref.RefSim(mjm, nconmax=16, njmax=64)
    s.qpos[:] = q
    s.stage("kinematics")
    s.stage("collision")
    report.append((name, line, float(dist.min()) if ncon else np.inf, exp, ncon, exp_n, s.ncon))
    _check(name, line, float(dist.min()) if ncon else np.inf, ncon, exp, tol, exp_n)
    if name in REFERENCE_F32:  # measured: identical to every printed digit (-2.51515598e-06, -0.000115788229, -4.93749976e-05)
      assert abs(float(dist.min()) - REFERENCE_F32[name]) <= 3e-9, (name, float(dist.min()), REFERENCE_F32[name])
    if exp_n == -1:
      continue
    assert ncon == s.ncon, (name, ncon, s.ncon)
    # the contact normal (frame row 0) agrees with the float64 oracle; flat face-face contacts have a unique normal
    # (the normal is the difference of two float32 witness points |dist| apart: its error grows as eps * |pos| / |dist| -- 3e-3 on
    # the 2.5e-6 deep test_box_box_early2)
    np.testing.assert_allclose(d.contact.frame.numpy()[adr].reshape(9)[:3], s.con_frame[0][:3], atol=max(2e-3, 4e-8 / abs(float(dist.min()))), err_msg=name)
    assert (d.overflow.numpy() == 0).all(), name
  for r in report:
    print("%-28s :%-4d dist %.9g expected %s ncon %d expected %s oracle %d" % r)
