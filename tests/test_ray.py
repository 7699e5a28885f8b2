"""Rays against the primitive geoms (reference ray.py, ray_test.py) and the rangefinder sensor: the oracle against closed forms and
surface equations (the reference's own tests compare with MuJoCo at run time and hold no stored numbers: parity with it is unpinned),
the HIP path (mjh_rays, k_sensor) against the oracle."""

import numpy as np
import pytest

import mujoco_warp_amd as mjw
from mujoco_warp_amd.device import DeviceArray
from oracle import ref

# primitive geoms laid out like the reference's test scene (test_data/ray.xml; its mesh and height-field geoms need files this tree does not have)
SCENE = """
<mujoco>
  <asset><material name="glass" rgba="1 1 1 0"/></asset>
  <worldbody>
    <geom name="plane" size="4 4 4" type="plane" rgba="0.1 0.1 0.1 1"/>
    <geom name="sphere" pos="0 0 1" size="0.5" type="sphere"/>
    <geom name="capsule" pos="0 1 1" quat="0 0.3826834 0 0.9238795" size="0.25 0.5" type="capsule"/>
    <geom name="box" pos="1 0 1" euler="0 0 90" size="0.5 0.25 0.3" type="box"/>
    <geom name="cylinder" pos="2 0 1" euler="0 0 30" type="cylinder" size=".25 .5" group="2"/>
    <geom name="ellipsoid" pos="-1.5 0 1" euler="20 30 40" type="ellipsoid" size=".5 .3 .2" group="1"/>
    <geom name="ghost" pos="0 0 2.5" size="0.3" type="sphere" rgba="1 1 1 0"/>
    <geom name="pane" pos="0 0 3.2" size="0.3 .3 .01" type="box" material="glass"/>
    <body name="mover" pos="-1 -1.5 1"><freejoint/><geom name="ball" size="0.2"/>
      <site name="eye" pos="0 0 -.25" euler="180 0 0"/><site name="side" pos="0 0 0" euler="0 90 0"/><site name="up" pos="0 0 .1"/>
    </body>
  </worldbody>
  <sensor><rangefinder site="eye"/><rangefinder site="side"/><rangefinder site="up"/><rangefinder site="eye" cutoff="0.5"/></sensor>
</mujoco>
"""


def _sim():
  mjm = mjw.mjcf.from_xml_string(SCENE)
  s = ref.RefSim(mjm, nconmax=8, njmax=32)
  s.forward()
  return mjm, s


def _n(v):
  v = np.asarray(v, dtype=np.float64)
  return v / np.linalg.norm(v)


def test_oracle_closed_forms():
  mjm, s = _sim()
  gid = {n: i for i, n in enumerate(mjm.geom_names)}
  assert s.ray([12.146, 1.865, 3.895], [0, 0, -1])[:2] == (-1.0, -1)  # nothing below (outside the plane's rectangle)
  v = _n([0.1, 0.2, -1.0])
  dist, g, nrm = s.ray([2.0, 1.0, 3.0], v)  # the plane: z = 0
  assert g == gid["plane"] and dist == pytest.approx(3.0 / -v[2], abs=1e-12) and np.allclose(nrm, [0, 0, 1])
  assert s.ray([0, 0, -0.5], v)[1] == -1  # wrong side of the plane
  dist, g, nrm = s.ray([0, 0, 1.6], v)  # the sphere: |p + t v - c| = r
  p = np.array([0, 0, 1.6]) + dist * v
  assert g == gid["sphere"] and np.linalg.norm(p - [0, 0, 1]) == pytest.approx(0.5, abs=1e-12) and np.allclose(nrm, (p - [0, 0, 1]) / 0.5)
  dist, g, nrm = s.ray([0, 0, 5], [0, 0, -1])  # straight down through the invisible sphere and the glass pane to the sphere's top
  assert g == gid["sphere"] and dist == pytest.approx(3.5, abs=1e-12)
  dist, g, nrm = s.ray([2, 0, 3], [0, 0, -1])  # the cylinder's top cap (its axis stays vertical under the rotation about z)
  assert g == gid["cylinder"] and dist == pytest.approx(1.5, abs=1e-12) and np.allclose(nrm, [0, 0, 1])
  dist, g, nrm = s.ray([2, -3, 1], [0, 1, 0])  # its round side
  assert g == gid["cylinder"] and dist == pytest.approx(2.75, abs=1e-12) and np.allclose(nrm, [0, -1, 0])
  dist, g, nrm = s.ray([1, -3, 1], [0, 2, 0])  # the box's long side (turned by 90 degrees: half-size 0.5 along y), in units of |vec|
  assert g == gid["box"] and dist == pytest.approx(2.5 / 2, abs=1e-12) and np.allclose(nrm, [0, -1, 0], atol=1e-7)
  dist, g, nrm = s.ray([0, 1, 3], [0, 0, -1])  # the capsule (axis in the x-z plane at 45 degrees): the round side right above its centre
  assert g == gid["capsule"] and dist == pytest.approx(2.0 - 0.25 * np.sqrt(2.0), abs=1e-6)
  # filters: groups (an entry of 0 hides the group), static geoms, the excluded body
  assert s.ray([2, 0, 3], [0, 0, -1], geomgroup=[1, 1, 0, 1, 1, 1])[1] == gid["plane"]
  assert s.ray([2, 0, 3], [0, 0, -1], geomgroup=[0, 0, 1, 0, 0, 0])[1] == gid["cylinder"]
  assert s.ray([-1, -1.5, 3], [0, 0, -1])[1] == gid["ball"]
  assert s.ray([-1, -1.5, 3], [0, 0, -1], bodyexclude=1)[1] == gid["plane"]
  assert s.ray([-1, -1.5, 3], [0, 0, -1], flg_static=False)[1] == gid["ball"] and s.ray([2, 0, 3], [0, 0, -1], flg_static=False)[1] == -1


def _implicit(mjm, s, g, p):
  """Surface function of geom g at world point p (0 on the surface) and its outward gradient direction."""
  t, size = int(mjm.geom_type[g]), np.asarray(mjm.geom_size[g], dtype=np.float64)
  R = s.geom_xmat[g].reshape(3, 3)
  l = R.T @ (p - s.geom_xpos[g])
  if t == 0:
    return l[2], R[:, 2]
  if t == 2:
    return np.linalg.norm(l) - size[0], R @ _n(l)
  if t == 4:
    return np.sum((l / size) ** 2) - 1.0, R @ _n(l / size**2)
  if t == 3:
    c = np.array([0, 0, np.clip(l[2], -size[1], size[1])])
    return np.linalg.norm(l - c) - size[0], R @ _n(l - c)
  if t == 5:
    dr, dz = np.hypot(l[0], l[1]) - size[0], abs(l[2]) - size[1]
    return max(dr, dz), R @ (_n([l[0], l[1], 0]) if dr > dz else np.array([0, 0, np.sign(l[2])]))
  d = np.abs(l) - size
  k = int(np.argmax(d))
  return d[k], R[:, k] * np.sign(l[k])


def _random_rays(n, seed):
  rng = np.random.default_rng(seed)
  pnt = rng.uniform([-3, -3, 0.2], [3, 3, 4], size=(n, 3))
  target = rng.uniform([-2, -2, 0], [2.5, 1.5, 1.5], size=(n, 3))
  return pnt, target - pnt


def test_oracle_hits_lie_on_the_surface():
  mjm, s = _sim()
  pnt, vec = _random_rays(600, 1)
  seen = set()
  for p, v in zip(pnt, vec):
    dist, g, nrm = s.ray(p, v)
    if g < 0:
      continue
    seen.add(int(mjm.geom_type[g]))
    f, grad = _implicit(mjm, s, g, p + dist * v)
    assert abs(f) < 1e-9 and np.allclose(nrm, grad, atol=1e-6), (g, f, nrm, grad)
    for other in range(mjm.ngeom - 1):  # no visible geom is entered before the reported hit (the ball body's geom aside: same test)
      if other != g and mjm.geom_names[other] not in ("ghost", "pane"):
        assert all(_implicit(mjm, s, other, p + t * v)[0] > -1e-9 for t in np.linspace(0, dist, 40)[:-1]) or _implicit(mjm, s, other, p)[0] < 0
  assert seen == {0, 2, 3, 4, 5, 6}


def test_oracle_rangefinder():
  mjm, s = _sim()
  # eye: 0.25 below the ball's centre, looking down (its own body is skipped) at the floor; side: along +x from the ball's centre to the
  # ellipsoid... which it misses (y = -1.5): nothing; up: nothing above; the fourth reads the first through a cutoff
  assert s.sensordata[0] == pytest.approx(0.75, abs=1e-9) and s.sensordata[1] == -1.0 and s.sensordata[2] == -1.0 and s.sensordata[3] == pytest.approx(0.5)


@pytest.mark.gpu
def test_gpu_rays_vs_oracle():
  mjm, s = _sim()
  m = mjw.put_model(mjm)
  d = mjw.make_data(mjm, nworld=3)
  q = d.qpos.numpy()
  q[1, :3] += [0.4, 0.2, 0.3]
  q[2, :3] += [1.0, 1.5, 1.0]  # (above the sphere)
  d.qpos.assign(q)
  mjw.forward(m, d)
  sims = []
  for w in range(3):
    sw = ref.RefSim(mjm, nconmax=8, njmax=32)
    sw.qpos[:] = q[w]
    sw.forward()
    sims.append(sw)
  pnt, vec = _random_rays(500, 2)
  total = 0
  for kw in (dict(), dict(geomgroup=[1, 0, 0, 1, 1, 1]), dict(flg_static=False), dict(bodyexclude=1)):
    P, V = DeviceArray.from_numpy(pnt[None].astype(np.float32)), DeviceArray.from_numpy(vec[None].astype(np.float32))
    dist, gid, nrm = DeviceArray.zeros((3, 500)), DeviceArray.zeros((3, 500), np.int32), DeviceArray.zeros((3, 500, 3))
    ex = DeviceArray.full((500,), kw.get("bodyexclude", -1), np.int32)
    mjw.rays(m, d, P, V, kw.get("geomgroup"), kw.get("flg_static", True), ex, dist, gid, nrm)
    dist, gid, nrm = dist.numpy(), gid.numpy(), nrm.numpy()
    hits = flips = 0
    for w in range(3):
      for r in range(500):
        rd, rg, rn = sims[w].ray(pnt[r].astype(np.float32), vec[r].astype(np.float32), **kw)
        if rg != gid[w, r]:  # a grazing ray may hit in one precision and miss in the other
          flips += 1
          continue
        if rg >= 0:
          hits += 1
          assert abs(dist[w, r] - rd) < 2e-5 * max(1.0, abs(rd)) and np.abs(nrm[w, r] - rn).max() < 2e-3, (w, r, dist[w, r], rd, nrm[w, r], rn)
        else:
          assert dist[w, r] == -1.0 and (nrm[w, r] == 0).all()
    assert flips <= 3 and hits > 10, (kw, flips, hits)
    total += hits
  assert total > 2000, total
  # ray(): one ray per world, broadcast from a single origin
  d1, g1, n1 = mjw.ray(m, d, DeviceArray.from_numpy(np.array([[[0, 0, 5]]], dtype=np.float32)), DeviceArray.from_numpy(np.array([[[0, 0, -1]]], dtype=np.float32)))
  assert np.allclose(d1.numpy()[:2, 0], 3.5, atol=1e-6) and (g1.numpy()[:2, 0] == 1).all() and np.allclose(n1.numpy()[:2, 0], [0, 0, 1], atol=1e-6)
  assert g1.numpy()[2, 0] == mjm.geom_names.index("ball")  # (world 2's ball sits above the sphere; the ghost and the pane are invisible)
  with pytest.raises(ValueError):
    mjw.ray(m, d, P, V)


@pytest.mark.gpu
def test_gpu_rangefinder_vs_oracle():
  mjm, s = _sim()
  m = mjw.put_model(mjm)
  d = mjw.make_data(mjm, nworld=2)
  q = d.qpos.numpy()
  q[1, :7] = [-1.5, -1.2, 2.0, *_n([0.9, 0.3, -0.2, 0.1])]  # above the ellipsoid, tilted
  d.qpos.assign(q)
  for step in range(30):
    sims = []
    for w in range(2):
      sw = ref.RefSim(mjm, nconmax=8, njmax=32)
      sw.qpos[:] = d.qpos.numpy()[w]
      sw.qvel[:] = d.qvel.numpy()[w]
      sw.forward()
      sims.append(sw)
    mjw.forward(m, d)
    for w in range(2):
      assert np.abs(d.sensordata.numpy()[w] - sims[w].sensordata).max() < 1e-4, (step, w, d.sensordata.numpy()[w], sims[w].sensordata)
    mjw.step(m, d)
  assert d.sensordata.numpy()[0, 0] < 0.75  # (the ball fell towards the floor)


def test_rays_argument_checks():
  # (host-side validation only: nothing is launched)
  mjm = mjw.mjcf.from_xml_string(SCENE)
  m = mjw.put_model(mjm)
  d = mjw.make_data(mjm, nworld=2)
  z = lambda *s, dt=np.float32: DeviceArray.zeros(s, dt)
  with pytest.raises(ValueError):
    mjw.rays(m, d, z(1, 4, 3), z(1, 5, 3), None, True, None, z(2, 4), None, None)  # pnt / vec shapes differ
  with pytest.raises(ValueError):
    mjw.rays(m, d, z(3, 4, 3), z(3, 4, 3), None, True, None, z(2, 4), None, None)  # neither 1 nor nworld origins
  with pytest.raises(ValueError):
    mjw.rays(m, d, z(1, 4, 3), z(1, 4, 3), None, True, None, z(2, 5), None, None)  # dist of the wrong shape
  with pytest.raises(ValueError):
    mjw.rays(m, d, z(1, 4, 3), z(1, 4, 3), [1, 1, 1], True, None, z(2, 4), None, None)  # five group entries missing
  with pytest.raises(ValueError):
    mjw.rays(m, d, z(1, 4, 3), z(1, 4, 3), None, True, z(3, dt=np.int32), z(2, 4), None, None)  # bodyexclude per ray
  with pytest.raises(NotImplementedError):
    mjw.rays(m, d, z(1, 4, 3), z(1, 4, 3), None, True, None, z(2, 4), None, None, rc=object())
  with pytest.raises(ValueError):
    mjw.ray(m, d, z(1, 4, 3), z(1, 4, 3))  # several rays: rays()
