"""Convex mesh geoms (first part): mesh support function in GJK / EPA (reference collision_gjk.py:154-169,
exhaustive vertex search), plane-mesh collider (collision_primitive.py:52-274 plane_convex, exhaustive branch), mesh assets in Model
(types.py:1707-1709: mesh_vertadr / mesh_vertnum / mesh_vert, geom_dataid).

Reference-held numbers reproduced by the float64 oracle (collision_gjk_test.py): box-mesh distance 0.1 (:341-366), mesh-mesh
penetration -0.01 (:405-439), the degenerate mesh pair -0.0031312597856874586 with one contact (:441-467), sphere-mesh with margin
-0.001 (:648-665); collision_driver_test.py:691-711 (plane and a separated tetrahedron: no contact closer than 0.05).  The loader's
mesh compilation (hull, centre of mass, principal frame) is pinned by a cube given as a mesh, which must equal the box primitive.
Not built: multi-contact recovery on mesh faces (models must set multiccd="disable" for box-mesh / mesh-mesh pairs), hill-climbing
support on meshes that carry a graph, mesh files.
"""

import numpy as np
import pytest

import mujoco_warp_amd as mjw
from mujoco_warp_amd import _npmath as nm
from oracle import ref
from tests.conftest import relerr

CUBE = "-1 -1 -1 1 -1 -1 1 1 -1 1 1 1 1 -1 1 -1 1 -1 -1 1 1 -1 -1 1"
DEGENERATE = """-0.0611590669 -0.13801524  -0.158372656  0.0620514415  0.135089189 -0.159879193  -0.105518319  -0.100999095 -0.188289702
    -0.107238553  -0.102976903  0.1569262  -0.0851279497 -0.122304708  0.156887323  -0.0590926372 -0.104567274 -0.242715642"""


def _geom_dist(xml, g1=0, g2=1, margin=0.0):
  """reference collision_gjk_test.py:_geom_dist on two static geoms of a model."""
  m = mjw.mjcf.from_xml_string(xml)

  def geom(g):
    vert = None
    if m.geom_type[g] == 7:
      i = m.geom_dataid[g]
      vert = m.mesh_vert[m.mesh_vertadr[i] : m.mesh_vertadr[i] + m.mesh_vertnum[i]]
    return int(m.geom_type[g]), m.geom_pos[g], nm.quat_to_mat(m.geom_quat[g]), m.geom_size[g], vert

  t1, p1, R1, s1, v1 = geom(g1)
  t2, p2, R2, s2, v2 = geom(g2)
  return ref.ccd(t1, p1, R1, s1, t2, p2, R2, s2, margin=margin, vert1=v1, vert2=v2)


def test_reference_held_mesh_vectors():
  d, n, _, _, _ = _geom_dist(f'<mujoco><asset><mesh name="smallbox" scale="0.1 0.1 0.1" vertex="{CUBE}"/></asset><worldbody>'
                             '<geom pos="0 0 .90" type="box" size="0.5 0.5 0.1"/><geom pos="0 0 1.2" type="mesh" mesh="smallbox"/></worldbody></mujoco>')
  assert abs(d - 0.1) < 1e-7
  d, n, _, _, _ = _geom_dist(f'<mujoco><asset><mesh name="box" scale=".5 .5 .1" vertex="{CUBE}"/><mesh name="smallbox" scale=".1 .1 .1" vertex="{CUBE}"/></asset>'
                             '<worldbody><geom pos="0 0 .09" type="mesh" mesh="smallbox"/><geom pos="0 0 -.1" type="mesh" mesh="box"/></worldbody></mujoco>')
  assert abs(d + 0.01) < 1e-7
  d, n, _, _, _ = _geom_dist(f'<mujoco><asset><mesh name="mesh" vertex="{DEGENERATE}"/></asset><worldbody>'
                             '<geom type="mesh" mesh="mesh" pos="-0.141666584 0 0" quat="0.5425650813 0.0029009761 0.0001424328 0.8400087479"/>'
                             '<geom type="mesh" mesh="mesh" pos="0.141666584 0 0" quat="0.5425650813 0.0029009761 0.0001424328 0.8400087479"/></worldbody></mujoco>')
  assert n == 1 and abs(d + 0.0031312597856874586) < 1e-7  # (assertAlmostEqual: 7 places)
  d, n, _, _, _ = _geom_dist(f'<mujoco><asset><mesh name="box" scale=".2 .2 .2" vertex="{CUBE}"/></asset><worldbody>'
                             '<geom type="sphere" pos="0 0 .349" size=".1"/><geom type="mesh" mesh="box"/></worldbody></mujoco>', margin=0.05)
  assert abs(d + 0.001) < 1e-7


MESH_SCENE = """
<mujoco>
  <option timestep="0.004"><flag multiccd="disable"/></option>
  <asset>
    <mesh name="cube" vertex="-.1 -.15 -.2  .1 -.15 -.2  -.1 .15 -.2  .1 .15 -.2  -.1 -.15 .2  .1 -.15 .2  -.1 .15 .2  .1 .15 .2"/>
    <mesh name="wedge" vertex="0 0 0  .3 0 0  0 .2 0  .3 .2 0  0 0 .15  0 .2 .15"/>
    <mesh name="gem" vertex=".12 0 0  -.12 0 0  0 .1 0  0 -.1 0  0 0 .16  0 0 -.09  .07 .06 .08  -.06 -.07 .05"/>
  </asset>
  <worldbody>
    <geom type="plane" size="5 5 .1"/>
    <body name="cube" pos="0 0 .25" euler="10 5 0"><freejoint/><geom type="mesh" mesh="cube" pos=".02 0 0"/></body>
    <body name="wedge" pos=".6 0 .2" euler="0 20 30"><freejoint/><geom type="mesh" mesh="wedge"/></body>
    <body name="gem" pos="0 .6 .3" euler="40 0 10"><freejoint/><geom type="mesh" mesh="gem"/></body>
    <body name="gem2" pos=".05 .62 .62" euler="0 30 0"><freejoint/><geom type="mesh" mesh="gem"/></body>
    <body name="ball" pos=".03 0 .62"><freejoint/><geom type="sphere" size=".08"/></body>
    <body name="cap" pos=".62 .05 .5" euler="0 80 0"><freejoint/><geom type="capsule" size=".04 .1"/></body>
    <body name="box" pos="-.6 0 .12"><freejoint/><geom type="box" size=".1 .1 .1"/></body>
    <body name="ell" pos="-.6 0 .4"><freejoint/><geom type="ellipsoid" size=".1 .07 .05"/></body>
    <body name="cyl" pos="-.6 .6 .3" euler="90 0 0"><freejoint/><geom type="cylinder" size=".06 .1"/></body>
    <body name="wedge2" pos="-.65 .62 .12" euler="0 0 50"><freejoint/><geom type="mesh" mesh="wedge"/></body>
    <body name="gem3" pos="-.58 .02 .62"><freejoint/><geom type="mesh" mesh="gem"/></body>
  </worldbody>
</mujoco>
"""


def test_loader_mesh_equals_box():
  box = MESH_SCENE.replace('<geom type="mesh" mesh="cube" pos=".02 0 0"/>', '<geom type="box" size=".1 .15 .2" pos=".02 0 0"/>')
  a, b = mjw.mjcf.from_xml_string(MESH_SCENE), mjw.mjcf.from_xml_string(box)
  assert abs(a.body_mass[1] - b.body_mass[1]) < 1e-9 and np.allclose(a.body_inertia[1], b.body_inertia[1], atol=1e-12) and np.allclose(a.body_ipos[1], b.body_ipos[1], atol=1e-12)
  assert np.allclose(a.geom_size[1], [0.1, 0.15, 0.2]) and abs(a.geom_rbound[1] - b.geom_rbound[1]) < 1e-12
  # every mesh sits in its principal frame, centred at its centre of mass
  for i in range(a.nmesh):
    v = a.mesh_vert[a.mesh_vertadr[i] : a.mesh_vertadr[i] + a.mesh_vertnum[i]]
    assert np.abs(mjw.mjcf._compile_mesh(v)["pos"]).max() < 1e-12
  with pytest.raises(NotImplementedError, match="multi-contact"):
    mjw.put_model(mjw.mjcf.from_xml_string(MESH_SCENE.replace('<flag multiccd="disable"/>', "")))


def test_oracle_mesh_cube_equals_box_primitive():
  """A cube given as a mesh must collide like the box: plane contacts (positions, distances) and GJK distances against the other shapes."""
  box = MESH_SCENE.replace('<geom type="mesh" mesh="cube" pos=".02 0 0"/>', '<geom type="box" size=".1 .15 .2" pos=".02 0 0"/>')
  a = ref.RefSim(mjw.mjcf.from_xml_string(MESH_SCENE), nconmax=64, njmax=256)
  b = ref.RefSim(mjw.mjcf.from_xml_string(box), nconmax=64, njmax=256)
  for s in (a, b):
    s.qpos[2] = 0.19  # cube touching the floor with one edge region, ball resting on it
    s.qpos[3:7] = nm.quat_normalize(np.array([1.0, 0.02, 0.0, 0.0]))
    s.qpos[7 * 4 + 2] = 0.47
    s.forward()

  def contacts(s, g):
    return sorted((round(float(s.con_dist[c]), 9), tuple(np.round(s.con_pos[c], 9))) for c in range(s.ncon) if g in s.con_geom[c])

  ca, cb = contacts(a, 1), contacts(b, 1)
  # plane_convex keeps the vertices within 1 mm of the deepest one (here the two corners of the lowest edge), plane_box every penetrating
  # corner: the mesh's contacts are the deepest of the box's; the sphere contact (GJK on the mesh, sphere_box on the primitive) is the same
  assert len(ca) == 3 and len(cb) == 5
  for da, pa in ca:
    assert any(abs(da - db) < 1e-6 and np.allclose(pa, pb, atol=1e-6) for db, pb in cb), (da, pa)
  assert abs(ca[0][0] - cb[0][0]) < 1e-9


def test_oracle_plane_tetrahedron_separated():  # collision_driver_test.py:691-711
  xml = ('<mujoco><asset><mesh name="tet" vertex="-1 0 0.1  1 0 0.1  0 1 0.1  0 0.5 1.1"/></asset><worldbody><geom type="plane" size="5 5 .1"/>'
         '<body><freejoint/><geom type="mesh" mesh="tet"/></body></worldbody></mujoco>')
  s = ref.RefSim(mjw.mjcf.from_xml_string(xml))
  s.forward()
  assert s.ncon == 0  # separated in z by 0.1: the reference asserts every distance > 0.05 (no contact is written)


@pytest.mark.gpu
def test_gpu_mesh_scene_vs_oracle():
  mjm = mjw.mjcf.from_xml_string(MESH_SCENE)
  m = mjw.put_model(mjm)
  nworld = 3
  d = mjw.make_data(mjm, nworld=nworld, nconmax=64, njmax=256)
  sims = [ref.RefSim(mjm, nconmax=64, njmax=256) for _ in range(nworld)]
  rng = np.random.default_rng(3)
  q = d.qpos.numpy()
  for w in range(1, nworld):
    q[w, 0::7] += rng.uniform(-0.02, 0.02, q[w, 0::7].shape)
  d.qpos.assign(q)
  for w, s in enumerate(sims):
    s.qpos[:] = q[w]
  ncon_seen = set()
  for step in range(150):
    for w, s in enumerate(sims):  # per-step parity from a common state (contact scenes are chaotic)
      s.qpos[:] = d.qpos.numpy()[w]
      s.qvel[:] = d.qvel.numpy()[w]
      s.qacc_warmstart[:] = d.qacc_warmstart.numpy()[w]
    mjw.step(m, d)
    nacon = 0
    for w, s in enumerate(sims):
      s.step()
      assert int(d.ws_ncon.numpy()[w]) == s.ncon, (step, w, int(d.ws_ncon.numpy()[w]), s.ncon)
      nacon += s.ncon
      assert relerr(d.qpos.numpy()[w], s.qpos) < 2e-4, (step, w)
      ncon_seen.add(s.ncon)
    assert int(d.nacon.numpy()[0]) == nacon
  assert (d.overflow.numpy() == 0).all() and max(ncon_seen) >= 12
