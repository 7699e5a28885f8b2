"""Convex mesh geoms (first part): mesh support function in GJK / EPA (reference collision_gjk.py:154-169,
exhaustive vertex search), plane-mesh collider (collision_primitive.py:52-274 plane_convex, exhaustive branch), mesh assets in Model
(types.py:1707-1709: mesh_vertadr / mesh_vertnum / mesh_vert, geom_dataid).

Reference-held numbers reproduced by the float64 oracle (collision_gjk_test.py): box-mesh distance 0.1 (:341-366), mesh-mesh
penetration -0.01 (:405-439), the degenerate mesh pair -0.0031312597856874586 with one contact (:441-467), sphere-mesh with margin
-0.001 (:648-665); collision_driver_test.py:691-711 (plane and a separated tetrahedron: no contact closer than 0.05).  The loader's
mesh compilation (hull, centre of mass, principal frame) is pinned by a cube given as a mesh, which must equal the box primitive.
Multi-contact recovery on mesh faces (collision_gjk.py:2076) and the hill-climbing branches for meshes of 10 or more vertices follow
further down.  Not built: mesh files, height fields.
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


def test_oracle_mesh_multicontact():
  """collision_gjk_test.py:528-546 test_mesh_mesh_ccd holds 4 contacts for two cube meshes face to face; a cube mesh against a box must
  give the very contacts of the box-box pair (same geometry, the box side drives the clip order); mesh-mesh picks its four points from
  the same clipped polygon (the greedy quadrilateral search starts from a different vertex: DESIGN section 6)."""
  def run(g1, g2):
    xml = f'<mujoco><asset><mesh name="smallbox" vertex="{CUBE}"/></asset><worldbody><geom pos="0 0 2" {g1}/><geom pos="0 1 3.99" euler="0 0 40" {g2}/></worldbody></mujoco>'
    s = ref.RefSim(mjw.mjcf.from_xml_string(xml))
    s.forward()
    return s.ccd_geoms(0, 1)

  mesh, box = 'type="mesh" mesh="smallbox"', 'type="box" size="1 1 1"'
  (dm, nm_, wm), (db, nb, wb), (dc, nc, wc), (de, ne, we) = run(mesh, mesh), run(box, box), run(box, mesh), run(mesh, box)
  assert nm_ == 4 and nb == 4 and nc == 4 and ne == 4  # (reference-held: 4)
  assert abs(dm + 0.01) < 1e-9 and abs(db + 0.01) < 1e-9
  key = lambda w: sorted(map(tuple, w.reshape(len(w), 6).round(6)))
  assert key(wc) == key(wb)
  # every mesh-mesh witness lies on the boundary of the overlap of the two faces (z = 3 / 2.99, inside both squares)
  c, s_ = np.cos(np.deg2rad(40)), np.sin(np.deg2rad(40))
  for w in np.concatenate([wm, we]):
    assert abs(w[0, 2] - 3.0) < 1e-9 and abs(w[1, 2] - 2.99) < 1e-9
    x, y = w[0, 0], w[0, 1]
    assert abs(x) <= 1 + 1e-9 and abs(y) <= 1 + 1e-9
    u, v = c * x + s_ * (y - 1), -s_ * x + c * (y - 1)
    assert abs(u) <= 1 + 1e-9 and abs(v) <= 1 + 1e-9


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
  # polygon tables: the cube has 6 quads, every vertex belongs to 3 of them, vertices counter-clockwise seen from outside
  assert a.mesh_polynum[0] == 6 and (a.mesh_polyvertnum[:6] == 4).all() and (a.mesh_polymapnum[:8] == 3).all()
  v = a.mesh_vert[:8]
  for p in range(6):
    ids = a.mesh_polyvert[a.mesh_polyvertadr[p] : a.mesh_polyvertadr[p] + 4]
    nrm = np.cross(v[ids[1]] - v[ids[0]], v[ids[2]] - v[ids[1]])
    assert np.dot(nrm, a.mesh_polynormal[p]) > 0 and np.allclose(np.cross(nrm, a.mesh_polynormal[p]), 0, atol=1e-12)


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


MESH_STACK = """
<mujoco>
  <option timestep="0.004"/>
  <asset>
    <mesh name="slab" vertex="-.4 -.35 -.05  .4 -.35 -.05  -.4 .35 -.05  .4 .35 -.05  -.4 -.35 .05  .4 -.35 .05  -.4 .35 .05  .4 .35 .05"/>
    <mesh name="wedge" vertex="0 0 0  .3 0 0  0 .2 0  .3 .2 0  0 0 .15  0 .2 .15"/>
    <mesh name="prism" vertex=".1 0 -.08  -.05 .0866 -.08  -.05 -.0866 -.08  .1 0 .08  -.05 .0866 .08  -.05 -.0866 .08"/>
  </asset>
  <worldbody>
    <geom type="plane" size="5 5 .1"/>
    <body pos="0 0 .051"><freejoint/><geom type="mesh" mesh="slab"/></body>
    <body pos=".05 .18 .183" euler="0 0 15"><freejoint/><geom type="box" size=".12 .1 .08"/></body>
    <body pos=".2 -.2 .183" euler="0 0 5"><freejoint/><geom type="mesh" mesh="prism"/></body>
    <body pos="-.35 -.3 .103" euler="0 0 10"><freejoint/><geom type="mesh" mesh="wedge"/></body>
    <body pos=".06 .17 .3435" euler="0 0 -20"><freejoint/><geom type="mesh" mesh="prism"/></body>
  </worldbody>
</mujoco>
"""
# (every body rests with a whole face inside the face below it, so that the clipped polygons are the faces themselves: a polygon of five
# vertices pruned to four by the reference's greedy search is ill-conditioned -- which four survive flips with rounding)


@pytest.mark.gpu
def test_gpu_mesh_multicontact_vs_oracle():
  """Multi-contact recovery on mesh faces (box on a mesh slab, prism and wedge meshes on it): contact counts per step identical to the
  oracle's, positions / frames / distances of every contact within float32 noise, per-step state parity."""
  mjm = mjw.mjcf.from_xml_string(MESH_STACK)
  m = mjw.put_model(mjm)
  assert m.npolygonmax == 4 and m.nmeshpoly == 6 + 5 + 5 and m.nmesh == 3
  d = mjw.make_data(mjm, nworld=2, nconmax=64, njmax=256)
  sims = [ref.RefSim(mjm, nconmax=64, njmax=256) for _ in range(2)]
  q = d.qpos.numpy()
  q[1, 7] += 0.01
  d.qpos.assign(q)
  sims[1].qpos[:] = q[1]
  multi = mismatch = groups = 0
  for step in range(120):
    for w, s in enumerate(sims):
      s.qpos[:] = d.qpos.numpy()[w]
      s.qvel[:] = d.qvel.numpy()[w]
      s.qacc_warmstart[:] = d.qacc_warmstart.numpy()[w]
    mjw.step(m, d)
    adr = 0
    for w, s in enumerate(sims):
      s.step()
      n = int(d.ws_ncon.numpy()[w])
      before = mismatch
      groups += 1
      if n != s.ncon:  # a face pair aligned within FACE_TOL (1.6 mrad) on one side only: 2 contacts instead of 4 (or the reverse) for a step
        mismatch += 1
        adr += n
        assert relerr(d.qpos.numpy()[w], s.qpos) < 3e-3, (step, w)
        continue
      geoms = d.contact.geom.numpy()[adr : adr + n]
      assert (geoms == s.con_geom[:n]).all()
      gpos, gdist, gfr = d.contact.pos.numpy()[adr : adr + n], d.contact.dist.numpy()[adr : adr + n], d.contact.frame.numpy()[adr : adr + n].reshape(n, 9)
      for g1, g2 in sorted(set(map(tuple, s.con_geom[:n]))):
        sel = (s.con_geom[:n] == (g1, g2)).all(axis=1)
        if g1 > 0 and int(sel.sum()) > 1:
          multi += 1
        groups += 1
        # the contacts of a pair as a set: which four vertices of a clipped 5-gon survive the greedy quadrilateral search, and the order
        # they come in, can flip with float32 rounding (as for boxes); distance and normal are those of the EPA face
        order_g = np.lexsort(np.round(gpos[sel], 4).T)
        order_s = np.lexsort(np.round(s.con_pos[:n][sel], 4).T)
        if np.abs(gpos[sel][order_g] - s.con_pos[:n][sel][order_s]).max() > 6e-4:
          mismatch += 1
          continue
        assert np.abs(gdist[sel][order_g] - s.con_dist[:n][sel][order_s]).max() < 1e-4, (step, w)  # (float32 EPA on penetrations of ~1e-4: tests/test_convex.py)
        assert np.abs(gfr[sel][order_g][:, :3] - s.con_frame[:n][sel][order_s][:, :3]).max() < 1e-2, (step, w)
      adr += n
      # (a step on which the two sides kept different vertices of a clipped polygon is solved with slightly different contact points)
      assert relerr(d.qpos.numpy()[w], s.qpos) < (3e-4 if mismatch == before else 3e-3), (step, w)
  assert multi > 200 and (d.overflow.numpy() == 0).all()
  assert mismatch <= 0.03 * groups, (mismatch, groups)


# ---- meshes of 10 or more vertices: hill climbing on the hull's vertex graph (collision_gjk.py:170-196, collision_primitive.py:131-243) ----
_PHI = (1 + 5**0.5) / 2
ICOSA = " ".join(f"{x * 0.08:.6f}" for v in ([0, 1, _PHI], [0, -1, _PHI], [0, 1, -_PHI], [0, -1, -_PHI], [1, _PHI, 0], [-1, _PHI, 0], [1, -_PHI, 0], [-1, -_PHI, 0],
                                              [_PHI, 0, 1], [-_PHI, 0, 1], [_PHI, 0, -1], [-_PHI, 0, -1]) for x in v)
CUBOCTA = " ".join(f"{x * 0.1:.6f}" for v in ([1, 1, 0], [1, -1, 0], [-1, 1, 0], [-1, -1, 0], [1, 0, 1], [1, 0, -1], [-1, 0, 1], [-1, 0, -1], [0, 1, 1], [0, 1, -1],
                                              [0, -1, 1], [0, -1, -1]) for x in v)
GRAPH_SCENE = f"""
<mujoco>
  <option timestep="0.004"/>
  <asset><mesh name="ico" vertex="{ICOSA}"/><mesh name="cubo" vertex="{CUBOCTA}"/></asset>
  <worldbody>
    <geom type="plane" size="5 5 .1"/>
    <body pos="0 0 .2" euler="20 10 0"><freejoint/><geom type="mesh" mesh="ico"/></body>
    <body pos=".4 0 .2" euler="0 30 10"><freejoint/><geom type="mesh" mesh="cubo"/></body>
    <body pos=".02 .01 .5" euler="50 0 20"><freejoint/><geom type="mesh" mesh="cubo"/></body>
    <body pos=".41 .02 .5"><freejoint/><geom type="sphere" size=".07"/></body>
    <body pos="-.4 0 .15"><freejoint/><geom type="box" size=".1 .1 .1"/></body>
    <body pos="-.38 .03 .45" euler="0 40 0"><freejoint/><geom type="mesh" mesh="ico"/></body>
    <body pos="0 .45 .2" euler="90 0 0"><freejoint/><geom type="capsule" size=".05 .12"/></body>
    <body pos="0 .47 .45" euler="10 10 10"><freejoint/><geom type="mesh" mesh="ico"/></body>
  </worldbody>
</mujoco>
"""


def test_oracle_hill_climb_equals_exhaustive_support():
  """On a convex mesh the graph walk ends at the global support vertex: GJK / EPA distances with the graph equal those of the exhaustive
  search (mesh_graphadr = -1) on random poses; the loader's graph has MuJoCo's layout."""
  mjm = mjw.mjcf.from_xml_string(GRAPH_SCENE)
  assert mjm.mesh_vertnum.tolist() == [12, 12] and (mjm.mesh_graphadr >= 0).all()
  g = mjm.mesh_graph[mjm.mesh_graphadr[0] :]
  nv, nf = int(g[0]), int(g[1])
  assert nv == 12 and nf == 20 and sorted(g[2 + nv : 2 + 2 * nv]) == list(range(12))
  edges = g[2 + 2 * nv : 2 + 2 * nv + nv + 3 * nf]
  assert (edges == -1).sum() == nv and all(int((edges[g[2 + l] :] == -1).argmax()) == 5 for l in range(nv))  # icosahedron: 5 neighbours each
  a = ref.RefSim(mjm, nconmax=64, njmax=256)
  nog = mjw.mjcf.from_xml_string(GRAPH_SCENE)
  nog.mesh_graphadr[:] = -1
  b = ref.RefSim(nog, nconmax=64, njmax=256)
  rng = np.random.default_rng(5)
  pairs = [(1, 2), (1, 3), (2, 6), (1, 4), (5, 6), (7, 8), (3, 8)]
  n = 0
  for trial in range(60):
    for s in (a, b):
      s.forward()
    g1, g2 = pairs[trial % len(pairs)]
    q = rng.normal(size=4)
    R = nm.quat_to_mat(q / np.linalg.norm(q))
    p2 = a.geom_xpos[g1] + rng.normal(size=3) * 0.09
    da, na, _ = a.ccd_geoms(g1, g2, pos2=p2, mat2=R, multiccd=False)
    db, nb, _ = b.ccd_geoms(g1, g2, pos2=p2, mat2=R, multiccd=False)
    assert na == nb and abs(da - db) < 1e-9, (trial, da, db)
    n += da < 0
  assert n > 10


def test_oracle_plane_mesh_hill_climb_contacts_are_low_vertices():
  mjm = mjw.mjcf.from_xml_string(GRAPH_SCENE)
  s = ref.RefSim(mjm, nconmax=64, njmax=256)
  s.qpos[2] = 0.12  # icosahedron resting into the floor
  s.forward()
  cons = [c for c in range(s.ncon) if tuple(s.con_geom[c]) == (0, 1)]
  assert 1 <= len(cons) <= 4
  v = mjm.mesh_vert[:12] @ s.geom_xmat[1].reshape(3, 3).T + s.geom_xpos[1]
  zmin = v[:, 2].min()
  for c in cons:  # each contact is a vertex within 1 mm of the lowest one, reported midway between vertex and plane
    k = int(np.argmin(np.linalg.norm(v[:, :2] - s.con_pos[c][:2], axis=1)))
    assert np.linalg.norm(v[k, :2] - s.con_pos[c][:2]) < 1e-9 and abs(s.con_dist[c] - v[k, 2]) < 1e-9 and v[k, 2] < zmin + 1e-3
    assert abs(s.con_pos[c][2] - 0.5 * v[k, 2]) < 1e-9
  assert min(s.con_dist[c] for c in cons) == pytest.approx(zmin, abs=1e-9)


@pytest.mark.gpu
def test_gpu_graph_meshes_vs_oracle():
  """12-vertex meshes (hill-climbing support and plane collider) against plane / sphere / box / capsule / mesh: per-step parity."""
  mjm = mjw.mjcf.from_xml_string(GRAPH_SCENE)
  m = mjw.put_model(mjm)
  assert m.nmeshgraph == len(mjm.mesh_graph) > 0
  d = mjw.make_data(mjm, nworld=2, nconmax=64, njmax=256)
  sims = [ref.RefSim(mjm, nconmax=64, njmax=256) for _ in range(2)]
  q = d.qpos.numpy()
  q[1, 0::7] += 0.013
  d.qpos.assign(q)
  seen = set()
  flicker = total = 0
  for step in range(150):
    for w, s in enumerate(sims):
      s.qpos[:] = d.qpos.numpy()[w]
      s.qvel[:] = d.qvel.numpy()[w]
      s.qacc_warmstart[:] = d.qacc_warmstart.numpy()[w]
    mjw.step(m, d)
    for w, s in enumerate(sims):
      s.step()
      total += 1
      if int(d.ws_ncon.numpy()[w]) != s.ncon:  # (plane_convex keeps the vertices within 1 mm of the deepest: a rocking mesh crosses that line)
        flicker += 1
        assert relerr(d.qpos.numpy()[w], s.qpos) < 3e-3, (step, w)
        continue
      assert relerr(d.qpos.numpy()[w], s.qpos) < 3e-4, (step, w)
      seen |= set(map(tuple, s.con_geom[: s.ncon]))
  assert flicker <= 0.03 * total, (flicker, total)
  assert {(0, 1), (0, 2)} <= seen and len(seen) >= 6, seen
  assert (d.overflow.numpy() == 0).all()
