"""Mesh / height-field FILE assets: binary + ASCII STL, OBJ, `maxhullvert`, PNG and binary height fields, `meshdir`.

The files are written by the tests (a regular tetrahedron and a dodecahedron -- the shapes of the reference's
test_data/meshes/{tetrahedron,dodecahedron}.stl, which are loaded too wherever /root/reference exists); a model compiled from a file
must equal the model compiled from the same vertices inline, and step like it on the GPU.
"""

import os
import struct

import numpy as np
import pytest

import mujoco_warp_amd as mjw
from mujoco_warp_amd import mjcf

REF_MESHES = "/root/reference/mujoco_warp/test_data/meshes"
PHI = (1 + 5**0.5) / 2
TETRA = np.array([[1, 1, 1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1]], dtype=np.float64)
SA, SB = np.array([0.1, 0.12, 0.15]), np.array([0.05, 0.06, 0.08])  # anisotropic scales: a regular solid has no unique principal frame
DODECA = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)] + [[0, s / PHI, t * PHI] for s in (-1, 1) for t in (-1, 1)]
                  + [[s / PHI, t * PHI, 0] for s in (-1, 1) for t in (-1, 1)] + [[s * PHI, 0, t / PHI] for s in (-1, 1) for t in (-1, 1)], dtype=np.float64)


def _triangles(verts):
  from scipy.spatial import ConvexHull

  h = ConvexHull(verts)
  tris = []
  for tri, eq in zip(h.simplices, h.equations):
    a, b, c = verts[tri]
    tris.append((a, b, c) if np.dot(np.cross(b - a, c - a), eq[:3]) > 0 else (a, c, b))
  return np.array(tris)


def write_binary_stl(path, verts):
  tris = _triangles(verts)
  with open(path, "wb") as f:
    f.write(b"written by tests/test_assets.py".ljust(80, b"\0"))
    f.write(struct.pack("<I", len(tris)))
    for t in tris:
      n = np.cross(t[1] - t[0], t[2] - t[0])
      f.write(struct.pack("<12fH", *(n / np.linalg.norm(n)), *t.reshape(-1), 0))


def write_ascii_stl(path, verts):
  with open(path, "w") as f:
    f.write("solid test\n")
    for t in _triangles(verts):
      n = np.cross(t[1] - t[0], t[2] - t[0])
      f.write("facet normal %r %r %r\n outer loop\n" % tuple(n / np.linalg.norm(n)))
      for v in t:
        f.write("  vertex %r %r %r\n" % tuple(float(x) for x in v))
      f.write(" endloop\nendfacet\n")
    f.write("endsolid test\n")


def write_obj(path, verts):
  lookup = {tuple(v): i + 1 for i, v in enumerate(verts)}
  with open(path, "w") as f:
    f.write("# test\n")
    for v in verts:
      f.write("v %r %r %r\n" % tuple(float(x) for x in v))
    for t in _triangles(verts):
      f.write("f " + " ".join("%d//1" % lookup[tuple(v)] for v in t) + "\n")


SCENE = """
<mujoco>
  <compiler meshdir="{meshdir}"/>
  <option timestep="0.002"/>
  <asset>{assets}</asset>
  <worldbody>
    <geom type="plane" size="0 0 .1"/>
    <body pos="0 0 .2" euler="20 30 0"><freejoint/><geom type="mesh" mesh="a"/></body>
    <body pos=".5 0 .2" euler="0 10 40"><freejoint/><geom type="mesh" mesh="b"/></body>
  </worldbody>
</mujoco>
"""


def _same_model(m1, m2):
  for f in ("mesh_vert", "mesh_vertnum", "mesh_graph", "mesh_polyvert", "mesh_polynormal", "geom_size", "geom_pos", "geom_quat", "geom_rbound", "geom_aabb",
            "body_mass", "body_inertia", "body_ipos", "body_iquat"):
    np.testing.assert_allclose(np.asarray(getattr(m1, f), dtype=np.float64), np.asarray(getattr(m2, f), dtype=np.float64), rtol=0, atol=1e-12, err_msg=f)


def _body_frame_vertices(m, k):
  from tests.test_reference_gjk_gpu import _quat_mat

  g = int(np.nonzero(np.asarray(m.geom_dataid) == k)[0][0])
  v = m.mesh_vert[m.mesh_vertadr[k] : m.mesh_vertadr[k] + m.mesh_vertnum[k]]
  return m.geom_pos[g] + v @ _quat_mat(m.geom_quat[g]).T


def _inline(verts, scale):
  return " ".join(repr(float(x)) for x in (verts * scale).reshape(-1))


@pytest.mark.parametrize("fmt", ["stl", "stl_ascii", "obj"])
def test_mesh_file_equals_inline_vertices(tmp_path, fmt):
  writer = {"stl": write_binary_stl, "stl_ascii": write_ascii_stl, "obj": write_obj}[fmt]
  ext = "obj" if fmt == "obj" else "stl"
  writer(tmp_path / f"tetra.{ext}", TETRA)
  writer(tmp_path / f"dodeca.{ext}", DODECA)
  from_file = mjcf.from_xml_string(SCENE.format(meshdir=tmp_path, assets=f'<mesh name="a" file="tetra.{ext}" scale=".1 .12 .15"/><mesh name="b" file="dodeca.{ext}" scale=".05 .06 .08"/>'))
  inline = mjcf.from_xml_string(SCENE.format(meshdir=".", assets=f'<mesh name="a" vertex="{_inline(TETRA, SA)}"/><mesh name="b" vertex="{_inline(DODECA, SB)}"/>'))
  assert from_file.nmesh == 2 and list(from_file.mesh_vertnum) == [4, 20]
  # (np.unique sorts the merged STL vertices: compare the vertex SETS and everything that does not depend on their order)
  # and the principal frame of a mesh is defined up to the signs of its axes (eigenvectors): compare the vertices in the BODY frame
  for k in range(2):
    a, b = _body_frame_vertices(from_file, k), _body_frame_vertices(inline, k)
    np.testing.assert_allclose(a[np.lexsort(a.round(9).T)], b[np.lexsort(b.round(9).T)], atol=1e-9)
  for f in ("geom_size", "geom_pos", "geom_rbound", "body_mass", "body_inertia", "body_ipos"):
    np.testing.assert_allclose(getattr(from_file, f), getattr(inline, f), atol=1e-9, err_msg=f)
  # closed forms: regular tetrahedron of edge 2 sqrt 2 (volume 8/3), dodecahedron of edge 2/phi (volume (15 + 7 sqrt 5) / 4 edge^3)
  np.testing.assert_allclose(from_file.body_mass[1], 1000 * 8 / 3 * SA.prod(), rtol=1e-6)
  np.testing.assert_allclose(from_file.body_mass[2], 1000 * (15 + 7 * 5**0.5) / 4 * (2 / PHI) ** 3 * SB.prod(), rtol=1e-6)  # (binary STL stores float32 vertices)


def test_binary_and_ascii_stl_agree(tmp_path):
  write_binary_stl(tmp_path / "b.stl", DODECA)
  write_ascii_stl(tmp_path / "a.stl", DODECA)
  vb, fb = mjcf.read_stl(tmp_path / "b.stl")
  va, fa = mjcf.read_stl(tmp_path / "a.stl")
  assert vb.shape == (20, 3) and fb.shape == (36, 3)
  np.testing.assert_allclose(vb, va, atol=1e-6)  # (binary STL stores float32)
  with pytest.raises(ValueError):
    (tmp_path / "bad.stl").write_bytes(b"\0" * 100)
    mjcf.read_stl(tmp_path / "bad.stl")
  with pytest.raises(NotImplementedError):
    mjcf.read_mesh_file("mesh.msh")


def test_maxhullvert_truncates_the_collision_hull_only(tmp_path):
  write_binary_stl(tmp_path / "d.stl", DODECA)
  xml = SCENE.format(meshdir=tmp_path, assets='<mesh name="a" file="d.stl" scale=".05 .06 .08" maxhullvert="{n}"/><mesh name="b" file="d.stl" scale=".05 .06 .08"/>')
  m = mjcf.from_xml_string(xml.format(n=8))
  assert m.mesh_graph[m.mesh_graphadr[0]] == 8 and m.mesh_graph[m.mesh_graphadr[1]] == 20  # hull vertices of the two assets
  np.testing.assert_allclose(m.body_mass[1], m.body_mass[2], rtol=1e-12)  # mass properties come from the whole mesh
  with pytest.raises(ValueError):
    mjcf.from_xml_string(xml.format(n=3))


def test_missing_mesh_file_is_an_error_only_for_colliding_geoms(tmp_path):
  xml = SCENE.format(meshdir=tmp_path, assets='<mesh name="a" file="nope.stl"/><mesh name="b" vertex="0 0 0 1 0 0 0 1 0 0 0 1"/>')
  with pytest.raises(FileNotFoundError):
    mjcf.from_xml_string(xml)
  m = mjcf.from_xml_string(xml.replace('mesh="a"/>', 'mesh="a" contype="0" conaffinity="0" mass="1"/>').replace("<freejoint/><geom", '<freejoint/><inertial pos="0 0 0" mass="1" diaginertia="1 1 1"/><geom', 1))
  assert m.nmesh == 1  # the visual mesh is not compiled (G1 / Panda rely on this: their STL files are not in the tree)


@pytest.mark.skipif(not os.path.isdir(REF_MESHES), reason="reference tree not present")
def test_reference_stl_fixtures_load():
  """test_data/ray.xml's assets: tetrahedron.stl (4 vertices) and dodecahedron.stl (20 vertices, Blender export)."""
  v, f = mjcf.read_stl(os.path.join(REF_MESHES, "tetrahedron.stl"))
  assert v.shape == (4, 3) and f.shape == (4, 3) and np.abs(v).max() == 1.0
  v, f = mjcf.read_stl(os.path.join(REF_MESHES, "dodecahedron.stl"))
  assert v.shape == (20, 3) and f.shape == (36, 3)
  np.testing.assert_allclose(np.linalg.norm(v, axis=1), np.linalg.norm(v[0]), rtol=1e-5)  # vertices on a sphere
  m = mjcf.from_xml_string(SCENE.format(meshdir=REF_MESHES, assets='<mesh name="a" file="tetrahedron.stl" scale=".4 .4 .4"/><mesh name="b" file="dodecahedron.stl" scale=".04 .04 .04"/>'))
  assert list(m.mesh_vertnum) == [4, 20]
  np.testing.assert_allclose(m.body_mass[1], 1000 * 4 / 3 * 0.4**3, rtol=1e-6)  # (its hull has volume 4/3)


HFIELD_SCENE = """
<mujoco>
  <compiler assetdir="{assetdir}"/>
  <option timestep="0.002"/>
  <asset>{hfield}</asset>
  <worldbody>
    <geom type="hfield" hfield="terrain"/>
    <body pos="0.1 0.2 .5"><freejoint/><geom type="sphere" size=".1"/></body>
    <body pos="-.4 .3 .5"><freejoint/><geom type="box" size=".08 .06 .05"/></body>
  </worldbody>
</mujoco>
"""
ELEV = np.array([[0, 1, 2, 1, 0], [1, 3, 5, 3, 1], [2, 5, 9, 5, 2], [1, 2, 4, 2, 1]], dtype=np.float64)  # as listed in MJCF: first row = +y edge


def test_hfield_files_equal_inline_elevation(tmp_path):
  from PIL import Image

  inline = mjcf.from_xml_string(HFIELD_SCENE.format(assetdir=".", hfield='<hfield name="terrain" nrow="4" ncol="5" size="1 1 .3 .1" elevation="%s"/>' % " ".join(str(x) for x in ELEV.reshape(-1))))
  Image.fromarray((ELEV * 28).astype(np.uint8), mode="L").save(tmp_path / "t.png")  # (top image row = +y, like the MJCF listing)
  png = mjcf.from_xml_string(HFIELD_SCENE.format(assetdir=tmp_path, hfield='<hfield name="terrain" file="t.png" size="1 1 .3 .1"/>'))
  with open(tmp_path / "t.bin", "wb") as f:  # MuJoCo's binary format stores row 0 = the -y edge
    f.write(struct.pack("<2i", 4, 5))
    f.write(ELEV[::-1].astype("<f4").tobytes())
  binf = mjcf.from_xml_string(HFIELD_SCENE.format(assetdir=tmp_path, hfield='<hfield name="terrain" file="t.bin" size="1 1 .3 .1"/>'))
  for m in (png, binf):
    assert (int(m.hfield_nrow[0]), int(m.hfield_ncol[0])) == (4, 5)
    np.testing.assert_allclose(m.hfield_data, inline.hfield_data, atol=1e-12)
    np.testing.assert_allclose(m.hfield_size, inline.hfield_size)
    np.testing.assert_allclose(m.geom_aabb, inline.geom_aabb)
  assert inline.hfield_data.max() == 1.0 and inline.hfield_data.min() == 0.0


@pytest.mark.gpu
def test_gpu_mesh_from_file_steps_like_inline_vertices(tmp_path):
  """A model whose meshes come from STL files steps bit-identically to the same model with inline vertices in the same order, and
  matches the oracle per step (the mesh colliders themselves are covered by tests/test_mesh.py)."""
  from oracle import ref
  from tests.conftest import relerr

  write_binary_stl(tmp_path / "tetra.stl", TETRA)
  write_binary_stl(tmp_path / "dodeca.stl", DODECA)
  mjm = mjcf.from_xml_string(SCENE.format(meshdir=tmp_path, assets='<mesh name="a" file="tetra.stl" scale=".1 .12 .15"/><mesh name="b" file="dodeca.stl" scale=".05 .06 .08"/>'))
  # the same vertices inline, in the file loader's (sorted) order
  va, _ = mjcf.read_stl(tmp_path / "tetra.stl")
  vb, _ = mjcf.read_stl(tmp_path / "dodeca.stl")
  mji = mjcf.from_xml_string(SCENE.format(meshdir=".", assets=f'<mesh name="a" vertex="{_inline(va, SA)}"/><mesh name="b" vertex="{_inline(vb, SB)}"/>'))
  _same_model(mjm, mji)
  s = ref.RefSim(mjm, nconmax=32, njmax=128, tolerance=1e-6)
  s.reset()
  m, mi = mjw.put_model(mjm), mjw.put_model(mji)
  d = mjw.put_data(mjm, mjw.MjData(mjm), nworld=2, nconmax=32, njmax=128)
  di = mjw.put_data(mji, mjw.MjData(mji), nworld=2, nconmax=32, njmax=128)
  worst_q = worst_v = 0.0
  ncon = 0
  for i in range(150):
    for name in ("qpos", "qvel", "qacc_warmstart"):
      v = np.tile(getattr(s, name).astype(np.float32), (2, 1))
      getattr(d, name).assign(v)
      getattr(di, name).assign(v)
    mjw.step(m, d)
    mjw.step(mi, di)
    s.step()
    assert (d.qpos.numpy() == di.qpos.numpy()).all() and (d.qvel.numpy() == di.qvel.numpy()).all()
    ncon += s.ncon
    if int(d.ws_ncon.numpy()[1]) != s.ncon:
      continue  # (a contact within float32 resolution of the detection boundary)
    worst_q = max(worst_q, relerr(d.qpos.numpy()[1], s.qpos))
    worst_v = max(worst_v, relerr(d.qvel.numpy()[1], s.qvel))
  assert ncon > 100  # both bodies land on the plane
  assert worst_q <= 1e-5 and worst_v <= 5e-3, (worst_q, worst_v)


@pytest.mark.gpu
def test_gpu_hfield_from_png_matches_oracle(tmp_path):
  from PIL import Image
  from oracle import ref
  from tests.conftest import relerr

  Image.fromarray((ELEV * 28).astype(np.uint8), mode="L").save(tmp_path / "t.png")
  mjm = mjcf.from_xml_string(HFIELD_SCENE.format(assetdir=tmp_path, hfield='<hfield name="terrain" file="t.png" size="1 1 .3 .1"/>'))
  s = ref.RefSim(mjm, nconmax=32, njmax=128, tolerance=1e-6)
  s.reset()
  m = mjw.put_model(mjm)
  d = mjw.put_data(mjm, mjw.MjData(mjm), nworld=2, nconmax=32, njmax=128)
  same = total = 0
  worst_q = 0.0
  for i in range(200):
    for name in ("qpos", "qvel", "qacc_warmstart"):
      getattr(d, name).assign(np.tile(getattr(s, name).astype(np.float32), (2, 1)))
    mjw.step(m, d)
    s.step()
    total += 1
    if int(d.ws_ncon.numpy()[1]) == s.ncon:
      same += 1
      worst_q = max(worst_q, relerr(d.qpos.numpy()[1], s.qpos))
  assert s.ncon >= 1 and same >= 0.7 * total  # (the prism contact selection is ill-conditioned: tests/test_hfield.py)
  assert worst_q <= 5e-3, worst_q  # (the same bound as tests/test_hfield.py: which prism contacts are kept is a float32-sensitive choice)
