"""Host <-> device IO: the drop-in boundary of the engine.

Mirrors /root/reference/mujoco_warp/_src/io.py: put_model 259 (validation 284-360, derived tables 495-652),
make_data 1680, put_data 1890, get_data_into 2184, reset_data 2435, reset_data_keyframe 2797,
override_model 2933; default sizing _default_nconmax 1284 / _default_njmax 1299 / _get_padded_sizes 1268.
Same signatures, shapes, dtypes and error behaviour (ValueError / NotImplementedError at put/make time;
capacity problems at run time are bits in `d.overflow`).
"""

import ctypes
from typing import Optional

import numpy as np

from . import _abi
from . import types
from .device import DeviceArray

from .mjcf import SENS as mjcf_SENS  # mjtSensor values of the supported sensors

_BATCHED_MODEL_FIELDS = [n[:-3] for n, k, p in _abi.MODEL_FIELDS if n.endswith("_nb")]
_MODEL_PTR_FIELDS = [(n, k) for n, k, p in _abi.MODEL_FIELDS if p]
_DATA_PTR_FIELDS = [(n, k) for n, k, p in _abi.DATA_FIELDS if p]


def _valid_sizes():
  return (2 + (np.arange(19) % 2)) * (2 ** (np.arange(19) // 2 + 3))


def _default_nconmax(mjm, mjd=None):
  vs = _valid_sizes()
  nconmax = max(45, getattr(mjd, "ncon", 0) if mjd is not None else 0)
  return int(nconmax) if nconmax > vs[-1] else int(vs[np.searchsorted(vs, nconmax)])


def _default_njmax(mjm, mjd=None):
  vs = _valid_sizes()
  njmax = max(53, getattr(mjd, "nefc", 0) if mjd is not None else 0)
  return int(njmax) if njmax > vs[-1] else int(vs[np.searchsorted(vs, njmax)])


def _get_padded_sizes(nv, njmax):
  def round_up(x, mult):
    return ((x + mult - 1) // mult) * mult

  return round_up(max(njmax, 1), 16), max(round_up(nv, 4), 4)


def is_sparse(mjm):
  jac = int(mjm.opt.jacobian)
  if jac == 2:  # auto
    return mjm.nv > 32
  return jac == 1


def geom_pairs(mjm):
  """Pre-filtered geom pairs in upper-triangular order (reference io.py:551-577, 631-640)."""
  filterparent = not (int(mjm.opt.disableflags) & types.DisableBit.FILTERPARENT)
  ng = mjm.ngeom
  if ng < 2:
    return np.zeros((0, 2), dtype=np.int32)
  g1, g2 = np.triu_indices(ng, k=1)
  b1, b2 = mjm.geom_bodyid[g1], mjm.geom_bodyid[g2]
  w1, w2 = mjm.body_weldid[b1], mjm.body_weldid[b2]
  wp1 = mjm.body_weldid[mjm.body_parentid[w1]]
  wp2 = mjm.body_weldid[mjm.body_parentid[w2]]
  self_col = w1 == w2
  parent_child = filterparent & (w1 != 0) & (w2 != 0) & ((w1 == wp2) | (w2 == wp1))
  mask = ((mjm.geom_contype[g1] & mjm.geom_conaffinity[g2]) | (mjm.geom_contype[g2] & mjm.geom_conaffinity[g1])) != 0
  exclude = np.isin((b1.astype(np.int64) << 16) + b2, np.asarray(mjm.exclude_signature, dtype=np.int64))
  keep = mask & ~self_col & ~parent_child & ~exclude
  return np.stack([g1[keep], g2[keep]], axis=1).astype(np.int32)


def geom_pairs_with_ids(mjm):
  """(pairs, pairid): the filtered pairs plus the explicit <contact><pair> entries, still in upper-triangular order; pairid is
  the explicit pair's index (its parameters override the geom mixing and it is included whatever the filters say) or -1
  (reference io.py:575-590 nxn_pairid)."""
  pairs = geom_pairs(mjm)
  npair = int(getattr(mjm, "npair", 0))
  if not npair:
    return pairs, np.full(len(pairs), -1, dtype=np.int32)
  table = {(int(a), int(b)): -1 for a, b in pairs}
  for i in range(npair):
    a, b = int(mjm.pair_geom1[i]), int(mjm.pair_geom2[i])
    table[(min(a, b), max(a, b))] = i
  keys = sorted(table)
  return np.array(keys, dtype=np.int32).reshape(-1, 2), np.array([table[k] for k in keys], dtype=np.int32)


_SUPPORTED_PAIRS = {(0, 2), (0, 3), (0, 4), (0, 5), (0, 6), (2, 2), (2, 3), (2, 5), (2, 6), (3, 3), (3, 6), (6, 6)}
# pairs of the reference's CONVEX class (collision_driver.py:47-80) that go through GJK / EPA (csrc/convex.hpp); box-box joins them
# unless DisableBit.NATIVECCD is set (collision_driver.py:867-870), see put_model
_CONVEX_PAIRS = {(2, 4), (3, 4), (3, 5), (4, 4), (4, 5), (4, 6), (5, 5), (5, 6), (2, 7), (3, 7), (4, 7), (5, 7), (6, 7), (7, 7)}
_HFIELD_PAIRS = {(1, 2), (1, 3), (1, 4), (1, 5), (1, 6), (1, 7)}  # height field vs convex geom: prism-wise GJK / EPA (collision_convex.py:164), heavy instantiation
_PLANE_MESH = (0, 7)  # primitive collider plane_convex (collision_primitive.py:52), heavy instantiation


def _pair_index(ngeom, pairs):
  """[ngeom (ngeom - 1) / 2] index into the filtered pair list of every unordered geom pair, -1 if filtered out (the SAP sweep's
  lookup; enumeration = math.upper_tri_index, reference math.py:323, collision_driver.py:485)."""
  out = np.full(max(ngeom * (ngeom - 1) // 2, 1), -1, dtype=np.int32)
  for p, (a, b) in enumerate(pairs):
    a, b = int(min(a, b)), int(max(a, b))
    out[(a * (2 * ngeom - a - 3)) // 2 + b - 1] = p
  return out


def _arr(x, dtype):
  return np.ascontiguousarray(np.asarray(x), dtype=dtype)


def cull_tables(geom_bodyid, pairs, pairid, geom_pos=None, geom_rbound=None):
  """Tables of k_broad_mask's group pre-test (csrc/collide.hpp): the pair list regrouped by PAIRS OF GEOM GROUPS, so that a world tests one
  bounding sphere per group pair before the bounding spheres of its geom pairs (the reference tests every geom pair, collision_driver.py:
  278-334; the pre-test only skips pairs whose sphere test must fail, the mask it produces is the same).

  A group is a maximal run of consecutive geoms of one moving body; every geom of the world body (static scenery spreads over the whole
  scene) is a group of its own; a group's sphere encloses its COLLIDING geoms only (the geoms of some non-explicit pair -- visual geoms have
  no bounding radius) and is centred on one of them, the CENTRE GEOM (the one that gives the smallest sphere in the body frame; any choice
  is valid, the kernel measures the radius in every world).  Returns
    cull_geom  [ncullgeom, 2]  geom + (group << 16) | the group's centre geom -- the colliding geoms, group by group
    cull_group [ncullgroup, 2] centre geom | number of colliding geoms (host-side description, not read by the kernel)
    cull_pair  [ncullpair, 4]  group | group | centre geom 1 + (centre geom 2 << 16) | first entry of cull_list + (count << 24), count <= 16
                               (longer group pairs are split into several rows); group -1: explicit pairs, whose margins are their own,
                               always tested
    cull_list  [npair, 2]      pair index | g1 + (g2 << 16)."""
  geom_bodyid = np.asarray(geom_bodyid).astype(np.int64)
  ng = len(geom_bodyid)
  i32 = np.int32
  if ng == 0 or len(pairs) == 0 or ng > 65535 or len(pairs) >= (1 << 24):
    return np.zeros((0, 2), i32), np.zeros((0, 2), i32), np.zeros((0, 4), i32), np.zeros((0, 2), i32)
  pairid = np.asarray(pairid)
  new = np.ones(ng, dtype=bool)
  new[1:] = (geom_bodyid[1:] != geom_bodyid[:-1]) | (geom_bodyid[1:] == 0)
  run = np.cumsum(new) - 1  # run of every geom
  used = np.zeros(ng, dtype=bool)
  used[pairs[pairid < 0].ravel()] = True
  geoms = np.flatnonzero(used)
  runs, first, count = np.unique(run[geoms], return_index=True, return_counts=True)  # (geoms ascending: a run's geoms are consecutive entries)
  gid = np.full(run.max() + 1, -1, dtype=np.int64)
  gid[runs] = np.arange(len(runs))
  centre = np.zeros(len(runs), dtype=np.int64)
  for k, (a, n) in enumerate(zip(first, count)):
    gs = geoms[a: a + n]
    centre[k] = gs[0]
    if n > 1 and geom_pos is not None and geom_rbound is not None:
      x, rb = np.asarray(geom_pos, dtype=np.float64).reshape(-1, 3)[gs], np.asarray(geom_rbound, dtype=np.float64).reshape(-1)[gs]
      radius = (np.linalg.norm(x[:, None, :] - x[None, :, :], axis=2) + rb[None, :]).max(axis=1)
      centre[k] = gs[int(np.argmin(radius))]
  cgeom = np.stack([geoms + (gid[run[geoms]] << 16), centre[gid[run[geoms]]]], axis=1)
  g1, g2 = pairs[:, 0].astype(np.int64), pairs[:, 1].astype(np.int64)
  key = gid[run[g1]] * (len(runs) + 1) + gid[run[g2]]
  key[pairid >= 0] = -1
  order = np.argsort(key, kind="stable")
  ks = key[order]
  start = np.flatnonzero(np.append(True, ks[1:] != ks[:-1]))
  cnt = np.diff(np.append(start, len(ks)))
  # (at most 16 entries per row: a quarter wavefront serves a surviving row, long group pairs are split)
  nchunk = (cnt + 15) // 16
  rows = np.repeat(np.arange(len(cnt)), nchunk)
  within = np.arange(len(rows)) - np.repeat(np.cumsum(nchunk) - nchunk, nchunk)
  ka, start, cnt = ks[start][rows], start[rows] + 16 * within, np.minimum(16, cnt[rows] - 16 * within)
  ga, gb = np.where(ka < 0, -1, ka // (len(runs) + 1)), np.where(ka < 0, -1, ka % (len(runs) + 1))
  cen = np.append(centre, 0)  # (rows of explicit pairs, group -1, have no centre geoms: the padding entry)
  cp = np.stack([ga, gb, np.where(ka < 0, 0, cen[ga] + (cen[gb] << 16)), start + (cnt << 24)], axis=1)
  cl = np.stack([order, g1[order] + (g2[order] << 16)], axis=1)
  return cgeom.astype(i32), np.stack([centre, count], axis=1).astype(i32), cp.astype(i32), cl.astype(i32)


def put_model(mjm, batch_sizes: Optional[dict] = None) -> types.Model:
  """Creates a model on device (reference io.py:259).

  Args:
    mjm: compiled model (mujoco.MjModel, or the numpy stand-in built by mujoco_warp_amd.mjcf).
    batch_sizes: optional {field name: nworld} making "*" fields per-world (domain randomisation).
  """
  opt = mjm.opt
  # ---- validation (io.py:284-360) ----
  if int(getattr(mjm, "neq", 0)) > 0 and (np.asarray(mjm.eq_type) != types.EqType.JOINT).any():
    raise NotImplementedError("only joint equality constraints are implemented")
  for name in ("ntendon", "nflex", "nplugin"):
    if int(getattr(mjm, name, 0)) > 0:
      raise NotImplementedError(f"{name} > 0 is outside the hot-path scope of this engine")
  if int(opt.integrator) not in (types.IntegratorType.EULER, types.IntegratorType.RK4, types.IntegratorType.IMPLICITFAST, types.IntegratorType.IMPLICIT):
    raise NotImplementedError(f"Integrator {int(opt.integrator)} is unsupported.")
  if int(opt.integrator) == types.IntegratorType.IMPLICIT and mjm.nv > 64:
    raise NotImplementedError("The fully implicit integrator is implemented for models of at most 64 dofs (csrc/implicit.hpp); use implicitfast.")
  # (the reference rejects PGS, io.py solver check / types.py:502; this engine implements MuJoCo C's dual PGS, csrc/pgs.hpp)
  if int(opt.solver) not in (types.SolverType.PGS, types.SolverType.CG, types.SolverType.NEWTON):
    raise NotImplementedError(f"Unknown solver {int(opt.solver)}.")
  if int(opt.cone) not in (types.ConeType.PYRAMIDAL, types.ConeType.ELLIPTIC):
    raise NotImplementedError(f"Unknown cone {int(opt.cone)}.")
  if mjm.nu and (np.asarray(mjm.actuator_trntype) != types.TrnType.JOINT).any():
    raise NotImplementedError("Only joint transmissions are supported.")
  if mjm.nu:
    # the kernels use gear[0], qpos[jnt_qposadr] and the joint's first dof only (smooth.hpp fwd_actuation, integrate.hpp): a
    # motor on a ball / free joint (gear[0:3] / gear[0:6] in the reference, smooth.py:2288-2400) would silently be wrong
    tj = np.asarray(mjm.jnt_type)[np.asarray(mjm.actuator_trnid).reshape(-1, 2)[:, 0]]
    if np.isin(tj, (int(types.JointType.BALL), int(types.JointType.FREE))).any():
      raise NotImplementedError("actuators on ball / free joints are not implemented (hinge and slide targets only).")
  if int(getattr(opt, "noslip_iterations", 0)) > 0:
    raise NotImplementedError("noslip solver is unsupported.")
  if mjm.nu:
    for name, allowed, what in (("actuator_dyntype", (0, 1, 2, 3), "none / integrator / filter / filterexact"),
                                ("actuator_gaintype", (0, 1), "fixed / affine"), ("actuator_biastype", (0, 1), "none / affine")):
      if not np.isin(np.asarray(getattr(mjm, name)), allowed).all():
        raise NotImplementedError(f"{name}: only {what} are implemented (no muscle / user types).")
    if np.asarray(getattr(mjm, "actuator_actearly", [0])).any():
      raise NotImplementedError("actuator actearly is not implemented.")
    if (np.asarray(getattr(mjm, "actuator_actnum", [0])) > 1).any():
      raise NotImplementedError("actuators with more than one activation variable are not implemented.")
  for name in ("jnt_stiffnesspoly", "dof_dampingpoly"):
    if np.asarray(getattr(mjm, name, [0.0])).any():
      raise NotImplementedError(f"{name} (polynomial stiffness / damping) is not implemented.")
  # physics this engine does not compute must not be dropped silently
  if float(getattr(opt, "density", 0.0)) != 0.0 or float(getattr(opt, "viscosity", 0.0)) != 0.0:
    raise NotImplementedError("fluid forces (option density / viscosity, passive.py _fluid_force) are not implemented.")
  unsupported_enable = int(opt.enableflags) & int(types.EnableBit.OVERRIDE | types.EnableBit.FWDINV | types.EnableBit.INVDISCRETE)
  if unsupported_enable:
    raise NotImplementedError(f"enable flags {types.EnableBit(unsupported_enable)!r} are not implemented.")
  # sleeping (reference io.py:356-360): Newton only; active when islands are not disabled (forward.py:345)
  if (int(opt.enableflags) & int(types.EnableBit.SLEEP)) and int(opt.solver) != int(types.SolverType.NEWTON):
    raise ValueError(f"sleeping requires the Newton solver (got solver={types.SolverType(int(opt.solver)).name})")
  if (int(opt.enableflags) & int(types.EnableBit.SLEEP)) and int(opt.integrator) == int(types.IntegratorType.RK4):
    raise NotImplementedError("sleeping with the RK4 integrator is not implemented.")
  pairs, pairid = geom_pairs_with_ids(mjm)
  cull = cull_tables(mjm.geom_bodyid, pairs, pairid, mjm.geom_pos, mjm.geom_rbound)
  gt = np.asarray(mjm.geom_type)
  for a, b in pairs:
    t = (int(min(gt[a], gt[b])), int(max(gt[a], gt[b])))
    if t in ((0, 1), (1, 1)):
      continue  # (plane / height-field against a height field: no collider in the reference either, collision_driver.py:47-59)
    if t not in _SUPPORTED_PAIRS and t not in _CONVEX_PAIRS and t != _PLANE_MESH and t not in _HFIELD_PAIRS:
      raise NotImplementedError(f"collision between geom types {t} is not implemented yet")
    if 7 in t:  # convex meshes: exhaustive vertex search, single contact
      for g in (a, b):
        if int(gt[g]) != 7:
          continue
        mid = int(np.asarray(getattr(mjm, "geom_dataid", np.full(len(gt), -1)))[g])
        if mid < 0 or not hasattr(mjm, "mesh_vert"):
          raise NotImplementedError("colliding mesh geom without mesh vertices (Model.mesh_vert / geom_dataid)")
  condims = set(int(c) for c in np.unique(np.asarray(mjm.geom_condim)[np.unique(pairs)])) if len(pairs) else set()
  nexplicit = int(getattr(mjm, "npair", 0))
  if nexplicit:
    condims |= set(int(c) for c in np.asarray(mjm.pair_dim))
  if not condims <= {1, 3, 4, 6}:
    raise NotImplementedError(f"unsupported condim values {condims}")

  ngeom_ = int(mjm.ngeom)
  m = types.Model()
  nv, nbody, njnt, ngeom, nu = int(mjm.nv), int(mjm.nbody), int(mjm.njnt), int(mjm.ngeom), int(mjm.nu)
  for name in ("nq", "nv", "nu", "na", "nbody", "njnt", "ngeom", "nsite", "nkey", "nmocap"):
    setattr(m, name, int(getattr(mjm, name, 0)))
  for name in ("neq", "ntendon", "nsensor", "nmesh", "nflex", "nhfield", "ncam", "nlight"):
    setattr(m, name, int(getattr(mjm, name, 0)))
  m.nC = int(np.sum(mjm.M_rownnz)) if nv else 0
  m.nM = m.nC
  # capsule-box / box-box pairs and explicit contact pairs select the kernel instantiation that carries them (include/mjhip.h)
  ptypes = [(int(min(gt[a], gt[b])), int(max(gt[a], gt[b]))) for a, b in pairs]
  # box-box is a CONVEX pair (GJK / EPA + multi-contact) unless DisableBit.NATIVECCD selects the primitive collider
  # (reference collision_driver.py:78, 867-870)
  box_ccd = not (int(opt.disableflags) & int(types.DisableBit.NATIVECCD))
  nboxbox = sum(t == (6, 6) for t in ptypes) if box_ccd else 0
  nconvex = sum(t in _CONVEX_PAIRS or t in _HFIELD_PAIRS for t in ptypes)
  m._convex_pairs = int(nconvex + nboxbox > 0)
  # EPA iteration cap (reference collision_convex.py:1223): 16 when every convex pair of the model is box-box
  m._epa_iterations = 16 if (nboxbox > 0 and nconvex == 0) else int(getattr(opt, "ccd_iterations", 35))
  m._heavy_pairs = int(any((int(min(gt[a], gt[b])), int(max(gt[a], gt[b]))) in ((3, 6), (6, 6), (0, 7)) for a, b in pairs) or int(getattr(mjm, "npair", 0)) > 0
                       or m._convex_pairs)
  m.heavy_colliders = m._heavy_pairs  # c_model() adds the broadphase options (they may be changed after put_model)
  # mesh / height-field geoms a ray could hit (visible colour: the reference's _ray_eliminate keeps them, ray.py:52): rays against
  # them are not implemented, so rays() and the rangefinder sensor refuse such models instead of silently reporting "no hit"
  g_rgba = np.asarray(getattr(mjm, "geom_rgba", np.tile([0.5, 0.5, 0.5, 1.0], (ngeom, 1))), dtype=np.float64).reshape(-1, 4)
  g_mat = np.asarray(getattr(mjm, "geom_matid", np.full(ngeom, -1))).reshape(-1)
  m_rgba = np.asarray(getattr(mjm, "mat_rgba", np.zeros((0, 4))), dtype=np.float64).reshape(-1, 4)
  m._ray_unsupported_geoms = int(sum(1 for g in range(ngeom) if int(gt[g]) in (int(types.GeomType.MESH), int(types.GeomType.HFIELD))
                                     and (m_rgba[g_mat[g], 3] if g_mat[g] >= 0 else g_rgba[g, 3]) != 0.0))
  if m._ray_unsupported_geoms and any(int(t) == 7 for t in np.asarray(getattr(mjm, "sensor_type", np.zeros(0))).reshape(-1)):
    raise NotImplementedError("rangefinder sensor in a model with visible mesh / height-field geoms: rays against those geoms are not implemented "
                              "(they would be reported as no hit); make them invisible to rays (rgba alpha 0) or drop the sensor")
  m.is_sparse = False

  m.nv_pad = _get_padded_sizes(nv, 1)[1]

  # ---- options (io.py:392-470) ----
  o = types.Option()
  o._root = m
  f32 = np.float32
  o.timestep = np.array([opt.timestep], dtype=f32)
  # float32 device arithmetic cannot resolve MuJoCo's default 1e-8 (io.py:398-401)
  o.tolerance = np.array([max(float(opt.tolerance), 1e-6)], dtype=f32)
  o.ls_tolerance = np.array([opt.ls_tolerance], dtype=f32)
  o.gravity = _arr(opt.gravity, f32).reshape(1, 3)
  o.impratio_invsqrt = np.array([1.0 / np.sqrt(max(float(opt.impratio), types.MJ_MINVAL))], dtype=f32)
  o.ccd_tolerance = np.array([float(getattr(opt, "ccd_tolerance", 1e-6))], dtype=f32)
  o.magnetic = _arr(getattr(opt, "magnetic", [0.0, -0.5, 0.0]), f32).reshape(1, 3)
  for name in ("integrator", "cone", "solver", "iterations", "ls_iterations", "disableflags", "enableflags"):
    setattr(o, name, int(getattr(opt, name)))
  o.ccd_iterations = int(getattr(opt, "ccd_iterations", 35))
  m.epa_iterations = m._epa_iterations
  # reference io.py:631-636: NXN below 250k filtered pairs, SAP above (tile sort below 1000 geoms)
  o.broadphase = (types.BroadphaseType.NXN if len(pairs) < 250_000 else
                  types.BroadphaseType.SAP_TILE if ngeom_ < 1000 else types.BroadphaseType.SAP_SEGMENTED)
  # the reference's default adds the OBB filter (io.py:405); contacts do not depend on the filter set (every filter is
  # conservative), only Data.ncollision does, and plane + sphere keeps the light collision kernel
  o.broadphase_filter = types.BroadphaseFilter.PLANE | types.BroadphaseFilter.SPHERE
  if m._heavy_pairs:
    # models that run the heavy collision instantiation anyway get the reference's full default (io.py:405): on the ALOHA scene the
    # sphere test alone lets 760 of 9,154 pairs through to GJK (long extrusions, a table-sized box), the box filters 15
    o.broadphase_filter |= types.BroadphaseFilter.AABB | types.BroadphaseFilter.OBB
  o.graph_conditional = False
  o.run_collision_detection = True
  o.warn_overflow = False
  s = types.Statistic()
  s._root = m
  s.meaninertia = np.array([mjm.stat.meaninertia], dtype=f32)

  # ---- derived tables ----
  parent = _arr(mjm.body_parentid, np.int32)
  depth = np.zeros(nbody, dtype=np.int32)
  for b in range(1, nbody):
    depth[b] = depth[parent[b]] + 1
  order = np.argsort(depth, kind="stable").astype(np.int32)
  nlevel = int(depth.max()) + 1 if nbody else 0
  leveladr = np.zeros(nlevel + 1, dtype=np.int32)
  for l in range(nlevel):
    leveladr[l + 1] = leveladr[l] + int(np.sum(depth == l))
  subtreenum = np.ones(nbody, dtype=np.int32)
  for b in range(nbody - 1, 0, -1):
    subtreenum[parent[b]] += subtreenum[b]
  for b in range(1, nbody):  # depth-first numbering => each subtree is a contiguous id range
    if parent[b] >= b or not (parent[b] < b < parent[b] + subtreenum[parent[b]]):
      raise ValueError("bodies must be numbered depth-first (MuJoCo order)")
  dofnum, dofadr = _arr(mjm.body_dofnum, np.int32), _arr(mjm.body_dofadr, np.int32)
  lastdof = np.full(nbody, -1, dtype=np.int32)
  for b in range(1, nbody):
    lastdof[b] = dofadr[b] + dofnum[b] - 1 if dofnum[b] > 0 else lastdof[parent[b]]
  dof_parent = _arr(mjm.dof_parentid, np.int32)
  # kinematic trees: dofs of the bodies below one child of the world, contiguous in MuJoCo's depth-first order
  dof_root = np.zeros(nv, dtype=np.int32)
  for i in range(nv):
    dof_root[i] = i if dof_parent[i] < 0 else dof_root[dof_parent[i]]
  roots = sorted(set(int(r) for r in dof_root))
  tree_dofadr = np.array(roots, dtype=np.int32)
  tree_dofnum = np.array([int(np.sum(dof_root == r)) for r in roots], dtype=np.int32)
  dof_treeid = np.array([roots.index(int(r)) for r in dof_root], dtype=np.int32)
  for t, r in enumerate(roots):
    if not (dof_treeid[r : r + tree_dofnum[t]] == t).all():
      raise ValueError("dofs of a kinematic tree must be contiguous (MuJoCo order)")
  body_treeid = np.array([dof_treeid[lastdof[b]] if lastdof[b] >= 0 else -1 for b in range(nbody)], dtype=np.int32)
  m.ntree = len(roots)
  if mjm.nu:
    adof = np.asarray(mjm.jnt_dofadr)[np.asarray(mjm.actuator_trnid).reshape(-1, 2)[:, 0]]
    m.act_dof_max = int(np.bincount(adof, minlength=max(nv, 1)).max())
    # d(force)/d(velocity) of an actuator = gainprm[2] * ctrl (affine gain) + biasprm[2] (affine bias): non-positive for every ctrl only
    # without a gain term and with biasprm[2] <= 0 (a position servo's -kv); otherwise the implicitfast matrix can lose definiteness
    gp, bp = np.asarray(mjm.actuator_gainprm, dtype=np.float64).reshape(nu, -1), np.asarray(mjm.actuator_biasprm, dtype=np.float64).reshape(nu, -1)
    gt_, bt_ = np.asarray(mjm.actuator_gaintype).reshape(-1), np.asarray(mjm.actuator_biastype).reshape(-1)
    m.act_velfeedback = int(bool(((gt_ == 1) & (gp[:, 2] != 0.0)).any() or ((bt_ == 1) & (bp[:, 2] > 0.0)).any()))
  else:
    m.act_dof_max = 0
    m.act_velfeedback = 0
  m.tree_nvmax = int(tree_dofnum.max()) if len(roots) else 0
  # nv > 64: worlds whose rows each touch one kinematic tree are solved per (world, tree) by the register-resident kernels
  m.tree_solve = int(nv > 64 and m.ntree > 1 and m.tree_nvmax <= 64 and int(opt.solver) != types.SolverType.PGS)
  # size classes of the island kernels: islands of <= 32 dofs run with 32 lanes and ceil(dofs / 4) quarter-rows -- the widest one is a
  # single tree unless two trees together fit --, islands of 33..64 dofs with 64 lanes
  small = sorted(int(n) for n in tree_dofnum if n <= 32)
  widest = max(small) if small else 0
  if len(small) >= 2 and small[0] + small[1] <= 32:
    widest = 32
  m.isl_nv4 = (widest + 3) // 4
  m.isl_wide = int(nv > 32)
  nw = max((nv + 31) // 32, 1)
  dofmask = np.zeros((nbody, nw), dtype=np.uint32)
  for b in range(nbody):  # io.py:536-549 body_isdofancestor as bit masks
    dsel = lastdof[b]
    while dsel >= 0:
      dofmask[b, dsel >> 5] |= np.uint32(1 << (dsel & 31))
      dsel = dof_parent[dsel]
  ddepth = np.zeros(nv, dtype=np.int32)
  for i in range(nv):
    ddepth[i] = 0 if dof_parent[i] < 0 else ddepth[dof_parent[i]] + 1
  dorder = np.argsort(ddepth, kind="stable").astype(np.int32)
  ndlevel = int(ddepth.max()) + 1 if nv else 0
  dleveladr = np.zeros(ndlevel + 1, dtype=np.int32)
  for l in range(ndlevel):
    dleveladr[l + 1] = dleveladr[l] + int(np.sum(ddepth == l))
  jnt_type, jnt_dofadr = _arr(mjm.jnt_type, np.int32), _arr(mjm.jnt_dofadr, np.int32)
  dof_jnt = _arr(mjm.dof_jntid, np.int32)
  grpadr = np.arange(nv, dtype=np.int32)
  for i in range(nv):
    j = dof_jnt[i]
    if jnt_type[j] == types.JointType.BALL:
      grpadr[i] = jnt_dofadr[j]
    elif jnt_type[j] == types.JointType.FREE and i - jnt_dofadr[j] >= 3:
      grpadr[i] = jnt_dofadr[j] + 3

  host = {}  # name -> numpy array in ABI dtype
  i32 = np.int32
  host.update(
    opt_timestep=o.timestep, opt_tolerance=o.tolerance, opt_ls_tolerance=o.ls_tolerance, opt_gravity=o.gravity,
    opt_impratio_invsqrt=o.impratio_invsqrt, opt_ccd_tolerance=o.ccd_tolerance, opt_magnetic=o.magnetic, stat_meaninertia=s.meaninertia,
    qpos0=_arr(mjm.qpos0, f32).reshape(1, -1), qpos_spring=_arr(mjm.qpos_spring, f32).reshape(1, -1),
    body_parentid=parent, body_rootid=_arr(mjm.body_rootid, i32), body_weldid=_arr(mjm.body_weldid, i32),
    body_jntnum=_arr(mjm.body_jntnum, i32), body_jntadr=_arr(mjm.body_jntadr, i32), body_dofnum=dofnum, body_dofadr=dofadr,
    body_lastdof=lastdof, body_mocapid=_arr(getattr(mjm, "body_mocapid", np.full(nbody, -1)), i32), body_subtreenum=subtreenum, body_tree=order, body_leveladr=leveladr, body_dofmask=dofmask,
    jnt_type=jnt_type, jnt_qposadr=_arr(mjm.jnt_qposadr, i32), jnt_dofadr=jnt_dofadr, jnt_bodyid=_arr(mjm.jnt_bodyid, i32),
    jnt_limited=_arr(mjm.jnt_limited, i32),
    jnt_actfrclimited=_arr(getattr(mjm, "jnt_actfrclimited", np.zeros(njnt)), i32), jnt_actgravcomp=_arr(getattr(mjm, "jnt_actgravcomp", np.zeros(njnt)), i32),
    jnt_actfrcrange=_arr(getattr(mjm, "jnt_actfrcrange", np.zeros((njnt, 2))), f32).reshape(1, njnt, 2),
    dof_bodyid=_arr(mjm.dof_bodyid, i32), dof_jntid=dof_jnt, dof_parentid=dof_parent, dof_grpadr=grpadr, dof_tree=dorder,
    dof_leveladr=dleveladr, tree_dofadr=tree_dofadr, tree_dofnum=tree_dofnum, dof_treeid=dof_treeid, body_treeid=body_treeid,
    M_rownnz=_arr(mjm.M_rownnz, i32), M_rowadr=_arr(mjm.M_rowadr, i32), M_colind=_arr(mjm.M_colind, i32), M_dense=_m_dense(mjm, nv),
    geom_type=_arr(mjm.geom_type, i32), geom_condim=_arr(mjm.geom_condim, i32), geom_bodyid=_arr(mjm.geom_bodyid, i32),
    geom_priority=_arr(mjm.geom_priority, i32), geom_group=_arr(getattr(mjm, "geom_group", np.zeros(ngeom)), i32),
    geom_matid=_arr(getattr(mjm, "geom_matid", np.full(ngeom, -1)), i32), geom_rgba=_arr(getattr(mjm, "geom_rgba", np.tile([0.5, 0.5, 0.5, 1.0], (ngeom, 1))), f32).reshape(-1, 4),
    mat_rgba=_arr(getattr(mjm, "mat_rgba", np.zeros((0, 4))), f32).reshape(-1, 4), geom_dataid=_arr(getattr(mjm, "geom_dataid", np.full(ngeom, -1)), i32),
    mesh_vertadr=_arr(getattr(mjm, "mesh_vertadr", np.zeros(0)), i32), mesh_vertnum=_arr(getattr(mjm, "mesh_vertnum", np.zeros(0)), i32),
    mesh_vert=_arr(getattr(mjm, "mesh_vert", np.zeros((0, 3))), f32).reshape(-1, 3),
    sensor_type=_arr(getattr(mjm, "sensor_type", np.zeros(0)), i32), sensor_datatype=_arr(getattr(mjm, "sensor_datatype", np.zeros(0)), i32),
    sensor_objtype=_arr(getattr(mjm, "sensor_objtype", np.zeros(0)), i32), sensor_objid=_arr(getattr(mjm, "sensor_objid", np.zeros(0)), i32),
    sensor_reftype=_arr(getattr(mjm, "sensor_reftype", np.zeros(0)), i32), sensor_refid=_arr(getattr(mjm, "sensor_refid", np.zeros(0)), i32),
    sensor_dim=_arr(getattr(mjm, "sensor_dim", np.zeros(0)), i32), sensor_adr=_arr(getattr(mjm, "sensor_adr", np.zeros(0)), i32),
    sensor_cutoff=_arr(getattr(mjm, "sensor_cutoff", np.zeros(0)), f32),
    hfield_size=_arr(getattr(mjm, "hfield_size", np.zeros((0, 4))), f32).reshape(-1, 4), hfield_nrow=_arr(getattr(mjm, "hfield_nrow", np.zeros(0)), i32),
    hfield_ncol=_arr(getattr(mjm, "hfield_ncol", np.zeros(0)), i32), hfield_adr=_arr(getattr(mjm, "hfield_adr", np.zeros(0)), i32),
    hfield_data=_arr(getattr(mjm, "hfield_data", np.zeros(0)), f32),
    mesh_graphadr=_arr(getattr(mjm, "mesh_graphadr", np.full(int(getattr(mjm, "nmesh", 0)), -1)), i32), mesh_graph=_arr(getattr(mjm, "mesh_graph", np.zeros(0)), i32),
    mesh_polyadr=_arr(getattr(mjm, "mesh_polyadr", np.zeros(0)), i32), mesh_polynormal=_arr(getattr(mjm, "mesh_polynormal", np.zeros((0, 3))), f32).reshape(-1, 3),
    mesh_polyvertadr=_arr(getattr(mjm, "mesh_polyvertadr", np.zeros(0)), i32), mesh_polyvertnum=_arr(getattr(mjm, "mesh_polyvertnum", np.zeros(0)), i32),
    mesh_polyvert=_arr(getattr(mjm, "mesh_polyvert", np.zeros(0)), i32), mesh_polymapadr=_arr(getattr(mjm, "mesh_polymapadr", np.zeros(0)), i32),
    mesh_polymapnum=_arr(getattr(mjm, "mesh_polymapnum", np.zeros(0)), i32), mesh_polymap=_arr(getattr(mjm, "mesh_polymap", np.zeros(0)), i32), nxn_geom_pair=pairs, nxn_pairid=pairid, nxn_pairindex=_pair_index(ngeom, pairs),
    cull_geom=cull[0], cull_group=cull[1], cull_pair=cull[2], cull_list=cull[3],
    pair_dim=_arr(getattr(mjm, "pair_dim", np.zeros(0)), i32), pair_friction=_arr(getattr(mjm, "pair_friction", np.zeros((0, 5))), f32).reshape(-1, 5),
    pair_solref=_arr(getattr(mjm, "pair_solref", np.zeros((0, 2))), f32).reshape(-1, 2),
    pair_solreffriction=_arr(getattr(mjm, "pair_solreffriction", np.zeros((0, 2))), f32).reshape(-1, 2),
    pair_solimp=_arr(getattr(mjm, "pair_solimp", np.zeros((0, 5))), f32).reshape(-1, 5),
    pair_margin=_arr(getattr(mjm, "pair_margin", np.zeros(0)), f32), pair_gap=_arr(getattr(mjm, "pair_gap", np.zeros(0)), f32),
    site_bodyid=_arr(getattr(mjm, "site_bodyid", np.zeros(0)), i32),
    site_type=_arr(getattr(mjm, "site_type", np.full(int(getattr(mjm, "nsite", 0)), 2)), i32), site_size=_arr(getattr(mjm, "site_size", np.full((int(getattr(mjm, "nsite", 0)), 3), 0.005)), f32).reshape(-1, 3),
    actuator_dyntype=_arr(mjm.actuator_dyntype, i32), actuator_gaintype=_arr(mjm.actuator_gaintype, i32),
    actuator_biastype=_arr(mjm.actuator_biastype, i32), actuator_trnid=_arr(mjm.actuator_trnid, i32).reshape(-1, 2),
    actuator_actadr=_arr(mjm.actuator_actadr, i32), actuator_ctrllimited=_arr(mjm.actuator_ctrllimited, i32),
    actuator_forcelimited=_arr(mjm.actuator_forcelimited, i32), actuator_actlimited=_arr(mjm.actuator_actlimited, i32),
    eq_obj1id=_arr(getattr(mjm, "eq_obj1id", np.zeros(0)), i32), eq_obj2id=_arr(getattr(mjm, "eq_obj2id", np.zeros(0)), i32),
  )
  m.nsensor, m.nsensordata = int(host["sensor_type"].shape[0]), int(getattr(mjm, "nsensordata", 0))
  supported_sensors = set(mjcf_SENS.values())
  m.nsensor_acc = int(sum(int(t) in (0, 1, 4, 5, 22, 33, 34) for t in host["sensor_type"]))
  m.nsensor_energy = int(sum(int(t) in (43, 44) for t in host["sensor_type"]))
  m.nsensor_frc = int(sum(int(t) in (4, 5) for t in host["sensor_type"]))
  m.nsensor_subtree = int(sum(int(t) in (36, 37) for t in host["sensor_type"]))
  bad = [int(t) for t in host["sensor_type"] if int(t) not in supported_sensors]
  if bad:  # (sensors do not enter the dynamics: the model still loads, their sensordata slots stay zero)
    import warnings

    warnings.warn(f"sensor types {sorted(set(bad))} are not computed by this engine (csrc/sensor.hpp computes {sorted(supported_sensors)}): their sensordata entries are 0")
  m.nmeshvert = int(host["mesh_vert"].shape[0])
  m.nmeshpoly = int(host["mesh_polyvertnum"].shape[0])
  m.nmeshgraph = int(host["mesh_graph"].shape[0])
  m.nhfield, m.nhfielddata = int(host["hfield_nrow"].shape[0]), int(host["hfield_data"].shape[0])
  m.nmeshpolyvert, m.nmeshpolymap = int(host["mesh_polyvert"].shape[0]), int(host["mesh_polymap"].shape[0])
  # clip buffers of the multi-contact recovery: 2 * npolygonmax points (reference collision_convex.py:1226-1234)
  nboxmesh, nmeshmesh = sum(t == (6, 7) for t in ptypes), sum(t == (7, 7) for t in ptypes)
  m.npolygonmax = 4 if nboxbox > 0 else 0
  if not (int(opt.disableflags) & int(types.DisableBit.MULTICCD)) and nboxmesh + nmeshmesh > 0:
    m.npolygonmax = max(int(host["mesh_polyvertnum"].max()) if m.nmeshpoly else 0, 4 if nboxmesh else m.npolygonmax)
    # polygons around one vertex (collision_convex.py:1233 nmeshdegmax): sizes the normal / index buffers of the multi-contact recovery
    m.nmeshdegmax = max(int(host["mesh_polymapnum"].max()) if len(host["mesh_polymapnum"]) else 0, 3)
  m.nmesh = int(host["mesh_vertadr"].shape[0])
  m.nmat = int(host["mat_rgba"].shape[0])
  m.sleep_enabled = int(bool(int(opt.enableflags) & int(types.EnableBit.SLEEP)) and not (int(opt.disableflags) & int(types.DisableBit.ISLAND)))
  m.opt_sleep_tolerance = float(getattr(opt, "sleep_tolerance", 1e-4))
  host["tree_sleep_policy"] = _arr(getattr(mjm, "tree_sleep_policy", np.full(m.ntree, int(types.SleepPolicy.AUTO_ALLOWED))), i32)
  host["dof_length"] = _arr(getattr(mjm, "dof_length", np.ones(nv)), f32)
  if m.sleep_enabled and (host["tree_sleep_policy"] > int(types.SleepPolicy.AUTO_ALLOWED)).any():
    raise NotImplementedError("sleep policies NEVER / ALLOWED / INIT are not implemented (reference types.py:308).")
  neq = int(getattr(mjm, "neq", 0))
  host["eq_solref"] = _arr(getattr(mjm, "eq_solref", np.zeros((0, 2))), f32).reshape(1, neq, 2)
  host["eq_solimp"] = _arr(getattr(mjm, "eq_solimp", np.zeros((0, 5))), f32).reshape(1, neq, 5)
  host["eq_data"] = _arr(getattr(mjm, "eq_data", np.zeros((0, 11))), f32).reshape(1, neq, 11)
  m.eq_active0 = _arr(getattr(mjm, "eq_active0", np.zeros(0)), i32)
  vec = {"geom_aabb": 6, "body_pos": 3, "body_quat": 4, "body_ipos": 3, "body_iquat": 4, "body_inertia": 3, "body_invweight0": 2,
         "jnt_solref": 2, "jnt_solimp": 5, "jnt_pos": 3, "jnt_axis": 3, "jnt_range": 2, "jnt_actfrcrange": 2, "dof_solref": 2, "dof_solimp": 5,
         "geom_solref": 2, "geom_solimp": 5, "geom_size": 3, "geom_pos": 3, "geom_quat": 4, "geom_friction": 3,
         "site_pos": 3, "site_quat": 4, "actuator_dynprm": 10, "actuator_gainprm": 10, "actuator_biasprm": 10,
         "actuator_ctrlrange": 2, "actuator_forcerange": 2, "actuator_actrange": 2, "actuator_gear": 6}
  for name in _BATCHED_MODEL_FIELDS:
    if name in host:
      continue
    src = getattr(mjm, name, None)
    if src is None:
      if name in ("site_pos", "site_quat"):
        src = np.zeros((0, vec[name]))
      else:
        raise AttributeError(f"model is missing field {name}")
    a = _arr(src, f32)
    host[name] = a.reshape(1, -1, vec[name]) if name in vec else a.reshape(1, -1)

  batch_sizes = batch_sizes or {}
  for name, n in batch_sizes.items():
    key = {"timestep": "opt_timestep", "tolerance": "opt_tolerance", "gravity": "opt_gravity",
           "meaninertia": "stat_meaninertia"}.get(name, name)
    if key not in _BATCHED_MODEL_FIELDS:
      raise ValueError(f"{name} is not a batched ('*') Model field")
    host[key] = np.repeat(host[key][:1], int(n), axis=0)

  # ---- upload; public attribute names follow the reference Model ----
  for name, kind in _MODEL_PTR_FIELDS:
    da = DeviceArray.from_numpy(host[name])
    if name.startswith("opt_"):
      setattr(o, name[4:], da)
    elif name.startswith("stat_"):
      setattr(s, name[5:], da)
    else:
      setattr(m, name, da)
  m.opt, m.stat = o, s
  m.npair = int(len(pairs))
  m.ncullgeom, m.ncullgroup, m.ncullpair = int(len(cull[0])), int(len(cull[1])), int(len(cull[2]))
  m.nexplicit = nexplicit
  m.nxn_geom_pair_filtered = m.nxn_geom_pair
  m.nbodylevel, m.ndoflevel = nlevel, ndlevel
  m.nmaxcondim = int(max(condims)) if condims else 1
  m.nmaxpyramid = max(1, 2 * (m.nmaxcondim - 1))
  # condim of every contact the model can make (collision_core.py contact_params: the geom of higher priority decides, the larger condim
  # at equal priority; explicit pairs carry their own): only 1 and 3 -> CG may run the contact-basis kernel (csrc/solver_cgp.hpp)
  gc_, gp_ = np.asarray(mjm.geom_condim).astype(int), np.asarray(mjm.geom_priority).astype(int)
  pair_condims = set(int(gc_[a] if gp_[a] > gp_[b] else gc_[b] if gp_[b] > gp_[a] else max(gc_[a], gc_[b])) for a, b in pairs)
  if nexplicit:
    pair_condims |= set(int(c) for c in np.asarray(mjm.pair_dim))
  # ... and no friction-loss rows (their three-zone cost is outside that kernel: it would hand EVERY world to its fallback launch, which
  # solves flagged worlds one after the other -- ADVICE round 5); the kernel still defers a world that shows up with nf > 0 (a model whose
  # frictionloss was edited after put_model), so this flag is about speed, not correctness
  has_floss = bool(np.any(np.asarray(getattr(mjm, "dof_frictionloss", 0.0)) > 0)) or bool(np.any(np.asarray(getattr(mjm, "tendon_frictionloss", 0.0)) > 0))
  m.cg_basis = int(pair_condims <= {1, 3} and not has_floss)
  m._ccd_flags_at_put = int(opt.disableflags)
  m.key_qpos = _arr(getattr(mjm, "key_qpos", np.zeros((0, m.nq))), f32)
  m.key_qvel = _arr(getattr(mjm, "key_qvel", np.zeros((0, nv))), f32)
  m.key_ctrl = _arr(getattr(mjm, "key_ctrl", np.zeros((0, nu))), f32)
  m.key_mpos = _arr(getattr(mjm, "key_mpos", np.zeros((0, 3 * m.nmocap))), f32)
  m.key_mquat = _arr(getattr(mjm, "key_mquat", np.zeros((0, 4 * m.nmocap))), f32)
  m.key_act = _arr(getattr(mjm, "key_act", np.zeros((0, m.na))), f32)
  m.key_time = _arr(getattr(mjm, "key_time", np.zeros(0)), f32)
  m._dirty = True
  m._c = None
  return m


def _m_dense(mjm, nv):
  """[nv, 4 ceil(nv / 4)] address in the CSR M-structure of the dense entry (i, c) (symmetric), -1 where M has no entry."""
  nvr = 4 * ((nv + 3) // 4)
  t = np.full((max(nv, 1), max(nvr, 4)), -1, dtype=np.int32)
  rowadr, rownnz, colind = np.asarray(mjm.M_rowadr), np.asarray(mjm.M_rownnz), np.asarray(mjm.M_colind)
  for i in range(nv):
    for a in range(int(rownnz[i])):
      j = int(colind[rowadr[i] + a])
      t[i, j] = t[j, i] = rowadr[i] + a
  return t[:nv] if nv else t[:0]


def c_model(m: types.Model):
  """ctypes MjhModel for `m`, rebuilt only when a field was re-bound."""
  if m._c is not None and not m._dirty:
    return m._c
  c = _abi.CModel()
  for name, kind, ptr in _abi.MODEL_FIELDS:
    if ptr:
      src = getattr(m.opt, name[4:]) if name.startswith("opt_") else getattr(m.stat, name[5:]) if name.startswith("stat_") else getattr(m, name)
      setattr(c, name, src.ptr)
      if name in _BATCHED_MODEL_FIELDS:
        setattr(c, name + "_nb", src.shape[0])
    elif name.endswith("_nb"):
      continue
    elif name in ("integrator", "cone", "solver", "iterations", "ls_iterations", "disableflags", "enableflags", "broadphase", "broadphase_filter", "ccd_iterations"):
      if name == "disableflags":
        # the convex narrowphase's buffers (Model.nmeshdegmax / npolygonmax, Data.ws_ccd) were sized from these two bits at put_model /
        # make_data; the reference re-derives them at every call (collision_convex.py:1226, 1346-1366), this engine cannot: a change needs a new
        # put_model (kernels that branch on the live flags would run past buffers sized for the old ones)
        ccd_bits = int(types.DisableBit.NATIVECCD) | int(types.DisableBit.MULTICCD)
        if (int(m.opt.disableflags) ^ int(m._ccd_flags_at_put)) & ccd_bits:
          raise ValueError("opt.disableflags: NATIVECCD / MULTICCD changed after put_model -- the convex narrowphase's buffers were sized from them; call put_model again")
      setattr(c, name, int(getattr(m.opt, name)))
    elif name == "opt_sleep_tolerance":
      setattr(c, name, float(m.opt_sleep_tolerance))
    elif name == "heavy_colliders":
      bf_ = int(m.opt.broadphase_filter)
      if bf_ & ~0xF:
        raise ValueError(f"unknown broadphase filter bits {bf_:#x}")
      if int(m.opt.broadphase) not in (0, 1, 2):
        raise ValueError(f"unknown broadphase {int(m.opt.broadphase)}")
      if int(m.opt.broadphase) == 0 and m._convex_pairs and int(m.npair) > 0 and int(m.ncullpair) == 0:
        # (io.cull_tables declines models beyond its 16-bit geom / 24-bit pair packing -- put_model picks SAP for those; NXN
        # forced on one would fail every step with MJH_E_ARG in launch_ccd_pre: say so here)
        raise NotImplementedError(f"opt.broadphase = NXN with convex pairs needs the pair-group tables, which hold at most 65535 geoms and 2^24 pairs "
                                  f"(this model: {int(m.ngeom)} geoms, {int(m.npair)} pairs): use SAP")
      setattr(c, name, int(bool(m._heavy_pairs or bf_ != 3 or int(m.opt.broadphase) != 0 or m.sleep_enabled)))  # (sleep filter: heavy instantiation only)  # (the light instantiation hard-wires plane + sphere)
    else:
      setattr(c, name, int(getattr(m, name)))
  object.__setattr__(m, "_c", c)
  object.__setattr__(m, "_dirty", False)
  return c


def _collide_ccap(npair: int, concap: int) -> int:
  """csrc/collide.hpp collide_ccap: candidate capacity of one world's narrowphase."""
  cap = max(8 * concap, 512)
  return (min(npair, cap) + 3) // 4 * 4


def _ccd_handcap(nworld: int, ccap: int) -> int:
  """csrc/collide.hpp ccd_handcap: EPA entries k_ccd_gjk can hand to k_ccd_epa per step (32 per world on average, at least 4096 or all)."""
  return max(nworld * min(ccap, 32), min(nworld * ccap, 4096))


def _ccd_words(nworld: int, iterations: int, hfield: int, npolygonmax: int, nmeshdegmax: int, ccap: int, npair: int, handcap: Optional[int] = None) -> int:
  """Per-lane words of Data.ws_ccd [nworld, words, 32] such that it holds csrc/convex.hpp ccd_layout(...).total floats: per world the
  height-field prisms' polytopes, the per-candidate result cache and the candidate list; then the EPA hand-over
  records (CCD_HAND_WORDS = 64 each) and the multi-contact buffers of the EPA groups (sized from the model like the reference's,
  collision_convex.py:1346-1366)."""
  it = min(int(iterations), 64)
  poly = 8 * (5 + it) + 5 * (6 + 5 * it) + 24
  cache0 = (poly + 4 * 24 + 7 * 50 + 1) * 32 if hfield else 0
  cand = cache0 + ccap * 24
  bmask = (cand + ccap + 4 + 3) // 4 * 4  # k_broad_mask's bit mask over the pair list (64-pair granules)
  world_stride = (bmask + 2 * ((npair + 63) // 64) + 3) // 4 * 4
  handcap = _ccd_handcap(nworld, ccap) if handcap is None else int(handcap)
  mcw = 0  # (the multi-contact recovery works in LDS)
  hand = (world_stride * nworld + 8 + 2 * ((npair + 63) // 64) + 3) // 4 * 4  # (behind the counters and the convex-pair mask)
  total = hand + handcap * 64 + min(handcap + 7, 16384) * mcw
  return (total + 32 * nworld - 1) // (32 * nworld)


def contact_cap(nconmax: int) -> int:
  """Per-world contact capacity of the collision -> constraint hand-off buffer (twice the average budget)."""
  return int(min(max(2 * nconmax, 16), 256))


def _data_shapes(m, nworld, nconmax, njmax, naconmax, handcap=None):
  njmax_pad, nv_pad = _get_padded_sizes(m.nv, njmax)
  nb, nv, nq, nu, na, ng, nj, ns, nC = m.nbody, m.nv, m.nq, m.nu, m.na, m.ngeom, m.njnt, m.nsite, m.nC
  W = nworld
  sh = dict(
    time=(W,), qpos=(W, nq), qvel=(W, nv), act=(W, na), ctrl=(W, nu), qacc_warmstart=(W, nv), qfrc_applied=(W, nv),
    xfrc_applied=(W, nb, 6), mocap_pos=(W, m.nmocap, 3), mocap_quat=(W, m.nmocap, 4), xpos=(W, nb, 3), xquat=(W, nb, 4), xmat=(W, nb, 3, 3), xipos=(W, nb, 3), ximat=(W, nb, 3, 3),
    xanchor=(W, nj, 3), xaxis=(W, nj, 3), geom_xpos=(W, ng, 3), geom_xmat=(W, ng, 3, 3), site_xpos=(W, ns, 3),
    site_xmat=(W, ns, 3, 3), subtree_com=(W, nb, 3), cinert=(W, nb, 10), cdof=(W, nv, 6), crb=(W, nb, 10), M=(W, nC),
    qLD=(W, nC), qLDiagInv=(W, nv), actuator_length=(W, nu), actuator_moment=(W, nu), actuator_velocity=(W, nu),
    cvel=(W, nb, 6), cdof_dot=(W, nv, 6), qfrc_spring=(W, nv), qfrc_damper=(W, nv), qfrc_gravcomp=(W, nv),
    qfrc_passive=(W, nv), qfrc_bias=(W, nv), cacc=(W, nb, 6), cfrc_int=(W, nb, 6), act_dot=(W, na),
    actuator_force=(W, nu), qfrc_actuator=(W, nv), qfrc_smooth=(W, nv), qacc_smooth=(W, nv), qacc=(W, nv),
    qfrc_constraint=(W, nv), efc_Ma=(W, nv), solver_niter=(W,), ne=(W,), nf=(W,), nl=(W,), nefc=(W,), overflow=(W,),
    nacon=(1,), ncollision=(1,),
    contact_dist=(naconmax,), contact_pos=(naconmax, 3), contact_frame=(naconmax, 3, 3), contact_includemargin=(naconmax,),
    contact_friction=(naconmax, 5), contact_solref=(naconmax, 2), contact_solreffriction=(naconmax, 2),
    contact_solimp=(naconmax, 5), contact_dim=(naconmax,), contact_geom=(naconmax, 2),
    contact_efc_address=(naconmax, m.nmaxpyramid), contact_worldid=(naconmax,), contact_type=(naconmax,),
    contact_geomcollisionid=(naconmax,),
    efc_type=(W, njmax), efc_id=(W, njmax), efc_state=(W, njmax), efc_J=(W, njmax_pad, nv_pad), efc_pos=(W, njmax),
    efc_margin=(W, njmax), efc_D=(W, njmax), efc_vel=(W, njmax), efc_aref=(W, njmax), efc_frictionloss=(W, njmax),
    efc_force=(W, njmax), ws_ncon=(W,), ws_conadr=(W,), ws_ncollision=(W,), ws_efc_con=(W, njmax), ws_tree_rowadr=(W, (m.ntree + 1) if m.tree_solve else 0), ws_tree_rowmap=(W, njmax if m.tree_solve else 0),
    ws_isl_dofadr=(W, (m.ntree + 1) if m.tree_solve else 0), ws_isl_dofmap=(W, nv if m.tree_solve else 0), ws_isl_dofinv=(W, nv if m.tree_solve else 0), ws_nisland=(W,), ws_isl_flags=(W,), ws_isl_list=(3, W if m.tree_solve else 0), ws_isl_count=(4,),
    ws_separable=(W,), ws_order=(W,), ws_ccd=(W if m._convex_pairs else 0, _ccd_words(W, max(int(m.opt.ccd_iterations), int(m.epa_iterations)), m.nhfield, m.npolygonmax, m.nmeshdegmax, _collide_ccap(int(m.npair), contact_cap(nconmax)), int(m.npair), handcap), 32),
    tree_asleep=(W, m.ntree), tree_awake=(W, m.ntree), body_awake=(W, nb), body_awake_ind=(W, nb), dof_awake_ind=(W, nv), ntree_awake=(W,), nbody_awake=(W,),
    nv_awake=(W,), tree_island=(W, m.ntree), nisland=(W,), ws_iacc=(W if int(m.opt.integrator) == int(types.IntegratorType.IMPLICIT) else 0, nv), ws_pgsB=(W if _needs_pgs_big(m) else 0, njmax_pad, nv_pad), ws_sleep_J=(W if m.sleep_enabled else 0, njmax_pad, nv_pad), ws_sleep_warm=(W if m.sleep_enabled else 0, nv),
    ws_sleep_flag=(W,),
    sensordata=(W, m.nsensordata), energy=(W, 2), subtree_linvel=(W, nb, 3), subtree_angmom=(W, nb, 3), cfrc_ext=(W, nb, 6), eq_active=(W, m.neq), ws_rk=(W, nq + 3 * nv + 2 * na), ws_contact=(W, contact_cap(nconmax), 32),
  )
  return sh, njmax_pad, nv_pad


def _needs_pgs_big(m: types.Model) -> bool:
  """PGS with more than 64 dofs or elliptic cones runs the generic kernel (csrc/pgs_big.hpp), which keeps J M^-1 in Data.ws_pgsB."""
  return int(m.opt.solver) == int(types.SolverType.PGS) and (m.nv > 64 or int(m.opt.cone) == int(types.ConeType.ELLIPTIC))


def _alloc_data(m: types.Model, nworld, nconmax, njmax, naconmax, mjd=None, nvmax=None, nccdmax=None, naccdmax=None):
  if nvmax is None:
    nvmax = m.nv
  if nvmax < 0 or nvmax > m.nv:  # reference io.py:1731
    raise ValueError(f"nvmax ({nvmax}) must be in [0, nv ({m.nv})]")
  if nconmax is None:
    nconmax = _default_nconmax(None, mjd)  # at least what the host data already holds (reference io.py:1284-1311)
  if njmax is None:
    njmax = _default_njmax(None, mjd)
  if nconmax < 0:
    raise ValueError("nconmax must be >= 0")
  if njmax < 0:
    raise ValueError("njmax must be >= 0")
  if nworld < 1:
    raise ValueError("nworld must be >= 1")
  if naconmax is None:
    naconmax = nworld * nconmax
  if naconmax < 0:
    raise ValueError("naconmax must be >= 0")
  # nccdmax / naccdmax (reference io.py:1741-1753: CCD contacts per world / in total): the number of penetrating convex pairs k_ccd_gjk can
  # hand to k_ccd_epa per step (Data.nccdhand).  Pairs beyond it are dropped and flagged in Data.overflow (OVF_CCD = 16); WHICH pairs depends
  # on the order the wavefronts reserve their entries, so size it for the scene.  Default: 32 per world on average (csrc/collide.hpp ccd_handcap).
  if nccdmax is not None:
    if nccdmax < 0:
      raise ValueError("nccdmax must be >= 0")
    if nccdmax > nconmax:
      raise ValueError(f"nccdmax ({nccdmax}) must be <= nconmax ({nconmax})")
  if naccdmax is not None and naccdmax < 0:
    raise ValueError("naccdmax must be >= 0")
  handcap = None
  if m._convex_pairs and (naccdmax is not None or nccdmax is not None):
    handcap = max(int(naccdmax) if naccdmax is not None else int(nccdmax) * nworld, 1)
  shapes, njmax_pad, nv_pad = _data_shapes(m, nworld, nconmax, njmax, naconmax, handcap)
  d = types.Data()
  d.contact = types.Contact()
  d.efc = types.Constraint()
  d.contact._root = d
  d.efc._root = d
  for name, kind in _DATA_PTR_FIELDS:
    arr = DeviceArray.zeros(shapes[name], dtype=np.int32 if kind == "int" else np.float32)
    _set_data_field(d, name, arr)
  d.nworld, d.nconmax, d.naconmax, d.njmax, d.njmax_pad, d.nv_pad = nworld, nconmax, naconmax, njmax, njmax_pad, nv_pad
  d.ws_order.assign(np.arange(nworld, dtype=np.int32))
  d.sleep_pass = 0
  d.nvmax = int(nvmax)
  d.nsleepworld = shapes["ws_sleep_J"][0]
  d.npgsworld = shapes["ws_pgsB"][0]
  d.nimpworld = shapes["ws_iacc"][0]
  _reset_sleep(m, d, None)
  if m.neq:
    d.eq_active.assign(np.tile(m.eq_active0, (nworld, 1)))
  _reset_mocap(m, d, None)
  d.nmaxpyramid = m.nmaxpyramid
  d.nccdworld, d.nccdword = shapes["ws_ccd"][0], shapes["ws_ccd"][1]
  d.world_offset = 0
  d.concap = contact_cap(nconmax)
  d.nccdhand = (handcap if handcap is not None else _ccd_handcap(nworld, _collide_ccap(int(m.npair), d.concap))) if shapes["ws_ccd"][0] else 0
  d.njmax_nnz = njmax * m.nv
  d._c = None
  d._dirty = True
  return d


def _set_data_field(d, name, arr):
  if name.startswith("contact_"):
    setattr(d.contact, name[8:], arr)
  elif name.startswith("efc_"):
    setattr(d.efc, name[4:], arr)
  else:
    setattr(d, name, arr)


def _get_data_field(d, name):
  if name.startswith("contact_"):
    return getattr(d.contact, name[8:])
  if name.startswith("efc_"):
    return getattr(d.efc, name[4:])
  return getattr(d, name)


def c_data(d: types.Data):
  if d._c is not None and not d._dirty:
    return d._c
  c = _abi.CData()
  for name, kind, ptr in _abi.DATA_FIELDS:
    if ptr:
      setattr(c, name, _get_data_field(d, name).ptr)
    else:
      setattr(c, name, int(getattr(d, name)))
  object.__setattr__(d, "_c", c)
  object.__setattr__(d, "_dirty", False)
  return c


def _model_of(mjm):
  """put_data/make_data take the host model like the reference; cache its device Model."""
  m = getattr(mjm, "_mjh_model", None)
  if m is None:
    m = put_model(mjm)
    try:
      mjm._mjh_model = m
    except AttributeError:
      pass
  return m


def make_data(mjm, nworld: int = 1, nconmax: Optional[int] = None, nccdmax: Optional[int] = None,
              njmax: Optional[int] = None, njmax_nnz: Optional[int] = None, naconmax: Optional[int] = None,
              naccdmax: Optional[int] = None, nvmax: Optional[int] = None) -> types.Data:
  """Creates a data object on device (reference io.py:1680); state = qpos0."""
  m = mjm if isinstance(mjm, types.Model) else _model_of(mjm)
  d = _alloc_data(m, nworld, nconmax, njmax, naconmax, nvmax=nvmax, nccdmax=nccdmax, naccdmax=naccdmax)
  d.qpos.assign(np.tile(m.qpos0.numpy()[0], (nworld, 1)))
  return d


def put_data(mjm, mjd, nworld: int = 1, nconmax: Optional[int] = None, nccdmax: Optional[int] = None,
             njmax: Optional[int] = None, njmax_nnz: Optional[int] = None, naconmax: Optional[int] = None,
             naccdmax: Optional[int] = None, nvmax: Optional[int] = None) -> types.Data:
  """Moves data from host to a device (reference io.py:1890): the single host state is tiled nworld times."""
  m = mjm if isinstance(mjm, types.Model) else _model_of(mjm)
  d = _alloc_data(m, nworld, nconmax, njmax, naconmax, mjd, nvmax=nvmax, nccdmax=nccdmax, naccdmax=naccdmax)
  if m.neq and getattr(mjd, "eq_active", None) is not None:
    d.eq_active.assign(np.tile(np.asarray(mjd.eq_active, dtype=np.int32).reshape(1, -1), (nworld, 1)))
  for name in ("qpos", "qvel", "act", "ctrl", "qacc_warmstart", "qfrc_applied", "xfrc_applied", "mocap_pos", "mocap_quat"):
    if not hasattr(mjd, name):
      continue
    src = np.asarray(getattr(mjd, name), dtype=np.float32)
    dst = getattr(d, name)
    if dst.size:
      dst.assign(np.broadcast_to(src.reshape((1,) + dst.shape[1:]), dst.shape))
  d.time.fill_(float(mjd.time))
  if m.ntree and getattr(mjd, "tree_asleep", None) is not None:  # reference io.py:2010-2012, 2138-2140: the host sleep state, tiled
    asleep = np.asarray(mjd.tree_asleep, dtype=np.int32).reshape(1, -1)
    d.tree_asleep.assign(np.tile(asleep, (nworld, 1)))
    d.tree_awake.assign(np.tile((asleep < 0).astype(np.int32), (nworld, 1)))
    if getattr(mjd, "body_awake", None) is not None:
      d.body_awake.assign(np.tile(np.asarray(mjd.body_awake, dtype=np.int32).reshape(1, -1), (nworld, 1)))
    elif d.body_awake.size and (asleep >= 0).any():
      # (a host MjData without body_awake, e.g. mjcf.MjData with `tree_asleep[:] = arange(ntree)`: derive the tables the first wake /
      # update_sleep stage would produce, sleep.py:171-215 -- the reference's host object carries MuJoCo's own)
      treeid, mocap = m.body_treeid.numpy(), m.body_mocapid.numpy()[m.body_rootid.numpy()]  # (a body welded under a mocap body is posed by it: the root's mocap id, as csrc/sleep.hpp)
      ba = np.where(treeid >= 0, np.where(asleep[0][np.maximum(treeid, 0)] < 0, int(types.SleepState.AWAKE), int(types.SleepState.ASLEEP)),
                    np.where(mocap >= 0, int(types.SleepState.AWAKE), int(types.SleepState.STATIC))).astype(np.int32)
      d.body_awake.assign(np.tile(ba, (nworld, 1)))
      bind = np.concatenate([np.flatnonzero(ba != int(types.SleepState.ASLEEP)), np.zeros(int((ba == int(types.SleepState.ASLEEP)).sum()), dtype=np.int64)]).astype(np.int32)
      d.body_awake_ind.assign(np.tile(bind, (nworld, 1)))
      d.nbody_awake.fill_(int((ba != int(types.SleepState.ASLEEP)).sum()))
      dof_awake = asleep[0][m.dof_treeid.numpy()] < 0
      dind = np.concatenate([np.flatnonzero(dof_awake), np.zeros(int((~dof_awake).sum()), dtype=np.int64)]).astype(np.int32)
      d.dof_awake_ind.assign(np.tile(dind, (nworld, 1)))
      d.nv_awake.fill_(int(dof_awake.sum()))
      d.ntree_awake.fill_(int((asleep < 0).sum()))
  return d


def get_data_into(result, mjm, d: types.Data, world_id: int = 0):
  """Gets data from a device into an existing host MjData-like object (reference io.py:2184)."""
  w = world_id
  for name in ("qpos", "qvel", "act", "ctrl", "qacc_warmstart", "qfrc_applied", "xfrc_applied", "mocap_pos", "mocap_quat", "qacc", "xpos", "xquat",
               "xmat", "xipos", "ximat", "xanchor", "xaxis", "geom_xpos", "geom_xmat", "subtree_com", "cinert", "cdof", "crb",
               "qLD", "qLDiagInv", "cvel", "cdof_dot", "qfrc_bias", "qfrc_passive", "qfrc_spring", "qfrc_damper",
               "qfrc_actuator", "qfrc_smooth", "qacc_smooth", "qfrc_constraint", "actuator_force", "actuator_length",
               "actuator_velocity", "cacc", "cfrc_int"):
    arr = getattr(d, name).numpy()[w]
    cur = getattr(result, name, None)
    if isinstance(cur, np.ndarray) and cur.size == arr.size:
      cur[...] = arr.reshape(cur.shape)
    else:
      setattr(result, name, arr.astype(np.float64))
  result.qM = d.M.numpy()[w].astype(np.float64)
  result.time = float(d.time.numpy()[w])
  # contacts of this world, in the engine's deterministic per-world order
  ncon, adr = int(d.ws_ncon.numpy()[w]), int(d.ws_conadr.numpy()[w])
  result.ncon = ncon
  sl = slice(adr, adr + ncon)
  result.contact_dist = d.contact.dist.numpy()[sl].astype(np.float64)
  result.contact_pos = d.contact.pos.numpy()[sl].astype(np.float64)
  result.contact_frame = d.contact.frame.numpy()[sl].reshape(ncon, 9).astype(np.float64)
  result.contact_geom = d.contact.geom.numpy()[sl]
  result.contact_dim = d.contact.dim.numpy()[sl]
  result.contact_efc_address = d.contact.efc_address.numpy()[sl]
  nefc = min(int(d.nefc.numpy()[w]), d.njmax)
  result.nefc, result.ne, result.nf, result.nl = nefc, int(d.ne.numpy()[w]), int(d.nf.numpy()[w]), int(d.nl.numpy()[w])
  nv = d.qvel.shape[1]
  result.efc_J = d.efc.J.numpy()[w, :nefc, :nv].astype(np.float64)
  for name in ("pos", "margin", "D", "vel", "aref", "frictionloss", "force"):
    setattr(result, "efc_" + name, getattr(d.efc, name).numpy()[w, :nefc].astype(np.float64))
  for name in ("type", "id", "state"):
    setattr(result, "efc_" + name, getattr(d.efc, name).numpy()[w, :nefc])
  result.solver_niter = np.array([int(d.solver_niter.numpy()[w])])
  for name in ("tree_asleep", "tree_awake", "body_awake"):  # reference io.py:2409-2412
    setattr(result, name, getattr(d, name).numpy()[w].copy())
  result.sensordata = d.sensordata.numpy()[w].astype(np.float64)  # reference io.py: sensordata / energy
  result.energy = d.energy.numpy()[w].astype(np.float64)


def _state_fields(m, d, sig):
  """(tensor viewed as [nworld, size]) of the components selected by sig, in bit order (reference support.py:674-960)."""
  if sig < 0 or sig >= (1 << int(types.State.NSTATE)):
    raise ValueError(f"invalid state signature {sig} >= 2^mjNSTATE")
  S = types.State
  fields = [(S.TIME, d.time, 1), (S.QPOS, d.qpos, m.nq), (S.QVEL, d.qvel, m.nv), (S.ACT, d.act, m.na), (S.WARMSTART, d.qacc_warmstart, m.nv),
            (S.CTRL, d.ctrl, m.nu), (S.QFRC_APPLIED, d.qfrc_applied, m.nv), (S.XFRC_APPLIED, d.xfrc_applied, 6 * m.nbody), (S.EQ_ACTIVE, d.eq_active, m.neq),
            (S.MOCAP_POS, d.mocap_pos, 3 * m.nmocap), (S.MOCAP_QUAT, d.mocap_quat, 4 * m.nmocap)]
  return [(arr, n) for bit, arr, n in fields if (sig & int(bit)) and n > 0]


def state_size(m: types.Model, sig: int) -> int:
  """Number of floats of the state components selected by sig (reference mj_stateSize)."""
  S = types.State
  sizes = {S.TIME: 1, S.QPOS: m.nq, S.QVEL: m.nv, S.ACT: m.na, S.WARMSTART: m.nv, S.CTRL: m.nu, S.QFRC_APPLIED: m.nv, S.XFRC_APPLIED: 6 * m.nbody,
           S.EQ_ACTIVE: m.neq, S.MOCAP_POS: 3 * m.nmocap, S.MOCAP_QUAT: 4 * m.nmocap}
  return int(sum(n for bit, n in sizes.items() if sig & int(bit)))


def get_state(m: types.Model, d: types.Data, state: DeviceArray, sig: int, active=None):
  """Copies the concatenated state components selected by sig (types.State bits) from Data into state [nworld, state_size(m, sig)]
  (reference support.get_state, support.py:674); `active` is an optional per-world mask.  Device-to-device copies."""
  import torch

  if tuple(state.shape) != (d.nworld, state_size(m, int(sig))):  # (checked before anything is written)
    raise ValueError(f"state must have shape ({d.nworld}, {state_size(m, int(sig))})")
  mask = None if active is None else torch.as_tensor(np.asarray(active.numpy() if hasattr(active, "numpy") else active, dtype=bool), device=state.t.device)
  adr = 0
  for arr, n in _state_fields(m, d, int(sig)):
    src = arr.t.reshape(d.nworld, n).to(state.t.dtype)
    dst = state.t[:, adr : adr + n]
    if mask is None:
      dst.copy_(src)
    else:
      dst[mask] = src[mask]
    adr += n


def set_state(m: types.Model, d: types.Data, state: DeviceArray, sig: int, active=None):
  """Copies the concatenated state components selected by sig from state into Data (reference support.set_state, support.py:829)."""
  import torch

  if tuple(state.shape) != (d.nworld, state_size(m, int(sig))):
    raise ValueError(f"state must have shape ({d.nworld}, {state_size(m, int(sig))})")
  mask = None if active is None else torch.as_tensor(np.asarray(active.numpy() if hasattr(active, "numpy") else active, dtype=bool), device=state.t.device)
  adr = 0
  for arr, n in _state_fields(m, d, int(sig)):
    dst = arr.t.reshape(d.nworld, n)
    src = state.t[:, adr : adr + n].to(dst.dtype)
    if mask is None:
      dst.copy_(src)
    else:
      dst[mask] = src[mask]
    adr += n


def reset_data(m: types.Model, d: types.Data, reset=None):
  """Resets data to qpos0 (reference io.py:2435); `reset` is an optional per-world bool mask."""
  mask = None if reset is None else np.asarray(reset.numpy() if hasattr(reset, "numpy") else reset, dtype=bool)
  _reset_state(m, d, np.tile(m.qpos0.numpy()[:1], (d.nworld, 1)) if m.qpos0.shape[0] == 1 else m.qpos0.numpy()[np.arange(d.nworld) % m.qpos0.shape[0]],
               None, None, None, 0.0, mask)


def reset_data_keyframe(m: types.Model, d: types.Data, key: int, reset=None):
  """Resets data to keyframe `key` (reference io.py:2797)."""
  if key < 0 or key >= m.nkey:
    raise ValueError(f"keyframe {key} out of range [0, {m.nkey})")
  mask = None if reset is None else np.asarray(reset.numpy() if hasattr(reset, "numpy") else reset, dtype=bool)
  W = d.nworld
  _reset_state(m, d, np.tile(m.key_qpos[key], (W, 1)), np.tile(m.key_qvel[key], (W, 1)),
               np.tile(m.key_act[key], (W, 1)) if m.na else None, np.tile(m.key_ctrl[key], (W, 1)) if m.nu else None,
               float(m.key_time[key]), mask)
  if m.nmocap and getattr(m, "key_mpos", None) is not None and len(m.key_mpos) > key:  # keyframe mocap poses (io.py:2855)
    for name, src, width in (("mocap_pos", m.key_mpos, 3), ("mocap_quat", m.key_mquat, 4)):
      val = np.tile(np.asarray(src[key], dtype=np.float32).reshape(1, m.nmocap, width), (W, 1, 1))
      dst = getattr(d, name)
      if mask is None:
        dst.assign(val)
      else:
        cur = dst.numpy().copy()
        cur[mask] = val[mask]
        dst.assign(cur)


def _reset_mocap(m, d, mask):
  """mocap_pos / mocap_quat = the model pose of the mocap bodies (reference io.py:2552 reset_mocap)."""
  if not m.nmocap:
    return
  mid = m.body_mocapid.numpy()
  order = np.argsort(mid[mid >= 0])
  bodies = np.flatnonzero(mid >= 0)[order]
  for name, src in (("mocap_pos", m.body_pos), ("mocap_quat", m.body_quat)):
    val = src.numpy()[np.arange(d.nworld) % src.shape[0]][:, bodies]
    dst = getattr(d, name)
    if mask is None:
      dst.assign(val)
    else:
      cur = dst.numpy().copy()
      cur[mask] = val[mask]
      dst.assign(cur)


def _reset_sleep(m, d, mask):
  """Every tree fully awake (reference io.py:2637-2670 reset_sleep; make_data io.py:1869-1871)."""
  nt, nb, nv = m.ntree, m.nbody, m.nv
  treeid = m.body_treeid.numpy()
  mocap = m.body_mocapid.numpy()[m.body_rootid.numpy()]  # (the root's mocap id, as csrc/sleep.hpp: bodies welded under a mocap body move with it)
  body_awake = np.where(treeid >= 0, int(types.SleepState.AWAKE), np.where(mocap >= 0, int(types.SleepState.AWAKE), int(types.SleepState.STATIC))).astype(np.int32)
  vals = dict(tree_asleep=np.full((d.nworld, nt), -(1 + types.MJ_MINAWAKE), np.int32), tree_awake=np.ones((d.nworld, nt), np.int32),
              body_awake=np.tile(body_awake, (d.nworld, 1)), body_awake_ind=np.tile(np.arange(nb, dtype=np.int32), (d.nworld, 1)),
              dof_awake_ind=np.tile(np.arange(nv, dtype=np.int32), (d.nworld, 1)), ntree_awake=np.full(d.nworld, nt, np.int32),
              nbody_awake=np.full(d.nworld, nb, np.int32), nv_awake=np.full(d.nworld, nv, np.int32), tree_island=np.full((d.nworld, nt), -1, np.int32),
              nisland=np.zeros(d.nworld, np.int32))
  for name, val in vals.items():
    dst = getattr(d, name)
    if dst.size == 0:
      continue
    if mask is not None:
      cur = dst.numpy().copy()
      cur[mask] = val[mask]
      val = cur
    dst.assign(val)


def _reset_state(m, d, qpos, qvel, act, ctrl, time, mask):
  def put(dst, val):
    if dst.size == 0:
      return
    if val is None:
      val = np.zeros(dst.shape, dtype=np.float32)
    if mask is None:
      dst.assign(val)
    else:
      cur = dst.numpy().copy()
      cur[mask] = np.asarray(val, dtype=cur.dtype)[mask]
      dst.assign(cur)

  put(d.qpos, qpos)
  put(d.qvel, qvel)
  put(d.act, act)
  put(d.ctrl, ctrl)
  put(d.qacc_warmstart, None)
  put(d.qacc, None)
  put(d.qfrc_applied, None)
  put(d.xfrc_applied, None)
  put(d.time, np.full(d.time.shape, time, dtype=np.float32))
  _reset_mocap(m, d, mask)
  _reset_sleep(m, d, mask)
  if m.neq:
    put(d.eq_active, np.tile(m.eq_active0, (d.nworld, 1)))
  if mask is None:
    for name in ("nefc", "ne", "nf", "nl", "solver_niter", "overflow", "nacon", "ncollision", "ws_ncon", "ws_conadr", "ws_ncollision"):
      getattr(d, name).zero_()
  else:  # per-world counters of the reset worlds (the global nacon / ncollision are recomputed by the next collision)
    for name in ("nefc", "ne", "nf", "nl", "solver_niter", "overflow", "ws_ncon", "ws_ncollision"):
      dst = getattr(d, name)
      cur = dst.numpy().copy()
      cur[mask] = 0
      dst.assign(cur)


_ENUMS = {"solver": types.SolverType, "integrator": types.IntegratorType, "cone": types.ConeType}


def override_model(model, overrides):
  """Overrides model parameters, e.g. {"opt.solver": "cg"} or ["opt.iterations=10"] (reference io.py:2933).

  Works on the host model (MjModel-like, before put_model) and on a device Model.
  """
  if isinstance(overrides, (list, tuple)):
    ov = {}
    for item in overrides:
      k, v = item.split("=", 1)
      ov[k.strip()] = v.strip()
    overrides = ov
  for key, val in overrides.items():
    obj = model
    parts = key.split(".")
    for p in parts[:-1]:
      obj = getattr(obj, p)
    attr = parts[-1]
    if not hasattr(obj, attr):
      raise ValueError(f"Unrecognized model field: {key}")
    if attr in _ENUMS and isinstance(val, str) and not val.lstrip("-").isdigit():
      val = int(_ENUMS[attr][val.upper()])
    elif attr in ("disableflags", "enableflags") and isinstance(val, str) and not val.lstrip("-").isdigit():
      enum_cls = types.DisableBit if attr == "disableflags" else types.EnableBit
      bits = 0
      for tok in val.split("|"):
        bits |= int(enum_cls[tok.strip().upper()])
      val = bits
    cur = getattr(obj, attr)
    if isinstance(cur, DeviceArray):
      a = np.asarray(val if not isinstance(val, str) else [float(x) for x in val.replace(",", " ").split()], dtype=np.float32)
      if attr == "tolerance":
        a = np.maximum(a, 1e-6)
      cur.assign(np.broadcast_to(a.reshape(-1) if a.ndim else a, cur.shape) if a.size == 1 else a.reshape(cur.shape))
    elif isinstance(cur, (int, np.integer)):
      setattr(obj, attr, int(val))
    elif isinstance(cur, float):
      setattr(obj, attr, float(val))
    elif isinstance(cur, np.ndarray):
      a = np.asarray(val if not isinstance(val, str) else [float(x) for x in val.replace(",", " ").split()], dtype=cur.dtype)
      cur[...] = a.reshape(cur.shape) if a.size == cur.size else a
    else:
      setattr(obj, attr, val)
    if isinstance(model, types.Model) and attr in ("solver", "integrator", "cone"):
      put = {"solver": (types.SolverType.PGS, types.SolverType.CG, types.SolverType.NEWTON), "integrator": (types.IntegratorType.EULER, types.IntegratorType.RK4, types.IntegratorType.IMPLICITFAST),
             "cone": (types.ConeType.PYRAMIDAL, types.ConeType.ELLIPTIC)}[attr]
      if int(getattr(obj, attr)) not in put:
        raise NotImplementedError(f"unsupported {attr} {val}")
    if hasattr(model, "_mjh_model"):
      try:
        del model._mjh_model
      except AttributeError:
        pass


def load_trajectory(npz_path, mjm, mjd) -> np.ndarray:
  """Control sequence of an NPZ recording resampled onto the model timestep, zero-order hold (reference io.py:3067).

  The file holds `ctrl` [n, nu] and `times` [n] (one timestamp per control; the last control is held for the previous
  interval, or one model timestep if there is only one) or [n + 1] (interval boundaries).  Optional `qpos` [1, nq] /
  `qvel` [1, nv] set the initial state of `mjd`.  Returns ctrl [nstep, nu] with nstep = round(duration / timestep);
  control intervals shorter than a timestep may be skipped.
  """
  z = np.load(npz_path)
  ctrl, times = np.asarray(z["ctrl"]), np.asarray(z["times"])
  if ctrl.ndim != 2 or ctrl.shape[0] == 0:
    raise ValueError(f"ctrl must have shape (nstep, nu) with nstep > 0, got {ctrl.shape}")
  if ctrl.shape[1] != mjm.nu:
    raise ValueError(f"ctrl shape {ctrl.shape} does not match model nu={mjm.nu}")
  n = ctrl.shape[0]
  if times.ndim != 1 or times.shape[0] not in (n, n + 1):
    raise ValueError(f"times shape {times.shape} must contain {n} or {n + 1} timestamps")
  if not np.isfinite(times).all():
    raise ValueError("times must be finite")
  if (np.diff(times) <= 0).any():
    raise ValueError("times must be strictly increasing")
  for name, size in (("qpos", mjm.nq), ("qvel", mjm.nv)):
    if name in z.files and z[name].ndim == 2 and z[name].shape[1] == size:
      getattr(mjd, name)[:] = z[name][0]
  dt = float(mjm.opt.timestep)
  if times.shape[0] == n:  # close the last interval
    hold = times[-1] - times[-2] if n > 1 else dt
    times = np.concatenate([times, [times[-1] + hold]])
  nstep = int(np.round((times[-1] - times[0]) / dt))
  sample = times[0] + (np.arange(nstep) + 1e-7) * dt  # nudged off the boundaries so that equal timestamps hold the new control
  return ctrl[np.searchsorted(times, sample, side="right") - 1]
