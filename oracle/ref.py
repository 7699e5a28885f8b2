"""ctypes front-end of the float64 CPU oracle (oracle/mjref.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package `mujoco_warp_amd` never does.  Pin status (externally pinned for the G1 chain since round 3): see oracle/mjref.h.
"""

import ctypes
import os
import re
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_DIR, "libmjref.so")
_LIB32_PATH = os.path.join(_DIR, "libmjref32.so")  # the same restatement in float32: the float32 FLOOR of the algorithm, not an oracle


def build(force=False):
  src = [os.path.join(_DIR, f) for f in ("mjref.c", "mjref.h", "ccd.c")]
  for path in (_LIB_PATH, _LIB32_PATH):
    if force or not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in src):
      subprocess.check_call(["make", "-C", _DIR, "-B", os.path.basename(path)], stdout=subprocess.DEVNULL)
  return _LIB_PATH


_NATIVE_FLAGS = ["-O3", "-march=native", "-fPIC", "-std=gnu99", "-Wno-unused-function", "-fno-fast-math"]


def use_native_build():
  """bench.py's cpu_baseline leg only (the stand-in for MuJoCo C is compiled `-O3 -march=native`): compile the float64 restatement
  for THIS host's cores into oracle/_native/ and make it the library the next RefSim loads.  Not the oracle of the parity tests (those keep
  the -O2 -ffp-contract=off build: bit-reproducible across hosts).  Returns the flags used, for the bench line."""
  global _LIB_PATH
  out = os.path.join(_DIR, "_native", "libmjref_native.so")
  os.makedirs(os.path.dirname(out), exist_ok=True)
  src = os.path.join(_DIR, "mjref.c")
  if not os.path.exists(out) or os.path.getmtime(src) > os.path.getmtime(out):
    tmp = out + f".tmp{os.getpid()}"
    subprocess.check_call(["gcc", *_NATIVE_FLAGS, "-shared", "-o", tmp, src, "-lm"], cwd=_DIR)
    os.replace(tmp, out)
  if _F64._lib is None:
    _LIB_PATH = out
  return "gcc " + " ".join(_NATIVE_FLAGS)


def _parse_struct(header, name):
  body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), header, re.S).group(1)
  fields = []
  for line in body.splitlines():
    line = line.strip()
    mm = re.match(r"(int|double)(\*?)\s+(\w+);", line)
    if mm:
      fields.append((mm.group(3), mm.group(1), bool(mm.group(2))))
  return fields


_HEADER = open(os.path.join(_DIR, "mjref.h")).read()
_MODEL_FIELDS = _parse_struct(_HEADER, "RefModel")
_DATA_FIELDS = _parse_struct(_HEADER, "RefData")


class _Real:
  """One precision of the library: ctypes mirror of the structs (every `double` of the header is `float` in the float32 build),
  numpy dtype, loaded library."""

  def __init__(self, f32):
    self.f32 = bool(f32)
    self.c_real = ctypes.c_float if f32 else ctypes.c_double
    self.np_real = np.float32 if f32 else np.float64
    c_real = self.c_real

    def ctype(kind, ptr):
      base = ctypes.c_int if kind == "int" else c_real
      return ctypes.POINTER(base) if ptr else base

    self.ctype = ctype
    self.CRefModel = type("CRefModel", (ctypes.Structure,), {"_fields_": [(n, ctype(k, p)) for n, k, p in _MODEL_FIELDS]})
    self.CRefData = type("CRefData", (ctypes.Structure,), {"_fields_": [(n, ctype(k, p)) for n, k, p in _DATA_FIELDS]})
    self._lib = None

  def ptr(self, a):
    return a.ctypes.data_as(ctypes.POINTER(self.c_real))

  def arr(self, x):
    return np.ascontiguousarray(x, dtype=self.np_real)

  def lib(self):
    if self._lib is None:
      build()
      L = ctypes.CDLL(_LIB32_PATH if self.f32 else _LIB_PATH)
      mp, dp = ctypes.POINTER(self.CRefModel), ctypes.POINTER(self.CRefData)
      for fn in ("kinematics", "com_pos", "crb", "factor_m", "collision", "make_constraint", "transmission", "com_vel",
                 "passive", "rne", "fwd_position", "fwd_velocity", "fwd_actuation", "fwd_acceleration", "solve",
                 "forward", "euler", "implicitfast", "implicit", "step"):
        f = getattr(L, "ref_" + fn)
        f.argtypes = [mp, dp]
        f.restype = None
      R = self.c_real
      dptr = ctypes.POINTER(R)
      L.ref_solve_m.argtypes = [mp, dp, dptr, dptr]
      L.ref_mul_m.argtypes = [mp, dp, dptr, dptr]
      L.ref_deriv_rne_vel.argtypes = [mp, dp, dptr]
      L.ref_deriv_rne_vel.restype = None
      L.ref_ctrl_noise.argtypes = [mp, dp, dptr, ctypes.c_int, ctypes.c_int, R, R]
      L.ref_halton.argtypes = [ctypes.c_int, ctypes.c_int]
      L.ref_halton.restype = R
      L.ref_rollout.argtypes = [mp, dp, ctypes.c_int, ctypes.c_int, R, R, dptr, dptr]
      L.ref_rollout.restype = ctypes.c_int
      L.ref_closest_segment_to_segment_points.argtypes = [dptr] * 6
      L.ref_closest_segment_to_segment_points.restype = None
      for fn in (L.ref_upper_tri_index, L.ref_upper_trid_index):
        fn.argtypes = [ctypes.c_int] * 3
        fn.restype = ctypes.c_int
      L.ref_ccd.argtypes = [ctypes.c_int, dptr, dptr, dptr, ctypes.c_int, dptr, dptr, dptr, R, R, R, ctypes.c_int, ctypes.c_int, dptr, dptr]
      L.ref_ccd.restype = ctypes.c_int
      self._lib = L
    return self._lib


_F64, _F32 = _Real(False), _Real(True)
CRefModel, CRefData = _F64.CRefModel, _F64.CRefData


def lib():
  return _F64.lib()


def ccd(type1, pos1, mat1, size1, type2, pos2, mat2, size2, margin=0.0, tolerance=1e-6, cutoff=1e30, iterations=35, multiccd=False, vert1=None, vert2=None):
  """GJK / EPA on two posed primitive convex geoms (reference collision_gjk.py:2529 `ccd`, called like its test harness
  collision_gjk_test.py:36-300).  Returns (dist, ncon, x1, x2, witness pairs [ncon, 2, 3])."""
  def arr(x, n):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1))
    assert a.size == n
    return a
  p1, m1, s1, p2, m2, s2 = arr(pos1, 3), arr(mat1, 9), arr(size1, 3), arr(pos2, 3), arr(mat2, 9), arr(size2, 3)
  out, wit = np.zeros(9), np.zeros(48)
  dp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
  v1 = np.ascontiguousarray(np.asarray(vert1 if vert1 is not None else np.zeros((0, 3)), dtype=np.float64).reshape(-1, 3))  # mesh vertices, geom frame
  v2 = np.ascontiguousarray(np.asarray(vert2 if vert2 is not None else np.zeros((0, 3)), dtype=np.float64).reshape(-1, 3))
  fn = lib().ref_ccd_mesh
  fn.restype = ctypes.c_int
  n = fn(int(type1), dp(p1), dp(m1), dp(s1), dp(v1) if len(v1) else None, len(v1), int(type2), dp(p2), dp(m2), dp(s2), dp(v2) if len(v2) else None, len(v2),
         ctypes.c_double(margin), ctypes.c_double(tolerance), ctypes.c_double(cutoff), int(iterations), int(bool(multiccd)), dp(out), dp(wit))
  return float(out[0]), n, out[1:4].copy(), out[4:7].copy(), wit.reshape(8, 2, 3)[: max(n, 0)].copy()


def _epa_iterations(mjm, pairs):
  """EPA iteration cap (reference collision_convex.py:1209-1223): 16 when every convex-class pair of the model is box-box."""
  convex = {(2, 4), (3, 4), (3, 5), (4, 4), (4, 5), (4, 6), (5, 5), (5, 6), (2, 7), (3, 7), (4, 7), (5, 7), (6, 7), (7, 7)}
  nativeccd_off = bool(int(mjm.opt.disableflags) & (1 << 17))
  gt = np.asarray(mjm.geom_type)
  nbb = nother = 0
  for a, b in np.asarray(pairs).reshape(-1, 2):
    t = (int(min(gt[a], gt[b])), int(max(gt[a], gt[b])))
    if t == (6, 6) and not nativeccd_off:
      nbb += 1
    elif t in convex:
      nother += 1
  return 16 if (nbb > 0 and nother == 0) else int(getattr(mjm.opt, "ccd_iterations", 35))


def npolygonmax(mjm, pairs):
  """Vertices per clip polygon the multi-contact buffers are sized for (reference collision_convex.py:1226-1234)."""
  gt = np.asarray(mjm.geom_type)
  types = [(int(min(gt[a], gt[b])), int(max(gt[a], gt[b]))) for a, b in np.asarray(pairs).reshape(-1, 2)]
  nativeccd_off = bool(int(mjm.opt.disableflags) & (1 << 17))
  nboxbox = 0 if nativeccd_off else sum(t == (6, 6) for t in types)
  nboxmesh, nmeshmesh = sum(t == (6, 7) for t in types), sum(t == (7, 7) for t in types)
  n = 4 if nboxbox > 0 else 0
  if not (int(mjm.opt.disableflags) & (1 << 19)) and nboxmesh + nmeshmesh > 0:
    pv = np.asarray(getattr(mjm, "mesh_polyvertnum", np.zeros(0)))
    n = max(int(pv.max()) if len(pv) else 0, 4 if nboxmesh else n)
  return n


def filtered_geom_pairs(mjm):
  """Pre-filtered geom pairs in upper-triangular order (reference io.py:551-577)."""
  DSBL_FILTERPARENT = 1 << 10
  filterparent = not (mjm.opt.disableflags & DSBL_FILTERPARENT)
  out = []
  excl = set(int(x) for x in np.atleast_1d(mjm.exclude_signature))
  for g1 in range(mjm.ngeom):
    for g2 in range(g1 + 1, mjm.ngeom):
      b1, b2 = int(mjm.geom_bodyid[g1]), int(mjm.geom_bodyid[g2])
      w1, w2 = int(mjm.body_weldid[b1]), int(mjm.body_weldid[b2])
      wp1, wp2 = int(mjm.body_weldid[mjm.body_parentid[w1]]), int(mjm.body_weldid[mjm.body_parentid[w2]])
      if w1 == w2:
        continue
      if filterparent and w1 != 0 and w2 != 0 and (w1 == wp2 or w2 == wp1):
        continue
      mask = (mjm.geom_contype[g1] & mjm.geom_conaffinity[g2]) | (mjm.geom_contype[g2] & mjm.geom_conaffinity[g1])
      if not mask:
        continue
      if ((b1 << 16) + b2) in excl:
        continue
      out.append((g1, g2))
  return np.array(out, dtype=np.int32).reshape(-1, 2)


_SIZES = {
  "qpos": "nq", "qvel": "nv", "act": "na", "ctrl": "nu", "qacc_warmstart": "nv", "qfrc_applied": "nv",
  "xfrc_applied": ("nbody", 6), "mocap_pos": ("nmocap", 3), "mocap_quat": ("nmocap", 4), "xpos": ("nbody", 3), "xquat": ("nbody", 4), "xmat": ("nbody", 9), "xipos": ("nbody", 3),
  "ximat": ("nbody", 9), "xanchor": ("njnt", 3), "xaxis": ("njnt", 3), "geom_xpos": ("ngeom", 3), "geom_xmat": ("ngeom", 9),
  "subtree_com": ("nbody", 3), "cinert": ("nbody", 10), "cdof": ("nv", 6), "crb": ("nbody", 10), "M": "nC", "qLD": "nC",
  "qLDiagInv": "nv", "cvel": ("nbody", 6), "cdof_dot": ("nv", 6), "qfrc_spring": "nv", "qfrc_damper": "nv",
  "qfrc_gravcomp": "nv", "qfrc_passive": "nv", "qfrc_bias": "nv", "cacc": ("nbody", 6), "cfrc_int": ("nbody", 6),
  "actuator_length": "nu", "actuator_velocity": "nu", "actuator_force": "nu", "act_dot": "na", "qfrc_actuator": "nv",
  "qfrc_smooth": "nv", "qacc_smooth": "nv", "qacc": "nv", "qfrc_constraint": "nv", "Ma": "nv",
  "con_dist": "nconmax", "con_pos": ("nconmax", 3), "con_frame": ("nconmax", 9), "con_includemargin": "nconmax",
  "con_friction": ("nconmax", 5), "con_solref": ("nconmax", 2), "con_solreffriction": ("nconmax", 2),
  "con_solimp": ("nconmax", 5), "con_dim": "nconmax", "con_geom": ("nconmax", 2), "con_efc_address": ("nconmax", 10),
  "efc_type": "njmax", "efc_id": "njmax", "efc_state": "njmax", "efc_J": ("njmax", "nv"), "efc_pos": "njmax",
  "efc_margin": "njmax", "efc_D": "njmax", "efc_vel": "njmax", "efc_aref": "njmax", "efc_frictionloss": "njmax",
  "efc_force": "njmax", "tree_asleep": "ntree", "tree_awake": "ntree", "body_awake": "nbody", "tree_island": "ntree",
  "body_awake_ind": "nbody", "dof_awake_ind": "nv", "sensordata": "nsensordata", "subtree_linvel": ("nbody", 3), "subtree_angmom": ("nbody", 3), "cfrc_ext": ("nbody", 6),
}
MJ_MINAWAKE = 10  # mjMINAWAKE (reference types.py:29)


class RefSim:
  """One float64 world: model struct + data arrays (numpy-owned) for the C oracle."""

  def __init__(self, mjm, nconmax=64, njmax=256, tolerance=None, solver=None, iterations=None, ls_iterations=None,
               integrator=None, broadphase=0, broadphase_filter=3, real="f64"):
    """real="f32" runs the float32 build of the same restatement (libmjref32.so): not an oracle but the float32 FLOOR of the
    reference's algorithm on a CPU -- what an error against the float64 oracle is worth (tools/parity_report.py)."""
    self.mjm = mjm
    self._keep = []
    R = self.R = {"f64": _F64, "f32": _F32}[real]
    CRefModel, CRefData, _ctype = R.CRefModel, R.CRefData, R.ctype
    cm = CRefModel()
    opt = mjm.opt
    sizes = dict(nq=mjm.nq, nv=mjm.nv, nu=mjm.nu, na=mjm.na, nbody=mjm.nbody, njnt=mjm.njnt, ngeom=mjm.ngeom,
                 nC=int(mjm.M_rownnz.sum()) if mjm.nv else 0, njmax=njmax, nconmax=nconmax)
    pairs = filtered_geom_pairs(mjm)
    nexplicit = int(getattr(mjm, "npair", 0))
    pairid = np.full(len(pairs), -1, dtype=np.int32)
    if nexplicit:  # explicit <contact><pair> entries override / extend the filtered list (reference io.py:575-590)
      table = {(int(a), int(b)): -1 for a, b in pairs}
      for i in range(nexplicit):
        a, b = int(mjm.pair_geom1[i]), int(mjm.pair_geom2[i])
        table[(min(a, b), max(a, b))] = i
      keys = sorted(table)
      pairs = np.array(keys, dtype=np.int32).reshape(-1, 2)
      pairid = np.array([table[k] for k in keys], dtype=np.int32)
    sizes["npair"] = len(pairs)
    sizes["nexplicit"] = nexplicit
    sizes["neq"] = int(getattr(mjm, "neq", 0))
    sizes["nmocap"] = int(getattr(mjm, "nmocap", 0))
    sizes["ntree"] = int(getattr(mjm, "ntree", 0))
    sizes["nsite"], sizes["nsensor"], sizes["nsensordata"] = int(getattr(mjm, "nsite", 0)), int(getattr(mjm, "nsensor", 0)), int(getattr(mjm, "nsensordata", 0))
    scalars = dict(
      integrator=int(opt.integrator if integrator is None else integrator), cone=int(opt.cone),
      solver=int(opt.solver if solver is None else solver),
      iterations=int(opt.iterations if iterations is None else iterations),
      ls_iterations=int(opt.ls_iterations if ls_iterations is None else ls_iterations),
      disableflags=int(opt.disableflags), enableflags=int(getattr(opt, 'enableflags', 0)), sleep_tolerance=float(getattr(opt, 'sleep_tolerance', 1e-4)), broadphase=int(broadphase), broadphase_filter=int(broadphase_filter), ccd_iterations=int(getattr(opt, 'ccd_iterations', 35)), epa_iterations=_epa_iterations(mjm, pairs),
      ccd_tolerance=float(getattr(opt, 'ccd_tolerance', 1e-6)), timestep=float(opt.timestep),
      tolerance=float(opt.tolerance if tolerance is None else tolerance), ls_tolerance=float(opt.ls_tolerance),
      impratio=float(opt.impratio), meaninertia=float(mjm.stat.meaninertia))
    special = {"gravity": np.asarray(opt.gravity, dtype=self.R.np_real), "magnetic": np.asarray(getattr(opt, "magnetic", [0.0, -0.5, 0.0]), dtype=self.R.np_real), "pair_geom": pairs, "nxn_pairid": pairid,
               "xpair_dim": getattr(mjm, "pair_dim", np.zeros(0)), "xpair_friction": getattr(mjm, "pair_friction", np.zeros(0)),
               "xpair_solref": getattr(mjm, "pair_solref", np.zeros(0)), "xpair_solreffriction": getattr(mjm, "pair_solreffriction", np.zeros(0)),
               "xpair_solimp": getattr(mjm, "pair_solimp", np.zeros(0)), "xpair_margin": getattr(mjm, "pair_margin", np.zeros(0)),
               "xpair_gap": getattr(mjm, "pair_gap", np.zeros(0)),
               "actuator_trnid": mjm.actuator_trnid, "M_colind": mjm.M_colind}
    for name, dflt in (("geom_dataid", np.full(mjm.ngeom, -1)), ("mesh_vertadr", np.zeros(0)), ("mesh_vertnum", np.zeros(0)), ("mesh_vert", np.zeros((0, 3))),
                       ("site_bodyid", np.zeros(0)), ("site_pos", np.zeros((0, 3))), ("site_quat", np.zeros((0, 4))), ("site_type", np.zeros(0)), ("site_size", np.zeros((0, 3))), ("geom_group", np.zeros(mjm.ngeom)), ("geom_matid", np.full(mjm.ngeom, -1)), ("geom_rgba", np.tile([0.5, 0.5, 0.5, 1.0], (mjm.ngeom, 1))), ("mat_rgba", np.zeros((0, 4))), ("sensor_type", np.zeros(0)), ("sensor_datatype", np.zeros(0)),
                       ("sensor_objtype", np.zeros(0)), ("sensor_objid", np.zeros(0)), ("sensor_reftype", np.zeros(0)), ("sensor_refid", np.zeros(0)), ("sensor_dim", np.zeros(0)),
                       ("sensor_adr", np.zeros(0)), ("sensor_cutoff", np.zeros(0)),
                       ("hfield_size", np.zeros((0, 4))), ("hfield_nrow", np.zeros(0)), ("hfield_ncol", np.zeros(0)), ("hfield_adr", np.zeros(0)), ("hfield_data", np.zeros(0)),
                       ("mesh_graphadr", np.full(max(int(getattr(mjm, "nmesh", 0)), 1), -1)), ("mesh_graph", np.zeros(0)), ("mesh_polyadr", np.zeros(0)), ("mesh_polynormal", np.zeros((0, 3))), ("mesh_polyvertadr", np.zeros(0)), ("mesh_polyvertnum", np.zeros(0)),
                       ("mesh_polyvert", np.zeros(0)), ("mesh_polymapadr", np.zeros(0)), ("mesh_polymapnum", np.zeros(0)), ("mesh_polymap", np.zeros(0))):
      special[name] = np.asarray(getattr(mjm, name, dflt))
    sizes["nmeshpoly"] = int(len(special["mesh_polyvertnum"]))
    sizes["npolygonmax"] = npolygonmax(mjm, pairs)
    for name, dt in (("tree_sleep_policy", np.int32), ("dof_length", np.float64)):  # (absent on models that predate sleeping)
      special[name] = np.asarray(getattr(mjm, name, np.full(sizes["ntree"], 2) if name == "tree_sleep_policy" else np.ones(mjm.nv)), dtype=dt)
    for name, kind, ptr in _MODEL_FIELDS:
      if not ptr:
        setattr(cm, name, sizes[name] if name in sizes else scalars[name])
        continue
      src = special[name] if name in special else getattr(mjm, name)
      arr = np.ascontiguousarray(np.asarray(src), dtype=np.int32 if kind == "int" else R.np_real)
      if arr.size == 0:
        arr = np.zeros(1, dtype=arr.dtype)
      self._keep.append(arr)
      setattr(cm, name, arr.ctypes.data_as(_ctype(kind, True)))
    self.cm = cm
    self.sizes = sizes
    cd = CRefData()
    self.arr = {}
    for name, kind, ptr in _DATA_FIELDS:
      if not ptr:
        continue
      spec = _SIZES[name]
      if isinstance(spec, str):
        shape = (sizes[spec],)
      else:
        shape = tuple(sizes[s] if isinstance(s, str) else s for s in spec)
      a = np.zeros(max(int(np.prod(shape)), 1), dtype=np.int32 if kind == "int" else R.np_real)
      self.arr[name] = a
      setattr(cd, name, a.ctypes.data_as(_ctype(kind, True)))
      setattr(self, "_shape_" + name, shape)
    self.cd = cd
    self.lib = R.lib()
    self.reset()

  def reset(self, key=None):
    m = self.mjm
    for a in self.arr.values():
      a[:] = 0
    self.qpos[:] = m.qpos0 if key is None else m.key_qpos[key]
    if key is not None:
      self.qvel[:] = m.key_qvel[key]
      self.ctrl[:] = m.key_ctrl[key]
      if m.na:
        self.act[:] = m.key_act[key]
    self.cd.time = 0.0
    self.cd.overflow = 0
    self.tree_asleep[:] = -(1 + MJ_MINAWAKE)  # fully awake (reference io.py:1869)
    self.tree_island[:] = -1
    self._call("update_sleep")
    for b in range(m.nbody):  # mocap bodies start at their model pose
      if int(m.body_mocapid[b]) >= 0:
        self.mocap_pos[int(m.body_mocapid[b])] = m.body_pos[b]
        self.mocap_quat[int(m.body_mocapid[b])] = m.body_quat[b]

  def __getattr__(self, name):
    arr = self.__dict__.get("arr", {})
    if name in arr:
      shape = self.__dict__["_shape_" + name]
      n = int(np.prod(shape))
      return arr[name][:n].reshape(shape)
    if name in ("ncon", "ne", "nf", "nl", "nefc", "solver_niter", "overflow", "time", "ncollision", "nisland", "ntree_awake", "nbody_awake", "nv_awake"):
      return getattr(self.__dict__["cd"], name)
    raise AttributeError(name)

  def _call(self, fn):
    getattr(self.lib, "ref_" + fn)(ctypes.byref(self.cm), ctypes.byref(self.cd))

  def step(self):
    self._call("step")

  def forward(self):
    self._call("forward")

  def stage(self, name):
    self._call(name)

  def ccd_trace_start(self):
    """Arm the branch trace of the convex narrowphase (oracle/ccd.c): the next collision pass records, per convex pair, the gate of
    gjk_phase (collision_gjk.py:2376-2414) it left through."""
    self.lib.ref_ccd_trace_start()

  def ccd_trace(self, cap=4096):
    """[(g1, g2, branch, simplex dim, separated flag, contacts, GJK distance, final distance)] of the traced pass; branch: 1 shrunk cores
    separated | 2 GJK distance > tolerance | 3 simplex < 2 points | 4 GJK `separated` | 5 degenerate polytope seed | 6 EPA failed | 7 EPA depth."""
    ints = np.zeros((cap, 6), dtype=np.int32)
    reals = np.zeros((cap, 2), dtype=self.R.np_real)
    self.lib.ref_ccd_trace_get.restype = ctypes.c_int
    n = self.lib.ref_ccd_trace_get(ints.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), reals.ctypes.data_as(ctypes.POINTER(self.R.c_real)), cap)
    return [tuple(int(x) for x in ints[i]) + (float(reals[i, 0]), float(reals[i, 1])) for i in range(n)]

  def solve_m(self, y):
    y = np.ascontiguousarray(y, dtype=self.R.np_real)
    x = np.zeros_like(y)
    dp = ctypes.POINTER(self.R.c_real)
    self.lib.ref_solve_m(ctypes.byref(self.cm), ctypes.byref(self.cd), x.ctypes.data_as(dp), y.ctypes.data_as(dp))
    return x

  def mul_m(self, v):
    v = np.ascontiguousarray(v, dtype=self.R.np_real)
    r = np.zeros_like(v)
    dp = ctypes.POINTER(self.R.c_real)
    self.lib.ref_mul_m(ctypes.byref(self.cm), ctypes.byref(self.cd), r.ctypes.data_as(dp), v.ctypes.data_as(dp))
    return r

  def ctrl_noise(self, step, worldid, noise_std=0.01, noise_rate=0.1, center=None):
    dp = ctypes.POINTER(self.R.c_real)
    c = np.zeros(max(self.mjm.nu, 1), dtype=self.R.np_real) if center is None else np.ascontiguousarray(center, dtype=self.R.np_real)
    self.lib.ref_ctrl_noise(ctypes.byref(self.cm), ctypes.byref(self.cd), c.ctypes.data_as(dp), step, worldid, noise_std, noise_rate)

  def rollout(self, nstep, worldid=0, noise_std=0.01, noise_rate=0.1, record=True):
    dp = ctypes.POINTER(self.R.c_real)
    qp = np.zeros((nstep, self.mjm.nq), dtype=self.R.np_real) if record else None
    qv = np.zeros((nstep, self.mjm.nv), dtype=self.R.np_real) if record else None
    ok = self.lib.ref_rollout(ctypes.byref(self.cm), ctypes.byref(self.cd), nstep, worldid, noise_std, noise_rate,
                              qp.ctypes.data_as(dp) if record else None, qv.ctypes.data_as(dp) if record else None)
    return ok, qp, qv

  def ccd_geoms(self, g1, g2, pos1=None, mat1=None, pos2=None, mat2=None, margin=0.0, tolerance=1e-6, cutoff=1e30, iterations=35, multiccd=True):
    """GJK / EPA / multi-contact on two geoms of the model at given poses (default: their current geom_xpos / geom_xmat); like the
    reference's test harness collision_gjk_test.py:_geom_dist.  Returns (dist, ncon, witness pairs [ncon, 2, 3])."""
    def arr(x, dflt):
      return np.ascontiguousarray(np.asarray(dflt if x is None else x, dtype=self.R.np_real).reshape(-1))
    p1, m1 = arr(pos1, self.geom_xpos[g1]), arr(mat1, self.geom_xmat[g1])
    p2, m2 = arr(pos2, self.geom_xpos[g2]), arr(mat2, self.geom_xmat[g2])
    out, wit = np.zeros(9, dtype=self.R.np_real), np.zeros(48, dtype=self.R.np_real)
    dp = lambda a: a.ctypes.data_as(ctypes.POINTER(self.R.c_real))
    fn = self.lib.ref_ccd_geoms
    fn.restype = ctypes.c_int
    n = fn(ctypes.byref(self.cm), int(g1), int(g2), dp(p1), dp(m1), dp(p2), dp(m2), self.R.c_real(margin), self.R.c_real(tolerance),
           self.R.c_real(cutoff), int(iterations), int(bool(multiccd)), dp(out), dp(wit))
    return float(out[0]), n, wit.reshape(8, 2, 3)[: max(n, 0)].copy()

  def ray(self, pnt, vec, geomgroup=None, flg_static=True, bodyexclude=-1):
    """Nearest intersection of one ray with the model's primitive geoms at their current poses (ray.py:907 _ray): (dist, geomid, normal)."""
    dp = lambda a: a.ctypes.data_as(ctypes.POINTER(self.R.c_real))
    p, v = np.ascontiguousarray(pnt, dtype=self.R.np_real), np.ascontiguousarray(vec, dtype=self.R.np_real)
    gg = None if geomgroup is None else np.ascontiguousarray(geomgroup, dtype=self.R.np_real)
    gid, nrm = ctypes.c_int(-1), np.zeros(3, dtype=self.R.np_real)
    fn = self.lib.ref_ray
    fn.restype = self.R.c_real
    dist = fn(ctypes.byref(self.cm), ctypes.byref(self.cd), dp(p), dp(v), None if gg is None else dp(gg), int(bool(flg_static)), int(bodyexclude), ctypes.byref(gid), dp(nrm))
    return float(dist), int(gid.value), nrm

  def dense_M(self):
    m = self.mjm
    nv = m.nv
    M = np.zeros((nv, nv))
    for i in range(nv):
      for k in range(m.M_rownnz[i]):
        j = m.M_colind[m.M_rowadr[i] + k]
        M[i, j] = M[j, i] = self.M[m.M_rowadr[i] + k]
    return M
