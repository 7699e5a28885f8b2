"""Enums and the Model/Data containers of the drop-in boundary.

Mirrors the hot-path subset of /root/reference/mujoco_warp/_src/types.py: enums 119-545, Option 836,
Statistic 922, Model 982-1960, Contact 1975, Constraint 2021, Data 2075-2374.  Field names, shapes and
dtypes (float32 / int32, world-major) are the reference's; fields that belong to out-of-scope subsystems
(tendons, flex, sensors, rendering, islands) are absent.  Arrays are `DeviceArray`s (device.py).

Deviation (documented in DESIGN.md): `Data.qLD` holds MuJoCo's sparse L'DL factor in the CSR M-structure
([nworld, nC], plus `qLDiagInv`), not the reference's packed dense per-tree Cholesky blocks.
"""

import dataclasses
import enum

import numpy as np

from .device import DeviceArray


class BroadphaseType(enum.IntEnum):
  NXN = 0
  SAP_TILE = 1
  SAP_SEGMENTED = 2


class BroadphaseFilter(enum.IntFlag):
  PLANE = 1 << 0
  SPHERE = 1 << 1
  AABB = 1 << 2
  OBB = 1 << 3


class OverflowType(enum.IntFlag):
  NEFC = 1 << 0
  NJMAX_NNZ = 1 << 1
  BROADPHASE = 1 << 2
  NARROWPHASE = 1 << 3
  CCD = 1 << 4
  HFIELD = 1 << 5
  CONTACT_MATCH = 1 << 6
  NVMAX = 1 << 7
  EPA_HORIZON = 1 << 8
  ITERATIONS = 1 << 9
  LS_ITERATIONS = 1 << 10


class DisableBit(enum.IntFlag):
  CONSTRAINT = 1 << 0
  EQUALITY = 1 << 1
  FRICTIONLOSS = 1 << 2
  LIMIT = 1 << 3
  CONTACT = 1 << 4
  SPRING = 1 << 5
  DAMPER = 1 << 6
  GRAVITY = 1 << 7
  CLAMPCTRL = 1 << 8
  WARMSTART = 1 << 9
  FILTERPARENT = 1 << 10
  ACTUATION = 1 << 11
  REFSAFE = 1 << 12
  SENSOR = 1 << 13
  MIDPHASE = 1 << 14
  EULERDAMP = 1 << 15
  AUTORESET = 1 << 16
  NATIVECCD = 1 << 17
  ISLAND = 1 << 18
  MULTICCD = 1 << 19


class EnableBit(enum.IntFlag):
  OVERRIDE = 1 << 0
  ENERGY = 1 << 1
  FWDINV = 1 << 2
  INVDISCRETE = 1 << 3
  SLEEP = 1 << 5


class ObjType(enum.IntEnum):
  """Object types (reference types.py:651; mjtObj values)."""

  UNKNOWN = 0
  BODY = 1
  XBODY = 2
  GEOM = 5
  SITE = 6
  CAMERA = 7
  FLEX = 9


class TrnType(enum.IntEnum):
  JOINT = 0
  JOINTINPARENT = 1
  SLIDERCRANK = 2
  TENDON = 3
  SITE = 4
  BODY = 5


class DynType(enum.IntEnum):
  NONE = 0
  INTEGRATOR = 1
  FILTER = 2
  FILTEREXACT = 3
  MUSCLE = 4
  USER = 5


class GainType(enum.IntEnum):
  FIXED = 0
  AFFINE = 1
  MUSCLE = 2
  USER = 3


class BiasType(enum.IntEnum):
  NONE = 0
  AFFINE = 1
  MUSCLE = 2
  USER = 3


class JointType(enum.IntEnum):
  FREE = 0
  BALL = 1
  SLIDE = 2
  HINGE = 3


class ConeType(enum.IntEnum):
  PYRAMIDAL = 0
  ELLIPTIC = 1


class IntegratorType(enum.IntEnum):
  EULER = 0
  RK4 = 1
  IMPLICIT = 2
  IMPLICITFAST = 3


class GeomType(enum.IntEnum):
  PLANE = 0
  HFIELD = 1
  SPHERE = 2
  CAPSULE = 3
  ELLIPSOID = 4
  CYLINDER = 5
  BOX = 6
  MESH = 7
  SDF = 8


class State(enum.IntFlag):
  """State components as bit flags (reference types.py:712; mjtState -- UNPINNED here: the mujoco package is absent, the order follows the
  reference's attribute list).  HISTORY, USERDATA and PLUGIN have no storage in this engine."""

  TIME = 1 << 0
  QPOS = 1 << 1
  QVEL = 1 << 2
  ACT = 1 << 3
  HISTORY = 1 << 4
  WARMSTART = 1 << 5
  CTRL = 1 << 6
  QFRC_APPLIED = 1 << 7
  XFRC_APPLIED = 1 << 8
  EQ_ACTIVE = 1 << 9
  MOCAP_POS = 1 << 10
  MOCAP_QUAT = 1 << 11
  USERDATA = 1 << 12
  PLUGIN = 1 << 13
  NSTATE = 14
  PHYSICS = TIME | QPOS | QVEL | ACT | HISTORY
  FULLPHYSICS = PHYSICS | PLUGIN
  USER = CTRL | QFRC_APPLIED | XFRC_APPLIED | EQ_ACTIVE | MOCAP_POS | MOCAP_QUAT | USERDATA
  INTEGRATION = FULLPHYSICS | USER | WARMSTART


MJ_MINAWAKE = 10  # mjMINAWAKE: steps a tree must stay below the sleep tolerance before it may sleep (reference types.py:29)


class SleepPolicy(enum.IntEnum):
  """Per-tree sleep policy (reference types.py:296; NEVER / ALLOWED / INIT unsupported there and here)."""

  AUTO = 0
  AUTO_NEVER = 1
  AUTO_ALLOWED = 2


class SleepState(enum.IntEnum):
  """Body sleep state (reference types.py:311; mjtSleepState)."""

  STATIC = -1
  ASLEEP = 0
  AWAKE = 1


class SolverType(enum.IntEnum):
  PGS = 0  # the reference has none (types.py:502); here: MuJoCo C's dual projected Gauss-Seidel (csrc/pgs.hpp)
  CG = 1
  NEWTON = 2


class ConstraintState(enum.IntEnum):
  SATISFIED = 0
  QUADRATIC = 1
  LINEARNEG = 2
  LINEARPOS = 3
  CONE = 4


class EqType(enum.IntEnum):
  CONNECT = 0
  WELD = 1
  JOINT = 2
  TENDON = 3
  FLEX = 4


class ConstraintType(enum.IntEnum):
  EQUALITY = 0
  FRICTION_DOF = 1
  FRICTION_TENDON = 2
  LIMIT_JOINT = 3
  LIMIT_TENDON = 4
  CONTACT_FRICTIONLESS = 5
  CONTACT_PYRAMIDAL = 6
  CONTACT_ELLIPTIC = 7


class ContactType(enum.IntFlag):
  CONSTRAINT = 1
  SENSOR = 2


MJ_MINVAL = 1e-15
MJ_MAXVAL = 1e10
MJ_MINIMP = 0.0001
MJ_MAXIMP = 0.9999
MJ_MINMU = 1e-5


def _arr(shape, dtype, host=False):
  """Declares an array field: `shape` is a tuple of sizes or size NAMES evaluated against the owning Model / Data ('*' = 1 or
  nworld, the reference's batched fields, types.py:822-833), `dtype` its numpy dtype; host arrays stay numpy (keyframes)."""
  return dataclasses.field(default=None, repr=False, metadata={"shape": tuple(shape), "dtype": dtype, "host": host})


class _Dirty:
  """Attribute container that remembers when a field was re-bound (so the C struct is rebuilt lazily)."""

  def __setattr__(self, name, value):
    object.__setattr__(self, name, value)
    if not name.startswith("_"):
      object.__setattr__(self, "_dirty", True)
      root = self.__dict__.get("_root")
      if root is not None:
        object.__setattr__(root, "_dirty", True)


@dataclasses.dataclass(eq=False)
class Option(_Dirty):
  """Physics options (reference types.py:836-905); batched ('*') fields carry a leading dimension of 1 or nworld."""

  timestep: DeviceArray = _arr(('*',), "float32")
  tolerance: DeviceArray = _arr(('*',), "float32")
  ls_tolerance: DeviceArray = _arr(('*',), "float32")
  gravity: DeviceArray = _arr(('*', 3), "float32")
  impratio_invsqrt: DeviceArray = _arr(('*',), "float32")
  ccd_tolerance: DeviceArray = _arr(('*',), "float32")
  magnetic: DeviceArray = _arr(('*', 3), "float32")
  integrator: int = 0
  cone: int = 0
  solver: int = 0
  iterations: int = 0
  ls_iterations: int = 0
  ccd_iterations: int = 0
  disableflags: int = 0
  enableflags: int = 0
  broadphase: int = 0
  broadphase_filter: int = 0
  graph_conditional: bool = False
  run_collision_detection: bool = False
  warn_overflow: bool = False

@dataclasses.dataclass(eq=False)
class Statistic(_Dirty):
  """Model statistics (reference types.py:922-931)."""

  meaninertia: DeviceArray = _arr(('*',), "float32")

@dataclasses.dataclass(eq=False)
class Model(_Dirty):
  """Device-resident model (reference types.py:982-1960, hot-path subset).  '*' = 1 or nworld (domain randomisation)."""

  nq: int = 0
  nv: int = 0
  nu: int = 0
  na: int = 0
  nbody: int = 0
  njnt: int = 0
  ngeom: int = 0
  nsite: int = 0
  nkey: int = 0
  nmocap: int = 0
  neq: int = 0
  ntendon: int = 0
  nsensor: int = 0
  nmesh: int = 0
  nflex: int = 0
  nhfield: int = 0
  ncam: int = 0
  nlight: int = 0
  nC: int = 0
  nM: int = 0
  heavy_colliders: int = 0
  is_sparse: bool = False
  nv_pad: int = 0
  eq_active0: np.ndarray = _arr(('neq',), "int32", host=True)
  qpos0: DeviceArray = _arr(('*', 'nq'), "float32")
  qpos_spring: DeviceArray = _arr(('*', 'nq'), "float32")
  body_parentid: DeviceArray = _arr(('nbody',), "int32")
  body_rootid: DeviceArray = _arr(('nbody',), "int32")
  body_weldid: DeviceArray = _arr(('nbody',), "int32")
  body_jntnum: DeviceArray = _arr(('nbody',), "int32")
  body_jntadr: DeviceArray = _arr(('nbody',), "int32")
  body_dofnum: DeviceArray = _arr(('nbody',), "int32")
  body_dofadr: DeviceArray = _arr(('nbody',), "int32")
  body_lastdof: DeviceArray = _arr(('nbody',), "int32")
  body_mocapid: DeviceArray = _arr(('nbody',), "int32")
  body_subtreenum: DeviceArray = _arr(('nbody',), "int32")
  body_tree: DeviceArray = _arr(('nbody',), "int32")
  body_leveladr: DeviceArray = _arr(('nbodylevel+1',), "int32")
  body_dofmask: DeviceArray = _arr(('nbody', '(nv+31)//32'), "uint32")
  body_pos: DeviceArray = _arr(('*', 'nbody', 3), "float32")
  body_quat: DeviceArray = _arr(('*', 'nbody', 4), "float32")
  body_ipos: DeviceArray = _arr(('*', 'nbody', 3), "float32")
  body_iquat: DeviceArray = _arr(('*', 'nbody', 4), "float32")
  body_mass: DeviceArray = _arr(('*', 'nbody'), "float32")
  body_subtreemass: DeviceArray = _arr(('*', 'nbody'), "float32")
  body_inertia: DeviceArray = _arr(('*', 'nbody', 3), "float32")
  body_invweight0: DeviceArray = _arr(('*', 'nbody', 2), "float32")
  body_gravcomp: DeviceArray = _arr(('*', 'nbody'), "float32")
  jnt_type: DeviceArray = _arr(('njnt',), "int32")
  jnt_qposadr: DeviceArray = _arr(('njnt',), "int32")
  jnt_dofadr: DeviceArray = _arr(('njnt',), "int32")
  jnt_bodyid: DeviceArray = _arr(('njnt',), "int32")
  jnt_limited: DeviceArray = _arr(('njnt',), "int32")
  jnt_solref: DeviceArray = _arr(('*', 'njnt', 2), "float32")
  jnt_solimp: DeviceArray = _arr(('*', 'njnt', 5), "float32")
  jnt_pos: DeviceArray = _arr(('*', 'njnt', 3), "float32")
  jnt_axis: DeviceArray = _arr(('*', 'njnt', 3), "float32")
  jnt_stiffness: DeviceArray = _arr(('*', 'njnt'), "float32")
  jnt_range: DeviceArray = _arr(('*', 'njnt', 2), "float32")
  jnt_margin: DeviceArray = _arr(('*', 'njnt'), "float32")
  jnt_actfrclimited: DeviceArray = _arr(('njnt',), "int32")
  jnt_actfrcrange: DeviceArray = _arr(('*', 'njnt', 2), "float32")
  jnt_actgravcomp: DeviceArray = _arr(('njnt',), "int32")
  dof_bodyid: DeviceArray = _arr(('nv',), "int32")
  dof_jntid: DeviceArray = _arr(('nv',), "int32")
  dof_parentid: DeviceArray = _arr(('nv',), "int32")
  dof_grpadr: DeviceArray = _arr(('nv',), "int32")
  dof_tree: DeviceArray = _arr(('nv',), "int32")
  dof_leveladr: DeviceArray = _arr(('ndoflevel+1',), "int32")
  tree_dofadr: DeviceArray = _arr(('ntree',), "int32")
  tree_dofnum: DeviceArray = _arr(('ntree',), "int32")
  dof_treeid: DeviceArray = _arr(('nv',), "int32")
  body_treeid: DeviceArray = _arr(('nbody',), "int32")
  dof_solref: DeviceArray = _arr(('*', 'nv', 2), "float32")
  dof_solimp: DeviceArray = _arr(('*', 'nv', 5), "float32")
  dof_frictionloss: DeviceArray = _arr(('*', 'nv'), "float32")
  dof_armature: DeviceArray = _arr(('*', 'nv'), "float32")
  dof_damping: DeviceArray = _arr(('*', 'nv'), "float32")
  dof_invweight0: DeviceArray = _arr(('*', 'nv'), "float32")
  M_rownnz: DeviceArray = _arr(('nv',), "int32")
  M_rowadr: DeviceArray = _arr(('nv',), "int32")
  M_colind: DeviceArray = _arr(('nC',), "int32")
  M_dense: DeviceArray = _arr(('nv', '4*((nv+3)//4)'), "int32")
  geom_type: DeviceArray = _arr(('ngeom',), "int32")
  geom_condim: DeviceArray = _arr(('ngeom',), "int32")
  geom_bodyid: DeviceArray = _arr(('ngeom',), "int32")
  geom_priority: DeviceArray = _arr(('ngeom',), "int32")
  nmat: int = 0
  geom_group: DeviceArray = _arr(('ngeom',), "int32")
  geom_matid: DeviceArray = _arr(('ngeom',), "int32")
  geom_rgba: DeviceArray = _arr(('ngeom', 4), "float32")
  mat_rgba: DeviceArray = _arr(('nmat', 4), "float32")
  geom_solmix: DeviceArray = _arr(('*', 'ngeom'), "float32")
  geom_solref: DeviceArray = _arr(('*', 'ngeom', 2), "float32")
  geom_solimp: DeviceArray = _arr(('*', 'ngeom', 5), "float32")
  geom_size: DeviceArray = _arr(('*', 'ngeom', 3), "float32")
  geom_rbound: DeviceArray = _arr(('*', 'ngeom'), "float32")
  geom_aabb: DeviceArray = _arr(('*', 'ngeom', 6), "float32")
  geom_pos: DeviceArray = _arr(('*', 'ngeom', 3), "float32")
  geom_quat: DeviceArray = _arr(('*', 'ngeom', 4), "float32")
  geom_friction: DeviceArray = _arr(('*', 'ngeom', 3), "float32")
  geom_margin: DeviceArray = _arr(('*', 'ngeom'), "float32")
  geom_gap: DeviceArray = _arr(('*', 'ngeom'), "float32")
  nxn_geom_pair: DeviceArray = _arr(('npair', 2), "int32")
  nxn_pairid: DeviceArray = _arr(('npair',), "int32")
  nxn_pairindex: DeviceArray = _arr(('ngeom*(ngeom-1)//2',), "int32")
  cull_geom: DeviceArray = _arr(('ncullgeom', 2), "int32")  # k_broad_mask's group pre-test (io.cull_tables)
  cull_group: DeviceArray = _arr(('ncullgroup', 2), "int32")
  cull_pair: DeviceArray = _arr(('ncullpair', 4), "int32")
  cull_list: DeviceArray = _arr(('npair', 2), "int32")
  pair_dim: DeviceArray = _arr(('nexplicit',), "int32")
  pair_friction: DeviceArray = _arr(('nexplicit', 5), "float32")
  pair_solref: DeviceArray = _arr(('nexplicit', 2), "float32")
  pair_solreffriction: DeviceArray = _arr(('nexplicit', 2), "float32")
  pair_solimp: DeviceArray = _arr(('nexplicit', 5), "float32")
  pair_margin: DeviceArray = _arr(('nexplicit',), "float32")
  pair_gap: DeviceArray = _arr(('nexplicit',), "float32")
  site_bodyid: DeviceArray = _arr(('nsite',), "int32")
  site_pos: DeviceArray = _arr(('*', 'nsite', 3), "float32")
  site_quat: DeviceArray = _arr(('*', 'nsite', 4), "float32")
  actuator_dyntype: DeviceArray = _arr(('nu',), "int32")
  actuator_gaintype: DeviceArray = _arr(('nu',), "int32")
  actuator_biastype: DeviceArray = _arr(('nu',), "int32")
  actuator_trnid: DeviceArray = _arr(('nu', 2), "int32")
  actuator_actadr: DeviceArray = _arr(('nu',), "int32")
  actuator_ctrllimited: DeviceArray = _arr(('nu',), "int32")
  actuator_forcelimited: DeviceArray = _arr(('nu',), "int32")
  actuator_actlimited: DeviceArray = _arr(('nu',), "int32")
  actuator_dynprm: DeviceArray = _arr(('*', 'nu', 10), "float32")
  actuator_gainprm: DeviceArray = _arr(('*', 'nu', 10), "float32")
  actuator_biasprm: DeviceArray = _arr(('*', 'nu', 10), "float32")
  actuator_ctrlrange: DeviceArray = _arr(('*', 'nu', 2), "float32")
  actuator_forcerange: DeviceArray = _arr(('*', 'nu', 2), "float32")
  actuator_actrange: DeviceArray = _arr(('*', 'nu', 2), "float32")
  actuator_gear: DeviceArray = _arr(('*', 'nu', 6), "float32")
  eq_obj1id: DeviceArray = _arr(('neq',), "int32")
  eq_obj2id: DeviceArray = _arr(('neq',), "int32")
  eq_solref: DeviceArray = _arr(('*', 'neq', 2), "float32")
  eq_solimp: DeviceArray = _arr(('*', 'neq', 5), "float32")
  eq_data: DeviceArray = _arr(('*', 'neq', 11), "float32")
  opt: "Option" = None
  stat: "Statistic" = None
  npair: int = 0
  nexplicit: int = 0
  ncullgeom: int = 0
  ncullgroup: int = 0
  ncullpair: int = 0
  nxn_geom_pair_filtered: DeviceArray = _arr(('npair', 2), "int32")
  nbodylevel: int = 0
  ndoflevel: int = 0
  nmaxcondim: int = 0
  epa_iterations: int = 0  # EPA iteration cap of the convex narrowphase (reference collision_convex.py:1223)
  act_dof_max: int = 0  # largest number of actuators acting on one dof
  cg_basis: int = 0  # 1: every contact has condim 1 or 3 -- CG at nv <= 32, njmax <= 64 runs the pooled contact-basis kernel (csrc/solver_cgp.hpp)
  act_velfeedback: int = 0  # 1: positive velocity feedback through an actuator is possible (implicitfast is then not fused into the solver)
  sleep_enabled: int = 0  # EnableBit.SLEEP set and DisableBit.ISLAND clear (reference forward.py:345)
  opt_sleep_tolerance: float = 0.0  # Option.sleep_tolerance (one value per model)
  tree_sleep_policy: DeviceArray = _arr(('ntree',), "int32")
  geom_dataid: DeviceArray = _arr(('ngeom',), "int32")
  mesh_vertadr: DeviceArray = _arr(('nmesh',), "int32")
  mesh_vertnum: DeviceArray = _arr(('nmesh',), "int32")
  mesh_vert: DeviceArray = _arr(('nmeshvert', 3), "float32")
  nmeshvert: int = 0
  nmeshpoly: int = 0
  nmeshpolyvert: int = 0
  nmeshpolymap: int = 0
  npolygonmax: int = 0
  nmeshdegmax: int = 0
  nsensordata: int = 0
  nsensor_acc: int = 0
  nsensor_subtree: int = 0
  nsensor_frc: int = 0
  nsensor_energy: int = 0
  site_type: DeviceArray = _arr(('nsite',), "int32")
  site_size: DeviceArray = _arr(('nsite', 3), "float32")
  sensor_type: DeviceArray = _arr(('nsensor',), "int32")
  sensor_datatype: DeviceArray = _arr(('nsensor',), "int32")
  sensor_objtype: DeviceArray = _arr(('nsensor',), "int32")
  sensor_objid: DeviceArray = _arr(('nsensor',), "int32")
  sensor_reftype: DeviceArray = _arr(('nsensor',), "int32")
  sensor_refid: DeviceArray = _arr(('nsensor',), "int32")
  sensor_dim: DeviceArray = _arr(('nsensor',), "int32")
  sensor_adr: DeviceArray = _arr(('nsensor',), "int32")
  sensor_cutoff: DeviceArray = _arr(('nsensor',), "float32")
  nmeshgraph: int = 0
  nhfielddata: int = 0
  hfield_size: DeviceArray = _arr(('nhfield', 4), "float32")
  hfield_nrow: DeviceArray = _arr(('nhfield',), "int32")
  hfield_ncol: DeviceArray = _arr(('nhfield',), "int32")
  hfield_adr: DeviceArray = _arr(('nhfield',), "int32")
  hfield_data: DeviceArray = _arr(('nhfielddata',), "float32")
  mesh_graphadr: DeviceArray = _arr(('nmesh',), "int32")
  mesh_graph: DeviceArray = _arr(('nmeshgraph',), "int32")
  mesh_polyadr: DeviceArray = _arr(('nmesh',), "int32")
  mesh_polynormal: DeviceArray = _arr(('nmeshpoly', 3), "float32")
  mesh_polyvertadr: DeviceArray = _arr(('nmeshpoly',), "int32")
  mesh_polyvertnum: DeviceArray = _arr(('nmeshpoly',), "int32")
  mesh_polyvert: DeviceArray = _arr(('nmeshpolyvert',), "int32")
  mesh_polymapadr: DeviceArray = _arr(('nmeshvert',), "int32")
  mesh_polymapnum: DeviceArray = _arr(('nmeshvert',), "int32")
  mesh_polymap: DeviceArray = _arr(('nmeshpolymap',), "int32")
  dof_length: DeviceArray = _arr(('nv',), "float32")
  ntree: int = 0  # kinematic trees with at least one dof
  tree_nvmax: int = 0  # dofs of the largest tree
  isl_nv4: int = 0  # quarter-rows of the widest island of <= 32 dofs (kernel size class)
  isl_wide: int = 0  # islands of 33..64 dofs can form
  tree_solve: int = 0  # 1: nv > 64, several trees: worlds whose constraint islands have <= 64 dofs are solved per island
  nmaxpyramid: int = 0
  key_qpos: np.ndarray = _arr(('nkey', 'nq'), "float32", host=True)
  key_qvel: np.ndarray = _arr(('nkey', 'nv'), "float32", host=True)
  key_ctrl: np.ndarray = _arr(('nkey', 'nu'), "float32", host=True)
  key_mpos: np.ndarray = _arr(('nkey', '3*nmocap'), "float32", host=True)
  key_mquat: np.ndarray = _arr(('nkey', '4*nmocap'), "float32", host=True)
  key_act: np.ndarray = _arr(('nkey', 'na'), "float32", host=True)
  key_time: np.ndarray = _arr(('nkey',), "float32", host=True)

@dataclasses.dataclass(eq=False)
class Contact(_Dirty):
  """Contact arrays, flat over all worlds (reference types.py:1975-2018)."""

  dist: DeviceArray = _arr(('naconmax',), "float32")
  pos: DeviceArray = _arr(('naconmax', 3), "float32")
  frame: DeviceArray = _arr(('naconmax', 3, 3), "float32")
  includemargin: DeviceArray = _arr(('naconmax',), "float32")
  friction: DeviceArray = _arr(('naconmax', 5), "float32")
  solref: DeviceArray = _arr(('naconmax', 2), "float32")
  solreffriction: DeviceArray = _arr(('naconmax', 2), "float32")
  solimp: DeviceArray = _arr(('naconmax', 5), "float32")
  dim: DeviceArray = _arr(('naconmax',), "int32")
  geom: DeviceArray = _arr(('naconmax', 2), "int32")
  efc_address: DeviceArray = _arr(('naconmax', 'nmaxpyramid'), "int32")
  worldid: DeviceArray = _arr(('naconmax',), "int32")
  type: DeviceArray = _arr(('naconmax',), "int32")
  geomcollisionid: DeviceArray = _arr(('naconmax',), "int32")

@dataclasses.dataclass(eq=False)
class Constraint(_Dirty):
  """Constraint arrays (reference types.py:2021-2072); J is dense [nworld, njmax_pad, nv_pad]."""

  Ma: DeviceArray = _arr(('nworld', 'nv'), "float32")
  type: DeviceArray = _arr(('nworld', 'njmax'), "int32")
  id: DeviceArray = _arr(('nworld', 'njmax'), "int32")
  state: DeviceArray = _arr(('nworld', 'njmax'), "int32")
  J: DeviceArray = _arr(('nworld', 'njmax_pad', 'nv_pad'), "float32")
  pos: DeviceArray = _arr(('nworld', 'njmax'), "float32")
  margin: DeviceArray = _arr(('nworld', 'njmax'), "float32")
  D: DeviceArray = _arr(('nworld', 'njmax'), "float32")
  vel: DeviceArray = _arr(('nworld', 'njmax'), "float32")
  aref: DeviceArray = _arr(('nworld', 'njmax'), "float32")
  frictionloss: DeviceArray = _arr(('nworld', 'njmax'), "float32")
  force: DeviceArray = _arr(('nworld', 'njmax'), "float32")

@dataclasses.dataclass(eq=False)
class Data(_Dirty):
  """Device-resident, batched simulation state (reference types.py:2075-2374, hot-path subset; ws_* = engine workspace)."""

  contact: "Contact" = None
  efc: "Constraint" = None
  time: DeviceArray = _arr(('nworld',), "float32")
  qpos: DeviceArray = _arr(('nworld', 'nq'), "float32")
  qvel: DeviceArray = _arr(('nworld', 'nv'), "float32")
  act: DeviceArray = _arr(('nworld', 'na'), "float32")
  ctrl: DeviceArray = _arr(('nworld', 'nu'), "float32")
  qacc_warmstart: DeviceArray = _arr(('nworld', 'nv'), "float32")
  qfrc_applied: DeviceArray = _arr(('nworld', 'nv'), "float32")
  xfrc_applied: DeviceArray = _arr(('nworld', 'nbody', 6), "float32")
  mocap_pos: DeviceArray = _arr(('nworld', 'nmocap', 3), "float32")
  mocap_quat: DeviceArray = _arr(('nworld', 'nmocap', 4), "float32")
  xpos: DeviceArray = _arr(('nworld', 'nbody', 3), "float32")
  xquat: DeviceArray = _arr(('nworld', 'nbody', 4), "float32")
  xmat: DeviceArray = _arr(('nworld', 'nbody', 3, 3), "float32")
  xipos: DeviceArray = _arr(('nworld', 'nbody', 3), "float32")
  ximat: DeviceArray = _arr(('nworld', 'nbody', 3, 3), "float32")
  xanchor: DeviceArray = _arr(('nworld', 'njnt', 3), "float32")
  xaxis: DeviceArray = _arr(('nworld', 'njnt', 3), "float32")
  geom_xpos: DeviceArray = _arr(('nworld', 'ngeom', 3), "float32")
  geom_xmat: DeviceArray = _arr(('nworld', 'ngeom', 3, 3), "float32")
  site_xpos: DeviceArray = _arr(('nworld', 'nsite', 3), "float32")
  site_xmat: DeviceArray = _arr(('nworld', 'nsite', 3, 3), "float32")
  subtree_com: DeviceArray = _arr(('nworld', 'nbody', 3), "float32")
  cinert: DeviceArray = _arr(('nworld', 'nbody', 10), "float32")
  cdof: DeviceArray = _arr(('nworld', 'nv', 6), "float32")
  crb: DeviceArray = _arr(('nworld', 'nbody', 10), "float32")
  M: DeviceArray = _arr(('nworld', 'nC'), "float32")
  qLD: DeviceArray = _arr(('nworld', 'nC'), "float32")
  qLDiagInv: DeviceArray = _arr(('nworld', 'nv'), "float32")
  actuator_length: DeviceArray = _arr(('nworld', 'nu'), "float32")
  actuator_moment: DeviceArray = _arr(('nworld', 'nu'), "float32")
  actuator_velocity: DeviceArray = _arr(('nworld', 'nu'), "float32")
  cvel: DeviceArray = _arr(('nworld', 'nbody', 6), "float32")
  cdof_dot: DeviceArray = _arr(('nworld', 'nv', 6), "float32")
  qfrc_spring: DeviceArray = _arr(('nworld', 'nv'), "float32")
  qfrc_damper: DeviceArray = _arr(('nworld', 'nv'), "float32")
  qfrc_gravcomp: DeviceArray = _arr(('nworld', 'nv'), "float32")
  qfrc_passive: DeviceArray = _arr(('nworld', 'nv'), "float32")
  qfrc_bias: DeviceArray = _arr(('nworld', 'nv'), "float32")
  cacc: DeviceArray = _arr(('nworld', 'nbody', 6), "float32")
  cfrc_int: DeviceArray = _arr(('nworld', 'nbody', 6), "float32")
  act_dot: DeviceArray = _arr(('nworld', 'na'), "float32")
  actuator_force: DeviceArray = _arr(('nworld', 'nu'), "float32")
  qfrc_actuator: DeviceArray = _arr(('nworld', 'nv'), "float32")
  qfrc_smooth: DeviceArray = _arr(('nworld', 'nv'), "float32")
  qacc_smooth: DeviceArray = _arr(('nworld', 'nv'), "float32")
  qacc: DeviceArray = _arr(('nworld', 'nv'), "float32")
  qfrc_constraint: DeviceArray = _arr(('nworld', 'nv'), "float32")
  solver_niter: DeviceArray = _arr(('nworld',), "int32")
  ne: DeviceArray = _arr(('nworld',), "int32")
  nf: DeviceArray = _arr(('nworld',), "int32")
  nl: DeviceArray = _arr(('nworld',), "int32")
  nefc: DeviceArray = _arr(('nworld',), "int32")
  overflow: DeviceArray = _arr(('nworld',), "int32")
  nacon: DeviceArray = _arr((1,), "int32")
  ncollision: DeviceArray = _arr((1,), "int32")
  ws_ncon: DeviceArray = _arr(('nworld',), "int32")
  ws_conadr: DeviceArray = _arr(('nworld',), "int32")
  ws_ncollision: DeviceArray = _arr(('nworld',), "int32")
  ws_efc_con: DeviceArray = _arr(('nworld', 'njmax'), "int32")
  ws_tree_rowadr: DeviceArray = _arr(('nworld', 'ntreeadr'), "int32")
  ws_tree_rowmap: DeviceArray = _arr(('nworld', 'ntreerow'), "int32")
  ws_isl_dofadr: DeviceArray = _arr(('nworld', 'ntreeadr'), "int32")
  ws_isl_dofmap: DeviceArray = _arr(('nworld', 'ntreedof'), "int32")
  ws_isl_dofinv: DeviceArray = _arr(('nworld', 'ntreedof'), "int32")
  ws_nisland: DeviceArray = _arr(('nworld',), "int32")
  ws_isl_flags: DeviceArray = _arr(('nworld',), "int32")
  ws_isl_list: DeviceArray = _arr((3, 'ntreeworld'), "int32")
  ws_isl_count: DeviceArray = _arr((4,), "int32")
  ws_separable: DeviceArray = _arr(('nworld',), "int32")
  ws_ccd: DeviceArray = _arr(('nccdworld', 'nccdword', 32), "float32")
  ws_order: DeviceArray = _arr(('nworld',), "int32")
  sensordata: DeviceArray = _arr(('nworld', 'nsensordata'), "float32")
  energy: DeviceArray = _arr(('nworld', 2), "float32")
  subtree_linvel: DeviceArray = _arr(('nworld', 'nbody', 3), "float32")
  subtree_angmom: DeviceArray = _arr(('nworld', 'nbody', 3), "float32")
  cfrc_ext: DeviceArray = _arr(('nworld', 'nbody', 6), "float32")
  tree_asleep: DeviceArray = _arr(('nworld', 'ntree'), "int32")  # reference types.py:2330-2345
  tree_awake: DeviceArray = _arr(('nworld', 'ntree'), "int32")
  body_awake: DeviceArray = _arr(('nworld', 'nbody'), "int32")
  body_awake_ind: DeviceArray = _arr(('nworld', 'nbody'), "int32")
  dof_awake_ind: DeviceArray = _arr(('nworld', 'nv'), "int32")
  ntree_awake: DeviceArray = _arr(('nworld',), "int32")
  nbody_awake: DeviceArray = _arr(('nworld',), "int32")
  nv_awake: DeviceArray = _arr(('nworld',), "int32")
  tree_island: DeviceArray = _arr(('nworld', 'ntree'), "int32")
  nisland: DeviceArray = _arr(('nworld',), "int32")
  ws_iacc: DeviceArray = _arr(('nimpworld', 'nv'), "float32")
  ws_pgsB: DeviceArray = _arr(('npgsworld', 'njmax_pad', 'nv_pad'), "float32")
  ws_sleep_J: DeviceArray = _arr(('nsleepworld', 'njmax_pad', 'nv_pad'), "float32")
  ws_sleep_warm: DeviceArray = _arr(('nsleepworld', 'nv'), "float32")
  ws_sleep_flag: DeviceArray = _arr(('nworld',), "int32")
  sleep_pass: int = 0
  nimpworld: int = 0  # leading size of ws_iacc (nworld for the fully implicit integrator, else 0)
  npgsworld: int = 0  # leading size of ws_pgsB (nworld for models solved by the generic PGS kernel, else 0)
  nvmax: int = 0  # capacity for awake dofs per world (reference types.py Data.nvmax): exceeding it raises OverflowType.NVMAX
  nsleepworld: int = 0
  eq_active: DeviceArray = _arr(('nworld', 'neq'), "int32")
  ws_rk: DeviceArray = _arr(('nworld', 'nq+3*nv+2*na'), "float32")
  ws_contact: DeviceArray = _arr(('nworld', 'concap', 32), "float32")
  nworld: int = 0
  nconmax: int = 0
  naconmax: int = 0
  njmax: int = 0
  njmax_pad: int = 0
  nv_pad: int = 0
  nmaxpyramid: int = 0
  nccdworld: int = 0  # worlds with an EPA workspace: nworld if the model has convex (GJK) pairs, else 0
  nccdword: int = 0  # workspace words per lane of a world such that the array holds csrc/convex.hpp ccd_layout(...).total floats
  world_offset: int = 0
  concap: int = 0
  nccdhand: int = 0  # EPA entries the convex narrowphase can hand over per step (csrc/collide.hpp ccd_handcap)
  njmax_nnz: int = 0

