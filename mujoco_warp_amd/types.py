"""Enums and the Model/Data containers of the drop-in boundary.

Mirrors the hot-path subset of /root/reference/mujoco_warp/_src/types.py: enums 119-545, Option 836,
Statistic 922, Model 982-1960, Contact 1975, Constraint 2021, Data 2075-2374.  Field names, shapes and
dtypes (float32 / int32, world-major) are the reference's; fields that belong to out-of-scope subsystems
(tendons, flex, sensors, rendering, islands) are absent.  Arrays are `DeviceArray`s (device.py).

Deviation (documented in DESIGN.md): `Data.qLD` holds MuJoCo's sparse L'DL factor in the CSR M-structure
([nworld, nC], plus `qLDiagInv`), not the reference's packed dense per-tree Cholesky blocks.
"""

import enum


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


class _Dirty:
  """Attribute container that remembers when a field was re-bound (so the C struct is rebuilt lazily)."""

  def __setattr__(self, name, value):
    object.__setattr__(self, name, value)
    if not name.startswith("_"):
      object.__setattr__(self, "_dirty", True)
      root = self.__dict__.get("_root")
      if root is not None:
        object.__setattr__(root, "_dirty", True)


class Option(_Dirty):
  """Physics options (reference types.py:836-905)."""


class Statistic(_Dirty):
  """Model statistics (reference types.py:922-931)."""


class Model(_Dirty):
  """Device-resident model (reference types.py:982-1960, hot-path subset)."""


class Contact(_Dirty):
  """Contact arrays, flat over all worlds (reference types.py:1975-2018)."""


class Constraint(_Dirty):
  """Constraint arrays (reference types.py:2021-2072); J is dense [nworld, njmax_pad, nv_pad]."""


class Data(_Dirty):
  """Device-resident, batched simulation state (reference types.py:2075-2374, hot-path subset)."""
