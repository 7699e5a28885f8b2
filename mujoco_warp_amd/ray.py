"""Ray casting against the primitive geoms (reference ray.py:1172 ray, 1219 rays): host mirror over mjh_rays (csrc/ray.hpp).

Mesh and height-field geoms are not intersected: `rays()` and `put_model` (for a rangefinder sensor) raise NotImplementedError on a model
that has such geoms with a visible colour (the reference's ray elimination, ray.py:52) instead of silently reporting "no hit"; there is
no BVH / render context."""

import ctypes
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _abi
from . import io
from .device import DeviceArray
from .forward import _stream


def rays(m, d, pnt: DeviceArray, vec: DeviceArray, geomgroup: Optional[Sequence[float]], flg_static: bool, bodyexclude: Optional[DeviceArray],
         dist: DeviceArray, geomid: Optional[DeviceArray], normal: Optional[DeviceArray], rc=None):
  """Nearest intersection of `nray` rays per world with the geoms of `d` (geom_xpos / geom_xmat must be current: run kinematics first).

  pnt, vec: [1 or nworld, nray, 3] float32 origins and directions (world frame; distances are in units of |vec|).
  geomgroup: six numbers (-1 six times or None: every group; otherwise geoms whose group's entry is 0 are skipped).
  flg_static: whether geoms of static bodies can be hit.  bodyexclude: [nray] int32 body whose geoms each ray ignores (-1: none), or None.
  dist [nworld, nray] (-1: no hit), geomid [nworld, nray] (-1), normal [nworld, nray, 3] are written; geomid / normal may be None."""
  if rc is not None:
    raise NotImplementedError("render contexts (BVH-accelerated mesh / flex rays) are not part of this engine")
  if getattr(m, "_ray_unsupported_geoms", 0):
    raise NotImplementedError(f"{m._ray_unsupported_geoms} mesh / height-field geom(s) can be hit by rays in this model: rays against them are not "
                              "implemented (they would silently report no hit); give those geoms rgba alpha 0 if rays are meant to pass through them")
  if len(pnt.shape) != 3 or pnt.shape[2] != 3 or tuple(pnt.shape) != tuple(vec.shape):
    raise ValueError(f"pnt {pnt.shape} and vec {vec.shape} must both be [1 or nworld, nray, 3]")
  if pnt.shape[0] not in (1, d.nworld):
    raise ValueError(f"pnt.shape[0] must be 1 or d.nworld ({d.nworld}), got {pnt.shape[0]}")
  nray = pnt.shape[1]
  if tuple(dist.shape) != (d.nworld, nray):
    raise ValueError(f"dist must have shape ({d.nworld}, {nray})")
  if geomid is not None and tuple(geomid.shape) != (d.nworld, nray):
    raise ValueError(f"geomid must have shape ({d.nworld}, {nray})")
  if normal is not None and tuple(normal.shape) != (d.nworld, nray, 3):
    raise ValueError(f"normal must have shape ({d.nworld}, {nray}, 3)")
  if bodyexclude is not None and int(np.prod(bodyexclude.shape)) != nray:
    raise ValueError(f"bodyexclude must have {nray} entries")
  gg = None
  if geomgroup is not None:
    if len(geomgroup) != 6:
      raise ValueError("geomgroup must have six entries")
    gg = (ctypes.c_float * 6)(*[float(x) for x in geomgroup])
  L = _abi.lib()
  _abi.check(L.mjh_rays(ctypes.byref(io.c_model(m)), ctypes.byref(io.c_data(d)), pnt.ptr, vec.ptr, int(pnt.shape[0]), int(nray), gg, int(bool(flg_static)),
                        bodyexclude.ptr if bodyexclude is not None else None, dist.ptr, geomid.ptr if geomid is not None else None,
                        normal.ptr if normal is not None else None, _stream()))


def ray(m, d, pnt: DeviceArray, vec: DeviceArray, geomgroup: Optional[Sequence[float]] = None, flg_static: bool = True, bodyexclude: int = -1,
        rc=None) -> Tuple[DeviceArray, DeviceArray, DeviceArray]:
  """One ray per world (pnt, vec of shape [1 or nworld, 1, 3]): returns (dist [nworld, 1], geomid [nworld, 1], normal [nworld, 1, 3])."""
  if len(pnt.shape) != 3 or pnt.shape[1] != 1:
    raise ValueError(f"ray() takes a single ray per world (shape (*, 1, 3)), got {pnt.shape}; use rays() for several")
  dist = DeviceArray.zeros((d.nworld, 1), np.float32)
  geomid = DeviceArray.zeros((d.nworld, 1), np.int32)
  normal = DeviceArray.zeros((d.nworld, 1, 3), np.float32)
  rays(m, d, pnt, vec, geomgroup, flg_static, DeviceArray.full((1,), int(bodyexclude), np.int32), dist, geomid, normal, rc)
  return dist, geomid, normal
