"""Build libmjhip_<tag>.so from the object cache with ONE unit recompiled with extra flags (about a minute instead of nine):
python tools/build_variant_fast.py <tag> <unit, e.g. mjhip.hip> [-DMACRO=value ...]      then  MJH_LIB=mujoco_warp_amd/libmjhip_<tag>.so"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mujoco_warp_amd import _abi
tag, unit, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
_abi.build()  # make sure the cache holds the current objects
cache = os.path.join(ROOT, "build", "objcache")
objs = []
for u in _abi.UNITS:
  if u == unit:
    obj = os.path.join(ROOT, "build", f"{u}.{tag}.o")
    subprocess.check_call(["hipcc", *_abi.HIPCC_FLAGS, *_abi.UNIT_FLAGS.get(u, []), *extra, "-c", "-o", obj, os.path.join(ROOT, "mujoco_warp_amd", "csrc", u)])
  else:
    obj = os.path.join(cache, f"{u}.{_abi._unit_key(u)}.o")
  objs.append(obj)
out = os.path.join(ROOT, "mujoco_warp_amd", f"libmjhip_{tag}.so")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
print(out)
