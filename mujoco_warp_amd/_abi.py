"""ctypes mirror of include/mjhip.h and loader/builder of libmjhip.so.

The header is the single source of truth: the struct layouts are parsed from it, so the Python side can
never drift from the C ABI.  There is NO CPU fallback: if the library is missing or no GPU is present the
engine raises (the product path must fail loudly, never route through oracle/).
"""

import ctypes
import os
import re
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
HEADER = os.path.join(_ROOT, "include", "mjhip.h")
LIB_PATH = os.path.join(_PKG, "libmjhip.so")
# translation units of the library (compiled in parallel; the solver kernels are ~70 template instantiations) and the
# headers they include
UNITS = ["mjhip.hip", "solve_cgp.hip", "solve_cg32.hip", "solve_ell_newton32_r1.hip", "solve_cgw.hip", "solve_newton32.hip", "solve_cg64.hip", "solve_newton64.hip", "solve_ell_cg32.hip", "solve_ell_newton32.hip",
         "solve_ell_cg64.hip", "solve_ell_newton64.hip", "solve_tree_cg.hip", "solve_tree_newton.hip", "solve_tree_ell_cg.hip", "solve_tree_ell_newton.hip", "pgs_tu.hip", "solve_big.hip", "build_id.hip"]
HEADERS = ["host.hpp", "solve_tu.hpp", "solve_tree.hpp", "dev_common.hpp", "smooth.hpp", "collide.hpp", "constraint.hpp", "solver.hpp", "solver_cgp.hpp", "solver_cgw.hpp", "solver_newton.hpp", "solver_big.hpp", "pgs.hpp",
           "integrate.hpp", "implicit.hpp", "pgs_big.hpp", "sleep.hpp", "convex.hpp", "sensor.hpp", "support.hpp", "ray.hpp", "contact_rec.hpp"]
SOURCES = [os.path.join(_PKG, "csrc", f) for f in UNITS + HEADERS]
# -fno-slp-vectorize: the SLP vectoriser packs scalar float ops into v_pk_* pairs, which on gfx950 issue at HALF the rate of the
# scalar forms (tools/ubench.hip: v_pk_fma_f32 4.3 cycles vs v_fma_f32 2.06) and need register pairs plus v_mov shuffles: the Newton
# kernel drops from 239 to 153 VGPRs and loses a third of its moves without it
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-fno-slp-vectorize", "-Wno-unused-value", "-Wno-pass-failed"]
# (hardware division / sqrt in EVERY unit was measured in round 3: humanoid CG 0.3417 -> 0.3369 ms, Newton 0.2821 -> 0.2799 ms (+ 1.4 % / + 0.8 %, two
# interleaved pairs), and one solver parity test moved past its bound (anisotropic elliptic friction, 1.06e-3 against 1e-3): not worth it.  Only
# the PGS unit takes the flag, below.)

# per-unit extra flags.  solve_cg32.hip (the headline's CG kernel: VALU-issue bound, divisions in every line-search step): 215.7 -> 212.0 us per
# launch, 0.3426 / 0.3441 -> 0.3391 / 0.3395 ms per step (two interleaved pairs, same box), the whole GPU suite unchanged.  pgs_tu.hip: the Gauss-Seidel sweeps are one dependent chain per island in which every row visit divides and the
# elliptic blocks' QCQP takes square roots; correctly rounded float32 division / sqrt are ~10-instruction sequences on gfx950, the hardware
# approximations (v_rcp_f32 / v_sqrt_f32 based, <= 2.5 ulp) one or two -- PGS iterates to a tolerance, the parity tests are unaffected
_FAST_DIV = ["-fno-hip-fp32-correctly-rounded-divide-sqrt"]
_NNAN = ["-fno-honor-nans", "-fno-signed-zeros"]  # no NaN / signed-zero bookkeeping around min / max / select chains (the solvers produce neither)
# (_NNAN on mjhip.hip -- kinematics, collision, constraint assembly -- was measured too: no gain, k_mid 84.4 vs 84.0 us)
UNIT_FLAGS = {"pgs_tu.hip": _FAST_DIV + _NNAN, "solve_cg32.hip": _FAST_DIV + _NNAN, "solve_cgp.hip": _FAST_DIV + _NNAN, "solve_cgw.hip": _FAST_DIV + _NNAN, "solve_cg64.hip": _FAST_DIV + _NNAN}
# (machine-scheduler strategies for the CG unit, -mllvm -amdgpu-sched-strategy=...: max-ilp 207 -> 223 us per launch, max-memory-clause 208 -> 214:
# the default stays)
# (-fno-honor-infinities on top: 208.2 vs 207.5 us, noise)
for _u in UNITS:  # every other solver unit: value-preserving for finite data, so the parity figures cannot move
  if _u.startswith("solve_") and _u not in UNIT_FLAGS:
    UNIT_FLAGS[_u] = _NNAN

_CTYPES = {"int": ctypes.c_int, "float": ctypes.c_float, "unsigned int": ctypes.c_uint}


def _parse_struct(text, name):
  body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), text, re.S).group(1)
  body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
  fields = []
  for stmt in body.split(";"):
    stmt = " ".join(stmt.split())
    if not stmt:
      continue
    mm = re.match(r"^(const )?(unsigned int|int|float)\s*(\*?)\s*(\w+)$", stmt)
    if not mm:
      raise ValueError(f"cannot parse declaration '{stmt}' in struct {name}")
    fields.append((mm.group(4), mm.group(2), bool(mm.group(3))))
  return fields


def _parse_defines(text):
  out = {}
  for mm in re.finditer(r"#define (MJH_\w+) (-?\d+)", text):
    out[mm.group(1)] = int(mm.group(2))
  return out


def _parse_functions(text):
  return re.findall(r"^\s*(?:const char\*|int)\s+(mjh_\w+)\(", text, flags=re.M)


_HEADER_TEXT = open(HEADER).read()
MODEL_FIELDS = _parse_struct(_HEADER_TEXT, "MjhModel")
DATA_FIELDS = _parse_struct(_HEADER_TEXT, "MjhData")
DEFINES = _parse_defines(_HEADER_TEXT)
FUNCTIONS = _parse_functions(_HEADER_TEXT)


def _mk(fields):
  return [(n, ctypes.c_void_p if ptr else _CTYPES[k]) for n, k, ptr in fields]


class CModel(ctypes.Structure):
  _fields_ = _mk(MODEL_FIELDS)


class CData(ctypes.Structure):
  _fields_ = _mk(DATA_FIELDS)


def source_build_id():
  """Hash of everything libmjhip.so is built from: every file under csrc/, the header and the compiler flags.  Baked into the library by
  build() (csrc/build_id.hip, mjh_build_id) and compared before every load: no mtime logic, so a pushed tree with fresh timestamps loads the
  library it was built with and a tree whose sources changed never runs old kernels."""
  import hashlib

  h = hashlib.sha256(repr((HIPCC_FLAGS, sorted(UNIT_FLAGS.items()), UNITS)).encode())
  d = os.path.join(_PKG, "csrc")
  for f in sorted(os.listdir(d)) + [HEADER]:
    path = f if os.path.isabs(f) else os.path.join(d, f)
    if os.path.isfile(path) and path.endswith((".hip", ".hpp", ".h")):
      h.update(os.path.basename(path).encode())
      h.update(open(path, "rb").read())
  return h.hexdigest()[:24]


def library_build_id(path=None):
  """The id baked into a built library (read from the file: the marker string of csrc/build_id.hip), None if the file is missing or carries none."""
  path = path or LIB_PATH
  if not os.path.exists(path):
    return None
  mm = re.search(rb"MJH_BUILD_ID=([0-9a-f]{24})", open(path, "rb").read())
  return mm.group(1).decode() if mm else None


def needs_build():
  return library_build_id() != source_build_id()


_TOOLCHAIN = None


def _toolchain_id():
  global _TOOLCHAIN
  if _TOOLCHAIN is None:
    try:
      _TOOLCHAIN = subprocess.run(["hipcc", "--version"], capture_output=True, text=True, timeout=60).stdout.strip()
    except Exception:
      _TOOLCHAIN = "unknown"
  return _TOOLCHAIN


def _unit_key(unit):
  """Hash of everything a translation unit is compiled from: its source, the headers it includes (transitively, quoted includes) and
  the flags.  Keys the object cache below, so that touching one header recompiles only the units that see it."""
  import hashlib

  seen, todo = {}, [os.path.join(_PKG, "csrc", unit)]
  while todo:
    f = os.path.normpath(todo.pop())
    if f in seen or not os.path.exists(f):
      continue
    seen[f] = open(f, "rb").read()
    for inc in re.findall(rb'^\s*#\s*include\s+"([^"]+)"', seen[f], flags=re.M):
      todo.append(os.path.join(os.path.dirname(f), inc.decode()))
  h = hashlib.sha256(" ".join(HIPCC_FLAGS + UNIT_FLAGS.get(unit, []) + ([source_build_id()] if unit == "build_id.hip" else [])).encode())
  h.update(_toolchain_id().encode())  # (a ROCm upgrade must not link objects of the old compiler)
  for f in sorted(seen):
    h.update(f.encode())
    h.update(seen[f])
  return h.hexdigest()[:20]


def build(force=False, verbose=False):
  """Compile libmjhip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).  Objects are cached under build/objcache by the hash
  of their inputs (MJH_NO_OBJCACHE=1 recompiles everything)."""
  if not force and not needs_build():
    return LIB_PATH
  # one builder at a time (N ranks of a multi-GPU launch may all find the library missing); write-then-rename so that a
  # concurrent loader never maps a half-written file
  import fcntl

  with open(LIB_PATH + ".lock", "w") as lock:
    fcntl.flock(lock, fcntl.LOCK_EX)
    if not force and not needs_build():
      return LIB_PATH
    tmp = LIB_PATH + f".tmp{os.getpid()}"
    objdir = os.path.join(_ROOT, "build", f"obj{os.getpid()}")
    os.makedirs(objdir, exist_ok=True)
    import shutil

    cache = os.path.join(_ROOT, "build", "objcache")
    use_cache = not os.environ.get("MJH_NO_OBJCACHE")
    os.makedirs(cache, exist_ok=True)
    procs = []
    for u in UNITS:
      obj = os.path.join(objdir, u + ".o")
      cached = os.path.join(cache, f"{u}.{_unit_key(u)}.o")
      if use_cache and os.path.exists(cached):
        shutil.copyfile(cached, obj)
        if verbose:
          print(f"(cached) {u}")
        continue
      cmd = ["hipcc", *HIPCC_FLAGS, *UNIT_FLAGS.get(u, []), *([f'-DMJH_BUILD_ID="{source_build_id()}"'] if u == "build_id.hip" else []), "-c", "-o", obj, os.path.join(_PKG, "csrc", u)]
      if verbose:
        print(" ".join(cmd))
      procs.append((cmd, subprocess.Popen(cmd), obj, cached))
    failed = None
    for cmd, pr, obj, cached in procs:
      if pr.wait() != 0:
        failed = failed or (pr.returncode, cmd)
      elif use_cache:
        for old in os.listdir(cache):  # one cached object per unit
          if old.startswith(os.path.basename(obj)[:-2] + "."):
            os.remove(os.path.join(cache, old))
        shutil.copyfile(obj, cached + ".tmp")  # (atomic: an interrupted build must not leave a truncated object under a valid key)
        os.replace(cached + ".tmp", cached)
    if failed:
      shutil.rmtree(objdir, ignore_errors=True)  # (objects of a failed build are of no use and pile up otherwise)
      raise subprocess.CalledProcessError(*failed)
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + [os.path.join(objdir, u + ".o") for u in UNITS]
    if verbose:
      print(" ".join(cmd))
    try:
      subprocess.check_call(cmd)
    finally:
      shutil.rmtree(objdir, ignore_errors=True)
    os.replace(tmp, LIB_PATH)
  return LIB_PATH


_lib = None


def lib():
  """Load libmjhip.so (building it if a compiler is available and the sources are newer)."""
  global _lib
  if _lib is not None:
    return _lib
  # MJH_LIB (developer knob, tools/ab.sh): a variant build of the same ABI -- loaded as it is.  Otherwise the library must have been built from
  # the sources on disk: a mismatch is rebuilt when a compiler is there and is an error when not (never a silent run of old kernels).
  if not os.environ.get("MJH_LIB") and needs_build():
    try:
      build()
    except (OSError, subprocess.CalledProcessError) as e:
      raise RuntimeError(f"libmjhip.so does not match the sources (library {library_build_id()}, sources {source_build_id()}) and could not be rebuilt: {e}") from e
    if needs_build():
      raise RuntimeError(f"libmjhip.so still does not match the sources after a rebuild (library {library_build_id()}, sources {source_build_id()})")
  # MJH_LIB: developer knob to A/B two builds of the same ABI in one GPU session (tools/ab.sh)
  L = ctypes.CDLL(os.environ.get("MJH_LIB", LIB_PATH))
  mp, dp, vp = ctypes.POINTER(CModel), ctypes.POINTER(CData), ctypes.c_void_p
  L.mjh_stage.argtypes = [mp, dp, ctypes.c_int, vp]
  L.mjh_step.argtypes = [mp, dp, vp]
  L.mjh_forward.argtypes = [mp, dp, vp]
  L.mjh_solve_m.argtypes = [mp, dp, vp, vp, vp]
  L.mjh_mul_m.argtypes = [mp, dp, vp, vp, vp]
  L.mjh_qld_dense.argtypes = [mp, dp, vp, ctypes.c_int, vp]
  L.mjh_contact_force.argtypes = [mp, dp, vp, ctypes.c_int, ctypes.c_int, vp, vp]
  L.mjh_jac.argtypes = [mp, dp, vp, vp, vp, vp, vp]
  L.mjh_rays.argtypes = [mp, dp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.c_int, vp, vp, vp, vp, vp]
  L.mjh_efc_j_sparse.argtypes = [mp, dp, ctypes.c_int, vp, vp, vp, vp, vp]
  L.mjh_ctrl_noise.argtypes = [mp, dp, vp, ctypes.c_int, ctypes.c_float, ctypes.c_float, vp]
  L.mjh_graph_create.argtypes = [mp, dp, vp, ctypes.POINTER(vp)]
  L.mjh_ws_ccd_floats.argtypes = [ctypes.c_int] * 7 + [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]
  L.mjh_graph_launch.argtypes = [vp, vp]
  L.mjh_graph_destroy.argtypes = [vp]
  L.mjh_timed_steps.argtypes = [mp, dp, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, vp,
                                ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), ctypes.c_int]
  L.mjh_last_error.restype = ctypes.c_char_p
  L.mjh_build_id.restype = ctypes.c_char_p
  L.mjh_dev_knob.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
  L.mjh_solver_kernel.argtypes = [mp, dp]
  L.mjh_solver_kernel.restype = ctypes.c_char_p
  for f in FUNCTIONS:
    if f not in ("mjh_last_error", "mjh_build_id", "mjh_solver_kernel"):
      getattr(L, f).restype = ctypes.c_int
  import atexit

  atexit.register(lambda: L.mjh_release_thread_resources())  # the main thread's side streams (stepping threads call it themselves)
  _lib = L
  return L


class EngineError(RuntimeError):
  pass


def set_knob(name, value):
  """The library's one test hook (mjh_dev_knob, include/mjhip.h): set (str / int) or clear (None) a developer knob of the loaded library.  The
  library never reads the environment after it was loaded, so os.environ has no effect on a running process."""
  check(lib().mjh_dev_knob(name.encode(), None if value is None else str(value).encode()))


class dev_knobs:
  """with dev_knobs(MJH_CG_KERNEL="pair"): ...  -- knobs set for the block (tests, A/B tools), cleared afterwards."""

  def __init__(self, **kv):
    self.kv = kv

  def __enter__(self):
    for k, v in self.kv.items():
      set_knob(k, v)
    return self

  def __exit__(self, *a):
    for k in self.kv:
      set_knob(k, None)


def check(rc):
  if rc != 0:
    msg = lib().mjh_last_error().decode()
    if rc == DEFINES["MJH_E_UNSUPPORTED"]:
      raise NotImplementedError(msg)
    raise EngineError(f"libmjhip error {rc}: {msg}")
