"""Physics pipeline entry points: thin Python shims over the C ABI (include/mjhip.h).

Same names and `(m, d) -> None` signatures as the reference's stage functions
(/root/reference/mujoco_warp/__init__.py:26-101, _src/forward.py:1341-1380); every call is one launch
sequence on torch's current HIP stream and returns without synchronising, like `wp.launch`.
"""

import ctypes
from typing import Optional

import numpy as np
import torch

from . import _abi
from . import io
from . import types
from .device import DeviceArray

_S = _abi.DEFINES


def _stream():
  if not torch.cuda.is_available():
    raise RuntimeError("mujoco_warp_amd needs an AMD GPU: no HIP device is visible (there is no CPU fallback).")
  return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _run(stage, m, d):
  L = _abi.lib()
  _abi.check(L.mjh_stage(ctypes.byref(io.c_model(m)), ctypes.byref(io.c_data(d)), stage, _stream()))


def step(m: types.Model, d: types.Data):
  """Advance simulation by one step (reference forward.py:1368)."""
  _run(_S["MJH_STAGE_STEP"], m, d)


def forward(m: types.Model, d: types.Data):
  """Forward dynamics (reference forward.py:1341)."""
  _run(_S["MJH_STAGE_FORWARD"], m, d)


def kinematics(m, d):
  _run(_S["MJH_STAGE_KINEMATICS"], m, d)


def com_pos(m, d):
  _run(_S["MJH_STAGE_COM_POS"], m, d)


def crb(m, d):
  _run(_S["MJH_STAGE_CRB"], m, d)


def factor_m(m, d):
  _run(_S["MJH_STAGE_FACTOR_M"], m, d)


def collision(m, d):
  _run(_S["MJH_STAGE_COLLISION"], m, d)


def make_constraint(m, d):
  _run(_S["MJH_STAGE_MAKE_CONSTRAINT"], m, d)


def transmission(m, d):
  _run(_S["MJH_STAGE_TRANSMISSION"], m, d)


def com_vel(m, d):
  _run(_S["MJH_STAGE_COM_VEL"], m, d)


def passive(m, d):
  _run(_S["MJH_STAGE_PASSIVE"], m, d)


def rne(m, d):
  _run(_S["MJH_STAGE_RNE"], m, d)


def fwd_position(m, d):
  _run(_S["MJH_STAGE_FWD_POSITION"], m, d)


def fwd_velocity(m, d):
  _run(_S["MJH_STAGE_FWD_VELOCITY"], m, d)


def fwd_actuation(m, d):
  _run(_S["MJH_STAGE_FWD_ACTUATION"], m, d)


def fwd_acceleration(m, d):
  _run(_S["MJH_STAGE_FWD_ACCELERATION"], m, d)


def solve(m, d):
  _run(_S["MJH_STAGE_SOLVE"], m, d)


def euler(m, d):
  _run(_S["MJH_STAGE_EULER"], m, d)


def implicit(m, d):
  _run(_S["MJH_STAGE_IMPLICIT"], m, d)


def rungekutta4(m, d):
  """Runge-Kutta 4 integration; call after `forward` (reference forward.py:524)."""
  _run(_S["MJH_STAGE_RUNGEKUTTA4"], m, d)


def sensor(m, d):
  """Data.sensordata from the current position / velocity / actuation results (reference sensor.sensor_pos / sensor_vel and the actuator
  forces of sensor_acc, sensor.py:810, 1432, 2512); `forward` and `step` call it themselves."""
  _run(_S["MJH_STAGE_SENSOR"], m, d)


def update_sleep(m, d):
  """Sleep tables (tree_awake, body_awake, awake index lists and counts) from Data.tree_asleep (reference sleep.py:171)."""
  _run(_S["MJH_STAGE_UPDATE_SLEEP"], m, d)


def wake(m, d):
  """Wakes sleeping trees the user touched: velocity, applied forces (reference sleep.py:721), then update_sleep."""
  _run(_S["MJH_STAGE_WAKE"], m, d)


def wake_collision(m, d):
  """Wakes sleeping trees in contact with awake ones (reference sleep.py:744), then update_sleep."""
  _run(_S["MJH_STAGE_WAKE_COLLISION"], m, d)


def wake_equality(m, d):
  """Wakes sleeping trees tied to awake ones by an active equality (reference sleep.py:793), then update_sleep."""
  _run(_S["MJH_STAGE_WAKE_EQUALITY"], m, d)


def island(m, d):
  """Constraint islands at tree granularity: Data.tree_island, Data.nisland (reference island.py:294)."""
  _run(_S["MJH_STAGE_ISLAND"], m, d)


def sleep(m, d):
  """Puts trees below the velocity tolerance to sleep, island by island (reference sleep.py:947), then update_sleep."""
  _run(_S["MJH_STAGE_SLEEP"], m, d)


def fwd_kinematics(m, d):
  """Kinematics-dependent computations (reference forward.py:616; no cameras / flex / tendons here)."""
  kinematics(m, d)
  com_pos(m, d)


def step1(m, d):
  """First half of `step`, before the user sets controls (reference forward.py:1384; no sensors / energy here)."""
  fwd_position(m, d)
  fwd_velocity(m, d)


def step2(m, d):
  """Second half of `step` (reference forward.py:1403): RK4 falls back to Euler, as in the reference."""
  fwd_actuation(m, d)
  fwd_acceleration(m, d)
  solve(m, d)
  if int(m.opt.integrator) in (int(types.IntegratorType.IMPLICITFAST), int(types.IntegratorType.IMPLICIT)):
    implicit(m, d)
  else:
    euler(m, d)


def solve_m(m, d, x: DeviceArray, y: DeviceArray):
  """x = M^-1 y using the stored factor (reference smooth.py:3214)."""
  L = _abi.lib()
  _abi.check(L.mjh_solve_m(ctypes.byref(io.c_model(m)), ctypes.byref(io.c_data(d)), x.ptr, y.ptr, _stream()))


def mul_m(m, d, res: DeviceArray, vec: DeviceArray):
  """res = M vec (reference support.py:218)."""
  L = _abi.lib()
  _abi.check(L.mjh_mul_m(ctypes.byref(io.c_model(m)), ctypes.byref(io.c_data(d)), res.ptr, vec.ptr, _stream()))


def qLD_dense(m, d):
  """Dense Cholesky factors of M's kinematic-tree blocks in the reference's packed `qLD` layout (reference io.py:173-211
  m_block_layout, smooth.py:3256: upper factor U, M = U^T U, n x n row-major per tree).  `Data.qLD` of this engine is MuJoCo's sparse
  L^T D L factor; this is the reference-layout copy on request (call after `factor_m` / `forward`).  Returns (qLD [nworld, total],
  block_adr [nv]: offset of each dof's block, -1 for trees of more than 64 dofs)."""
  nums = m.tree_dofnum.numpy()
  adrs = m.tree_dofadr.numpy()
  block_adr = np.full(m.nv, -1, dtype=np.int32)
  total = 0
  for a, n in zip(adrs, nums):
    if n <= 64:
      block_adr[a : a + n] = total
      total += int(n) * int(n)
  out = DeviceArray.zeros((d.nworld, total), dtype=np.float32)
  L = _abi.lib()
  _abi.check(L.mjh_qld_dense(ctypes.byref(io.c_model(m)), ctypes.byref(io.c_data(d)), out.ptr, total, _stream()))
  return out, block_adr


def contact_force(m, d, contact_ids: DeviceArray, to_world_frame: bool, force: DeviceArray):
  """6D forces of the contacts `contact_ids` (int32 [n]) into `force` ([n, 6] float32): normal, two tangents, spin, two rolls, in the
  contact frame or rotated to world axes (reference support.contact_force, support.py:445)."""
  n = int(np.prod(contact_ids.shape))
  if tuple(force.shape) != (n, 6):
    raise ValueError(f"force must have shape ({n}, 6)")
  L = _abi.lib()
  _abi.check(L.mjh_contact_force(ctypes.byref(io.c_model(m)), ctypes.byref(io.c_data(d)), contact_ids.ptr, n, int(bool(to_world_frame)), force.ptr, _stream()))


def jac(m, d, jacp: Optional[DeviceArray], jacr: Optional[DeviceArray], point: DeviceArray, body: DeviceArray):
  """Translational / rotational Jacobians ([nworld, 3, nv]; either may be None) of `point` ([nworld, 3], world coordinates) moving with
  `body` ([nworld] int32) (reference support.jac, support.py:581)."""
  for j in (jacp, jacr):
    if j is not None and tuple(j.shape) != (d.nworld, 3, m.nv):
      raise ValueError(f"Jacobian outputs must have shape ({d.nworld}, 3, {m.nv})")
  L = _abi.lib()
  _abi.check(L.mjh_jac(ctypes.byref(io.c_model(m)), ctypes.byref(io.c_data(d)), jacp.ptr if jacp is not None else None,
                       jacr.ptr if jacr is not None else None, point.ptr, body.ptr, _stream()))


def rne_postconstraint(m, d):
  """Data.cacc / cfrc_int / cfrc_ext with the constraint forces in (reference smooth.rne_postconstraint, smooth.py:1744); call after the
  solver.  `forward` runs it when a force / torque sensor reads the result."""
  _run(_S["MJH_STAGE_RNE_POSTCONSTRAINT"], m, d)


def subtree_vel(m, d):
  """Data.subtree_linvel / subtree_angmom: velocity of every subtree's centre of mass and its angular momentum about it (reference
  smooth.subtree_vel, smooth.py:3614); call after fwd_velocity.  `forward` runs it when a sensor reads the result."""
  _run(_S["MJH_STAGE_SUBTREE_VEL"], m, d)


def energy_pos(m, d):
  """Data.energy = (potential, kinetic) from the current kinematics, M and qvel (reference sensor.energy_pos / energy_vel, sensor.py:2934,
  3003: one launch computes both here).  `forward` / `step` call it when EnableBit.ENERGY is set."""
  _run(_S["MJH_STAGE_ENERGY"], m, d)


def energy_vel(m, d):
  """See energy_pos (both components are computed together)."""
  _run(_S["MJH_STAGE_ENERGY"], m, d)


def sensor_pos(m, d):
  """Position-stage sensors (reference sensor.sensor_pos, sensor.py:810).  This engine computes the position and velocity stages (and
  Data.energy) in one launch: sensor_pos and sensor_vel both run it; call them after fwd_position / fwd_velocity.  Nothing of the
  acceleration stage runs (no rne_postconstraint on stale forces, cacc / cfrc_* untouched)."""
  _run(_S["MJH_STAGE_SENSOR_POSVEL"], m, d)


def sensor_vel(m, d):
  """Velocity-stage sensors (reference sensor.sensor_vel, sensor.py:1432); see sensor_pos."""
  _run(_S["MJH_STAGE_SENSOR_POSVEL"], m, d)


def sensor_acc(m, d):
  """Acceleration-stage sensors only (reference sensor.sensor_acc, sensor.py:2512): accelerometer, force / torque / touch (after an
  rne_postconstraint launch when one of them is present), frame accelerations, actuator forces; call after the solver."""
  _run(_S["MJH_STAGE_SENSOR_ACC"], m, d)


def efc_J_sparse(m, d, njmax_nnz: int = None):
  """CSR copy of `d.efc.J` in the reference's sparse layout (reference types.py:2021-2070; the reference stores efc.J like this for
  nv > 32, io.py:1804-1808; this engine keeps the dense tile and offers the CSR form on request).  Returns
  (J_rownnz [nworld, njmax], J_rowadr [nworld, njmax], J_colind [nworld, 1, njmax_nnz], J [nworld, 1, njmax_nnz])."""
  nnz = int(d.njmax_nnz if njmax_nnz is None else njmax_nnz)
  rownnz = DeviceArray.zeros((d.nworld, d.njmax), dtype=np.int32)
  rowadr = DeviceArray.zeros((d.nworld, d.njmax), dtype=np.int32)
  colind = DeviceArray.zeros((d.nworld, 1, nnz), dtype=np.int32)
  vals = DeviceArray.zeros((d.nworld, 1, nnz), dtype=np.float32)
  L = _abi.lib()
  _abi.check(L.mjh_efc_j_sparse(ctypes.byref(io.c_model(m)), ctypes.byref(io.c_data(d)), nnz, rownnz.ptr, rowadr.ptr, colind.ptr, vals.ptr, _stream()))
  return rownnz, rowadr, colind, vals


def ctrl_noise(m, d, step_index: int, noise_std: float = 0.01, noise_rate: float = 0.1, center: DeviceArray = None):
  """Ornstein-Uhlenbeck/Halton control noise of the benchmark harness (reference cli.py:103-145)."""
  L = _abi.lib()
  _abi.check(L.mjh_ctrl_noise(ctypes.byref(io.c_model(m)), ctypes.byref(io.c_data(d)), center.ptr if center is not None else None,
                              int(step_index), float(noise_std), float(noise_rate), _stream()))


class StepGraph:
  """hipGraph of one `step` (the reference captures step in a CUDA graph: cli.py:262-265)."""

  def __init__(self, m, d):
    self._m, self._d = m, d  # keep the buffers alive
    self._exec = ctypes.c_void_p()
    L = _abi.lib()
    _stream()  # raises without a GPU
    # stream capture is illegal on the legacy default stream: capture on a side stream ordered after the current one
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    # mjh_graph_create runs one real step before capturing (kernel attributes cannot be set during capture): keep the state,
    # constructing a graph must not advance the simulation (the reference's capture does not either)
    names = ["qpos", "qvel", "act", "ctrl", "time", "qacc_warmstart", "qacc", "solver_niter", "overflow", "sensordata", "energy", "act_dot"]
    # sleeping models: the warm-up step would advance the sleep counters / wake state / island tables by one step
    names += ["tree_asleep", "tree_awake", "body_awake", "body_awake_ind", "dof_awake_ind", "ntree_awake", "nbody_awake", "nv_awake", "tree_island", "nisland"]
    keep = {k: getattr(d, k).t.clone() for k in names if getattr(d, k, None) is not None and getattr(d, k).size}
    rc = L.mjh_graph_create(ctypes.byref(io.c_model(m)), ctypes.byref(io.c_data(d)), ctypes.c_void_p(side.cuda_stream), ctypes.byref(self._exec))
    torch.cuda.current_stream().wait_stream(side)
    for k, v in keep.items():
      getattr(d, k).t.copy_(v)
    _abi.check(rc)

  def launch(self):
    _abi.check(_abi.lib().mjh_graph_launch(self._exec, _stream()))

  def __del__(self):
    try:
      if self._exec:
        _abi.lib().mjh_graph_destroy(self._exec)
    except Exception:
      pass


def solver_kernel(m, d) -> str:
  """Name of the solver mapping the library's dispatch picks for (m, d) in a fused step (mjh_solver_kernel): "cgp", "cgw", "pair",
  "newton_mfma", ... -- what a test or a bench line measured, without a developer knob."""
  return _abi.lib().mjh_solver_kernel(ctypes.byref(io.c_model(m)), ctypes.byref(io.c_data(d))).decode()


def timed_steps(m, d, nstep: int, step0: int = 0, noise_std: float = 0.01, noise_rate: float = 0.1, per_kernel: bool = False,
                plain_kernels: bool = False):
  """Run nstep x (ctrl_noise + step) bracketed by HIP events on the launch stream.

  Returns (elapsed_ms, per_kernel_ms or None); per-kernel times come from event pairs around each launch
  and are only meaningful for profiling (the extra events perturb the total slightly).  per_kernel times the four
  launches of the fused step (KERNEL_NAMES: ctrl_noise, fwd_pos, mid, solve, integrate); with plain_kernels the step
  runs one plain kernel per stage instead, which gives the per-stage trace of the reference's event tracer.
  """
  L = _abi.lib()
  ms = ctypes.c_float(0.0)
  pk = (ctypes.c_float * _S["MJH_NKERNEL"])() if per_kernel else None
  _abi.check(L.mjh_timed_steps(ctypes.byref(io.c_model(m)), ctypes.byref(io.c_data(d)), int(nstep), int(step0), float(noise_std),
                               float(noise_rate), _stream(), ctypes.byref(ms), pk, int(bool(plain_kernels))))
  return ms.value, (list(pk) if per_kernel else None)


KERNEL_NAMES = ["ctrl_noise", "fwd_pos", "collision", "make_constraint", "fwd_vel", "solve", "integrate", "other", "mid"]
