"""GPU parity tests of the pooled contact-basis CG kernel (csrc/solver_cgp.hpp), forced at small batch sizes through the developer knob
MJH_CG_KERNEL (set through the library's test hook mjh_dev_knob; tests/test_headline_batch.py runs the same kernel through the product
dispatch at 8192 worlds): against the float64 oracle, against the two-worlds-per-wavefront kernel it
replaces (csrc/solver.hpp), and through its fallback launch (worlds it flags solver_niter = -1)."""

import numpy as np
import pytest

import conftest
import mujoco_warp_amd as mjw
from conftest import relerr
from oracle import ref
from test_gpu import _check_solution, _pair, _sync

pytestmark = pytest.mark.gpu


from mujoco_warp_amd._abi import dev_knobs as _knob  # (the library's one test hook, mjh_dev_knob: it never reads the environment after load)


def _cg_humanoid(nworld, warm_steps=15, **kw):
  mjm = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  s, m, d = _pair(mjm, nworld=nworld, nconmax=24, njmax=64, solver=int(mjw.SolverType.CG), warm_steps=warm_steps, **kw)
  assert m.cg_basis == 1
  return mjm, s, m, d


@pytest.mark.parametrize("nworld,threads", [(3, 192), (16, 512), (37, 384), (70, 768)])
def test_cgp_forward_matches_oracle_and_the_paired_kernel(nworld, threads):
  """The same state through the pooled kernel (whole and partial workgroups, every workgroup size) and through k_solve<cg>: both
  inside the oracle tolerance, the same iteration count, and the same answer to the solver tolerance."""
  mjm, s, m, d = _cg_humanoid(nworld)
  s.forward()
  out = {}
  for kernel in ("cgp", "pair"):
    with _knob(MJH_CG_KERNEL=kernel, MJH_CGP_THREADS=threads):
      d.qacc.fill_(float("inf"))
      d.solver_niter.fill_(-7)
      mjw.forward(m, d)
    _check_solution(s, d)
    q = d.qacc.numpy()
    assert (q == q[0]).all()  # the same inputs in every world: bitwise the same outputs
    assert (d.solver_niter.numpy() == d.solver_niter.numpy()[0]).all()
    assert abs(int(d.solver_niter.numpy()[0]) - s.solver_niter) <= 3  # (float32 CG against the float64 oracle: measured 18 / 21)
    out[kernel] = (q[0].copy(), d.efc.force.numpy()[0].copy(), d.efc.state.numpy()[0].copy(), int(d.solver_niter.numpy()[0]))
  assert relerr(out["cgp"][0], out["pair"][0]) <= 2e-4
  assert relerr(out["cgp"][1][: s.nefc], out["pair"][1][: s.nefc]) <= 2e-3
  assert (out["cgp"][2][: s.nefc] == out["pair"][2][: s.nefc]).mean() >= 0.95
  assert abs(out["cgp"][3] - out["pair"][3]) <= 4  # (float32 CG: the two kernels sum in different orders and may stop an iteration or three apart)


def test_cgp_per_step_parity_resynced():
  """North-star parity through the pooled kernel and its fused Euler epilogue: one step from the oracle's state, 150 steps along its
  trajectory (key 0: free fall, first contacts, the squat) -- the bounds of test_gpu.test_per_step_parity_resynced[CG]."""
  mjm, s, m, d = _cg_humanoid(2, warm_steps=0)
  worst_q = worst_v = 0.0
  with _knob(MJH_CG_KERNEL="cgp"):
    for i in range(150):
      s.ctrl_noise(i, 0)
      _sync(s, d)
      mjw.step(m, d)
      s.step()
      worst_q = max(worst_q, relerr(d.qpos.numpy()[1], s.qpos))
      worst_v = max(worst_v, relerr(d.qvel.numpy()[1], s.qvel))
  assert worst_q <= 2e-6, worst_q
  assert worst_v <= 4e-4, worst_v
  assert (d.overflow.numpy() == 0).all()


def test_cgp_worlds_in_different_states_against_the_paired_kernel():
  """64 worlds driven apart by control noise (different contact sets and row counts in one workgroup's pool): every step both kernels
  start from the same states and must agree world by world."""
  mjm = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  mjm.opt.solver = int(mjw.SolverType.CG)
  m = mjw.put_model(mjm)
  da = mjw.make_data(mjm, nworld=64, nconmax=24, njmax=64)
  db = mjw.make_data(mjm, nworld=64, nconmax=24, njmax=64)
  mjw.reset_data_keyframe(m, da, 0)
  seen = set()
  for i in range(400):
    mjw.ctrl_noise(m, da, i, 0.3, 0.1)
    if i % 20 == 0:
      for name in ("qpos", "qvel", "ctrl", "qacc_warmstart", "time"):
        getattr(db, name).assign(getattr(da, name).numpy())
      da.overflow.zero_()
      db.overflow.zero_()
      with _knob(MJH_CG_KERNEL="pair"):
        mjw.forward(m, db)
      with _knob(MJH_CG_KERNEL="cgp", MJH_CGP_THREADS=256):
        mjw.forward(m, da)
      nefc = da.nefc.numpy()
      seen |= set(int(x) for x in nefc)
      assert (nefc == db.nefc.numpy()).all()
      qa, qb = da.qacc.numpy(), db.qacc.numpy()
      errs = np.array([relerr(qa[w], qb[w]) for w in range(64)])
      # (both stop at the solver tolerance, possibly iterations apart; in the violent first-contact states -- control noise 0.3 -- one world
      # in a few hundred ends 4e-3 apart, measured)
      assert np.median(errs) <= 2e-4 and errs.max() <= 1e-2, (i, float(np.median(errs)), float(errs.max()), int(errs.argmax()))
      # (iteration counts of two float32 CG runs with different summation orders scatter widely world by world -- measured up to 31 apart at
      # tolerance 1e-6 -- but must not differ systematically)
      na, nb_ = da.solver_niter.numpy().mean(), db.solver_niter.numpy().mean()
      assert abs(na - nb_) <= 0.15 * nb_ + 1.0, (i, na, nb_)
      # (the violent states of this test run some worlds into the iteration cap -- whichever kernel: no more of them with the pooled one)
      capped_a, capped_b = (da.overflow.numpy() & 512) != 0, (db.overflow.numpy() & 512) != 0
      assert (capped_a & ~capped_b).sum() <= 2 and ((da.overflow.numpy() & ~(512 | 1024)) == 0).all(), (i, int(capped_a.sum()), int(capped_b.sum()))
    with _knob(MJH_CG_KERNEL="cgp", MJH_CGP_THREADS=256):
      mjw.step(m, da)
  assert len(seen) >= 4, seen  # (the batch really held worlds with different row counts)
  assert np.isfinite(da.qpos.numpy()).all()


def test_cgp_pool_overflow_goes_to_the_fallback_launch():
  """A pool too small for its workgroup's worlds (developer knob MJH_CGP_LDS): the worlds that do not fit are flagged solver_niter = -1
  by the pooled kernel and solved by k_solve<cg> in the launch behind it -- every world ends inside the oracle tolerance."""
  mjm, s, m, d = _cg_humanoid(24)
  s.forward()
  n6 = int((s.efc_type[: s.nefc] == 6).sum())
  nb = n6 // 4 * 3 + s.nefc - n6
  lds = 4 * (32 + 136 * 8 + 28 * (2 * ((nb + 15) // 16 * 16) + 4))  # room for two worlds of this state (and not three) per workgroup of 8
  with _knob(MJH_CG_KERNEL="cgp", MJH_CGP_THREADS=256, MJH_CGP_LDS=lds):
    mjw.forward(m, d)
  niter = d.solver_niter.numpy()
  assert (niter >= 0).all()
  for w in range(d.nworld):
    _check_solution(s, d, w=w)
  q = d.qacc.numpy()
  assert len({q[w].tobytes() for w in range(d.nworld)}) == 2  # two kernels took part
  with _knob(MJH_CG_KERNEL="cgp", MJH_CGP_THREADS=256, MJH_CGP_LDS=lds):
    for _ in range(5):  # and through the fused step: both kernels integrate their worlds
      mjw.step(m, d)
  qp = d.qpos.numpy()
  assert relerr(qp, np.tile(qp[0], (d.nworld, 1))) <= 1e-5
  assert (d.time.numpy() == d.time.numpy()[0]).all()


def test_cgp_friction_loss_rows_are_left_to_the_fallback():
  """Friction-loss rows (three-zone cost) are outside the pooled kernel: such worlds take the fallback launch and match the oracle."""
  xml = open(conftest.HUMANOID_XML).read().replace('<joint name="abdomen_z"', '<joint frictionloss="0.4" name="abdomen_z"')
  assert "frictionloss" in xml
  mjm = mjw.mjcf.from_xml_string(xml)
  s, m, d = _pair(mjm, nworld=5, nconmax=24, njmax=64, solver=int(mjw.SolverType.CG))
  # a model WITH friction loss is not dispatched to the pooled kernel at all (every world would take the fallback launch: ADVICE round 5) ...
  assert m.cg_basis == 0
  assert mjw.solver_kernel(m, mjw.make_data(mjm, nworld=8192, nconmax=24, njmax=64)) == "pair"
  s.forward()
  assert s.nf == 1
  # ... and the kernel's own guard (a world that shows up with nf > 0, e.g. frictionloss edited after put_model) still hands it over
  m.cg_basis = 1
  m._dirty = True
  with _knob(MJH_CG_KERNEL="cgp"):
    assert mjw.solver_kernel(m, d) == "cgp"
    mjw.forward(m, d)
  assert (d.solver_niter.numpy() >= 0).all()
  _check_solution(s, d)


@pytest.mark.parametrize("nworld", [700, 9000])
def test_solver_schedule_is_a_permutation_sorted_by_the_previous_iteration_count(nworld):
  """k_fwd_pos_plus' schedule workgroup (csrc/integrate.hpp schedule_body): one batch of register-resident keys up to 32 x 256 worlds, the
  looped path beyond -- either way Data.ws_order is a permutation of the worlds, longest previous solve first."""
  mjm = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  mjm.opt.solver = int(mjw.SolverType.CG)
  m = mjw.put_model(mjm)
  d = mjw.make_data(mjm, nworld=nworld, nconmax=24, njmax=64)
  mjw.reset_data_keyframe(m, d, 0)
  for i in range(12):
    mjw.ctrl_noise(m, d, i, 0.5, 0.1)
    mjw.step(m, d)
  prev = d.solver_niter.numpy().copy()
  mjw.step(m, d)
  order = d.ws_order.numpy()
  assert sorted(order.tolist()) == list(range(nworld))
  keys = np.minimum(prev[order], 127)
  assert (np.diff(keys) <= 0).all()
  assert len(set(prev.tolist())) > 3  # (the worlds really differed)


def _snake_xml(nlink):
  """A free-floating snake of `nlink` hinge links lying on the floor (nv = 6 + nlink): every third link's sphere (condim 3) touches the plane,
  so that the rows stay inside njmax = 64."""
  lines = ['<mujoco><option timestep="0.003" solver="CG"/><default><geom condim="3" friction="0.9 0.02 0.001"/><joint armature="0.01" damping="0.05"/></default>',
           '<worldbody><geom name="floor" type="plane" size="0 0 .05"/><body name="l0" pos="0 0 .05"><freejoint/><geom type="sphere" size=".05"/>']
  for k in range(nlink):
    ax = "0 1 0" if k % 2 else "0 0 1"
    touch = "" if (k + 1) % 3 == 0 else ' contype="0" conaffinity="0"'
    lines.append(f'<body name="l{k + 1}" pos=".11 0 0"><joint type="hinge" axis="{ax}" range="-40 40" limited="true"/><geom type="sphere" size=".05"{touch}/>')
  lines.append("</body>" * (nlink + 1) + "</worldbody></mujoco>")
  return "\n".join(lines)


@pytest.mark.parametrize("nlink", [14, 18, 22, 26])
def test_cgp_other_widths(nlink):
  """The instantiations of the pooled kernel the humanoid does not use: nv = 20, 24, 28, 32 (NV4 = 5 .. 8; the last two at the register
  budget) -- a snake settling on the floor, per-step parity with the oracle through the pooled kernel and its fused Euler epilogue."""
  mjm = mjw.mjcf.from_xml_string(_snake_xml(nlink))
  assert mjm.nv == 6 + nlink
  s, m, d = _pair(mjm, nworld=2, nconmax=32, njmax=64, solver=int(mjw.SolverType.CG), warm_steps=0)
  assert m.cg_basis == 1
  worst_q = worst_v = 0.0
  rows = set()
  mixed = lambda a, b: float(np.abs(np.asarray(a, dtype=np.float64) - b).max() / max(1.0, np.abs(b).max()))  # (the snake comes to rest: velocities of 1e-4)
  with _knob(MJH_CG_KERNEL="cgp"):
    for i in range(120):
      _sync(s, d)
      mjw.step(m, d)
      s.step()
      rows.add(int(s.nefc))
      worst_q = max(worst_q, mixed(d.qpos.numpy()[1], s.qpos))
      worst_v = max(worst_v, mixed(d.qvel.numpy()[1], s.qvel))
  assert 0 < max(rows) <= 64, rows
  assert worst_q <= 5e-6, worst_q
  assert worst_v <= 1e-3, worst_v
  assert ((d.overflow.numpy() & ~(512 | 1024)) == 0).all()
