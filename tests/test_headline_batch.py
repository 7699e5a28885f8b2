"""The headline kernel at the headline size, through the PRODUCT dispatch (no developer knob): humanoid.xml, 8192 worlds driven apart
by control noise, `opt.solver = cg` -- the pooled contact-basis CG kernel (csrc/solver_cgp.hpp) is what `launch_solve_any` picks there
(`mjw.solver_kernel` says so).

  * sampled worlds against the float64 oracle: a world's (qpos, qvel, ctrl, qacc_warmstart) copied into the oracle, one step each, the
    bounds of tests/test_cgp.py::test_cgp_per_step_parity_resynced -- in the free-fall window (step 40) and on the floor (step 300);
  * bitwise shard invariance at this size: 8192 worlds as ONE Data == two Data of 4096 with world_offset 0 / 4096 (the solver's schedule,
    the wavefront pairing and the row pools of a workgroup all differ between the two layouts);
  * bitwise run-to-run determinism with CG.

Reference: forward.py:1368 (step), solver.py:3283-3450 (CG); BASELINE.json configs[1]."""

import numpy as np
import pytest

import conftest
import mujoco_warp_amd as mjw
from conftest import relerr
from oracle import ref

pytestmark = pytest.mark.gpu

NWORLD = 8192
STATE = ("qpos", "qvel", "ctrl", "qacc_warmstart")


def _cg_model():
  mjm = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  mjm.opt.solver = int(mjw.SolverType.CG)
  return mjm


def _rollout(m, d, first, last):
  for i in range(first, last):
    mjw.ctrl_noise(m, d, i)
    mjw.step(m, d)


def test_product_dispatch_picks_the_pooled_kernel_at_8192_worlds_only():
  mjm = _cg_model()
  m = mjw.put_model(mjm)
  assert mjw.solver_kernel(m, mjw.make_data(mjm, nworld=NWORLD, nconmax=24, njmax=64)) == "cgp"
  assert mjw.solver_kernel(m, mjw.make_data(mjm, nworld=4096, nconmax=24, njmax=64)) == "cgp"
  assert mjw.solver_kernel(m, mjw.make_data(mjm, nworld=1024, nconmax=24, njmax=64)) == "cgw"  # (configs[3]-sized shards: one world per wavefront)
  mjm.opt.solver = int(mjw.SolverType.NEWTON)
  assert mjw.solver_kernel(mjw.put_model(mjm), mjw.make_data(mjm, nworld=NWORLD, nconmax=24, njmax=64)) == "newton_mfma"


def test_pooled_cg_at_8192_diverged_worlds_against_the_oracle():
  mjm = _cg_model()
  m = mjw.put_model(mjm)
  d = mjw.make_data(mjm, nworld=NWORLD, nconmax=24, njmax=64)
  assert mjw.solver_kernel(m, d) == "cgp"
  mjw.reset_data_keyframe(m, d, 0)
  s = ref.RefSim(mjm, nconmax=24, njmax=64, tolerance=max(mjm.opt.tolerance, 1e-6))
  rng = np.random.default_rng(6)
  step = 0
  for at in (40, 300):
    _rollout(m, d, step, at)
    step = at
    mjw.ctrl_noise(m, d, step)  # (the control of the compared step: the oracle takes it from the device)
    before = {f: getattr(d, f).numpy().copy() for f in STATE}
    assert len(np.unique(before["qpos"][:, 2])) > NWORLD // 2  # the worlds really are in different states
    mjw.step(m, d)
    step += 1
    qpos, qvel, nefc, niter = d.qpos.numpy(), d.qvel.numpy(), d.nefc.numpy(), d.solver_niter.numpy()
    assert (niter >= 0).all() and (d.overflow.numpy() == 0).all()
    # 32 worlds: the extremes of the row count and of the iteration count (the worlds a pool or a schedule would get wrong first) + random ones
    picks = {int(np.argmax(nefc)), int(np.argmin(nefc)), int(np.argmax(niter)), int(np.argmin(niter)), 0, NWORLD - 1}
    while len(picks) < 32:
      picks.add(int(rng.integers(NWORLD)))
    worst_q = worst_v = 0.0
    it_dev, it_ref = [], []
    for w in sorted(picks):
      s.reset(key=0)
      for f in STATE:
        getattr(s, f)[:] = before[f][w]
      s.step()
      assert s.nefc == int(nefc[w]), (at, w, s.nefc, int(nefc[w]))
      # (float32 CG against float64 CG at tolerance 1e-6: the stopping iterate scatters world by world -- tests/test_cgp.py measured up to 31 apart
      # between two float32 kernels -- most of all on the longest solves, which are among the picks on purpose; no systematic difference allowed)
      assert abs(int(niter[w]) - s.solver_niter) <= max(8, 0.4 * s.solver_niter), (at, w, int(niter[w]), s.solver_niter)
      it_dev.append(int(niter[w]))
      it_ref.append(int(s.solver_niter))
      worst_q = max(worst_q, relerr(qpos[w], s.qpos))
      # qvel relative to max(1, max|qvel|): on the floor (step 300) the humanoid is nearly at rest -- max|qvel| ~ 0.1 rad/s -- and CG's stopping
      # rule leaves ~1e-4 rad/s open in ABSOLUTE terms whatever the velocity (float64 oracle at tolerance 1e-6 vs the same oracle converged:
      # 5e-5 .. 1.5e-4 absolute = 4e-4 .. 1.7e-3 relative there; its float32 twin vs the oracle up to 2.3e-4 absolute, measured)
      worst_v = max(worst_v, float(np.abs(qvel[w].astype(np.float64) - s.qvel).max() / max(1.0, np.abs(s.qvel).max())))
    # (the bounds of test_cgp_per_step_parity_resynced; what the float64 reference CG leaves open at tolerance 1e-6 is 2.9e-4 in the
    # free-fall window: profiles/round6_precision_split.txt)
    assert abs(np.mean(it_dev) - np.mean(it_ref)) <= 0.1 * np.mean(it_ref) + 1.0, (at, np.mean(it_dev), np.mean(it_ref))
    assert worst_q <= 2e-6, (at, worst_q)
    assert worst_v <= 4e-4, (at, worst_v)


def test_pooled_cg_is_bitwise_shard_invariant_and_deterministic_at_8192_worlds():
  mjm = _cg_model()
  m = mjw.put_model(mjm)
  whole = mjw.make_data(mjm, nworld=NWORLD, nconmax=24, njmax=64)
  again = mjw.make_data(mjm, nworld=NWORLD, nconmax=24, njmax=64)
  halves = [mjw.make_data(mjm, nworld=NWORLD // 2, nconmax=24, njmax=64) for _ in range(2)]
  halves[1].world_offset = NWORLD // 2
  assert all(mjw.solver_kernel(m, x) == "cgp" for x in [whole, again] + halves)
  for x in [whole, again] + halves:
    mjw.reset_data_keyframe(m, x, 0)
  # 120 steps: free fall, landing, first steps on the floor (worlds of 8 .. 60 rows side by side in one workgroup's pool)
  for x in [whole, again] + halves:
    _rollout(m, x, 0, 120)
  for f in ("qpos", "qvel", "qacc", "ctrl", "qacc_warmstart"):
    a = getattr(whole, f).numpy()
    assert (a == getattr(again, f).numpy()).all(), f  # run to run
    assert (a == np.concatenate([getattr(h, f).numpy() for h in halves])).all(), f  # one Data vs two shards
  assert (whole.solver_niter.numpy() == np.concatenate([h.solver_niter.numpy() for h in halves])).all()
  assert len(np.unique(whole.qpos.numpy(), axis=0)) > NWORLD // 2
  nefc = whole.nefc.numpy()
  assert nefc.max() - nefc.min() >= 8  # different row counts in the batch
