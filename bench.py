"""Headline benchmark: env-steps/sec for humanoid.xml at 8192 worlds (BASELINE.json configs[1]), 1/2/4/8 GPUs.

    python bench.py --gpus N --steps K --warmup W [--solver cg|newton|pgs] [--scaling weak|strong]

N > 1 without a torch.distributed environment: bench.py re-executes itself through
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (one rank per GPU, RCCL);
launched by torchrun it reads RANK / LOCAL_RANK / WORLD_SIZE.  Every rank checks that the group really has N ranks and
that each rank owns a distinct device.

A "step" is one pass of the mj_step hot path (control-noise kernel + step) over a batch of synthetic worlds that start
from keyframe 0 and are decorrelated by the harness' Halton/OU control noise (reference cli.py:103-145, 240-297).  State
is resident in HBM before the timed region starts; the timed region is bracketed by barrier + torch.cuda.synchronize()
on both sides; `value` = all worlds of all ranks x K / MAX elapsed over ranks.  Worlds shard with no data-path
collective; the only collective is the metrics all-reduce after the timed region.
  --scaling weak   (default) every GPU steps its own 8192 worlds  -> "scaling": "weak"
  --scaling strong 8192 worlds in total, shard_worlds(8192, rank, N) per GPU
With N > 1 the line also carries the other mode as `other_scaling`.

Extra objects on the JSON line:
  roofline     -- dominant kernel (the solver launch): SURVEY 8(d) algorithmic HBM bytes per launch / mean launch time from
                  HIP events recorded on the launch stream (instrumented replay of the same K steps); `traffic` is null
                  unless --pmc-profile names a rocprofv3 PMC summary of THIS solver's kernel
  steady_1000  -- the reference's own measurement (testspeed.py:362, cli.py:289-292): 1000 steps from key 0, device sync per
                  step, control noise outside the timed region -- the state the published metric averages over, whatever
                  --steps / --warmup the caller passed
  cpu_baseline -- the float64 oracle ("port": restatement, NOT MuJoCo C) on the host cores, bounded sample
"""

import argparse
import json
import multiprocessing
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
NWORLD = 8192
SOLVERS = {"pgs": 0, "cg": 1, "newton": 2}


def _cpu_worker(args):
  xml, worldid, budget_s, nstep, solver = args
  import mujoco_warp_amd as mjw
  from oracle import ref

  mjm = mjw.mjcf.load_xml(xml)
  mjm.opt.solver = solver
  s = ref.RefSim(mjm, nconmax=24, njmax=64, tolerance=1e-6)
  t0 = time.perf_counter()
  done = r = 0
  while time.perf_counter() - t0 < budget_s:
    s.reset(key=0)
    s.rollout(nstep, worldid=worldid * 1000 + r, record=False)
    done += nstep
    r += 1
  return done, time.perf_counter() - t0


def usable_cores():
  """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
  n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
  try:
    quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
    if quota != "max":
      n = max(1, min(n, int(float(quota) / float(period))))
  except Exception:
    pass
  return n


def cpu_baseline(xml, solver, budget_s=8.0, nstep=500):
  """Oracle ("port") on the host: single-thread rate, then one independent rollout stream per usable core."""
  one = _cpu_worker((xml, 0, 3.0, nstep, solver))
  cores = usable_cores()
  t0 = time.perf_counter()
  with multiprocessing.get_context("fork").Pool(cores) as pool:
    res = pool.map(_cpu_worker, [(xml, w, budget_s, nstep, solver) for w in range(cores)])
  wall = time.perf_counter() - t0
  total = sum(r[0] for r in res)
  return {
    "value": total / max(r[1] for r in res), "unit": "env-steps/s", "cores": cores, "kind": "port",
    "single_thread": one[0] / one[1],
    "sample": f"{cores} processes x ~{budget_s:.0f}s of {nstep}-step humanoid.xml rollouts (key 0 + control noise), float64 oracle "
              f"(restatement of the reference algorithm, NOT MuJoCo C: not installable here or on the GPU box); single-thread figure from a 3 s run; wall {wall:.1f}s",
  }


def _free_port():
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  p = s.getsockname()[1]
  s.close()
  return p


def respawn_under_torchrun(gpus):
  """`python bench.py --gpus N` without a torch.distributed environment: launch N ranks of this script on this node."""
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
         "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
  env = dict(os.environ)
  env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
  return subprocess.call(cmd, env=env)


def check_group(gpus):
  """The process group really has `gpus` ranks and every rank drives a distinct device.  Returns (rank, local_rank, world_size)."""
  import torch
  import torch.distributed as dist

  from mujoco_warp_amd import shard

  rank, local_rank, world_size = shard.init_process_group()
  if world_size != gpus:
    raise SystemExit(f"--gpus {gpus} but the launcher started WORLD_SIZE={world_size} ranks")
  if world_size > 1:
    if not dist.is_initialized() or dist.get_world_size() != gpus:
      raise SystemExit(f"process group has {dist.get_world_size() if dist.is_initialized() else 0} ranks, expected {gpus}")
    sharing = os.environ.get("MJH_DIST_BACKEND") == "gloo"  # developer knob: several ranks on one GPU (RCCL refuses that)
    ndev = torch.cuda.device_count()
    if ndev < gpus and not sharing:
      raise SystemExit(f"--gpus {gpus} but only {ndev} HIP device(s) are visible")
    dev = local_rank % max(ndev, 1)
    ids = [None] * world_size
    dist.all_gather_object(ids, (socket.gethostname(), dev))
    if len(set(ids)) != world_size and not sharing:
      raise SystemExit(f"ranks share devices: {ids}")
  return rank, local_rank, world_size


def measure(mjw, shard, m, mjm, nworld, world_offset, steps, warmup):
  """W untimed warm-up steps, then exactly K timed steps (barrier + synchronize on both sides).  Returns the Data (for the
  replays), a snapshot of the state at the start of the timed window and the per-rank measurement."""
  import torch

  d = mjw.make_data(mjm, nworld=nworld, nconmax=24, njmax=64)
  d.world_offset = world_offset  # global world ids: trajectories independent of the number of GPUs
  mjw.reset_data_keyframe(m, d, 0)
  if warmup:
    mjw.timed_steps(m, d, warmup, step0=0)
  snapshot = {k: getattr(d, k).numpy().copy() for k in ("qpos", "qvel", "ctrl", "qacc_warmstart", "time")}
  shard.barrier()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  ev_ms, _ = mjw.timed_steps(m, d, steps, step0=warmup)
  torch.cuda.synchronize()
  shard.barrier()
  elapsed = time.perf_counter() - t0
  qpos = d.qpos.numpy()
  res = {
    "elapsed": elapsed, "ev_ms": ev_ms, "nan_worlds": float(np.isnan(qpos).any(axis=1).sum()),
    "ovf_worlds": float((d.overflow.numpy() != 0).sum()), "nefc": np.minimum(d.nefc.numpy(), d.njmax), "niter": d.solver_niter.numpy(),
  }
  return d, snapshot, res


def steady_1000(mjw, m, mjm, nworld, world_offset, nstep=1000):
  """The reference's measurement loop (cli.py:270-292): per step, noise kernel + sync untimed, then step + sync timed."""
  import torch

  d = mjw.make_data(mjm, nworld=nworld, nconmax=24, njmax=64)
  d.world_offset = world_offset
  mjw.reset_data_keyframe(m, d, 0)
  mjw.step(m, d)  # function attributes / first-touch outside the loop (the reference captures its graph before timing)
  mjw.reset_data_keyframe(m, d, 0)
  torch.cuda.synchronize()
  total = 0.0
  nefc_sum = niter_sum = 0.0
  for i in range(nstep):
    mjw.ctrl_noise(m, d, i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mjw.step(m, d)
    torch.cuda.synchronize()
    total += time.perf_counter() - t0
    if i % 100 == 99:  # untimed statistics, like the reference's callback
      nefc_sum += float(np.minimum(d.nefc.numpy(), d.njmax).mean())
      niter_sum += float(d.solver_niter.numpy().mean())
  ok = int(nworld - np.isnan(d.qpos.numpy()).any(axis=1).sum())
  return {"value": nworld * nstep / total, "unit": "env-steps/s", "nstep": nstep, "ms_per_step": 1e3 * total / nstep,
          "nefc_mean": nefc_sum / (nstep // 100), "solver_niter_mean": niter_sum / (nstep // 100), "converged_worlds": ok,
          "timing": "reference placement: per-step device sync, control noise outside the timed region (cli.py:289-292), eager launches"}


def other_configs(mjw, nstep=200):
  """BASELINE.json configs[2] and [3] on this GPU, off the timed headline: the model as authored (Newton, implicitfast), the
  reference's measurement loop (per-step device sync, noise / replayed control untimed) over `nstep` steps after an untimed lead-in."""
  import torch

  out = {}
  for name, rel, nworld, nconmax, njmax, replay in (
      ("unitree_g1_flat", ("unitree_g1", "scene_flat.xml"), 4096, 48, 192, "shuffle_dance.npz"),
      ("franka_emika_panda", ("franka_emika_panda", "scene.xml"), 8192, 1, 5, None)):
    mjm = mjw.mjcf.load_xml(os.path.join(ROOT, "benchmarks", *rel))
    m = mjw.put_model(mjm)
    d = mjw.make_data(mjm, nworld=nworld, nconmax=nconmax, njmax=njmax)
    if mjm.nkey:
      mjw.reset_data_keyframe(m, d, 0)
    center = None
    if replay:  # the recorded controls are the noise centre (reference cli.py:119-145 with --replay)
      ctrl = mjw.load_trajectory(os.path.join(ROOT, "benchmarks", rel[0], replay), mjm, mjw.MjData(mjm))
      center = [mjw.DeviceArray.from_numpy(np.asarray(c, dtype=np.float32)) for c in ctrl[: 100 + nstep]]  # [nu]: the noise centre of every world
      z = np.load(os.path.join(ROOT, "benchmarks", rel[0], replay))
      if "qpos" in z.files and z["qpos"].shape[1] == mjm.nq:  # start where the recording starts: the controls were made for that state
        d.qpos.assign(np.tile(z["qpos"][0].astype(np.float32), (nworld, 1)))
        d.qvel.assign(np.tile(z["qvel"][0].astype(np.float32), (nworld, 1)))
    total = nefc = niter = 0.0
    for i in range(100 + nstep):
      mjw.ctrl_noise(m, d, i, center=center[i] if center else None)
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      mjw.step(m, d)
      torch.cuda.synchronize()
      if i >= 100:
        total += time.perf_counter() - t0
        if i % 50 == 49:
          nefc += float(np.minimum(d.nefc.numpy(), d.njmax).mean())
          niter += float(d.solver_niter.numpy().mean())
    ok = bool(np.isfinite(d.qpos.numpy()).all())
    ms_b2b, _ = mjw.timed_steps(m, d, nstep, step0=100 + nstep)
    out[name] = {"workload": f"{rel[1]}, nworld={nworld}, nconmax={nconmax}, njmax={njmax}, solver / integrator / iteration caps as authored"
                             + (f", initial state and control centre from {replay} + noise" if replay else ", control noise"),
                 "value": nworld * nstep / total, "unit": "env-steps/s", "nstep": nstep, "ms_per_step": 1e3 * total / nstep,
                 "back_to_back_value": nworld * nstep / (ms_b2b * 1e-3), "nefc_mean": nefc / (nstep // 50), "solver_niter_mean": niter / (nstep // 50),
                 "finite": ok, "overflow_bits": int(np.bitwise_or.reduce(d.overflow.numpy())),
                 "timing": "reference placement (per-step sync, control untimed); back_to_back_value = the same steps enqueued without syncs"}
    del d
  return out


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=200)
  ap.add_argument("--warmup", type=int, default=20)
  ap.add_argument("--nworld", type=int, default=NWORLD, help="worlds per GPU (weak) / in total (strong)")
  ap.add_argument("--solver", default="cg", choices=sorted(SOLVERS))
  ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--no-roofline", action="store_true")
  ap.add_argument("--no-steady", action="store_true", help="skip the 1000-step reference-placement figure")
  ap.add_argument("--pmc-profile", default="auto", help="rocprofv3 PMC summary (tools/make_pmc_summary.py) of this solver's kernel; "
                  "auto = the committed profiles/round3_pmc_<solver>.json (labelled as such in traffic_source), none = null")
  ap.add_argument("--no-configs", action="store_true", help="skip the BASELINE configs[2] / [3] figures (G1 4096 worlds, Panda 8192 worlds)")
  args = ap.parse_args()
  if args.gpus < 1:
    raise SystemExit("--gpus must be >= 1")
  if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
    raise SystemExit(respawn_under_torchrun(args.gpus))

  xml = os.path.join(ROOT, "benchmarks", "humanoid", "humanoid.xml")
  cpu = None
  if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_cpu_baseline:
    cpu = cpu_baseline(xml, SOLVERS[args.solver])  # before any HIP context exists (fork-safe)

  import torch

  import mujoco_warp_amd as mjw
  from mujoco_warp_amd import shard

  if not torch.cuda.is_available():
    raise SystemExit("bench.py needs an MI355X: no HIP device visible")
  rank, local_rank, world_size = check_group(args.gpus)
  torch.cuda.set_device(local_rank % torch.cuda.device_count())

  mjm = mjw.mjcf.load_xml(xml)
  mjw.override_model(mjm, {"opt.solver": args.solver})
  m = mjw.put_model(mjm)

  def shard_of(mode):
    if mode == "weak":
      return rank * args.nworld, args.nworld
    return shard.shard_worlds(args.nworld, rank, world_size)

  def run(mode):
    off, cnt = shard_of(mode)
    d, snap, r = measure(mjw, shard, m, mjm, cnt, off, args.steps, args.warmup)
    t_max, env_steps, nan_tot, ovf_tot = shard.reduce_metrics(r["elapsed"], float(cnt * args.steps), r["nan_worlds"], r["ovf_worlds"])
    return d, snap, r, cnt, t_max, env_steps, nan_tot, ovf_tot

  d, snapshot, r, nworld, t_max, env_steps, nan_tot, ovf_tot = run(args.scaling)
  nefc, niter = r["nefc"], r["niter"]
  total_worlds = args.nworld * world_size if args.scaling == "weak" else args.nworld

  out = None
  if rank == 0:
    out = {
      "metric": "env-steps/sec (nworld x steps/s) for humanoid.xml at 8192 worlds, 1/2/4/8 GPUs",
      "value": env_steps / t_max, "unit": "env-steps/s", "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
      "ms_per_step": 1e3 * t_max / args.steps, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
      "dtype": "f32", "data": "synthetic",
      "config": {"workload": f"humanoid.xml, nworld={args.nworld} {'per GPU' if args.scaling == 'weak' else 'in total'}, {args.solver.upper()} solver, Euler, pyramidal, "
                             "nconmax=24, njmax=64, key 0 + Halton/OU control noise (std 0.01, rate 0.1)",
                 "nworld_per_gpu": nworld, "nworld_total": total_worlds, "parallelism": f"worlds sharded over {world_size} GPU(s), no data-path collective"},
      "timing": "K steps enqueued back to back on one stream (no per-step sync), control noise inside the timed region; the "
                "reference's placement (per-step sync, noise untimed) is reported as steady_1000",
      "event_ms_per_step_rank0": r["ev_ms"] / args.steps,
      "converged_worlds": int(total_worlds - nan_tot), "overflow_worlds": int(ovf_tot),
      "nefc_mean": float(nefc.mean()), "nefc_p95": float(np.percentile(nefc, 95)),
      "solver_niter_mean": float(niter.mean()), "solver_niter_p95": float(np.percentile(niter, 95)),
      "reference_published_env_steps_per_s": {"value": 2729192, "note": "Newton solver, unstated NVIDIA GPU (benchmarks/README.md:48); not the same metric"},
    }

  # ---- roofline of the dominant kernel: instrumented replay of the same K steps (rank 0) ----
  if rank == 0 and not args.no_roofline:
    for k, v in snapshot.items():
      getattr(d, k).assign(v)
    # HIP event pairs (recorded on the launch stream, inside mjh_timed_steps) around each launch of the fused step
    ms2, pk = mjw.timed_steps(m, d, args.steps, step0=args.warmup, per_kernel=True)
    names = mjw.KERNEL_NAMES
    fused_us = {n: 1e3 * t / args.steps for n, t in zip(names, pk) if t > 0}
    ne = float(nefc.mean())
    # algorithmic bytes per world-step: SURVEY.md section 8(d), float32/int32 words (restated in DESIGN.md section 4)
    words_solve = (1135 if args.solver == "cg" else 406) + 33 * ne  # solver pass: M (CG: + qLD), J, D, aref, type/id, qacc_*, outputs
    words_crb = 1501                                                # CRBA(+factor) pass: cinert, cdof in; crb, M, qLD, qLDiagInv out
    t_dom = fused_us["solve"] * 1e-6
    achieved = 4 * words_solve * nworld / t_dom / 1e9
    traffic, traffic_src = _traffic_from_profile(args.pmc_profile, args.solver)
    out["roofline"] = {"kernel": {"pgs": "k_solve_pgs", "cg": "k_solve_plus<CG> (solver + L'DL-factor / publication riders)",
                                  "newton": "k_solve_plus<NEWTON>"}[args.solver], "bound": "hbm",
                       "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                       "traffic": traffic, "traffic_source": traffic_src, "bytes_per_launch": 4 * words_solve * nworld,
                       "us_per_launch": fused_us["solve"],
                       "note": "the solver is VALU-issue / latency bound, not an HBM stream (profiles/): the HBM fraction is reported because the contract asks for it"}
    t_pass = (fused_us["fwd_pos"] + fused_us["solve"]) * 1e-6
    pass_bytes = 4 * (words_crb + words_solve) * nworld
    out["pass_crba_solver"] = {"bytes": pass_bytes, "us": t_pass * 1e6, "achieved_GBs": pass_bytes / t_pass / 1e9,
                               "frac_of_8TBs": pass_bytes / t_pass / 1e9 / HBM_PEAK_GBS,
                               "note": "time = k_fwd_pos_plus (FK+CoM+CRBA fused) + the solver launch; bytes = SURVEY 8(d)"}
    out["fused_launch_us"] = fused_us
    # per-stage trace: one plain kernel per stage (the reference's event-tracer granularity)
    for k, v in snapshot.items():
      getattr(d, k).assign(v)
    ms3, pk3 = mjw.timed_steps(m, d, args.steps, step0=args.warmup, per_kernel=True, plain_kernels=True)
    out["per_kernel_us"] = {n: 1e3 * t / args.steps for n, t in zip(names, pk3) if n != "mid"}
  del d

  # ---- the other scaling mode (N > 1 only): same K / W, its own Data ----
  if world_size > 1:
    other = "strong" if args.scaling == "weak" else "weak"
    d2, _, r2, cnt2, t2, steps2, nan2, _ = run(other)
    del d2
    if rank == 0:
      out["other_scaling"] = {"scaling": other, "value": steps2 / t2, "ms_per_step": 1e3 * t2 / args.steps, "nworld_per_gpu": cnt2,
                              "nworld_total": args.nworld * world_size if other == "weak" else args.nworld}

  # ---- the reference's own 1000-step measurement (rank 0's shard; N = 1 semantics) ----
  if rank == 0 and not args.no_steady:
    off, cnt = shard_of(args.scaling)
    out["steady_1000"] = steady_1000(mjw, m, mjm, cnt, off)
  shard.barrier()

  if rank == 0 and world_size == 1 and not args.no_configs:
    out["configs"] = other_configs(mjw)
  if rank == 0 and cpu is not None:
    out["cpu_baseline"] = cpu
  if rank == 0:
    print(json.dumps(out))
  if world_size > 1:
    import torch.distributed as dist

    dist.destroy_process_group()


def _traffic_from_profile(path, solver):
  """HBM bytes per solver launch from a rocprofv3 PMC summary of the SAME solver's kernel (FETCH_SIZE x2 on gfx950 + WRITE_SIZE,
  separate passes: MI355X_MICROARCH.md).  Without such a profile the field is null: a number from another run is not a measurement."""
  committed = False
  if path == "auto":
    path = os.path.join(ROOT, "profiles", f"round3_pmc_{solver}.json")
    committed = True
    if not os.path.exists(path):
      path = None
  if not path or path == "none":
    return None, "not collected in this run (rocprofv3 --pmc needs its own passes; see profiles/)"
  try:
    with open(path) as f:
      j = json.load(f)
    if j.get("solver") != solver:
      return None, f"{path} holds solver={j.get('solver')}, this run is {solver}"
    src = f"{os.path.relpath(path, ROOT)} (rocprofv3 PMC passes of the same workload and solver, NOT this process"
    return j.get("k_solve_hbm_bytes_per_launch"), src + ("; committed with the repo: re-collect with tools/profile_round.sh after kernel changes)" if committed else ")")
  except Exception as e:
    return None, f"{path}: {e}"


if __name__ == "__main__":
  main()
