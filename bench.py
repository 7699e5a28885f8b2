"""Headline benchmark: env-steps/sec for humanoid.xml at 8192 worlds per GPU (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the mj_step hot path (control-noise kernel + step) over a batch of `nworld` synthetic
worlds that start from keyframe 0 and are decorrelated by the harness' Halton/OU control noise
(reference cli.py:103-145, 240-297).  State is resident in HBM before the timed region starts; the timed
region is bracketed by barrier + torch.cuda.synchronize() on both sides; `value` = all worlds of all ranks x K
divided by the MAX elapsed time over ranks.  Worlds shard with no data-path collective ("scaling": "weak":
every GPU steps its own 8192 worlds); the only collective is the metrics all-reduce after the timed region.

Extra objects on the JSON line:
  roofline     -- dominant kernel (k_solve): algorithmic HBM bytes per launch / mean launch time from HIP events
                  recorded on the launch stream (second, instrumented pass over the same K steps)
  cpu_baseline -- the float64 oracle ("port": restatement, NOT MuJoCo C) on the host cores, bounded sample
"""

import argparse
import json
import multiprocessing
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


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
              f"(restatement of the reference algorithm, NOT MuJoCo C); single-thread figure from a 3 s run; wall {wall:.1f}s",
  }


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=200)
  ap.add_argument("--warmup", type=int, default=20)
  ap.add_argument("--nworld", type=int, default=8192, help="worlds per GPU")
  ap.add_argument("--solver", default="cg", choices=["cg", "newton", "pgs"])
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--no-roofline", action="store_true")
  args = ap.parse_args()

  xml = os.path.join(ROOT, "benchmarks", "humanoid", "humanoid.xml")
  cpu = None
  if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_cpu_baseline:
    cpu = cpu_baseline(xml, {"pgs": 0, "cg": 1, "newton": 2}[args.solver])  # before any HIP context exists (fork-safe)

  import torch

  import mujoco_warp_amd as mjw
  from mujoco_warp_amd import shard

  rank, local_rank, world_size = shard.init_process_group()
  if world_size != args.gpus and world_size > 1:
    raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_size}")
  if not torch.cuda.is_available():
    raise SystemExit("bench.py needs an MI355X: no HIP device visible")
  torch.cuda.set_device(local_rank % torch.cuda.device_count())

  mjm = mjw.mjcf.load_xml(xml)
  mjw.override_model(mjm, {"opt.solver": args.solver})
  m = mjw.put_model(mjm)
  nworld = args.nworld
  d = mjw.make_data(mjm, nworld=nworld, nconmax=24, njmax=64)
  d.world_offset = rank * nworld  # global world ids: trajectories independent of the number of GPUs
  mjw.reset_data_keyframe(m, d, 0)

  # warmup (untimed)
  if args.warmup:
    mjw.timed_steps(m, d, args.warmup, step0=0)
  snapshot = {k: getattr(d, k).numpy().copy() for k in ("qpos", "qvel", "ctrl", "qacc_warmstart", "time")}

  shard.barrier()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  ev_ms, _ = mjw.timed_steps(m, d, args.steps, step0=args.warmup)
  torch.cuda.synchronize()
  shard.barrier()
  elapsed = time.perf_counter() - t0

  qpos = d.qpos.numpy()
  nan_worlds = float(np.isnan(qpos).any(axis=1).sum())
  ovf_worlds = float((d.overflow.numpy() != 0).sum())
  nefc = np.minimum(d.nefc.numpy(), d.njmax)
  niter = d.solver_niter.numpy()
  t_max, env_steps, nan_tot, ovf_tot = shard.reduce_metrics(elapsed, float(nworld * args.steps), nan_worlds, ovf_worlds)

  out = None
  if rank == 0:
    out = {
      "metric": "env-steps/sec (nworld x steps/s) for humanoid.xml at 8192 worlds, 1/2/4/8 GPUs",
      "value": env_steps / t_max, "unit": "env-steps/s", "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
      "ms_per_step": 1e3 * t_max / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
      "dtype": "f32", "data": "synthetic",
      "config": {"workload": f"humanoid.xml, nworld={nworld} per GPU, {args.solver.upper()} solver, Euler, pyramidal, nconmax=24, njmax=64, "
                             "key 0 + Halton/OU control noise (std 0.01, rate 0.1)",
                 "nworld_per_gpu": nworld, "nworld_total": nworld * world_size, "parallelism": f"worlds sharded over {world_size} GPU(s)"},
      "event_ms_per_step_rank0": ev_ms / args.steps,
      "converged_worlds": int(nworld * world_size - nan_tot), "overflow_worlds": int(ovf_tot),
      "nefc_mean": float(nefc.mean()), "nefc_p95": float(np.percentile(nefc, 95)),
      "solver_niter_mean": float(niter.mean()), "solver_niter_p95": float(np.percentile(niter, 95)),
      "reference_published_env_steps_per_s": {"value": 2729192, "note": "Newton solver, unstated NVIDIA GPU (benchmarks/README.md:48); not the same metric"},
    }

  # ---- roofline of the dominant kernel: instrumented replay of the same K steps (rank 0, N=1 semantics) ----
  if rank == 0 and not args.no_roofline:
    for k, v in snapshot.items():
      getattr(d, k).assign(v)
    # HIP event pairs (recorded on the launch stream, inside mjh_timed_steps) around each launch of the fused step
    ms2, pk = mjw.timed_steps(m, d, args.steps, step0=args.warmup, per_kernel=True)
    names = mjw.KERNEL_NAMES
    fused_us = {n: 1e3 * t / args.steps for n, t in zip(names, pk) if t > 0}
    ne = float(nefc.mean())
    cg = args.solver == "cg"
    # algorithmic bytes per world-step: SURVEY.md section 8(d), float32/int32 words (restated in DESIGN.md section 4)
    words_solve = (1135 if cg else 406) + 33 * ne  # solver pass: M/qLD, J, D, aref, type/id, qacc_*, outputs
    words_crb = 1501                               # CRBA(+factor) pass: cinert, cdof in; crb, M, qLD, qLDiagInv out
    t_dom = fused_us["solve"] * 1e-6
    achieved = 4 * words_solve * nworld / t_dom / 1e9
    out["roofline"] = {"kernel": "k_solve_pgs" if args.solver == "pgs" else "k_solve_plus (solver workgroups + L'DL factor workgroups of the fused step)", "bound": "hbm",
                       "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                       "traffic": _traffic_from_profile(), "bytes_per_launch": 4 * words_solve * nworld,
                       "us_per_launch": fused_us["solve"],
                       "note": "latency/VALU-bound kernel: 59% VALU issue utilisation (profiles/), not an HBM stream"}
    t_pass = (fused_us["fwd_pos"] + fused_us["solve"]) * 1e-6
    pass_bytes = 4 * (words_crb + words_solve) * nworld
    out["pass_crba_solver"] = {"bytes": pass_bytes, "us": t_pass * 1e6, "achieved_GBs": pass_bytes / t_pass / 1e9,
                               "frac_of_8TBs": pass_bytes / t_pass / 1e9 / HBM_PEAK_GBS,
                               "note": "time = k_fwd_pos_plus (FK+CoM+CRBA fused) + k_solve_plus (solver + factor); bytes = SURVEY 8(d)"}
    out["fused_launch_us"] = fused_us
    # per-stage trace: one plain kernel per stage (the reference's event-tracer granularity)
    for k, v in snapshot.items():
      getattr(d, k).assign(v)
    ms3, pk3 = mjw.timed_steps(m, d, args.steps, step0=args.warmup, per_kernel=True, plain_kernels=True)
    out["per_kernel_us"] = {n: 1e3 * t / args.steps for n, t in zip(names, pk3) if n != "mid"}

  if rank == 0 and cpu is not None:
    out["cpu_baseline"] = cpu

  if rank == 0:
    print(json.dumps(out))
  if world_size > 1:
    import torch.distributed as dist

    dist.destroy_process_group()


def _traffic_from_profile():
  """HBM bytes per k_solve launch from the committed PMC summary (profiles/), if present; else null."""
  p = os.path.join(ROOT, "profiles", "round1_pmc_summary.json")
  try:
    with open(p) as f:
      return json.load(f).get("k_solve_hbm_bytes_per_launch")
  except Exception:
    return None


if __name__ == "__main__":
  main()
