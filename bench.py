"""Headline benchmark: env-steps/sec for humanoid.xml at 8192 worlds (BASELINE.json configs[1]), 1/2/4/8 GPUs.

    python bench.py --gpus N --steps K --warmup W [--solver cg|newton|pgs] [--scaling weak|strong]

N > 1 without a torch.distributed environment: bench.py re-executes itself through
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (one rank per GPU, RCCL);
launched by torchrun it reads RANK / LOCAL_RANK / WORLD_SIZE.  Every rank checks that the group really has N ranks and
that each rank owns a distinct device.

A "step" is one pass of the mj_step hot path over a batch of synthetic worlds that start from keyframe 0 and are decorrelated by
the harness' Halton/OU control noise (reference cli.py:103-145, 240-297).  State is resident in HBM before the timed region
starts.  TIMER PLACEMENT = THE REFERENCE'S (cli.py:289-292, testspeed.py:362): per step the control-noise kernel runs
and is synchronised UNTIMED, then `step` + device synchronise is timed; `value` = all worlds of all ranks x K / MAX over ranks of the
summed step times.  The window of exactly K steps (after W untimed warm-up steps) is bracketed by barrier + synchronize on both
sides and measured THREE times from the same state: `value` is the median window, `windows` lists all three (box-to-box noise on
this fleet is 20-40 %, run-to-run noise ~1 %: `box` carries a fingerprint -- the state-independent k_fwd_pos launch time and
rocm-smi clocks / power when readable).  `back_to_back` = the same K steps enqueued without per-step syncs, noise inside.
Worlds shard with no data-path collective; the only collective is the metrics all-reduce after the timed region.
  --scaling weak   (default) every GPU steps its own 8192 worlds  -> "scaling": "weak"
  --scaling strong 8192 worlds in total, shard_worlds(8192, rank, N) per GPU
With N > 1 the line also carries the other mode as `other_scaling`.

Extra objects on the JSON line:
  roofline     -- dominant kernel (the solver launch): SURVEY 8(d) algorithmic HBM bytes per launch / mean launch time from
                  HIP events recorded on the launch stream (instrumented replay of the same K steps); `traffic` is null
                  unless --pmc-profile names a rocprofv3 PMC summary of THIS solver's kernel; `issue` = what actually bounds the three
                  launches, from the same PMC summary: VALU issue fraction (SQ_INSTS_VALU x 2 cycles / SIMD busy cycles), LDS pipe
                  busy fraction, resident waves per SIMD, wait fraction, HBM fraction of each kernel
  steady_1000  -- the reference's own measurement (testspeed.py:362, cli.py:289-292): 1000 steps from key 0, device sync per
                  step, control noise outside the timed region -- the state the published metric averages over, whatever
                  --steps / --warmup the caller passed
  configs      -- the other BASELINE configs on this GPU, reference placement: unitree_g1_flat, franka_emika_panda, aloha_pot (the
                  reference's in-tree ALOHA model, its registry sizes) and the configs[4]-class clutter_synth (Newton + sleep, PGS)
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
  """Oracle ("port") on the host: single-thread rate, then one independent rollout stream per usable core.  Compiled for this host
  (`-O3 -march=native`) before the workers fork; the flags are part of `sample`."""
  from oracle import ref

  try:
    flags = ref.use_native_build()
  except Exception as e:  # (no compiler on the box: the -O2 build that travelled with the repo)
    flags = f"prebuilt oracle/libmjref.so (-O2; native build failed: {type(e).__name__})"
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
              f"(restatement of the reference algorithm, NOT MuJoCo C: not installable here or on the GPU box), built on this host with `{flags}`; single-thread figure from a 3 s run; wall {wall:.1f}s",
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


STATE_FIELDS = ("qpos", "qvel", "ctrl", "qacc_warmstart", "time", "solver_niter")
NWINDOW = 3


def reference_window(mjw, m, d, steps, step0):
  """K steps with the reference's timer placement (cli.py:289-292): noise kernel + sync untimed, step + sync timed.  Returns seconds."""
  import torch

  total = 0.0
  for i in range(steps):
    mjw.ctrl_noise(m, d, step0 + i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mjw.step(m, d)
    torch.cuda.synchronize()
    total += time.perf_counter() - t0
  return total


def measure(mjw, shard, m, mjm, nworld, world_offset, steps, warmup):
  """W untimed warm-up steps, then exactly K timed steps (barrier + synchronize on both sides), NWINDOW times from the same state.
  Returns the Data (for the replays), a snapshot of the state at the start of the timed window and the per-rank measurement."""
  import torch

  d = mjw.make_data(mjm, nworld=nworld, nconmax=24, njmax=64)
  d.world_offset = world_offset  # global world ids: trajectories independent of the number of GPUs
  mjw.reset_data_keyframe(m, d, 0)
  mjw.step(m, d)  # function attributes / first touch outside the timed region (the reference captures its graph before timing)
  mjw.reset_data_keyframe(m, d, 0)
  if warmup:
    mjw.timed_steps(m, d, warmup, step0=0)
  torch.cuda.synchronize()
  snapshot = {k: getattr(d, k).numpy().copy() for k in STATE_FIELDS}
  windows = []
  for rep in range(NWINDOW):
    for k, v in snapshot.items():
      getattr(d, k).assign(v)
    shard.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    windows.append(reference_window(mjw, m, d, steps, warmup))
    torch.cuda.synchronize()
    shard.barrier()
    wall = time.perf_counter() - t0
  qpos = d.qpos.numpy()
  nefc, niter = np.minimum(d.nefc.numpy(), d.njmax), d.solver_niter.numpy()
  nan_worlds, ovf_worlds = float(np.isnan(qpos).any(axis=1).sum()), float((d.overflow.numpy() != 0).sum())
  # the same K steps enqueued back to back (no per-step sync, noise kernel inside): the engine's throughput when nothing waits for it
  for k, v in snapshot.items():
    getattr(d, k).assign(v)
  shard.barrier()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  ev_ms, _ = mjw.timed_steps(m, d, steps, step0=warmup)
  torch.cuda.synchronize()
  shard.barrier()
  b2b = time.perf_counter() - t0
  res = {"windows": windows, "wall_last_window": wall, "b2b": b2b, "ev_ms": ev_ms, "nan_worlds": nan_worlds, "ovf_worlds": ovf_worlds, "nefc": nefc, "niter": niter}
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
  out = {"value": nworld * nstep / total, "unit": "env-steps/s", "nstep": nstep, "ms_per_step": 1e3 * total / nstep,
         "nefc_mean": nefc_sum / (nstep // 100), "solver_niter_mean": niter_sum / (nstep // 100), "converged_worlds": ok,
         "timing": "reference placement: per-step device sync, control noise outside the timed region (cli.py:289-292), eager launches"}
  # the fused step's launches at THIS state (HIP event pairs on the launch stream, 50 more steps): the steady-state roofline's clock
  _, pk = mjw.timed_steps(m, d, 50, step0=nstep, per_kernel=True)
  out["fused_launch_us"] = {n: 1e3 * t / 50 for n, t in zip(mjw.KERNEL_NAMES, pk) if t > 0}
  out["nefc_mean_at_end"] = float(np.minimum(d.nefc.numpy(), d.njmax).mean())
  return out


def other_configs(mjw, nstep=200, only=None):
  """The other BASELINE.json configs on this GPU, off the timed headline: each model as authored (solver, integrator, cone, iteration caps),
  the reference's measurement loop (per-step device sync, noise / replayed control untimed) over `nstep` steps after an untimed lead-in.
    unitree_g1_flat, franka_emika_panda  -- configs[2] / [3]
    aloha_pot                           -- the reference's in-tree ALOHA model (mujoco_warp/test_data/aloha_pot, all 134 assets) at the sizes of
                                           its benchmark registry (benchmarks/aloha/__init__.py:15-25), replaying lift_pot.npz
    clutter_synth, clutter_synth_pgs    -- configs[4] class (aloha_clutter's 216 meshes are not in the reference tree): nv 136, elliptic, impratio 10,
                                           2048 worlds; Newton + sleeping + init_asleep as the reference's registry runs it, and the PGS solver configs[4] names"""
  import torch

  B = os.path.join(ROOT, "benchmarks")
  entries = (
    dict(name="unitree_g1_flat", xml=os.path.join(B, "unitree_g1", "scene_flat.xml"), nworld=4096, nconmax=48, njmax=192, replay=os.path.join(B, "unitree_g1", "shuffle_dance.npz")),
    dict(name="humanoid_shard", xml=os.path.join(B, "humanoid", "humanoid.xml"), nworld=1024, nconmax=24, njmax=64, override=["opt.solver=cg"], lead=300, launch_us=True,
         note="the headline under STRONG scaling: 8192 worlds over 8 GPUs = 1024 each (`solver_kernel` names the CG mapping the dispatch picks at this size); "
              "8 x this figure is the single-GPU lower bound of an 8-GPU strong-scaling run"),
    dict(name="franka_emika_panda", xml=os.path.join(B, "franka_emika_panda", "scene.xml"), nworld=8192, nconmax=1, njmax=5),
    dict(name="franka_emika_panda_shard", xml=os.path.join(B, "franka_emika_panda", "scene.xml"), nworld=1024, nconmax=1, njmax=5,
         note="configs[3] per GPU: 8192 worlds over 8 GPUs = 1024 each -- a quarter of the device's wavefront slots, the step is launch latency"),
    dict(name="aloha_pot", xml=os.path.join(B, "aloha_pot", "scene.xml"), nworld=8192, nconmax=24, njmax=128, replay=os.path.join(B, "aloha_pot", "lift_pot.npz")),
    dict(name="clutter_synth", xml=os.path.join(B, "clutter_synth", "scene_clutter_synth.xml"), nworld=2048, nconmax=256, njmax=384, nvmax=56,
         override=["opt.enableflags=SLEEP"], init_asleep=True, hold_key_ctrl=True),
    dict(name="clutter_synth_shard", xml=os.path.join(B, "clutter_synth", "scene_clutter_synth.xml"), nworld=256, nconmax=256, njmax=384, nvmax=56,
         override=["opt.enableflags=SLEEP"], init_asleep=True, hold_key_ctrl=True, nstep=100, lead=50,
         note="configs[4] per GPU: 2048 worlds over 8 GPUs = 256 each"),
    dict(name="clutter_synth_pgs", xml=os.path.join(B, "clutter_synth", "scene_clutter_synth.xml"), nworld=2048, nconmax=256, njmax=384,
         override=["opt.solver=pgs", "opt.enableflags=0"], nstep=100, lead=50, hold_key_ctrl=True),
  )
  out = {}
  for e in entries:
    if only and not any(e["name"] == o for o in only):
      continue
    try:
      out[e["name"]] = _config_run(mjw, torch, e, e.get("nstep", nstep), e.get("lead", 100))
    except Exception as ex:  # a side config must not take the headline line down with it
      out[e["name"]] = {"error": f"{type(ex).__name__}: {ex}"}
  return out


def _config_run(mjw, torch, e, nstep, lead):
  mjm = mjw.mjcf.load_xml(e["xml"])
  if e.get("override"):
    mjw.override_model(mjm, e["override"])
  m = mjw.put_model(mjm)
  mjd = mjw.MjData(mjm)
  if mjm.nkey:
    mjw.mj_resetDataKeyframe(mjm, mjd, 0)
  center = None
  if e.get("replay"):  # the recorded controls are the noise centre (reference cli.py:119-145 with --replay); the recording's start state is applied
    ctrl = mjw.load_trajectory(e["replay"], mjm, mjd)
    center = [mjw.DeviceArray.from_numpy(np.asarray(c, dtype=np.float32)) for c in ctrl[: lead + nstep]]
  if e.get("init_asleep"):  # reference cli.py:167-168
    mjd.tree_asleep[:] = np.arange(mjm.ntree, dtype=np.int32)
  d = mjw.put_data(mjm, mjd, nworld=e["nworld"], nconmax=e["nconmax"], njmax=e["njmax"], nvmax=e.get("nvmax"))
  hold = mjw.DeviceArray.from_numpy(np.asarray(mjd.ctrl, dtype=np.float32)) if (mjm.nu and center is None and e.get("hold_key_ctrl")) else None  # noise around the keyframe's controls
  total = nefc = niter = ncon = 0.0
  nstat = 0
  # Steps of MANY launches (sleeping: ~20 plain kernels; more than 64 dofs: per-island solver launches on forked streams) are replayed from a
  # hipGraph of `step`, the reference's own mechanism (cli.py:262-290: capture once, launch + sync per step); the few-launch steps are launched
  # eagerly like the headline (a graph replay is ~6 us slower there: LAB_NOTES R5.6).  Round 6, clutter_synth: eager 1.28 M, graph 1.53 M.
  graph = mjw.StepGraph(m, d) if (m.sleep_enabled or mjm.nv > 64) else None
  for i in range(lead + nstep):
    mjw.ctrl_noise(m, d, i, center=center[min(i, len(center) - 1)] if center else hold)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if graph is not None:
      graph.launch()
    else:
      mjw.step(m, d)
    torch.cuda.synchronize()
    if i >= lead:
      total += time.perf_counter() - t0
      if i % 50 == 49:
        nefc += float(np.minimum(d.nefc.numpy(), d.njmax).mean())
        niter += float(d.solver_niter.numpy().mean())
        ncon += float(d.ws_ncon.numpy().mean())
        nstat += 1
  ok = bool(np.isfinite(d.qpos.numpy()).all())
  ovf = int(np.bitwise_or.reduce(d.overflow.numpy()))
  res = {"workload": f"{os.path.basename(e['xml'])}, nworld={e['nworld']}, nconmax={e['nconmax']}, njmax={e['njmax']}"
                     + (f", nvmax={e['nvmax']}" if e.get("nvmax") else "") + (f", {' '.join(e['override'])}" if e.get("override") else "")
                     + (", init_asleep" if e.get("init_asleep") else "") + ", solver / integrator / cone / iteration caps as authored"
                     + (f", initial state and control centre from {os.path.basename(e['replay'])} + noise" if e.get("replay") else ", control noise"),
         "nv": int(mjm.nv), "ngeom": int(mjm.ngeom), "solver": ["PGS", "CG", "NEWTON"][int(mjm.opt.solver)], "solver_kernel": mjw.solver_kernel(m, d), "cone": ["pyramidal", "elliptic"][int(mjm.opt.cone)],
         "value": e["nworld"] * nstep / total, "unit": "env-steps/s", "nstep": nstep, "ms_per_step": 1e3 * total / nstep,
         "ncon_mean": ncon / max(nstat, 1), "nefc_mean": nefc / max(nstat, 1), "solver_niter_mean": niter / max(nstat, 1),
         "finite": ok, "overflow_bits": ovf, "iteration_cap_worlds": int(((d.overflow.numpy() >> 9) & 1).sum()),
         "timing": "reference placement (per-step sync, control untimed)" + (", hipGraph replay of step as the reference (cli.py:262-290)" if graph is not None else ", eager launches")}
  if e.get("note"):
    res["note"] = e["note"]
    res["x8"] = {"value": 8 * res["value"], "note": "8 shards of this size, one per GPU, no data-path collective: the single-GPU figure x 8 (a projection, not a measured 8-GPU run)"}
  if int(mjm.opt.solver) != 0 and not (int(mjm.opt.enableflags) & int(mjw.EnableBit.SLEEP)) and not e.get("replay") and not e.get("hold_key_ctrl"):  # (timed_steps: fused launch sequence, noise around the ctrl-range midpoint -- the same workload only without a replayed / held control centre)
    ms_b2b, _ = mjw.timed_steps(m, d, nstep, step0=lead + nstep)
    res["back_to_back_value"] = e["nworld"] * nstep / (ms_b2b * 1e-3)
    if e.get("launch_us"):
      _, pk = mjw.timed_steps(m, d, 50, step0=lead + 2 * nstep, per_kernel=True)
      res["fused_launch_us"] = {n: 1e3 * t / 50 for n, t in zip(mjw.KERNEL_NAMES, pk) if t > 0}
  del d
  return res


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
                  "auto = the newest committed profiles/round<N>_pmc_<solver>_<early|steady>.json whose window matches --warmup (labelled as such in traffic_source), none = null")
  ap.add_argument("--configs-only", default="", help="comma-separated names: run only these entries of `configs` (developer: A/B of one config)")
  ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE configs (G1 4096 worlds, Panda 8192, aloha_pot 8192, clutter_synth 2048 Newton + PGS)")
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
    # every window: MAX over ranks of the summed step times; counters summed over ranks
    red = [shard.reduce_metrics(w, float(cnt * args.steps), r["nan_worlds"], r["ovf_worlds"]) for w in r["windows"]]
    r["windows_max"] = [x[0] for x in red]
    r["b2b_max"] = shard.reduce_metrics(r["b2b"], 0.0, 0.0, 0.0)[0]
    t_med = float(np.median(r["windows_max"]))
    return d, snap, r, cnt, t_med, red[0][1], red[0][2], red[0][3]

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
      "timing": "the reference's placement (cli.py:289-292): per step the control-noise kernel + sync untimed, then step + device sync timed; "
                f"value = median of {NWINDOW} windows of exactly K steps from the same state (max over ranks per window); back_to_back = the same steps without per-step syncs",
      "windows": [{"value": env_steps / t, "ms_per_step": 1e3 * t / args.steps} for t in r["windows_max"]],
      "window_spread": (max(r["windows_max"]) - min(r["windows_max"])) / t_max,
      "back_to_back": {"value": env_steps / r["b2b_max"], "ms_per_step": 1e3 * r["b2b_max"] / args.steps, "event_ms_per_step_rank0": r["ev_ms"] / args.steps,
                       "timing": "K steps enqueued on one stream, no per-step sync, control noise inside the timed region"},
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
    traffic, traffic_src = _traffic_from_profile(args.pmc_profile, args.solver, _window_of(args.warmup))
    out["roofline"] = {"kernel": {"pgs": "k_solve_pgs", "cg": "k_solve_cgp_plus (pooled contact-basis CG + L'DL-factor / publication riders)",
                                  "newton": "k_solve_plus<NEWTON>"}[args.solver], "solver_kernel": mjw.solver_kernel(m, d), "bound": "hbm",
                       "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                       "traffic": traffic, "traffic_source": traffic_src, "bytes_per_launch": 4 * words_solve * nworld,
                       "us_per_launch": fused_us["solve"],
                       "note": "not an HBM stream: the kernel is a latency chain at ~3 waves per SIMD -- see `issue` (VALU issue ~0.4, HBM 3-5 %); "
                               "the HBM fraction is reported because the contract asks for it"}
    t_pass = (fused_us["fwd_pos"] + fused_us["solve"]) * 1e-6
    pass_bytes = 4 * (words_crb + words_solve) * nworld
    out["pass_crba_solver"] = {"bytes": pass_bytes, "us": t_pass * 1e6, "achieved_GBs": pass_bytes / t_pass / 1e9,
                               "frac_of_8TBs": pass_bytes / t_pass / 1e9 / HBM_PEAK_GBS,
                               "note": "time = k_fwd_pos_plus (FK+CoM+CRBA fused) + the solver launch; bytes = SURVEY 8(d)"}
    out["fused_launch_us"] = fused_us
    out["roofline"]["issue"] = _issue_from_profile(args.pmc_profile, args.solver, _window_of(args.warmup))
    out["roofline"]["window"] = f"steps {args.warmup}..{args.warmup + args.steps} of the rollout from key 0 ({_window_of(args.warmup)}); the steady state is `roofline_steady`"
    out["box"] = box_fingerprint(fused_us.get("fwd_pos"), fused_us.get("mid"))
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
    if not args.no_roofline and "solve" in out["steady_1000"].get("fused_launch_us", {}):
      # the same roofline at the state the published metric averages over: the launch clock and nefc of the END of the 1000-step rollout, the
      # counter traffic of the committed steady-window PMC summary
      st = out["steady_1000"]
      words = (1135 if args.solver == "cg" else 406) + 33 * st["nefc_mean_at_end"]
      us = st["fused_launch_us"]["solve"]
      ach = 4 * words * cnt / (us * 1e-6) / 1e9
      tr, tr_src = _traffic_from_profile(args.pmc_profile, args.solver, "steady")
      out["roofline_steady"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": tr, "traffic_source": tr_src,
                                "bytes_per_launch": 4 * words * cnt, "us_per_launch": us, "nefc_mean": st["nefc_mean_at_end"],
                                "window": "steps 1000..1050 of the rollout from key 0 (steady)", "issue": _issue_from_profile(args.pmc_profile, args.solver, "steady")}
  shard.barrier()

  if rank == 0 and world_size == 1 and not args.no_configs:
    out["configs"] = other_configs(mjw, only=[x for x in args.configs_only.split(",") if x] or None)
  if rank == 0 and cpu is not None:
    out["cpu_baseline"] = cpu
  if rank == 0:
    print(json.dumps(out))
  if world_size > 1:
    import torch.distributed as dist

    dist.destroy_process_group()


def _committed_profile(solver, window="steady"):
  """The committed rocprofv3 PMC summary of this solver's workload whose WINDOW of the rollout matches the timed one: "early" = the
  driver's --warmup 5 window (free fall and first contacts, nefc ~12), "steady" = steps 300+ (the humanoid on the floor, nefc ~45)."""
  for r in ("round6", "round5", "round4"):
    for name in (f"{r}_pmc_{solver}_{window}.json", f"{r}_pmc_{solver}.json"):
      p = os.path.join(ROOT, "profiles", name)
      if os.path.exists(p):
        return p
  return None


def _window_of(warmup):
  return "early" if warmup < 100 else "steady"


def _issue_from_profile(path, solver, window="steady"):
  """What bounds the three launches of the step, from the rocprofv3 PMC summary of the same workload and solver (tools/make_pmc_summary.py):
  per kernel the VALU issue fraction (a wave64 VALU instruction occupies a SIMD-32 for 2 cycles: SQ_INSTS_VALU x 2 / (1,024 SIMDs x
  busy cycles)), the LDS pipe's busy fraction, the resident waves per SIMD, the share of wave-cycles spent waiting and the HBM fraction."""
  if path == "auto":
    path = _committed_profile(solver, window)
  if not path or path == "none" or not os.path.exists(path):
    return None
  try:
    j = json.load(open(path))
    if j.get("solver") != solver:
      return None
    out = {"source": os.path.relpath(path, ROOT) + " (PMC passes of the same workload and solver, not this process)", "window": j.get("window")}
    for k, v in j.get("kernels", {}).items():
      if v.get("valu_issue_frac") is None:
        continue
      if k in ("k_fwd_pos", "k_mid") or k == j.get("k_solve_kernel"):
        out[k] = {q: v.get(q) for q in ("mean_us", "valu_issue_frac", "lds_busy_frac", "waves_per_simd", "wait_frac", "hbm_frac", "hbm_bytes_per_launch")}
    return out
  except Exception as e:
    return {"error": f"{path}: {e}"}


def box_fingerprint(fwd_pos_us, mid_us=None):
  """Box-to-box variance on this fleet is 20-40 % and does NOT follow the shader clock: round 4 measured the same build at
  27.6 M (sclk 2150 MHz, 354 W; k_mid 69 us, solver 172 us) and at 19.5 M (sclk 2395 MHz, 465 W; k_mid 124 us, solver 236 us) env-steps/s with
  k_fwd_pos at 52 / 54 us on both.  The memory-heavy k_mid launch is the discriminating fingerprint (69-85 us on a fast box, whatever the
  contact state); rocm-smi adds clocks and power where it can be read."""
  out = {"k_fwd_pos_us": fwd_pos_us, "k_mid_us": mid_us,
         "slow_box": bool(mid_us is not None and mid_us > 95.0),
         "note": "k_mid (collision + constraint assembly + velocity stage of 8192 humanoids, 150 MB of traffic) takes 69-85 us on a fast box and 105-125 us on a slow one; "
                 "k_fwd_pos (52-54 us) does not tell them apart; slow_box = k_mid above 95 us"}
  try:
    o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20).stdout
    card = next(iter(json.loads(o).values()))
    for key, val in card.items():
      kl = key.lower()
      if "sclk" in kl or "mclk" in kl or "power" in kl:
        out[key] = val
  except Exception as e:
    out["rocm_smi"] = f"unavailable ({type(e).__name__})"
  return out


def _traffic_from_profile(path, solver, window="steady"):
  """HBM bytes per solver launch from a rocprofv3 PMC summary of the SAME solver's kernel (FETCH_SIZE x2 on gfx950 + WRITE_SIZE,
  separate passes: MI355X_MICROARCH.md).  Without such a profile the field is null: a number from another run is not a measurement."""
  committed = False
  if path == "auto":
    path = _committed_profile(solver, window)
    committed = True
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
