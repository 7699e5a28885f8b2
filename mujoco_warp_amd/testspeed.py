"""mjwarp-testspeed work-alike: benchmark a mujoco_warp_amd function on batched worlds.

Mirrors /root/reference/mujoco_warp/testspeed.py (flags 51-61, metrics 359-378) and the unroll loop of
/root/reference/mujoco_warp/_src/cli.py:240-297 with argparse in place of absl (absent here):

    python -m mujoco_warp_amd.testspeed benchmarks/humanoid/humanoid.xml --nworld=8192 --nconmax=24 --njmax=64 \
        -o opt.solver=cg --format=short --event_trace=true --measure_solver=true --measure_alloc=true

Timer placement is the reference's: per step, the control-noise kernel runs and is synchronised OUTSIDE the timed
region; the timed region is one hipGraph launch of `step` + device synchronise (cli.py:289-292).  Output keys of
--format=short|json are the reference's (jit_duration, run_time, steps_per_second, converged_worlds, *_memory,
ncon_mean/p95, nefc_mean/p95, solver_niter_mean/p95) plus the flattened event trace in ns per world-step under the reference's
@event_scope names (step, step.forward, step.forward.fwd_position, step.forward.fwd_position.fwd_kinematics.kinematics, ...).

Installed as the console script `mjwarp-testspeed` (pyproject.toml; reference pyproject.toml:72-73).
"""

import argparse
import json
import sys
import time

import numpy as np


def _bool(s):
  return str(s).lower() in ("1", "true", "yes", "on")


def _memory(obj, prefix=""):
  from .device import DeviceArray

  out = {}
  for k, v in vars(obj).items():
    if k.startswith("_"):
      continue
    if isinstance(v, DeviceArray):
      out[prefix + k] = v.capacity
    elif hasattr(v, "__dict__") and not isinstance(v, (int, float, str)) and type(v).__module__.endswith("types"):
      out.update(_memory(v, prefix + k + "."))
  return out


def _event_trace(mjw, mjm, m, d, args, centers, center, nstep=20):
  """The reference's EventTracer output (`step`, `step.forward`, `step.forward.fwd_position`, ... : the @event_scope nesting of
  forward.py:388-1380 / smooth.py / collision_driver.py:885 / constraint.py:4898 / solver.py:3671), ns per world-step: `nstep` extra steps run
  stage by stage -- ONE plain kernel per reference stage function, bracketed by HIP events on the launch stream -- so every key maps 1:1 to the
  reference function of that name.  (The timed rollout above runs the fused step; a fused launch has no per-stage time.)  Models with sleeping
  enabled run extra sleep stages between these (forward.py:652-678): their trace keeps the fused launches' granularity."""
  import torch

  sleep = bool(int(mjm.opt.enableflags) & int(mjw.EnableBit.SLEEP)) and not bool(int(mjm.opt.disableflags) & int(mjw.DisableBit.ISLAND))
  if sleep:
    ms, pk = mjw.timed_steps(m, d, nstep, step0=args.nstep, noise_std=args.noise_std, noise_rate=args.noise_rate, per_kernel=True, plain_kernels=True)
    ns = {k: 1e6 * v / nstep / args.nworld for k, v in zip(mjw.KERNEL_NAMES, pk)}
    fwd_position = ns["fwd_pos"] + ns["collision"] + ns["make_constraint"]
    forward = fwd_position + ns["fwd_vel"] + ns["solve"]
    return {"step": forward + ns["integrate"], "step.forward": forward, "step.forward.fwd_position": fwd_position,
            "step.forward.fwd_position.collision": ns["collision"], "step.forward.fwd_position.make_constraint": ns["make_constraint"],
            "step.forward.fwd_velocity": ns["fwd_vel"], "step.forward.solve": ns["solve"], "step.euler": ns["integrate"]}
  integ = {int(mjw.IntegratorType.EULER): ("euler", mjw.euler), int(mjw.IntegratorType.RK4): ("rungekutta4", mjw.rungekutta4),
           int(mjw.IntegratorType.IMPLICIT): ("implicit", mjw.implicit), int(mjw.IntegratorType.IMPLICITFAST): ("implicit", mjw.implicit)}[int(mjm.opt.integrator)]
  # (reference order: forward.py:1342-1366 with fwd_position(factorize=False) 636-678 and fwd_acceleration(factorize=True) 1291)
  stages = [("step.forward.fwd_position.fwd_kinematics.kinematics", mjw.kinematics), ("step.forward.fwd_position.fwd_kinematics.com_pos", mjw.com_pos),
            ("step.forward.fwd_position.crb", mjw.crb), ("step.forward.fwd_position.collision", mjw.collision),
            ("step.forward.fwd_position.make_constraint", mjw.make_constraint), ("step.forward.fwd_position.transmission", mjw.transmission),
            ("step.forward.fwd_velocity.com_vel", mjw.com_vel), ("step.forward.fwd_velocity.passive", mjw.passive), ("step.forward.fwd_velocity.rne", mjw.rne),
            ("step.forward.fwd_actuation", mjw.fwd_actuation), ("step.forward.fwd_acceleration.factor_m", mjw.factor_m),
            ("step.forward.fwd_acceleration", mjw.fwd_acceleration), ("step.forward.solve", mjw.solve), ("step." + integ[0], integ[1])]
  if mjm.nsensor:
    stages.insert(6, ("step.forward.sensor_pos", mjw.sensor_pos))
    stages.insert(-1, ("step.forward.sensor_acc", mjw.sensor_acc))
  ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in stages] for _ in range(nstep)]
  for i in range(nstep):
    if mjm.nu:
      mjw.ctrl_noise(m, d, args.nstep + i, args.noise_std, args.noise_rate, centers[-1] if centers is not None else center)
    for (name, fn), (e0, e1) in zip(stages, ev[i]):
      e0.record()
      fn(m, d)
      e1.record()
  torch.cuda.synchronize()
  leaf = {name: sum(e0.elapsed_time(e1) for e0, e1 in (ev[i][k] for i in range(nstep))) * 1e6 / nstep / args.nworld for k, (name, _) in enumerate(stages)}
  trace = {}
  for name, v in leaf.items():  # a scope's time = its own launch (if it has one) + its children's
    parts = name.split(".")
    for j in range(1, len(parts) + 1):
      key = ".".join(parts[:j])
      trace[key] = trace.get(key, 0.0) + v
  return trace


def main(argv=None):
  ap = argparse.ArgumentParser(prog="mujoco_warp_amd.testspeed", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
  ap.add_argument("mjcf")
  ap.add_argument("--function", default="step")
  ap.add_argument("--nworld", type=int, default=8192)
  ap.add_argument("--nstep", type=int, default=None, help="default 1000, or the length of the --replay sequence")
  ap.add_argument("--nconmax", type=int, default=None)
  ap.add_argument("--njmax", type=int, default=None)
  ap.add_argument("--nvmax", type=int, default=None, help="capacity for awake dofs per world (reference cli.py:45)")
  ap.add_argument("--nccdmax", type=int, default=None, help="accepted for run.py compatibility (the CCD workspace is sized from nconmax here)")
  ap.add_argument("--init_asleep", type=_bool, default=False, help="initialize all trees as asleep before simulation (requires sleep enabled; reference cli.py:48)")
  ap.add_argument("--keyframe", type=int, default=0)
  ap.add_argument("--replay", default=None, help="NPZ file with a `ctrl` [nstep, nu] sequence used as the centre of the control noise "
                  "(reference cli.py:53, 153-161); nstep defaults to its length; an initial qpos/qvel in the file is applied")
  ap.add_argument("--noise_std", type=float, default=0.01)
  ap.add_argument("--noise_rate", type=float, default=0.1)
  ap.add_argument("-o", "--override", action="append", default=[])
  ap.add_argument("--format", default="human", choices=["human", "short", "json"])
  ap.add_argument("--event_trace", type=_bool, default=False)
  ap.add_argument("--measure_alloc", type=_bool, default=False)
  ap.add_argument("--measure_solver", type=_bool, default=False)
  ap.add_argument("--memory", type=_bool, default=False)
  ap.add_argument("--overflow_behavior", default="error", choices=["error", "continue"])
  ap.add_argument("--clear_warp_cache", type=_bool, default=False, help="accepted for run.py compatibility (no JIT cache here)")
  ap.add_argument("--device", default=None)
  args = ap.parse_args(argv)

  import torch

  import mujoco_warp_amd as mjw
  from mujoco_warp_amd import device as mdev

  if not torch.cuda.is_available():
    raise ValueError("testspeed available for gpu only")
  if args.device:
    mdev.set_device(args.device)
  fn = getattr(mjw, args.function)

  if args.format == "human":
    print(f"Loading model from: {args.mjcf}...\n")
  mjm = mjw.mjcf.load_xml(args.mjcf)
  if args.override:
    mjw.override_model(mjm, args.override)
  free0 = torch.cuda.mem_get_info()[0]
  m = mjw.put_model(mjm)
  mjd = mjw.MjData(mjm)
  if mjm.nkey > args.keyframe:
    mjw.mj_resetDataKeyframe(mjm, mjd, args.keyframe)
  ctrls = None
  if args.replay:
    z = np.load(args.replay)
    if "times" in z.files:  # the reference's recording format: resampled onto the model timestep (io.load_trajectory)
      ctrls = np.asarray(mjw.load_trajectory(args.replay, mjm, mjd), dtype=np.float32)
    else:  # bare [nstep, nu] sequence, one control per step
      ctrls = np.asarray(z["ctrl"], dtype=np.float32)
      if ctrls.ndim != 2 or ctrls.shape[1] != mjm.nu:
        raise ValueError(f"replay ctrl has shape {ctrls.shape}, expected [nstep, {mjm.nu}]")
      if "qpos" in z.files and z["qpos"].shape[-1] == mjm.nq:
        mjd.qpos[:] = z["qpos"][0]
      if "qvel" in z.files and z["qvel"].shape[-1] == mjm.nv:
        mjd.qvel[:] = z["qvel"][0]
    args.nstep = len(ctrls) if args.nstep is None else min(args.nstep, len(ctrls))
  if args.nstep is None:
    args.nstep = 1000
  if args.init_asleep:  # reference cli.py:167-168
    if not (int(mjm.opt.enableflags) & int(mjw.EnableBit.SLEEP)):
      raise ValueError("--init_asleep requires sleep to be enabled (-o opt.enableflags=SLEEP)")
    mjd.tree_asleep[:] = np.arange(mjm.ntree, dtype=np.int32)
  d = mjw.put_data(mjm, mjd, nworld=args.nworld, nconmax=args.nconmax, njmax=args.njmax, nvmax=args.nvmax)
  center = mjw.DeviceArray.from_numpy(np.asarray(mjd.ctrl, dtype=np.float32)) if mjm.nu else None
  centers = [mjw.DeviceArray.from_numpy(c) for c in ctrls[: args.nstep]] if ctrls is not None else None

  if args.format == "human":
    print(f"Model\n  nq: {m.nq} nv: {m.nv} nu: {m.nu} nbody: {m.nbody} ngeom: {m.ngeom}")
    print(f"  solver: {mjw.SolverType(m.opt.solver).name} cone: {mjw.ConeType(m.opt.cone).name} integrator: {mjw.IntegratorType(m.opt.integrator).name}"
          f" iterations: {m.opt.iterations} ls_iterations: {m.opt.ls_iterations} is_sparse: {m.is_sparse}")
    print(f"Data\n  nworld: {d.nworld} naconmax: {d.naconmax} njmax: {d.njmax}\n")
    print(f"Rolling out {args.nstep} steps at dt = {m.opt.timestep.numpy()[0]:.3f}...")

  # "JIT": graph capture of the benchmarked function (kernels are precompiled; nothing is compiled at run time)
  t0 = time.perf_counter()
  graph = mjw.StepGraph(m, d) if args.function == "step" else None  # (leaves the state untouched)
  torch.cuda.synchronize()
  jit_duration = time.perf_counter() - t0

  nacon, nefc, niter = [], [], []
  runtime = 0.0
  trace_ms = np.zeros(len(mjw.KERNEL_NAMES))
  for i in range(args.nstep):
    if mjm.nu:
      mjw.ctrl_noise(m, d, i, args.noise_std, args.noise_rate, centers[i] if centers is not None else center)
      torch.cuda.synchronize()
    t1 = time.perf_counter()
    if graph is not None:
      graph.launch()
    else:
      fn(m, d)
    torch.cuda.synchronize()
    runtime += time.perf_counter() - t1
    if args.measure_alloc:
      nacon.append(int(d.nacon.numpy()[0]))
      nefc.append(float(np.max(d.nefc.numpy())))
    if args.measure_solver:
      niter.append(float(np.max(d.solver_niter.numpy())))
    if args.overflow_behavior == "error":
      # every capacity bit of OverflowType aborts the run as in the reference (testspeed.py:266-279), NVMAX included: a world with more awake
      # dofs than --nvmax is flagged exactly where the reference flags it (island.py:1010-1018) although nothing is truncated here (islands are
      # solved one by one, not gathered into an nvmax-wide problem).  The two iteration-limit bits are reported, not fatal: float32 worlds at
      # the caps of `iterations` / `ls_iterations` are ordinary on the G1 (iterations = 10), and the reference's abort on them would end its own run.
      ovf = d.overflow.numpy() & 0x1FF
      if ovf.any():
        wids = np.nonzero(ovf)[0]
        print(f"\nSimulation aborted: overflow detected in {len(wids)} world{'s' if len(wids) > 1 else ''} at step {i}:", file=sys.stderr)
        for wid in wids[:10]:
          print(f"  World {wid}: {', '.join(f.name for f in mjw.OverflowType if int(ovf[wid]) & int(f))}", file=sys.stderr)
        raise RuntimeError(f"overflow (OverflowType bits {int(np.bitwise_or.reduce(ovf))}) at step {i}: raise nconmax / njmax / nvmax or pass --overflow_behavior=continue")

  trace = {}
  if args.event_trace and args.function == "step":
    trace = _event_trace(mjw, mjm, m, d, args, centers, center)

  nconverged = int(np.sum(~np.any(np.isnan(d.qpos.numpy()), axis=1)))
  steps = args.nworld * args.nstep
  model_mem, data_mem = _memory(m), _memory(d)
  total_mem = free0 - torch.cuda.mem_get_info()[0]
  if args.format == "human":
    dt = float(m.opt.timestep.numpy()[0])
    print(f"""
Summary for {d.nworld} parallel rollouts

Total JIT time: {jit_duration:.2f} s
Total simulation time: {runtime:.2f} s
Total steps per second: {steps / runtime:,.0f}
Total realtime factor: {steps * dt / runtime:,.2f} x
Total time per step: {1e9 * runtime / steps:.2f} ns
Total converged worlds: {nconverged} / {d.nworld}""")
    if trace:
      print("\nEvent trace (ns per world-step):\n")
      for k, v in trace.items():
        print(f"{'  ' * k.count('.')}{k.split('.')[-1]}: {v:.2f}")
    if args.memory:
      print(f"\nModel memory {sum(model_mem.values()) / 2**20:.2f} MiB, Data memory {sum(data_mem.values()) / 2**20:.2f} MiB, total {total_mem / 2**20:.2f} MiB")
  else:
    metrics = {
      "jit_duration": jit_duration, "run_time": runtime, "steps_per_second": steps / runtime, "converged_worlds": nconverged,
      "model_memory": sum(model_mem.values()), "data_memory": sum(data_mem.values()), "total_memory": total_mem,
      "ncon_mean": float(np.mean(nacon)) / args.nworld if nacon else 0.0,
      "ncon_p95": float(np.percentile(nacon, 95)) / args.nworld if nacon else 0.0,
      "nefc_mean": float(np.mean(nefc)) if nefc else 0.0, "nefc_p95": float(np.percentile(nefc, 95)) if nefc else 0.0,
      "solver_niter_mean": float(np.mean(niter)) if niter else 0.0, "solver_niter_p95": float(np.percentile(niter, 95)) if niter else 0.0,
    }
    metrics.update(trace)
    if args.format == "short":
      for k, v in metrics.items():
        print(f"{k}: {v}")
    else:
      print(json.dumps(metrics))
  return 0


if __name__ == "__main__":
  sys.exit(main())
