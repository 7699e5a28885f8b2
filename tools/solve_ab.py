"""A/B of solver-kernel variants inside ONE process on the headline workload (humanoid, 8192 worlds, CG): the knobs the library looks up at
every launch (MJH_CG_KERNEL, MJH_CGP_THREADS, MJH_CGP_LDS; set through mjh_dev_knob) are switched between timed windows, so that every variant sees the same box, the
same clocks and (nearly) the same states.

usage: python tools/solve_ab.py [--nworld 8192] [--at 5,300] [--steps 20] [--reps 3] "VAR=x VAR2=y" "..." ...
Each positional argument is one variant (space-separated env assignments; "" = library defaults).  Per window (`--at`: rollout step at
which it starts) and variant: ms per step (back to back) and the per-launch event times of the fused step, median over `--reps` rounds
(variants interleaved, the state restored before every timed run)."""
import argparse, json, os, sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import mujoco_warp_amd as mjw
import importlib

fw = importlib.import_module("mujoco_warp_amd.forward")

p = argparse.ArgumentParser()
p.add_argument("variants", nargs="*", default=[""])
p.add_argument("--nworld", type=int, default=8192)
p.add_argument("--at", default="5,300")
p.add_argument("--steps", type=int, default=20)
p.add_argument("--reps", type=int, default=3)
p.add_argument("--solver", default="cg")
p.add_argument("--json", default="")
a = p.parse_args()

mjm = mjw.mjcf.load_xml(os.path.join(ROOT, "benchmarks", "humanoid", "humanoid.xml"))
mjm.opt.solver = int({"cg": mjw.SolverType.CG, "newton": mjw.SolverType.NEWTON}[a.solver])
m = mjw.put_model(mjm)
d = mjw.make_data(mjm, nworld=a.nworld, nconmax=24, njmax=64)
mjw.reset_data_keyframe(m, d, 0)
STATE = ("qpos", "qvel", "ctrl", "qacc_warmstart", "time", "solver_niter")


from mujoco_warp_amd import _abi


def setenv(spec):  # (through the library's test hook mjh_dev_knob: it does not read the environment after load)
  changed = []
  for kv in spec.split():
    k, v = kv.split("=", 1)
    _abi.set_knob(k, v)
    changed.append(k)
  return changed


def restore(changed):
  for k in changed:
    _abi.set_knob(k, None)


out = {}
step = 0
for at in [int(x) for x in a.at.split(",")]:
  if at > step:
    fw.timed_steps(m, d, at - step, step0=step)
    step = at
  torch.cuda.synchronize()
  keep = {k: getattr(d, k).t.clone() for k in STATE}
  res = {v: [] for v in a.variants}
  for rep in range(a.reps + 1):  # (round 0 warms every variant's kernels up)
    for v in a.variants:
      for k, t in keep.items():
        getattr(d, k).t.copy_(t)
      ch = setenv(v)
      ms, _ = fw.timed_steps(m, d, a.steps, step0=step)
      for k, t in keep.items():
        getattr(d, k).t.copy_(t)
      ms2, pk = fw.timed_steps(m, d, a.steps, step0=step, per_kernel=True)
      restore(ch)
      torch.cuda.synchronize()
      if rep:
        ni = d.solver_niter.numpy()
        res[v].append((ms / a.steps, [x / a.steps * 1e3 for x in pk], float(ni.mean()), float(d.nefc.numpy().mean()),
                       int((ni < 0).sum()), bool(np.isfinite(d.qpos.numpy()).all()), int(ni.max()), float(np.percentile(ni, 99)), int(d.nefc.numpy().max())))
  for k, t in keep.items():
    getattr(d, k).t.copy_(t)
  for v in a.variants:
    r = res[v]
    msm = float(np.median([x[0] for x in r]))
    pkm = np.median(np.array([x[1] for x in r]), axis=0)
    names = fw.KERNEL_NAMES
    row = {"ms_per_step": round(msm, 4), "env_steps_per_s_M": round(a.nworld / msm / 1e3, 2), "niter": round(r[-1][2], 2), "nefc": round(r[-1][3], 1),
           "unsolved": r[-1][4], "finite": r[-1][5], "niter_max": r[-1][6], "niter_p99": r[-1][7], "nefc_max": r[-1][8], **{names[i]: round(float(pkm[i]), 1) for i in range(len(names)) if pkm[i] > 0}}
    out.setdefault(str(at), {})[v or "default"] = row
    print(f"at {at:4d} [{v or 'default':40s}]", json.dumps(row), flush=True)
if a.json:
  json.dump(out, open(a.json, "w"), indent=1)
