"""One BASELINE config (bench.py other_configs entry) through the reference's timer placement plus the fused step's per-launch event times:
python tools/config_ab.py unitree_g1_flat [nstep]        (MJH_LIB selects the library: A/B of builds on one box)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import mujoco_warp_amd as mjw
import bench

name = sys.argv[1]
nstep = int(sys.argv[2]) if len(sys.argv) > 2 else 200
B = os.path.join(ROOT, "benchmarks")
E = {"unitree_g1_flat": dict(name="unitree_g1_flat", xml=os.path.join(B, "unitree_g1", "scene_flat.xml"), nworld=4096, nconmax=48, njmax=192, replay=os.path.join(B, "unitree_g1", "shuffle_dance.npz")),
     "franka_emika_panda": dict(name="franka_emika_panda", xml=os.path.join(B, "franka_emika_panda", "scene.xml"), nworld=8192, nconmax=1, njmax=5),
     "aloha_pot": dict(name="aloha_pot", xml=os.path.join(B, "aloha_pot", "scene.xml"), nworld=8192, nconmax=24, njmax=128, replay=os.path.join(B, "aloha_pot", "lift_pot.npz"))}[name]
r = bench._config_run(mjw, torch, E, nstep, 100)
print(name, os.environ.get("MJH_LIB", "default"), json.dumps({k: r[k] for k in ("value", "ms_per_step", "nefc_mean", "solver_niter_mean")}))
# per-launch times of the fused step at the end state
mjm = mjw.mjcf.load_xml(E["xml"])
m = mjw.put_model(mjm)
mjd = mjw.MjData(mjm)
if mjm.nkey:
  mjw.mj_resetDataKeyframe(mjm, mjd, 0)
if E.get("replay"):
  mjw.load_trajectory(E["replay"], mjm, mjd)
d = mjw.put_data(mjm, mjd, nworld=E["nworld"], nconmax=E["nconmax"], njmax=E["njmax"])
mjw.timed_steps(m, d, 150, step0=0)
_, pk = mjw.timed_steps(m, d, 50, step0=150, per_kernel=True)
print("  fused launches (us):", {n: round(1e3 * t / 50, 1) for n, t in zip(mjw.KERNEL_NAMES, pk) if t > 0})
_, pk = mjw.timed_steps(m, d, 50, step0=200, per_kernel=True, plain_kernels=True)
print("  plain kernels (us):", {n: round(1e3 * t / 50, 1) for n, t in zip(mjw.KERNEL_NAMES, pk) if t > 0})
