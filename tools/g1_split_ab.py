"""GPU box: the G1 replay (4096 worlds, reference timer placement) with the 64-lane solver's row classes one after the other vs beside one
another (MJH_SOLVE64_SPLIT, read once per process: one subprocess per variant, interleaved twice)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, time, json
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
import mujoco_warp_amd as mjw
B = os.path.join(sys.argv[1], "benchmarks")
name = sys.argv[2]
cfg = {"g1": dict(xml=os.path.join(B, "unitree_g1", "scene_flat.xml"), nworld=4096, nconmax=48, njmax=192, replay=os.path.join(B, "unitree_g1", "shuffle_dance.npz")),
       "three": dict(xml=os.path.join(B, "humanoid", "three_humanoids.xml"), nworld=8192, nconmax=100, njmax=192)}[name]
mjm = mjw.mjcf.load_xml(cfg["xml"])
m = mjw.put_model(mjm)
mjd = mjw.MjData(mjm)
if mjm.nkey: mjw.mj_resetDataKeyframe(mjm, mjd, 0)
center = None
if cfg.get("replay"):
  ctrl = mjw.load_trajectory(cfg["replay"], mjm, mjd)
  center = [mjw.DeviceArray.from_numpy(np.asarray(c, dtype=np.float32)) for c in ctrl[:400]]
d = mjw.put_data(mjm, mjd, nworld=cfg["nworld"], nconmax=cfg["nconmax"], njmax=cfg["njmax"])
tot = 0.0
for i in range(400):
  mjw.ctrl_noise(m, d, i, center=center[i] if center else None)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  mjw.step(m, d)
  torch.cuda.synchronize()
  if i >= 100: tot += time.perf_counter() - t0
print(json.dumps({"M_env_steps_per_s": round(cfg["nworld"] * 300 / tot / 1e6, 3), "us_per_step": round(tot / 300 * 1e6, 1), "niter": float(d.solver_niter.numpy().mean()), "nefc": float(d.nefc.numpy().mean()),
                  "finite": bool(np.isfinite(d.qpos.numpy()).all()), "qpos_sum": float(np.abs(d.qpos.numpy()).sum())}))
'''
variants = [v for v in (sys.argv[1:] or ["MJH_SOLVE64_SPLIT=0", "MJH_SOLVE64_SPLIT=1"])]
for rep in range(2):
  for name in ("g1",):
    for v in variants:
      env = dict(os.environ)
      k, val = v.split("=")
      env[k] = val
      p = subprocess.run([sys.executable, "-c", code, ROOT, name], env=env, capture_output=True, text=True, timeout=600)
      print(name, v, p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-300:], flush=True)
