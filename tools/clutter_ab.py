"""GPU box: clutter_synth (configs[4] class, Newton + sleeping + init_asleep, 2048 worlds) in the bench line's window (steps 100-300) and in testspeed's
(steps 0-300), eager launches vs the hipGraph replay, 2 vs 4 solver side streams (MJH_NAUX) or none (MJH_NO_AUX); one subprocess per variant."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, time, json
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
import mujoco_warp_amd as mjw
graph_mode = sys.argv[2] == "graph"
mjm = mjw.mjcf.load_xml(os.path.join(sys.argv[1], "benchmarks", "clutter_synth", "scene_clutter_synth.xml"))
mjw.override_model(mjm, ["opt.enableflags=SLEEP"])
m = mjw.put_model(mjm)
mjd = mjw.MjData(mjm)
mjw.mj_resetDataKeyframe(mjm, mjd, 0)
mjd.tree_asleep[:] = np.arange(mjm.ntree, dtype=np.int32)
d = mjw.put_data(mjm, mjd, nworld=2048, nconmax=256, njmax=384, nvmax=56)
hold = mjw.DeviceArray.from_numpy(np.asarray(mjd.ctrl, dtype=np.float32))
g = mjw.StepGraph(m, d) if graph_mode else None
t = [0.0, 0.0]
for i in range(300):
  mjw.ctrl_noise(m, d, i, center=hold)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  g.launch() if g else mjw.step(m, d)
  torch.cuda.synchronize()
  t[i >= 100] += time.perf_counter() - t0
print(json.dumps({"M_0_100": round(2048 * 100 / t[0] / 1e6, 3), "M_100_300": round(2048 * 200 / t[1] / 1e6, 3), "us_100_300": round(t[1] / 200 * 1e6, 1),
                  "awake": float(d.ntree_awake.numpy().mean()), "nefc": float(d.nefc.numpy().mean()), "qsum": float(np.abs(d.qpos.numpy()).sum())}))
'''
for rep in range(2):
  for mode in ("eager", "graph"):
    for env_s in ("MJH_NO_AUX=1", "MJH_NAUX=2", "MJH_NAUX=4"):
      env = dict(os.environ)
      k, v = env_s.split("=")
      env[k] = v
      p = subprocess.run([sys.executable, "-c", code, ROOT, mode], env=env, capture_output=True, text=True, timeout=600)
      print(f"{mode:6s} {env_s:14s}", p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-300:], flush=True)
