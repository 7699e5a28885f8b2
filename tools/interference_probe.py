"""GPU box: does stepping model A earlier in the process slow clutter_synth down?  (bench.py `configs`: clutter_synth 1.42 M alone, 1.07 M after aloha_pot.)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, time, json, gc
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
import mujoco_warp_amd as mjw
B = os.path.join(sys.argv[1], "benchmarks")
first = sys.argv[2]
def run_first():
  if first == "none": return
  name, nworld = first.split(":")
  xml = {"aloha": os.path.join(B, "aloha_pot", "scene.xml"), "humanoid_newton": os.path.join(B, "humanoid", "humanoid.xml")}[name]
  mjm = mjw.mjcf.load_xml(xml)
  m = mjw.put_model(mjm)
  mjd = mjw.MjData(mjm)
  if mjm.nkey: mjw.mj_resetDataKeyframe(mjm, mjd, 0)
  d = mjw.put_data(mjm, mjd, nworld=int(nworld), nconmax=24, njmax=128 if name == "aloha" else 64)
  for i in range(50):
    mjw.step(m, d)
  torch.cuda.synchronize()
  del d, m
  gc.collect()
  if os.environ.get("PROBE_EMPTY_CACHE"): torch.cuda.empty_cache()
run_first()
mjm = mjw.mjcf.load_xml(os.path.join(B, "clutter_synth", "scene_clutter_synth.xml"))
mjw.override_model(mjm, ["opt.enableflags=SLEEP"])
m = mjw.put_model(mjm)
mjd = mjw.MjData(mjm)
mjw.mj_resetDataKeyframe(mjm, mjd, 0)
mjd.tree_asleep[:] = np.arange(mjm.ntree, dtype=np.int32)
d = mjw.put_data(mjm, mjd, nworld=2048, nconmax=256, njmax=384, nvmax=56)
hold = mjw.DeviceArray.from_numpy(np.asarray(mjd.ctrl, dtype=np.float32))
g = mjw.StepGraph(m, d)
t = 0.0
for i in range(300):
  mjw.ctrl_noise(m, d, i, center=hold)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  g.launch()
  torch.cuda.synchronize()
  if i >= 100: t += time.perf_counter() - t0
print(json.dumps({"first": first, "M_100_300": round(2048 * 200 / t / 1e6, 3), "mem_GB": round(torch.cuda.memory_reserved() / 2**30, 2)}))
'''
for first, env_s in (("none", ""), ("aloha:8192", ""), ("aloha:8192", "MJH_NO_SIDE=1"), ("aloha:8192", "PROBE_EMPTY_CACHE=1"), ("aloha:64", ""), ("humanoid_newton:8192", "")):
  env = dict(os.environ)
  if env_s:
    k, v = env_s.split("=")
    env[k] = v
  p = subprocess.run([sys.executable, "-c", code, ROOT, first], env=env, capture_output=True, text=True, timeout=600)
  print(f"{env_s:22s}", p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-400:], flush=True)
