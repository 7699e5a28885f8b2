"""Which gate of the reference's gjk_phase (collision_gjk.py:2376-2414) drops the ALOHA pot's resting contact in float32?  (round-4 verdict, item 7)

Drives the float64 oracle along the lift_pot trajectory of mujoco_warp/test_data/aloha_pot; every step its float32 twin (the same C
restatement compiled in float32, oracle/libmjref32.so) restarts from the oracle's state.  Both run their collision pass with the branch trace
of oracle/ccd.c armed.  For every step on which the twin reports fewer contacts than the oracle the pairs that differ are listed with the gate
each build left gjk_phase through, GJK's distance and the final distance.  CPU only:  python tools/ccd_branch_report.py [--cone pyramidal|elliptic]"""
import argparse, collections, os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mujoco_warp_amd as mjw
from oracle import ref
from tests.test_aloha_pot import find_keys, make_trajectory, _with_cone

ap = argparse.ArgumentParser()
ap.add_argument("--cone", default="pyramidal")
ap.add_argument("--steps", type=int, default=1001)
a = ap.parse_args()
BR = {1: "cores separated (inflate)", 2: "GJK dist > tolerance", 3: "simplex < 2 points", 4: "GJK `separated`", 5: "degenerate seed", 6: "EPA failed", 7: "EPA depth"}
mjm = mjw.mjcf.load_xml(os.path.join(ROOT, "benchmarks", "aloha_pot", "scene.xml"))
mjm = _with_cone(mjm, mjw.ConeType.PYRAMIDAL if a.cone == "pyramidal" else mjw.ConeType.ELLIPTIC)
keys = find_keys(mjm, "lift_pot")
s = ref.RefSim(mjm, nconmax=64, njmax=256, broadphase_filter=15)
t = ref.RefSim(mjm, nconmax=64, njmax=256, broadphase_filter=15, real="f32")
s.reset(key=keys[0])
t.reset(key=keys[0])
tol = float(mjm.opt.ccd_tolerance) if hasattr(mjm.opt, "ccd_tolerance") else 1e-6
lost_steps, n = 0, 0
hist = collections.Counter()
examples = []
for i, ctrl in enumerate(make_trajectory(mjm, keys)):
  if i >= a.steps:
    break
  s.ctrl[:] = ctrl
  for name in ("qpos", "qvel", "qacc_warmstart", "ctrl"):
    getattr(t, name)[:] = getattr(s, name)
  for sim in (s, t):  # the collision pass of THIS state, traced
    sim.stage("kinematics"); sim.stage("com_pos")
    sim.ccd_trace_start()
    sim.stage("collision")
  tr64 = {(r[0], r[1]): r for r in s.ccd_trace()}
  tr32 = {(r[0], r[1]): r for r in t.ccd_trace()}
  n += 1
  if t.ncon < s.ncon:
    lost_steps += 1
    for pair, r64 in tr64.items():
      r32 = tr32.get(pair)
      if r32 is None or (r64[5] > 0) != (r32[5] > 0) or (r64[7] < 0) != (r32[7] < 0):
        key = (BR.get(r64[2], r64[2]), BR.get(r32[2], r32[2]) if r32 else "not a candidate")
        hist[key] += 1
        if len(examples) < 6:
          examples.append((i, pair, r64, r32))
  s.step()
print(f"aloha lift, cone {a.cone}: {n} steps; the float32 twin holds fewer contacts than the float64 oracle on {lost_steps} of them (ccd tolerance {tol:g})")
print("pairs with a contact in float64 and none in float32, by the gate each build left gjk_phase through (float64 -> float32):")
for (b64, b32), c in hist.most_common():
  print(f"  {c:5d}  {b64:28s} -> {b32}")
for i, pair, r64, r32 in examples:
  gn = [mjm.geom_names[g] if hasattr(mjm, "geom_names") and mjm.geom_names[g] else f"geom{g}" for g in pair]
  print(f"  step {i} {gn}: float64 branch {r64[2]} dim {r64[3]} sep {r64[4]} GJK dist {r64[6]:+.3e} final {r64[7]:+.3e} | float32 "
        + (f"branch {r32[2]} dim {r32[3]} sep {r32[4]} GJK dist {r32[6]:+.3e} final {r32[7]:+.3e}" if r32 else "no candidate"))
