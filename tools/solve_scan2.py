"""Solve-launch time as a function of the world count and of an iteration cap (CG, humanoid): separates prologue, bulk and tail.
Every measurement restarts from the same warmed-up state.  python tools/solve_scan2.py [cg|newton]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mujoco_warp_amd as mjw
solver = sys.argv[1] if len(sys.argv) > 1 else "cg"
PLAIN = os.environ.get("PLAIN", "1") == "1"
caps = tuple(int(x) for x in os.environ.get('CAPS', '100,32,16,8,4,2,1,0').split(','))
for nworld in tuple(int(x) for x in os.environ.get('NWORLDS', '1024,8192').split(',')):
  mjm = mjw.mjcf.load_xml(os.path.join(ROOT, "benchmarks", "humanoid", "humanoid.xml"))
  mjw.override_model(mjm, [f"opt.solver={solver}"])
  m0 = mjw.put_model(mjm)
  d = mjw.make_data(mjm, nworld=nworld, nconmax=24, njmax=64)
  mjw.reset_data_keyframe(m0, d, 0)
  mjw.timed_steps(m0, d, 200)
  snap = {k: getattr(d, k).numpy().copy() for k in ("qpos", "qvel", "ctrl", "qacc_warmstart", "time")}
  row = []
  for cap in caps:
    mjw.override_model(mjm, [f"opt.iterations={cap}"])
    m = mjw.put_model(mjm)
    for k, v in snap.items():
      getattr(d, k).assign(v)
    mjw.timed_steps(m, d, 5, step0=200)
    for k, v in snap.items():
      getattr(d, k).assign(v)
    ms, pk = mjw.timed_steps(m, d, 20, step0=200, per_kernel=True, plain_kernels=PLAIN)
    it = d.solver_niter.numpy()
    row.append((cap, pk[mjw.KERNEL_NAMES.index('solve')] / 20 * 1e3, it.mean(), it.max()))
  print(f"nworld {nworld:6d}: " + "  ".join(f"cap{c}:{t:6.1f}us(m{mn:4.1f},x{mx})" for c, t, mn, mx in row), flush=True)
  pk_all = {n: pk[i] / 20 * 1e3 for i, n in enumerate(mjw.KERNEL_NAMES)}
  print("   per-kernel us (cap 0):", {k: round(v, 1) for k, v in pk_all.items()}, flush=True)
