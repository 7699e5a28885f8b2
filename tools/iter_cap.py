"""How much of the solver launch is the straggler chain?  Caps opt.iterations (the state is kept on the uncapped trajectory:
every measurement starts from the same warmed-up state) and reports the solve launch time and the iteration histogram."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mujoco_warp_amd as mjw
solver = sys.argv[1] if len(sys.argv) > 1 else "newton"
caps = (100, 4, 3, 2, 1) if solver == "newton" else (100, 40, 30, 22, 15)
mjm = mjw.mjcf.load_xml(os.path.join(ROOT, "benchmarks", "humanoid", "humanoid.xml"))
mjw.override_model(mjm, [f"opt.solver={solver}"])
m0 = mjw.put_model(mjm)
d = mjw.make_data(mjm, nworld=8192, nconmax=24, njmax=64)
mjw.reset_data_keyframe(m0, d, 0)
mjw.timed_steps(m0, d, 200)
snap = {k: getattr(d, k).numpy().copy() for k in ("qpos", "qvel", "ctrl", "qacc_warmstart", "time")}
for cap in caps:
  mjw.override_model(mjm, [f"opt.iterations={cap}"])
  m = mjw.put_model(mjm)
  for k, v in snap.items():
    getattr(d, k).assign(v)
  mjw.timed_steps(m, d, 5, step0=200)
  for k, v in snap.items():
    getattr(d, k).assign(v)
  ms, pk = mjw.timed_steps(m, d, 20, step0=200, per_kernel=True)
  it = d.solver_niter.numpy()
  hist = np.bincount(it, minlength=8)[:8]
  print(f"cap {cap:3d}: solve launch {pk[mjw.KERNEL_NAMES.index('solve')] / 20 * 1e3:6.1f} us  niter mean {it.mean():5.2f} p95 {np.percentile(it, 95):4.0f} max {it.max()}  hist {hist.tolist()}")
