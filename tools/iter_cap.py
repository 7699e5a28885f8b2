"""How much of k_solve's duration is the straggler chain?  Caps opt.iterations and reports the fused solve launch time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mujoco_warp_amd as mjw
for cap in (100, 40, 30, 22, 15):
  mjm = mjw.mjcf.load_xml(os.path.join(ROOT, "benchmarks", "humanoid", "humanoid.xml"))
  mjw.override_model(mjm, ["opt.solver=cg", f"opt.iterations={cap}"])
  m = mjw.put_model(mjm)
  d = mjw.make_data(mjm, nworld=8192, nconmax=24, njmax=64)
  mjw.reset_data_keyframe(m, d, 0)
  mjw.timed_steps(m, d, 120)
  ms, pk = mjw.timed_steps(m, d, 100, step0=120, per_kernel=True)
  it = d.solver_niter.numpy()
  print(f"cap {cap:3d}: step {ms / 100 * 1e3:6.1f} us  solve launch {pk[mjw.KERNEL_NAMES.index('solve')] / 100 * 1e3:6.1f} us  niter mean {it.mean():5.1f} p95 {np.percentile(it, 95):4.0f} max {it.max()}")
