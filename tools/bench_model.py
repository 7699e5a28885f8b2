"""Fused-step timing of any benchmark model (not the contract bench): python tools/bench_model.py <mjcf> nworld nconmax njmax [solver]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mujoco_warp_amd as mjw
xml, nworld, nconmax, njmax = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
mjm = mjw.mjcf.load_xml(xml)
if len(sys.argv) > 5:
  mjw.override_model(mjm, [f"opt.solver={sys.argv[5]}"])
m = mjw.put_model(mjm)
d = mjw.make_data(mjm, nworld=nworld, nconmax=nconmax, njmax=njmax)
if mjm.nkey:
  mjw.reset_data_keyframe(m, d, 0)
mjw.timed_steps(m, d, 100)
ms, _ = mjw.timed_steps(m, d, 200, step0=100)
ms2, pk = mjw.timed_steps(m, d, 100, step0=300, per_kernel=True)
ms3, pk3 = mjw.timed_steps(m, d, 100, step0=400, per_kernel=True, plain_kernels=True)
n = mjw.KERNEL_NAMES
print(f"{os.path.basename(xml)} nworld {nworld} nv {mjm.nv}: {nworld * 200 / ms * 1e3:,.0f} env-steps/s  {ms / 200 * 1e3:.1f} us/step  niter {d.solver_niter.numpy().mean():.1f} nefc {d.nefc.numpy().mean():.1f}")
print("  fused us:", {k: round(v * 10, 1) for k, v in zip(n, pk) if v > 0})
print("  plain us:", {k: round(v * 10, 1) for k, v in zip(n, pk3) if v > 0})
print("  finite:", bool(np.isfinite(d.qpos.numpy()).all()), "overflow bits:", int(np.bitwise_or.reduce(d.overflow.numpy())))
