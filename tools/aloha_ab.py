"""aloha_pot, 8192 worlds at the benchmark registry's sizes: per-launch times of the fused step for a list of Model.opt overrides
(e.g. the broadphase: python tools/aloha_ab.py "" "broadphase=1" "broadphase=2")."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mujoco_warp_amd as mjw

B = os.path.join(ROOT, "benchmarks", "aloha_pot")
for spec in (sys.argv[1:] or [""]):
  mjm = mjw.mjcf.load_xml(os.path.join(B, "scene.xml"))
  m = mjw.put_model(mjm)
  for kv in spec.split():
    k, v = kv.split("=")
    setattr(m.opt, k, int(v))
  mjd = mjw.MjData(mjm)
  mjw.mj_resetDataKeyframe(mjm, mjd, 0)
  mjw.load_trajectory(os.path.join(B, "lift_pot.npz"), mjm, mjd)
  d = mjw.put_data(mjm, mjd, nworld=8192, nconmax=24, njmax=128)
  mjw.timed_steps(m, d, 150, step0=0)
  ms, _ = mjw.timed_steps(m, d, 100, step0=150)
  _, pk = mjw.timed_steps(m, d, 50, step0=250, per_kernel=True)
  _, pp = mjw.timed_steps(m, d, 50, step0=300, per_kernel=True, plain_kernels=True)
  print(f"[{spec or 'default'}] {8192 * 100 / ms * 1e3 / 1e6:.2f} M env-steps/s back to back ({ms * 10:.0f} us/step); ncon {d.ws_ncon.numpy().mean():.2f} overflow {int(np.bitwise_or.reduce(d.overflow.numpy()))}")
  print("   fused:", {n: round(1e3 * t / 50, 1) for n, t in zip(mjw.KERNEL_NAMES, pk) if t > 0})
  print("   plain:", {n: round(1e3 * t / 50, 1) for n, t in zip(mjw.KERNEL_NAMES, pp) if t > 0})
