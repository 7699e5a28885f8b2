"""GPU box: ALOHA scene, per-launch time of the fused step with the NXN and the SAP candidate generators (same candidates by construction)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import mujoco_warp_amd as mjw
nworld = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
mjm = mjw.mjcf.load_xml(os.path.join(ROOT, "benchmarks", "aloha_pot", "scene.xml"))
for bp in (0, 1, 0, 1):
  m = mjw.put_model(mjm)
  m.opt.broadphase = bp
  mjd = mjw.MjData(mjm)
  mjw.mj_resetDataKeyframe(mjm, mjd, 3)
  ctrl = mjw.load_trajectory(os.path.join(ROOT, "benchmarks", "aloha_pot", "lift_pot.npz"), mjm, mjd)
  d = mjw.put_data(mjm, mjd, nworld=nworld, nconmax=24, njmax=128)
  centers = [mjw.DeviceArray.from_numpy(np.asarray(c, dtype=np.float32)) for c in ctrl[:300]]
  tot = 0.0
  for i in range(300):
    mjw.ctrl_noise(m, d, i, center=centers[i])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mjw.step(m, d)
    torch.cuda.synchronize()
    if i >= 100:
      tot += time.perf_counter() - t0
  print(f"broadphase {bp}: {tot / 200 * 1e3:.3f} ms/step, {nworld * 200 / tot / 1e6:.2f} M env-steps/s, ncollision mean {d.ws_ncollision.numpy().mean():.2f} ncon {d.ws_ncon.numpy().mean():.2f} ovf {int(np.bitwise_or.reduce(d.overflow.numpy()))}", flush=True)
  del d
