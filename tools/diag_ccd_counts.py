"""GPU box: candidates / convex candidates / EPA entries per world and step on the aloha_pot replay (counters of Data.ws_ccd, read after
every step):  python tools/diag_ccd_counts.py [nworld] [nstep]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mujoco_warp_amd as mjw
from mujoco_warp_amd import io as mio

nworld = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nstep = int(sys.argv[2]) if len(sys.argv) > 2 else 600
folder = os.path.join(ROOT, "benchmarks", "aloha_pot")
mjm = mjw.mjcf.load_xml(os.path.join(folder, "scene.xml"))
m = mjw.put_model(mjm)
mjd = mjw.MjData(mjm)
mjw.mj_resetDataKeyframe(mjm, mjd, 0)
ctrl = np.asarray(mjw.load_trajectory(os.path.join(folder, "lift_pot.npz"), mjm, mjd), dtype=np.float32)
d = mjw.put_data(mjm, mjd, nworld=nworld, nconmax=24, njmax=128)
nstep = min(nstep, len(ctrl))
ccap = mio._collide_ccap(int(m.npair), mio.contact_cap(24))
it = min(max(int(m.opt.ccd_iterations), int(m.epa_iterations)), 64)
cand = ccap * 24
bmask = (cand + ccap + 4 + 3) // 4 * 4
stride = (bmask + 2 * ((int(m.npair) + 63) // 64) + 3) // 4 * 4
print("npair", m.npair, "ccap", ccap, "world stride", stride, "ctrl", ctrl.shape)
for i in range(nstep):
  d.ctrl.assign(np.tile(ctrl[min(i, len(ctrl) - 1)].astype(np.float32), (nworld, 1)))
  mjw.step(m, d)
  if i % 50 == 0 or i == nstep - 1:
    ws = d.ws_ccd.numpy().reshape(-1)
    cnt = ws[stride * nworld: stride * nworld + 8].view(np.int32)
    tail = ws[: stride * nworld].reshape(nworld, stride)[:, cand + ccap: cand + ccap + 3].view(np.int32)
    print(f"step {i:4d}: longest convex list {cnt[0]:3d}, EPA entries {cnt[1] / nworld:6.1f} / world; candidates {tail[:, 0].mean():6.1f} (before cap {tail[:, 1].mean():6.1f}), convex {tail[:, 2].mean():6.1f}; ncon {d.ws_ncon.numpy().mean():.1f} nefc {d.nefc.numpy().mean():.1f}"
          + (f" | EPA clock per entry (ticks): epa {16 * cnt[2] / max(cnt[1], 1):.0f}, mc normals {16 * cnt[3] / max(cnt[1], 1):.0f}, mc match+faces {16 * cnt[4] / max(cnt[1], 1):.0f}, clip {16 * cnt[5] / max(cnt[1], 1):.0f}, prune {16 * cnt[6] / max(cnt[1], 1):.0f}, clipped polygon {cnt[7] / max(cnt[1], 1):.1f} verts" if os.environ.get("MJH_LIB", "").endswith("epaclock.so") else "")
          + (f" | GJK clock per pair (ticks): set-up {1024 * cnt[5] / max(cnt[0], 1):.0f}, whole phase {1024 * cnt[6] / max(cnt[0], 1):.0f} of which supports {1024 * cnt[2] / max(cnt[0], 1):.0f}, simplex {1024 * cnt[3] / max(cnt[0], 1):.0f}, closing supports {1024 * cnt[4] / max(cnt[0], 1):.0f}" if os.environ.get("MJH_LIB", "").endswith("gjkclock.so") else "")
          + (f" | GJK iterations {cnt[2] / max(cnt[0], 1):.1f} / pair (max {cnt[3]}), hill-climb steps {cnt[4] / max(cnt[0], 1):.1f} / pair, neighbours {cnt[5] / max(cnt[0], 1):.1f} / pair" if cnt[2] and not os.environ.get("MJH_LIB", "").endswith("clock.so") else ""))
