"""Phase clock of k_ccd_epa on the ALOHA scene (8192 worlds, the bench line's replay): run with a library whose mjhip.hip unit was built with
-DMJH_DBG_EPA_CLOCK (tools/build_variant_fast.py epaclk mjhip.hip -DMJH_DBG_EPA_CLOCK; MJH_LIB=...).  The kernel adds, per lane group, the
shader-clock ticks (/16) of its phases to the counters 2..7 behind the worlds' slices of Data.ws_ccd; k_ccd_reset zeroes them every step, so
after a step they hold that step's sums."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import mujoco_warp_amd as mjw
from mujoco_warp_amd import io

B = os.path.join(ROOT, "benchmarks", "aloha_pot")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
mjm = mjw.mjcf.load_xml(os.path.join(B, "scene.xml"))
m = mjw.put_model(mjm)
mjd = mjw.MjData(mjm)
mjw.mj_resetDataKeyframe(mjm, mjd, 0)
ctrl = mjw.load_trajectory(os.path.join(B, "lift_pot.npz"), mjm, mjd)
d = mjw.put_data(mjm, mjd, nworld=8192, nconmax=24, njmax=128)
for i in range(n):
  mjw.ctrl_noise(m, d, i, center=mjw.DeviceArray.from_numpy(np.asarray(ctrl[min(i, len(ctrl) - 1)], dtype=np.float32)))
  mjw.step(m, d)
torch.cuda.synchronize()
# csrc/convex.hpp ccd_layout: cnt sits behind the worlds' slices
it = min(max(int(m.opt.ccd_iterations), int(m.epa_iterations)), 64)
ccap = io._collide_ccap(int(m.npair), d.concap)
cand = ccap * 24  # (no height fields in this scene)
bmask = (cand + ccap + 4 + 3) // 4 * 4
world_stride = (bmask + 2 * ((int(m.npair) + 63) // 64) + 3) // 4 * 4
cnt = d.ws_ccd.t.reshape(-1)[world_stride * d.nworld: world_stride * d.nworld + 8].view(torch.int32).cpu().numpy()
names = {0: "max convex candidates of a world", 1: "EPA entries (penetrating pairs)", 2: "EPA proper", 3: "multi-contact: normals", 4: "match + faces", 5: "clip loop", 6: "pruning to four points", 7: "(sum of clipped polygon sizes)"}
print("ncon mean %.2f" % d.ws_ncon.numpy().mean())
for k in range(8):
  v = int(cnt[k])
  print(f"  cnt[{k}] {names[k]:36s} {v:12d}" + (f"   = {16 * v / max(int(cnt[1]), 1):9.0f} ticks per entry" if 2 <= k <= 6 else ""))
