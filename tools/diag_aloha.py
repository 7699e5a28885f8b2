"""GPU box: where the HIP narrowphase and the oracle disagree along the aloha_pot lift (per re-synchronised step)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mujoco_warp_amd as mjw
from oracle import ref
from tests.test_aloha_pot import XML, find_keys, make_trajectory, _sync

cone = int(sys.argv[1]) if len(sys.argv) > 1 else 1
mjm = mjw.mjcf.load_xml(XML)
mjm.opt.cone = cone
keys = find_keys(mjm, "lift_pot")
traj = make_trajectory(mjm, keys)
s = ref.RefSim(mjm, nconmax=64, njmax=256, broadphase_filter=15)
s.reset(key=keys[0])
m = mjw.put_model(mjm)
d = mjw.make_data(mjm, nworld=2, nconmax=64, njmax=256)
mjw.reset_data_keyframe(m, d, keys[0])
shown = 0
kinds = {}
for i, ctrl in enumerate(traj):
  s.ctrl[:] = ctrl
  _sync(d, s, 2)
  mjw.step(m, d)
  s.step()
  nc, ne = int(d.ws_ncon.numpy()[1]), int(d.nefc.numpy()[1])
  if nc == s.ncon and ne == s.nefc:
    continue
  a = int(d.ws_conadr.numpy()[1])
  gg = [tuple(x) for x in d.contact.geom.numpy()[a:a + nc]]
  gs = [tuple(int(y) for y in x) for x in s.con_geom[:s.ncon]]
  key = (nc - s.ncon, ne - s.nefc)
  kinds[key] = kinds.get(key, 0) + 1
  if shown < 12:
    shown += 1
    print(f"step {i}: gpu ncon {nc} nefc {ne} ncoll {int(d.ws_ncollision.numpy()[1])} | oracle ncon {s.ncon} nefc {s.nefc} ncoll {s.ncollision}")
    print("   gpu   ", [(g, round(float(x), 7)) for g, x in zip(gg, d.contact.dist.numpy()[a:a + nc])])
    print("   oracle", [(g, round(float(x), 7)) for g, x in zip(gs, s.con_dist[:s.ncon])])
    print("   efc types gpu", np.bincount(d.efc.type.numpy()[1][:ne], minlength=8), "oracle", np.bincount(s.efc_type[:s.nefc], minlength=8))
print("mismatch kinds (dncon, dnefc):", kinds)
