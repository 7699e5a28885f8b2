"""Throughput of the feature scenes of the test suite at scale (not the contract bench): convex meshes, mesh stack with multi-contact,
height-field terrain, sleeping pile, sensor scene.  python tools/bench_scenes.py [nworld]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mujoco_warp_amd as mjw
from tests import test_hfield, test_mesh, test_sensor, test_sleep

nworld = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
SCENES = [("mesh scene (11 bodies, meshes vs all shapes)", test_mesh.MESH_SCENE, 64, 256), ("mesh stack (multi-contact on mesh faces)", test_mesh.MESH_STACK, 64, 256),
          ("12-vertex meshes (hill climbing)", test_mesh.GRAPH_SCENE, 64, 256), ("height-field terrain, 6 shapes", test_hfield.BUMPY, 64, 256),
          ("sleeping pile (sleep enabled)", test_sleep.PILE_XML, 48, 160), ("sensor scene (39 sensors, all stages)", test_sensor.SENSOR_XML, 16, 64)]
for name, xml, nconmax, njmax in SCENES:
  mjm = mjw.mjcf.from_xml_string(xml)
  m = mjw.put_model(mjm)
  d = mjw.make_data(mjm, nworld=nworld, nconmax=nconmax, njmax=njmax)
  rng = np.random.default_rng(0)
  q = d.qpos.numpy()
  q += rng.normal(size=q.shape).astype(np.float32) * 0.002  # decorrelate the worlds a little
  d.qpos.assign(q)
  mjw.timed_steps(m, d, 100, noise_std=0.0)
  ms, _ = mjw.timed_steps(m, d, 200, step0=100, noise_std=0.0)
  print(f"{name}: nv {mjm.nv}, {nworld} worlds: {nworld * 200 / ms * 1e3:,.0f} env-steps/s ({ms / 200 * 1e3:.0f} us/step), ncon {d.ws_ncon.numpy().mean():.1f}, "
        f"finite {bool(np.isfinite(d.qpos.numpy()).all())}, overflow bits {int(np.bitwise_or.reduce(d.overflow.numpy()))}")

# rays(): a 256-ray fan per world against the 9 primitive geoms of the ray test scene
import torch
from mujoco_warp_amd.device import DeviceArray
from tests import test_ray
mjm = mjw.mjcf.from_xml_string(test_ray.SCENE)
m = mjw.put_model(mjm)
d = mjw.make_data(mjm, nworld=nworld)
mjw.forward(m, d)
nray = 256
pnt, vec = test_ray._random_rays(nray, 3)
P, V = DeviceArray.from_numpy(pnt[None].astype(np.float32)), DeviceArray.from_numpy(vec[None].astype(np.float32))
dist, gid, nrm = DeviceArray.zeros((nworld, nray)), DeviceArray.zeros((nworld, nray), np.int32), DeviceArray.zeros((nworld, nray, 3))
for _ in range(3):
  mjw.rays(m, d, P, V, None, True, None, dist, gid, nrm)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
  mjw.rays(m, d, P, V, None, True, None, dist, gid, nrm)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"rays: {nworld} worlds x {nray} rays x {mjm.ngeom} geoms: {nworld * nray / ms * 1e-6:,.2f} G rays/s ({ms * 1e3:.0f} us per call), hits {(gid.numpy() >= 0).mean():.2f}")
