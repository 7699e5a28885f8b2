"""Experiment: does stepping the batch as C independent chunks on C streams (tails of one chunk's launches overlap the bulk of
another's) beat one 8192-world launch sequence?  python tools/chunk_overlap.py [cg|newton]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import mujoco_warp_amd as mjw
solver = sys.argv[1] if len(sys.argv) > 1 else "cg"
NW, K = 8192, int(os.environ.get('K', '200'))
mjm = mjw.mjcf.load_xml(os.path.join(ROOT, "benchmarks", "humanoid", "humanoid.xml"))
mjw.override_model(mjm, [f"opt.solver={solver}"])
m = mjw.put_model(mjm)

def run(nchunk, sync_every, prio):
  nw = NW // nchunk
  ds = []
  for c in range(nchunk):
    d = mjw.make_data(mjm, nworld=nw, nconmax=24, njmax=64)
    d.world_offset = c * nw
    mjw.reset_data_keyframe(m, d, 0)
    ds.append(d)
  lo, hi = -1, 0
  streams = [torch.cuda.Stream(priority=(lo if (prio and c % 2 == 0) else hi)) for c in range(nchunk)]
  def loop(k0, k1, timed):
    for i in range(k0, k1):
      for c in range(nchunk):
        with torch.cuda.stream(streams[c]):
          mjw.ctrl_noise(m, ds[c], i)
          mjw.step(m, ds[c])
      if sync_every:
        torch.cuda.synchronize()
  loop(0, int(os.environ.get('WARM', '200')), False)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  loop(200, 200 + K, True)
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  nit = np.mean([d.solver_niter.numpy().mean() for d in ds])
  print(f"chunks {nchunk} sync_per_step {int(sync_every)} prio {int(prio)}: {NW * K / dt / 1e6:6.2f} M env-steps/s  {dt / K * 1e3:.4f} ms/step niter {nit:.2f}", flush=True)

for nchunk in tuple(int(x) for x in os.environ.get('CHUNKS', '1,2,4,8').split(',')):
  for sync_every in ((False,) if os.environ.get('NOSYNC') else (False, True)):
    for prio in ((False, True) if nchunk > 1 else (False,)):
      run(nchunk, sync_every, prio)
