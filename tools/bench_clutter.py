"""BASELINE configs[4]-class line: tests/models/clutter_synth.xml at nworld = 2048, nconmax = 256, njmax = 384 (the aloha_clutter registry
sizes, benchmarks/aloha/__init__.py:46-61), Newton + elliptic + sleeping, with and without init_asleep, control noise on.
python tools/bench_clutter.py [nworld] [nstep] [newton|pgs] [iterations]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import mujoco_warp_amd as mjw
nworld = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
nstep = int(sys.argv[2]) if len(sys.argv) > 2 else 500
only = sys.argv[3] if len(sys.argv) > 3 else None          # "newton" / "pgs": run that solver's configurations only
iterations = int(sys.argv[4]) if len(sys.argv) > 4 else 0  # override opt.iterations (PGS: sweep cap; 1 = set-up cost of the solve)
mjm = mjw.mjcf.load_xml(os.path.join(ROOT, "tests", "models", "clutter_synth.xml"))
for init_asleep, solver in ((False, "newton"), (True, "newton"), (False, "pgs")):
  if only and solver != only:
    continue
  mjm = mjw.mjcf.load_xml(os.path.join(ROOT, "tests", "models", "clutter_synth.xml"))
  if iterations:
    mjw.override_model(mjm, [f"opt.iterations={iterations}"])
  if solver == "pgs":  # BASELINE configs[4] names PGS: the generic kernel; sleeping needs Newton (as in the reference), so it is off here
    mjw.override_model(mjm, ["opt.solver=pgs", "opt.enableflags=0"])
  m = mjw.put_model(mjm)
  mjd = mjw.MjData(mjm)
  mjw.mj_resetDataKeyframe(mjm, mjd, 0)
  if init_asleep:
    mjd.tree_asleep[:] = np.arange(mjm.ntree, dtype=np.int32)
  d = mjw.put_data(mjm, mjd, nworld=nworld, nconmax=256, njmax=384, nvmax=56)
  center = mjw.DeviceArray.from_numpy(np.asarray(mjd.ctrl, dtype=np.float32))
  total = 0.0
  stats = []
  for i in range(nstep):
    mjw.ctrl_noise(m, d, i, center=center)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mjw.step(m, d)
    torch.cuda.synchronize()
    total += time.perf_counter() - t0
    if i % 100 == 99:
      stats.append((i + 1, float(d.nefc.numpy().mean()), float(d.ws_ncon.numpy().mean()), float(d.ntree_awake.numpy().mean()), float(d.solver_niter.numpy().mean())))
  ovf = int(np.bitwise_or.reduce(d.overflow.numpy()))
  print(f"clutter_synth nv {mjm.nv} nworld {nworld} solver {solver} init_asleep {int(init_asleep)}: {nworld * nstep / total:,.0f} env-steps/s ({total / nstep * 1e3:.3f} ms/step, per-step sync, noise untimed), "
        f"finite {bool(np.isfinite(d.qpos.numpy()).all())}, overflow bits {ovf:#x} (NVMAX {bool(ovf & 128)})")
  for st in stats:
    print("   step %4d: nefc %.1f ncon %.1f trees awake %.1f niter %.2f" % st)
