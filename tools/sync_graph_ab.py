"""GPU box: the reference's measurement loop (control kernel + sync untimed, step + sync timed) with eager launches against hipGraph replays of the
step (the reference replays a captured graph: cli.py:262-292), for the three benchmark models.  python tools/sync_graph_ab.py [nstep]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import mujoco_warp_amd as mjw

nstep = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for name, rel, nworld, nconmax, njmax, ov in (("humanoid cg", ("humanoid", "humanoid.xml"), 8192, 24, 64, ["opt.solver=cg"]),
                                              ("unitree_g1_flat", ("unitree_g1", "scene_flat.xml"), 4096, 48, 192, []),
                                              ("franka_emika_panda", ("franka_emika_panda", "scene.xml"), 8192, 1, 5, [])):
  mjm = mjw.mjcf.load_xml(os.path.join(ROOT, "benchmarks", *rel))
  if ov:
    mjw.override_model(mjm, ov)
  m = mjw.put_model(mjm)
  d = mjw.make_data(mjm, nworld=nworld, nconmax=nconmax, njmax=njmax)
  if mjm.nkey:
    mjw.reset_data_keyframe(m, d, 0)
  mjw.timed_steps(m, d, 200)
  g = mjw.StepGraph(m, d)
  out = {}
  for rep in range(2):
    for mode in ("eager", "graph"):
      total = 0.0
      for i in range(nstep):
        mjw.ctrl_noise(m, d, 200 + i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode == "eager":
          mjw.step(m, d)
        else:
          g.launch()
        torch.cuda.synchronize()
        total += time.perf_counter() - t0
      out[mode] = total / nstep * 1e6
    print(f"{name}: eager {out['eager']:.1f} us/step ({nworld / out['eager']:.2f} M), graph {out['graph']:.1f} us/step ({nworld / out['graph']:.2f} M)")
