"""GPU box: long rollouts of the BASELINE models through the reference's loop (hipGraph replay, per-step sync): finite states, overflow bits,
iteration statistics at the end.  python tools/long_run.py [nstep]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import mujoco_warp_amd as mjw
B = os.path.join(ROOT, "benchmarks")
nstep = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
CASES = [
  dict(name="humanoid_cg", xml=os.path.join(B, "humanoid", "humanoid.xml"), nworld=8192, nconmax=24, njmax=64, override=["opt.solver=cg"]),
  dict(name="humanoid_newton", xml=os.path.join(B, "humanoid", "humanoid.xml"), nworld=8192, nconmax=24, njmax=64),
  dict(name="unitree_g1_flat", xml=os.path.join(B, "unitree_g1", "scene_flat.xml"), nworld=4096, nconmax=48, njmax=192, replay=os.path.join(B, "unitree_g1", "shuffle_dance.npz")),
  dict(name="franka_emika_panda", xml=os.path.join(B, "franka_emika_panda", "scene.xml"), nworld=8192, nconmax=1, njmax=5),
  dict(name="clutter_synth", xml=os.path.join(B, "clutter_synth", "scene_clutter_synth.xml"), nworld=2048, nconmax=256, njmax=384, nvmax=56, override=["opt.enableflags=SLEEP"], init_asleep=True),
]
for e in CASES:
  mjm = mjw.mjcf.load_xml(e["xml"])
  if e.get("override"):
    mjw.override_model(mjm, e["override"])
  m = mjw.put_model(mjm)
  mjd = mjw.MjData(mjm)
  if mjm.nkey:
    mjw.mj_resetDataKeyframe(mjm, mjd, 0)
  center = None
  if e.get("replay"):
    ctrl = mjw.load_trajectory(e["replay"], mjm, mjd)
    center = [mjw.DeviceArray.from_numpy(np.asarray(c, dtype=np.float32)) for c in ctrl]
  if e.get("init_asleep"):
    mjd.tree_asleep[:] = np.arange(mjm.ntree, dtype=np.int32)
  d = mjw.put_data(mjm, mjd, nworld=e["nworld"], nconmax=e["nconmax"], njmax=e["njmax"], nvmax=e.get("nvmax"))
  hold = mjw.DeviceArray.from_numpy(np.asarray(mjd.ctrl, dtype=np.float32)) if mjm.nu else None
  g = mjw.StepGraph(m, d)
  n = min(nstep, len(center)) if center else nstep
  t0 = time.perf_counter()
  capped = 0
  for i in range(n):
    if mjm.nu:
      mjw.ctrl_noise(m, d, i, center=center[i] if center else hold)
    g.launch()
    if i % 500 == 499:
      capped = max(capped, int(((d.overflow.numpy() >> 9) & 1).sum()))
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  ovf = d.overflow.numpy()
  q = d.qpos.numpy()
  print(json.dumps({"name": e["name"], "nstep": n, "finite_worlds": int(np.isfinite(q).all(axis=1).sum()), "nworld": e["nworld"], "capacity_overflow_worlds": int(((ovf & 0x1FF) != 0).sum()),
                    "iteration_cap_worlds": int(((ovf >> 9) & 1).sum()), "ls_cap_worlds": int(((ovf >> 10) & 1).sum()), "niter_mean": round(float(d.solver_niter.numpy().mean()), 2),
                    "nefc_mean": round(float(np.minimum(d.nefc.numpy(), d.njmax).mean()), 1), "M_env_steps_per_s_incl_noise": round(e["nworld"] * n / dt / 1e6, 2)}), flush=True)
  del g, d
