"""Randomised-model parity (40 + 12 PGS + 16 extra-collider + 16 convex-pair seeds by default; MJH_FUZZ_SEEDS / MJH_FUZZ_PGS_SEEDS /
MJH_FUZZ_COLLIDER_SEEDS / MJH_FUZZ_CONVEX_SEEDS for more.  Last full sweep (final round-2 code): 480 + 160 clean; of 320 extra-collider seeds 318 pass and 2
(262, 315: box-box pairs, which run CCD + multi-contact by default) exceed the bounds through the same face-alignment decision as below; of 320 convex-pair seeds -- box pairs included,
through CCD + multi-contact -- 313 pass, 3 skip on the mass-matrix condition and 4 (46, 242, 275, 315) exceed the per-step bounds by EPA
facet noise or a face-alignment decision at its 1.6 mrad threshold: qpos 3e-5 .. 2.3e-3; + 16 random-convex-mesh seeds, MJH_FUZZ_MESH_SEEDS: 200 run, 182 pass, 18 skip on the row budget): random kinematic trees with mixed joint / geom / actuator types against the float64 oracle.

Round-5 sweep (final build; MJH_FUZZ_SEEDS=200, PGS 12, COLLIDER / CONVEX / MESH 160 each, MJH_FUZZ_CGP_SEEDS=300 through the pooled CG kernel): 967 pass, 22 skip,
3 exceed a bound -- convex 46 (above), collider 86 (CG: qpos 3.1e-5) and pooled 91 (CG: qpos 3.0e-5); the last two are float32 CG stopping iterations before the
float64 oracle (5 vs 8, 14 vs 16): the round-4 CG kernel exceeds the same bounds on the same seeds (qacc 3e-2 on 91 against 1e-5 for the pooled kernel), Newton passes both.
Round-6 sweep (final build: incremental Hessian in the Newton kernels of solver.hpp, knob table, class streams; same seed counts): the same 967 / 22 / 3, the three
exceeding seeds with the same figures to every printed digit (profiles/round6_fuzz_wide.log).

The fixed models (humanoid, G1, Panda, pendula, free bodies, pile) pin specific code paths; these seeds sweep the
combinations: free / ball / hinge / slide joints at random depths, limits, damping, springs, armature, friction loss,
sphere / capsule / box / ellipsoid / cylinder geoms resting on or falling to a plane, sphere-sphere / sphere-capsule /
capsule-capsule contacts between bodies, motor and position actuators, Euler and implicitfast, both solvers.
"""

import os

import numpy as np
import pytest

import mujoco_warp_amd as mjw
from conftest import relerr
from oracle import ref

pytestmark = pytest.mark.gpu


def random_model_xml(seed, more_colliders=False, convex_pairs=False, meshes=False):
  r = np.random.default_rng(seed)
  assets = []
  integrator = "implicitfast" if seed % 2 else "Euler"
  lines = [f'<mujoco><option timestep="0.003" integrator="{integrator}"/>',
           '<default><geom condim="3" friction="0.8 0.02 0.001"/><joint armature="0.02"/></default>', "<worldbody>",
           '<geom name="floor" type="plane" size="0 0 .05" contype="3" conaffinity="0"/>']
  # more_colliders: same trees and geoms, but boxes also collide with spheres and capsules (sphere_box, capsule_box) and
  # cylinders with spheres (sphere_cylinder); contype/conaffinity bits: 0 floor->round, 1 floor->rest, 2 round<->round,
  # 3 round->box, 4 sphere->cylinder, 5 box<->box (cylinder-* and ellipsoid-* pairs stay off: no collider for them)
  bits = {"sphere": (4 + 8 + 16, 1 + 4), "capsule": (4 + 8, 1 + 4), "box": (32, 2 + 8 + 32), "cylinder": (0, 2 + 16), "ellipsoid": (0, 2)}
  joints, close = [], []
  nb = int(r.integers(3, 9))
  depth = 0
  for b in range(nb):
    up = int(r.integers(0, depth + 1)) if b else 0   # pop back up the tree
    for _ in range(up):
      lines.append(close.pop())
      depth -= 1
    if depth == 0:
      pos = f"{r.uniform(-1, 1):.3f} {r.uniform(-1, 1):.3f} {r.uniform(0.25, 0.6):.3f}"
    else:
      pos = f"{r.uniform(-.1, .1):.3f} {r.uniform(-.1, .1):.3f} {r.uniform(-.25, -.12):.3f}"
    lines.append(f'<body name="b{b}" pos="{pos}">')
    close.append("</body>")
    depth += 1
    jt = r.choice(["free", "ball", "hinge", "slide", "hinge"]) if depth == 1 else r.choice(["ball", "hinge", "slide", "hinge"])
    jn = f"j{b}"
    if jt == "free":
      lines.append(f'<freejoint name="{jn}"/>')
    elif jt == "ball":
      lim = 'range="0 50" limited="true"' if r.random() < 0.5 else ""
      lines.append(f'<joint name="{jn}" type="ball" damping="{r.uniform(0, .05):.3f}" {lim}/>')
    else:
      ax = r.standard_normal(3)
      ax /= np.linalg.norm(ax)
      lim = f'range="{-r.uniform(20, 60):.1f} {r.uniform(20, 60):.1f}" limited="true"' if jt == "hinge" else f'range="-.08 .08" limited="true"'
      extra = ""
      if r.random() < 0.4:
        extra += f' stiffness="{r.uniform(.5, 3):.2f}" springref="{r.uniform(-5, 5):.2f}"'
      if r.random() < 0.3:
        extra += f' frictionloss="{r.uniform(.01, .08):.3f}"'
      if r.random() < 0.5:
        extra += f' armature="{r.uniform(.01, .05):.4f}"'
      lines.append(f'<joint name="{jn}" type="{jt}" axis="{ax[0]:.3f} {ax[1]:.3f} {ax[2]:.3f}" damping="{r.uniform(0, .2):.3f}" {lim if r.random() < .6 else ""}{extra}/>')
      joints.append((jn, jt))
    gt = r.choice(["sphere", "capsule", "capsule", "box", "ellipsoid", "cylinder"])
    s = r.uniform(.03, .07)
    # only spheres / capsules may touch each other (supported pair types); everything touches the floor
    # (floor: contype bits 0|1; spheres/capsules: contype bit 2, conaffinity bits 0|2; the rest: conaffinity bit 1 only)
    body_contact = 'contype="4" conaffinity="5"' if gt in ("sphere", "capsule") else 'contype="0" conaffinity="2"'
    if more_colliders:
      body_contact = f'contype="{bits[gt][0]}" conaffinity="{bits[gt][1]}"'
    if convex_pairs:  # everything collides with everything: cylinder-* and ellipsoid-* pairs go through GJK / EPA
      body_contact = 'contype="1" conaffinity="1"'
    if meshes and gt in ("box", "ellipsoid", "cylinder"):
      # a random convex polytope instead: 6..8 vertices with multi-contact recovery on (vertex degree <= 7), up to 14 with it off
      nvert = int(r.integers(6, 9)) if seed % 2 == 0 else int(r.integers(6, 15))
      pts = r.standard_normal((nvert, 3))
      pts = pts / np.linalg.norm(pts, axis=1, keepdims=True) * r.uniform(0.8, 1.2, (nvert, 1)) * np.array([s, s * r.uniform(.7, 1.4), s * r.uniform(.7, 1.4)])
      assets.append(f'<mesh name="m{b}" vertex="{" ".join(f"{x:.4f}" for x in pts.reshape(-1))}"/>')
      lines.append(f'<geom type="mesh" mesh="m{b}" {body_contact}/>')
    elif gt == "sphere":
      lines.append(f'<geom type="sphere" size="{s:.3f}" {body_contact}/>')
    elif gt == "capsule":
      lines.append(f'<geom type="capsule" fromto="0 0 0 {r.uniform(-.1, .1):.3f} {r.uniform(-.1, .1):.3f} {-r.uniform(.1, .2):.3f}" size="{s * .6:.3f}" {body_contact}/>')
    elif gt == "box":
      lines.append(f'<geom type="box" size="{s:.3f} {s * r.uniform(.6, 1.5):.3f} {s * r.uniform(.6, 1.5):.3f}" {body_contact}/>')
    elif gt == "ellipsoid":
      lines.append(f'<geom type="ellipsoid" size="{s:.3f} {s * 1.3:.3f} {s * .8:.3f}" {body_contact}/>')
    else:
      lines.append(f'<geom type="cylinder" size="{s:.3f} {s * 1.5:.3f}" {body_contact}/>')
  while close:
    lines.append(close.pop())
  lines.append("</worldbody>")
  if joints:
    lines.append("<actuator>")
    for jn, jt in joints[: 4]:
      if r.random() < 0.5:
        lines.append(f'<motor joint="{jn}" gear="{r.uniform(.5, 3):.2f}" ctrlrange="-1 1" ctrllimited="true"/>')
      else:
        lines.append(f'<position joint="{jn}" kp="{r.uniform(5, 40):.1f}" kv="{r.uniform(.1, 2):.2f}"/>')
    lines.append("</actuator>")
  lines.append("</mujoco>")
  if meshes:
    lines.insert(1, "<asset>" + "".join(assets) + "</asset>" + ('<option><flag multiccd="disable"/></option>' if seed % 2 else ""))
  return "\n".join(lines)


def _run_seed(seed, solver, njmax_dev=128, more_colliders=False, convex_pairs=False, meshes=False):
  mjm = mjw.mjcf.from_xml_string(random_model_xml(seed, more_colliders, convex_pairs, meshes))
  mjm.opt.solver = int(solver)
  mjm.opt.iterations, mjm.opt.ls_iterations = 100, 50
  s = ref.RefSim(mjm, nconmax=48, njmax=128, tolerance=1e-6)
  rng = np.random.default_rng(1000 + seed)
  s.qvel[:] = 0.5 * rng.standard_normal(mjm.nv)
  if mjm.nu:
    s.ctrl[:] = rng.uniform(-1, 1, mjm.nu)
  for i in range(30 + 10 * (seed % 4)):  # let things fall and touch
    s.step()
  s.forward()
  cond = np.linalg.cond(s.dense_M())
  if cond > 3e4:  # float32 cannot resolve M^-1 to the test tolerances (cond * eps); a statement about the model, not the engine
    pytest.skip(f"ill-conditioned mass matrix (cond {cond:.1e})")
  m = mjw.put_model(mjm)
  d = mjw.put_data(mjm, mjw.MjData(mjm), nworld=3, nconmax=48, njmax=njmax_dev)
  worst_q = worst_v = worst_a = 0.0
  for i in range(25):
    for name in ("qpos", "qvel", "act", "ctrl", "qacc_warmstart"):
      dst = getattr(d, name)
      if dst.size:
        dst.assign(np.tile(getattr(s, name).astype(np.float32), (d.nworld, 1)))
    s.forward()
    if s.nefc > njmax_dev:
      pytest.skip(f"model needs more than {njmax_dev} rows")
    if i == 0:
      mjw.forward(m, d)
      assert int(d.nefc.numpy()[2]) == s.nefc and int(d.ws_ncon.numpy()[2]) == s.ncon
      worst_a = relerr(d.qacc.numpy()[2], s.qacc)
    # convex pairs: deep interpenetrations (bodies spawned or pushed into each other) leave EPA's facet normal ill-determined --
    # its polytope has at most 40 vertices, float32 and float64 stop on neighbouring facets up to 6e-2 rad apart (tests/test_convex.py);
    # such states are stepped but not compared
    deep = convex_pairs and any(s.con_dist[c] < -0.004 and (int(mjm.geom_type[s.con_geom[c][0]]) in (4, 5, 7) or int(mjm.geom_type[s.con_geom[c][1]]) in (4, 5, 7))
                                and int(mjm.geom_type[s.con_geom[c][0]]) != 0 for c in range(s.ncon))
    mjw.step(m, d)
    same_rows = int(d.nefc.numpy()[1]) == s.nefc
    s.step()
    if deep:
      continue
    if not same_rows and solver == mjw.SolverType.PGS:
      continue  # a contact at the detection boundary within float32 resolution (see tests/test_pgs.py); re-synchronised next step
    worst_q = max(worst_q, relerr(d.qpos.numpy()[1], s.qpos))
    worst_v = max(worst_v, relerr(d.qvel.numpy()[1], s.qvel))
  assert np.isfinite(d.qpos.numpy()).all()
  print(f"seed {seed}: nv {mjm.nv} qacc {worst_a:.2e} qpos {worst_q:.2e} qvel {worst_v:.2e} niter {int(d.solver_niter.numpy()[1])} vs {s.solver_niter}")
  # CG stops on a float32-noisy improvement/gradient test: the converged accelerations agree less tightly than Newton's;
  # PGS stops far from its fixed point (linear convergence), one sweep more or less moves qacc by ~1e-3
  # convex pairs: an EPA normal is a facet of a <= 40-vertex polytope and float32 / float64 may stop on neighbouring facets (measured
  # 2e-3 rad on a 9 um deep capsule-ellipsoid contact, seed 22); the friction torque that difference puts on a capsule's spin axis
  # (inertia ~2e-4) moves qacc of that dof by tens of rad/s^2 while positions and velocities stay inside their bounds below
  if not convex_pairs:
    assert worst_a <= (5e-3 if solver == mjw.SolverType.NEWTON else 2e-2), worst_a
  assert worst_q <= 2e-5, worst_q
  assert worst_v <= 3e-3, worst_v


@pytest.mark.parametrize("seed", range(int(os.environ.get("MJH_FUZZ_SEEDS", "40"))))
def test_random_model_forward_and_steps(seed):
  _run_seed(seed, mjw.SolverType.NEWTON if seed % 3 else mjw.SolverType.CG)


@pytest.mark.parametrize("seed", range(int(os.environ.get("MJH_FUZZ_PGS_SEEDS", "12"))))
def test_random_model_pgs(seed):
  """PGS on the same random models: even seeds with njmax 64 (register-resident sweep), odd seeds with 128 (LDS sweep)."""
  _run_seed(seed, mjw.SolverType.PGS, 64 if seed % 2 == 0 else 128)


@pytest.mark.parametrize("seed", range(int(os.environ.get("MJH_FUZZ_COLLIDER_SEEDS", "16"))))
def test_random_model_more_colliders(seed):
  """The same random trees with boxes colliding against spheres and capsules and cylinders against spheres: random poses for
  sphere_box, capsule_box, box_box (the instantiation that carries the large colliders) and sphere_cylinder."""
  _run_seed(seed, mjw.SolverType.NEWTON if seed % 2 else mjw.SolverType.CG, more_colliders=True)


@pytest.mark.parametrize("seed", range(int(os.environ.get("MJH_FUZZ_CONVEX_SEEDS", "16"))))
def test_random_model_convex_pairs(seed):
  """The same random trees with every geom colliding with every other: ellipsoid-* and cylinder-* pairs run GJK / EPA
  (csrc/convex.hpp) next to the primitive colliders."""
  _run_seed(seed, mjw.SolverType.NEWTON if seed % 2 else mjw.SolverType.CG, convex_pairs=True)


@pytest.mark.parametrize("seed", range(int(os.environ.get("MJH_FUZZ_MESH_SEEDS", "16"))))
def test_random_model_meshes(seed):
  """The same random trees with boxes / ellipsoids / cylinders replaced by random convex polytopes (inline mesh assets), every geom
  colliding with every other: mesh support in GJK / EPA, plane-mesh, and (even seeds) multi-contact recovery on mesh faces."""
  _run_seed(seed, mjw.SolverType.NEWTON if seed % 2 else mjw.SolverType.CG, convex_pairs=True, meshes=True)


@pytest.mark.parametrize("seed", range(int(os.environ.get("MJH_FUZZ_CGP_SEEDS", "24"))))
def test_random_model_pooled_cg(seed):
  """CG through the pooled contact-basis kernel (csrc/solver_cgp.hpp; forced at this batch size by the developer knob MJH_CG_KERNEL, set through
  the library's test hook mjh_dev_knob) on the same random trees: every nv / 4 instantiation, equality-free trees with limit rows and mixed
  condim-1 / condim-3 contacts, the fused Euler and implicitfast epilogues, and -- the seeds with friction loss -- its fallback launch."""
  from mujoco_warp_amd._abi import dev_knobs

  with dev_knobs(MJH_CG_KERNEL="cgp"):
    _run_seed(seed, mjw.SolverType.CG, njmax_dev=64)
