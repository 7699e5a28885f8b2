"""In-kernel phase timing: where does a world's time go inside each kernel at full occupancy?

Builds libmjhip_clk.so with -DMJH_PHASE_CLOCK (lane 0 of every world adds shader-clock ticks between phase marks to a
device table), runs N humanoid steps and prints each phase's share of its kernel and the absolute ticks per world-step.
Run on the GPU box:  python tools/phase_clock.py [--solver cg|newton] [--build-only]
Round 4: `--lib` takes a library in which ONE solver unit was compiled with -DMJH_PHASE_CLOCK (tools/build_variant_fast.py <tag> solve_cg32.hip
-DMJH_PHASE_CLOCK; the units solve_cg32 / solve_newton32 / solve_newton64 carry the reader): profiles/round4_phase_*.txt.  (The unity build below
was last verified with ABI v16 and does not include the units added since.)
"""
import argparse, ctypes, os, subprocess, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "mujoco_warp_amd", "libmjhip_clk.so")

PHASES = {
  1: ("fwd_pos", ["kin: copy out", "com_pos", "crb", "factor", "kin: qpos + joint-local", "kin: level loop (+ table swap)", "kin: body mats", "kin: geoms/sites"]),
  2: ("collision", ["stage geoms", "broadphase", "narrow pass1", "window init", "pass2 stage", "write records"]),
  3: ("make_constraint", ["load", "friction+limits", "J rows fl", "contact list", "contact J", "contact rows"]),
  4: ("fwd_vel", ["load", "com_vel", "passive", "rne", "actuation", "qfrc_smooth"]),
  6: ("integrate", ["load + implicit?", "M copy + qDeriv (damping, actuators)", "factor_ld", "solve_ld", "advance"]),
  5: ("solve", ["M rows", "Ma+Minv", "J+rows", "it: update+JTf+grad", "it: Mgrad/chol", "it: conv+mv+jv", "it: linesearch",
                "it: move", "exit", "store", "it: H build (mfma)", "it: H swap + M", "it: chol factor+solve", "-", "(count) ls bracketing iterations", "(count) ls calls"]),
}


def build():
  src = os.path.join(ROOT, "mujoco_warp_amd", "csrc", "unity.hip")  # one TU: a single copy of the device counters
  subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-fno-slp-vectorize", "-Wno-unused-value",
                         "-DMJH_PHASE_CLOCK", "-o", LIB, src])


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--solver", default="cg")
  ap.add_argument("--nworld", type=int, default=8192)
  ap.add_argument("--steps", type=int, default=50)
  ap.add_argument("--build-only", action="store_true")
  ap.add_argument("--xml", default=os.path.join(ROOT, "benchmarks", "humanoid", "humanoid.xml"))
  ap.add_argument("--nconmax", type=int, default=24)
  ap.add_argument("--njmax", type=int, default=64)
  ap.add_argument("--lib", default=None, help="an instrumented library built elsewhere (one unit with -DMJH_PHASE_CLOCK, tools/build_variant_fast.py) instead of the unity build")
  ap.add_argument("--warm", type=int, default=100, help="rollout steps before the measured window")
  ap.add_argument("--iterations", type=int, default=-1, help="cap opt.iterations for the measured steps (state from the uncapped warm-up)")
  args = ap.parse_args()
  if args.lib:
    os.environ["MJH_LIB"] = os.path.abspath(args.lib)
  else:
    if args.build_only or not os.path.exists(LIB):
      build()
      if args.build_only:
        return
    os.environ["MJH_LIB"] = LIB
  import mujoco_warp_amd as mjw
  from mujoco_warp_amd import _abi

  mjm = mjw.mjcf.load_xml(args.xml)
  mjw.override_model(mjm, [f"opt.solver={args.solver}"])
  m = mjw.put_model(mjm)
  mjd = mjw.MjData(mjm)
  mjw.mj_resetDataKeyframe(mjm, mjd, 0)
  d = mjw.put_data(mjm, mjd, nworld=args.nworld, nconmax=args.nconmax, njmax=args.njmax)
  mjw.timed_steps(m, d, args.warm, step0=0)  # warm-up (default: into the steady contact regime)
  L = _abi.lib()
  L.mjh_debug_phase_ticks.argtypes = [ctypes.c_void_p, ctypes.c_int]
  if args.iterations >= 0:
    snap = {k: getattr(d, k).numpy().copy() for k in ("qpos", "qvel", "ctrl", "qacc_warmstart", "time")}
    mjw.override_model(mjm, [f"opt.iterations={args.iterations}"])
    m = mjw.put_model(mjm)
    args.steps = 1
  L.mjh_debug_phase_ticks(None, 1)
  ms, _ = mjw.timed_steps(m, d, args.steps, step0=args.warm)
  buf = np.zeros((64, 8, 16), dtype=np.uint64)
  L.mjh_debug_phase_ticks(buf.ctypes.data, 0)
  print(f"{args.solver}: {ms / args.steps * 1e3:.1f} us/step (instrumented build)")
  per = buf.astype(np.float64).sum(axis=0) / (args.steps * args.nworld)
  for k, (name, phases) in PHASES.items():
    tot = per[k].sum()
    print(f"{name}: {tot:.0f} ticks per world-step")
    for i, ph in enumerate(phases):
      print(f"    {ph:24s} {per[k, i]:9.0f}  {100 * per[k, i] / max(tot, 1):5.1f}%")


if __name__ == "__main__":
  main()
