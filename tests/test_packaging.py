"""The drop-in boundary as a package: the reference's console-script name, its registry format run unchanged
through benchmarks/run.py, its event-trace key names, and the library's one developer hook.  Reference: pyproject.toml:72-73,
benchmarks/run.py:128-149, benchmarks/humanoid/__init__.py, testspeed.py:75-89, 359-378."""

import os
import re
import subprocess
import sys
import textwrap

import pytest

import conftest

# the reference's registry entries for the two scenes named in BASELINE.json whose entries this repo can run, AS THE REFERENCE WRITES THEM
# (benchmarks/humanoid/__init__.py:1-16; benchmarks/aloha/__init__.py:46-61 -- data, not code; `assets` are its git checkouts)
REFERENCE_HUMANOID_REGISTRY = textwrap.dedent('''
  BENCHMARKS = [
    {
      "name": "humanoid",
      "mjcf": "humanoid.xml",
      "nworld": 8192,
      "nconmax": 24,
      "njmax": 64,
    },
    {
      "name": "three_humanoids",
      "mjcf": "three_humanoids.xml",
      "nworld": 8192,
      "nconmax": 100,
      "njmax": 192,
    },
  ]
''')
REFERENCE_CLUTTER_ENTRY = {"name": "aloha_clutter", "mjcf": "scene_clutter.xml", "nworld": 2048, "nconmax": 256, "nccdmax": 16, "njmax": 384, "nvmax": 56,
                           "override": "opt.enableflags=SLEEP", "init_asleep": True, "replay": "pick_clutter.npz", "assets": [("repo", "aloha")]}


def _run_py():
  sys.path.insert(0, os.path.join(conftest.ROOT, "benchmarks"))
  import importlib

  return importlib.import_module("run")


def test_pyproject_declares_the_reference_entry_point():
  text = open(os.path.join(conftest.ROOT, "pyproject.toml")).read()
  assert re.search(r'^mjwarp-testspeed\s*=\s*"mujoco_warp_amd\.testspeed:main"', text, flags=re.M)
  from mujoco_warp_amd import testspeed

  assert callable(testspeed.main)


def test_run_py_forwards_registry_fields_as_the_reference_does(tmp_path):
  run = _run_py()
  folder = tmp_path / "humanoid"
  folder.mkdir()
  (folder / "__init__.py").write_text(REFERENCE_HUMANOID_REGISTRY)
  found = run.discover(str(tmp_path))
  assert [b["name"] for _, b in found] == ["humanoid", "three_humanoids"]
  cmd = run.command(*found[0])
  assert cmd[1:4] == ["-m", "mujoco_warp_amd.testspeed", str(folder / "humanoid.xml")]
  for flag in ("--format=short", "--event_trace=true", "--memory=true", "--measure_solver=true", "--measure_alloc=true", "--nworld=8192", "--nconmax=24", "--njmax=64"):
    assert flag in cmd, flag
  cmd = run.command("/x", REFERENCE_CLUTTER_ENTRY)
  for flag in ("--nvmax=56", "--nccdmax=16", "--override=opt.enableflags=SLEEP", "--init_asleep=True", "--replay=/x/pick_clutter.npz"):
    assert flag in cmd, flag
  assert not any("assets" in c or c.startswith("--name") or c.startswith("--mjcf") for c in cmd)


def test_testspeed_parses_every_flag_the_reference_registry_uses():
  """argparse of the work-alike accepts what run.py forwards (no GPU: parsing stops at the device check)."""
  from mujoco_warp_amd import testspeed

  argv = ["x.xml", "--nworld=2048", "--nconmax=256", "--nccdmax=16", "--njmax=384", "--nvmax=56", "--override=opt.enableflags=SLEEP", "--init_asleep=True",
          "--replay=r.npz", "--clear_warp_cache=false", "--format=short", "--event_trace=true", "--memory=true", "--measure_solver=true", "--measure_alloc=true", "--nstep=3"]
  import torch

  if torch.cuda.is_available():
    pytest.skip("parsing is covered by the GPU run below")
  with pytest.raises(ValueError, match="gpu only"):
    testspeed.main(argv)


@pytest.mark.gpu
def test_reference_registry_entry_runs_unchanged_and_reports_the_reference_keys(tmp_path):
  """The reference's benchmarks/humanoid/__init__.py (text above) in a registry folder of its own + the model file: `run.py --registry` prints the
  reference's metric keys (testspeed.py:359-374) and its event-trace names (the @event_scope nesting of forward.py / smooth.py)."""
  folder = tmp_path / "humanoid"
  folder.mkdir()
  (folder / "__init__.py").write_text(REFERENCE_HUMANOID_REGISTRY)
  os.symlink(conftest.HUMANOID_XML, folder / "humanoid.xml")
  p = subprocess.run([sys.executable, os.path.join(conftest.ROOT, "benchmarks", "run.py"), "--registry", str(tmp_path), "-f", "^humanoid$", "--nstep", "60"],
                     capture_output=True, text=True, timeout=600, cwd=conftest.ROOT)
  assert p.returncode == 0, p.stdout + p.stderr
  out = dict(line.split(" ", 1) for line in p.stdout.splitlines() if line.startswith("humanoid."))
  for key in ("jit_duration", "run_time", "steps_per_second", "converged_worlds", "model_memory", "data_memory", "total_memory", "ncon_mean", "ncon_p95",
              "nefc_mean", "nefc_p95", "solver_niter_mean", "solver_niter_p95",
              "step", "step.forward", "step.forward.fwd_position", "step.forward.fwd_position.fwd_kinematics", "step.forward.fwd_position.fwd_kinematics.kinematics",
              "step.forward.fwd_position.fwd_kinematics.com_pos", "step.forward.fwd_position.crb", "step.forward.fwd_position.collision",
              "step.forward.fwd_position.make_constraint", "step.forward.fwd_position.transmission", "step.forward.fwd_velocity", "step.forward.fwd_velocity.com_vel",
              "step.forward.fwd_velocity.passive", "step.forward.fwd_velocity.rne", "step.forward.fwd_actuation", "step.forward.fwd_acceleration",
              "step.forward.fwd_acceleration.factor_m", "step.forward.solve", "step.euler"):
    assert "humanoid." + key in out, (key, sorted(out))
  assert int(out["humanoid.converged_worlds"]) == 8192 and float(out["humanoid.steps_per_second"]) > 1e6
  # a scope's time is the sum of what runs inside it
  f = lambda k: float(out["humanoid." + k])
  assert abs(f("step") - f("step.forward") - f("step.euler")) <= 1e-6 * f("step")
  assert f("step.forward.fwd_position") > f("step.forward.fwd_position.collision") > 0


@pytest.mark.gpu
def test_dev_knob_hook_replaces_the_environment():
  """The library snapshots MJH_* at load and never calls getenv afterwards: os.environ changes nothing in a running process, mjh_dev_knob does."""
  import mujoco_warp_amd as mjw
  from mujoco_warp_amd import _abi

  mjm = mjw.mjcf.load_xml(conftest.HUMANOID_XML)
  mjm.opt.solver = int(mjw.SolverType.CG)
  m = mjw.put_model(mjm)
  d = mjw.make_data(mjm, nworld=8192, nconmax=24, njmax=64)
  assert mjw.solver_kernel(m, d) == "cgp"
  os.environ["MJH_CG_KERNEL"] = "pair"
  try:
    assert mjw.solver_kernel(m, d) == "cgp"
  finally:
    del os.environ["MJH_CG_KERNEL"]
  with _abi.dev_knobs(MJH_CG_KERNEL="pair"):
    assert mjw.solver_kernel(m, d) == "pair"
  assert mjw.solver_kernel(m, d) == "cgp"
  with pytest.raises(_abi.EngineError):
    _abi.set_knob("NOT_A_KNOB", "1")
