#!/usr/bin/env python3
"""Runs the registered benchmarks through `mujoco_warp_amd.testspeed` and prints `name.metric value` lines.

Work-alike of the reference's benchmarks/run.py for the part that matters here (discovery of `BENCHMARKS` lists in
benchmarks/*/__init__.py, one testspeed run per entry, `--format short` metrics prefixed with the benchmark name);
the reference's git checkout / uv / asset-fetching machinery is out of scope.

  python benchmarks/run.py                 # all
  python benchmarks/run.py -f humanoid     # names matching a regex
  python benchmarks/run.py -f g1 --nstep 100
"""

import argparse
import importlib.util
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def discover(registry=HERE):
  out = []
  for entry in sorted(os.listdir(registry)):
    init = os.path.join(registry, entry, "__init__.py")
    if not os.path.isfile(init):
      continue
    spec = importlib.util.spec_from_file_location(f"_bench_{entry}", init)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for b in getattr(mod, "BENCHMARKS", []):
      out.append((os.path.join(registry, entry), dict(b)))
  return out


def command(folder, b, nstep=None, event_trace=True):
  """The testspeed command of one registry entry -- the reference's rule (benchmarks/run.py:128-149): fixed flags, `replay` resolved against
  the benchmark's folder, EVERY other field forwarded as --field=value (a list: once per item), so that an entry written for the reference runs
  unchanged.  Fields of the reference's registry that configure things outside the hot path (`assets`: git checkouts; `note`) are not flags."""
  cmd = [sys.executable, "-m", "mujoco_warp_amd.testspeed", os.path.join(folder, b["mjcf"]), "--clear_warp_cache=false", "--format=short",
         f"--event_trace={'true' if event_trace else 'false'}", "--memory=true", "--measure_solver=true", "--measure_alloc=true"]
  for field, value in b.items():
    if field == "replay":
      cmd.append("--replay=" + os.path.join(folder, value))
    elif field == "nstep" and nstep is not None:
      continue
    elif field not in ("name", "assets", "mjcf", "_dir", "note"):
      for item in (value if isinstance(value, (list, tuple)) else [value]):
        cmd.append(f"--{field}={item}")
  if nstep is not None:
    cmd.append(f"--nstep={nstep}")
  return cmd


def main():
  ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
  ap.add_argument("-f", "--filter", default=".*", help="regex on benchmark names")
  ap.add_argument("--nstep", type=int, default=None)
  ap.add_argument("--list", action="store_true")
  ap.add_argument("--no-trace", action="store_true", help="skip the per-stage event trace after the rollout (profiling runs: the rollout's launches only)")
  ap.add_argument("--registry", default=HERE, help="directory of benchmark folders (default: this one); a copy of the reference's benchmarks/ works as it is")
  args = ap.parse_args()
  rc = 0
  for folder, b in discover(args.registry):
    if not re.search(args.filter, b["name"]):
      continue
    if args.list:
      print(b["name"], b)
      continue
    p = subprocess.run(command(folder, b, args.nstep, not args.no_trace), cwd=ROOT, capture_output=True, text=True)
    if p.returncode != 0:
      print(f"{b['name']}.error {p.stderr.strip().splitlines()[-1] if p.stderr.strip() else 'failed'}")
      rc = 1
      continue
    for line in p.stdout.splitlines():
      if ": " in line:
        k, v = line.split(": ", 1)
        print(f"{b['name']}.{k} {v}")
  return rc


if __name__ == "__main__":
  sys.exit(main())
