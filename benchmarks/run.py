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


def discover():
  out = []
  for entry in sorted(os.listdir(HERE)):
    init = os.path.join(HERE, entry, "__init__.py")
    if not os.path.isfile(init):
      continue
    spec = importlib.util.spec_from_file_location(f"_bench_{entry}", init)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for b in getattr(mod, "BENCHMARKS", []):
      out.append((os.path.join(HERE, entry), dict(b)))
  return out


def main():
  ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
  ap.add_argument("-f", "--filter", default=".*", help="regex on benchmark names")
  ap.add_argument("--nstep", type=int, default=None)
  ap.add_argument("--list", action="store_true")
  args = ap.parse_args()
  rc = 0
  for folder, b in discover():
    if not re.search(args.filter, b["name"]):
      continue
    if args.list:
      print(b["name"], b)
      continue
    cmd = [sys.executable, "-m", "mujoco_warp_amd.testspeed", os.path.join(folder, b["mjcf"]), f"--nworld={b['nworld']}",
           f"--nconmax={b['nconmax']}", f"--njmax={b['njmax']}", "--format=short", "--event_trace=true", "--measure_alloc=true",
           "--measure_solver=true"]
    nstep = args.nstep if args.nstep is not None else b.get("nstep")
    if nstep is not None:
      cmd.append(f"--nstep={nstep}")
    for key in ("nvmax", "nccdmax"):
      if b.get(key) is not None:
        cmd.append(f"--{key}={b[key]}")
    if b.get("init_asleep"):
      cmd.append("--init_asleep=true")
    if b.get("replay"):
      cmd.append("--replay=" + os.path.join(folder, b["replay"]))
    for o in b.get("override", []):
      cmd += ["-o", o]
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True)
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
