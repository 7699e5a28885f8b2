"""Condense rocprofv3 (rocpd sqlite) output dirs into one small JSON summary.

usage: python tools/summarize_profile.py gpurun_out/prof_<tag> [out.json]
Per kernel: calls, mean/min/max duration from the kernel trace; per-dispatch mean of every PMC counter
(summed over hardware instances, i.e. SEs/XCCs).  FETCH_SIZE/WRITE_SIZE are in KiB (rocprofv3 convention);
on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section) -- the
corrected byte figures are emitted next to the raw ones.
"""

import glob
import json
import os
import sqlite3
import sys
from collections import defaultdict


def _tables(cur):
  t = {}
  for (name,) in cur.execute("select name from sqlite_master where type='table'"):
    t[name.rsplit("_0000", 1)[0]] = name
  return t


def _short(n):
  n = n.replace(".kd", "")
  if "k_solve_pgs" in n:
    return "k_solve_pgs" + ("<reg>" if "Lb1" in n else "<lds>")
  if "k_solve_big" in n:
    return "k_solve_big"
  if "k_solve_newton" in n:
    return "k_solve_newton"
  if "k_solve_cgp" in n:
    return "k_solve_cgp"
  if "k_solve_cgw" in n:
    return "k_solve_cgw"
  if "k_rk4" in n:
    return "k_rk4"
  if "k_mid" in n:
    return "k_mid"
  for key in ("k_fwd_pos", "k_fwd_vel", "k_collision", "k_make_constraint", "k_solve_m", "k_solve", "k_integrate", "k_ctrl_noise", "k_contact_scan", "k_publish_contacts", "k_factor_smooth", "k_schedule_worlds"):
    if key in n:
      if key == "k_solve" and "ILi" in n:
        i = n.index("ILi")
        return "k_solve<" + n[i + 3 : i + 5].rstrip("E") + ("," + ("newton" if "Lb1" in n else "cg")) + ">"
      return key
  return n[:60]


LAST = int(os.environ.get("PROFILE_LAST", "0"))  # > 0: only the last N dispatches of every kernel (the timed windows, not the warm-up)


def kernel_trace(dbpath):
  db = sqlite3.connect(dbpath)
  cur = db.cursor()
  t = _tables(cur)
  names = {r[0]: r[1] for r in cur.execute(f"select id, kernel_name from '{t['rocpd_info_kernel_symbol']}'")}
  agg = defaultdict(list)
  for kid, s, e in cur.execute(f"select kernel_id, start, end from '{t['rocpd_kernel_dispatch']}' order by start"):
    agg[_short(names.get(kid, str(kid)))].append(e - s)
  if LAST > 0:
    agg = {k: v[-LAST:] for k, v in agg.items()}
  total = sum(sum(v) for v in agg.values())
  out = []
  for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    out.append({"kernel": k, "calls": len(v), "mean_us": sum(v) / len(v) / 1e3, "min_us": min(v) / 1e3, "max_us": max(v) / 1e3,
                "total_ms": sum(v) / 1e6, "pct": 100.0 * sum(v) / max(total, 1)})
  return out


def timeline(dbpath, ndispatch=40):
  """Start offset / duration / queue of the last dispatches: shows overlap and the gaps between dependent kernels."""
  db = sqlite3.connect(dbpath)
  cur = db.cursor()
  t = _tables(cur)
  names = {r[0]: r[1] for r in cur.execute(f"select id, kernel_name from '{t['rocpd_info_kernel_symbol']}'")}
  cols = [r[1] for r in cur.execute(f"pragma table_info('{t['rocpd_kernel_dispatch']}')")]
  qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
  rows = list(cur.execute(f"select kernel_id, start, end{', ' + qcol if qcol else ''} from '{t['rocpd_kernel_dispatch']}' order by start"))
  rows = rows[-ndispatch:]
  t0 = rows[0][1]
  return [{"kernel": _short(names.get(r[0], str(r[0]))), "start_us": (r[1] - t0) / 1e3, "dur_us": (r[2] - r[1]) / 1e3,
           "end_us": (r[2] - t0) / 1e3, "queue": (r[3] if qcol else None)} for r in rows]


def pmc(dbpath):
  db = sqlite3.connect(dbpath)
  cur = db.cursor()
  t = _tables(cur)
  names = {r[0]: r[1] for r in cur.execute(f"select id, kernel_name from '{t['rocpd_info_kernel_symbol']}'")}
  pmcn = {r[0]: r[1] for r in cur.execute(f"select id, name from '{t['rocpd_info_pmc']}'")}
  ev2k, order = {}, defaultdict(list)
  for ev, kid in cur.execute(f"select event_id, kernel_id from '{t['rocpd_kernel_dispatch']}' order by start"):
    ev2k[ev] = _short(names.get(kid, str(kid)))
    order[ev2k[ev]].append(ev)
  if LAST > 0:
    keep = set(ev for evs in order.values() for ev in evs[-LAST:])
    ev2k = {ev: k for ev, k in ev2k.items() if ev in keep}
  per = defaultdict(lambda: defaultdict(float))
  ndisp = defaultdict(set)
  for ev, pid, val in cur.execute(f"select event_id, pmc_id, value from '{t['rocpd_pmc_event']}'"):
    k = ev2k.get(ev)
    if k is None:
      continue
    per[k][pmcn.get(pid, str(pid))] += val
    ndisp[k].add(ev)
  return {k: {c: v / max(len(ndisp[k]), 1) for c, v in cs.items()} for k, cs in per.items()}


def main():
  out = sys.argv[1]
  summary = {"source": out}
  for f in glob.glob(os.path.join(out, "trace", "*.db")):
    summary["kernel_trace"] = kernel_trace(f)
    summary["timeline"] = timeline(f)
  counters = defaultdict(dict)
  for tag in ("pmc_sq", "pmc_sq2", "pmc_mfma", "pmc_fetch", "pmc_write"):
    for f in glob.glob(os.path.join(out, tag, "*.db")):
      for k, cs in pmc(f).items():
        counters[k].update(cs)
  for k, cs in counters.items():
    if "FETCH_SIZE" in cs:
      cs["fetch_bytes_raw"] = cs["FETCH_SIZE"] * 1024
      cs["fetch_bytes_gfx950_corrected"] = cs["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in cs:
      cs["write_bytes"] = cs["WRITE_SIZE"] * 1024
  summary["pmc_per_dispatch"] = counters
  text = json.dumps(summary, indent=1)
  if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text)
  print(text)


if __name__ == "__main__":
  main()
