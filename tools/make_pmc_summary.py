"""profiles/roundN_pmc_<solver>.json from a tools/profile.sh summary: HBM traffic per launch of every kernel, MFMA counters.

usage: python tools/make_pmc_summary.py gpurun_out/prof_<tag>/summary.json profiles/roundN_pmc_<solver>.json <solver> [window label]
FETCH_SIZE / WRITE_SIZE are collected in their own rocprofv3 --pmc passes (tools/profile.sh), both in KiB.  On gfx950 FETCH_SIZE
under-reports WIDE coalesced reads (16 bytes per lane) by 2x (MI355X_MICROARCH.md); other access widths are uncalibrated there.  Round 3
calibrates per kernel instead of doubling everything: a kernel whose reads are 4-byte lane loads is taken at face value -- checked on
k_fwd_pos, whose only Data read is qpos (8192 x 28 x 4 = 0.92 MB: raw 0.86 MB, doubled 1.72 MB) --, a kernel that reads mostly with
16-byte lane loads is doubled, and the solver (J rows by float4, everything else by dword) is reported as the [raw, 2 x raw] bracket
with the J share doubled as the point estimate.
"""
import json, sys
src, dst = sys.argv[1], sys.argv[2]
s = json.load(open(src))
out = {"source": src, "solver": sys.argv[3] if len(sys.argv) > 3 else None, "kernels": {}}
# read width of each kernel's Data loads (csrc): gcopy / scalar loads = 4 bytes per lane; the solvers stage J with float4 loads
WIDE_SHARE = {"k_fwd_pos": 0.0, "k_mid": 0.0, "k_integrate": 0.0, "k_fwd_vel": 0.0, "k_collision": 0.0, "k_make_constraint": 0.15, "k_publish_contacts": 0.0,
              "k_factor_smooth": 0.0, "k_ctrl_noise": 0.0}
for k, c in s.get("pmc_per_dispatch", {}).items():
  if "fetch_bytes_raw" in c:
    raw = c["fetch_bytes_raw"]
    share = WIDE_SHARE.get(k, 0.75 if k.startswith("k_solve") else None)  # (k_solve_cgp too: J rows by float4)  # solver: J is ~3/4 of what it reads
    est = raw * (1.0 + share) if share is not None else None
    out["kernels"][k] = {"fetch_bytes_raw": raw, "fetch_bytes_x2": 2 * raw, "fetch_bytes_calibrated": est, "wide_load_share_assumed": share,
                         "write_bytes_per_launch": c.get("write_bytes"),
                         "hbm_bytes_per_launch": (est if est is not None else 2 * raw) + c.get("write_bytes", 0.0),
                         "valu_insts_per_launch": c.get("SQ_INSTS_VALU"), "busy_cycles_sum": c.get("SQ_BUSY_CYCLES"),
                         "mfma_mops_f32": c.get("SQ_INSTS_VALU_MFMA_MOPS_F32"), "mfma_busy_cycles": c.get("SQ_VALU_MFMA_BUSY_CYCLES"), "mfma_insts": c.get("SQ_INSTS_MFMA"),
                         "lds_idx_active": c.get("SQ_LDS_IDX_ACTIVE"), "wave_cycles": c.get("SQ_WAVE_CYCLES"), "wait_any": c.get("SQ_WAIT_ANY"), "waves": c.get("SQ_WAVES")}
    k = out["kernels"][k]
    # What bounds the kernel (round 4).  SQ_BUSY_CYCLES is summed over the 32 shader engines: busy / 32 = the kernel's duration in shader
    # cycles.  A wave64 VALU instruction occupies its SIMD-32 for 2 cycles (MI355X_MICROARCH.md "Wave scheduling", profiles/round2_ubench.txt:
    # v_fma_f32 2.06), 1,024 SIMDs; the LDS pipe is one per CU (256); SQ_WAVE_CYCLES / SQ_WAIT_ANY are in quad-cycles, so waves per SIMD =
    # 4 x WAVE_CYCLES / (1,024 x duration) and the wait share is the plain ratio WAIT_ANY / WAVE_CYCLES.
    dur = (k["busy_cycles_sum"] or 0.0) / 32.0
    if dur > 0:
      if k["valu_insts_per_launch"] is not None:
        k["valu_issue_frac"] = k["valu_insts_per_launch"] * 2.0 / (1024.0 * dur)
      if k["lds_idx_active"] is not None:
        k["lds_busy_frac"] = k["lds_idx_active"] / (256.0 * dur)
      if k["wave_cycles"]:
        k["waves_per_simd"] = 4.0 * k["wave_cycles"] / (1024.0 * dur)  # (SQ_WAVE_CYCLES and SQ_WAIT_ANY tick once per 4 cycles; SQ_BUSY_CYCLES counts cycles: it reproduces the launch durations)
        if k["wait_any"] is not None:
          k["wait_frac"] = k["wait_any"] / k["wave_cycles"]
      k["duration_cycles"] = dur
dom = [k for k in out["kernels"] if k.startswith("k_solve<") or k in ("k_solve_newton", "k_solve_cgp", "k_solve_cgw")]
dom.sort(key=lambda k: -(out["kernels"][k].get("busy_cycles_sum") or 0.0))
if dom:
  out["k_solve_hbm_bytes_per_launch"] = out["kernels"][dom[0]]["hbm_bytes_per_launch"]
  out["k_solve_kernel"] = dom[0]
if len(sys.argv) > 4:
  out["window"] = sys.argv[4]  # which steps of the rollout the passes covered, e.g. "steps 10-60 (nefc 20)" / "steps 300-350 (steady, nefc 45)"
for t in s.get("kernel_trace", []):
  if t["kernel"] in out["kernels"]:
    out["kernels"][t["kernel"]]["mean_us"] = t["mean_us"]
    k = out["kernels"][t["kernel"]]
    if k.get("hbm_bytes_per_launch"):
      k["hbm_frac"] = k["hbm_bytes_per_launch"] / (t["mean_us"] * 1e-6) / 8.0e12  # of the 8 TB/s spec peak
    if k.get("mfma_busy_cycles"):
      # MFMA pipe utilisation: SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1,024 SIMDs) / (kernel duration x 2.4 GHz x 1,024 SIMDs); and the
      # arithmetic rate from the instruction count (v_mfma_f32_32x32x1_2b_f32: 2 blocks x 32 x 32 x 1 MACs = 4,096 flop per instruction)
      k["mfma_util"] = k["mfma_busy_cycles"] / (t["mean_us"] * 2400.0 * 1024.0)
      if k.get("mfma_insts"):
        k["mfma_tflops"] = k["mfma_insts"] * 4096.0 / (t["mean_us"] * 1e-6) / 1e12
        k["mfma_peak_note"] = "dense f32 MFMA peak of MI355X = 157 TFLOP/s (MI355X_MICROARCH.md); the H build is one phase of this kernel (16 % of it)"
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out)[:600])
