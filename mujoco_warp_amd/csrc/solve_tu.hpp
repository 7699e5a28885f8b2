// solve_tu.hpp -- k_solve_plus and its launchers; included by the four solver translation units (see host.hpp).
#pragma once
#include "host.hpp"

#include "contact_rec.hpp"
#include "smooth.hpp"
#include "solver.hpp"

#ifndef MJH_N64_WAVES
#define MJH_N64_WAVES 1
#endif

template <int NV4, int NR, bool NEWTON, int SG, bool ELL = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((!NEWTON && !ELL && SG == 32 && NR == 2) ? 3 : ((NEWTON && SG == 64 && MJH_N64_WAVES > 1) ? MJH_N64_WAVES : 1), 8))) k_solve_plus(MjhModel m, MjhData d, int nsolve, int nfac, int nefc_lo, int nefc_hi, int fuse_euler, int rider_at) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wpb = blockDim.x / SG;
  // workgroups are dispatched in index order: [0, rider_at) solver (longest expected solves first), then the 2 nfac rider workgroups, then the
  // remaining (short) solves
  const int bx = (int)blockIdx.x, nrider = NEWTON ? 0 : 2 * nfac;
  const int sb = bx < rider_at ? bx : bx - nrider;  // solver workgroup index when this is one
  if (bx < rider_at || bx >= rider_at + nrider) solve_body<NV4, NR, NEWTON, SG, ELL>(m, d, smem, Blk{sb * wpb, wpb, (int)blockDim.x}, nefc_lo, nefc_hi, fuse_euler);
  // CG only: the Newton kernel holds 256 VGPRs (one wave per SIMD), which would throttle the riders too (measured
  // +120 us); for Newton they ride along with the integrator launch instead
  else if (!NEWTON) {
    const int wf = blockDim.x / 32, bi = bx - rider_at;
    if (bi < nfac) factor_smooth_body<32>(m, d, 0, smem, Blk{bi * wf, wf, (int)blockDim.x});
    else publish_body<32>(d, 1, reinterpret_cast<int*>(smem), Blk{(bi - nfac) * wf, wf, (int)blockDim.x}, m.nexplicit ? m.pair_solreffriction : nullptr);
  }
}
// SG = lanes per world: 32 (two worlds per wavefront) for nv <= 32, 64 for 32 < nv <= 64.  with_factor appends the
// L'DL-factor workgroups (fused step, CG); without it the launch is the plain `solve` stage.
template <int NV4, int NR, bool NEWTON, int SG, bool ELL = false>
static int launch_solve_t(const MjhModel* m, const MjhData* d, bool with_factor, int fuse_euler, hipStream_t s, int nefc_lo, int nefc_hi) {
  const SolveLayout lay = solve_layout<NV4, NR, SG, NEWTON, ELL>(std::min(d->njmax, nefc_hi));  // (as solve_body: rows this launch can meet)
  const FacLayout fl = fac_layout(m->nv, m->nC);
  const size_t ms_bytes = sizeof(int) * mstruct_ints(m->nv, m->nC);  // riders only: the solver keeps no shared tables
  size_t lds;
  int threads = pick_block(0, sizeof(float) * lay.total, SG, &lds, true);
  if (!threads) return fail(MJH_E_UNSUPPORTED, "k_solve: njmax x nv does not fit in LDS");
  if (const char* e = mjh_knob("MJH_SOLVE_THREADS")) {  // tuning knob (developer only)
    threads = std::max(atoi(e), SG);
    lds = sizeof(float) * lay.total * (threads / SG);
  }
  const int wpb = threads / SG, wf = threads / 32;
  with_factor = with_factor && !NEWTON;
  if (with_factor) lds = std::max(lds, ms_bytes + sizeof(float) * fl.total * wf);
  HIPCHK(set_lds((k_solve_plus<NV4, NR, NEWTON, SG, ELL>), lds));
  const int nsolve = (d->nworld + wpb - 1) / wpb, nfac = with_factor ? (d->nworld + wf - 1) / wf : 0;
  // riders (fused step, CG): factor workgroups, then as many contact-publication workgroups
  debug_occupancy(NEWTON ? "k_solve_plus<newton>" : "k_solve_plus<cg>", k_solve_plus<NV4, NR, NEWTON, SG, ELL>, nsolve + 2 * nfac, threads, lds);
  // where the riders sit in the dispatch order, in per cent of the solver workgroups (developer knob; 100 = after all of them, the round-1 layout).
  // Measured (round 3, humanoid CG, two interleaved rounds on one box): 100 -> 215.9 / 215.3 us per launch, 75 -> 280.4 / 280.7, 55 -> 268.6 / 267.6,
  // 35 -> 279.1 / 279.0: riders dispatched among the solver workgroups cost four times what they cost in the launch's tail
  static const int rider_pct = mjh_knob("MJH_RIDER_AT") ? atoi(mjh_knob("MJH_RIDER_AT")) : 100;
  const int rider_at = std::min(nsolve, (int)((long long)nsolve * std::max(rider_pct, 0) / 100));
  hipLaunchKernelGGL((k_solve_plus<NV4, NR, NEWTON, SG, ELL>), dim3(nsolve + 2 * nfac), dim3(threads), lds, s, *m, *d, nsolve, nfac, nefc_lo, nefc_hi, fuse_euler, rider_at);
  return MJH_OK;
}
// The fallback launch behind the pooled CG kernel (solver_cgp.hpp): every lane group scans SG worlds' flags with one load and solves the worlds
// flagged solver_niter = -1 one after the other -- in the common case none, and the launch is a few workgroups that load and leave.  A kernel of
// its own (instantiated in solve_cg32.hip only): as a mode of k_solve_plus the second inlined copy of solve_body cost EVERY instantiation registers
// (the 64-lane Newton kernel went from 207 to 280 and from two wavefronts per SIMD to one: G1 8.2 -> 6.9 M env-steps/s, measured in round 5).
template <int NV4, int NR, bool NEWTON, int SG, bool ELL = false>
__global__ void __launch_bounds__(256) k_solve_deferred(MjhModel m, MjhData d, int fuse_euler) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wpb = blockDim.x / SG;
  const int lig = threadIdx.x & (SG - 1), gid = (int)blockIdx.x * wpb + (int)threadIdx.x / SG;
  for (int c0 = gid * SG; c0 < d.nworld; c0 += (int)gridDim.x * wpb * SG) {
    unsigned long long todo = gballot<SG>(c0 + lig < d.nworld && d.solver_niter[c0 + lig] == -1);
    while (todo) {
      const int bq = __ffsll((long long)todo) - 1;
      todo &= todo - 1ull;
      solve_body<NV4, NR, NEWTON, SG, ELL>(m, d, smem, Blk{0, wpb, (int)blockDim.x}, -1, 0x7fffffff, fuse_euler, 0, 0x7fffffff, c0 + bq);
      gsync();
    }
  }
}
template <int NV4, int NR, bool NEWTON, int SG, bool ELL = false>
static int launch_solve_deferred_t(const MjhModel* m, const MjhData* d, int fuse_euler, hipStream_t s) {
  const SolveLayout lay = solve_layout<NV4, NR, SG, NEWTON, ELL>(d->njmax);
  size_t lds;
  const int threads = pick_block(0, sizeof(float) * lay.total, SG, &lds, true);
  if (!threads) return fail(MJH_E_UNSUPPORTED, "k_solve_deferred: njmax x nv does not fit in LDS");
  HIPCHK(set_lds((k_solve_deferred<NV4, NR, NEWTON, SG, ELL>), lds));
  const int wpb = threads / SG;
  hipLaunchKernelGGL((k_solve_deferred<NV4, NR, NEWTON, SG, ELL>), dim3(std::max(1, (d->nworld + wpb * SG - 1) / (wpb * SG))), dim3(threads), lds, s, *m, *d, fuse_euler);
  return MJH_OK;
}
template <int NR, bool NEWTON, bool ELL = false>
static int launch_solve_32(const MjhModel* m, const MjhData* d, bool wf, int fe, hipStream_t s, int lo, int hi) {
  switch ((m->nv + 3) / 4) {  // kernels are specialised on ceil(nv/4): no padded matrix columns
    case 0:
    case 1: return launch_solve_t<1, NR, NEWTON, 32, ELL>(m, d, wf, fe, s, lo, hi);
    case 2: return launch_solve_t<2, NR, NEWTON, 32, ELL>(m, d, wf, fe, s, lo, hi);
    case 3: return launch_solve_t<3, NR, NEWTON, 32, ELL>(m, d, wf, fe, s, lo, hi);
    case 4: return launch_solve_t<4, NR, NEWTON, 32, ELL>(m, d, wf, fe, s, lo, hi);
    case 5: return launch_solve_t<5, NR, NEWTON, 32, ELL>(m, d, wf, fe, s, lo, hi);
    case 6: return launch_solve_t<6, NR, NEWTON, 32, ELL>(m, d, wf, fe, s, lo, hi);
    case 7: return launch_solve_t<7, NR, NEWTON, 32, ELL>(m, d, wf, fe, s, lo, hi);
    default: return launch_solve_t<8, NR, NEWTON, 32, ELL>(m, d, wf, fe, s, lo, hi);
  }
}
template <int NR, bool NEWTON, bool ELL = false>
static int launch_solve_64(const MjhModel* m, const MjhData* d, bool wf, int fe, hipStream_t s, int lo, int hi) {
  const int nv4 = (m->nv + 3) / 4;  // rounded up to an instantiated size (lanes past nv hold identity rows)
  if (nv4 <= 9) return launch_solve_t<9, NR, NEWTON, 64, ELL>(m, d, wf, fe, s, lo, hi);
  if (nv4 <= 10) return launch_solve_t<10, NR, NEWTON, 64, ELL>(m, d, wf, fe, s, lo, hi);
  if (nv4 <= 12) return launch_solve_t<12, NR, NEWTON, 64, ELL>(m, d, wf, fe, s, lo, hi);
  if (nv4 <= 14) return launch_solve_t<14, NR, NEWTON, 64, ELL>(m, d, wf, fe, s, lo, hi);
  return launch_solve_t<16, NR, NEWTON, 64, ELL>(m, d, wf, fe, s, lo, hi);
}
