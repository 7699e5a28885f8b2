// solve_newton32.hip -- Newton for models with nv <= 32 (one translation unit of libmjhip.so, see host.hpp):
//   k_solve_newton (solver_newton.hpp: MFMA Hessian, blocked Cholesky, shared J pool)  njmax <= 64
//   k_solve_plus<.., NEWTON, 32> (solver.hpp: the register-resident VALU solver)        njmax > 64, or MJH_OLD_NEWTON=1
#include "solve_tu.hpp"

#include "solver_newton.hpp"

template <int NV4, int WV>
static int launch_newton_t(const MjhModel* m, const MjhData* d, int fuse_euler, bool riders, hipStream_t s) {
  constexpr int NVR = 4 * NV4;
  constexpr int JS = (NV4 & 1) ? NVR : NVR + 4;
  // pool rows: what 4 WV wavefronts per CU leave for J after the per-world scratch, at most 2 x njmax
  const NewtonLayout l0 = newton_layout<NV4>(0);
  const int words_per_wave = kLdsPerCU / (4 * WV) / 4;
  int pool = (words_per_wave - l0.total) / JS;
  const int cap = d->njmax < 64 ? d->njmax : 64;
  pool = std::min(pool, 2 * cap);
  pool = std::max(pool, newton_min_pool<NV4>(d->njmax));
  if (const char* e = mjh_knob("MJH_NEWTON_POOL")) pool = std::max(atoi(e), newton_min_pool<NV4>(d->njmax));  // developer knob
  const NewtonLayout lay = newton_layout<NV4>(pool);
  size_t lds = sizeof(float) * lay.total;
  const int nsolve = (d->nworld + 1) / 2, nrider = riders ? nsolve : 0;
  if (riders) lds = std::max(lds, sizeof(int) * mstruct_ints(m->nv, m->nC) + sizeof(float) * fac_layout(m->nv, m->nC).total * 2);
  if (lds > (size_t)kLdsPerCU) return fail(MJH_E_UNSUPPORTED, "k_solve_newton: does not fit in LDS");
  static const int rider_pct = mjh_knob("MJH_RIDER_AT") ? atoi(mjh_knob("MJH_RIDER_AT")) : 100;  // see solve_tu.hpp
  const int rider_at = std::min(nsolve, (int)((long long)nsolve * std::max(rider_pct, 0) / 100));
  const dim3 grid(nsolve + 2 * nrider), block(64);
  if (mjh_knob("MJH_DEBUG_OCC")) {  // developer knob: resident workgroups per CU the runtime computes for this launch
    int nb = -1;
    if (pool >= 2 * cap) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_solve_newton<NV4, WV, true>, 64, lds);
    else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_solve_newton<NV4, WV, false>, 64, lds);
    fprintf(stderr, "k_solve_newton<%d,%d>: pool %d rows, LDS %zu B per wavefront, %d wavefronts per CU\n", NV4, WV, pool, lds, nb);
  }
  if (pool >= 2 * cap) {  // every pair fits: the single-turn instantiation
    HIPCHK(set_lds((k_solve_newton<NV4, WV, true>), lds));
    hipLaunchKernelGGL((k_solve_newton<NV4, WV, true>), grid, block, lds, s, *m, *d, pool, fuse_euler, nrider, rider_at);
  } else {
    HIPCHK(set_lds((k_solve_newton<NV4, WV, false>), lds));
    hipLaunchKernelGGL((k_solve_newton<NV4, WV, false>), grid, block, lds, s, *m, *d, pool, fuse_euler, nrider, rider_at);
  }
  return MJH_OK;
}
template <int WV>
static int launch_newton_w(const MjhModel* m, const MjhData* d, int fuse_euler, bool riders, hipStream_t s) {
  switch ((m->nv + 3) / 4) {
    case 0:
    case 1: return launch_newton_t<1, WV>(m, d, fuse_euler, riders, s);
    case 2: return launch_newton_t<2, WV>(m, d, fuse_euler, riders, s);
    case 3: return launch_newton_t<3, WV>(m, d, fuse_euler, riders, s);
    case 4: return launch_newton_t<4, WV>(m, d, fuse_euler, riders, s);
    case 5: return launch_newton_t<5, WV>(m, d, fuse_euler, riders, s);
    case 6: return launch_newton_t<6, WV>(m, d, fuse_euler, riders, s);
    case 7: return launch_newton_t<7, WV>(m, d, fuse_euler, riders, s);
    default: return launch_newton_t<8, WV>(m, d, fuse_euler, riders, s);
  }
}

static int launch_newton(const MjhModel* m, const MjhData* d, int fuse_euler, bool riders, hipStream_t s) {
  // two wavefronts per SIMD is the measured optimum: at three (168 VGPRs) the register allocator still spills in the Cholesky
  static const int waves = mjh_knob("MJH_NEWTON_WAVES") ? atoi(mjh_knob("MJH_NEWTON_WAVES")) : 2;  // developer knob
  return waves == 2 ? launch_newton_w<2>(m, d, fuse_euler, riders, s) : launch_newton_w<3>(m, d, fuse_euler, riders, s);
}

int launch_solve_32_newton(const MjhModel* m, const MjhData* d, int nr, bool with_factor, int fuse_euler, hipStream_t s, int lo, int hi) {
  static const bool old_path = mjh_knob("MJH_OLD_NEWTON") != nullptr;  // developer knob: A/B against the VALU solver
  switch (nr) {
    case 2:
      // (32-bit byte offsets inside the kernel: nworld * max(nv, njmax) * 4 must fit)
      if (!old_path && lo < 0 && hi >= 64 && d->njmax <= 64 && (double)d->nworld * std::max(m->nv, d->njmax) * 4.0 < 4.0e9) return launch_newton(m, d, fuse_euler, with_factor, s);
      return launch_solve_32<2, true>(m, d, with_factor, fuse_euler, s, lo, hi);
    case 6: return launch_solve_32<6, true>(m, d, with_factor, fuse_euler, s, lo, hi);
    default: return fail(MJH_E_ARG, "k_solve: unsupported rows per lane");
  }
}

#ifdef MJH_PHASE_CLOCK
// profiling variant (see solve_cg32.hip): this unit's copy of the per-phase tick sums
extern "C" __attribute__((visibility("default"))) int mjh_debug_phase_ticks(unsigned long long* out, int reset) {
  if (out) HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_ticks), sizeof(unsigned long long) * 64 * 8 * 16));
  if (reset) {
    static unsigned long long zeros[64 * 8 * 16] = {0};
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_ticks), zeros, sizeof(zeros)));
  }
  return MJH_OK;
}
#endif
