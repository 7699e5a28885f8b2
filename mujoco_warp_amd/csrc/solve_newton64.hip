// solve_newton64.hip -- k_solve_plus instantiations: Newton, 64 lanes per world (one translation unit of libmjhip.so, see host.hpp)
#include "solve_tu.hpp"

int launch_solve_64_newton(const MjhModel* m, const MjhData* d, int nr, bool with_factor, int fuse_euler, hipStream_t s, int lo, int hi) {
  switch (nr) {
    case 1: return launch_solve_64<1, true>(m, d, with_factor, fuse_euler, s, lo, hi);
    case 2: return launch_solve_64<2, true>(m, d, with_factor, fuse_euler, s, lo, hi);
    case 3: return launch_solve_64<3, true>(m, d, with_factor, fuse_euler, s, lo, hi);
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
