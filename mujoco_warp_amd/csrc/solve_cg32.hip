// solve_cg32.hip -- k_solve_plus instantiations: CG, 32 lanes per world (one translation unit of libmjhip.so, see host.hpp)
#include "solve_tu.hpp"

int launch_solve_32_cg(const MjhModel* m, const MjhData* d, int nr, bool with_factor, int fuse_euler, hipStream_t s, int lo, int hi) {
  switch (nr) {
    case 2: return launch_solve_32<2, false>(m, d, with_factor, fuse_euler, s, lo, hi);
    case 6: return launch_solve_32<6, false>(m, d, with_factor, fuse_euler, s, lo, hi);
    default: return fail(MJH_E_ARG, "k_solve: unsupported rows per lane");
  }
}

// the worlds the pooled CG kernel flagged (solver_cgp.hpp): k_solve<cg>'s body, two rows per lane
int launch_solve_32_cg_deferred(const MjhModel* m, const MjhData* d, int fuse_euler, hipStream_t s) {
  switch ((m->nv + 3) / 4) {
    case 0:
    case 1: return launch_solve_deferred_t<1, 2, false, 32>(m, d, fuse_euler, s);
    case 2: return launch_solve_deferred_t<2, 2, false, 32>(m, d, fuse_euler, s);
    case 3: return launch_solve_deferred_t<3, 2, false, 32>(m, d, fuse_euler, s);
    case 4: return launch_solve_deferred_t<4, 2, false, 32>(m, d, fuse_euler, s);
    case 5: return launch_solve_deferred_t<5, 2, false, 32>(m, d, fuse_euler, s);
    case 6: return launch_solve_deferred_t<6, 2, false, 32>(m, d, fuse_euler, s);
    case 7: return launch_solve_deferred_t<7, 2, false, 32>(m, d, fuse_euler, s);
    default: return launch_solve_deferred_t<8, 2, false, 32>(m, d, fuse_euler, s);
  }
}

#ifdef MJH_PHASE_CLOCK
// profiling variant (tools/build_variant_fast.py clk32 solve_cg32.hip -DMJH_PHASE_CLOCK; tools/phase_clock.py --lib ...): this unit's copy of
// the per-phase tick sums of solve_body
extern "C" __attribute__((visibility("default"))) int mjh_debug_phase_ticks(unsigned long long* out, int reset) {
  if (out) HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_ticks), sizeof(unsigned long long) * 64 * 8 * 16));
  if (reset) {
    static unsigned long long zeros[64 * 8 * 16] = {0};
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_ticks), zeros, sizeof(zeros)));
  }
  return MJH_OK;
}
#endif
