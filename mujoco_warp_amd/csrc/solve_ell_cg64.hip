// solve_ell_cg64.hip -- k_solve_plus instantiations: CG, elliptic cones, 64 lanes per world (one translation unit of libmjhip.so, see host.hpp)
#include "solve_tu.hpp"

int launch_solve_64_cg_ell(const MjhModel* m, const MjhData* d, int nr, bool with_factor, int fuse_euler, hipStream_t s, int lo, int hi) {
  switch (nr) {
    case 1: return launch_solve_64<1, false, true>(m, d, with_factor, fuse_euler, s, lo, hi);
    case 2: return launch_solve_64<2, false, true>(m, d, with_factor, fuse_euler, s, lo, hi);
    case 3: return launch_solve_64<3, false, true>(m, d, with_factor, fuse_euler, s, lo, hi);
    default: return fail(MJH_E_ARG, "k_solve: unsupported rows per lane");
  }
}
