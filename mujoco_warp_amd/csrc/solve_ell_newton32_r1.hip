// solve_ell_newton32_r1.hip -- k_solve_plus instantiations: Newton, elliptic cones, 32 lanes per world, ONE row per lane (worlds of at most
// 32 rows; one translation unit of libmjhip.so, see host.hpp)
#include "solve_tu.hpp"

int launch_solve_32_newton_ell_r1(const MjhModel* m, const MjhData* d, bool with_factor, int fuse_euler, hipStream_t s, int lo, int hi) {
  return launch_solve_32<1, true, true>(m, d, with_factor, fuse_euler, s, lo, hi);
}
