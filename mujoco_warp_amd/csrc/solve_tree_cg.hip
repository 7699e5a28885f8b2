// solve_tree_cg.hip -- k_solve_tree instantiations, CG (one translation unit of libmjhip.so, see host.hpp)
#include "solve_tree.hpp"

// nv4 = ceil(max tree dofs / 4); rows per tree in (lo, hi]: 2 rows per lane cover 64, 6 cover 192
int launch_solve_tree_cg(const MjhModel* m, const MjhData* d, int nv4, int nr, hipStream_t s, int lo, int hi) {
  if (nr == 2) return launch_tree_nv<2, false>(m, d, nv4, s, lo, hi);
  if (nr == 6) return launch_tree_nv<6, false>(m, d, nv4, s, lo, hi);
  return fail(MJH_E_ARG, "k_solve_tree: unsupported rows per lane");
}
