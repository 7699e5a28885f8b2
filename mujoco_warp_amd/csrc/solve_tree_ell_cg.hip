// solve_tree_ell_cg.hip -- k_solve_tree instantiations, CG with elliptic friction cones (one translation unit of libmjhip.so, see host.hpp)
#include "solve_tree.hpp"

int launch_solve_tree_cg_ell(const MjhModel* m, const MjhData* d, hipStream_t s, hipStream_t sr, hipStream_t sr2, hipStream_t sr3) { return launch_tree_all<false, true>(m, d, s, sr, sr2, sr3); }
