// solve_tree_cg.hip -- k_solve_tree instantiations, CG (one translation unit of libmjhip.so, see host.hpp)
#include "solve_tree.hpp"

int launch_solve_tree_cg(const MjhModel* m, const MjhData* d, hipStream_t s, hipStream_t sr, hipStream_t sr2, hipStream_t sr3) { return launch_tree_all<false>(m, d, s, sr, sr2, sr3); }
