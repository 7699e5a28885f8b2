// solve_tree_newton.hip -- k_solve_tree instantiations, Newton (one translation unit of libmjhip.so, see host.hpp)
#include "solve_tree.hpp"

int launch_solve_tree_newton(const MjhModel* m, const MjhData* d, hipStream_t s, hipStream_t sr, hipStream_t sr2, hipStream_t sr3) { return launch_tree_all<true>(m, d, s, sr, sr2, sr3); }
