// solve_tree.hpp -- per-(world, tree) launches of the register-resident solver for models with more than 64 dofs (see solve_body TREE)
#pragma once
#include "host.hpp"

#include "collide.hpp"
#include "smooth.hpp"
#include "solver.hpp"

template <int NV4, int NR, bool NEWTON>
__global__ void __launch_bounds__(256) k_solve_tree(MjhModel m, MjhData d, int nefc_lo, int nefc_hi) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wpb = blockDim.x / 32;
  solve_body<NV4, NR, NEWTON, 32, false, true>(m, d, smem, Blk{(int)blockIdx.x * wpb, wpb, (int)blockDim.x}, nefc_lo, nefc_hi, 0);
}
template <int NV4, int NR, bool NEWTON>
static int launch_tree_t(const MjhModel* m, const MjhData* d, hipStream_t s, int lo, int hi) {
  const SolveLayout lay = solve_layout<NV4, NR, 32, NEWTON, false, true>(d->njmax);
  size_t lds;
  const int threads = pick_block(0, sizeof(float) * lay.total, 32, &lds, true);
  if (!threads) return fail(MJH_E_UNSUPPORTED, "k_solve_tree: rows x dofs of a tree do not fit in LDS");
  HIPCHK(set_lds((k_solve_tree<NV4, NR, NEWTON>), lds));
  const int wpb = threads / 32, n = d->nworld * m->ntree;
  debug_occupancy(NEWTON ? "k_solve_tree<newton>" : "k_solve_tree<cg>", k_solve_tree<NV4, NR, NEWTON>, (n + wpb - 1) / wpb, threads, lds);
  hipLaunchKernelGGL((k_solve_tree<NV4, NR, NEWTON>), dim3((n + wpb - 1) / wpb), dim3(threads), lds, s, *m, *d, lo, hi);
  return MJH_OK;
}
template <int NR, bool NEWTON>
static int launch_tree_nv(const MjhModel* m, const MjhData* d, int nv4, hipStream_t s, int lo, int hi) {
  switch (nv4) {
    case 0:
    case 1: return launch_tree_t<1, NR, NEWTON>(m, d, s, lo, hi);
    case 2: return launch_tree_t<2, NR, NEWTON>(m, d, s, lo, hi);
    case 3: return launch_tree_t<3, NR, NEWTON>(m, d, s, lo, hi);
    case 4: return launch_tree_t<4, NR, NEWTON>(m, d, s, lo, hi);
    case 5: return launch_tree_t<5, NR, NEWTON>(m, d, s, lo, hi);
    case 6: return launch_tree_t<6, NR, NEWTON>(m, d, s, lo, hi);
    case 7: return launch_tree_t<7, NR, NEWTON>(m, d, s, lo, hi);
    default: return launch_tree_t<8, NR, NEWTON>(m, d, s, lo, hi);
  }
}
