// solve_big.hip -- k_solve_big: CG / Newton for models with more than 64 dofs (one translation unit of libmjhip.so, see host.hpp)
#include "host.hpp"

#include "solver_big.hpp"

__global__ void __launch_bounds__(64) k_solve_big(MjhModel m, MjhData d, int nefc_lo) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // with constraint islands (MjhModel.tree_solve) only the worlds holding an island of more than 64 dofs come here: a small grid
  // walks their list (an LDS-heavy block per world would cost ~100 us of empty rounds)
  if (m.tree_solve) {
    const int nlist = d.ws_isl_count[ISL_LIST_GENERIC];
#pragma nounroll
    for (int i = blockIdx.x; i < nlist; i += gridDim.x) {
      int w = __builtin_amdgcn_readfirstlane(d.ws_isl_list[(size_t)ISL_LIST_GENERIC * d.nworld + i]);  // (block-uniform)
      asm volatile("" : "+s"(w));
      solve_big_body<64>(m, d, smem, Blk{w, 1, 64}, true);
    }
    return;
  }
  solve_big_body<64>(m, d, smem, Blk{(int)blockIdx.x, 1, 64}, false, nefc_lo);
}

int launch_solve_big(const MjhModel* m, const MjhData* d, hipStream_t s, int nefc_lo) {
  const BigLayout lay = big_layout(m->nv, m->nC, d->njmax, m->solver == SOL_NEWTON, m->cone == CONE_ELLIPTIC && d->nmaxpyramid > 1);
  const size_t lds = sizeof(int) * mstruct_ints(m->nv, m->nC) + sizeof(float) * lay.total;
  if (lds > (size_t)kLdsPerCU) return fail(MJH_E_UNSUPPORTED, "k_solve_big: nv / njmax do not fit in LDS");
  HIPCHK(set_lds(k_solve_big, lds));
  hipLaunchKernelGGL(k_solve_big, dim3(m->tree_solve ? std::min(d->nworld, 1024) : d->nworld), dim3(64), lds, s, *m, *d, nefc_lo);
  return MJH_OK;
}
