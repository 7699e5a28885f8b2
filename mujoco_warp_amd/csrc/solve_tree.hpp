// solve_tree.hpp -- per-(world, island) launches of the register-resident solver for models with more than 64 dofs (solve_body TREE)
#pragma once
#include "host.hpp"

#include "contact_rec.hpp"
#include "smooth.hpp"
#include "solver.hpp"

// SG lanes per island: 32 for islands of at most 32 dofs, 64 for 33..64 dofs; (lo, hi] = rows of the island, (nv_lo, nv_hi] = its dofs
// LOOPED: the launch serves a rare class of islands (more than 64 rows, or 33..64 dofs).  k_tree_rows lists the worlds that hold
// such an island (ws_isl_list / ws_isl_count, one atomic per listed world); a small grid walks that list, so a batch without such
// islands costs one load per wavefront instead of one empty LDS-heavy block per island slot (measured: two mostly empty launches cost
// 140 us on 8192 three-humanoid worlds) and a batch with many of them is spread evenly over the grid.
template <int NV4, int NR, bool NEWTON, int SG, bool LOOPED, bool ELL = false>
__global__ void __launch_bounds__(256) k_solve_tree(MjhModel m, MjhData d, int nefc_lo, int nefc_hi, int nv_lo, int nv_hi, int need) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wpb = blockDim.x / SG;
  if (LOOPED) {
    const int cls = need == ISL_WIDE ? ISL_LIST_WIDE : ISL_LIST_MANYROWS, nlist = d.ws_isl_count[cls];
#pragma nounroll
    for (int i = blockIdx.x; i < nlist; i += gridDim.x) {
      const int w = __builtin_amdgcn_readfirstlane(d.ws_isl_list[(size_t)cls * d.nworld + i]);  // (block-uniform)
      const int nisl = d.ws_nisland[w];
#pragma nounroll
      for (int k0 = 0; k0 < nisl; k0 += wpb) {
        int slot = w * m.ntree + k0;
        asm volatile("" : "+s"(slot));  // (keeps LICM from hoisting the body's address arithmetic out of the loops)
        solve_body<NV4, NR, NEWTON, SG, ELL, true>(m, d, smem, Blk{slot, min(wpb, nisl - k0), (int)blockDim.x}, nefc_lo, nefc_hi, 0, nv_lo, nv_hi);
      }
    }
    return;
  }
  solve_body<NV4, NR, NEWTON, SG, ELL, true>(m, d, smem, Blk{(int)blockIdx.x * wpb, wpb, (int)blockDim.x}, nefc_lo, nefc_hi, 0, nv_lo, nv_hi);
}
template <int NV4, int NR, bool NEWTON, int SG, bool LOOPED, bool ELL = false>
static int launch_tree_t(const MjhModel* m, const MjhData* d, hipStream_t s, int lo, int hi, int nv_lo, int nv_hi, int need = 0) {
  const SolveLayout lay = solve_layout<NV4, NR, SG, NEWTON, ELL, true>(d->njmax);
  size_t lds;
  int threads = pick_block(0, sizeof(float) * lay.total, SG, &lds, true);
  if (!threads) return fail(MJH_E_UNSUPPORTED, "k_solve_tree: rows x dofs of an island do not fit in LDS");
  if (LOOPED) {  // one wavefront per block
    threads = 64;
    lds = sizeof(float) * lay.total * (64 / SG);
  }
  HIPCHK(set_lds((k_solve_tree<NV4, NR, NEWTON, SG, LOOPED, ELL>), lds));
  const int wpb = threads / SG, n = d->nworld * m->ntree;
  const int grid = LOOPED ? std::min(d->nworld, 2048) : (n + wpb - 1) / wpb;
  debug_occupancy(NEWTON ? "k_solve_tree<newton>" : "k_solve_tree<cg>", k_solve_tree<NV4, NR, NEWTON, SG, LOOPED, ELL>, grid, threads, lds);
  hipLaunchKernelGGL((k_solve_tree<NV4, NR, NEWTON, SG, LOOPED, ELL>), dim3(grid), dim3(threads), lds, s, *m, *d, lo, hi, nv_lo, nv_hi, need);
  return MJH_OK;
}
// (nv_lo, nv_hi]: dofs of the islands this launch solves; nv4 >= ceil(nv_hi / 4)
template <int NR, bool NEWTON, bool LOOPED, bool ELL = false>
static int launch_tree_32(const MjhModel* m, const MjhData* d, int nv4, hipStream_t s, int lo, int hi, int need, int nv_lo = 0, int nv_hi = 32) {
  switch (nv4) {
    case 0:
    case 1: return launch_tree_t<1, NR, NEWTON, 32, LOOPED, ELL>(m, d, s, lo, hi, nv_lo, nv_hi, need);
    case 2: return launch_tree_t<2, NR, NEWTON, 32, LOOPED, ELL>(m, d, s, lo, hi, nv_lo, nv_hi, need);
    case 3: return launch_tree_t<3, NR, NEWTON, 32, LOOPED, ELL>(m, d, s, lo, hi, nv_lo, nv_hi, need);
    case 4: return launch_tree_t<4, NR, NEWTON, 32, LOOPED, ELL>(m, d, s, lo, hi, nv_lo, nv_hi, need);
    case 5: return launch_tree_t<5, NR, NEWTON, 32, LOOPED, ELL>(m, d, s, lo, hi, nv_lo, nv_hi, need);
    case 6: return launch_tree_t<6, NR, NEWTON, 32, LOOPED, ELL>(m, d, s, lo, hi, nv_lo, nv_hi, need);
    case 7: return launch_tree_t<7, NR, NEWTON, 32, LOOPED, ELL>(m, d, s, lo, hi, nv_lo, nv_hi, need);
    default: return launch_tree_t<8, NR, NEWTON, 32, LOOPED, ELL>(m, d, s, lo, hi, nv_lo, nv_hi, need);
  }
}
// the two-size row dispatch of the whole-world solver (mjhip.hip launch_solve_any), per island.  `s`: the launch of the common class (islands
// of at most 32 dofs and 64 rows); `sr`, `sr2`, `sr3`: the launches of the rare classes, which touch disjoint islands and run beside it (round 3:
// on one stream the five launches of a three-humanoid step -- each mostly tail -- took 920 us in sequence; round 6: every class a stream of
// its own -- on clutter_synth the 8..16-dof, 16..32-dof and many-row launches followed one another on ONE side stream, 324 + 134 + 134 us of
// latency chains that touch disjoint islands)
template <bool NEWTON, bool ELL = false>
static int launch_tree_all(const MjhModel* m, const MjhData* d, hipStream_t s, hipStream_t sr, hipStream_t sr2, hipStream_t sr3) {
  const int all = 0x7fffffff;
  // islands of at most 32 dofs: 2 rows per lane cover 64 rows (the common case: one block per two island slots), 6 cover 192
  // Size classes (round 3): the kernel is specialised on ceil(dofs / 4) of the WIDEST island the model can form, so a free body (6 dofs)
  // was solved with 32-wide rows of M / H and an 8-block Cholesky when the model also holds wide islands (clutter_synth: isl_nv4 = 8).
  // Models with small trees (the trees but the largest average at most 8 dofs) solve the islands of at most 8 / 16 dofs with the
  // instantiations of that size; the classes touch disjoint islands, so the wider ones go to the stream of the rare classes.
  static const bool no_classes = mjh_knob("MJH_NO_ISLAND_CLASSES") != nullptr;  // developer knob (A/B)
  const int top4 = m->isl_nv4;
  const bool small = !no_classes && m->ntree > 1 && top4 > 2 && (m->nv - m->tree_nvmax) <= 8 * (m->ntree - 1);
  if (small) {
    if (int rc = launch_tree_32<2, NEWTON, false, ELL>(m, d, 2, s, -1, 64, 0, 0, 8)) return rc;
    if (top4 > 4) {
      if (int rc = launch_tree_32<2, NEWTON, false, ELL>(m, d, 4, sr, -1, 64, 0, 8, 16)) return rc;
      if (int rc = launch_tree_32<2, NEWTON, false, ELL>(m, d, top4, sr2, -1, 64, 0, 16, 32)) return rc;
    } else if (int rc = launch_tree_32<2, NEWTON, false, ELL>(m, d, top4, sr, -1, 64, 0, 8, 32)) return rc;
  } else if (int rc = launch_tree_32<2, NEWTON, false, ELL>(m, d, top4, s, -1, 64, 0)) return rc;
  if (d->njmax > 64)
    if (int rc = launch_tree_32<6, NEWTON, true, ELL>(m, d, m->isl_nv4, sr3, 64, all, ISL_MANYROWS)) return rc;
  // islands of 33..64 dofs (several trees joined, or a wide tree): one island per wavefront, padded to 64 columns
  if (d->njmax <= 64) return launch_tree_t<16, 1, NEWTON, 64, true, ELL>(m, d, sr3, -1, all, 32, 64, ISL_WIDE);
  if (int rc = launch_tree_t<16, 2, NEWTON, 64, true, ELL>(m, d, sr3, -1, 128, 32, 64, ISL_WIDE)) return rc;
  if (d->njmax > 128) return launch_tree_t<16, 3, NEWTON, 64, true, ELL>(m, d, sr3, 128, all, 32, 64, ISL_WIDE);
  return MJH_OK;
}
