// solver.hpp -- persistent per-world constraint solver (Newton and CG, pyramidal/frictionless cones).
//
// Reference: solver.py:3671-3743 (solve/_solve), 3525-3620 (_solver_iteration), 3622-3668 (init_context),
// 835-1347 (_linesearch_iterative_kernel), 1698-1822 (_update_constraint_efc), 1912-1947 (qfrc_constraint),
// 3061-3220 (_update_gradient), 2365-2440 (JTDAJ), 2567-2603 (Cholesky solve), 3283-3450 (CG), 3454-3497
// (_solve_done).  The reference runs ~12 launches per iteration inside a CUDA conditional-graph while loop and
// keeps J/H/vectors in global memory; every world iterates until ALL worlds converge.
//
// MI355X mapping: one 32-lane group (half a wavefront) owns a world for the WHOLE solve.  J (njmax x JS) and all
// solver vectors are LDS resident; lane i keeps row i of M, of the matrix being factored (H = M + J^T D J for
// Newton, M itself for CG) and column i of its Cholesky factor in VGPRs (the kernel is LDS-capacity bound, so
// VGPRs are free).  The factorisation is a right-looking Cholesky whose pivot column is broadcast through one
// LDS line, the triangular solves broadcast with v_readlane, and every row reduction of the line search is a DPP
// row_shr/row_bcast tree.  The kernel is specialised on NV4 = ceil(nv/4): all matrix loops are fully unrolled
// over exactly NVR = 4*NV4 columns.  Each world leaves the loop as soon as ITS convergence test passes.
// HBM traffic is the algorithmic minimum: J, D, aref, M, three nv-vectors in; qacc, qfrc_constraint, Ma, force,
// state out.
#pragma once
#include "dev_common.hpp"
#include "smooth.hpp"

// ---- DPP reduction over a 32-lane group; every lane receives the total ---------------------------------
// G = 32: two worlds per wavefront (nv <= 32); G = 64: one world per wavefront (32 < nv <= 64)
template <int G>
DEV float bcastg(float v, int k) {  // value of lane k of this lane's group (k compile-time)
  const int a = __builtin_amdgcn_readlane(__float_as_int(v), k);
  if (G == 64) return __int_as_float(a);
  const int b = __builtin_amdgcn_readlane(__float_as_int(v), k + 32);
  return __int_as_float((threadIdx.x & 32) ? b : a);
}
template <int G>
DEV float gsumg(float v) {
  v = dpp_add_f<0x111, 0xf, 0xf>(v);  // row_shr:1
  v = dpp_add_f<0x112, 0xf, 0xf>(v);  // row_shr:2
  v = dpp_add_f<0x114, 0xf, 0xf>(v);  // row_shr:4
  v = dpp_add_f<0x118, 0xf, 0xf>(v);  // row_shr:8  -> lane 15 of each row holds the row sum
  v = dpp_add_f<0x142, 0xa, 0xf>(v);  // row_bcast:15 into rows 1 and 3 -> lanes 31 / 63 hold the 32-lane sums
  if (G == 64) {
    v = dpp_add_f<0x143, 0xc, 0xf>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave sum
    return bcastg<64>(v, 63);
  }
  // every lane reads lane 31 of its 32-lane group through the LDS crossbar (ds_swizzle, no memory access): one DS
  // instruction instead of two v_readlane + v_cndmask -- the kernel is VALU-issue bound (measured -5 % kernel time)
  return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x3E0));
}
// N sums at once: the five DPP steps run over all N values before the next step starts, so each value's dependent DPP chain
// (14 cycles per step for a lone wavefront, tools/ubench.hip) is covered by the other values' issue slots, and the N crossbar
// broadcasts share one wait.  Measured (tools/ubench_ls.hip): a line-search iteration -- nine sums -- drops from 1470 cycles.
template <int G, int N>
DEV void gsumg_n(float (&v)[N]) {
#ifdef MJH_F64_LS
  // experiment: the cross-lane sums of the line search / CG scalars accumulated in float64
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double x = (double)v[i];
#pragma unroll
    for (int off = G / 2; off >= 1; off >>= 1) x += __shfl_xor(x, off, G);
    v[i] = (float)x;
  }
  return;
#endif
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = dpp_add_f<0x111, 0xf, 0xf>(v[i]);  // row_shr:1
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = dpp_add_f<0x112, 0xf, 0xf>(v[i]);  // row_shr:2
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = dpp_add_f<0x114, 0xf, 0xf>(v[i]);  // row_shr:4
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = dpp_add_f<0x118, 0xf, 0xf>(v[i]);  // row_shr:8
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = dpp_add_f<0x142, 0xa, 0xf>(v[i]);  // row_bcast:15 into rows 1 and 3
  if (G == 64) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = bcastg<64>(dpp_add_f<0x143, 0xc, 0xf>(v[i]), 63);  // row_bcast:31, lane 63 holds the sum
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v[i]), 0x3E0));
  }
}
// The same sums without the LDS crossbar (round 5, 32-lane groups): after the four row_shr steps lane 15 of each 16-lane row holds its row's
// sum; v_permlane16_swap hands both rows' registers to both rows and two row_newbcast:15 reads add the two row sums -- the same bits in
// every lane.  Six (round 5: seven) VALU instructions per value instead of five + one ds_swizzle: a reduction is no LDS round trip any more (the CG
// iteration had 4-5 of them on its dependent chain, and they were 23 of its 59 DS instructions).
template <int N>
DEV void gsum32_valu_n(float (&v)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = dpp_add_f<0x111, 0xf, 0xf>(v[i]);  // row_shr:1
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = dpp_add_f<0x112, 0xf, 0xf>(v[i]);  // row_shr:2
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = dpp_add_f<0x114, 0xf, 0xf>(v[i]);  // row_shr:4
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = dpp_add_f<0x118, 0xf, 0xf>(v[i]);  // row_shr:8
  // (round 6: add the two rows' copies first, broadcast once -- lane 15 then holds even-row sum + odd-row sum, the same two operands in the
  // same order as the two broadcasts + add this replaces: bit-identical, one VALU instruction per value fewer)
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i]), false, false);
    v[i] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);  // every lane: even row's copy + odd row's copy of its lane position
  }
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[i]), 0x15F, 0xf, 0xf, true));  // row_newbcast:15
}
// RED selects the broadcast of a group sum: 0 = gsumg_n (ds_swizzle), 1 = gsum32_valu_n (G = 32 only)
template <int G, int N, int RED>
DEV void gsum_sel(float (&v)[N]) {
  if (RED == 1 && G == 32) gsum32_valu_n<N>(v);
  else gsumg_n<G, N>(v);
}
// (A butterfly all-reduce -- quad_perm, row_half_mirror, row_mirror, v_permlane16_swap or ds_swizzle -- is two VALU ops
// shorter and passes in isolation, but inside k_solve it broke parity with both cross-row variants; not pursued.)
// division for step-size candidates and ratios that only steer the search (v_rcp_f32, 1 ulp; an IEEE divide is ~10 ops)
DEV float fast_div(float x, float y) { return x * __builtin_amdgcn_rcpf(y != 0.0f ? y : MJ_MINVAL); }

// ---- dense Cholesky with lane i owning row i (NVR <= 32), all indices compile-time ----------------------
// h: row i of the SPD matrix on entry, row i of L on exit (entries above the diagonal are junk);
// lt: column i of L below the diagonal (for the transposed solve); rdiag = 1 / L[i][i].
// `col` is an LDS scratch of max(64, 8*JS) floats private to the group.  Lanes >= nv must hold identity rows.
template <int NVR, int JS, int G>
DEV void chol_factor_rows(float (&h)[NVR], float (&lt)[NVR], float& rdiag, float* col, int lig) {
  const bool own = lig < NVR;
  const int ligc = own ? lig : NVR - 1;
  rdiag = 1.0f;
#pragma unroll
  for (int j = 0; j < NVR; ++j) {  // right-looking; pivot column broadcast through LDS (double buffered)
    float* cb = col + (j & 1) * G;
    // (unconditional: lanes past the matrix write slots nobody reads; a branch here lets LLVM sink the updates below across it)
    cb[lig] = h[j];
    gsync();
    // pivot by v_readlane (off the LDS round trip); 1/sqrt = v_rsq_f32 + one Newton step (~0.5 ulp)
    const float pv = fmaxf(bcastg<G>(h[j], j), MJ_MINVAL);
    float inv = __builtin_amdgcn_rsqf(pv);
    inv = inv * (1.5f - 0.5f * pv * inv * inv);
    const float piv = pv * inv;
    const float lij = (lig == j) ? piv : h[j] * inv;
    h[j] = lij;
    rdiag = (lig == j) ? inv : rdiag;
    const float t = lij * inv;
#pragma unroll
    for (int k = j + 1; k < NVR; ++k) h[k] -= t * cb[k];
  }
  gsync();
#pragma unroll
  for (int c0 = 0; c0 < NVR; c0 += 8) {  // column i of L: rows pass through an 8-row LDS tile
    if (lig >= c0 && lig < c0 + 8 && own) {
#pragma unroll
      for (int c4 = 0; c4 < NVR / 4; ++c4)
        *reinterpret_cast<float4*>(col + (lig - c0) * JS + 4 * c4) = make_float4(h[4 * c4], h[4 * c4 + 1], h[4 * c4 + 2], h[4 * c4 + 3]);
    }
    gsync();
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      if (c0 + kk < NVR) {
        const float v = col[kk * JS + ligc];
        lt[c0 + kk] = (own && c0 + kk > lig) ? v : 0.0f;
      }
    }
    gsync();
  }
}
// x = (L L^T)^-1 g for the lane's component; broadcasts via v_readlane (no LDS traffic)
template <int NVR, int G>
DEV float chol_solve_rows(const float (&h)[NVR], const float (&lt)[NVR], float rdiag, float g) {
  const int lig = threadIdx.x & (G - 1);
  float acc = g, y = 0.0f, x = 0.0f;
#pragma unroll
  for (int k = 0; k < NVR; ++k) {
    const float yk = bcastg<G>(acc * rdiag, k);
    if (lig == k) y = yk;
    acc -= h[k] * yk;
  }
  acc = y;
#pragma unroll
  for (int k = NVR - 1; k >= 0; --k) {
    const float xk = bcastg<G>(acc * rdiag, k);
    if (lig == k) x = xk;
    acc -= lt[k] * xk;
  }
  return x;
}

// ---- the same solve by a BLOCKED right-looking Cholesky (round 2 for the MFMA Newton kernel, round 3 generic in the group size):
// factorisation, forward and backward substitution in one routine, nothing but the H row (NVR registers) held across it.
//   h: row i of the SPD matrix on entry (the FULL row: the trailing block is kept symmetric), destroyed.  Lanes >= nv hold identity
//   rows.  LDS, private to the lane group: panel (4 G floats), vec (G), save (12 NV4).
// Per 4-column block: every lane parks its four raw panel entries and its running right-hand side in LDS (one round trip), refactors
// the 4 x 4 diagonal block redundantly in registers (no cross-lane traffic), takes its own entries of L and applies the Schur update
// from the RAW panel (h[k] -= (x Ld^-1) . p_k = L_i . L_k); the forward substitution rides along.  The backward substitution needs
// columns of L, which live across lanes: per block four DPP reductions of h[c] * x, then the 4 x 4 back substitution in registers
// from the block factors saved in LDS.  NVR / 4 LDS round trips per factorisation where chol_factor_rows needs NVR, and NVR^2 / 8
// float4 panel reads where it needs NVR^2 / 2 scalar ones (the G1's 64-lane Newton kernel spent 65 % of its time there).
DEV float rsqrt_nr(float pv) {  // v_rsq_f32 + one Newton step (~0.5 ulp), as chol_factor_rows
  float inv = __builtin_amdgcn_rsqf(pv);
  return inv * (1.5f - 0.5f * pv * inv * inv);
}
// FACTOR = false (round 6, the Newton solver's factor reuse: solver.py:2670-2733 _update_gradient_cholesky_blocked_skip_unchanged): `h` already
// holds the rows of L and `save` the block factors of an earlier call -- only the two substitutions run (NV4 LDS round trips for the
// right-hand side, no panel, no Schur update).
template <int NV4, int G, bool FACTOR = true>
DEV float chol_factor_solve_g(float (&h)[4 * NV4], float g, float* panel, float* vec, float* save, int lig) {
  constexpr int NVR = 4 * NV4;
  float gacc = g, y = 0.0f;
  if constexpr (!FACTOR) {
#pragma unroll
    for (int jb = 0; jb < NV4; ++jb) {
      const int j0 = 4 * jb;
      gsync();
      vec[lig] = gacc;
      gsync();
      const float4 g4 = *reinterpret_cast<const float4*>(vec + j0);
      const float4 sa = *reinterpret_cast<const float4*>(save + 12 * jb);      // l10 l20 l30 r0
      const float4 sb = *reinterpret_cast<const float4*>(save + 12 * jb + 4);  // l21 l31 l32 r1
      const float2 sc = *reinterpret_cast<const float2*>(save + 12 * jb + 8);  // r2 r3
      const float y0 = g4.x * sa.w;
      const float y1 = (g4.y - sa.x * y0) * sb.w;
      const float y2 = (g4.z - sa.y * y0 - sb.x * y1) * sc.x;
      const float y3 = (g4.w - sa.z * y0 - sb.y * y1 - sb.z * y2) * sc.y;
      y = lig == j0 ? y0 : y;
      y = lig == j0 + 1 ? y1 : y;
      y = lig == j0 + 2 ? y2 : y;
      y = lig == j0 + 3 ? y3 : y;
      gacc -= h[j0] * y0 + h[j0 + 1] * y1 + h[j0 + 2] * y2 + h[j0 + 3] * y3;
    }
  }
#pragma unroll
  for (int jb = 0; jb < (FACTOR ? NV4 : 0); ++jb) {
    const int j0 = 4 * jb;
    gsync();  // the previous block's panel reads are issued before this write (one wavefront: LDS ops complete in order)
    // unconditional (lanes >= NVR park junk in their own slot): a branch here splits the block, the compiler then sinks the
    // previous block's Schur FMAs past it while their panel loads must stay before this write
    *reinterpret_cast<float4*>(panel + 4 * lig) = make_float4(h[j0], h[j0 + 1], h[j0 + 2], h[j0 + 3]);
    vec[lig] = gacc;
    gsync();
    const float4 d0 = *reinterpret_cast<const float4*>(panel + 4 * j0);
    const float4 d1 = *reinterpret_cast<const float4*>(panel + 4 * (j0 + 1));
    const float4 d2 = *reinterpret_cast<const float4*>(panel + 4 * (j0 + 2));
    const float4 d3 = *reinterpret_cast<const float4*>(panel + 4 * (j0 + 3));
    const float4 g4 = *reinterpret_cast<const float4*>(vec + j0);
    const float r0 = rsqrt_nr(fmaxf(d0.x, MJ_MINVAL));
    const float l10 = d1.x * r0, l20 = d2.x * r0, l30 = d3.x * r0;
    const float r1 = rsqrt_nr(fmaxf(d1.y - l10 * l10, MJ_MINVAL));
    const float l21 = (d2.y - l20 * l10) * r1, l31 = (d3.y - l30 * l10) * r1;
    const float r2 = rsqrt_nr(fmaxf(d2.z - l20 * l20 - l21 * l21, MJ_MINVAL));
    const float l32 = (d3.z - l30 * l20 - l31 * l21) * r2;
    const float r3 = rsqrt_nr(fmaxf(d3.w - l30 * l30 - l31 * l31 - l32 * l32, MJ_MINVAL));
    *reinterpret_cast<float4*>(save + 12 * jb) = make_float4(l10, l20, l30, r0);
    *reinterpret_cast<float4*>(save + 12 * jb + 4) = make_float4(l21, l31, l32, r1);
    *reinterpret_cast<float2*>(save + 12 * jb + 8) = make_float2(r2, r3);
    const float x0 = h[j0] * r0;
    const float x1 = (h[j0 + 1] - x0 * l10) * r1;
    const float x2 = (h[j0 + 2] - x0 * l20 - x1 * l21) * r2;
    const float x3 = (h[j0 + 3] - x0 * l30 - x1 * l31 - x2 * l32) * r3;
    h[j0] = x0;
    h[j0 + 1] = x1;
    h[j0 + 2] = x2;
    h[j0 + 3] = x3;
    const float y0 = g4.x * r0;
    const float y1 = (g4.y - l10 * y0) * r1;
    const float y2 = (g4.z - l20 * y0 - l21 * y1) * r2;
    const float y3 = (g4.w - l30 * y0 - l31 * y1 - l32 * y2) * r3;
    y = lig == j0 ? y0 : y;  // (four selects: a nested conditional becomes branches)
    y = lig == j0 + 1 ? y1 : y;
    y = lig == j0 + 2 ? y2 : y;
    y = lig == j0 + 3 ? y3 : y;
    gacc -= x0 * y0 + x1 * y1 + x2 * y2 + x3 * y3;
    if (jb + 1 < NV4) {
      const float u3 = x3 * r3;
      const float u2 = (x2 - l32 * u3) * r2;
      const float u1 = (x1 - l21 * u2 - l31 * u3) * r1;
      const float u0 = (x0 - l10 * u1 - l20 * u2 - l30 * u3) * r0;
#pragma unroll
      for (int k = j0 + 4; k < NVR; ++k) {
        const float4 pk = *reinterpret_cast<const float4*>(panel + 4 * k);
        h[k] = fmaf(-u3, pk.w, fmaf(-u2, pk.z, fmaf(-u1, pk.y, fmaf(-u0, pk.x, h[k]))));
      }
      __builtin_amdgcn_sched_barrier(0);  // finish the update here: deferring it keeps the panel rows in registers
    }
  }
  gsync();
  vec[lig] = y;
  gsync();
  float x = 0.0f;  // lanes >= NVR never receive a value: their (junk but finite) h entries are multiplied by zero
#pragma unroll
  for (int jb = NV4 - 1; jb >= 0; --jb) {
    const int j0 = 4 * jb;
    float t4[4] = {h[j0] * x, h[j0 + 1] * x, h[j0 + 2] * x, h[j0 + 3] * x};
    gsumg_n<G, 4>(t4);
    const float4 y4 = *reinterpret_cast<const float4*>(vec + j0);
    const float4 sa = *reinterpret_cast<const float4*>(save + 12 * jb);      // l10 l20 l30 r0
    const float4 sb = *reinterpret_cast<const float4*>(save + 12 * jb + 4);  // l21 l31 l32 r1
    const float2 sc = *reinterpret_cast<const float2*>(save + 12 * jb + 8);  // r2 r3
    const float x3 = (y4.w - t4[3]) * sc.y;
    const float x2 = ((y4.z - t4[2]) - sb.z * x3) * sc.x;
    const float x1 = ((y4.y - t4[1]) - sb.x * x2 - sb.y * x3) * sb.w;
    const float x0 = ((y4.x - t4[0]) - sa.x * x1 - sa.y * x2 - sa.z * x3) * sa.w;
    x = lig == j0 ? x0 : x;
    x = lig == j0 + 1 ? x1 : x;
    x = lig == j0 + 2 ? x2 : x;
    x = lig == j0 + 3 ? x3 : x;
  }
  return x;
}

// implicitfast fused into the solver's epilogue (round 3; forward.py:578-612, derivative.py:1117): x = (M + h D - h dA/dv)^-1 (M qacc) from
// the register-resident row of M, for models without activations.  The integrator launch did this with the sparse L'DL factor of the
// modified M -- a serial chain over the dofs -- and was 23 % of the Panda's step; here it is one blocked Cholesky of the dense row the
// solver already holds.  `scratch`: 6 G + 12 NV4 floats of group-private LDS that nothing else uses any more.
template <int NV4, int G>
DEV float impfast_acc(const MjhModel& m, const MjhData& d, int w, int lig, bool active, const float (&mrow)[4 * NV4], float Ma, float* scratch) {
  constexpr int NVR = 4 * NV4;
  float *diag = scratch, *panel = scratch + G, *vec = panel + 4 * G, *save = vec + G;
  const float h = bf(m.opt_timestep, m.opt_timestep_nb, w, 1)[0];
  diag[lig] = 0.0f;
  gsync();
  if (!(m.disableflags & DSBL_ACTUATION)) actuator_vel_diag<G>(m, d, w, lig, h, diag);
  gsync();
  float dg = 0.0f;
  if (active) {
    dg = diag[lig];
    if (!(m.disableflags & DSBL_DAMPER)) dg += h * bf(m.dof_damping, m.dof_damping_nb, w, m.nv)[lig];
  }
  float hh[NVR];
#pragma unroll
  for (int c = 0; c < NVR; ++c) hh[c] = mrow[c] + (c == lig ? dg : 0.0f);
  const float x = chol_factor_solve_g<NV4, G>(hh, active ? Ma : 0.0f, panel, vec, save, lig);
  return active ? x : 0.0f;
}

struct SolveLayout {
  int J, force, da, bsearch, bgrad, col, ex, cone, fl, total;
};
template <int NV4, int NR, int G, bool NEWTON, bool ELL = false, bool TREE = false>
__host__ __device__ inline SolveLayout solve_layout(int njmax) {
  constexpr int NVR = 4 * NV4;
  constexpr int JS = (NV4 & 1) ? NVR : NVR + 4;  // JS/4 odd: row-per-lane 16-byte reads hit distinct banks
  // J^T f runs over 16-row chunks (rows past nefc are zero); a kernel never holds more rows than its lanes cover
  const int njp = min(((njmax + 15) / 16) * 16, G * NR);
  SolveLayout p;
  int o = 0;
  {  // also stages the dense NVR x NVR copy of M, and lends 6 G + 12 NV4 words to the fused implicitfast update
    const int jw = (njp > NVR ? njp : NVR) * JS;
    p.J = o; o += jw > 6 * G + 12 * NV4 ? jw : 6 * G + 12 * NV4;
  }
  p.force = o; o += G * NR;                     // efc_force of the current iterate (for J^T f)
  p.da = o; o += NEWTON ? G * NR : 0;           // D * [state == QUADRATIC] (Newton: J^T D J)
  p.bsearch = o; o += G;                        // broadcast copies of the two nv-vectors other lanes read
  p.bgrad = o; o += G;
  // Newton: the blocked Cholesky's panel, right-hand side and block factors; CG: the double-buffered Gauss-Jordan pivot row only
  p.col = o; o += NEWTON ? 5 * G + 12 * NV4 : 2 * (NVR > G ? NVR : G);  // (Newton: panel 4 G | vec G | save 12 NV4)
  // elliptic cones: per-row exchange lines (scaled Jaref / jv, the three quadratic-cost terms, the row's friction scale) through
  // which the rows of one contact see each other; Newton adds the cone Hessian block rows (6 words) + first row / size
  p.ex = o; o += ELL ? 6 * G * NR : 0;
  p.cone = o; o += (ELL && NEWTON) ? 7 * G * NR : 0;
  p.fl = o; o += TREE ? G * NR : 0;  // per-tree solve: frictionloss of this tree's rows, gathered through the row map
  p.total = ((o + 3) / 4) * 4;
  return p;
}

// force and state of one row at Jaref = ja (solver.py:1698-1822): equality rows are always active, limit/contact rows
// when violated, padding rows (D = 0) never -- branch-free; friction-loss rows (rare) have three zones
DEV void row_force(int kind, float ja, float D, bool has_fl, const float* floss, float& force, int& state) {
  const bool quad = kind == 0 || (kind == 2 && ja < 0.0f);
  force = quad ? -D * ja : 0.0f;
  state = quad ? ST_QUADRATIC : ST_SATISFIED;
  if (has_fl && kind == 1) {
    const float f = *floss, rf = safe_div(f, D);
    if (ja <= -rf) { force = f; state = ST_LINEARNEG; }
    else if (ja >= rf) { force = -f; state = ST_LINEARPOS; }
    else { force = -D * ja; state = ST_QUADRATIC; }
  }
}

// Explicit Euler step fused into the solver's epilogue (forward.py:387-417 without the implicit-damping branch, _advance
// 276-349 for models without activations): lane i holds qacc[i]; `vbuf` is a group-private LDS line of >= nv floats.
// Saves the integrator launch (~11 us per step) when the host knows the velocity update is explicit.
// qacc_i: the acceleration the velocity update uses (the solver's qacc for Euler, the implicitfast solution otherwise); warm_i: the
// solver's qacc, which becomes the next step's warm start either way.
template <int G>
DEV void euler_advance(const MjhModel& m, const MjhData& d, int w, int lig, bool active, float qacc_i, float* vbuf, float warm_i) {
  const int nv = m.nv;
  const float h = bf(m.opt_timestep, m.opt_timestep_nb, w, 1)[0];
  const size_t vo = (size_t)w * nv;
  if (active) {
    const float v = d.qvel[vo + lig] + qacc_i * h;
    vbuf[lig] = v;
    d.qvel[vo + lig] = v;
    d.qacc_warmstart[vo + lig] = warm_i;
  }
  gsync();
  float* qpos = d.qpos + (size_t)w * m.nq;
  for (int j = lig; j < m.njnt; j += G) {  // _next_position forward.py:53
    const int qa = m.jnt_qposadr[j], dof = m.jnt_dofadr[j], t = m.jnt_type[j];
    if (t == JNT_FREE) {
      for (int k = 0; k < 3; ++k) qpos[qa + k] += h * vbuf[dof + k];
      st4(qpos + qa + 3, quat_integrate(ld4(qpos + qa + 3), ld3(vbuf + dof + 3), h));
    } else if (t == JNT_BALL) {
      st4(qpos + qa, quat_integrate(ld4(qpos + qa), ld3(vbuf + dof), h));
    } else {
      qpos[qa] += h * vbuf[dof];
    }
  }
  if (lig == 0) d.time[w] += h;
}

// (cost - cost(0), grad, hess) of ONE constraint row on the ray at step alpha
// (solver.py:518-556 _compute_efc_eval_pt_pyramidal; alpha = 0 variant 620-647)
struct P3 {
  float c, g, h;
};
DEV P3 eval_row(float ja, float jv, float D, float f, int kind, float a) {
  const float jvD = jv * D, hess = jv * jvD, grad0 = jvD * ja;
  const float x = ja + a * jv;
  P3 r = P3{0.0f, 0.0f, 0.0f};
  if (kind == 2) {  // limit / contact: active only when x < 0
    const float quad0 = 0.5f * D * ja * ja;
    const float cost0 = ja < 0.0f ? quad0 : 0.0f;
    if (x < 0.0f) {
      r.c = a * (grad0 + 0.5f * a * hess) + (quad0 - cost0);
      r.g = grad0 + a * hess;
      r.h = hess;
    } else {
      r.c = -cost0;
    }
  } else if (kind == 1) {  // friction loss
    const float rf = safe_div(f, D);
    const float cost0 = (-rf < ja && ja < rf) ? 0.5f * D * ja * ja : (ja <= -rf ? f * (-0.5f * rf - ja) : f * (-0.5f * rf + ja));
    if (-rf < x && x < rf) {
      r.c = 0.5f * D * x * x - cost0;
      r.g = jvD * x;
      r.h = hess;
    } else if (x <= -rf) {
      r.c = f * (-0.5f * rf - x) - cost0;
      r.g = -f * jv;
    } else {
      r.c = f * (-0.5f * rf + x) - cost0;
      r.g = f * jv;
    }
  } else if (kind == 0) {  // equality
    r.c = a * (grad0 + 0.5f * a * hess);
    r.g = grad0 + a * hess;
    r.h = hess;
  }
  return r;
}
DEV bool in_bracket(P3 x, P3 y) { return (x.g < y.g && y.g < 0.0f) || (x.g > y.g && y.g > 0.0f); }

#ifndef MJH_LS_NOISE_ULPS
#define MJH_LS_NOISE_ULPS 4
#endif
// ---- the line search (solver.py:835-1347) over rows held in registers; every sum of an evaluation round is reduced together
// HAS_FL: friction-loss rows present (three-zone cost, rare): a compile-time switch, the common instantiation is branch-free
template <int NR, int G, bool HAS_FL, int RED = 0>
DEV void line_search_rows(const float (&rja)[NR], const float (&rjv)[NR], const float (&rD)[NR], const int (&rkind)[NR],
                          const float* floss_lane, float gauss1_lane, float gauss2_lane, float gauss1_abs_lane, float gtol_in, int ls_iterations,
                          float& alpha_out, float& improvement_out, bool& converged_out, int* iters_out = nullptr) {
  float ehess[NR], egrad0[NR], ecact[NR], ecin[NR];
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const float jvD = rjv[k] * rD[k], quad0 = 0.5f * rD[k] * rja[k] * rja[k];
    const float cost0 = (rkind[k] == 0 || rja[k] < 0.0f) ? quad0 : 0.0f;
    ehess[k] = rjv[k] * jvD;
    egrad0[k] = jvD * rja[k];
    ecact[k] = quad0 - cost0;
    ecin[k] = -cost0;
  }
  auto eval = [&](float a) __attribute__((always_inline)) {
    P3 s = P3{0.0f, 0.0f, 0.0f};
    if (!HAS_FL) {
      const float ha = 0.5f * a;
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        const bool act = rkind[k] == 0 || (rja[k] + a * rjv[k] < 0.0f);
        s.c += act ? a * (egrad0[k] + ha * ehess[k]) + ecact[k] : ecin[k];
        s.g += act ? egrad0[k] + a * ehess[k] : 0.0f;
        s.h += act ? ehess[k] : 0.0f;
      }
    } else {
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        const P3 t = eval_row(rja[k], rjv[k], rD[k], rkind[k] == 1 ? floss_lane[G * k] : 0.0f, rkind[k], a);
        s.c += t.c;
        s.g += t.g;
        s.h += t.h;
      }
    }
    return s;
  };
  const P3 e = eval(0.0f);
#if MJH_LS_NOISE_ULPS > 0
  // float32: the ray derivative is a sum of ~nefc + nv terms that mostly cancel at the minimum, so it carries rounding noise of
  // about eps * sum |term|.  A derivative inside that noise is zero as far as float32 can tell: bracketing on (the reference's
  // tolerance floor of 1e-6 is an absolute number chosen for float64) only chases noise -- measured 2.5 bracketing iterations
  // per call against 0.75 for the float64 oracle, with no effect on the iterates.
  // (round 3: the three sums of the Gauss quadratic -- search . (Ma - qfrc_smooth), search . M search / 2 and the absolute terms of the
  // first -- ride in the same reduction: one dependent DPP chain per solver iteration less)
  float eabs = 0.0f;
#pragma unroll
  for (int k = 0; k < NR; ++k) eabs += fabsf(egrad0[k]);
  float r2[6] = {e.g, e.h, eabs, gauss1_lane, gauss2_lane, gauss1_abs_lane};
  gsum_sel<G, 6, RED>(r2);
  const float gauss1 = r2[3], gauss2 = r2[4];
  const float gtol = fmaxf(gtol_in, (MJH_LS_NOISE_ULPS * 5.96e-8f) * (r2[5] + r2[2]));
#else
  float r2[4] = {e.g, e.h, gauss1_lane, gauss2_lane};
  gsum_sel<G, 4, RED>(r2);
  const float gauss1 = r2[2], gauss2 = r2[3];
  const float gtol = gtol_in;
#endif
  // group sums + the Gauss (smooth) quadratic; the sums of up to three ray points are reduced together (gsumg_n)
  auto finish = [&](float c, float g, float h, float a) __attribute__((always_inline)) {
    return P3{a * a * gauss2 + a * gauss1 + c, 2.0f * a * gauss2 + gauss1 + g, 2.0f * gauss2 + h};
  };
  const P3 p0 = P3{0.0f, gauss1 + r2[0], 2.0f * gauss2 + r2[1]};
  const float lo_alpha_in = -fast_div(p0.g, p0.h);
  const P3 el = eval(lo_alpha_in);
  float r3[3] = {el.c, el.g, el.h};
  gsum_sel<G, 3, RED>(r3);
  const P3 lo_in = finish(r3[0], r3[1], r3[2], lo_alpha_in);
  float alpha = 0.0f, improvement = 0.0f;
  bool ls_converged = fabsf(lo_in.g) < gtol && lo_in.c < 0.0f;
  if (ls_converged) {
    alpha = lo_alpha_in;
    improvement = -lo_in.c;
  } else {
    const bool lo_less = lo_in.g < p0.g;
    P3 lo = lo_less ? lo_in : p0, hi = lo_less ? p0 : lo_in;
    float lo_alpha = lo_less ? lo_alpha_in : 0.0f, hi_alpha = lo_less ? 0.0f : lo_alpha_in;
    for (int it = 0; it < ls_iterations; ++it) {
      if (iters_out) ++*iters_out;
      const float a_lo = lo_alpha - fast_div(lo.g, lo.h), a_hi = hi_alpha - fast_div(hi.g, hi.h);
      const float a_mid = 0.5f * (lo_alpha + hi_alpha);
      const P3 e1 = eval(a_lo), e2 = eval(a_hi), e3 = eval(a_mid);
      float r9[9] = {e1.c, e1.g, e1.h, e2.c, e2.g, e2.h, e3.c, e3.g, e3.h};
      gsum_sel<G, 9, RED>(r9);
      const P3 lo_next = finish(r9[0], r9[1], r9[2], a_lo), hi_next = finish(r9[3], r9[4], r9[5], a_hi), mid = finish(r9[6], r9[7], r9[8], a_mid);
      // Bracket update (solver.py:1222-1290).  The reference takes a candidate when its derivative lies strictly between the
      // bracket end's derivative and zero, for three candidates in turn -- each test on the end the previous one may have
      // replaced.  The end therefore finishes on the candidate whose derivative is closest to zero among those strictly
      // between the ORIGINAL end and zero (the earliest on ties), which needs no chain: three keys, one minimum, one select.
      auto pick = [](P3& end, float& end_a, const P3& y1, float a1, const P3& y2, float a2, const P3& y3, float a3) __attribute__((always_inline)) {
        const float g0 = end.g, m0 = fabsf(g0);
        auto key = [&](float g) __attribute__((always_inline)) {
          const float mg = fabsf(g);
          const bool same = (__float_as_int(g) ^ __float_as_int(g0)) >= 0;  // equal sign bits
          return (same && mg > 0.0f && mg < m0) ? mg : 3.0e38f;
        };
        const float k1 = key(y1.g), k2 = key(y2.g), k3 = key(y3.g);
        const float kb = fminf(k1, fminf(k2, k3));
        const bool any = kb < 3.0e38f;
        const bool u1 = k1 == kb, u2 = k2 == kb;
        const float sc = u1 ? y1.c : (u2 ? y2.c : y3.c), sg = u1 ? y1.g : (u2 ? y2.g : y3.g), sh = u1 ? y1.h : (u2 ? y2.h : y3.h);
        const float sa = u1 ? a1 : (u2 ? a2 : a3);
        end.c = any ? sc : end.c;
        end.g = any ? sg : end.g;
        end.h = any ? sh : end.h;
        end_a = any ? sa : end_a;
        return any;
      };
      const bool swap_lo = pick(lo, lo_alpha, lo_next, a_lo, mid, a_mid, hi_next, a_hi);
      const bool swap_hi = pick(hi, hi_alpha, hi_next, a_hi, mid, a_mid, lo_next, a_lo);
      const bool ls_done = (!swap_lo && !swap_hi) || (lo.c < 0.0f && lo.g < 0.0f && lo.g > -gtol) || (hi.c < 0.0f && hi.g > 0.0f && hi.g < gtol);
      const bool improved = lo.c < 0.0f || hi.c < 0.0f;
      const bool lo_better = lo.c < hi.c;
      alpha = improved ? (lo_better ? lo_alpha : hi_alpha) : alpha;
      improvement = improved ? -(lo_better ? lo.c : hi.c) : improvement;
      if (ls_done) {
        ls_converged = true;
        break;
      }
    }
  }
  alpha_out = alpha;
  improvement_out = improvement;
  converged_out = ls_converged;
}

// ---- elliptic friction cones (solver.py:272-421) -------------------------------------------------------------
// One contact = `dim` consecutive rows (normal, then friction directions).  In the reference's scaled coordinates
// N = mu * Jaref_0 and T = |(fri_j * Jaref_j)_j>=1| the contact is SATISFIED for N >= mu T (top zone), QUADRATIC for
// mu N + T <= 0 (bottom zone, every row an independent quadratic) and in the CONE (middle) zone otherwise, where its cost is
// dm/2 (N - mu T)^2, dm = D_0 / (mu^2 (1 + mu^2)), mu = friction_0 / sqrt(impratio).
DEV int ell_zone(float mu, float N, float T) {
  if (N >= mu * T || (T <= 0.0f && N >= 0.0f)) return ST_SATISFIED;
  if (mu * N + T <= 0.0f || (T <= 0.0f && N < 0.0f)) return ST_QUADRATIC;
  return ST_CONE;
}
// constants of one contact on the search ray, held by the lane that owns the contact's first row
struct EllRay {
  float mu, dm, q0, q1, q2, u0, v0, uu, uv, vv;  // q = sum_j (D ja^2 / 2, D jv ja, D jv^2 / 2); u = scaled Jaref, v = scaled jv
  float cost0, T0, r0;                            // the reference point alpha = 0 (_eval_elliptic_reference solver.py:275-298)
  int state0;
};
DEV void ell_ray_reference(EllRay& e) {
  e.T0 = 0.0f;
  e.r0 = 0.0f;
  if (e.uu <= 0.0f) {
    e.state0 = e.u0 < 0.0f ? ST_QUADRATIC : ST_SATISFIED;
    e.cost0 = e.u0 < 0.0f ? e.q0 : 0.0f;
    return;
  }
  e.T0 = sqrtf(e.uu);
  if (e.u0 >= e.mu * e.T0) {
    e.state0 = ST_SATISFIED;
    e.cost0 = 0.0f;
  } else if (e.mu * e.u0 + e.T0 <= 0.0f) {
    e.state0 = ST_QUADRATIC;
    e.cost0 = e.q0;
  } else {
    e.r0 = e.u0 - e.mu * e.T0;
    e.state0 = ST_CONE;
    e.cost0 = 0.5f * e.dm * e.r0 * e.r0;
  }
}
// (cost(alpha) - cost(0), d/dalpha, d2/dalpha2) of one elliptic contact: the shifted forms of the reference
// (_eval_elliptic_shifted solver.py:343-403), which difference against the reference point analytically -- in float32
// the plain difference of two costs loses the line search's whole signal near convergence
DEV P3 ell_eval(const EllRay& e, float a) {
  const float N = e.u0 + a * e.v0;
  const float Td = a * (2.0f * e.uv + a * e.vv), Tsqr = e.uu + Td;
  const float aq2 = a * e.q2;
  bool quadz = false, conez = false;
  float T = 0.0f;
  if (Tsqr <= 0.0f) {
    quadz = N < 0.0f;
  } else {
    T = sqrtf(Tsqr);
    if (N >= e.mu * T) {
    } else if (e.mu * N + T <= 0.0f) quadz = true;
    else conez = true;
  }
  if (quadz) {  // _eval_elliptic_quadratic_shifted solver.py:318-340
    float cost = a * (aq2 + e.q1);
    if (e.state0 == ST_CONE) {
      const float b0 = e.mu * e.u0 + e.T0;
      cost += 0.5f * e.dm * b0 * b0;
    } else if (e.state0 == ST_SATISFIED) {
      cost = 0.5f * e.dm * (1.0f + e.mu * e.mu) * (N * N + fmaxf(Tsqr, 0.0f));
    }
    return P3{cost, 2.0f * aq2 + e.q1, 2.0f * e.q2};
  }
  if (conez) {
    const float Tinv = 1.0f / T;
    const float T1 = (e.uv + a * e.vv) * Tinv, T2 = (e.vv - T1 * T1) * Tinv;
    const float r = N - e.mu * T, r1 = e.v0 - e.mu * T1;
    float cost;
    if (e.state0 == ST_CONE) {  // rationalised T - T0
      const float Tdelta = Td / (T + e.T0), rdelta = a * e.v0 - e.mu * Tdelta;
      cost = 0.5f * e.dm * rdelta * (2.0f * e.r0 + rdelta);
    } else if (e.state0 == ST_QUADRATIC) {
      const float b = e.mu * N + T;
      cost = a * (aq2 + e.q1) - 0.5f * e.dm * b * b;
    } else {
      cost = 0.5f * e.dm * r * r;
    }
    return P3{cost, e.dm * r * r1, e.dm * (r1 * r1 - e.mu * r * T2)};
  }
  return P3{-e.cost0, 0.0f, 0.0f};
}

// Rows of M^-1 by Gauss-Jordan with lane i owning row i of [A | B] (A = M, B = I); no pivoting (M is SPD).
// At step k lane k publishes the entries the other rows need -- B[k][0..k] and A[k][k+1..] -- as ONE LDS line
// (double buffered), the pivot A[k][k] travels by v_readlane; every lane then applies one rank-1 update.
// The scale step of row k is folded into the same update with the multiplier 1 - 1/pivot.
template <int NVR, int G>
DEV void invert_rows(const float (&mrow)[NVR], float (&b)[NVR], float* buf, int lig) {
  float a[NVR];
#pragma unroll
  for (int c = 0; c < NVR; ++c) {
    a[c] = mrow[c];
    b[c] = (c == lig) ? 1.0f : 0.0f;
  }
#pragma unroll
  for (int k = 0; k < NVR; ++k) {
    float* pb = buf + (k & 1) * G;
    if (lig == k) {
#pragma unroll
      for (int c4 = 0; c4 < NVR / 4; ++c4) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (4 * c4 + e <= k) ? b[4 * c4 + e] : a[4 * c4 + e];
        *reinterpret_cast<float4*>(pb + 4 * c4) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
    gsync();
    const float ipk = 1.0f / bcastg<G>(a[k], k);
    const float f = (lig == k) ? (1.0f - ipk) : a[k] * ipk;
#pragma unroll
    for (int c4 = 0; c4 < NVR / 4; ++c4) {
      const float4 v4 = *reinterpret_cast<const float4*>(pb + 4 * c4);
      const float pr[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = 4 * c4 + e;
        if (c <= k) b[c] -= f * pr[e];
        else a[c] -= f * pr[e];
      }
    }
  }
}

// The same inverse, four pivots per step (round 3): the lanes of the pivot block publish their rows as one 4 x NVR LDS tile, every
// lane inverts the 4 x 4 pivot block P redundantly in registers and applies one rank-4 update -- NVR / 4 LDS round trips and
// divisions chains instead of NVR (the single-pivot version spent 30 k cycles per world, 1 k per step, on the dependent
// write -> read -> divide -> update chain).  Compact storage as above: column c holds the inverse-in-progress B once its block
// was eliminated, A before.  For rows outside the block F = A_iK P^-1 and row_i -= F row_K; a row of the block becomes
// P^-1 row_K, written as the same update with F = e_i - (P^-1)_i; the block's own columns end as delta_ic - F_c.
// `buf` holds 2 x 4 x NVR floats (double buffered).
template <int NVR, int G>
DEV void invert_rows_b4(const float (&mrow)[NVR], float (&s)[NVR], float* buf, int lig) {
#pragma unroll
  for (int c = 0; c < NVR; ++c) s[c] = mrow[c];
#pragma unroll
  for (int kb = 0; kb < NVR / 4; ++kb) {
    constexpr int T = 4 * NVR;
    const int k = 4 * kb;
    float* pb = buf + (kb & 1) * T;
    const int q = lig - k;  // 0..3 for the lanes of the pivot block
    if (q >= 0 && q < 4) {
#pragma unroll
      for (int c4 = 0; c4 < NVR / 4; ++c4) *reinterpret_cast<float4*>(pb + q * NVR + 4 * c4) = make_float4(s[4 * c4], s[4 * c4 + 1], s[4 * c4 + 2], s[4 * c4 + 3]);
    }
    gsync();
    // P = rows K, columns K (symmetric positive definite); P^-1 by unrolled Gauss-Jordan without pivoting
    float P[4][4], Pi[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const float4 v = *reinterpret_cast<const float4*>(pb + p * NVR + k);
      P[p][0] = v.x; P[p][1] = v.y; P[p][2] = v.z; P[p][3] = v.w;
#pragma unroll
      for (int e = 0; e < 4; ++e) Pi[p][e] = p == e ? 1.0f : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float ip = 1.0f / P[j][j];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        P[j][e] *= ip;
        Pi[j][e] *= ip;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (r != j) {
          const float f = P[r][j];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            P[r][e] -= f * P[j][e];
            Pi[r][e] -= f * Pi[j][e];
          }
        }
    }
    const bool inb = q >= 0 && q < 4;
    float F[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const float out = s[k] * Pi[0][p] + s[k + 1] * Pi[1][p] + s[k + 2] * Pi[2][p] + s[k + 3] * Pi[3][p];
      const float own = (q == p ? 1.0f : 0.0f) - (q == 0 ? Pi[0][p] : (q == 1 ? Pi[1][p] : (q == 2 ? Pi[2][p] : Pi[3][p])));
      F[p] = inb ? own : out;
    }
#pragma unroll
    for (int c4 = 0; c4 < NVR / 4; ++c4) {
      if (c4 == kb) continue;
      const float4 r0 = *reinterpret_cast<const float4*>(pb + 4 * c4), r1 = *reinterpret_cast<const float4*>(pb + NVR + 4 * c4),
                   r2 = *reinterpret_cast<const float4*>(pb + 2 * NVR + 4 * c4), r3 = *reinterpret_cast<const float4*>(pb + 3 * NVR + 4 * c4);
      s[4 * c4] -= F[0] * r0.x + F[1] * r1.x + F[2] * r2.x + F[3] * r3.x;
      s[4 * c4 + 1] -= F[0] * r0.y + F[1] * r1.y + F[2] * r2.y + F[3] * r3.y;
      s[4 * c4 + 2] -= F[0] * r0.z + F[1] * r1.z + F[2] * r2.z + F[3] * r3.z;
      s[4 * c4 + 3] -= F[0] * r0.w + F[1] * r1.w + F[2] * r2.w + F[3] * r3.w;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) s[k + e] = (q == e ? 1.0f : 0.0f) - F[e];
  }
}

// TREE (nv > 64, MjhModel.tree_solve): the unit of work is one constraint island of one world -- the dofs of the kinematic trees
// that k_tree_rows found connected by coupling rows (M is block diagonal over trees, so the problem separates exactly over islands)
// and the rows grouped under that island; every global index goes through the island's dof map / row map, everything else is the
// same kernel.  An instantiation serves the islands whose dof count lies in (nv_lo, nv_hi] (32 lanes: <= 32 dofs, 64 lanes: 33..64).
template <int NV4, int NR, bool NEWTON, int G, bool ELL = false, bool TREE = false>
DEV void solve_body(const MjhModel& m, const MjhData& d, float* smem, const Blk& b, int nefc_lo = -1, int nefc_hi = 0x7fffffff,
                    int fuse_euler = 0, int nv_lo = 0, int nv_hi = 0x7fffffff, int w_direct = -1) {
  if ((int)threadIdx.x >= b.nthreads) return;
  constexpr int NVR = 4 * NV4;
  constexpr int JS = (NV4 & 1) ? NVR : NVR + 4;
  const int nv_all = m.nv, nC = m.nC, njmax = d.njmax, nvp = d.nv_pad;
  // (the J tile is sized for the rows THIS launch can meet: a row-class launch that stops at nefc_hi < njmax holds fewer rows -- more worlds per CU)
  const SolveLayout lay = solve_layout<NV4, NR, G, NEWTON, ELL, TREE>(TREE ? njmax : min(njmax, nefc_hi));
  const int lig = threadIdx.x & (G - 1), gib = threadIdx.x / G;
  const int slot = b.w0 + gib;
  if (TREE && gib >= b.nw) return;  // (a looped launch hands a block fewer islands than it has lane groups)
  if (w_direct < 0 && slot >= (TREE ? d.nworld * m.ntree : d.nworld)) return;
  // worlds are scheduled longest-expected-solve first and paired with a similar neighbour (k_schedule_worlds); w_direct: the caller names
  // the world (the fallback launch behind the pooled CG kernel, solve_tu.hpp)
  const int w = TREE ? slot / m.ntree : (w_direct >= 0 ? w_direct : d.ws_order[slot]);
  const int tree = TREE ? slot - w * m.ntree : 0;  // island index
  if (TREE && (!d.ws_separable[w] || tree >= d.ws_nisland[w])) return;  // (an island of more than 64 dofs: the generic solver)
  const int* dadr = TREE ? d.ws_isl_dofadr + (size_t)w * (m.ntree + 1) + tree : nullptr;
  const int nv = TREE ? dadr[1] - dadr[0] : nv_all;
  if (TREE && (nv <= nv_lo || nv > nv_hi)) return;
  const int* dmap = TREE ? d.ws_isl_dofmap + (size_t)w * nv_all + dadr[0] : nullptr;  // island dof -> dof
  const int* dinv = TREE ? d.ws_isl_dofinv + (size_t)w * nv_all : nullptr;            // dof -> island dof
  const int* rmap = TREE ? d.ws_tree_rowmap + (size_t)w * njmax + d.ws_tree_rowadr[(size_t)w * (m.ntree + 1) + tree] : nullptr;
  auto R = [&](int r) __attribute__((always_inline)) { return TREE ? rmap[r] : r; };  // global row of this problem's row r
  // LDS decides how many worlds a CU holds (the kernel runs 2-3 rounds): no block-shared tables here, the M-structure
  // is read once from global (L1 hits)
  float* S = smem + (size_t)gib * lay.total;
  float *Jl = S + lay.J, *eforce = S + lay.force, *eda = S + lay.da, *bsearch = S + lay.bsearch, *bgrad = S + lay.bgrad, *col = S + lay.col;
  constexpr int GR = G * NR;
  // elliptic: exchange lines (see solve_layout); exs = friction scale of each row, persistent over the solve
  float *exu = S + lay.ex, *exv = exu + GR, *exq0 = exv + GR, *exq1 = exq0 + GR, *exq2 = exq1 + GR, *exs = exq2 + GR;
  float* econe = S + lay.cone;                              // Newton: row r of the contact's cone Hessian block, 6 words
  int* einfo = reinterpret_cast<int*>(S + lay.cone) + 6 * GR;  // first row | rows << 8 of a contact in the cone zone, else -1

  // Two-size dispatch (njmax > 64): the same world list is offered to a small-row and a big-row instantiation; a world
  // is solved by the one whose range (nefc_lo, nefc_hi] holds its row count and skipped by the other.  LDS per world
  // follows the instantiation's row capacity, so the common few-row worlds run at several times the occupancy.
  const int nefc_all = TREE ? d.ws_tree_rowadr[(size_t)w * (m.ntree + 1) + tree + 1] - d.ws_tree_rowadr[(size_t)w * (m.ntree + 1) + tree]
                            : min(d.nefc[w], njmax);
  if (nefc_all <= nefc_lo || nefc_all > nefc_hi) return;
  const int nefc = min(nefc_all, G * NR);
  const int ne = TREE ? 0 : d.ne[w], nf = TREE ? 0 : d.nf[w];
  // vo + lig = this lane's entry of a world's nv-vector (TREE: through the island's dof map, hence a per-lane base)
  const size_t vo = TREE ? (size_t)w * nv_all + (lig < nv ? dmap[lig] : 0) - lig : (size_t)w * nv_all, eo = (size_t)w * njmax;
  // friction-loss rows present: the line search needs the three-zone cost (rare).  TREE: row kinds come from efc.type and the
  // frictionloss values are gathered into an LDS line (the row map breaks the lane-strided addressing of the plain path)
  float* flds = S + lay.fl;
  bool has_fl = nf > 0;
  if (TREE) {
    bool any = false;
    for (int r = lig; r < G * NR; r += G) {
      const bool fr = r < nefc && (d.efc_type[eo + rmap[r]] == CT_FRICTION_DOF || d.efc_type[eo + rmap[r]] == CT_FRICTION_TENDON);
      flds[r] = fr ? d.efc_frictionloss[eo + rmap[r]] : 0.0f;
      any |= fr;
    }
    has_fl = gballot<G>(any) != 0ull;
    gsync();
  }
  const float* floss = TREE ? flds : d.efc_frictionloss + eo;  // floss[r]: frictionloss of this problem's row r
  const bool active = lig < nv;
  const int ligr = lig < NVR ? lig : NVR - 1;  // clamped row index for lanes beyond the matrix

  PhaseClock pc(5, lig);
  // ---- M row of this lane into registers (dense staging in the J region) ----------------------------------
  float mrow[NVR];
  if (!TREE) {
    // every lane gathers its dense row through the model-wide address table M_dense: 2 x NV4 independent loads, all in flight
    // (the sparse -> dense staging through LDS that this replaces was a chain of dependent global loads: 9 k cycles per world)
    const float* Mg = d.M + (size_t)w * nC;
    const int nv4r = (nv_all + 3) >> 2;  // table row stride in 16-byte units (<= NV4: the 64-lane kernels round NV4 up)
    const int4* tab = reinterpret_cast<const int4*>(m.M_dense) + (size_t)(active ? lig : 0) * nv4r;
    int idx[NVR];
#pragma unroll
    for (int c4 = 0; c4 < NV4; ++c4) {
      const int4 t4 = c4 < nv4r ? tab[c4] : make_int4(-1, -1, -1, -1);
      idx[4 * c4] = t4.x; idx[4 * c4 + 1] = t4.y; idx[4 * c4 + 2] = t4.z; idx[4 * c4 + 3] = t4.w;
    }
#pragma unroll
    for (int c = 0; c < NVR; ++c) {
      const float v = Mg[idx[c] < 0 ? 0 : idx[c]];
      mrow[c] = active ? (idx[c] < 0 ? 0.0f : v) : (c == lig ? 1.0f : 0.0f);
    }
  } else {
    for (int idx = lig; idx < NVR * JS; idx += G) Jl[idx] = 0.0f;
    gsync();
    const float* Mg = d.M + (size_t)w * nC;
    for (int i = lig; i < nv; i += G) {
      const int gi = TREE ? dmap[i] : i;
      const int start = m.M_rowadr[gi], n = m.M_rownnz[gi];
      for (int a = 0; a < n; ++a) {
        const int j = TREE ? dinv[m.M_colind[start + a]] : m.M_colind[start + a];  // (ancestors of a dof lie in its own tree)
        const float v = Mg[start + a];
        Jl[i * JS + j] = v;
        Jl[j * JS + i] = v;
      }
    }
    gsync();
#pragma unroll
    for (int c4 = 0; c4 < NV4; ++c4) {
      const float4 v4 = *reinterpret_cast<const float4*>(Jl + ligr * JS + 4 * c4);
      mrow[4 * c4] = active ? v4.x : (4 * c4 == lig ? 1.0f : 0.0f);
      mrow[4 * c4 + 1] = active ? v4.y : (4 * c4 + 1 == lig ? 1.0f : 0.0f);
      mrow[4 * c4 + 2] = active ? v4.z : (4 * c4 + 2 == lig ? 1.0f : 0.0f);
      mrow[4 * c4 + 3] = active ? v4.w : (4 * c4 + 3 == lig ? 1.0f : 0.0f);
    }
    gsync();
  }

  pc.mark(0);
  // ---- lane-owned dof scalars (element `lig` of each nv-vector lives in a register) ------------------------
  const bool warm = !(m.disableflags & DSBL_WARMSTART);
  const float fs = active ? d.qfrc_smooth[vo + lig] : 0.0f;
  // lane i: sum_c M[i][c] vec[c], vec broadcast from an LDS line
  auto mul_row = [&](const float (&row)[NVR], const float* vec) __attribute__((always_inline)) {
    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
    for (int c4 = 0; c4 < NV4; ++c4) {
      const float4 v4 = *reinterpret_cast<const float4*>(vec + 4 * c4);
      s0 += row[4 * c4] * v4.x + row[4 * c4 + 2] * v4.z;
      s1 += row[4 * c4 + 1] * v4.y + row[4 * c4 + 3] * v4.w;
    }
    return active ? s0 + s1 : 0.0f;
  };
  // ---- qacc_smooth = M^-1 qfrc_smooth from the register-resident factor (the L'DL factor qLD is produced beside
  // the solver by k_factor_smooth).  CG: rows of M^-1, computed once per solve and reused as the preconditioner; it
  // also writes the public qacc_smooth.  Newton: only a cold start / an unconstrained world needs it (Cholesky of M).
  float h[NVR];  // Newton: H row, destroyed by the factorisation (rebuilt every iteration); CG: h = row of M^-1
  // Incremental Hessian + factor reuse (round 6; solver.py:3222-3280 _update_gradient_incremental, 2670-2733 skip_unchanged; pyramidal /
  // frictionless only, as the reference: _use_incremental 3509-3511).  One world per wavefront, widths whose registers hold a second copy of
  // the row: `hs` keeps H of the last build; an iteration adds (D_new - D_old) J_r^T J_r for the rows whose QUADRATIC state flipped --
  // typically a handful of the G1's ~70 rows -- and re-factorises; when no row flipped it keeps the factor and only substitutes.
  // 32 lanes per world / island (two per wavefront; the per-island kernels of models beyond 64 dofs: three_humanoids): the same, the flipped
  // rows of the two halves walked in lockstep, each half deciding for itself whether it re-factorises -- no world's arithmetic depends on
  // its wavefront partner.
  constexpr bool INC = NEWTON && !ELL && ((G == 64 && !TREE && NV4 <= 10) || (G == 32 && NV4 <= 8 && NR <= 2));
  float hs[INC ? NVR : 1];
  float pda[NR];  // D [state == QUADRATIC] of this lane's rows at the last build
#pragma unroll
  for (int k = 0; k < NR; ++k) pda[k] = 0.0f;
  float qs = 0.0f;
  if (!NEWTON) {
    // (the J region is free until the rows are loaded below: it lends the 2 x 4 x NVR-word tile buffer)
    if (NVR * JS >= 8 * NVR) invert_rows_b4<NVR, G>(mrow, h, Jl, lig);
    else invert_rows<NVR, G>(mrow, h, col, lig);
    bgrad[lig] = fs;
    gsync();
    qs = mul_row(h, bgrad);
    // one step of iterative refinement with the exact M: the float32 explicit inverse alone carries cond(M) * eps of
    // relative error (found by the randomised-model tests), fine for a preconditioner but not for qacc_smooth
    bsearch[lig] = qs;
    gsync();
    const float res = fs - mul_row(mrow, bsearch);
    bgrad[lig] = active ? res : 0.0f;
    gsync();
    qs += mul_row(h, bgrad);
    if (active) d.qacc_smooth[vo + lig] = qs;
    gsync();
  } else if (nefc == 0 || !warm) {
#pragma unroll
    for (int c = 0; c < NVR; ++c) h[c] = mrow[c];
    qs = chol_factor_solve_g<NV4, G>(h, fs, col, col + 4 * G, col + 5 * G, lig);
    if (!active) qs = 0.0f;
  }
  float q = 0.0f;
  if (active) q = nefc > 0 && warm ? d.qacc_warmstart[vo + lig] : qs;
  pc.mark(1);
  bsearch[lig] = q;
  gsync();
  float Ma = mul_row(mrow, bsearch);

  if (nefc == 0) {  // unconstrained: qacc = qacc_smooth (solver.py:3684-3686)
    if (active) {
      d.qacc[vo + lig] = q;
      d.qfrc_constraint[vo + lig] = 0.0f;
      d.efc_Ma[vo + lig] = Ma;
    }
    if (lig == 0 && !TREE) d.solver_niter[w] = 0;  // (TREE: k_tree_rows zeroed it; trees report with atomicMax)
    if (fuse_euler && !TREE) {
      float qi = q;
      if (fuse_euler == 2) {
        gsync();
        qi = impfast_acc<NV4, G>(m, d, w, lig, active, mrow, Ma, Jl);
      }
      euler_advance<G>(m, d, w, lig, active, qi, bsearch, q);
    }
    return;
  }


  // ---- J into LDS; this lane's rows (D, aref, frictionloss, Jaref) into registers ---------------------------
  const int nefc4 = (nefc + 3) & ~3;
  {
    const float* Jg = d.efc_J + (size_t)w * d.njmax_pad * nvp;
    if (TREE) {  // this tree's rows and columns
      for (int r = 0; r < nefc; ++r)
        for (int c = lig; c < JS; c += G) Jl[r * JS + c] = c < nv ? Jg[(size_t)rmap[r] * nvp + dmap[c]] : 0.0f;
    } else if (nvp == JS) {
      // same row stride in HBM and LDS: one flat copy with 16-byte loads, all in flight (a world's J block is 16-byte
      // aligned: njmax_pad * nv_pad is a multiple of 4)
      const float4* src = reinterpret_cast<const float4*>(Jg);
      float4* dst = reinterpret_cast<float4*>(Jl);
      const int n4 = nefc * (JS / 4);
#pragma unroll 4
      for (int i = lig; i < n4; i += G) dst[i] = src[i];
    } else {
      for (int r = 0; r < nefc; ++r)
        for (int c = lig; c < JS; c += G) Jl[r * JS + c] = c < nvp ? Jg[(size_t)r * nvp + c] : 0.0f;
    }
    for (int r = nefc; r < ((nefc + 15) & ~15); ++r)  // zero rows up to the next 16-row chunk boundary
      for (int c = lig; c < JS; c += G) Jl[r * JS + c] = 0.0f;
  }
  // persistent per-row registers are kept to the minimum (the kernel sits at the 3-waves-per-SIMD VGPR boundary):
  // frictionloss is re-read in the rare friction-loss path, force/state are recomputed once at the end
  float rD[NR], rja[NR], rjv[NR];
  int rkind[NR];
  // elliptic rows (kind 4: first row of a contact, 5: its other rows): friction scale of the row (mu for the first), the
  // contact's mu and dm, first row | rows << 8
  float rs[NR], rmu[NR], rdm[NR];
  int rcon[NR];
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const int r = lig + G * k;
    const bool has = r < nefc;
    rD[k] = has ? d.efc_D[eo + R(r)] : 0.0f;
    rkind[k] = !has ? 3 : (r >= ne + nf ? 2 : (r >= ne ? 1 : 0));  // 3: padding row (contributes nothing)
    if (TREE && has) {
      const int ty = d.efc_type[eo + rmap[r]];
      rkind[k] = ty == CT_EQUALITY ? 0 : ((ty == CT_FRICTION_DOF || ty == CT_FRICTION_TENDON) ? 1 : 2);
    }
    rjv[k] = 0.0f;
    eforce[r] = 0.0f;
    if (NEWTON) eda[r] = 0.0f;  // CG has no eda region
    if (ELL) {
      rs[k] = rmu[k] = rdm[k] = 0.0f;
      rcon[k] = 0;
      // (TREE: the rows are the island's, in island order -- a contact's rows stay consecutive --, and reach the world's tables through R())
      if (has && (TREE ? d.efc_type[eo + R(r)] == CT_CONTACT_ELLIPTIC : r >= ne + nf + d.nl[w])) {
        const int cid = d.ws_efc_con[eo + R(r)], c = cid >> 4, dimid = cid & 15;
        const float* cr = d.ws_contact + ((size_t)w * d.concap + c) * CON_STRIDE;
        const int* cri = reinterpret_cast<const int*>(cr);
        if (cri[24] > 1) {
          const int r0 = r - dimid, dim = min(cri[29], nefc - r0);
          const float impr2 = bf(m.opt_impratio_invsqrt, m.opt_impratio_invsqrt_nb, w, 1)[0];
          const float mu = cr[14] * impr2;
          rkind[k] = dimid == 0 ? 4 : 5;
          rmu[k] = mu;
          rs[k] = dimid == 0 ? mu : cr[CON_FRICTION_WORD(dimid - 1)];
          rdm[k] = safe_div(d.efc_D[eo + R(r0)], mu * mu * (1.0f + mu * mu));
          rcon[k] = r0 | (dim << 8);
        }
      }
      exs[r] = rs[k];
      exu[r] = 0.0f;
    }
  }
  gsync();
  // J[r,:] . vec  (row-per-lane, conflict-free 16-byte reads)
  auto j_dot = [&](const float* vec, int r) __attribute__((always_inline)) {
    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
    for (int c4 = 0; c4 < NV4; ++c4) {
      const float4 j4 = *reinterpret_cast<const float4*>(Jl + r * JS + 4 * c4);
      const float4 v4 = *reinterpret_cast<const float4*>(vec + 4 * c4);
      s0 += j4.x * v4.x + j4.z * v4.z;
      s1 += j4.y * v4.y + j4.w * v4.w;
    }
    return s0 + s1;
  };
#pragma unroll
  for (int k = 0; k < NR; ++k) rja[k] = rkind[k] != 3 ? j_dot(bsearch, lig + G * k) - d.efc_aref[eo + R(lig + G * k)] : 0.0f;
  gsync();

  pc.mark(2);
  const float tolerance = bf(m.opt_tolerance, m.opt_tolerance_nb, w, 1)[0];
  const float ls_tolerance = bf(m.opt_ls_tolerance, m.opt_ls_tolerance_nb, w, 1)[0];
  const float meaninertia = bf(m.stat_meaninertia, m.stat_meaninertia_nb, w, 1)[0];
  const float scale = meaninertia * (float)nv_all;
  const float rscale = 1.0f / scale;

  float grad_dot = 0.0f, search_dot = 0.0f, decrement = 0.0f;
  float g = 0.0f, Mg = 0.0f, pg = 0.0f, pMg = 0.0f, srch = 0.0f, qc = 0.0f;
  float cg5[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};

  // force and state of one row of an elliptic contact (_eval_constraint solver.py:455-472, _eval_elliptic_middle 406-421): the
  // rows of a contact decide together from the scaled Jaref the lanes just published in exu.  Newton also gets the row of the
  // contact's cone Hessian block C (H += J_c^T C J_c, _update_gradient_JTCJ_dense solver.py:2466-2564):
  //   C_ab = dm s_a s_b [ d_a0 d_b0 - (mu/T)(d_a0 u_b + u_a d_b0) + (mu N / T^3) u_a u_b + (mu^2 - N mu / T) d_ab [a >= 1] ],
  // u_0 := 0, s = friction scales (s_0 = mu)
  auto ell_row_force = [&](int k, float& force, int& state, bool& cone) __attribute__((always_inline)) {
    const int r = lig + G * k, r0 = rcon[k] & 255, dim = rcon[k] >> 8, a = r - r0;
    const float mu = rmu[k], dm = rdm[k];
    float u[6];
    float tt = 0.0f;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      u[j] = j < dim ? exu[r0 + j] : 0.0f;
      if (j > 0) tt += u[j] * u[j];
    }
    const float N = u[0], T = tt <= 0.0f ? 0.0f : sqrtf(tt);
    state = ell_zone(mu, N, T);
    cone = state == ST_CONE;
    if (state == ST_SATISFIED) force = 0.0f;
    else if (state == ST_QUADRATIC) force = -rD[k] * rja[k];
    else {
      const float fn = -dm * (N - mu * T) * mu;
      force = a == 0 ? fn : -safe_div(fn, T) * (exu[r] * rs[k]);
      if (NEWTON) {
        const float t = fmaxf(T, MJ_MINVAL), ttt = fmaxf(t * t * t, MJ_MINVAL);
        const float mu_tinv = safe_div(mu, t), mu_n_ttt = mu * safe_div(N, ttt), tdiag = mu * mu - N * mu_tinv;
        const float ua = a == 0 ? 0.0f : exu[r];
#pragma unroll
        for (int bq = 0; bq < 6; ++bq) {
          const float ub = bq == 0 ? 0.0f : u[bq];
          float cab = mu_n_ttt * ua * ub;
          if (a == 0 && bq == 0) cab += 1.0f;
          if (a == 0) cab -= mu_tinv * ub;
          if (bq == 0) cab -= mu_tinv * ua;
          if (a == bq && a > 0) cab += tdiag;
          econe[6 * r + bq] = bq < dim ? dm * rs[k] * exs[r0 + (bq < dim ? bq : 0)] * cab : 0.0f;
        }
      }
    }
  };

  int niter = 0;
  const int maxiter = m.iterations, ls_iterations = m.ls_iterations;
  int ovf = 0;
  float improvement = 0.0f;
  // One loop body = [constraint update, gradient/search update, convergence test, line search + move], so that the
  // (large, fully unrolled) gradient code has a single call site.  Iteration 0 is init_context (solver.py:3622).
  for (;;) {
    // ---- force/state of this lane's rows (solver.py:1698-1822) ------------------------------------------------
    if (ELL) {
#pragma unroll
      for (int k = 0; k < NR; ++k) exu[lig + G * k] = rja[k] * rs[k];
      gsync();
    }
    bool chg[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) chg[k] = false;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const float ja = rja[k], D = rD[k];
      // equality rows are always active, limit/contact rows when violated, padding rows (D = 0) never: branch-free
      float force;
      int state;
      row_force(rkind[k], ja, D, has_fl, floss + lig + G * k, force, state);
      bool cone = false;
      if (ELL && rkind[k] >= 4) ell_row_force(k, force, state, cone);
      eforce[lig + G * k] = force;
      if (NEWTON) {
        const float da = state == ST_QUADRATIC ? D : 0.0f;
        // (INC, after the first build: the line carries the CHANGE of D [quadratic] since the last build)
        eda[lig + G * k] = (INC && niter > 0) ? da - pda[k] : da;
        if (INC) {
          if (niter > 0 && da != pda[k]) chg[k] = true;
          pda[k] = da;
        }
      }
      if (ELL && NEWTON) einfo[lig + G * k] = cone ? rcon[k] : -1;
    }
    gsync();
    // ---- qfrc_constraint = J^T force (solver.py:1912-1947): lane = dof, 4 rows per step ------------------------
    {
      float s0 = 0.0f, s1 = 0.0f;
      const float* Jc = Jl + ligr;
      // fully unrolled 16-row chunks: every LDS read has an immediate offset (no address arithmetic); rows between
      // nefc and the chunk boundary carry zero force and zero J
#pragma unroll
      for (int r0 = 0; r0 < G * NR; r0 += 16) {
        if (r0 < nefc) {
#pragma unroll
          for (int r = r0; r < r0 + 16; r += 4) {
            const float4 f4 = *reinterpret_cast<const float4*>(eforce + r);
            s0 += Jc[r * JS] * f4.x + Jc[(r + 2) * JS] * f4.z;
            s1 += Jc[(r + 1) * JS] * f4.y + Jc[(r + 3) * JS] * f4.w;
          }
        }
      }
      qc = active ? s0 + s1 : 0.0f;
    }
    // ---- gradient and search direction (solver.py:3061-3220) ---------------------------------------------------
    g = active ? (Ma - fs - qc) : 0.0f;
    if (NEWTON) grad_dot = gsumg<G>(g * g);  // (CG: reduced with the Polak-Ribiere sums below)
    // improvement / gradient tests need no search direction: a world that passes them skips the H rebuild + Cholesky
    const bool done_early = niter > 0 && ((improvement * rscale < tolerance) || (sqrtf(grad_dot) * rscale < tolerance));
    pc.mark(3);
    if (NEWTON && done_early) break;
    if (NEWTON) {
      // H row = M row + sum_r (D_r [state_r == QUADRATIC]) J[r][i] J[r][:]   (JTDAJ, solver.py:2365-2440)
      bool refactor = true;
      bool full = true;
      if constexpr (INC) {
        if (niter > 0) {
          // rows whose quadratic state flipped since the last build (one world per wavefront: the masks are wavefront-uniform)
          full = false;
          refactor = false;
#pragma unroll
          for (int k = 0; k < NR; ++k) {
            unsigned long long mk = gballot<G>(chg[k]);  // this lane group's flipped rows (G = 64: the wavefront's)
            refactor |= mk != 0ull;
            while (__ballot(mk != 0ull)) {
              const bool on = mk != 0ull;
              const int r = on ? __builtin_ctzll(mk) + G * k : 0;
              mk &= mk - 1ull;  // (0 stays 0)
              const float jd = on ? Jl[r * JS + ligr] * eda[r] : 0.0f;
#pragma unroll
              for (int c4 = 0; c4 < NV4; ++c4) {
                const float4 a4 = *reinterpret_cast<const float4*>(Jl + r * JS + 4 * c4);
                hs[4 * c4] += jd * a4.x;
                hs[4 * c4 + 1] += jd * a4.y;
                hs[4 * c4 + 2] += jd * a4.z;
                hs[4 * c4 + 3] += jd * a4.w;
              }
            }
          }
          if (refactor) {
#pragma unroll
            for (int c = 0; c < NVR; ++c) h[c] = hs[c];
          }
        }
      }
      if (full) {
#pragma unroll
        for (int c = 0; c < NVR; ++c) h[c] = mrow[c];
        for (int r = 0; r < nefc4; r += 2) {  // two rows per LDS round trip (padding rows have J = 0, D = 0)
          float jd0 = Jl[r * JS + ligr] * eda[r], jd1 = Jl[(r + 1) * JS + ligr] * eda[r + 1];
          if (ELL) {  // rows of a contact in the cone zone: (C J_c)[a][lane's dof] replaces D J[r][lane's dof]
            const int i0 = einfo[r], i1 = r + 1 < nefc ? einfo[r + 1] : -1;
            if (i0 >= 0) {
              jd0 = 0.0f;
              for (int bq = 0; bq < (i0 >> 8); ++bq) jd0 += econe[6 * r + bq] * Jl[((i0 & 255) + bq) * JS + ligr];
            }
            if (i1 >= 0) {
              jd1 = 0.0f;
              for (int bq = 0; bq < (i1 >> 8); ++bq) jd1 += econe[6 * (r + 1) + bq] * Jl[((i1 & 255) + bq) * JS + ligr];
            }
          }
#pragma unroll
          for (int c4 = 0; c4 < NV4; ++c4) {
            const float4 a4 = *reinterpret_cast<const float4*>(Jl + r * JS + 4 * c4);
            const float4 b4 = *reinterpret_cast<const float4*>(Jl + (r + 1) * JS + 4 * c4);
            h[4 * c4] += jd0 * a4.x + jd1 * b4.x;
            h[4 * c4 + 1] += jd0 * a4.y + jd1 * b4.y;
            h[4 * c4 + 2] += jd0 * a4.z + jd1 * b4.z;
            h[4 * c4 + 3] += jd0 * a4.w + jd1 * b4.w;
          }
        }
        if constexpr (INC) {
#pragma unroll
          for (int c = 0; c < NVR; ++c) hs[c] = h[c];
        }
      }
      if constexpr (INC) {
        // (G = 32: `refactor` is uniform over a lane group, not over the wavefront -- halves that disagree run both paths one after the other)
        if (refactor) Mg = chol_factor_solve_g<NV4, G>(h, g, col, col + 4 * G, col + 5 * G, lig);
        else Mg = chol_factor_solve_g<NV4, G, false>(h, g, col, col + 4 * G, col + 5 * G, lig);
      } else {
        Mg = chol_factor_solve_g<NV4, G>(h, g, col, col + 4 * G, col + 5 * G, lig);
      }
      if (!active) Mg = 0.0f;
      srch = -Mg;
      search_dot = gsumg<G>(Mg * Mg);
      decrement = gsumg<G>(g * Mg);
    } else {
      bgrad[lig] = g;
      gsync();
      Mg = mul_row(h, bgrad);  // Mgrad = M^-1 grad
      // one reduction for everything this iteration's direction needs (round 3; was three dependent DPP chains): |grad|^2, the
      // Polak-Ribiere numerator and denominator, and |search|^2 of the NEW direction -Mgrad + beta search expanded into
      // |Mgrad|^2 - 2 beta Mgrad . search + beta^2 |search|^2 (it only scales the line search's gradient tolerance)
      cg5[0] = g * g; cg5[1] = g * (Mg - pMg); cg5[2] = pg * pMg; cg5[3] = Mg * Mg; cg5[4] = Mg * srch;
      gsumg_n<G, 5>(cg5);
      grad_dot = cg5[0];
    }
    pc.mark(4);
    if (niter == 0) {
      if (!NEWTON) {  // CG: search = -Mgrad (solver.py:1663-1695)
        srch = -Mg;
        search_dot = cg5[3];
        pg = g;
        pMg = Mg;
      }
    } else {
      const float imp = improvement * rscale, gradient = sqrtf(grad_dot) * rscale;
      bool done;
      if (NEWTON) {
        done = (imp < tolerance) || (gradient < tolerance) || (0.5f * decrement * rscale < tolerance);
      } else {
        // Polak-Ribiere (solver.py:3283-3450)
        const float num = cg5[1], den = cg5[2];
        const float beta = fmaxf(0.0f, num * __builtin_amdgcn_rcpf(fmaxf(MJ_MINVAL, den)));
        done = (imp < tolerance) || (gradient < tolerance);
        if (!done) {
          srch = -Mg + beta * srch;
          search_dot = fmaxf(cg5[3] + beta * (beta * search_dot - 2.0f * cg5[4]), 0.0f);
          pg = g;
          pMg = Mg;
        }
      }
      if (done) break;
      if (niter >= maxiter) {
        ovf |= OVF_ITERATIONS;
        break;
      }
    }
    if (maxiter == 0) break;
    // ---- mv = M search, jv = J search --------------------------------------------------------------------
    bsearch[lig] = srch;
    gsync();
    const float mvi = mul_row(mrow, bsearch);
#pragma unroll
    for (int k = 0; k < NR; ++k) rjv[k] = rkind[k] != 3 ? j_dot(bsearch, lig + G * k) : 0.0f;
    pc.mark(5);
    // ---- line search (solver.py:835-1347); rows and all sums stay in registers ----------------------------------
    float gs[3] = {srch * (Ma - fs), 0.5f * srch * mvi, fabsf(srch * (Ma - fs))};
    const float gtol = fmaxf(tolerance * ls_tolerance * sqrtf(search_dot) * scale, 1e-6f);
    float alpha = 0.0f;
    improvement = 0.0f;
    bool ls_converged = false;
    if (!ELL) {
      const float* floss_lane = floss + lig;
      if (has_fl) line_search_rows<NR, G, true>(rja, rjv, rD, rkind, floss_lane, gs[0], gs[1], gs[2], gtol, ls_iterations, alpha, improvement, ls_converged);
      else line_search_rows<NR, G, false>(rja, rjv, rD, rkind, floss_lane, gs[0], gs[1], gs[2], gtol, ls_iterations, alpha, improvement, ls_converged);
    } else {
      gsumg_n<G, 3>(gs);
      const float gauss1 = gs[0], gauss2 = gs[1];
      // per-row constants of the ray (solver.py:518-556): cost(a) - cost(0) = a (grad0 + a hess / 2) + cact when the
      // row is active at a, cin otherwise; equality rows are always active, padding rows have D = jv = 0
      float ehess[NR], egrad0[NR], ecact[NR], ecin[NR];
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        const float jvD = rjv[k] * rD[k], quad0 = 0.5f * rD[k] * rja[k] * rja[k];
        const float cost0 = (rkind[k] == 0 || rja[k] < 0.0f) ? quad0 : 0.0f;
        ehess[k] = rjv[k] * jvD;
        egrad0[k] = jvD * rja[k];
        ecact[k] = quad0 - cost0;
        ecin[k] = -cost0;
        if (ELL) {  // every row of an elliptic contact publishes its terms; the contact is evaluated by its first row's lane
          const int r = lig + G * k;
          exu[r] = rja[k] * rs[k];
          exv[r] = rjv[k] * rs[k];
          exq0[r] = quad0;
          exq1[r] = egrad0[k];
          exq2[r] = 0.5f * ehess[k];
          if (rkind[k] >= 4) ehess[k] = egrad0[k] = ecact[k] = ecin[k] = 0.0f;
        }
      }
      EllRay ray[NR];
      if (ELL) {
        gsync();
#pragma unroll
        for (int k = 0; k < NR; ++k) {
          EllRay& e = ray[k];
          e.mu = rmu[k];
          e.dm = rdm[k];
          e.q0 = e.q1 = e.q2 = e.uu = e.uv = e.vv = 0.0f;
          const int r0 = lig + G * k, dim = rkind[k] == 4 ? rcon[k] >> 8 : 0;
          e.u0 = rkind[k] == 4 ? exu[r0] : 0.0f;
          e.v0 = rkind[k] == 4 ? exv[r0] : 0.0f;
#pragma unroll
          for (int j = 0; j < 6; ++j)
            if (j < dim) {
              e.q0 += exq0[r0 + j];
              e.q1 += exq1[r0 + j];
              e.q2 += exq2[r0 + j];
              if (j > 0) {
                const float uj = exu[r0 + j], vj = exv[r0 + j];
                e.uu += uj * uj;
                e.uv += uj * vj;
                e.vv += vj * vj;
              }
            }
          ell_ray_reference(e);
        }
      }
      auto eval = [&](float a) __attribute__((always_inline)) {
        P3 s = P3{0.0f, 0.0f, 0.0f};
        if (ELL) {
#pragma unroll
          for (int k = 0; k < NR; ++k)
            if (rkind[k] == 4) {
              const P3 t = ell_eval(ray[k], a);
              s.c += t.c;
              s.g += t.g;
              s.h += t.h;
            }
        }
        if (!has_fl) {  // equality / limit / contact rows only: branch-free
          const float ha = 0.5f * a;
#pragma unroll
          for (int k = 0; k < NR; ++k) {
            const bool act = rkind[k] == 0 || (rja[k] + a * rjv[k] < 0.0f);  // (elliptic rows: all four terms are zero)
            s.c += act ? a * (egrad0[k] + ha * ehess[k]) + ecact[k] : ecin[k];
            s.g += act ? egrad0[k] + a * ehess[k] : 0.0f;
            s.h += act ? ehess[k] : 0.0f;
          }
        } else {
#pragma unroll
          for (int k = 0; k < NR; ++k) {
            const P3 t = eval_row(rja[k], rjv[k], rD[k], rkind[k] == 1 ? floss[lig + G * k] : 0.0f, rkind[k], a);
            s.c += t.c;
            s.g += t.g;
            s.h += t.h;
          }
        }
        return s;
      };
      // group sums + the Gauss (smooth) quadratic
      auto total = [&](P3 s, float a) __attribute__((always_inline)) {
        return P3{a * a * gauss2 + a * gauss1 + gsumg<G>(s.c), 2.0f * a * gauss2 + gauss1 + gsumg<G>(s.g), 2.0f * gauss2 + gsumg<G>(s.h)};
      };
      const P3 e = eval(0.0f);
      const P3 p0 = P3{0.0f, gauss1 + gsumg<G>(e.g), 2.0f * gauss2 + gsumg<G>(e.h)};
      const float lo_alpha_in = -fast_div(p0.g, p0.h);
      const P3 lo_in = total(eval(lo_alpha_in), lo_alpha_in);
      ls_converged = fabsf(lo_in.g) < gtol && lo_in.c < 0.0f;
      if (ls_converged) {
        alpha = lo_alpha_in;
        improvement = -lo_in.c;
      } else {
        const bool lo_less = lo_in.g < p0.g;
        P3 lo = lo_less ? lo_in : p0, hi = lo_less ? p0 : lo_in;
        float lo_alpha = lo_less ? lo_alpha_in : 0.0f, hi_alpha = lo_less ? 0.0f : lo_alpha_in;
        for (int it = 0; it < ls_iterations; ++it) {
          const float a_lo = lo_alpha - fast_div(lo.g, lo.h), a_hi = hi_alpha - fast_div(hi.g, hi.h);
          const float a_mid = 0.5f * (lo_alpha + hi_alpha);
          const P3 lo_next = total(eval(a_lo), a_lo), hi_next = total(eval(a_hi), a_hi), mid = total(eval(a_mid), a_mid);
          // conditional moves, not branches: the six updates are wave-divergent between the two worlds of a wavefront
          auto take = [](bool c, P3& dst, float& da, const P3& src, float sa) __attribute__((always_inline)) {
            dst.c = c ? src.c : dst.c;
            dst.g = c ? src.g : dst.g;
            dst.h = c ? src.h : dst.h;
            da = c ? sa : da;
          };
          const bool s1 = in_bracket(lo, lo_next);
          take(s1, lo, lo_alpha, lo_next, a_lo);
          const bool s2 = in_bracket(lo, mid);
          take(s2, lo, lo_alpha, mid, a_mid);
          const bool s3 = in_bracket(lo, hi_next);
          take(s3, lo, lo_alpha, hi_next, a_hi);
          const bool h1 = in_bracket(hi, hi_next);
          take(h1, hi, hi_alpha, hi_next, a_hi);
          const bool h2 = in_bracket(hi, mid);
          take(h2, hi, hi_alpha, mid, a_mid);
          const bool h3 = in_bracket(hi, lo_next);
          take(h3, hi, hi_alpha, lo_next, a_lo);
          const bool swap_lo = s1 || s2 || s3, swap_hi = h1 || h2 || h3;
          const bool ls_done = (!swap_lo && !swap_hi) || (lo.c < 0.0f && lo.g < 0.0f && lo.g > -gtol) || (hi.c < 0.0f && hi.g > 0.0f && hi.g < gtol);
          const bool improved = lo.c < 0.0f || hi.c < 0.0f;
          const bool lo_better = lo.c < hi.c;
          alpha = improved ? (lo_better ? lo_alpha : hi_alpha) : alpha;
          improvement = improved ? -(lo_better ? lo.c : hi.c) : improvement;
          if (ls_done) {
            ls_converged = true;
            break;
          }
        }
      }
    }
    if (!ls_converged) ovf |= OVF_LS_ITERATIONS;
    pc.mark(6);
    // ---- move along the ray (registers only) ---------------------------------------------------------------
    q += alpha * srch;
    Ma += alpha * mvi;
#pragma unroll
    for (int k = 0; k < NR; ++k) rja[k] += alpha * rjv[k];
    ++niter;
    pc.mark(7);
  }
  pc.mark(8);

  // ---- outputs ---------------------------------------------------------------------------------------------
  if (active) {
    d.qacc[vo + lig] = q;
    d.qfrc_constraint[vo + lig] = qc;
    d.efc_Ma[vo + lig] = Ma;
  }
  if (ELL) {
#pragma unroll
    for (int k = 0; k < NR; ++k) exu[lig + G * k] = rja[k] * rs[k];
    gsync();
  }
#pragma unroll
  for (int k = 0; k < NR; ++k)
    if (rkind[k] != 3) {  // force/state at the final iterate: the same expression the last constraint update evaluated
      float force;
      int state;
      row_force(rkind[k], rja[k], rD[k], has_fl, floss + lig + G * k, force, state);
      if (ELL && rkind[k] >= 4) {
        bool cone;
        ell_row_force(k, force, state, cone);
      }
      d.efc_force[eo + R(lig + G * k)] = force;
      d.efc_state[eo + R(lig + G * k)] = state;
    }
  if (lig == 0) {
    if (TREE) atomicMax(d.solver_niter + w, niter);
    else d.solver_niter[w] = niter;
    if (ovf) atomicOr(d.overflow + w, ovf);
  }
  if (fuse_euler && !TREE) {
    gsync();
    float qi = q;
    if (fuse_euler == 2) qi = impfast_acc<NV4, G>(m, d, w, lig, active, mrow, Ma, Jl);
    euler_advance<G>(m, d, w, lig, active, qi, bsearch, q);
  }
  pc.mark(9);
}
