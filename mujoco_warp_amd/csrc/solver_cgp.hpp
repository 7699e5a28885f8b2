// solver_cgp.hpp -- CG for the headline class (nv <= 32, njmax <= 64, pyramidal cones, contacts of condim 1 or 3): contact-basis rows in one
// row pool per workgroup, no vector through LDS, every LDS phase of an iteration one round trip (round 5).
//
// Reference: the same functions as solver.hpp (solver.py:3283-3450 CG, 835-1347 line search, 1698-1822 constraint update, 1912-1947
// qfrc_constraint); the pyramid rows of a contact are constraint.py:3751-3879 (_efc_contact_jac_dense): J_n + mu J_t1, J_n - mu J_t1,
// J_n + mu J_t2, J_n - mu J_t2.
//
// Why a third CG mapping.  k_solve<cg> (solver.hpp) holds `64 x JS` words of J per world whatever its row count: 7.9 KB, ten two-world
// wavefronts per CU = 2.5 per SIMD (2.08 measured), and every wavefront is one latency chain at 29 % VALU issue
// (profiles/round4_pmc_cg_steady.json): per iteration eleven dependent LDS round trips (J^T f in 16-row chunks, two vector broadcasts
// through LDS lines, the line search's cross-lane sums).  Here:
//   * contact-basis rows: the four pyramid rows of a condim-3 contact span N = J_n, T1 = mu J_t1, T2 = mu J_t2.  LDS holds the three
//     basis rows (recovered from efc.J as (r0 + r1) / 2, (r0 - r1) / 2, (r2 - r3) / 2), J x of the four rows is formed inside the
//     contact's lane quad as d_N +- d_T1, d_N +- d_T2 (three quad_perm DPP reads), and J^T f = N^T (f0 + f1 + f2 + f3) + T1^T (f0 - f1)
//     + T2^T (f2 - f3) runs over three rows instead of four: - 25 % LDS words and - 25 % of the J^T f reads and FMAs;
//   * one row POOL per workgroup: a world takes the rows it needs (3 per condim-3 contact + 1 per other row, rounded to 4) instead of 64;
//     three wavefronts per SIMD (the register budget: 168) leave 58 rows per world on average where the humanoid's steady state needs
//     36-40.  A world that does not fit (or has friction-loss rows / a contact cut by njmax) is flagged solver_niter = -1 and solved by
//     the fallback launch that follows on the same stream (k_solve<cg> restricted to flagged worlds: empty in the common case);
//   * no vector ever goes through LDS: an nv-vector lives one element per lane, v_permlane16_swap + DPP row_newbcast feed the
//     matrix-vector FMAs (solver_cgw.hpp's scheme in a 32-lane group) -- as v_fmac_f32_dpp, written in inline assembly: the compiler only
//     folds a DPP move into VOP2 opcodes and selects the three-operand v_fma_f32 here (238 v_mov_b32_dpp + v_fma pairs measured);
//   * J^T f in ONE round trip: lane (cq, rg) = (lane / 4, lane % 4) reads the columns 4 cq .. 4 cq + 3 of the rows 4 rg .. 4 rg + 3 of every
//     16-row batch as 16-byte words (a 36-row world: 12 reads per lane where the row loop needed 36 + 9; round 6: rows 4 apart across the
//     quad's lanes, conflict-free), the lane's four basis forces as one more, two quad_perm adds fold the four row classes and lane l --
//     dof l -- finds its column in its own quad;
//   * J is dead when the solve ends: the fused integrator's scratch lines alias the world's pool rows.
//   * a wavefront whose two worlds have at most 32 rows (the driver's window of the headline rollout: free fall, first contacts) runs an
//     instantiation of the iteration loop with ONE row per lane -- row dots, forces and line search over one slot, J^T f without the second
//     16-row batch when both worlds are within 16 basis rows --, picked by a ballot in front of the loop.
// Rows in the lanes: slot s = lane + 32 k (k = 0, 1).  The rows of condim-3 contacts first -- the q-th such contact owns the quad of slots
// 4q .. 4q + 3 --, then every other row (equality, limit, frictionless contact) in efc order: a stable partition of the efc rows by
// "pyramid row or not", computed per world from efc.type with two ballots; slotR[s] = efc row of slot s.  Basis row of slot (q, e < 3):
// 3q + e; of the p-th other row: 3 nq + p.
#pragma once
#include <type_traits>
#include "solver.hpp"
#include "solver_cgw.hpp"

// LDS of a workgroup: [CGP_HEAD ints: the worlds' row requests] [per world: slotR 64 ints | fbT 64 floats + 8 (the word lanes without a
// basis force write to)] [pool: pool_rows x JS]
#ifndef MJH_CGP_HOIST
#define MJH_CGP_HOIST 1  // 1: the J words of batch 0 in front of the force hand-over (round 6: - 0.5 .. 1.4 % on the launch, bit-identical); 2: batch 1 as well
#endif
#define CGP_HEAD 32
#define CGP_WORLD 136
template <int NV4>
__host__ __device__ inline int cgp_pool_rows(size_t lds_bytes, int wpb) {
  constexpr int NVR = 4 * NV4, JS = (NV4 & 1) ? NVR : NVR + 4;
  const int words = (int)(lds_bytes / sizeof(float)) - CGP_HEAD - CGP_WORLD * wpb;
  return words <= 0 ? 0 : ((words / JS) & ~3);
}
// rows every solved world allocates at least: the Gauss-Jordan tile of the prologue (2 x 4 x NVR words), the fused integrator's lines
template <int NV4>
__host__ __device__ inline int cgp_min_rows(int fuse_euler) {
  constexpr int NVR = 4 * NV4, JS = (NV4 & 1) ? NVR : NVR + 4;
  const int words = fuse_euler == 2 ? (6 * 32 + 12 * NV4 + 4 + 32) : (8 * NVR > 32 ? 8 * NVR : 32);
  return (((words + JS - 1) / JS) + 3) & ~3;
}

template <int CTRL>
DEV float qperm(float v) {  // quad_perm DPP read (CTRL = a | b << 2 | c << 4 | d << 6); every lane has a source: no old value to set up
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
// acc += a * (lane C of the reader's 16-lane row of xs): one VALU instruction
template <int C>
DEV void fmac_rbc(float& acc, float a, float xs) {
  asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(xs), "v"(a), "n"(C));
}
// The hazard recogniser does not see into inline assembly: a DPP read needs two wait states after the VALU write of its source and five
// after a VALU write of EXEC.  Tying the broadcast pair to the s_nop orders its producers in front of it and every fmac_rbc behind it.
DEV void dpp_fence(BV& x) { asm volatile("s_nop 4" : "+v"(x.a), "+v"(x.b)); }
// four of them in ONE asm statement (the compiler otherwise pads every second v_fmac_f32_dpp with an s_nop: it cannot see that the
// accumulators -- plain operands -- need no wait states)
template <int C>
DEV void fmac4_rbc(float& s0, float& s1, float a0, float a1, float a2, float a3, float xs) {
  asm volatile(
      "v_fmac_f32_dpp %0, %2, %3 row_newbcast:%7 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %1, %2, %4 row_newbcast:%8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %0, %2, %5 row_newbcast:%9 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %1, %2, %6 row_newbcast:%10 row_mask:0xf bank_mask:0xf bound_ctrl:1"
      : "+v"(s0), "+v"(s1)
      : "v"(xs), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "n"(C), "n"(C + 1), "n"(C + 2), "n"(C + 3));
}
// N (a multiple of 4) terms: s[C & 1] += row[OFF + C] * x[C], x[C] = lane C of the reader's 16-lane row of xs
template <int OFF, int N, int NT>
DEV void fmac_seq(float (&s)[2], const float (&row)[N], float xs) {
  static_assert(NT % 4 == 0, "four terms per asm statement");
  if constexpr (NT >= 4) fmac4_rbc<0>(s[0], s[1], row[OFF], row[OFF + 1], row[OFF + 2], row[OFF + 3], xs);
  if constexpr (NT >= 8) fmac4_rbc<4>(s[0], s[1], row[OFF + 4], row[OFF + 5], row[OFF + 6], row[OFF + 7], xs);
  if constexpr (NT >= 12) fmac4_rbc<8>(s[0], s[1], row[OFF + 8], row[OFF + 9], row[OFF + 10], row[OFF + 11], xs);
  if constexpr (NT >= 16) fmac4_rbc<12>(s[0], s[1], row[OFF + 12], row[OFF + 13], row[OFF + 14], row[OFF + 15], xs);
}
// two rows against the same vector, interleaved: four independent accumulator chains
template <int N, int NT>
DEV void fmac_seq2(float (&s)[4], const float (&r0)[N], const float (&r1)[N], float xs) {
  static_assert(NT % 4 == 0, "four terms per asm statement");
  if constexpr (NT >= 4) { fmac4_rbc<0>(s[0], s[1], r0[0], r0[1], r0[2], r0[3], xs); fmac4_rbc<0>(s[2], s[3], r1[0], r1[1], r1[2], r1[3], xs); }
  if constexpr (NT >= 8) { fmac4_rbc<4>(s[0], s[1], r0[4], r0[5], r0[6], r0[7], xs); fmac4_rbc<4>(s[2], s[3], r1[4], r1[5], r1[6], r1[7], xs); }
  if constexpr (NT >= 12) { fmac4_rbc<8>(s[0], s[1], r0[8], r0[9], r0[10], r0[11], xs); fmac4_rbc<8>(s[2], s[3], r1[8], r1[9], r1[10], r1[11], xs); }
  if constexpr (NT >= 16) { fmac4_rbc<12>(s[0], s[1], r0[12], r0[13], r0[14], r0[15], xs); fmac4_rbc<12>(s[2], s[3], r1[12], r1[13], r1[14], r1[15], xs); }
}

// invert_rows_b4 (solver.hpp) with the 4 x 4 pivot block inverted through its L D L^T factors: the block is symmetric positive definite (a
// Schur complement of M), and every lane inverts it redundantly -- the unrolled Gauss-Jordan of solver.hpp is 150 of the 280 VALU instructions
// of a block step, this is 70.  P^-1 = L^-T D^-1 L^-1 from the lower triangle of P.
template <int NVR, int G>
DEV void invert_rows_b4s(const float (&mrow)[NVR], float (&s)[NVR], float* buf, int lig) {
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
    const float4 P0 = *reinterpret_cast<const float4*>(pb + k), P1 = *reinterpret_cast<const float4*>(pb + NVR + k),
                 P2 = *reinterpret_cast<const float4*>(pb + 2 * NVR + k), P3 = *reinterpret_cast<const float4*>(pb + 3 * NVR + k);
    // L D L^T of the lower triangle
    const float i0 = 1.0f / P0.x;
    const float l10 = P1.x * i0, l20 = P2.x * i0, l30 = P3.x * i0;
    const float i1 = 1.0f / (P1.y - l10 * P1.x);
    const float t21 = P2.y - l20 * P1.x, t31 = P3.y - l30 * P1.x;
    const float l21 = t21 * i1, l31 = t31 * i1;
    const float i2 = 1.0f / (P2.z - l20 * P2.x - l21 * t21);
    const float t32 = P3.z - l30 * P2.x - l31 * t21;
    const float l32 = t32 * i2;
    const float i3 = 1.0f / (P3.w - l30 * P3.x - l31 * t31 - l32 * t32);
    // N = L^-1 (unit lower triangular)
    const float n10 = -l10, n21 = -l21, n32 = -l32;
    const float n20 = -l20 - l21 * n10, n31 = -l31 - l32 * n21;
    const float n30 = -l30 - l31 * n10 - l32 * n20;
    // P^-1 = N^T D^-1 N (symmetric)
    const float e30 = i3 * n30, e31 = i3 * n31, e32 = i3 * n32, e20 = i2 * n20, e21 = i2 * n21, e10 = i1 * n10;
    float Pi[4][4];
    Pi[3][3] = i3;
    Pi[2][3] = Pi[3][2] = e32;
    Pi[1][3] = Pi[3][1] = e31;
    Pi[0][3] = Pi[3][0] = e30;
    Pi[2][2] = i2 + n32 * e32;
    Pi[1][2] = Pi[2][1] = e21 + n31 * e32;
    Pi[0][2] = Pi[2][0] = e20 + n30 * e32;
    Pi[1][1] = i1 + n21 * e21 + n31 * e31;
    Pi[0][1] = Pi[1][0] = e10 + n20 * e21 + n30 * e31;
    Pi[0][0] = i0 + n10 * e10 + n20 * e20 + n30 * e30;
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

template <int NV4>
DEV void solve_cgp_body(const MjhModel& m, const MjhData& d, float* smem, int slot0, int nwb, int pool_rows, int fuse_euler) {
  constexpr int G = 32, NR = 2, NVR = 4 * NV4, JS = (NV4 & 1) ? NVR : NVR + 4, J4 = JS / 4;
  constexpr int NA = NVR < 16 ? NVR : 16, NBX = NVR - NA;  // columns served by the a / b half of a broadcast pair
  const int nv = m.nv, nC = m.nC, njmax = d.njmax, nvp = d.nv_pad;
  const int lig = threadIdx.x & (G - 1), gib = threadIdx.x / G;
  const int slot = slot0 + gib;
  const bool valid = gib < nwb && slot < d.nworld;
  int* cnt = reinterpret_cast<int*>(smem);
  int* slotR = reinterpret_cast<int*>(smem) + CGP_HEAD + CGP_WORLD * gib;  // efc row of every slot of this world
  float* fbT = smem + CGP_HEAD + CGP_WORLD * gib + 64;                      // basis forces fbT[b] (+ the spare word 64)
  float* pool = smem + CGP_HEAD + CGP_WORLD * (blockDim.x / G);
  const bool active = lig < nv;

  // ---- every global load of the prologue that only needs the world index, issued together: the row counts and types, the dense address
  // table of M (model-wide) and this lane's row of M through it, the dof scalars -------------------------------------------------------
  const int w = valid ? d.ws_order[slot] : 0;  // longest expected solve first (k_schedule_worlds)
  const size_t vo = (size_t)w * nv, eo = (size_t)w * njmax;
  int idx[NVR];
  {
    const int nv4r = (nv + 3) >> 2;
    const int4* tab = reinterpret_cast<const int4*>(m.M_dense) + (size_t)(active ? lig : 0) * nv4r;
#pragma unroll
    for (int c4 = 0; c4 < NV4; ++c4) {
      const int4 t4 = c4 < nv4r ? tab[c4] : make_int4(-1, -1, -1, -1);
      idx[4 * c4] = t4.x; idx[4 * c4 + 1] = t4.y; idx[4 * c4 + 2] = t4.z; idx[4 * c4 + 3] = t4.w;
    }
  }
  int nefc = 0, ne = 0, nf = 0, npf = 0;
  int ty[2] = {0, 0};
  if (valid) {
    nefc = min(d.nefc[w], njmax);
    ne = d.ne[w];
    nf = d.nf[w];
    npf = ne + nf + d.nl[w];
#pragma unroll
    for (int k = 0; k < 2; ++k) ty[k] = lig + G * k < njmax ? d.efc_type[eo + lig + G * k] : 0;
  }
  float mrow[NVR];
  {
    const float* Mg = d.M + (size_t)w * nC;
#pragma unroll
    for (int c = 0; c < NVR; ++c) {
      const float v = Mg[idx[c] < 0 ? 0 : idx[c]];
      mrow[c] = active ? (idx[c] < 0 ? 0.0f : v) : (c == lig ? 1.0f : 0.0f);
    }
  }
  const float fs = active ? d.qfrc_smooth[vo + lig] : 0.0f;
  const float qwarm = active ? d.qacc_warmstart[vo + lig] : 0.0f;

  // ---- stable partition of the efc rows: pyramid rows (four per condim-3 contact) to the front, the others behind them in order --------
  int np = 0, nq = 0, need = 0;
  bool defer = false;
  if (valid) {
    npf = min(npf, nefc);
    bool is6[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int r = lig + G * k;
      is6[k] = r >= npf && r < nefc && ty[k] == CT_CONTACT_PYRAMIDAL;
    }
    const unsigned m0 = (unsigned)gballot<G>(is6[0]), m1 = (unsigned)gballot<G>(is6[1]);
    const int n60 = __popc(m0), n6 = n60 + __popc(m1);
    const unsigned below = (1u << lig) - 1u;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int r = lig + G * k;
      const int c6 = k == 0 ? __popc(m0 & below) : n60 + __popc(m1 & below);
      if (r < nefc && r < 64) slotR[is6[k] ? c6 : n6 + r - c6] = r;
      fbT[r] = 0.0f;  // (basis rows past nb carry zero force for good)
    }
    nq = n6 >> 2;
    np = nefc - n6;
    defer = nf > 0 || (n6 & 3) != 0 || nefc > 64;  // friction loss (three-zone rows), a contact cut by njmax: the fallback launch
    need = max((np + 3 * nq + 15) & ~15, cgp_min_rows<NV4>(fuse_euler));  // (16-row batches of J^T f: the rows up to the batch boundary are zero rows)
  }
  // ---- the workgroup's (sequential, identical in every lane) pool allocation ---------------------------------------------------------
  if (lig == 0 && gib < CGP_HEAD) cnt[gib] = (valid && !defer) ? need : 0;
  __syncthreads();
  int base = 0;
  {
    int acc = 0;
    for (int j = 0; j < nwb; ++j) {
      const int c = cnt[j];
      const bool fits = acc + c <= pool_rows;
      if (j == gib) {
        base = acc;
        if (!fits) defer = true;
      }
      acc += fits ? c : 0;
    }
  }
  if (!valid) return;
  if (defer) {
    if (lig == 0) d.solver_niter[w] = -1;  // (the fallback launch solves the worlds flagged -1 and writes their iteration count)
    return;
  }
  float* Jl = pool + (size_t)base * JS;
  const int nb = np + 3 * nq, nb16 = (nb + 15) & ~15;

  PhaseClock pc(5, lig);
  pc.mark(0);
  const bool warm = !(m.disableflags & DSBL_WARMSTART);
  // lane i: sum_c A[i][c] x[c]; x arrives as a broadcast pair (a: x[l % 16], b: x[16 + l % 16]) and every x[c] an FMA needs is the DPP
  // operand row_newbcast:c of one of the two -- no LDS.  (Every lane of the group must execute this: a DPP read from a lane that sits
  // out a branch is zero.)
  auto mul_row = [&](const float (&row)[NVR], BV x) __attribute__((always_inline)) {
    float s[2] = {0.0f, 0.0f};
    dpp_fence(x);
    fmac_seq<0, NVR, NA>(s, row, x.a);
    if constexpr (NBX > 0) fmac_seq<NA, NVR, NBX>(s, row, x.b);
    return active ? s[0] + s[1] : 0.0f;
  };
  // ---- M^-1 (the CG preconditioner; the world's pool rows lend the tile buffer) and qacc_smooth with one step of refinement ----------
  float h[NVR];
  invert_rows_b4s<NVR, G>(mrow, h, Jl, lig);
  float qs = mul_row(h, bcast_prep(fs));
  {
    const float res = fs - mul_row(mrow, bcast_prep(qs));
    qs += mul_row(h, bcast_prep(active ? res : 0.0f));
  }
  if (active) d.qacc_smooth[vo + lig] = qs;
  float q = active ? (nefc > 0 && warm ? qwarm : qs) : 0.0f;
  pc.mark(1);
  float Ma = mul_row(mrow, bcast_prep(q));
  if (nefc == 0) {  // unconstrained: qacc = qacc_smooth (solver.py:3684-3686)
    if (active) {
      d.qacc[vo + lig] = q;
      d.qfrc_constraint[vo + lig] = 0.0f;
      d.efc_Ma[vo + lig] = Ma;
    }
    if (lig == 0) d.solver_niter[w] = 0;
    if (fuse_euler) {
      float qi = q;
      gsync();
      if (fuse_euler == 2) qi = impfast_acc<NV4, G>(m, d, w, lig, active, mrow, Ma, Jl);
      euler_advance<G>(m, d, w, lig, active, qi, Jl + (fuse_euler == 2 ? 6 * G + 12 * NV4 + 4 : 0), q);
    }
    return;
  }

  // ---- this lane's rows: efc row, D, aref (loads first), kind, where their basis row and basis force live ----------------------------
  float rD[NR], rja[NR], rjv[NR];
  int rkind[NR];  // 0 equality, 2 limit / contact, 3 padding (no friction-loss rows here)
  int jro[NR];    // word offset in the world's pool of the basis row this lane reads in the row dots
  int fbo[NR];    // word of fbT that takes this lane's basis force (64: none -- the spare word)
  bool isq[NR];
  int rer[NR];    // efc row of the slot
  const int qd = lig & 3;
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const int s = lig + G * k;
    const bool has = s < nefc;
    isq[k] = s < 4 * nq;
    const int er = has ? slotR[s] : 0;
    rer[k] = er;
    const int br = isq[k] ? 3 * (s >> 2) + (qd < 3 ? qd : 2) : (has ? s - nq : 0);
    const int bf_ = isq[k] ? (qd < 3 ? 3 * (s >> 2) + qd : -1) : s - nq;  // basis row that takes this slot's force
    jro[k] = br * JS;
    fbo[k] = (!has || bf_ < 0) ? 64 : bf_;
    rD[k] = has ? d.efc_D[eo + er] : 0.0f;
    rkind[k] = !has ? 3 : (isq[k] ? 2 : (er < ne ? 0 : 2));
    rjv[k] = 0.0f;
    rja[k] = has ? d.efc_aref[eo + er] : 0.0f;  // (aref for now: Jaref = J q - aref below)
  }
  // ---- basis rows into the pool: contact c -> N, T1, T2 from its four pyramid rows; the other rows as they are -------------------------
  gsync();  // (the tile reads of the inversion are done)
  {
    const float* Jg = d.efc_J + (size_t)w * d.njmax_pad * nvp;
    const int nvp4 = nvp >> 2;  // (nv_pad is a multiple of 4: rows of efc.J are 16-byte aligned)
    for (int it = lig; it < nq * J4; it += G) {
      const int c = it / J4, c4 = it - c * J4;
      float4 r0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), r1 = r0, r2 = r0, r3 = r0;
      if (c4 < nvp4) {
        const float4* src = reinterpret_cast<const float4*>(Jg + (size_t)slotR[4 * c] * nvp) + c4;  // (a contact's four rows are consecutive)
        r0 = src[0]; r1 = src[nvp4]; r2 = src[2 * nvp4]; r3 = src[3 * nvp4];
      }
      float4* dst = reinterpret_cast<float4*>(Jl + (size_t)(3 * c) * JS) + c4;
      dst[0] = make_float4(0.5f * (r0.x + r1.x), 0.5f * (r0.y + r1.y), 0.5f * (r0.z + r1.z), 0.5f * (r0.w + r1.w));
      dst[J4] = make_float4(0.5f * (r0.x - r1.x), 0.5f * (r0.y - r1.y), 0.5f * (r0.z - r1.z), 0.5f * (r0.w - r1.w));
      dst[2 * J4] = make_float4(0.5f * (r2.x - r3.x), 0.5f * (r2.y - r3.y), 0.5f * (r2.z - r3.z), 0.5f * (r2.w - r3.w));
    }
    for (int it = lig; it < (nb16 - 3 * nq) * J4; it += G) {  // the other rows, then zero rows up to the multiple of 16
      const int p = it / J4, c4 = it - p * J4;
      float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (p < np && c4 < nvp4) v = (reinterpret_cast<const float4*>(Jg + (size_t)slotR[4 * nq + p] * nvp))[c4];
      (reinterpret_cast<float4*>(Jl + (size_t)(3 * nq + p) * JS))[c4] = v;
    }
  }
  gsync();
  // J[row of slot k, :] . x for both slots: the two basis-row dots interleaved (four accumulator chains), then -- inside a contact's quad --
  // d_N +- d_T1 / d_N +- d_T2
  // `mv` (optional): M x rides between the J loads and their use -- its 28 FMAs cover the LDS latency of the first batch
  // NX = 1: a wavefront whose worlds have at most G rows -- every lane's second slot is empty -- reads and multiplies one row per lane
  auto j_dots = [&](auto nxt, BV x, float (&out)[NR], float* mv) __attribute__((always_inline)) {
    constexpr int NX = decltype(nxt)::value;
    float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    float r0[NA], r1[NA];
    auto load_a = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int c4 = 0; c4 < NA / 4; ++c4) {
        const float4 a4 = *reinterpret_cast<const float4*>(Jl + jro[0] + 4 * c4);
        r0[4 * c4] = a4.x; r0[4 * c4 + 1] = a4.y; r0[4 * c4 + 2] = a4.z; r0[4 * c4 + 3] = a4.w;
        if constexpr (NX > 1) {
          const float4 b4 = *reinterpret_cast<const float4*>(Jl + jro[1] + 4 * c4);
          r1[4 * c4] = b4.x; r1[4 * c4 + 1] = b4.y; r1[4 * c4 + 2] = b4.z; r1[4 * c4 + 3] = b4.w;
        }
      }
    };
    if constexpr (NV4 <= 7) load_a();  // (32 more live registers across the M product: the 32-column instantiation has none to spare)
    dpp_fence(x);
    if (mv) {
      float t[2] = {0.0f, 0.0f};
      fmac_seq<0, NVR, NA>(t, mrow, x.a);
      if constexpr (NBX > 0) fmac_seq<NA, NVR, NBX>(t, mrow, x.b);
      *mv = active ? t[0] + t[1] : 0.0f;
    }
    if constexpr (NV4 > 7) load_a();
    constexpr int NB_ = NBX > 0 ? NBX : 4;
    float q0[NB_], q1[NB_];
    if constexpr (NBX > 0) {
#pragma unroll
      for (int c4 = 0; c4 < NBX / 4; ++c4) {
        const float4 a4 = *reinterpret_cast<const float4*>(Jl + jro[0] + NA + 4 * c4);
        q0[4 * c4] = a4.x; q0[4 * c4 + 1] = a4.y; q0[4 * c4 + 2] = a4.z; q0[4 * c4 + 3] = a4.w;
        if constexpr (NX > 1) {
          const float4 b4 = *reinterpret_cast<const float4*>(Jl + jro[1] + NA + 4 * c4);
          q1[4 * c4] = b4.x; q1[4 * c4 + 1] = b4.y; q1[4 * c4 + 2] = b4.z; q1[4 * c4 + 3] = b4.w;
        }
      }
    }
    if constexpr (NX > 1) {
      fmac_seq2<NA, NA>(s, r0, r1, x.a);
      if constexpr (NBX > 0) fmac_seq2<NB_, NBX>(s, q0, q1, x.b);
    } else {
      float s2[2] = {0.0f, 0.0f};
      fmac_seq<0, NA, NA>(s2, r0, x.a);
      if constexpr (NBX > 0) fmac_seq<0, NB_, NBX>(s2, q0, x.b);
      s[0] = s2[0];
      s[1] = s2[1];
    }
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      if (k >= NX) {
        out[k] = 0.0f;
        continue;
      }
      const float dt = s[2 * k] + s[2 * k + 1];
      const float dn = qperm<0x00>(dt), d1 = qperm<0x55>(dt), d2 = qperm<0xAA>(dt);
      const float tq = qd < 2 ? d1 : d2;
      const float dq = dn + ((qd & 1) ? -tq : tq);
      out[k] = rkind[k] == 3 ? 0.0f : (isq[k] ? dq : dt);
    }
  };
  {
    float jq[NR];
    j_dots(std::integral_constant<int, NR>{}, bcast_prep(q), jq, nullptr);
#pragma unroll
    for (int k = 0; k < NR; ++k) rja[k] = rkind[k] != 3 ? jq[k] - rja[k] : 0.0f;
  }
  pc.mark(2);
  const float tolerance = bf(m.opt_tolerance, m.opt_tolerance_nb, w, 1)[0];
  const float ls_tolerance = bf(m.opt_ls_tolerance, m.opt_ls_tolerance_nb, w, 1)[0];
  const float meaninertia = bf(m.stat_meaninertia, m.stat_meaninertia_nb, w, 1)[0];
  const float scale = meaninertia * (float)nv;
  const float rscale = 1.0f / scale;
  // J^T f: lane (cq, rg) reads the columns 4 cq .. 4 cq + 3 of the rows rg + 4 i: one base address, every row an immediate offset
  // Round 6: lane (cq, rg) takes the rows 4 rg .. 4 rg + 3 of a 16-row batch (it took rg, rg + 4, rg + 8, rg + 12).  The four lanes of a column
  // quad then read rows 4 apart at the same time: with rows of JS = 12 / 20 / 28 / 36 words their bank offsets are 0 / 16 / 32 / 48 (mod 64),
  // and the 16 float4 of a ds_read_b128 lane group tile the 64 banks exactly -- the reads are conflict-free, where rows 1 apart collided pairwise
  // (twice the LDS cycles: SQ_LDS_BANK_CONFLICT 7.0 M cycles per launch in round 5, 23 % of the LDS pipe).  Lanes past the last column quad
  // (their sums are dropped) read the quad four below their own instead of the last one, which would break the tiling.
  const int qd4 = lig >> 2;
  const int cq = qd4 < J4 ? qd4 : (qd4 - 4 >= 0 && qd4 - 4 < J4 ? qd4 - 4 : J4 - 1), rg = lig & 3;
  const float* Jq = Jl + 4 * cq + 4 * rg * JS;
  // (a batch past the world's rows is redirected to its last one: finite numbers against zero forces instead of another world's LDS)
  const float *Jq1 = Jq + min(16, nb16 - 16) * JS, *Jq2 = Jq + min(32, nb16 - 16) * JS, *Jq3 = Jq + min(48, nb16 - 16) * JS;

  float grad_dot = 0.0f, search_dot = 0.0f;
  float g = 0.0f, Mg = 0.0f, pg = 0.0f, pMg = 0.0f, srch = 0.0f, qc = 0.0f;
  float cg5[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  int niter = 0;
  const int maxiter = m.iterations, ls_iterations = m.ls_iterations;
  int ovf = 0;
  float improvement = 0.0f;
  const bool short16 = __ballot(nb16 > 16) == 0ull;
  auto iterate = [&](auto nxt) __attribute__((always_inline)) {
  constexpr int NX = decltype(nxt)::value;
  for (;;) {
#if MJH_CGP_HOIST
    // the J words of batch 0 are loop invariant: issued in front of the force update and its LDS hand-over instead of behind the fence, their
    // latency rides under both (steps 300-320: solver launch 158.3 -> 156.1 us, three interleaved rounds; steps 5-25 unchanged)
    const float4 hj0 = *reinterpret_cast<const float4*>(Jq), hj1 = *reinterpret_cast<const float4*>(Jq + JS),
                 hj2 = *reinterpret_cast<const float4*>(Jq + 2 * JS), hj3 = *reinterpret_cast<const float4*>(Jq + 3 * JS);
#endif
#if MJH_CGP_HOIST > 1
    float4 gj0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), gj1 = gj0, gj2 = gj0, gj3 = gj0;
    if (!(NX == 1 && short16)) {
      gj0 = *reinterpret_cast<const float4*>(Jq1); gj1 = *reinterpret_cast<const float4*>(Jq1 + JS);
      gj2 = *reinterpret_cast<const float4*>(Jq1 + 2 * JS); gj3 = *reinterpret_cast<const float4*>(Jq1 + 3 * JS);
    }
#endif
    // ---- force of this lane's rows (solver.py:1698-1822), folded to basis forces inside the contact quads -------------------------
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const bool quad = (rkind[k] == 0) | ((rkind[k] == 2) & (rja[k] < 0.0f));  // (bitwise: no short-circuit branches on the chain)
      const float f = quad ? -rD[k] * rja[k] : 0.0f;
      const float a = qperm<0xB1>(f);          // the pair partner: f1 f0 f3 f2
      const float sm = f + a, df = f - a;      // f0 + f1 | f2 + f3 ;  f0 - f1, f1 - f0, f2 - f3, f3 - f2
      const float tot = sm + qperm<0x4E>(sm);  // f0 + f1 + f2 + f3
      const float bq = qd == 0 ? tot : (qd == 1 ? -df : df);
      fbT[fbo[k]] = isq[k] ? bq : f;
    }
    gsync();
    // ---- qfrc_constraint = J^T force (solver.py:1912-1947) over the basis rows, one LDS round trip ------------------------------------
    {
      float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
      auto batch = [&](int kb, const float* Jb) __attribute__((always_inline)) {
        const float4 f4 = *reinterpret_cast<const float4*>(fbT + 16 * kb + 4 * rg);  // the basis forces of the lane's four rows
        const float4 j0 = *reinterpret_cast<const float4*>(Jb), j1 = *reinterpret_cast<const float4*>(Jb + JS),
                     j2 = *reinterpret_cast<const float4*>(Jb + 2 * JS), j3 = *reinterpret_cast<const float4*>(Jb + 3 * JS);
        a0 += j0.x * f4.x; a1 += j0.y * f4.x; a2 += j0.z * f4.x; a3 += j0.w * f4.x;
        a0 += j1.x * f4.y; a1 += j1.y * f4.y; a2 += j1.z * f4.y; a3 += j1.w * f4.y;
        a0 += j2.x * f4.z; a1 += j2.y * f4.z; a2 += j2.z * f4.z; a3 += j2.w * f4.z;
        a0 += j3.x * f4.w; a1 += j3.y * f4.w; a2 += j3.z * f4.w; a3 += j3.w * f4.w;
      };
      // two round trips at most: batches 0 + 1 always (batch 1 of a world of at most 16 rows re-reads batch 0 against zero forces), batches
      // 2 + 3 behind one branch
#if MJH_CGP_HOIST
      {
        const float4 f4 = *reinterpret_cast<const float4*>(fbT + 4 * rg);
        a0 += hj0.x * f4.x; a1 += hj0.y * f4.x; a2 += hj0.z * f4.x; a3 += hj0.w * f4.x;
        a0 += hj1.x * f4.y; a1 += hj1.y * f4.y; a2 += hj1.z * f4.y; a3 += hj1.w * f4.y;
        a0 += hj2.x * f4.z; a1 += hj2.y * f4.z; a2 += hj2.z * f4.z; a3 += hj2.w * f4.z;
        a0 += hj3.x * f4.w; a1 += hj3.y * f4.w; a2 += hj3.z * f4.w; a3 += hj3.w * f4.w;
      }
#else
      batch(0, Jq);
#endif
#if MJH_CGP_HOIST > 1
      if (!(NX == 1 && short16)) {
        const float4 f4 = *reinterpret_cast<const float4*>(fbT + 16 + 4 * rg);
        a0 += gj0.x * f4.x; a1 += gj0.y * f4.x; a2 += gj0.z * f4.x; a3 += gj0.w * f4.x;
        a0 += gj1.x * f4.y; a1 += gj1.y * f4.y; a2 += gj1.z * f4.y; a3 += gj1.w * f4.y;
        a0 += gj2.x * f4.z; a1 += gj2.y * f4.z; a2 += gj2.z * f4.z; a3 += gj2.w * f4.z;
        a0 += gj3.x * f4.w; a1 += gj3.y * f4.w; a2 += gj3.z * f4.w; a3 += gj3.w * f4.w;
      }
#else
      if (!(NX == 1 && short16)) batch(1, Jq1);
#endif  // (uniform over the wavefront: both worlds within 16 basis rows)
      if (nb16 > 32) {
        batch(2, Jq2);
        batch(3, Jq3);
      }
      // the four row classes of a column quad are one lane quad: two butterfly adds, then lane l (dof l) takes column l % 4
      a0 += qperm<0xB1>(a0); a1 += qperm<0xB1>(a1); a2 += qperm<0xB1>(a2); a3 += qperm<0xB1>(a3);
      a0 += qperm<0x4E>(a0); a1 += qperm<0x4E>(a1); a2 += qperm<0x4E>(a2); a3 += qperm<0x4E>(a3);
      const float mine = qd == 0 ? a0 : (qd == 1 ? a1 : (qd == 2 ? a2 : a3));
      qc = active ? mine : 0.0f;
    }
    gsync();  // (the next iteration's force writes follow these reads)
    // ---- gradient, preconditioned gradient, Polak-Ribiere direction (solver.py:3061-3220, 3283-3450) ----------------------------------
    g = active ? (Ma - fs - qc) : 0.0f;
    pc.mark(3);
    Mg = mul_row(h, bcast_prep(g));
    cg5[0] = g * g; cg5[1] = g * (Mg - pMg); cg5[2] = pg * pMg; cg5[3] = Mg * Mg; cg5[4] = Mg * srch;
    gsum32_valu_n<5>(cg5);
    grad_dot = cg5[0];
    pc.mark(4);
    if (niter == 0) {
      srch = -Mg;
      search_dot = cg5[3];
      pg = g;
      pMg = Mg;
    } else {
      const float imp = improvement * rscale, gradient = sqrtf(grad_dot) * rscale;
      const float beta = fmaxf(0.0f, cg5[1] * __builtin_amdgcn_rcpf(fmaxf(MJ_MINVAL, cg5[2])));
      const bool done = (imp < tolerance) || (gradient < tolerance);
      if (done) break;
      srch = -Mg + beta * srch;
      search_dot = fmaxf(cg5[3] + beta * (beta * search_dot - 2.0f * cg5[4]), 0.0f);
      pg = g;
      pMg = Mg;
      if (niter >= maxiter) {
        ovf |= OVF_ITERATIONS;
        break;
      }
    }
    if (maxiter == 0) break;
    // ---- mv = M search, jv = J search -----------------------------------------------------------------------------------------------
    float mvi;
    j_dots(nxt, bcast_prep(srch), rjv, &mvi);
    pc.mark(5);
    // ---- line search (solver.py:835-1347); rows and all sums stay in registers -----------------------------------------------------
    const float g1 = srch * (Ma - fs);
    const float gtol = fmaxf(tolerance * ls_tolerance * sqrtf(search_dot) * scale, 1e-6f);
    float alpha = 0.0f;
    improvement = 0.0f;
    bool ls_converged = false;
    if constexpr (NX == NR) {
      line_search_rows<NR, G, false, 1>(rja, rjv, rD, rkind, nullptr, g1, 0.5f * srch * mvi, fabsf(g1), gtol, ls_iterations, alpha, improvement, ls_converged);
    } else {
      const float ja1[1] = {rja[0]}, jv1[1] = {rjv[0]}, D1[1] = {rD[0]};
      const int kind1[1] = {rkind[0]};
      line_search_rows<1, G, false, 1>(ja1, jv1, D1, kind1, nullptr, g1, 0.5f * srch * mvi, fabsf(g1), gtol, ls_iterations, alpha, improvement, ls_converged);
    }
    if (!ls_converged) ovf |= OVF_LS_ITERATIONS;
    pc.mark(6);
    q += alpha * srch;
    Ma += alpha * mvi;
#pragma unroll
    for (int k = 0; k < NX; ++k) rja[k] += alpha * rjv[k];
    ++niter;
    pc.mark(7);
  }
  };
  // (uniform over the wavefront: both its worlds within G rows -- the driver's window of the headline rollout, free fall and first contacts)
  if (__ballot(nefc > G) == 0ull) iterate(std::integral_constant<int, 1>{});
  else iterate(std::integral_constant<int, NR>{});
  pc.mark(8);
  // ---- outputs -----------------------------------------------------------------------------------------------------------------------
  if (active) {
    d.qacc[vo + lig] = q;
    d.qfrc_constraint[vo + lig] = qc;
    d.efc_Ma[vo + lig] = Ma;
  }
#pragma unroll
  for (int k = 0; k < NR; ++k)
    if (rkind[k] != 3) {  // force / state at the final iterate: the expression of the last constraint update
      const bool quad = (rkind[k] == 0) | (rja[k] < 0.0f);
      d.efc_force[eo + rer[k]] = quad ? -rD[k] * rja[k] : 0.0f;
      d.efc_state[eo + rer[k]] = quad ? ST_QUADRATIC : ST_SATISFIED;
    }
  if (lig == 0) {
    d.solver_niter[w] = niter;
    if (ovf) atomicOr(d.overflow + w, ovf);
  }
  if (fuse_euler) {
    gsync();  // (J is dead: the integrator's lines alias the world's pool rows)
    float qi = q;
    if (fuse_euler == 2) qi = impfast_acc<NV4, G>(m, d, w, lig, active, mrow, Ma, Jl);
    euler_advance<G>(m, d, w, lig, active, qi, Jl + (fuse_euler == 2 ? 6 * G + 12 * NV4 + 4 : 0), q);
  }
  pc.mark(9);
}
