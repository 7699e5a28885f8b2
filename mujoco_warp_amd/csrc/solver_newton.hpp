// solver_newton.hpp -- Newton solver for models with nv <= 32 and njmax <= 64: MFMA Hessian, blocked Cholesky, three
// wavefronts per SIMD.
//
// Reference: the same functions as solver.hpp (solver.py:3671-3743 solve/_solve, 3525-3620 _solver_iteration, 835-1347 line
// search, 1698-1822 constraint update, 2365-2440 _update_gradient_JTDAJ_dense_tiled, 2567-2603 Cholesky solve, 3454-3497
// _solve_done) -- the iteration is the one solver.hpp's solve_body<NEWTON> runs, step for step; what changes is the mapping.
//
// Why a second mapping (measured, profiles/round2_ubench.txt, round2_phase_newton_before.txt): solve_body<NEWTON> needs 256
// VGPRs + 76 AGPRs = ONE wavefront per SIMD, and a wavefront spends 110 k cycles per solve at 14 % VALU issue, half of it
// in the Hessian build + Cholesky, every LDS round trip (64 cycles) and every v_readlane hop (23) exposed.  A wave64 VALU op
// costs 2 cycles, so the whole launch holds ~30 us of VALU work inside 216 us.  Here:
//   * H = M + J^T D J comes from the matrix pipe: one v_mfma_f32_32x32x1_2b_f32 per constraint row does the rank-1 update
//     of BOTH worlds of the wavefront (block b = lanes 32b..32b+31: A = D_r J[r][:], B = J[r][:]); 16 v_permlane32_swap then
//     leave column i (= row i, H is symmetric) of world b's H in lane (b, i).  Per row: two LDS reads, one multiply, one
//     select; the VALU version needed 16 LDS reads and 60 FMAs per row pair.  f32 MFMA is bitwise an fmaf chain (k-ordered),
//     so the result is an ordinary float32 sum in row order.
//   * the Cholesky is right-looking over 4-column blocks: one LDS panel write per block, every lane refactors the 4x4
//     diagonal block redundantly in registers (no cross-lane traffic) and applies the Schur update from the raw panel:
//     7 LDS round trips per factorisation instead of 28.
//   * register discipline: the only long-lived array is the M row (28); factorisation, forward and backward substitution
//     are one routine that holds nothing but the H row (no transposed copy of L): <= 168 VGPRs = three wavefronts per SIMD.
//   * LDS decides the rest: the two worlds of a wavefront share ONE pool of J rows sized for the steady state (the humanoid
//     sits at 49-51 rows per world: pool 100 rows instead of 2 x 64), so that twelve wavefronts fit a CU.  A pair that does
//     not fit (rare) is solved one world after the other by the same wavefront ("two turns").
//   * wave-uniform control flow: the MFMA and the half swap are wavefront-wide, so the iteration loop runs until BOTH
//     worlds are finished; a finished (or parked, or absent) world keeps executing on frozen state with alpha = 0.
#pragma once
#include "solver.hpp"

typedef float f32x32 __attribute__((ext_vector_type(32)));

struct NewtonLayout {
  int pool;                       // J rows of both worlds (pool_rows x JS); stages the two dense M copies first
  int world, wstride;             // per-world scratch blocks: world + hf * wstride
  int force, da, bvec, panel;     // offsets inside a per-world block
  int total;                      // words per wavefront
};
template <int NV4>
__host__ __device__ inline NewtonLayout newton_layout(int pool_rows) {
  constexpr int NVR = 4 * NV4;
  constexpr int JS = (NV4 & 1) ? NVR : NVR + 4;  // JS/4 odd: row-per-lane 16-byte reads hit distinct banks
  NewtonLayout p;
  p.pool = 0;
  p.world = pool_rows * JS;
  int q = 0;
  p.force = q; q += 64;   // efc_force of the current iterate (J^T f)
  p.da = q; q += 64;      // D * [state == QUADRATIC] (J^T D J)
  p.bvec = q; q += 32;    // broadcast copy of the nv-vector other lanes read
  p.panel = q; q += (4 * JS > 128 ? 4 * JS : 128);  // Cholesky panel (32 lanes x 4) / 4-row transpose tile (4 x JS)
  p.wstride = q;
  p.total = ((p.world + 2 * q + 3) / 4) * 4;
  return p;
}
// J rows a wavefront's pool must hold at least: the dense staging copies of M (2 x NVR rows) and one world's rows
template <int NV4>
__host__ __device__ inline int newton_min_pool(int njmax) {
  const int one = njmax < 64 ? njmax : 64;
  return (2 * 4 * NV4 > one ? 2 * 4 * NV4 : one);
}

// x = H^-1 g by a blocked right-looking Cholesky with lane i owning row i (NVR <= 32) -- factorisation, forward and backward
// substitution in one routine, nothing but the H row (NVR registers) held across it.
//   h: row i of the SPD matrix on entry (the FULL row: the trailing block is kept symmetric), destroyed.  Lanes >= nv hold
//   identity rows.  LDS, private to the 32-lane group: panel (128 floats), vec (32), save (12 NV4, may alias dead arrays).
// Per 4-column block: every lane parks its four raw panel entries and its running right-hand side in LDS (one round trip),
// refactors the 4x4 diagonal block redundantly in registers (no cross-lane traffic), takes its own entries of L and applies
// the Schur update from the RAW panel (h[k] -= (x Ld^-1) . p_k = L_i . L_k); the forward substitution L y = g rides along
// (y_blk = Ld^-1 g_blk, g_i -= L_i,blk . y_blk).  The backward substitution L^T x = y needs columns of L, which live
// across lanes: per block four 32-lane DPP reductions of h[c] * x (lanes without an x yet contribute zero), then the 4x4
// back substitution in registers from the block factors saved in LDS.  No transposed copy of L, no v_readlane chain.
template <int NV4>
DEV float chol_factor_solve(float (&h)[4 * NV4], float g, float* panel, float* vec, float* save, int lig) {
  constexpr int NVR = 4 * NV4, G = 32;
  float gacc = g, y = 0.0f;
#pragma unroll
  for (int jb = 0; jb < NV4; ++jb) {
    const int j0 = 4 * jb;
    gsync();  // the previous block's panel reads are issued before this write (one wavefront: LDS ops complete in order)
    // unconditional (lanes >= NVR park junk in their own slot): a branch here splits the block, the compiler then sinks the
    // previous block's Schur FMAs past it while their panel loads must stay before this write -- 96 VGPRs held, spilled
    *reinterpret_cast<float4*>(panel + 4 * lig) = make_float4(h[j0], h[j0 + 1], h[j0 + 2], h[j0 + 3]);
    vec[lig] = gacc;
    gsync();
    // 4x4 diagonal block (rows j0..j0+3 of the panel), factored redundantly by every lane
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
    // (every lane stores the same twelve words: the block factors the backward pass needs again)
    *reinterpret_cast<float4*>(save + 12 * jb) = make_float4(l10, l20, l30, r0);
    *reinterpret_cast<float4*>(save + 12 * jb + 4) = make_float4(l21, l31, l32, r1);
    *reinterpret_cast<float2*>(save + 12 * jb + 8) = make_float2(r2, r3);
    // this lane's entries of L in the block's columns: x = p Ld^-T (for the block's own lanes: the rows of Ld)
    const float x0 = h[j0] * r0;
    const float x1 = (h[j0 + 1] - x0 * l10) * r1;
    const float x2 = (h[j0 + 2] - x0 * l20 - x1 * l21) * r2;
    const float x3 = (h[j0 + 3] - x0 * l30 - x1 * l31 - x2 * l32) * r3;
    h[j0] = x0;
    h[j0 + 1] = x1;
    h[j0 + 2] = x2;
    h[j0 + 3] = x3;
    // forward substitution of the block, then this lane's right-hand side loses the block's contribution
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
      // Schur update of the trailing rows from the RAW panel: h[k] -= (x Ld^-1) . p_k  (= L_i . L_k)
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
  // ---- backward substitution ------------------------------------------------------------------------------------------
  gsync();
  vec[lig] = y;
  gsync();
  float x = 0.0f;  // lanes >= NVR never receive a value: their (junk but finite) h entries are multiplied by zero
#pragma unroll
  for (int jb = NV4 - 1; jb >= 0; --jb) {
    const int j0 = 4 * jb;
    float t4[4] = {h[j0] * x, h[j0 + 1] * x, h[j0 + 2] * x, h[j0 + 3] * x};
    gsumg_n<G, 4>(t4);
    const float t0 = t4[0], t1 = t4[1], t2 = t4[2], t3 = t4[3];
    const float4 y4 = *reinterpret_cast<const float4*>(vec + j0);
    const float4 sa = *reinterpret_cast<const float4*>(save + 12 * jb);      // l10 l20 l30 r0
    const float4 sb = *reinterpret_cast<const float4*>(save + 12 * jb + 4);  // l21 l31 l32 r1
    const float2 sc = *reinterpret_cast<const float2*>(save + 12 * jb + 8);  // r2 r3
    const float x3 = (y4.w - t3) * sc.y;
    const float x2 = ((y4.z - t2) - sb.z * x3) * sc.x;
    const float x1 = ((y4.y - t1) - sb.x * x2 - sb.y * x3) * sb.w;
    const float x0 = ((y4.x - t0) - sa.x * x1 - sa.y * x2 - sa.z * x3) * sa.w;
    x = lig == j0 ? x0 : x;
    x = lig == j0 + 1 ? x1 : x;
    x = lig == j0 + 2 ? x2 : x;
    x = lig == j0 + 3 ? x3 : x;
  }
  return x;
}

// exact line search on the convex cost of ONE world (solver.py:835-1347), rows and sums in registers; the same arithmetic as
DEV bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

// Global accesses with a uniform base and a 32-bit per-lane BYTE offset: one VGPR per address (saddr + voffset form) instead
// of a 64-bit pair per array; several arrays of one world share the same offset register.  The host only selects this
// kernel when nworld * max(nv, njmax) * 4 fits 32 bits.
DEV float ldf(const float* base, unsigned boff) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + boff); }
DEV int ldi(const int* base, unsigned boff) { return *reinterpret_cast<const int*>(reinterpret_cast<const char*>(base) + boff); }
DEV void stf(float* base, unsigned boff, float v) { *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + boff) = v; }
DEV void sti(int* base, unsigned boff, int v) { *reinterpret_cast<int*>(reinterpret_cast<char*>(base) + boff) = v; }

// One world of a wavefront's pair (lanes of half hf); `live` = this half solves world w in this turn, otherwise it runs the
// same instruction stream on an empty problem and touches no global memory.  rbase = first pool row of this world.
template <int NV4>
DEV void newton_pair(const MjhModel& m, const MjhData& d, float* S, const NewtonLayout& lay, int w, bool live, int nefc_in, int rbase,
                     int fuse_euler, int tid) {
  constexpr int NVR = 4 * NV4, G = 32, NR = 2;
  constexpr int JS = (NV4 & 1) ? NVR : NVR + 4;
  const int nv = m.nv, nC = m.nC, njmax = d.njmax, nvp = d.nv_pad;
  const int lane = tid & 63, hf = lane >> 5, lig = lane & 31;
  const int nefc = live ? nefc_in : 0;
  float* Jl = S + lay.pool + rbase * JS;
  float* W = S + lay.world + hf * lay.wstride;
  float *eforce = W + lay.force, *eda = W + lay.da, *bvec = W + lay.bvec, *panel = W + lay.panel;
  const int ne = live ? d.ne[w] : 0, nf = live ? d.nf[w] : 0;
  const bool has_fl = nf > 0;
  const unsigned voff = 4u * ((unsigned)w * (unsigned)nv + (unsigned)lig);     // byte offset of element lig of this world's nv-vectors
  const unsigned eoff = 4u * ((unsigned)w * (unsigned)njmax + (unsigned)lig);  // ... of row lig of its efc vectors
  const bool active = live && lig < nv;
  const int ligr = lig < NVR ? lig : NVR - 1;

  PhaseClock pc(5, lig);
  // ---- M row of this lane into registers: dense staging in the pool (world hf: rows hf*NVR .. of the pool); rows that
  // carry no dof (lanes >= nv, a parked world) are identity rows ----------------------------------------------------------
  float mrow[NVR];
  {
    float* Ms = S + lay.pool + hf * NVR * JS;
    for (int idx = lig; idx < NVR * JS / 4; idx += G) reinterpret_cast<float4*>(Ms)[idx] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    gsync();
    if (!active) Ms[ligr * JS + ligr] = 1.0f;
    if (live) {
      const float* Mg = d.M + (size_t)w * nC;
      for (int i = lig; i < nv; i += G) {
        const int start = m.M_rowadr[i], n = m.M_rownnz[i];
        for (int a = 0; a < n; ++a) {
          const int j = m.M_colind[start + a];
          const float v = Mg[start + a];
          Ms[i * JS + j] = v;
          Ms[j * JS + i] = v;
        }
      }
    }
    gsync();
#pragma unroll
    for (int c4 = 0; c4 < NV4; ++c4) {
      const float4 v4 = *reinterpret_cast<const float4*>(Ms + ligr * JS + 4 * c4);
      mrow[4 * c4] = v4.x;
      mrow[4 * c4 + 1] = v4.y;
      mrow[4 * c4 + 2] = v4.z;
      mrow[4 * c4 + 3] = v4.w;
    }
    gsync();
  }
  pc.mark(0);
  const bool warm = !(m.disableflags & DSBL_WARMSTART);
  const float fs = active ? ldf(d.qfrc_smooth, voff) : 0.0f;
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
  float h[NVR];
  float* save = eforce;  // block factors of the Cholesky: eforce | eda (128 floats) are dead between H build and the next pass
  // cold start / unconstrained world: qacc_smooth from the Cholesky factor of M (wave-uniform: both halves factor)
  float qs = 0.0f;
  const bool cold = live && (nefc == 0 || !warm);
  if (wave_any(cold)) {
#pragma unroll
    for (int c = 0; c < NVR; ++c) h[c] = mrow[c];
    qs = chol_factor_solve<NV4>(h, fs, panel, bvec, save, lig);
    if (!active) qs = 0.0f;
  }
  float q = 0.0f;
  if (active) q = (nefc > 0 && warm) ? ldf(d.qacc_warmstart, voff) : qs;
  pc.mark(1);
  gsync();
  bvec[lig] = q;
  gsync();
  float Ma = mul_row(mrow, bvec);

  // ---- J rows into the pool; this lane's rows (D, aref, Jaref) into registers -------------------------------------------
  {
    const float* Jg = d.efc_J + (size_t)w * d.njmax_pad * nvp;
    if (nvp == JS) {
      const float4* src = reinterpret_cast<const float4*>(Jg);
      float4* dst = reinterpret_cast<float4*>(Jl);
      const int n4 = nefc * (JS / 4);
#pragma unroll 4
      for (int i = lig; i < n4; i += G) dst[i] = src[i];
    } else {
      for (int r = 0; r < nefc; ++r)
        for (int c = lig; c < JS; c += G) Jl[r * JS + c] = c < nvp ? Jg[(size_t)r * nvp + c] : 0.0f;
    }
  }
  float rD[NR], rja[NR], rjv[NR];
  int rkind[NR];
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const int r = lig + G * k;
    const bool has = r < nefc;
    rD[k] = has ? ldf(d.efc_D, eoff + 4u * G * k) : 0.0f;
    rkind[k] = !has ? 3 : (r >= ne + nf ? 2 : (r >= ne ? 1 : 0));  // 3: no row
    rjv[k] = 0.0f;
    eforce[r] = 0.0f;
    eda[r] = 0.0f;
  }
  gsync();
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
  for (int k = 0; k < NR; ++k) rja[k] = rkind[k] != 3 ? j_dot(bvec, lig + G * k) - ldf(d.efc_aref, eoff + 4u * G * k) : 0.0f;
  gsync();
  pc.mark(2);

  const float tolerance = bf(m.opt_tolerance, m.opt_tolerance_nb, w, 1)[0];
  const float ls_tolerance = bf(m.opt_ls_tolerance, m.opt_ls_tolerance_nb, w, 1)[0];
  const float meaninertia = bf(m.stat_meaninertia, m.stat_meaninertia_nb, w, 1)[0];
  const float scale = meaninertia * (float)nv;
  const float rscale = 1.0f / scale;
  const int maxiter = m.iterations, ls_iterations = m.ls_iterations;
  const float* floss_lane = reinterpret_cast<const float*>(reinterpret_cast<const char*>(d.efc_frictionloss) + eoff);

  float grad_dot = 0.0f, search_dot = 0.0f, decrement = 0.0f;
  float g = 0.0f, Mg = 0.0f, srch = 0.0f, qc = 0.0f, improvement = 0.0f;
  int niter = 0, ovf = 0;
  // fin: this world's solve is over (its state is frozen from here on); an unconstrained / parked / absent world starts finished
  bool fin = nefc == 0;
  // rows the MFMA loop covers (wave-uniform: the larger of the two worlds); lanes beyond the matrix feed zeros
  const int n_a = __builtin_amdgcn_readlane(nefc, 0), n_b = __builtin_amdgcn_readlane(nefc, 32);
  const int nmax = n_a > n_b ? n_a : n_b;
  const int nrow_lane = lig < NVR ? nefc : 0;

  for (;;) {
    // ---- force/state of this lane's rows (solver.py:1698-1822) ---------------------------------------------------------
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      float force;
      int state;
      row_force(rkind[k], rja[k], rD[k], has_fl, floss_lane + G * k, force, state);
      eforce[lig + G * k] = force;
      eda[lig + G * k] = state == ST_QUADRATIC ? rD[k] : 0.0f;
    }
    gsync();
    // ---- qfrc_constraint = J^T force: lane = dof, sixteen rows per trip (loads first), then 4-row steps, then the tail ---
    {
      float s0 = 0.0f, s1 = 0.0f;
      const float* Jc = Jl + ligr;
      const int n16 = nefc & ~15, n4 = nefc & ~3;
      int r = 0;
      for (; r < n16; r += 16) {
        float jj[16];
        float4 ff[4];
#pragma unroll
        for (int e = 0; e < 16; ++e) jj[e] = Jc[(r + e) * JS];
#pragma unroll
        for (int e = 0; e < 4; ++e) ff[e] = *reinterpret_cast<const float4*>(eforce + r + 4 * e);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s0 += jj[4 * e] * ff[e].x + jj[4 * e + 2] * ff[e].z;
          s1 += jj[4 * e + 1] * ff[e].y + jj[4 * e + 3] * ff[e].w;
        }
      }
      for (; r < n4; r += 4) {
        const float4 f4 = *reinterpret_cast<const float4*>(eforce + r);
        s0 += Jc[r * JS] * f4.x + Jc[(r + 2) * JS] * f4.z;
        s1 += Jc[(r + 1) * JS] * f4.y + Jc[(r + 3) * JS] * f4.w;
      }
      for (; r < nefc; ++r) s0 += Jc[r * JS] * eforce[r];
      qc = active ? s0 + s1 : 0.0f;
    }
    // ---- gradient; the improvement / gradient tests need no search direction (solver.py:3454-3497) ---------------------
    g = active ? (Ma - fs - qc) : 0.0f;
    grad_dot = gsumg<G>(g * g);
    if (!fin && niter > 0 && ((improvement * rscale < tolerance) || (sqrtf(grad_dot) * rscale < tolerance))) fin = true;
    pc.mark(3);
    if (!wave_any(!fin)) break;
    // ---- H = M + J^T D J on the matrix pipe, both worlds of the wavefront at once (solver.py:2365-2440) -----------------
    {
      f32x32 acc;
#pragma unroll
      for (int i = 0; i < 32; ++i) acc[i] = 0.0f;
      const float* Jc = Jl + ligr;
      // software pipeline: the four rows of trip t+1 are loaded (unconditionally: the address stays inside the wavefront's
      // LDS, a row that is not this world's is replaced by zeros afterwards) while the four MFMAs of trip t issue
      float jn[4];
      float4 dn = *reinterpret_cast<const float4*>(eda);
#pragma unroll
      for (int e = 0; e < 4; ++e) jn[e] = Jc[e * JS];
      for (int r0 = 0; r0 < nmax; r0 += 4) {
        float jc[4];
        const float dc[4] = {dn.x, dn.y, dn.z, dn.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) jc[e] = (r0 + e < nrow_lane) ? jn[e] : 0.0f;
        const int rn = r0 + 4 < 60 ? r0 + 4 : 60;  // (the prefetch of the last trip is unused)
        dn = *reinterpret_cast<const float4*>(eda + rn);
#pragma unroll
        for (int e = 0; e < 4; ++e) jn[e] = Jc[(rn + e) * JS];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x1f32(jc[e] * dc[e], jc[e], acc, 0, 0, 0);
      }
      pc.mark(10);
      // lane (b, i) gathers column i of world b's 32x32 block: registers 0-15 hold block 0, 16-31 block 1, each lane half
      // the rows of its column; one half swap per register pair completes the column (tools/ubench.hip checks this map)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[j]), __float_as_uint(acc[16 + j]), false, false);
        const int ra = 8 * (j >> 2) + (j & 3), rb = ra + 4;
        if (ra < NVR) h[ra] = mrow[ra] + __uint_as_float(sw[0]);
        if (rb < NVR) h[rb] = mrow[rb] + __uint_as_float(sw[1]);
      }
    }
    pc.mark(11);
    gsync();  // (eda is read by the H build above; the block factors overwrite it)
    Mg = chol_factor_solve<NV4>(h, g, panel, bvec, save, lig);
    pc.mark(12);
    if (!active) Mg = 0.0f;
    const float srch_new = -Mg;
    float sd2[2] = {Mg * Mg, g * Mg};
    gsumg_n<G, 2>(sd2);
    const float sd_new = sd2[0], dec_new = sd2[1];
    if (!fin) {
      srch = srch_new;
      search_dot = sd_new;
      decrement = dec_new;
      if (niter > 0) {
        if (0.5f * decrement * rscale < tolerance) fin = true;
        else if (niter >= maxiter) {
          ovf |= OVF_ITERATIONS;
          fin = true;
        }
      }
      if (maxiter == 0) fin = true;
    }
    pc.mark(4);
    if (!wave_any(!fin)) break;
    // ---- mv = M search, jv = J search ---------------------------------------------------------------------------------
    gsync();
    bvec[lig] = srch;
    gsync();
    const float mvi = mul_row(mrow, bvec);
#pragma unroll
    for (int k = 0; k < NR; ++k) rjv[k] = rkind[k] != 3 ? j_dot(bvec, lig + G * k) : 0.0f;
    pc.mark(5);
    // ---- line search ---------------------------------------------------------------------------------------------------
    const float gs2[3] = {srch * (Ma - fs), 0.5f * srch * mvi, fabsf(srch * (Ma - fs))};  // lane-local: reduced inside the line search with its alpha = 0 sums
    const float gtol = fmaxf(tolerance * ls_tolerance * sqrtf(search_dot) * scale, 1e-6f);
    float alpha, imp_new;
    bool ls_ok;
    // (a finished world rides along on frozen state: no bracketing iterations for it)
#ifdef MJH_PHASE_CLOCK
    int ls_its = 0;
    if (wave_any(has_fl)) line_search_rows<NR, G, true>(rja, rjv, rD, rkind, floss_lane, gs2[0], gs2[1], gs2[2], gtol, fin ? 0 : ls_iterations, alpha, imp_new, ls_ok, &ls_its);
    else line_search_rows<NR, G, false>(rja, rjv, rD, rkind, floss_lane, gs2[0], gs2[1], gs2[2], gtol, fin ? 0 : ls_iterations, alpha, imp_new, ls_ok, &ls_its);
    if (lig == 0 && !fin) {  // profiling build: bracketing iterations and calls of the line search (phase slots 14, 15 of kernel 5)
      atomicAdd(&g_phase_ticks[blockIdx.x & 63][5][14], (unsigned long long)ls_its);
      atomicAdd(&g_phase_ticks[blockIdx.x & 63][5][15], 1ull);
    }
#else
    if (wave_any(has_fl)) line_search_rows<NR, G, true>(rja, rjv, rD, rkind, floss_lane, gs2[0], gs2[1], gs2[2], gtol, fin ? 0 : ls_iterations, alpha, imp_new, ls_ok);
    else line_search_rows<NR, G, false>(rja, rjv, rD, rkind, floss_lane, gs2[0], gs2[1], gs2[2], gtol, fin ? 0 : ls_iterations, alpha, imp_new, ls_ok);
#endif
    pc.mark(6);
    if (!fin) {
      if (!ls_ok) ovf |= OVF_LS_ITERATIONS;
      improvement = imp_new;
      q += alpha * srch;
      Ma += alpha * mvi;
#pragma unroll
      for (int k = 0; k < NR; ++k) rja[k] += alpha * rjv[k];
      ++niter;
    }
    pc.mark(7);
  }
  pc.mark(8);

  // ---- outputs ---------------------------------------------------------------------------------------------------------
  if (active) {
    stf(d.qacc, voff, q);
    stf(d.qfrc_constraint, voff, qc);
    stf(d.efc_Ma, voff, Ma);
  }
#pragma unroll
  for (int k = 0; k < NR; ++k)
    if (rkind[k] != 3) {
      float force;
      int state;
      row_force(rkind[k], rja[k], rD[k], has_fl, floss_lane + G * k, force, state);
      stf(d.efc_force, eoff + 4u * G * k, force);
      sti(d.efc_state, eoff + 4u * G * k, state);
    }
  if (live && lig == 0) {
    d.solver_niter[w] = niter;
    if (ovf) atomicOr(d.overflow + w, ovf);
  }
  if (fuse_euler) {  // (wave-uniform: the blocked Cholesky of the implicitfast update runs in both halves, a parked world on identity rows)
    gsync();
    float qi = q;
    // scratch: eforce | eda | bvec | panel are contiguous (288 floats >= 6 G + 12 NV4) and dead by now
    if (fuse_euler == 2) qi = impfast_acc<NV4, G>(m, d, live ? w : 0, lig, active, mrow, Ma, eforce);
    if (live) euler_advance<G>(m, d, w, lig, active, qi, bvec, q);
  }
  gsync();
  pc.mark(9);
}

// One wavefront = two schedule slots.  FULLPOOL: the pool holds 2 x min(njmax, 64) rows, every pair fits (no second turn).
// Otherwise the pair shares the pool when its rows fit and is solved in two turns when they do not.
template <int NV4, bool FULLPOOL>
DEV void newton_body(const MjhModel& m, const MjhData& d, float* smem, int pool_rows, int fuse_euler, int block) {
  const NewtonLayout lay = newton_layout<NV4>(pool_rows);
  const int wave = block * ((int)blockDim.x >> 6) + ((int)threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, hf = lane >> 5;
  if (2 * wave >= d.nworld) return;  // whole wavefront beyond the world list
  float* S = smem + (size_t)((int)threadIdx.x >> 6) * lay.total;
  const int slot = 2 * wave + hf;
  const bool have = slot < d.nworld;
  int w = have ? d.ws_order[slot] : 0;
  const int cap = d.njmax < 64 ? d.njmax : 64;
  int nefc = have ? d.nefc[w] : 0;
  nefc = nefc < cap ? nefc : cap;
  const int n0 = __builtin_amdgcn_readlane(nefc, 0);
  if (FULLPOOL) {
    newton_pair<NV4>(m, d, S, lay, w, have, nefc, hf ? n0 : 0, fuse_euler, (int)threadIdx.x);
  } else {
    const int n1 = __builtin_amdgcn_readlane(nefc, 32);
    const bool shared = n0 + n1 <= pool_rows;  // both worlds' rows fit the pool: one turn, world 1's rows follow world 0's
    const int nturn = shared ? 1 : 2;
    for (int turn = 0; turn < nturn; ++turn) {  // (one call site: the body is inlined once)
      // opaque per turn: otherwise every lane- and world-derived value is hoisted out of this loop and stays live
      int tid = threadIdx.x;
      asm volatile("" : "+v"(tid), "+v"(w));
      newton_pair<NV4>(m, d, S, lay, w, have && (shared || hf == turn), nefc, (shared && hf) ? n0 : 0, fuse_euler, tid);
    }
  }
}

// WV = wavefronts per SIMD the register allocation is held to (3: 168 VGPRs, 2: 256)
template <int NV4, int WV, bool FULLPOOL>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WV, 8))) k_solve_newton(MjhModel m, MjhData d, int pool_rows, int fuse_euler, int nrider, int rider_at) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // riders (round 3; small models, see mjhip.hip newton_inline_ok): 2 nrider one-wavefront workgroups -- the L'DL factor + qacc_smooth, then the contact
  // publication, two worlds each -- inserted at workgroup rider_at of the dispatch order instead of running on a side stream
  const int bx = (int)blockIdx.x;
  if (bx >= rider_at && bx < rider_at + 2 * nrider) {
    const int bi = bx - rider_at;
    if (bi < nrider) factor_smooth_body<32>(m, d, 1, smem, Blk{bi * 2, 2, 64});
    else publish_body<32>(d, 1, reinterpret_cast<int*>(smem), Blk{(bi - nrider) * 2, 2, 64}, m.nexplicit ? m.pair_solreffriction : nullptr);
    return;
  }
  newton_body<NV4, FULLPOOL>(m, d, smem, pool_rows, fuse_euler, bx < rider_at ? bx : bx - 2 * nrider);
}
