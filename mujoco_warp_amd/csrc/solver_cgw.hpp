// solver_cgw.hpp -- CG with one world per WAVEFRONT, both halves of the wavefront at work in every phase (round 3).
//
// Reference: the same functions as solver.hpp (solver.py:3283-3450 CG, 835-1347 line search, 1698-1822 constraint update).
//
// Why a second CG kernel.  k_solve<cg> (solver.hpp) gives a world 32 lanes: lane i owns dof i -- a full row of M and of M^-1 -- and
// two constraint rows.  Its launch is a latency problem, not a throughput one: a CU holds 20 worlds whatever the
// kernel (LDS: the J tile), every world is one long dependent chain (~900 instructions per iteration, prologue of ~10 k), VALU and LDS
// are each < 40 % busy.  So the lever is the length of a world's chain, and this kernel halves the per-lane work of every phase
// instead of pairing two worlds in a wavefront:
//   * lane (i, h), h = lane / 32: HALF of row i of M and of M^-1 (columns [16 h, 16 h + 16)); a matrix-vector product is 16 FMAs and
//     one v_permlane32_swap that adds the two halves (both halves end with the same bits);
//   * lane r: ONE constraint row (64 rows per world) -- J r . search is one row dot, the line search evaluates one row per lane;
//   * J^T f: lane (rho, j), rho = lane / 16, sums the columns j and 16 + j over the rows [16 rho, 16 rho + 16); two swaps add the four parts;
//   * M^-1: the 4 x 4-blocked Gauss-Jordan of solver.hpp with the rank-4 update split over the column halves.
// No vector ever goes through LDS: an nv-vector lives one element per lane (lane l holds x[l % 32]); one v_permlane16_swap turns it into
// the pair (x[l % 16], x[16 + l % 16]), and every broadcast x[c] an FMA needs is the DPP operand row_newbcast:c of one of the two (lane c of
// the reader's own 16-lane row).  The first version of this kernel read the vectors back from LDS lines with ds_read_b128 -- 8 LDS clocks
// for 64 lanes whether or not they read the same address -- and ran exactly as fast as solve_body (209 vs 207 us): 185 of its 305 LDS
// clocks per world and iteration were such broadcasts.  Now LDS holds J only: a row read (7 x b128) and a column read (32 x b32) per
// iteration.  Global loads of the prologue are issued together, as soon as the world index is known (they were a chain of six round
// trips).  Worlds with more than 64 rows, elliptic cones, Newton and per-island solves stay with solve_body.
#pragma once
#include "solver.hpp"

// x(lane) + x(lane ^ 32), the same bits in both lanes
DEV float half_sum(float x) {
  const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);  // (x of the lower half) + (x of the upper half)
}
// x as held by the lanes of half H (compile-time), in every lane
template <int H>
DEV float half_of(float x) {
  const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(sw[H]);
}

#include <utility>

struct CgwLayout {
  int J, tile, vbuf, total;
};
template <int NV4>
__host__ __device__ inline CgwLayout cgw_layout(int njmax) {
  constexpr int NVR = 4 * NV4, JS = (NV4 & 1) ? NVR : NVR + 4;
  const int njp = min(((njmax + 15) / 16) * 16, 64);
  CgwLayout p;
  int o = 0;
  p.J = o; o += njp * JS + 32;  // (+ 32: the column reads of J^T f reach column 31 of the last row)
  p.tile = o; o += 2 * 4 * 32;  // double-buffered 4 x 32 pivot tile of the Gauss-Jordan inverse
  p.vbuf = o; o += 32;          // velocity line of the fused Euler step
  p.total = ((o + 3) / 4) * 4;
  return p;
}

// DPP operand row_newbcast:C -- lane C of the reader's own 16-lane row
template <int C>
DEV float rbc(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + C, 0xf, 0xf, true));
}
// x in dof layout (lane l holds x[l % 32]) -> a: x[l % 16], b: x[16 + l % 16] in every lane (v_permlane16_swap exchanges the odd rows of
// its first operand with the even rows of its second)
struct BV {
  float a, b;
};
DEV BV bcast_prep(float x) {
  const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return BV{__uint_as_float(sw[0]), __uint_as_float(sw[1])};
}
// s[C & 1] += row[OFF + C] * x[C] for the listed C, x[C] = lane C of the reader's 16-lane row of `xs`
template <int OFF, int N, int... C>
DEV void fma_rbc(float (&s)[2], const float (&row)[N], float xs, std::integer_sequence<int, C...>) {
  ((s[C & 1] += row[OFF + C] * rbc<C>(xs)), ...);
}
// J^T f part of one 16-row chunk: columns j and 16 + j (Jc points at row 0 of the chunk, column j), f = this lane's row force
template <int JS, bool TWO, int... R>
DEV void jtf_rbc(float (&s)[4], const float* Jc, float f, std::integer_sequence<int, R...>) {
  ((s[R & 1] += Jc[R * JS] * rbc<R>(f), s[2 + (R & 1)] += TWO ? Jc[R * JS + 16] * rbc<R>(f) : 0.0f), ...);
}

// Rows of M^-1, four pivots per step (invert_rows_b4 of solver.hpp), lane (i, h) holding the columns [16 h, 16 h + 16) of row i of the
// matrix padded with identity to 32 columns.  The pivot block's entries of a row (needed for F = A_iK P^-1) live in one half; the other
// half gets them by v_permlane32_swap.
template <int NV4>
DEV void invert_rows_split(const float (&mrow)[16], float (&s)[16], float* buf, int ld, int hf) {
  constexpr int NB = 4, HC = 16, NVP = 32;
#pragma unroll
  for (int c = 0; c < HC; ++c) s[c] = mrow[c];
#pragma unroll
  for (int kb = 0; kb < NV4; ++kb) {
    const int k = 4 * kb, hK = kb / NB, lb = kb % NB;  // (compile-time after unrolling)
    float* pb = buf + (kb & 1) * 4 * NVP;
    const int q = ld - k;  // 0..3 for the lanes of the pivot block
    const bool inb = q >= 0 && q < 4;
    if (inb) {
#pragma unroll
      for (int c4 = 0; c4 < NB; ++c4) *reinterpret_cast<float4*>(pb + q * NVP + hf * HC + 4 * c4) = make_float4(s[4 * c4], s[4 * c4 + 1], s[4 * c4 + 2], s[4 * c4 + 3]);
    }
    gsync();
    float P[4][4], Pi[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const float4 v = *reinterpret_cast<const float4*>(pb + p * NVP + k);
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
    float a4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) a4[e] = hK ? half_of<1>(s[4 * lb + e]) : half_of<0>(s[4 * lb + e]);
    float F[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const float out = a4[0] * Pi[0][p] + a4[1] * Pi[1][p] + a4[2] * Pi[2][p] + a4[3] * Pi[3][p];
      const float own = (q == p ? 1.0f : 0.0f) - (q == 0 ? Pi[0][p] : (q == 1 ? Pi[1][p] : (q == 2 ? Pi[2][p] : Pi[3][p])));
      F[p] = inb ? own : out;
    }
    const bool pivot_half = hf == hK;
#pragma unroll
    for (int c4 = 0; c4 < NB; ++c4) {
      const float* t = pb + hf * HC + 4 * c4;
      const float4 r0 = *reinterpret_cast<const float4*>(t), r1 = *reinterpret_cast<const float4*>(t + NVP),
                   r2 = *reinterpret_cast<const float4*>(t + 2 * NVP), r3 = *reinterpret_cast<const float4*>(t + 3 * NVP);
      const float u0 = s[4 * c4] - (F[0] * r0.x + F[1] * r1.x + F[2] * r2.x + F[3] * r3.x);
      const float u1 = s[4 * c4 + 1] - (F[0] * r0.y + F[1] * r1.y + F[2] * r2.y + F[3] * r3.y);
      const float u2 = s[4 * c4 + 2] - (F[0] * r0.z + F[1] * r1.z + F[2] * r2.z + F[3] * r3.z);
      const float u3 = s[4 * c4 + 3] - (F[0] * r0.w + F[1] * r1.w + F[2] * r2.w + F[3] * r3.w);
      if (c4 == lb) {  // the pivot block's own columns in the half that holds them: delta - F
        s[4 * c4] = pivot_half ? (q == 0 ? 1.0f : 0.0f) - F[0] : u0;
        s[4 * c4 + 1] = pivot_half ? (q == 1 ? 1.0f : 0.0f) - F[1] : u1;
        s[4 * c4 + 2] = pivot_half ? (q == 2 ? 1.0f : 0.0f) - F[2] : u2;
        s[4 * c4 + 3] = pivot_half ? (q == 3 ? 1.0f : 0.0f) - F[3] : u3;
      } else {
        s[4 * c4] = u0; s[4 * c4 + 1] = u1; s[4 * c4 + 2] = u2; s[4 * c4 + 3] = u3;
      }
    }
  }
}

template <int NV4>
DEV void solve_cgw_body(const MjhModel& m, const MjhData& d, float* smem, const Blk& b, int nefc_lo, int nefc_hi, int fuse_euler) {
  if ((int)threadIdx.x >= b.nthreads) return;
  constexpr int G = 64, NVR = 4 * NV4, JS = (NV4 & 1) ? NVR : NVR + 4, HC = 16, J4 = JS / 4;
  constexpr int NA = NVR < 16 ? NVR : 16, NBX = NVR - NA;  // columns served by the a / b half of a broadcast pair
  const int nv = m.nv, nC = m.nC, njmax = d.njmax, nvp = d.nv_pad;
  const CgwLayout lay = cgw_layout<NV4>(njmax);
  const int lig = threadIdx.x & 63, gib = threadIdx.x >> 6, ld = lig & 31, hf = lig >> 5, rho = lig >> 4, j16 = lig & 15;
  const int slot = b.w0 + gib;
  if (slot >= d.nworld) return;
  float* S = smem + (size_t)gib * lay.total;
  float *Jl = S + lay.J, *tile = S + lay.tile, *vbuf = S + lay.vbuf;
  const bool active = ld < nv;

  PhaseClock pc(5, lig);
  // ---- every global load of the prologue, issued together: the dense address table of M (model-wide, independent of the world), then --
  // as soon as the world index is back -- the row counts, this lane's half row of M, its dof scalars, its constraint row and its share of J
  int idx[HC];
  {
    const int nv4r = (nv + 3) >> 2;
    const int4* tab = reinterpret_cast<const int4*>(m.M_dense) + (size_t)(active ? ld : 0) * nv4r;
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      const int gb = hf * 4 + c4;
      const int4 t4 = gb < nv4r ? tab[gb] : make_int4(-1, -1, -1, -1);
      idx[4 * c4] = t4.x; idx[4 * c4 + 1] = t4.y; idx[4 * c4 + 2] = t4.z; idx[4 * c4 + 3] = t4.w;
    }
  }
  const int w = d.ws_order[slot];  // longest expected solve first (k_schedule_worlds)
  const size_t vo = (size_t)w * nv, eo = (size_t)w * njmax;
  const int nefc_raw = d.nefc[w], ne = d.ne[w], nf = d.nf[w];
  const float* Jg = d.efc_J + (size_t)w * d.njmax_pad * nvp;
  const bool flat = nvp == JS;  // same row stride in HBM and LDS: J travels as float4s through registers
  const int njp = min(((njmax + 15) / 16) * 16, 64);
  const int n4max = min(njp, d.njmax_pad) * J4;
  float4 jreg[J4];
#pragma unroll
  for (int k = 0; k < J4; ++k) {
    const int i = lig + G * k;
    jreg[k] = flat && i < n4max ? reinterpret_cast<const float4*>(Jg)[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
  float mrow[HC];
  {
    const float* Mg = d.M + (size_t)w * nC;
#pragma unroll
    for (int c = 0; c < HC; ++c) {
      const float v = Mg[idx[c] < 0 ? 0 : idx[c]];
      mrow[c] = active ? (idx[c] < 0 ? 0.0f : v) : (hf * HC + c == ld ? 1.0f : 0.0f);
    }
  }
  const float fs = active ? d.qfrc_smooth[vo + ld] : 0.0f;
  const float qwarm = active ? d.qacc_warmstart[vo + ld] : 0.0f;
  const bool inrow = lig < njmax;
  const float rD_raw = inrow ? d.efc_D[eo + lig] : 0.0f;
  const float aref = inrow ? d.efc_aref[eo + lig] : 0.0f;
  const float tolerance = bf(m.opt_tolerance, m.opt_tolerance_nb, w, 1)[0];
  const float ls_tolerance = bf(m.opt_ls_tolerance, m.opt_ls_tolerance_nb, w, 1)[0];
  const float meaninertia = bf(m.stat_meaninertia, m.stat_meaninertia_nb, w, 1)[0];

  const int nefc_all = min(nefc_raw, njmax);
  if (nefc_all <= nefc_lo || nefc_all > nefc_hi) return;  // (two-size dispatch: see solve_body)
  const int nefc = min(nefc_all, 64);
  const float* floss = d.efc_frictionloss + eo;
  const bool has_fl = nf > 0;
  // ---- J into LDS (rows past nefc as zeros: the chunked column reads of J^T f rely on it) -------------------------------------------
  if (flat) {
#pragma unroll
    for (int k = 0; k < J4; ++k) {
      const int i = lig + G * k;
      if (i < njp * J4) reinterpret_cast<float4*>(Jl)[i] = i < nefc * J4 ? jreg[k] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
  } else {
    for (int r = 0; r < njp; ++r)
      for (int c = lig; c < JS; c += G) Jl[r * JS + c] = (r < nefc && c < nvp) ? Jg[(size_t)r * nvp + c] : 0.0f;
  }
  pc.mark(0);
  const bool warm = !(m.disableflags & DSBL_WARMSTART);
  // dof i: sum_c A[i][c] x[c] from the two half rows; x comes as a broadcast pair
  auto mul_row = [&](const float (&row)[HC], const BV& x) __attribute__((always_inline)) {
    float s[2] = {0.0f, 0.0f};
    fma_rbc<0>(s, row, hf ? x.b : x.a, std::make_integer_sequence<int, HC>{});
    const float t = half_sum(s[0] + s[1]);
    return active ? t : 0.0f;
  };
  // ---- M^-1 (the CG preconditioner) and qacc_smooth = M^-1 qfrc_smooth with one step of iterative refinement ----------------
  float h[HC];
  invert_rows_split<NV4>(mrow, h, tile, ld, hf);
  float qs = mul_row(h, bcast_prep(fs));
  const float res = fs - mul_row(mrow, bcast_prep(qs));
  qs += mul_row(h, bcast_prep(active ? res : 0.0f));
  if (active && hf == 0) d.qacc_smooth[vo + ld] = qs;
  float q = active ? (nefc > 0 && warm ? qwarm : qs) : 0.0f;
  pc.mark(1);
  BV qb = bcast_prep(q);
  float Ma = mul_row(mrow, qb);

  if (nefc == 0) {  // unconstrained: qacc = qacc_smooth (solver.py:3684-3686)
    if (active && hf == 0) {
      d.qacc[vo + ld] = q;
      d.qfrc_constraint[vo + ld] = 0.0f;
      d.efc_Ma[vo + ld] = Ma;
    }
    if (lig == 0) d.solver_niter[w] = 0;
    if (fuse_euler) euler_advance<G>(m, d, w, lig, active && hf == 0, q, vbuf, q);
    return;
  }

  float rD[1], rja[1], rjv[1];
  int rkind[1];
  {
    const bool has = lig < nefc;
    rD[0] = has ? rD_raw : 0.0f;
    rkind[0] = !has ? 3 : (lig >= ne + nf ? 2 : (lig >= ne ? 1 : 0));  // 3: padding row
    rjv[0] = 0.0f;
  }
  gsync();
  auto j_dot = [&](const BV& x) __attribute__((always_inline)) {  // J[lig, :] . x
    float jr[NVR];
#pragma unroll
    for (int c4 = 0; c4 < NV4; ++c4) {
      const float4 j4 = *reinterpret_cast<const float4*>(Jl + lig * JS + 4 * c4);
      jr[4 * c4] = j4.x; jr[4 * c4 + 1] = j4.y; jr[4 * c4 + 2] = j4.z; jr[4 * c4 + 3] = j4.w;
    }
    float s[2] = {0.0f, 0.0f};
    fma_rbc<0>(s, jr, x.a, std::make_integer_sequence<int, NA>{});
    if (NBX > 0) fma_rbc<NA>(s, jr, x.b, std::make_integer_sequence<int, NBX>{});
    return s[0] + s[1];
  };
  // (row dots run in EVERY lane, whatever its row: a DPP operand read from a lane that sits out a branch is zero -- the first cut had the
  // dot inside `rkind != 3 ? ... : 0`, and rows 16..31 lost x[14], x[15] whenever nefc was 30)
  {
    const float jq = j_dot(qb);
    rja[0] = rkind[0] != 3 ? jq - aref : 0.0f;
  }

  pc.mark(2);
  const float scale = meaninertia * (float)nv;
  const float rscale = 1.0f / scale;
  const float own = hf == 0 ? 1.0f : 0.0f;  // dof scalars live in both halves: sums over dofs count the lower half only

  float grad_dot = 0.0f, search_dot = 0.0f;
  float g = 0.0f, Mg = 0.0f, pg = 0.0f, pMg = 0.0f, srch = 0.0f, qc = 0.0f;
  float cg5[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  int niter = 0;
  const int maxiter = m.iterations, ls_iterations = m.ls_iterations;
  int ovf = 0;
  float improvement = 0.0f;
  for (;;) {
    // ---- force of this lane's row (solver.py:1698-1822) ---------------------------------------------------------------------
    float force;
    {
      int state;
      row_force(rkind[0], rja[0], rD[0], has_fl, floss + lig, force, state);
    }
    // ---- qfrc_constraint = J^T force: lane (rho, j) sums columns j and 16 + j over the rows [16 rho, 16 rho + 16) ------------------
    {
      float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      if (16 * rho < nefc) jtf_rbc<JS, (NVR > 16)>(s, Jl + 16 * rho * JS + j16, force, std::make_integer_sequence<int, 16>{});
      // the four row chunks: lanes l, l ^ 16, l ^ 32, l ^ 48 -- every lane ends with the same bits
      auto quad_sum = [](float x) __attribute__((always_inline)) {
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
        return half_sum(__uint_as_float(sw[0]) + __uint_as_float(sw[1]));
      };
      const float c0 = quad_sum(s[0] + s[1]), c1 = NVR > 16 ? quad_sum(s[2] + s[3]) : 0.0f;
      qc = active ? ((rho & 1) ? c1 : c0) : 0.0f;
    }
    // ---- gradient and search direction (solver.py:3061-3220, 3283-3450) ----------------------------------------------------------
    g = active ? (Ma - fs - qc) : 0.0f;
    pc.mark(3);
    Mg = mul_row(h, bcast_prep(g));
    cg5[0] = own * g * g; cg5[1] = own * g * (Mg - pMg); cg5[2] = own * pg * pMg; cg5[3] = own * Mg * Mg; cg5[4] = own * Mg * srch;
    gsumg_n<G, 5>(cg5);
    grad_dot = cg5[0];
    pc.mark(4);
    if (niter == 0) {
      srch = -Mg;
      search_dot = cg5[3];
      pg = g;
      pMg = Mg;
    } else {
      const float imp = improvement * rscale, gradient = sqrtf(grad_dot) * rscale;
      const float beta = fmaxf(0.0f, cg5[1] * __builtin_amdgcn_rcpf(fmaxf(MJ_MINVAL, cg5[2])));  // Polak-Ribiere
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
    // ---- mv = M search, jv = J search --------------------------------------------------------------------------------------------
    const BV sb = bcast_prep(srch);
    const float mvi = mul_row(mrow, sb);
    {
      const float js = j_dot(sb);
      rjv[0] = rkind[0] != 3 ? js : 0.0f;
    }
    pc.mark(5);
    // ---- line search (solver.py:835-1347): one row per lane ------------------------------------------------------------------------
    const float g1 = own * srch * (Ma - fs);
    const float gtol = fmaxf(tolerance * ls_tolerance * sqrtf(search_dot) * scale, 1e-6f);
    float alpha = 0.0f;
    improvement = 0.0f;
    bool ls_converged = false;
    if (has_fl) line_search_rows<1, G, true>(rja, rjv, rD, rkind, floss + lig, g1, 0.5f * own * srch * mvi, fabsf(g1), gtol, ls_iterations, alpha, improvement, ls_converged);
    else line_search_rows<1, G, false>(rja, rjv, rD, rkind, floss + lig, g1, 0.5f * own * srch * mvi, fabsf(g1), gtol, ls_iterations, alpha, improvement, ls_converged);
    if (!ls_converged) ovf |= OVF_LS_ITERATIONS;
    pc.mark(6);
    q += alpha * srch;
    Ma += alpha * mvi;
    rja[0] += alpha * rjv[0];
    ++niter;
    pc.mark(7);
  }
  pc.mark(8);
  // ---- outputs ---------------------------------------------------------------------------------------------------------------------
  if (active && hf == 0) {
    d.qacc[vo + ld] = q;
    d.qfrc_constraint[vo + ld] = qc;
    d.efc_Ma[vo + ld] = Ma;
  }
  if (rkind[0] != 3) {  // force/state at the final iterate
    float force;
    int state;
    row_force(rkind[0], rja[0], rD[0], has_fl, floss + lig, force, state);
    d.efc_force[eo + lig] = force;
    d.efc_state[eo + lig] = state;
  }
  if (lig == 0) {
    d.solver_niter[w] = niter;
    if (ovf) atomicOr(d.overflow + w, ovf);
  }
  if (fuse_euler) euler_advance<G>(m, d, w, lig, active && hf == 0, q, vbuf, q);
  pc.mark(9);
}
