// pgs_big.hpp -- generic projected Gauss-Seidel: any number of dofs, pyramidal AND elliptic friction cones (round 3).
//
// The register / LDS resident PGS kernels (pgs.hpp) end at 64 dofs and know pyramidal rows only.  BASELINE configs[4] (aloha_clutter) asks
// for PGS on a 136-dof model with elliptic cones, which the reference cannot run at all (it has no PGS: types.py:502).  This kernel is the
// correctness path for those cases: one wavefront per world, the running acceleration and the row vectors in LDS, J and B = J M^-1 in
// HBM / L2 (Data.ws_pgsB), M^-1 through the sparse L'DL factor (block diagonal over kinematic trees, no fill-in).
//
// Algorithm: MuJoCo C's dual PGS (engine_solver.c mj_solPGS) as restated in float64 by oracle/mjref.c:solve_pgs, which this kernel is
// tested against.  Scalar rows: projected coordinate step (pgs.hpp).  Elliptic contact = block of `dim` rows updated together:
//   * at the apex (no normal force): exact minimisation along the steepest feasible ray of the cone, v = (1, -mu_j^2 res_j / |mu o res_f|);
//   * otherwise: exact minimisation along the current force direction (ray update: scales normal and friction together), then
//   * the friction forces at fixed normal force: QCQP  min 0.5 v'A_ff v + v'b_c  s.t.  sum_j (v_j / mu_j)^2 <= f_n^2  (Newton on the
//     multiplier, dense Cholesky of at most 5 x 5 in registers);
//   * a block step that would increase the dual cost (round-off) is rejected, as for scalar rows.
// The fixed point of these steps is the solution of the dual problem, which the oracle tests show equal to the Newton / CG solution of the
// primal elliptic problem (tests/test_pgs.py: qacc to 2e-6 for condim 3 / 4 / 6, impratio 1 and 10).
#pragma once
#include <type_traits>

#include "solver.hpp"

#define PGSB_MAXWAVES 8
#define PGSB_MAXTRACKS (4 * PGSB_MAXWAVES)
// ctl words: 0 state (1: nothing to solve) | 1 compact | 2 packed elements | 3 quarter tracks | 8.. list bounds [PGSB_MAXTRACKS + 1] |
// 48.. improvement partials [2][PGSB_MAXWAVES] | 64.. tree labels [64] | 128.. tree group / row counts [64]
#define PGSB_CTL 192
// sums over a track of TW lanes (64: the wavefront; 16: one DPP row), result in every lane of the track
template <int TW>
DEV float tsum(float v) {
  if (TW == 64) return gsumg<64>(v);
  v = dpp_add_f<0x111, 0xf, 0xf>(v);  // row_shr:1 .. 8: lane 15 of the row holds the row sum
  v = dpp_add_f<0x112, 0xf, 0xf>(v);
  v = dpp_add_f<0x114, 0xf, 0xf>(v);
  v = dpp_add_f<0x118, 0xf, 0xf>(v);
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + 15, 0xf, 0xf, true));  // row_newbcast:15
}
template <int TW, int N>
DEV void tsum_n(float (&v)[N]) {
  if (TW == 64) {
    gsumg_n<64, N>(v);
    return;
  }
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = dpp_add_f<0x111, 0xf, 0xf>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = dpp_add_f<0x112, 0xf, 0xf>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = dpp_add_f<0x114, 0xf, 0xf>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = dpp_add_f<0x118, 0xf, 0xf>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[i]), 0x150 + 15, 0xf, 0xf, true));
}
struct PgsBigLayout {
  int q, qs, tmp, force, aref, R, Ad, mu, info, tmask, blk, coff, rlist, ctl, xs, M, L, dinv, total;
};
__host__ __device__ inline PgsBigLayout pgs_big_layout(int nv, int nC, int njmax, bool ell) {
  PgsBigLayout p;
  int o = 0;
  p.q = o; o += nv;
  p.qs = o; o += nv;
  p.tmp = o; o += nv;
  p.force = o; o += njmax;
  p.aref = o; o += njmax;
  p.R = o; o += njmax;
  p.Ad = o; o += njmax;
  p.mu = o; o += ell ? njmax : 0;      // friction coefficient of a friction row of an elliptic contact
  p.tmask = o; o += 2 * njmax;          // kinematic trees a row touches (bit t of a 64-bit mask; trees >= 64 set every bit: full range)
  p.info = o; o += njmax;               // row kind: 0 equality, 1 friction loss, 2 limit / contact, 8 + dim: first row of an elliptic contact, 7: its other rows
  p.blk = o; o += ell ? 6 * njmax : 0;  // row r of an elliptic contact starting at r0: (A + R)[r][r0 .. r0 + 5]
  p.coff = o; o += njmax + 1;           // compact rows (see "compact rows" below): first element of row r in the packed arrays
  p.rlist = o; o += njmax;              // the sweep's visits (first rows of blocks) grouped by wavefront (see "islands")
  p.ctl = o; o += PGSB_CTL;             // hand-over from the set-up wavefront to the sweep: flags, island labels, list bounds, improvement sums
  p.xs = o; o += (65 * nv > 2 * njmax ? 65 * nv : 2 * njmax);  // 64 right-hand sides of the batched sparse solves, [dof][65]; later two row vectors
  p.M = o; o += nC;
  p.L = o; o += nC;
  p.dinv = o; o += nv;
  p.total = ((o + 3) / 4) * 4;
  return p;
}

// x = argmin 0.5 x'A x + x'b  s.t.  sum (x_i / mu_i)^2 <= r^2 in N = 2, 3 or 5 dimensions (condim 3 / 4 / 6); oracle/mjref.c qcqp.
// A, b, mu are read from the leading N x N / N entries of 5-wide arrays.  Specialised on N: a contact of condim 3 costs a 2 x 2
// factorisation per Newton step on the multiplier, not a padded 5 x 5 one (the padded version was 90 % of a sweep's time).
template <int N>
DEV void qcqpN(const float (&A)[5][5], const float (&b)[5], const float (&mu)[5], float r, float (&x)[5]) {
  float As[N][N], bs[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    bs[i] = b[i] * mu[i];
#pragma unroll
    for (int j = 0; j < N; ++j) As[i][j] = A[i][j] * mu[i] * mu[j];
  }
  float la = 0.0f, y[N];
#pragma unroll
  for (int i = 0; i < N; ++i) y[i] = 0.0f;
  bool active = false;
  for (int it = 0; it < 20; ++it) {
    float L[N][N];
#pragma unroll
    for (int j = 0; j < N; ++j) {  // Cholesky of As + la I (lower)
      float s = As[j][j] + la;
#pragma unroll
      for (int k = 0; k < N; ++k)
        if (k < j) s -= L[j][k] * L[j][k];
      const float l = sqrtf(fmaxf(s, MJ_MINVAL));
      L[j][j] = l;
#pragma unroll
      for (int i = 0; i < N; ++i)
        if (i > j) {
          float t = As[i][j];
#pragma unroll
          for (int k = 0; k < N; ++k)
            if (k < j) t -= L[i][k] * L[j][k];
          L[i][j] = t / l;
        }
    }
    auto solve = [&](const float (&rhs)[N], float (&out)[N]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        float t = rhs[i];
#pragma unroll
        for (int k = 0; k < N; ++k)
          if (k < i) t -= L[i][k] * out[k];
        out[i] = t / L[i][i];
      }
#pragma unroll
      for (int i = N - 1; i >= 0; --i) {
        float t = out[i];
#pragma unroll
        for (int k = 0; k < N; ++k)
          if (k > i) t -= L[k][i] * out[k];
        out[i] = t / L[i][i];
      }
    };
    float nb[N];
#pragma unroll
    for (int i = 0; i < N; ++i) nb[i] = -bs[i];
    solve(nb, y);
    float val = -r * r;
#pragma unroll
    for (int i = 0; i < N; ++i) val += y[i] * y[i];
    if (val < 1e-7f * r * r + 1e-20f) break;  // inside (or on) the ball, to float32 resolution
    active = true;
    float t[N];
    solve(y, t);
    float deriv = 0.0f;
#pragma unroll
    for (int i = 0; i < N; ++i) deriv -= 2.0f * y[i] * t[i];
    const float delta = -val / deriv;
    if (!(delta > 1e-7f * (la + 1e-20f))) break;
    la += delta;
  }
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < N; ++i) s += y[i] * y[i];
  const float sc = (active && s > r * r && s > 0.0f) ? r / sqrtf(s) : 1.0f;  // round-off: land on the boundary
#pragma unroll
  for (int i = 0; i < 5; ++i) x[i] = 0.0f;
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] = y[i] * mu[i] * sc;
}
DEV void qcqp5(int n, const float (&A)[5][5], const float (&b)[5], const float (&mu)[5], float r, float (&x)[5]) {
  if (n == 2) qcqpN<2>(A, b, mu, r, x);
  else if (n == 3) qcqpN<3>(A, b, mu, r, x);
  else qcqpN<5>(A, b, mu, r, x);
}

template <int G>
DEV void pgs_big_body(const MjhModel& m, const MjhData& d, float* smem, int w) {
  static_assert(G == 64, "wavefront-wide row updates");
  const int nv = m.nv, nC = m.nC, njmax = d.njmax, nvp = d.nv_pad;
  const bool ell = m.cone == CONE_ELLIPTIC && d.nmaxpyramid > 1;
  const PgsBigLayout lay = pgs_big_layout(nv, nC, njmax, ell);
  int* shi = reinterpret_cast<int*>(smem);
  const MStruct ms = load_mstruct<G>(m, shi, blockDim.x);  // (whole workgroup; ends with a barrier)
  const int lig = threadIdx.x & (G - 1);
  // One world per workgroup of up to PGSB_MAXWAVES wavefronts (round 3).  Wavefront 0 sets the problem up; the sweeps then run on all
  // wavefronts: rows of different constraint islands do not interact (A is block diagonal over islands), so Gauss-Seidel over the rows of
  // one island group is independent of the other groups -- each wavefront sweeps the rows of its own group of islands in row order, which
  // leaves every island's iterates exactly those of the sequential sweep.  The convergence test stays the model-wide one (improvement summed
  // over the groups after every sweep).
  const int wv = threadIdx.x >> 6, nwv = min((int)(blockDim.x >> 6), PGSB_MAXWAVES);
  float* S = smem + mstruct_ints(nv, nC);
  float *q = S + lay.q, *qs = S + lay.qs, *tmp = S + lay.tmp, *force = S + lay.force, *aref = S + lay.aref, *Rr = S + lay.R, *Ad = S + lay.Ad,
        *rmu = S + lay.mu, *blk = S + lay.blk, *xs = S + lay.xs, *Ml = S + lay.M, *Ll = S + lay.L, *dinv = S + lay.dinv;
  int* info = reinterpret_cast<int*>(S + lay.info);
  int* coff = reinterpret_cast<int*>(S + lay.coff);
  int* rlist = reinterpret_cast<int*>(S + lay.rlist);
  int* ctl = reinterpret_cast<int*>(S + lay.ctl);
  int* ladr = ctl + 8;
  float* imp = S + lay.ctl + 48;
  unsigned* tmask = reinterpret_cast<unsigned*>(S + lay.tmask);
  const int ntree = m.ntree;
  const bool sparse_rows = ntree > 1 && ntree <= 64;  // rows touch one or two trees: the sweep visits those dof ranges only
  const int nefc = min(d.nefc[w], njmax), ne = d.ne[w], nf = d.nf[w], nl = d.nl[w];
  const size_t vo = (size_t)w * nv, eo = (size_t)w * njmax;
  const float* Jg = d.efc_J + (size_t)w * d.njmax_pad * nvp;
  float* Bg = d.ws_pgsB + (size_t)w * d.njmax_pad * nvp;

  if (wv == 0) [&]() __attribute__((always_inline)) {  // ---- set-up (wavefront 0; `return` leaves the set-up only) --------------------------
  if (lig == 0) ctl[0] = 1;
  // ---- M, sparse factor, qacc_smooth ------------------------------------------------------------------------------------------
  gcopy<G>(Ml, d.M + (size_t)w * nC, nC, lig);
  gcopy<G>(Ll, d.M + (size_t)w * nC, nC, lig);
  gcopy<G>(qs, d.qfrc_smooth + vo, nv, lig);
  gsync();
  factor_ld<G>(ms, Ll, dinv, nv, lig, &m);
  solve_ld<G>(m, ms, Ll, dinv, qs, nv, lig);
  gsync();
  for (int i = lig; i < nv; i += G) d.qacc_smooth[vo + i] = qs[i];
  if (nefc == 0) {
    for (int i = lig; i < nv; i += G) {
      d.qacc[vo + i] = qs[i];
      d.qfrc_constraint[vo + i] = 0.0f;
      d.efc_Ma[vo + i] = d.qfrc_smooth[vo + i];
    }
    if (lig == 0) d.solver_niter[w] = 0;
    return;
  }
  // ---- B = J M^-1, 64 rows per batch.  Global traffic is always lane = dof (consecutive addresses of one row); the sparse L'DL solve
  // runs lane = row on the LDS copy xs[dof * XS + row], XS = 65 (odd: both access directions are bank-conflict free).  (The first
  // version read and wrote J / B with lane = row: 64 cache lines per load instruction, 1e9 L2 transactions per step -- 100 ms.)
  constexpr int XS = G + 1;
  for (int r0 = 0; r0 < nefc; r0 += G) {
    const int nb = min(G, nefc - r0);
    for (int rr = 0; rr < G; ++rr)
      for (int c = lig; c < nv; c += G) xs[c * XS + rr] = rr < nb ? Jg[(size_t)(r0 + rr) * nvp + c] : 0.0f;
    gsync();
    const int r = r0 + lig;
    const bool has = r < nefc;
    if (has) {  // trees this row touches (from the J copy, before the solve overwrites it)
      unsigned m0 = 0u, m1 = 0u;
      if (sparse_rows) {
        for (int t = 0; t < ntree; ++t) {
          const int a0 = m.tree_dofadr[t], n0 = m.tree_dofnum[t];
          bool any = false;
          for (int c = a0; c < a0 + n0; ++c) any = any || xs[c * XS + lig] != 0.0f;
          if (any) {
            if (t < 32) m0 |= 1u << t;
            else m1 |= 1u << (t - 32);
          }
        }
      }
      tmask[2 * r] = m0;
      tmask[2 * r + 1] = m1;
    }
    for (int k = nv - 1; k >= 0; --k) {  // x <- L^-T x
      const int start = ms.rowadr[k], n = ms.rownnz[k];
      const float xk = xs[k * XS + lig];
      for (int a = 0; a < n - 1; ++a) xs[ms.colind[start + a] * XS + lig] -= Ll[start + a] * xk;
    }
    for (int k = 0; k < nv; ++k) xs[k * XS + lig] *= dinv[k];
    for (int k = 0; k < nv; ++k) {  // x <- L^-1 x (ancestors have smaller indices)
      const int start = ms.rowadr[k], n = ms.rownnz[k];
      float s = xs[k * XS + lig];
      for (int a = 0; a < n - 1; ++a) s -= Ll[start + a] * xs[ms.colind[start + a] * XS + lig];
      xs[k * XS + lig] = s;
    }
    gsync();
    for (int rr = 0; rr < nb; ++rr) {  // store the B rows (lane = dof) and (A + R)_rr = J_r . B_r + R_r
      float part = 0.0f;
      for (int c = lig; c < nvp; c += G) {
        const float bv = c < nv ? xs[c * XS + rr] : 0.0f;
        Bg[(size_t)(r0 + rr) * nvp + c] = bv;
        part += c < nv ? Jg[(size_t)(r0 + rr) * nvp + c] * bv : 0.0f;
      }
      const float sAR = gsumg<G>(part);
      if (lig == 0) {
        const int rw = r0 + rr;
        const float D = d.efc_D[eo + rw];
        aref[rw] = d.efc_aref[eo + rw];
        Rr[rw] = 1.0f / D;
        Ad[rw] = sAR + 1.0f / D;
        info[rw] = rw >= ne + nf ? 2 : (rw >= ne ? 1 : 0);
        if (ell) rmu[rw] = 1.0f;
      }
    }
    gsync();
  }
  __threadfence_block();
  gsync();
  // ---- elliptic contacts: row kinds and friction coefficients (lane = row) ------------------------------------------------------------
  if (ell) {
    for (int r = ne + nf + nl + lig; r < nefc; r += G) {
      const int cid = d.ws_efc_con[eo + r], c = cid >> 4, dimid = cid & 15;
      const float* cr = d.ws_contact + ((size_t)w * d.concap + c) * CON_STRIDE;
      const int* cri = reinterpret_cast<const int*>(cr);
      if (cri[24] > 1) {
        const int r0 = r - dimid, dim = min(cri[29], nefc - r0);
        info[r] = dimid == 0 ? 8 + dim : 7;
        rmu[r] = dimid == 0 ? 1.0f : cr[CON_FRICTION_WORD(dimid - 1)];
      }
    }
    gsync();
  }
  // ---- compact rows (round 3) ---------------------------------------------------------------------------------------------------------
  // A sweep visits the rows one after the other and every visit used to start with a round trip to L2 for the row of J and end with one
  // for the row of B -- 2 x nefc x sweeps dependent global loads per solve (clutter_synth: 50 k).  A row is non-zero on the dofs of the one or
  // two kinematic trees it touches only, so the non-zero parts of J and B (and their column indices) of all rows are packed once into the LDS
  // region the B build used for its right-hand sides: element l of row r sits at coff[r] + l and belongs to lane l.  The block matrices, the warm
  // start, the sweeps and the final products then run on LDS alone.  Falls back to the global rows when a row touches more than 64 dofs or the packed rows do not fit.
  bool compact = sparse_rows;
  int tot_c = 0, widest_c = 0;
  if (compact) {
    const int xs_words = (65 * nv > 2 * njmax ? 65 * nv : 2 * njmax);
    if (ell) {  // the rows of an elliptic contact are updated together: give them one column set (the union of their trees)
      for (int r0 = 0; r0 < nefc; r0 += G) {
        const int r = r0 + lig;
        unsigned m0 = 0u, m1 = 0u;
        int dim = 0;
        if (r < nefc && info[r] >= 8) {
          dim = info[r] - 8;
          for (int a = 0; a < dim; ++a) {
            m0 |= tmask[2 * (r + a)];
            m1 |= tmask[2 * (r + a) + 1];
          }
        }
        gsync();
        for (int a = 0; a < dim; ++a) {
          tmask[2 * (r + a)] = m0;
          tmask[2 * (r + a) + 1] = m1;
        }
        gsync();
      }
    }
    int tot = 0, widest = 0;
    for (int r0 = 0; r0 < nefc; r0 += G) {  // lane = row: its element count, then an exclusive scan over the batch
      const int r = r0 + lig;
      int n = 0;
      if (r < nefc) {
        unsigned long long mk = (unsigned long long)tmask[2 * r] | ((unsigned long long)tmask[2 * r + 1] << 32);
        while (mk) {
          n += m.tree_dofnum[__builtin_ctzll(mk)];
          mk &= mk - 1;
        }
      }
      int inc = n;
#pragma unroll
      for (int off = 1; off < G; off <<= 1) {
        const int t = __shfl_up(inc, off, G);
        if (lig >= off) inc += t;
      }
      if (r < nefc) coff[r] = tot + inc - n;
      tot += __shfl(inc, G - 1, G);
      widest = max(widest, n);
    }
    widest = max(widest, __shfl_xor(widest, 32, G));
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) widest = max(widest, __shfl_xor(widest, off, G));
    if (lig == 0) coff[nefc] = tot;
    compact = widest <= G && 3 * tot <= xs_words - 2 * njmax;  // (the first 2 njmax words stay the warm start's row vectors)
    tot_c = tot;
    widest_c = widest;
    gsync();
    if (compact) {
      float *Jc = xs + 2 * njmax, *Bc = Jc + tot;
      int* colc = reinterpret_cast<int*>(Bc + tot);
#pragma unroll 4
      for (int r = 0; r < nefc; ++r) {
        const int o = coff[r], n = coff[r + 1] - o;
        unsigned long long mk = (unsigned long long)tmask[2 * r] | ((unsigned long long)tmask[2 * r + 1] << 32);
        int base = 0, col = -1;
        while (mk) {
          const int t = __builtin_ctzll(mk);
          mk &= mk - 1;
          const int a0 = m.tree_dofadr[t], n0 = m.tree_dofnum[t];
          if (lig >= base && lig < base + n0) col = a0 + lig - base;
          base += n0;
        }
        if (lig < n) {
          Jc[o + lig] = Jg[(size_t)r * nvp + col];
          Bc[o + lig] = Bg[(size_t)r * nvp + col];
          colc[o + lig] = col;
        }
      }
      gsync();
    }
  }

  const float *Jcs = xs + 2 * njmax, *Bcs = Jcs + tot_c;  // the packed rows, for the rest of the set-up
  const int* colcs = reinterpret_cast<const int*>(Bcs + tot_c);
  // dot of two rows of J / B over the dofs of the trees row r touches (lane = dof: partial sums, to be reduced over the wavefront)
  auto row_dots6 = [&](int r, int r0b, int dim, float (&part)[6]) __attribute__((always_inline)) {
#pragma unroll
    for (int bq = 0; bq < 6; ++bq) part[bq] = 0.0f;
    unsigned long long mk = sparse_rows ? ((unsigned long long)tmask[2 * r] | ((unsigned long long)tmask[2 * r + 1] << 32)) : 1ull;
    while (mk) {
      const int t = __builtin_ctzll(mk);
      mk &= mk - 1;
      const int a0 = sparse_rows ? m.tree_dofadr[t] : 0, n0 = sparse_rows ? m.tree_dofnum[t] : nv;
      for (int c = a0 + lig; c < a0 + n0; c += G) {
        const float j = Jg[(size_t)r * nvp + c];
#pragma unroll
        for (int bq = 0; bq < 6; ++bq)
          if (bq < dim) part[bq] += j * Bg[(size_t)(r0b + bq) * nvp + c];
      }
    }
  };
  // ---- elliptic contacts: row kinds, friction coefficients, the dim x dim blocks of A + R -----------------------------------------
  if (ell) {
    for (int r = ne + nf + nl; r < nefc; ++r) {  // (rows in turn, lanes over the row's non-zeros)
      if (info[r] >= 7) {
        int r0 = r;
        while (info[r0] == 7) --r0;
        const int dimid = r - r0, dim = info[r0] - 8;
        float part[6];
        if (compact) {  // (the rows of a contact share one column set: element l of every row is the same dof)
          const int o = coff[r], n = coff[r + 1] - o;
          const float j = lig < n ? Jcs[o + lig] : 0.0f;
#pragma unroll
          for (int bq = 0; bq < 6; ++bq) part[bq] = (bq < dim && lig < n) ? j * Bcs[coff[r0 + bq] + lig] : 0.0f;
        } else {
          row_dots6(r, r0, dim, part);
        }
        gsumg_n<G, 6>(part);
        if (lig < 6) blk[6 * r + lig] = (lig < dim ? (lig == 0 ? part[0] : lig == 1 ? part[1] : lig == 2 ? part[2] : lig == 3 ? part[3] : lig == 4 ? part[4] : part[5]) : 0.0f) + (lig == dimid ? Rr[r] : 0.0f);
      }
    }
    gsync();
  }
  // ---- warm start: primal forces at qacc_warmstart, kept if their dual cost is negative (engine_forward.c warmstart) ----------------
  const bool warm = !(m.disableflags & DSBL_WARMSTART);
  for (int i = lig; i < nv; i += G) tmp[i] = d.qacc_warmstart[vo + i];
  gsync();
  float cpart = 0.0f;
  for (int r = 0; r < nefc; ++r) {  // Jaref at the warm-start point and b_r, rows in turn (lane = dof of the row's trees)
    float jw = 0.0f, jb = 0.0f;
    unsigned long long mk = sparse_rows ? ((unsigned long long)tmask[2 * r] | ((unsigned long long)tmask[2 * r + 1] << 32)) : 1ull;
    if (compact) {
      const int o = coff[r];
      if (lig < coff[r + 1] - o) {
        const float j = Jcs[o + lig];
        const int c = colcs[o + lig];
        jw = j * tmp[c];
        jb = j * qs[c];
      }
      mk = 0ull;
    }
    while (mk) {
      const int t = __builtin_ctzll(mk);
      mk &= mk - 1;
      const int a0 = sparse_rows ? m.tree_dofadr[t] : 0, n0 = sparse_rows ? m.tree_dofnum[t] : nv;
      for (int c = a0 + lig; c < a0 + n0; c += G) {
        const float j = Jg[(size_t)r * nvp + c];
        jw += j * tmp[c];
        jb += j * qs[c];
      }
    }
    float two[2] = {jw, jb};
    gsumg_n<G, 2>(two);
    if (lig == 0) {
      xs[r] = two[0] - aref[r];          // (xs is free now: reused as two row vectors)
      xs[njmax + r] = two[1] - aref[r];  // b_r
    }
  }
  gsync();
  for (int r = lig; r < nefc; r += G) {
    float f = 0.0f;
    if (warm) {
      const int k = info[r];
      if (k <= 2) {
        int st;
        row_force(k, xs[r], 1.0f / Rr[r], nf > 0, d.efc_frictionloss + eo + r, f, st);
      } else {  // a row of an elliptic contact: the primal zones (solver.py:455-472), decided from the contact's rows together
        int r0 = r;
        while (info[r0] == 7) --r0;
        const int dim = info[r0] - 8;
        const float mu = d.ws_contact[((size_t)w * d.concap + (d.ws_efc_con[eo + r0] >> 4)) * CON_STRIDE + 14] * bf(m.opt_impratio_invsqrt, m.opt_impratio_invsqrt_nb, w, 1)[0];
        float tt = 0.0f;
        for (int j = 1; j < dim; ++j) tt += (xs[r0 + j] * rmu[r0 + j]) * (xs[r0 + j] * rmu[r0 + j]);
        const float N = xs[r0] * mu, T = tt <= 0.0f ? 0.0f : sqrtf(tt);
        const int zone = ell_zone(mu, N, T);
        if (zone == ST_QUADRATIC) f = -xs[r] / Rr[r];
        else if (zone == ST_CONE) {
          const float fnm = -safe_div(1.0f / Rr[r0], mu * mu * (1.0f + mu * mu)) * (N - mu * T) * mu;
          f = r == r0 ? fnm : -safe_div(fnm, T) * (xs[r] * rmu[r] * rmu[r]);
        }
      }
    }
    force[r] = f;
    cpart += f * (xs[njmax + r] + 0.5f * Rr[r] * f);
  }
  gsync();
  float ypart = 0.0f;
  if (compact) {  // rows in turn, every lane adds its element into z = B' f (tmp) and y = J' f (q: free until the line after the cost)
    for (int c = lig; c < nv; c += G) tmp[c] = q[c] = 0.0f;
    gsync();
    if (warm)
      for (int r = 0; r < nefc; ++r) {
        const int o = coff[r];
        const float f = force[r];
        if (lig < coff[r + 1] - o && f != 0.0f) {
          const int c = colcs[o + lig];
          tmp[c] += f * Bcs[o + lig];
          q[c] += f * Jcs[o + lig];
        }
        gsync();  // (the next row may reach the same dofs from other lanes)
      }
    for (int c = lig; c < nv; c += G) ypart += 0.5f * q[c] * tmp[c];
  } else
  for (int c = lig; c < nv; c += G) {  // q - qacc_smooth = B' f; the A part of the cost is 0.5 (J' f) . (B' f)
    float z = 0.0f, y = 0.0f;
    if (warm)
      for (int r = 0; r < nefc; ++r) {
        const float f = force[r];
        z += f * Bg[(size_t)r * nvp + c];
        y += f * Jg[(size_t)r * nvp + c];
      }
    tmp[c] = z;
    ypart += 0.5f * y * z;
  }
  const float cost = gsumg<G>(cpart + ypart);
  const bool keep = warm && !(cost > 0.0f);
  gsync();
  for (int c = lig; c < nv; c += G) q[c] = qs[c] + (keep ? tmp[c] : 0.0f);
  if (!keep)
    for (int r = lig; r < nefc; r += G) force[r] = 0.0f;
  gsync();

  // ---- islands: the sweep's visits (scalar rows, first rows of elliptic contacts) grouped by wavefront ------------------------------------
  // Trees joined by a row belong to one island (min-label propagation over the rows' tree masks, as k_tree_rows does for CG / Newton); islands
  // go to the wavefront with the fewest rows so far; every wavefront gets the visits of its islands in row order.  Without tree masks (one
  // tree, or more than 64) and without packed rows everything stays on wavefront 0.
  int* lab = ctl + 64;
  int* grp = ctl + 128;
  const bool split = compact && nwv > 1;
  // tracks: a wavefront sweeps one island at a time with all 64 lanes, or -- when no row has more than 16 non-zeros -- four islands at a
  // time, one per 16-lane DPP row (a row of J / B occupies one lane per non-zero; the small dense block problem is solved redundantly by
  // the lanes of the track)
  const bool quarter = split && widest_c <= 16;
  const int ntrk = split ? nwv * (quarter ? 4 : 1) : 1;
  if (split) {
    if (lig < ntree) lab[lig] = lig;
    gsync();
    for (int pass = 0; pass < ntree; ++pass) {
      bool changed = false;
      for (int r = lig; r < nefc; r += G) {
        unsigned long long mk = (unsigned long long)tmask[2 * r] | ((unsigned long long)tmask[2 * r + 1] << 32);
        if (mk & (mk - 1)) {  // two trees or more
          int lo = ntree;
          for (unsigned long long t = mk; t; t &= t - 1) lo = min(lo, lab[__builtin_ctzll(t)]);
          for (unsigned long long t = mk; t; t &= t - 1) {
            const int u = __builtin_ctzll(t);
            if (lab[u] > lo) {
              atomicMin(&lab[u], lo);
              changed = true;
            }
          }
        }
      }
      gsync();
      if (lig < ntree) lab[lig] = lab[lab[lig]];
      gsync();
      if (!__any(changed)) break;
    }
    if (lig < ntree) grp[lig] = 0;
    gsync();
    for (int r = lig; r < nefc; r += G) {  // rows per island (at its root tree)
      const unsigned long long mk = (unsigned long long)tmask[2 * r] | ((unsigned long long)tmask[2 * r + 1] << 32);
      if (mk) atomicAdd(&grp[lab[__builtin_ctzll(mk)]], 1);
    }
    gsync();
    // islands in tree order, each to the least loaded track: lane k keeps the load of track k
    int myload = 0;
    for (int t = 0; t < ntree; ++t) {
      if (lab[t] != t) continue;  // (wave-uniform: LDS value)
      int mn = lig < ntrk ? myload : 0x7fffffff;
      const int cand = mn;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) mn = min(mn, __shfl_xor(mn, off, G));
      const int best = __builtin_ctzll(__ballot(cand == mn));
      if (lig == best) myload += grp[t];
      gsync();
      if (lig == 0) grp[t] = -1 - best;  // (negative: a track, no longer a count)
      gsync();
    }
    if (lig < ntree) lab[lig] = -1 - grp[lab[lig]];  // tree -> track of its island (each lane rewrites its own entry only; roots' grp is final)
    gsync();
  }
  {
    int adr = 0;
    for (int k = 0; k < ntrk; ++k) {
      if (lig == 0) ladr[k] = adr;
      for (int r0 = 0; r0 < nefc; r0 += G) {
        const int r = r0 + lig;
        bool mine = r < nefc && info[r] != 7;
        if (mine) {
          int g = 0;
          if (split) {
            const unsigned long long mk = (unsigned long long)tmask[2 * r] | ((unsigned long long)tmask[2 * r + 1] << 32);
            g = mk ? lab[__builtin_ctzll(mk)] : 0;
          }
          mine = g == k;
        }
        const unsigned long long bits = __ballot(mine);
        if (mine) rlist[adr + __popcll(bits & ((1ull << lig) - 1ull))] = r;
        adr += __popcll(bits);
      }
    }
    if (lig == 0) {
      for (int k = ntrk; k <= PGSB_MAXTRACKS; ++k) ladr[k] = adr;
      ctl[1] = compact ? 1 : 0;
      ctl[2] = tot_c;
      ctl[3] = quarter ? 1 : 0;
      ctl[0] = 0;
    }
  }
  }();  // ---- end of the set-up ----------------------------------------------------------------------------------------------------------
  __syncthreads();
  if (ctl[0]) return;  // no active row: wavefront 0 wrote the unconstrained solution
  const bool compact = ctl[1] != 0;
  float *Jc = xs + 2 * njmax, *Bc = Jc + ctl[2];
  int* colc = reinterpret_cast<int*>(Bc + ctl[2]);

  // ---- sweeps ---------------------------------------------------------------------------------------------------------------------
  const float tolerance = bf(m.opt_tolerance, m.opt_tolerance_nb, w, 1)[0];
  const float rscale = 1.0f / (bf(m.stat_meaninertia, m.stat_meaninertia_nb, w, 1)[0] * (float)max(nv, 1));
  const int maxiter = m.iterations;
  int niter = 0;
  // J_r . q over the lanes' dofs (partial; reduce with gsumg / gsumg_n)
  // (M^-1 is block diagonal over kinematic trees: row r of J and of B are zero outside the trees the row touches)
  int l = lig;  // lane within the track (set by run_sweeps)
  auto jq_part = [&](int r) __attribute__((always_inline)) {
    float s = 0.0f;
    if (compact) {
      const int o = coff[r];
      if (l < coff[r + 1] - o) s = Jc[o + l] * q[colc[o + l]];
    } else if (sparse_rows) {
      unsigned long long mk = (unsigned long long)tmask[2 * r] | ((unsigned long long)tmask[2 * r + 1] << 32);
      while (mk) {
        const int t = __builtin_ctzll(mk);
        mk &= mk - 1;
        const int a0 = m.tree_dofadr[t], n0 = m.tree_dofnum[t];
        for (int c = a0 + lig; c < a0 + n0; c += G) s += Jg[(size_t)r * nvp + c] * q[c];
      }
    } else {
      for (int c = lig; c < nv; c += G) s += Jg[(size_t)r * nvp + c] * q[c];
    }
    return s;
  };
  auto q_add = [&](int r, float delta) __attribute__((always_inline)) {
    if (compact) {
      const int o = coff[r];
      if (l < coff[r + 1] - o) q[colc[o + l]] += delta * Bc[o + l];
    } else if (sparse_rows) {
      unsigned long long mk = (unsigned long long)tmask[2 * r] | ((unsigned long long)tmask[2 * r + 1] << 32);
      while (mk) {
        const int t = __builtin_ctzll(mk);
        mk &= mk - 1;
        const int a0 = m.tree_dofadr[t], n0 = m.tree_dofnum[t];
        for (int c = a0 + lig; c < a0 + n0; c += G) q[c] += delta * Bg[(size_t)r * nvp + c];
      }
    } else {
      for (int c = lig; c < nv; c += G) q[c] += delta * Bg[(size_t)r * nvp + c];
    }
  };
  auto run_sweeps = [&](auto TWc) __attribute__((always_inline)) {
  constexpr int TW = decltype(TWc)::value, NTW = 64 / TW;
  l = lig % TW;
  const int trk = wv * NTW + lig / TW;
  const int v0 = ladr[trk], nvis = ladr[trk + 1] - v0;
  int nmax = nvis;  // the tracks of a wavefront visit in lockstep
  if (NTW > 1) {
    nmax = max(nmax, __shfl_xor(nmax, 16, 64));
    nmax = max(nmax, __shfl_xor(nmax, 32, 64));
  }
  while (niter < maxiter) {
    float improvement = 0.0f;
    for (int st = 0; st < nmax; ++st) {
      const bool on = st < nvis;
      const int i = on ? rlist[v0 + st] : 0;
      const int k = on ? info[i] : -1;
      if (k >= 0 && k <= 2) {
        const float fold = force[i];
        const float res = tsum<TW>(jq_part(i)) - aref[i] + Rr[i] * fold;
        float fn = fold - res / Ad[i];
        if (k == 2) fn = fmaxf(fn, 0.0f);
        else if (k == 1) {
          const float fl = d.efc_frictionloss[eo + i];
          fn = fminf(fmaxf(fn, -fl), fl);
        }
        float delta = fn - fold;
        float change = delta * (0.5f * delta * Ad[i] + res);
        if (change > 1e-10f) {
          delta = 0.0f;
          change = 0.0f;
        }
        improvement -= change;
        if (delta != 0.0f) q_add(i, delta);
        if (l == 0) force[i] = fold + delta;
      } else if (k >= 8) {
      // ---- elliptic contact block (all lanes of the track compute the small dense problem redundantly) ----------------------------
      const int dim = k - 8;
      float res[6], fold[6], A[6][6], mu[5], fnew[6];
      {
        float part[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) part[a] = a < dim ? jq_part(i + a) : 0.0f;
        tsum_n<TW, 6>(part);
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          fold[a] = a < dim ? force[i + a] : 0.0f;
          res[a] = a < dim ? part[a] - aref[i + a] + Rr[i + a] * fold[a] : 0.0f;
#pragma unroll
          for (int bq = 0; bq < 6; ++bq) A[a][bq] = (a < dim && bq < dim) ? blk[6 * (i + a) + bq] : (a == bq ? 1.0f : 0.0f);
          if (a > 0) mu[a - 1] = a < dim ? rmu[i + a] : 1.0f;
        }
      }
      float fn;
      if (fold[0] < MJ_MINVAL) {  // apex: steepest feasible ray of the cone
        float sres = 0.0f, v[6], vAv = 0.0f;
#pragma unroll
        for (int a = 1; a < 6; ++a) sres += a < dim ? mu[a - 1] * mu[a - 1] * res[a] * res[a] : 0.0f;
        sres = sqrtf(sres);
        v[0] = 1.0f;
#pragma unroll
        for (int a = 1; a < 6; ++a) v[a] = (a < dim && sres > MJ_MINVAL) ? -mu[a - 1] * mu[a - 1] * res[a] / sres : 0.0f;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int bq = 0; bq < 6; ++bq) vAv += (a < dim && bq < dim) ? v[a] * A[a][bq] * v[bq] : 0.0f;
        const float slope = res[0] - sres, t = (slope < 0.0f && vAv >= MJ_MINVAL) ? -slope / vAv : 0.0f;
#pragma unroll
        for (int a = 0; a < 6; ++a) fnew[a] = a < dim ? t * v[a] : 0.0f;
        fn = fnew[0];
      } else {  // ray update
        float vAv = 0.0f, vr = 0.0f;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          float Av = 0.0f;
#pragma unroll
          for (int bq = 0; bq < 6; ++bq) Av += (a < dim && bq < dim) ? A[a][bq] * fold[bq] : 0.0f;
          vAv += fold[a] * Av;
          vr += fold[a] * res[a];
        }
        const float x = fmaxf(vAv >= MJ_MINVAL ? -vr / vAv : 0.0f, -1.0f);
#pragma unroll
        for (int a = 0; a < 6; ++a) fnew[a] = fold[a] + x * fold[a];
        fn = fnew[0];
      }
      if (fn < MJ_MINVAL) {
#pragma unroll
        for (int a = 1; a < 6; ++a) fnew[a] = 0.0f;
      } else {  // friction at fixed normal force
        float Ac[5][5], bc[5], vv[5];
#pragma unroll
        for (int a = 1; a < 6; ++a) {
          float bb = res[a] + A[a][0] * (fn - fold[0]);
#pragma unroll
          for (int c2 = 1; c2 < 6; ++c2) {
            Ac[a - 1][c2 - 1] = A[a][c2];
            bb -= (a < dim && c2 < dim) ? A[a][c2] * fold[c2] : 0.0f;
          }
          bc[a - 1] = a < dim ? bb : 0.0f;
        }
        qcqp5(dim - 1, Ac, bc, mu, fn, vv);
#pragma unroll
        for (int a = 1; a < 6; ++a) fnew[a] = a < dim ? vv[a - 1] : 0.0f;
      }
      float change = 0.0f, dl[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) dl[a] = a < dim ? fnew[a] - fold[a] : 0.0f;
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        float s2 = 0.0f;
#pragma unroll
        for (int bq = 0; bq < 6; ++bq) s2 += (a < dim && bq < dim) ? A[a][bq] * dl[bq] : 0.0f;
        change += dl[a] * (0.5f * s2 + res[a]);
      }
      const bool bad = change > 1e-10f;
      if (!bad) {
        improvement -= change;
        if (compact) {  // (the rows of a contact touch the same trees: same columns, one read-modify-write of q per lane)
          const int o = coff[i], n = coff[i + 1] - o;
          if (l < n) {
            float dq = 0.0f;
#pragma unroll
            for (int a = 0; a < 6; ++a)
              if (a < dim) dq += dl[a] * Bc[o + a * n + l];
            q[colc[o + l]] += dq;
          }
        } else
        for (int a = 0; a < dim; ++a)
          if (dl[a] != 0.0f) q_add(i + a, dl[a]);  // (the rows of a contact touch the same trees: disjoint lanes per dof, no race)
        if (l < dim) force[i + l] = fnew[0] * (l == 0) + fnew[1] * (l == 1) + fnew[2] * (l == 2) + fnew[3] * (l == 3) + fnew[4] * (l == 4) + fnew[5] * (l == 5);
      }
      }
      gsync();
    }
    // model-wide convergence test: the groups' improvements through LDS (double-buffered by sweep parity: one barrier per sweep)
    if (NTW > 1) {
      improvement += __shfl_xor(improvement, 16, 64);
      improvement += __shfl_xor(improvement, 32, 64);
    }
    if (lig == 0) imp[(niter & 1) * PGSB_MAXWAVES + wv] = improvement;
    __syncthreads();
    float total = 0.0f;
    for (int k = 0; k < nwv; ++k) total += imp[(niter & 1) * PGSB_MAXWAVES + k];
    ++niter;
    if (total * rscale < tolerance) break;
  }
  };
  if (ctl[3]) run_sweeps(std::integral_constant<int, 16>{});
  else run_sweeps(std::integral_constant<int, 64>{});
  __syncthreads();
  if (wv != 0) return;

  // ---- finish: qfrc_constraint = J' f, qacc = qacc_smooth + B' f, dual states ---------------------------------------------------------
  if (compact) {  // J' f and B' f from the packed rows (rows in turn; tmp and q are free now)
    for (int c = lig; c < nv; c += G) tmp[c] = q[c] = 0.0f;
    gsync();
    for (int r = 0; r < nefc; ++r) {
      const int o = coff[r];
      const float f = force[r];
      if (lig < coff[r + 1] - o && f != 0.0f) {
        const int c = colc[o + lig];
        tmp[c] += f * Bc[o + lig];
        q[c] += f * Jc[o + lig];
      }
      gsync();
    }
    for (int c = lig; c < nv; c += G) {
      d.qacc[vo + c] = qs[c] + tmp[c];
      d.qfrc_constraint[vo + c] = q[c];
      d.efc_Ma[vo + c] = d.qfrc_smooth[vo + c] + q[c];
    }
  } else
  for (int c = lig; c < nv; c += G) {
    float qc = 0.0f, dq = 0.0f;
    for (int r = 0; r < nefc; ++r) {
      const float f = force[r];
      qc += f * Jg[(size_t)r * nvp + c];
      dq += f * Bg[(size_t)r * nvp + c];
    }
    d.qacc[vo + c] = qs[c] + dq;
    d.qfrc_constraint[vo + c] = qc;
    d.efc_Ma[vo + c] = d.qfrc_smooth[vo + c] + qc;
  }
  for (int r = lig; r < nefc; r += G) {
    const float f = force[r];
    const int k = info[r];
    int state;
    if (k == 0) state = ST_QUADRATIC;
    else if (k == 1) {
      const float fl = d.efc_frictionloss[eo + r];
      state = f <= -fl ? ST_LINEARPOS : (f >= fl ? ST_LINEARNEG : ST_QUADRATIC);
    } else if (k == 2) state = f <= 0.0f ? ST_SATISFIED : ST_QUADRATIC;
    else {  // elliptic contact: no normal force -> satisfied; friction on the cone's surface -> cone; strictly inside -> quadratic
      int r0 = r;
      while (info[r0] == 7) --r0;
      const int dim = info[r0] - 8;
      float tt = 0.0f;
      for (int a = 1; a < dim; ++a) tt += (force[r0 + a] / rmu[r0 + a]) * (force[r0 + a] / rmu[r0 + a]);
      state = force[r0] <= 0.0f ? ST_SATISFIED : (tt >= force[r0] * force[r0] * (1.0f - 1e-5f) ? ST_CONE : ST_QUADRATIC);
    }
    d.efc_force[eo + r] = f;
    d.efc_state[eo + r] = state;
  }
  if (lig == 0) d.solver_niter[w] = niter;
}
