// solver_big.hpp -- CG / Newton constraint solver for models with more than 64 dofs (three_humanoids: nv = 81).
//
// Same algorithm as solver.hpp (reference solver.py:3671-3743 solve, 3525-3620 iteration, 835-1347 line search,
// 1698-1822 constraint update, 3061-3220 gradient, 3283-3450 CG) but without the "lane i owns row i" register layout,
// which ends at 64 dofs: one wavefront per world, every vector in LDS, every loop strided over the 64 lanes.
//   * J stays in HBM/L2 (njmax x nv is 64 KB at nv 81, njmax 192: LDS-resident it would leave one wave per CU); it is
//     read twice per iteration: row-per-lane for J v, column-per-lane (coalesced) for J' f.
//   * CG applies M^-1 through the sparse L'DL factor (factor_ld / solve_ld, smooth.hpp: no fill-in for tree-structured M),
//     so independent trees (three humanoids) cost the sum of their sizes, not the square of the total.
//   * Newton builds the dense H = M + J' diag(D active) J in LDS (nv^2 floats) from 16-row blocks of J and factors it with a
//     left-looking Cholesky (sums in registers).  (The reference switches to a blocked 16x16 Cholesky + sparse J here,
//     solver.py:2801-3052; at nv ~ 100 the dense factor is 26-64 KB of LDS and a few hundred microseconds.)
// This is the correctness path for big models, not a tuned one: the register-resident kernels cover nv <= 64.
#pragma once
#include "solver.hpp"

struct BigLayout {
  int q, Ma, grad, Mgrad, search, mv, pgrad, pMgrad, qc, fs, x;  // nv-vectors
  int ja, jv, D, fl, force, da, kind;                            // row vectors (njmax)
  int M, L, dinv, H, stage, HS, total;
  int rs, rmu, rdm, rcon, econe;  // elliptic cones: friction scale / mu / dm of the row's contact, first row | dim << 16, cone Hessian rows (Newton)
};
constexpr int BIG_RB = 16;  // J rows staged per block of the H build
__host__ __device__ inline BigLayout big_layout(int nv, int nC, int njmax, bool newton, bool ell = false) {
  BigLayout p;
  int o = 0;
  int* nvv[] = {&p.q, &p.Ma, &p.grad, &p.Mgrad, &p.search, &p.mv, &p.pgrad, &p.pMgrad, &p.qc, &p.fs, &p.x};
  for (int* f : nvv) { *f = o; o += nv; }
  int* rv[] = {&p.ja, &p.jv, &p.D, &p.fl, &p.force, &p.da, &p.kind};
  for (int* f : rv) { *f = o; o += njmax; }
  int* ev[] = {&p.rs, &p.rmu, &p.rdm, &p.rcon};
  for (int* f : ev) { *f = o; o += ell ? njmax : 0; }
  p.econe = o; o += (ell && newton) ? 6 * njmax : 0;
  p.M = o; o += nC;
  p.L = o; o += nC;
  p.dinv = o; o += nv;
  o = ((o + 3) / 4) * 4;
  const int nvp = ((nv + 3) / 4) * 4;
  p.HS = ((nvp / 4) & 1) ? nvp : nvp + 4;  // H row stride: a multiple of 4 with HS/4 odd (row-per-lane 16-byte reads, no conflicts)
  p.H = o; o += newton ? nv * p.HS : 0;
  p.stage = o; o += newton ? BIG_RB * nvp : 0;
  p.total = ((o + 3) / 4) * 4;
  return p;
}

template <int G>
DEV void solve_big_body(const MjhModel& m, const MjhData& d, float* smem, const Blk& b, bool direct = false, int nefc_lo = -1) {
  if ((int)threadIdx.x >= b.nthreads) return;
  const int nv = m.nv, nC = m.nC, njmax = d.njmax, nvp = d.nv_pad;
  const bool newton = m.solver == SOL_NEWTON;
  const bool ell = m.cone == CONE_ELLIPTIC && d.nmaxpyramid > 1;
  const BigLayout lay = big_layout(nv, nC, njmax, newton, ell);
  int* shi = reinterpret_cast<int*>(smem);
  const MStruct ms = load_mstruct<G>(m, shi, b.nthreads);
  const int lig = threadIdx.x & (G - 1), gib = threadIdx.x / G;
  const int slot = b.w0 + gib;
  if (slot >= d.nworld) return;
  const int w = direct ? slot : d.ws_order[slot];  // (direct: the caller hands world ids, not schedule slots)
  float* S = smem + mstruct_ints(nv, nC) + (size_t)gib * lay.total;
  float *q = S + lay.q, *Ma = S + lay.Ma, *grad = S + lay.grad, *Mgrad = S + lay.Mgrad, *search = S + lay.search, *mv = S + lay.mv,
        *pgrad = S + lay.pgrad, *pMgrad = S + lay.pMgrad, *qc = S + lay.qc, *fs = S + lay.fs, *x = S + lay.x;
  float *ja = S + lay.ja, *jv = S + lay.jv, *rD = S + lay.D, *rfl = S + lay.fl, *force = S + lay.force, *da = S + lay.da;
  int* kind = reinterpret_cast<int*>(S + lay.kind);
  float *rs = S + lay.rs, *rmu = S + lay.rmu, *rdm = S + lay.rdm, *econe = S + lay.econe;
  int* rcon = reinterpret_cast<int*>(S + lay.rcon);
  float *Ml = S + lay.M, *Ll = S + lay.L, *dinv = S + lay.dinv, *H = S + lay.H, *stage = S + lay.stage;
  const int HS = lay.HS;

  const int nefc = min(d.nefc[w], njmax);
  if (nefc <= nefc_lo) return;  // (njmax > 192 on a small model: the register-resident kernels took the worlds with fewer rows)
  const int ne = d.ne[w], nf = d.nf[w];
  const size_t vo = (size_t)w * nv, eo = (size_t)w * njmax;
  const float* Jg = d.efc_J + (size_t)w * d.njmax_pad * nvp;
  const bool warm = !(m.disableflags & DSBL_WARMSTART);

  // ---- M, its sparse factor, qacc_smooth = M^-1 qfrc_smooth ------------------------------------------------------
  gcopy<G>(Ml, d.M + (size_t)w * nC, nC, lig);
  gcopy<G>(Ll, d.M + (size_t)w * nC, nC, lig);
  gcopy<G>(fs, d.qfrc_smooth + vo, nv, lig);
  gcopy<G>(x, d.qfrc_smooth + vo, nv, lig);
  gsync();
  factor_ld<G>(ms, Ll, dinv, nv, lig, &m);
  solve_ld<G>(m, ms, Ll, dinv, x, nv, lig);
  gsync();
  for (int i = lig; i < nv; i += G) {
    d.qacc_smooth[vo + i] = x[i];
    q[i] = nefc > 0 && warm ? d.qacc_warmstart[vo + i] : x[i];
  }
  gsync();
  mul_m_ld<G>(ms, Ml, q, Ma, nv, lig);
  gsync();
  if (nefc == 0) {  // unconstrained: qacc = qacc_smooth (solver.py:3684-3686)
    for (int i = lig; i < nv; i += G) {
      d.qacc[vo + i] = q[i];
      d.qfrc_constraint[vo + i] = 0.0f;
      d.efc_Ma[vo + i] = Ma[i];
    }
    if (lig == 0) d.solver_niter[w] = 0;
    return;
  }

  // J[r,:] . vec for this lane's rows (row-per-lane; a row is nvp contiguous floats, 16-byte aligned)
  auto j_dot = [&](int r, const float* vec) __attribute__((always_inline)) {
    const float4* Jr = reinterpret_cast<const float4*>(Jg + (size_t)r * nvp);
    float s0 = 0.0f, s1 = 0.0f;
    for (int c4 = 0; c4 < nvp / 4; ++c4) {
      const float4 j4 = Jr[c4];
      const int c = 4 * c4;  // (padding columns of J are zero; the vectors have nv entries)
      s0 += j4.x * vec[c] + (c + 2 < nv ? j4.z * vec[c + 2] : 0.0f);
      s1 += (c + 1 < nv ? j4.y * vec[c + 1] : 0.0f) + (c + 3 < nv ? j4.w * vec[c + 3] : 0.0f);
    }
    return s0 + s1;
  };
  for (int r = lig; r < nefc; r += G) {
    rD[r] = d.efc_D[eo + r];
    rfl[r] = d.efc_frictionloss[eo + r];
    kind[r] = r >= ne + nf ? 2 : (r >= ne ? 1 : 0);
    ja[r] = j_dot(r, q) - d.efc_aref[eo + r];
    jv[r] = 0.0f;
    if (ell) {  // rows of an elliptic contact (solver.hpp: kind 4 = first row, 5 = the others); the first row's lane handles the contact
      rs[r] = rmu[r] = rdm[r] = 0.0f;
      rcon[r] = -1;
      if (r >= ne + nf + d.nl[w]) {
        const int cid = d.ws_efc_con[eo + r], c = cid >> 4, dimid = cid & 15;
        const float* cr = d.ws_contact + ((size_t)w * d.concap + c) * CON_STRIDE;
        const int* cri = reinterpret_cast<const int*>(cr);
        if (cri[24] > 1) {
          const int r0 = r - dimid, dim = min(cri[29], nefc - r0);
          const float mu = cr[14] * bf(m.opt_impratio_invsqrt, m.opt_impratio_invsqrt_nb, w, 1)[0];
          kind[r] = dimid == 0 ? 4 : 5;
          rmu[r] = mu;
          rs[r] = dimid == 0 ? mu : cr[CON_FRICTION_WORD(dimid - 1)];
          rdm[r] = safe_div(d.efc_D[eo + r0], mu * mu * (1.0f + mu * mu));
          rcon[r] = r0 | (dim << 16);
        }
      }
    }
  }
  gsync();
  // force / state of the rows of one elliptic contact (first row r0): zones of _eval_constraint solver.py:455-472, middle-zone forces
  // 406-421, and for Newton the rows of the cone Hessian block C (solver.py:2466-2564; same formulas as solver.hpp ell_row_force)
  auto ell_contact = [&](int r0, bool final_state) __attribute__((always_inline)) {
    const int dim = rcon[r0] >> 16;
    const float mu = rmu[r0], dm = rdm[r0];
    float u[6];
    float tt = 0.0f;
    for (int j = 0; j < 6; ++j) {
      u[j] = j < dim ? ja[r0 + j] * rs[r0 + j] : 0.0f;
      if (j > 0) tt += u[j] * u[j];
    }
    const float N = u[0], T = tt <= 0.0f ? 0.0f : sqrtf(tt);
    const int zone = ell_zone(mu, N, T);
    const float fn = -dm * (N - mu * T) * mu;
    for (int a = 0; a < dim; ++a) {
      const int r = r0 + a;
      float f = 0.0f;
      if (zone == ST_QUADRATIC) f = -rD[r] * ja[r];
      else if (zone == ST_CONE) f = a == 0 ? fn : -safe_div(fn, T) * (u[a] * rs[r]);
      force[r] = f;
      da[r] = zone == ST_QUADRATIC ? rD[r] : 0.0f;
      if (final_state) {
        d.efc_force[eo + r] = f;
        d.efc_state[eo + r] = zone;
      }
      if (newton && !final_state) {
        kind[r] = (kind[r] & 7) | (zone == ST_CONE ? 8 : 0);  // bit 3: the row's contact is in the cone zone (H takes C J_c instead of D J)
        if (zone == ST_CONE) {
          const float t = fmaxf(T, MJ_MINVAL), ttt = fmaxf(t * t * t, MJ_MINVAL);
          const float mu_tinv = safe_div(mu, t), mu_n_ttt = mu * safe_div(N, ttt), tdiag = mu * mu - N * mu_tinv;
          const float ua = a == 0 ? 0.0f : u[a];
          for (int bq = 0; bq < 6; ++bq) {
            const float ub = bq == 0 ? 0.0f : u[bq];
            float cab = mu_n_ttt * ua * ub;
            if (a == 0 && bq == 0) cab += 1.0f;
            if (a == 0) cab -= mu_tinv * ub;
            if (bq == 0) cab -= mu_tinv * ua;
            if (a == bq && a > 0) cab += tdiag;
            econe[6 * r + bq] = bq < dim ? dm * rs[r] * rs[r0 + bq] * cab : 0.0f;
          }
        }
      }
    }
  };

  const float tolerance = bf(m.opt_tolerance, m.opt_tolerance_nb, w, 1)[0];
  const float ls_tolerance = bf(m.opt_ls_tolerance, m.opt_ls_tolerance_nb, w, 1)[0];
  const float meaninertia = bf(m.stat_meaninertia, m.stat_meaninertia_nb, w, 1)[0];
  const float scale = meaninertia * (float)nv, rscale = 1.0f / scale;
  const int maxiter = m.iterations, ls_iterations = m.ls_iterations;
  const bool has_fl = nf > 0;
  int niter = 0, ovf = 0;
  float improvement = 0.0f, search_dot = 0.0f;

  for (;;) {
    // ---- force / state of every row (solver.py:1698-1822), qfrc_constraint = J' force (1912-1947) -------------------
    for (int r = lig; r < nefc; r += G) {
      const int kr = kind[r] & 7;
      if (kr == 4) ell_contact(r, false);
      if (kr >= 4) continue;
      float f;
      int st;
      row_force(kr, ja[r], rD[r], has_fl, rfl + r, f, st);
      force[r] = f;
      da[r] = st == ST_QUADRATIC ? rD[r] : 0.0f;
    }
    gsync();
    for (int c = lig; c < nv; c += G) {  // column-per-lane: consecutive lanes read consecutive addresses of every row
      float s = 0.0f;
      for (int r = 0; r < nefc; ++r) s += Jg[(size_t)r * nvp + c] * force[r];
      qc[c] = s;
    }
    gsync();
    // ---- gradient (solver.py:3061-3220) -------------------------------------------------------------------------
    float gd = 0.0f;
    for (int i = lig; i < nv; i += G) {
      const float g = Ma[i] - fs[i] - qc[i];
      grad[i] = g;
      gd += g * g;
    }
    const float grad_dot = gsumg<G>(gd);
    gsync();
    float decrement = 0.0f;
    if (newton) {
      // H = M + J' diag(da) J, dense lower triangle.  J is staged BIG_RB rows at a time (one coalesced copy: the rows of a
      // world are contiguous); lane i then adds the block to row i four columns at a time with the sums in registers, so H
      // sees one read-modify-write per block instead of one per J row.
      for (int i = lig; i < nv; i += G)
        for (int j = 0; j < HS; ++j) H[i * HS + j] = 0.0f;
      gsync();
      for (int i = lig; i < nv; i += G) {
        const int start = ms.rowadr[i], n = ms.rownnz[i];
        for (int a = 0; a < n; ++a) H[i * HS + ms.colind[start + a]] = Ml[start + a];  // colind <= i
      }
      gsync();
      for (int r0 = 0; r0 < nefc; r0 += BIG_RB) {
        const int nr = min(BIG_RB, nefc - r0);
        {
          const float4* src = reinterpret_cast<const float4*>(Jg + (size_t)r0 * nvp);
          float4* dst = reinterpret_cast<float4*>(stage);
          for (int idx = lig; idx < nr * (nvp / 4); idx += G) dst[idx] = src[idx];
        }
        gsync();
        for (int i = lig; i < nv; i += G) {
          float t[BIG_RB];
          bool any = false;
#pragma unroll
          for (int k = 0; k < BIG_RB; ++k) {
            t[k] = k < nr ? da[r0 + k] * stage[k * nvp + i] : 0.0f;
            if (ell && k < nr && (kind[r0 + k] & 8)) {  // row of a contact in the cone zone: (C J_c)[row][i] (its rows may straddle the staged block: from L2)
              const int rc0 = rcon[r0 + k] & 0xffff, dim = rcon[r0 + k] >> 16;
              float acc = 0.0f;
              for (int bq = 0; bq < dim; ++bq) acc += econe[6 * (r0 + k) + bq] * Jg[(size_t)(rc0 + bq) * nvp + i];
              t[k] = acc;
            }
            any = any || t[k] != 0.0f;
          }
          if (!any) continue;
          for (int j0 = 0; j0 <= i; j0 += 4) {
            float4 acc = *reinterpret_cast<const float4*>(H + i * HS + j0);
#pragma unroll
            for (int k = 0; k < BIG_RB; ++k) {
              const float4 j4 = *reinterpret_cast<const float4*>(stage + k * nvp + j0);  // (rows >= nr hold stale data: t = 0)
              acc.x += t[k] * j4.x;
              acc.y += t[k] * j4.y;
              acc.z += t[k] * j4.z;
              acc.w += t[k] * j4.w;
            }
            *reinterpret_cast<float4*>(H + i * HS + j0) = acc;
          }
        }
        gsync();
      }
      // Cholesky in place (lower), left-looking: column j of L from the finished columns k < j,
      //   s_i = H[i][j] - sum_k L[i][k] L[j][k]   (lane i: own row i against the broadcast row j, sums in registers)
      for (int j = 0; j < nv; ++j) {
        for (int i = j + lig; i < nv; i += G) {
          float s0 = 0.0f, s1 = 0.0f;
          int k = 0;
          for (; k + 4 <= j; k += 4) {
            const float4 a4 = *reinterpret_cast<const float4*>(H + i * HS + k);
            const float4 b4 = *reinterpret_cast<const float4*>(H + j * HS + k);
            s0 += a4.x * b4.x + a4.z * b4.z;
            s1 += a4.y * b4.y + a4.w * b4.w;
          }
          for (; k < j; ++k) s0 += H[i * HS + k] * H[j * HS + k];
          H[i * HS + j] -= s0 + s1;
        }
        gsync();
        const float pv = fmaxf(H[j * HS + j], MJ_MINVAL);
        const float inv = 1.0f / sqrtf(pv);
        gsync();
        for (int i = j + lig; i < nv; i += G) H[i * HS + j] = i == j ? pv * inv : H[i * HS + j] * inv;
        gsync();
      }
      for (int i = lig; i < nv; i += G) Mgrad[i] = grad[i];
      gsync();
      for (int j = 0; j < nv; ++j) {  // forward substitution, column oriented
        if (lig == 0) Mgrad[j] = Mgrad[j] / H[j * HS + j];
        gsync();
        const float yj = Mgrad[j];
        for (int i = j + 1 + lig; i < nv; i += G) Mgrad[i] -= H[i * HS + j] * yj;
        gsync();
      }
      for (int j = nv - 1; j >= 0; --j) {  // backward substitution with L'
        if (lig == 0) Mgrad[j] = Mgrad[j] / H[j * HS + j];
        gsync();
        const float xj = Mgrad[j];
        for (int i = lig; i < j; i += G) Mgrad[i] -= H[j * HS + i] * xj;
        gsync();
      }
      float sd = 0.0f, dc = 0.0f;
      for (int i = lig; i < nv; i += G) {
        search[i] = -Mgrad[i];
        sd += Mgrad[i] * Mgrad[i];
        dc += grad[i] * Mgrad[i];
      }
      search_dot = gsumg<G>(sd);
      decrement = gsumg<G>(dc);
    } else {
      for (int i = lig; i < nv; i += G) Mgrad[i] = grad[i];
      gsync();
      solve_ld<G>(m, ms, Ll, dinv, Mgrad, nv, lig);  // Mgrad = M^-1 grad
      gsync();
    }
    // ---- convergence, search direction ----------------------------------------------------------------------------
    if (niter == 0) {
      if (!newton) {
        float sd = 0.0f;
        for (int i = lig; i < nv; i += G) {
          search[i] = -Mgrad[i];
          sd += Mgrad[i] * Mgrad[i];
          pgrad[i] = grad[i];
          pMgrad[i] = Mgrad[i];
        }
        search_dot = gsumg<G>(sd);
      }
    } else {
      const float imp = improvement * rscale, gradient = sqrtf(grad_dot) * rscale;
      bool done;
      if (newton) {
        done = (imp < tolerance) || (gradient < tolerance) || (0.5f * decrement * rscale < tolerance);
      } else {
        float num = 0.0f, den = 0.0f;
        for (int i = lig; i < nv; i += G) {
          num += grad[i] * (Mgrad[i] - pMgrad[i]);
          den += pgrad[i] * pMgrad[i];
        }
        num = gsumg<G>(num);
        den = gsumg<G>(den);
        const float beta = fmaxf(0.0f, num / fmaxf(MJ_MINVAL, den));
        done = (imp < tolerance) || (gradient < tolerance);
        if (!done) {
          float sd = 0.0f;
          for (int i = lig; i < nv; i += G) {
            const float s = -Mgrad[i] + beta * search[i];
            search[i] = s;
            sd += s * s;
            pgrad[i] = grad[i];
            pMgrad[i] = Mgrad[i];
          }
          search_dot = gsumg<G>(sd);
        }
      }
      if (done) break;
      if (niter >= maxiter) {
        ovf |= OVF_ITERATIONS;
        break;
      }
    }
    if (maxiter == 0) break;
    gsync();
    // ---- mv = M search, jv = J search ---------------------------------------------------------------------------
    mul_m_ld<G>(ms, Ml, search, mv, nv, lig);
    gsync();
    for (int r = lig; r < nefc; r += G) jv[r] = j_dot(r, search);
    float g1 = 0.0f, g2 = 0.0f;
    for (int i = lig; i < nv; i += G) {
      g1 += search[i] * (Ma[i] - fs[i]);
      g2 += 0.5f * search[i] * mv[i];
    }
    const float gauss1 = gsumg<G>(g1), gauss2 = gsumg<G>(g2);
    gsync();
    // ---- line search (solver.py:835-1347) ------------------------------------------------------------------------
    const float gtol = fmaxf(tolerance * ls_tolerance * sqrtf(search_dot) * scale, 1e-6f);
    auto total = [&](float a) __attribute__((always_inline)) {
      P3 s = P3{0.0f, 0.0f, 0.0f};
      for (int r = lig; r < nefc; r += G) {
        const int kr = kind[r] & 7;
        if (kr == 5) continue;  // (evaluated with its contact by the first row's lane)
        P3 t;
        if (kr == 4) {  // the contact's ray constants (solver.hpp EllRay), then the reference's shifted forms
          const int dim = rcon[r] >> 16;
          EllRay e;
          e.mu = rmu[r];
          e.dm = rdm[r];
          e.q0 = e.q1 = e.q2 = e.uu = e.uv = e.vv = 0.0f;
          e.u0 = ja[r] * rs[r];
          e.v0 = jv[r] * rs[r];
          for (int j = 0; j < dim; ++j) {
            const float jaj = ja[r + j], jvj = jv[r + j], Dj = rD[r + j];
            e.q0 += 0.5f * Dj * jaj * jaj;
            e.q1 += Dj * jvj * jaj;
            e.q2 += 0.5f * Dj * jvj * jvj;
            if (j > 0) {
              const float uj = jaj * rs[r + j], vj = jvj * rs[r + j];
              e.uu += uj * uj;
              e.uv += uj * vj;
              e.vv += vj * vj;
            }
          }
          ell_ray_reference(e);
          t = ell_eval(e, a);
        } else {
          t = eval_row(ja[r], jv[r], rD[r], rfl[r], kr, a);
        }
        s.c += t.c;
        s.g += t.g;
        s.h += t.h;
      }
      return P3{a * a * gauss2 + a * gauss1 + gsumg<G>(s.c), 2.0f * a * gauss2 + gauss1 + gsumg<G>(s.g), 2.0f * gauss2 + gsumg<G>(s.h)};
    };
    P3 p0 = total(0.0f);
    p0.c = 0.0f;
    const float lo_alpha_in = -fast_div(p0.g, p0.h);
    const P3 lo_in = total(lo_alpha_in);
    float alpha = 0.0f;
    improvement = 0.0f;
    bool ls_converged = fabsf(lo_in.g) < gtol && lo_in.c < 0.0f;
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
        const P3 lo_next = total(a_lo), hi_next = total(a_hi), mid = total(a_mid);
        auto take = [](bool c, P3& dst, float& da_, const P3& src, float sa) __attribute__((always_inline)) {
          dst.c = c ? src.c : dst.c;
          dst.g = c ? src.g : dst.g;
          dst.h = c ? src.h : dst.h;
          da_ = c ? sa : da_;
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
    if (!ls_converged) ovf |= OVF_LS_ITERATIONS;
    // ---- move along the ray ---------------------------------------------------------------------------------------
    for (int i = lig; i < nv; i += G) {
      q[i] += alpha * search[i];
      Ma[i] += alpha * mv[i];
    }
    for (int r = lig; r < nefc; r += G) ja[r] += alpha * jv[r];
    gsync();
    ++niter;
  }

  // ---- outputs ------------------------------------------------------------------------------------------------------
  for (int i = lig; i < nv; i += G) {
    d.qacc[vo + i] = q[i];
    d.qfrc_constraint[vo + i] = qc[i];
    d.efc_Ma[vo + i] = Ma[i];
  }
  for (int r = lig; r < nefc; r += G) {
    const int kr = kind[r] & 7;
    if (kr == 4) ell_contact(r, true);
    if (kr >= 4) continue;
    float f;
    int st;
    row_force(kr, ja[r], rD[r], has_fl, rfl + r, f, st);
    d.efc_force[eo + r] = f;
    d.efc_state[eo + r] = st;
  }
  if (lig == 0) {
    d.solver_niter[w] = niter;
    if (ovf) atomicOr(d.overflow + w, ovf);
  }
}
