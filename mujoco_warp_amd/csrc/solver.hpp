// solver.hpp -- persistent per-world constraint solver (Newton and CG, pyramidal/frictionless cones).
//
// Reference: solver.py:3671-3743 (solve/_solve), 3525-3620 (_solver_iteration), 3622-3668 (init_context),
// 835-1347 (_linesearch_iterative_kernel), 1698-1822 (_update_constraint_efc), 1912-1947 (qfrc_constraint),
// 3061-3220 (_update_gradient), 2365-2440 (JTDAJ), 2567-2603 (Cholesky solve), 3283-3450 (CG), 3454-3497
// (_solve_done).  The reference runs ~12 launches per iteration inside a CUDA conditional-graph while loop and
// keeps J/H/vectors in global memory; every world iterates until ALL worlds converge.
//
// MI355X mapping: one 32-lane group (half a wavefront) owns a world for the WHOLE solve.  J (njmax x nv) and
// all solver vectors are LDS resident; for Newton, lane i keeps row i of M, of H = M + J^T D J and column i
// of its Cholesky factor in VGPRs (the kernel is LDS-capacity bound, so VGPRs are free), the factorisation
// is a right-looking Cholesky whose pivot column is broadcast through a 128 B LDS line, the triangular
// solves broadcast with v_readlane, and every row reduction of the line search is a DPP row_shr/row_bcast
// tree.  Each world leaves the loop as soon as ITS convergence test passes.  HBM traffic is the algorithmic
// minimum: J, D, aref, M, three nv-vectors in; qacc, qfrc_constraint, Ma, force, state out.
#pragma once
#include "dev_common.hpp"
#include "smooth.hpp"

// ---- DPP reduction over a 32-lane group; every lane receives the total ---------------------------------
template <int CTRL, int ROW_MASK, int BANK_MASK>
DEV float dpp_add_f(float v) {
  const int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, true);
  return v + __int_as_float(r);
}
DEV float bcast32(float v, int k) {  // value of lane k of this lane's 32-group (k compile-time)
  const int a = __builtin_amdgcn_readlane(__float_as_int(v), k);
  const int b = __builtin_amdgcn_readlane(__float_as_int(v), k + 32);
  return __int_as_float((threadIdx.x & 32) ? b : a);
}
DEV float gsum32(float v) {
  v = dpp_add_f<0x111, 0xf, 0xf>(v);  // row_shr:1
  v = dpp_add_f<0x112, 0xf, 0xf>(v);  // row_shr:2
  v = dpp_add_f<0x114, 0xf, 0xf>(v);  // row_shr:4
  v = dpp_add_f<0x118, 0xf, 0xf>(v);  // row_shr:8  -> lane 15 of each row holds the row sum
  v = dpp_add_f<0x142, 0xa, 0xf>(v);  // row_bcast:15 into rows 1 and 3 -> lanes 31 / 63 hold the group sums
  return bcast32(v, 31);
}

// ---- dense Cholesky with lane i owning row i (NVP <= 32), all indices compile-time ----------------------
// h: row i of the SPD matrix on entry, row i of L on exit (entries above the diagonal are junk);
// lt: column i of L below the diagonal (for the transposed solve); rdiag = 1 / L[i][i].
// `col` is an LDS scratch of 8*(NVP+4) floats private to the group.  Lanes >= NVP must hold identity rows.
template <int NVP>
DEV void chol_factor_rows(float (&h)[NVP], float (&lt)[NVP], float& rdiag, float* col, int lig) {
  constexpr int JS = NVP + 4;
  const bool own = lig < NVP;
  const int ligc = own ? lig : NVP - 1;
  rdiag = 1.0f;
#pragma unroll
  for (int j = 0; j < NVP; ++j) {  // right-looking; pivot column broadcast through LDS (double buffered)
    float* cb = col + (j & 1) * NVP;
    if (own) cb[lig] = h[j];
    gsync();
    const float piv = sqrtf(fmaxf(cb[j], MJ_MINVAL));
    const float inv = 1.0f / piv;
    const float lij = (lig == j) ? piv : h[j] * inv;
    h[j] = lij;
    if (lig == j) rdiag = inv;
#pragma unroll
    for (int k = j + 1; k < NVP; ++k) h[k] -= lij * (cb[k] * inv);
  }
  gsync();
#pragma unroll
  for (int c0 = 0; c0 < NVP; c0 += 8) {  // column i of L: rows pass through an 8-row LDS tile
    if (lig >= c0 && lig < c0 + 8 && own) {
#pragma unroll
      for (int c4 = 0; c4 < NVP / 4; ++c4)
        *reinterpret_cast<float4*>(col + (lig - c0) * JS + 4 * c4) = make_float4(h[4 * c4], h[4 * c4 + 1], h[4 * c4 + 2], h[4 * c4 + 3]);
    }
    gsync();
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const float v = col[kk * JS + ligc];
      lt[c0 + kk] = (own && c0 + kk > lig) ? v : 0.0f;
    }
    gsync();
  }
}
// x = (L L^T)^-1 g for the lane's component; broadcasts via v_readlane (no LDS traffic)
template <int NVP>
DEV float chol_solve_rows(const float (&h)[NVP], const float (&lt)[NVP], float rdiag, float g) {
  const int lig = threadIdx.x & 31;
  float acc = g, y = 0.0f, x = 0.0f;
#pragma unroll
  for (int k = 0; k < NVP; ++k) {
    const float yk = bcast32(acc * rdiag, k);
    if (lig == k) y = yk;
    acc -= h[k] * yk;
  }
  acc = y;
#pragma unroll
  for (int k = NVP - 1; k >= 0; --k) {
    const float xk = bcast32(acc * rdiag, k);
    if (lig == k) x = xk;
    acc -= lt[k] * xk;
  }
  return x;
}

struct SolveLayout {
  int J, D, Jaref, jv, floss, force, state, qacc, Ma, grad, search, mv, fs, Mg, pg, pMg, qc, col, L, dinv, total;
};
template <int NVP>
__host__ __device__ inline SolveLayout solve_layout(int njmax, int nC, bool cg) {
  SolveLayout p;
  int o = 0;
  p.J = o; o += (njmax > NVP ? njmax : NVP) * (NVP + 4);  // also stages the dense NVP x NVP copy of M
  p.D = o; o += njmax;
  p.Jaref = o; o += njmax;
  p.jv = o; o += njmax;
  p.floss = o; o += njmax;
  p.force = o; o += njmax;
  p.state = o; o += njmax;
  p.qacc = o; o += 32;  // nv-vectors are indexed by lane id (0..31)
  p.Ma = o; o += 32;
  p.grad = o; o += 32;
  p.search = o; o += 32;
  p.mv = o; o += 32;
  p.fs = o; o += 32;
  p.Mg = o; o += 32;
  p.pg = o; o += 32;
  p.pMg = o; o += 32;
  p.qc = o; o += 32;
  p.col = o; o += 8 * (NVP + 4);  // pivot-column double buffer (2*NVP) / 8-row transpose tile
  p.L = o;
  p.dinv = o;
  p.total = ((o + 3) / 4) * 4;
  return p;
}

// (cost - cost(0), grad, hess) of the constraint part along the ray at three step sizes
// (solver.py:702-755 _compute_efc_eval_pt_3alphas_pyramidal; single-alpha variants 518-556, 620-647)
struct Pt3 {
  float c[3], g[3], h[3];
};
DEV void eval_rows(const float* eJaref, const float* ejv, const float* eD, const float* efl, int nefc, int ne, int nf,
                   int lig, float a0, float a1, float a2, Pt3& out) {
  float c0 = 0, c1 = 0, c2 = 0, g0 = 0, g1 = 0, g2 = 0, h0 = 0, h1 = 0, h2 = 0;
  for (int r = lig; r < nefc; r += 32) {
    const float ja = eJaref[r], jv = ejv[r], D = eD[r];
    const float jvD = jv * D, hess = jv * jvD, grad0 = jvD * ja;
    const float x0 = ja + a0 * jv, x1 = ja + a1 * jv, x2 = ja + a2 * jv;
    if (r >= ne + nf) {
      const float quad0 = 0.5f * D * ja * ja;
      const float cost0 = ja < 0.0f ? quad0 : 0.0f;
      const float offset = quad0 - cost0;
      if (x0 < 0.0f) { c0 += a0 * (grad0 + 0.5f * a0 * hess) + offset; g0 += grad0 + a0 * hess; h0 += hess; } else c0 -= cost0;
      if (x1 < 0.0f) { c1 += a1 * (grad0 + 0.5f * a1 * hess) + offset; g1 += grad0 + a1 * hess; h1 += hess; } else c1 -= cost0;
      if (x2 < 0.0f) { c2 += a2 * (grad0 + 0.5f * a2 * hess) + offset; g2 += grad0 + a2 * hess; h2 += hess; } else c2 -= cost0;
    } else if (r >= ne) {
      const float f = efl[r], rf = safe_div(f, D);
      const float cost0 = (-rf < ja && ja < rf) ? 0.5f * D * ja * ja : (ja <= -rf ? f * (-0.5f * rf - ja) : f * (-0.5f * rf + ja));
      const float fjv = f * jv;
      if (-rf < x0 && x0 < rf) { c0 += 0.5f * D * x0 * x0 - cost0; g0 += jvD * x0; h0 += hess; }
      else if (x0 <= -rf) { c0 += f * (-0.5f * rf - x0) - cost0; g0 -= fjv; } else { c0 += f * (-0.5f * rf + x0) - cost0; g0 += fjv; }
      if (-rf < x1 && x1 < rf) { c1 += 0.5f * D * x1 * x1 - cost0; g1 += jvD * x1; h1 += hess; }
      else if (x1 <= -rf) { c1 += f * (-0.5f * rf - x1) - cost0; g1 -= fjv; } else { c1 += f * (-0.5f * rf + x1) - cost0; g1 += fjv; }
      if (-rf < x2 && x2 < rf) { c2 += 0.5f * D * x2 * x2 - cost0; g2 += jvD * x2; h2 += hess; }
      else if (x2 <= -rf) { c2 += f * (-0.5f * rf - x2) - cost0; g2 -= fjv; } else { c2 += f * (-0.5f * rf + x2) - cost0; g2 += fjv; }
    } else {
      c0 += a0 * (grad0 + 0.5f * a0 * hess); g0 += grad0 + a0 * hess; h0 += hess;
      c1 += a1 * (grad0 + 0.5f * a1 * hess); g1 += grad0 + a1 * hess; h1 += hess;
      c2 += a2 * (grad0 + 0.5f * a2 * hess); g2 += grad0 + a2 * hess; h2 += hess;
    }
  }
  out.c[0] = gsum32(c0); out.c[1] = gsum32(c1); out.c[2] = gsum32(c2);
  out.g[0] = gsum32(g0); out.g[1] = gsum32(g1); out.g[2] = gsum32(g2);
  out.h[0] = gsum32(h0); out.h[1] = gsum32(h1); out.h[2] = gsum32(h2);
}
struct P3 {
  float c, g, h;
};
DEV bool in_bracket(P3 x, P3 y) { return (x.g < y.g && y.g < 0.0f) || (x.g > y.g && y.g > 0.0f); }

template <int NVP, bool NEWTON>
__global__ void __launch_bounds__(256) k_solve(MjhModel m, MjhData d) {
  constexpr int G = 32;
  constexpr int JS = NVP + 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int nv = m.nv, nC = m.nC, njmax = d.njmax, nvp = d.nv_pad;
  const SolveLayout lay = solve_layout<NVP>(njmax, nC, !NEWTON);
  int* shi = reinterpret_cast<int*>(smem);
  const MStruct ms = load_mstruct<G>(m, shi);
  const int lig = threadIdx.x & (G - 1), gib = threadIdx.x / G;
  const int w = blockIdx.x * (blockDim.x / G) + gib;
  if (w >= d.nworld) return;
  float* S = smem + mstruct_ints(nv, nC) + (size_t)gib * lay.total;
  float* Jl = S + lay.J;
  float *eD = S + lay.D, *eJaref = S + lay.Jaref, *ejv = S + lay.jv, *efl = S + lay.floss, *eforce = S + lay.force;
  int* estate = reinterpret_cast<int*>(S + lay.state);
  float *vq = S + lay.qacc, *vMa = S + lay.Ma, *vgrad = S + lay.grad, *vsearch = S + lay.search, *vmv = S + lay.mv,
        *vfs = S + lay.fs, *vMg = S + lay.Mg, *vpg = S + lay.pg, *vpMg = S + lay.pMg, *vqc = S + lay.qc, *col = S + lay.col;

  const int nefc = min(d.nefc[w], njmax);
  const int ne = d.ne[w], nf = d.nf[w];
  const size_t vo = (size_t)w * nv, eo = (size_t)w * njmax;
  const bool active = lig < nv;

  // ---- M row of this lane into registers (dense staging in the J region) ----------------------------------
  float mrow[NVP];
  {
    for (int idx = lig; idx < NVP * JS; idx += G) Jl[idx] = 0.0f;
    gsync();
    const float* Mg = d.M + (size_t)w * nC;
    for (int i = lig; i < nv; i += G) {
      const int start = ms.rowadr[i], n = ms.rownnz[i];
      for (int a = 0; a < n; ++a) {
        const int j = ms.colind[start + a];
        const float v = Mg[start + a];
        Jl[i * JS + j] = v;
        Jl[j * JS + i] = v;
      }
    }
    gsync();
#pragma unroll
    for (int c = 0; c < NVP; ++c) mrow[c] = active ? Jl[lig * JS + c] : (c == lig ? 1.0f : 0.0f);
    gsync();
  }

  // ---- dof vectors --------------------------------------------------------------------------------------
  const bool warm = !(m.disableflags & DSBL_WARMSTART);
  {
    float q = 0.0f, fs = 0.0f;
    if (active) {
      q = nefc > 0 && warm ? d.qacc_warmstart[vo + lig] : d.qacc_smooth[vo + lig];
      fs = d.qfrc_smooth[vo + lig];
    }
    vq[lig] = q;
    vfs[lig] = fs;
    vsearch[lig] = 0.0f;
  }
  gsync();
  auto mul_m_row = [&](const float* vec) __attribute__((always_inline)) {  // lane i: sum_c M[i][c] vec[c], vec broadcast from LDS
    float s = 0.0f;
#pragma unroll
    for (int c4 = 0; c4 < NVP / 4; ++c4) {
      const float4 v4 = *reinterpret_cast<const float4*>(vec + 4 * c4);
      s += mrow[4 * c4] * v4.x + mrow[4 * c4 + 1] * v4.y + mrow[4 * c4 + 2] * v4.z + mrow[4 * c4 + 3] * v4.w;
    }
    return active ? s : 0.0f;
  };
  vMa[lig] = mul_m_row(vq);

  if (nefc == 0) {  // unconstrained: qacc = qacc_smooth (solver.py:3684-3686)
    if (active) {
      d.qacc[vo + lig] = vq[lig];
      d.qfrc_constraint[vo + lig] = 0.0f;
      d.efc_Ma[vo + lig] = vMa[lig];
    }
    if (lig == 0) d.solver_niter[w] = 0;
    return;
  }

  // ---- J, D, aref, frictionloss into LDS ------------------------------------------------------------------
  {
    const float* Jg = d.efc_J + (size_t)w * d.njmax_pad * nvp;
    for (int r = 0; r < nefc; ++r)
      for (int c = lig; c < JS; c += G) Jl[r * JS + c] = c < nvp ? Jg[(size_t)r * nvp + c] : 0.0f;
    for (int r = lig; r < nefc; r += G) {
      eD[r] = d.efc_D[eo + r];
      efl[r] = d.efc_frictionloss[eo + r];
    }
  }
  gsync();
  auto j_dot = [&](const float* vec, int r) __attribute__((always_inline)) {  // J[r,:] . vec  (row-per-lane, conflict-free b128 reads)
    float s = 0.0f;
#pragma unroll
    for (int c4 = 0; c4 < NVP / 4; ++c4) {
      const float4 j4 = *reinterpret_cast<const float4*>(Jl + r * JS + 4 * c4);
      const float4 v4 = *reinterpret_cast<const float4*>(vec + 4 * c4);
      s += j4.x * v4.x + j4.y * v4.y + j4.z * v4.z + j4.w * v4.w;
    }
    return s;
  };
  for (int r = lig; r < nefc; r += G) eJaref[r] = j_dot(vq, r) - d.efc_aref[eo + r];
  gsync();

  const float tolerance = bf(m.opt_tolerance, m.opt_tolerance_nb, w, 1)[0];
  const float ls_tolerance = bf(m.opt_ls_tolerance, m.opt_ls_tolerance_nb, w, 1)[0];
  const float meaninertia = bf(m.stat_meaninertia, m.stat_meaninertia_nb, w, 1)[0];
  const float scale = meaninertia * (float)nv;
  const float rscale = 1.0f / scale;

  float grad_dot = 0.0f, search_dot = 0.0f, decrement = 0.0f;
  float h[NVP], lt[NVP];
  float mrdiag = 1.0f;

  // force/state per row + qfrc_constraint = J^T force (solver.py:1698-1822, 1912-1947)
  auto update_constraint = [&]() __attribute__((always_inline)) {
    for (int r = lig; r < nefc; r += G) {
      const float ja = eJaref[r], D = eD[r];
      float force;
      int state;
      if (r < ne) {
        force = -D * ja;
        state = ST_QUADRATIC;
      } else if (r < ne + nf) {
        const float f = efl[r], rf = safe_div(f, D);
        if (ja <= -rf) { force = f; state = ST_LINEARNEG; }
        else if (ja >= rf) { force = -f; state = ST_LINEARPOS; }
        else { force = -D * ja; state = ST_QUADRATIC; }
      } else if (ja >= 0.0f) {
        force = 0.0f;
        state = ST_SATISFIED;
      } else {
        force = -D * ja;
        state = ST_QUADRATIC;
      }
      eforce[r] = force;
      estate[r] = state;
    }
    gsync();
    float s = 0.0f;
    if (lig < JS)
      for (int r = 0; r < nefc; ++r) s += Jl[r * JS + lig] * eforce[r];
    vqc[lig] = active ? s : 0.0f;
  };

  // grad, then search direction (solver.py:3061-3220)
  auto update_gradient = [&]() __attribute__((always_inline)) {
    const float g = active ? (vMa[lig] - vfs[lig] - vqc[lig]) : 0.0f;
    vgrad[lig] = g;
    grad_dot = gsum32(g * g);
    if (NEWTON) {
      // H row = M row + sum over QUADRATIC rows of D * J[r][i] * J[r][:]   (JTDAJ, solver.py:2365-2440)
#pragma unroll
      for (int c = 0; c < NVP; ++c) h[c] = mrow[c];
      for (int r = 0; r < nefc; ++r) {
        if (estate[r] != ST_QUADRATIC) continue;
        const float jd = Jl[r * JS + lig] * eD[r];
#pragma unroll
        for (int c4 = 0; c4 < NVP / 4; ++c4) {
          const float4 j4 = *reinterpret_cast<const float4*>(Jl + r * JS + 4 * c4);
          h[4 * c4] += jd * j4.x;
          h[4 * c4 + 1] += jd * j4.y;
          h[4 * c4 + 2] += jd * j4.z;
          h[4 * c4 + 3] += jd * j4.w;
        }
      }
      float rdiag;
      chol_factor_rows<NVP>(h, lt, rdiag, col, lig);
      float x = chol_solve_rows<NVP>(h, lt, rdiag, g);
      if (!active) x = 0.0f;
      vMg[lig] = x;
      vsearch[lig] = -x;
      search_dot = gsum32(x * x);
      decrement = gsum32(g * x);
      gsync();
    } else {
      // CG: Mgrad = M^-1 grad with the dense Cholesky factor of M held in registers (factored once per solve)
      float x = chol_solve_rows<NVP>(h, lt, mrdiag, g);
      if (!active) x = 0.0f;
      vMg[lig] = x;
      gsync();
    }
  };
  if (!NEWTON) {
#pragma unroll
    for (int c = 0; c < NVP; ++c) h[c] = mrow[c];
    chol_factor_rows<NVP>(h, lt, mrdiag, col, lig);
  }

  int niter = 0;
  const int maxiter = m.iterations, ls_iterations = m.ls_iterations;
  int ovf = 0;
  float improvement = 0.0f;
  // One loop body = [constraint update, gradient/search update, convergence test, line search + move], so that the
  // (large, fully unrolled) gradient code has a single call site.  Iteration 0 is init_context (solver.py:3622).
  for (;;) {
    update_constraint();
    gsync();
    update_gradient();
    if (niter == 0) {
      if (!NEWTON) {  // CG: search = -Mgrad (solver.py:1663-1695)
        const float mg = vMg[lig];
        vsearch[lig] = -mg;
        search_dot = gsum32(active ? mg * mg : 0.0f);
        vpg[lig] = vgrad[lig];
        vpMg[lig] = mg;
        gsync();
      }
    } else {
      const float imp = improvement * rscale, gradient = sqrtf(grad_dot) * rscale;
      bool done;
      if (NEWTON) {
        done = (imp < tolerance) || (gradient < tolerance) || (0.5f * decrement * rscale < tolerance);
      } else {
        // Polak-Ribiere (solver.py:3283-3450)
        const float mg = vMg[lig], pm = vpMg[lig];
        const float num = gsum32(active ? vgrad[lig] * (mg - pm) : 0.0f);
        const float den = gsum32(active ? vpg[lig] * pm : 0.0f);
        const float beta = fmaxf(0.0f, num / fmaxf(MJ_MINVAL, den));
        done = (imp < tolerance) || (gradient < tolerance);
        if (!done) {
          const float s = -mg + beta * vsearch[lig];
          vsearch[lig] = s;
          search_dot = gsum32(active ? s * s : 0.0f);
          vpg[lig] = vgrad[lig];
          vpMg[lig] = mg;
          gsync();
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
    vmv[lig] = mul_m_row(vsearch);
    for (int r = lig; r < nefc; r += G) ejv[r] = j_dot(vsearch, r);
    gsync();
    // ---- line search (solver.py:835-1347) -------------------------------------------------------------------
    const float sr = vsearch[lig];
    const float gauss1 = gsum32(active ? sr * (vMa[lig] - vfs[lig]) : 0.0f);
    const float gauss2 = gsum32(active ? 0.5f * sr * vmv[lig] : 0.0f);
    const float gtol = fmaxf(tolerance * ls_tolerance * sqrtf(search_dot) * scale, 1e-6f);
    Pt3 e;
    eval_rows(eJaref, ejv, eD, efl, nefc, ne, nf, lig, 0.0f, 0.0f, 0.0f, e);
    const P3 p0 = P3{0.0f, gauss1 + e.g[0], 2.0f * gauss2 + e.h[0]};
    const float lo_alpha_in = -safe_div(p0.g, p0.h);
    eval_rows(eJaref, ejv, eD, efl, nefc, ne, nf, lig, lo_alpha_in, lo_alpha_in, lo_alpha_in, e);
    const P3 lo_in = P3{lo_alpha_in * lo_alpha_in * gauss2 + lo_alpha_in * gauss1 + e.c[0],
                        2.0f * lo_alpha_in * gauss2 + gauss1 + e.g[0], 2.0f * gauss2 + e.h[0]};
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
        const float a_lo = lo_alpha - safe_div(lo.g, lo.h), a_hi = hi_alpha - safe_div(hi.g, hi.h);
        const float a_mid = 0.5f * (lo_alpha + hi_alpha);
        eval_rows(eJaref, ejv, eD, efl, nefc, ne, nf, lig, a_lo, a_hi, a_mid, e);
        const P3 lo_next = P3{a_lo * a_lo * gauss2 + a_lo * gauss1 + e.c[0], 2.0f * a_lo * gauss2 + gauss1 + e.g[0], 2.0f * gauss2 + e.h[0]};
        const P3 hi_next = P3{a_hi * a_hi * gauss2 + a_hi * gauss1 + e.c[1], 2.0f * a_hi * gauss2 + gauss1 + e.g[1], 2.0f * gauss2 + e.h[1]};
        const P3 mid = P3{a_mid * a_mid * gauss2 + a_mid * gauss1 + e.c[2], 2.0f * a_mid * gauss2 + gauss1 + e.g[2], 2.0f * gauss2 + e.h[2]};
        const bool s1 = in_bracket(lo, lo_next);
        if (s1) { lo = lo_next; lo_alpha = a_lo; }
        const bool s2 = in_bracket(lo, mid);
        if (s2) { lo = mid; lo_alpha = a_mid; }
        const bool s3 = in_bracket(lo, hi_next);
        if (s3) { lo = hi_next; lo_alpha = a_hi; }
        const bool h1 = in_bracket(hi, hi_next);
        if (h1) { hi = hi_next; hi_alpha = a_hi; }
        const bool h2 = in_bracket(hi, mid);
        if (h2) { hi = mid; hi_alpha = a_mid; }
        const bool h3 = in_bracket(hi, lo_next);
        if (h3) { hi = lo_next; hi_alpha = a_lo; }
        const bool swap_lo = s1 || s2 || s3, swap_hi = h1 || h2 || h3;
        const bool ls_done = (!swap_lo && !swap_hi) || (lo.c < 0.0f && lo.g < 0.0f && lo.g > -gtol) || (hi.c < 0.0f && hi.g > 0.0f && hi.g < gtol);
        const bool improved = lo.c < 0.0f || hi.c < 0.0f;
        const bool lo_better = lo.c < hi.c;
        if (improved) {
          alpha = lo_better ? lo_alpha : hi_alpha;
          improvement = -(lo_better ? lo.c : hi.c);
        }
        if (ls_done) {
          ls_converged = true;
          break;
        }
      }
    }
    if (!ls_converged) ovf |= OVF_LS_ITERATIONS;
    // ---- move along the ray ------------------------------------------------------------------------------
    vq[lig] += alpha * sr;
    vMa[lig] += alpha * vmv[lig];
    for (int r = lig; r < nefc; r += G) eJaref[r] += alpha * ejv[r];
    gsync();
    ++niter;
  }

  // ---- outputs ---------------------------------------------------------------------------------------------
  if (active) {
    d.qacc[vo + lig] = vq[lig];
    d.qfrc_constraint[vo + lig] = vqc[lig];
    d.efc_Ma[vo + lig] = vMa[lig];
  }
  for (int r = lig; r < nefc; r += G) {
    d.efc_force[eo + r] = eforce[r];
    d.efc_state[eo + r] = estate[r];
  }
  if (lig == 0) {
    d.solver_niter[w] = niter;
    if (ovf) atomicOr(d.overflow + w, ovf);
  }
}
