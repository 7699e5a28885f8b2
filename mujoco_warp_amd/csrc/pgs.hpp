// pgs.hpp -- projected Gauss-Seidel constraint solver (dual problem, pyramidal/frictionless cones), one world per lane group.
//
// The reference has no PGS (types.py:502, io.py solver check): the algorithm is the one MuJoCo C documents and implements
// (engine_solver.c mj_solPGS; warm start as in engine_forward.c), restated in float64 by oracle/mjref.c:solve_pgs, which
// this kernel is tested against.  Dual problem:  min_f 0.5 f'(A+R)f + f'b,  A = J M^-1 J', R = 1/D,
// b = J qacc_smooth - aref;  f free on equality rows, |f| <= frictionloss on friction-loss rows, f >= 0 on limit/contact rows.
//
// MI355X mapping.  Gauss-Seidel is sequential over rows, so the parallelism is across the nv dofs of ONE row update and
// across worlds.  The nefc x nefc matrix A is never formed: with B = J M^-1 (nefc x nv, LDS resident next to J) and the
// running acceleration q = qacc_smooth + B' f held one dof per lane, the residual of row i is
//     res_i = b_i + (A+R)_i . f = J_i . q - aref_i + R_i f_i            (one LDS read + a DPP reduction)
// and a force change delta moves q by delta * B_i                        (one LDS read + one FMA).
// LDS per world is 2 (njmax x nv) instead of njmax^2 + njmax x nv, and a sweep costs nefc x (reduction + ~25 VALU).
// M^-1 comes from the same register-resident Gauss-Jordan the CG kernel uses (lane i owns row i) and each B row gets one
// step of iterative refinement with the exact M.
//
// REG variant (njmax <= 64, the humanoid/Panda case): the sweep above is one dependent chain per row through two LDS
// round trips and a six-step DPP reduction (~670 cycles per row measured, one wave per SIMD).  With at most 64 rows the
// matrix A+R fits in VGPRs instead: one world per wavefront, lane j owns row j of A+R (64 registers), the residual
// r_j = b_j + (A+R)_j . f and the force f_j.  A row update is then: read r_i and f_i from lane i (v_readlane into SGPRs;
// the row loop is fully unrolled, every index is a constant), ~10 branch-free VALU for the projected step (every row
// kind is a clamp to [lo_i, hi_i]), and one FMA per lane (r_j += (A+R)_ji delta): no LDS access and no reduction on the
// chain.  A+R is built once per solve as J_j . B_c (own J row in registers, B rows broadcast from LDS): this is the dense
// J M^-1 J' contraction, 2 nefc^2 nv flops per world = 0.1 Mflop, a few per cent of the solve even on VALU, so MFMA
// would not pay here.
#pragma once
#include "solver.hpp"

struct PgsLayout {
  int J, B, rc, force, va, vb, col, total;
};
template <int NV4, int G>
__host__ __device__ inline PgsLayout pgs_layout(int njmax) {
  constexpr int NVR = 4 * NV4;
  constexpr int JS = (NV4 & 1) ? NVR : NVR + 4;
  const int njp = ((njmax + 15) / 16) * 16;
  PgsLayout p;
  int o = 0;
  p.J = o; o += (njp > NVR ? njp : NVR) * JS;  // also stages the dense copy of M
  p.B = o; o += njp * JS;                       // rows of J M^-1
  p.rc = o; o += njp * 4;                       // per row: aref, R, (A+R)_ii, 1 / (A+R)_ii
  p.force = o; o += njp;
  p.va = o; o += G;                             // broadcast lines for the nv-vectors other lanes read
  p.vb = o; o += G;
  p.col = o; o += 2 * (NVR > G ? NVR : G);      // Gauss-Jordan pivot row, double buffered
  p.total = ((o + 3) / 4) * 4;
  return p;
}

template <int NV4, int G, bool REG>
DEV void pgs_body(const MjhModel& m, const MjhData& d, float* smem, const Blk& b, int refresh = 1) {
  if ((int)threadIdx.x >= b.nthreads) return;
  constexpr int NVR = 4 * NV4;
  constexpr int JS = (NV4 & 1) ? NVR : NVR + 4;
  const int nv = m.nv, nC = m.nC, njmax = d.njmax, nvp = d.nv_pad;
  const PgsLayout lay = pgs_layout<NV4, G>(njmax);
  const int lig = threadIdx.x & (G - 1), gib = threadIdx.x / G;
  const int slot = b.w0 + gib;
  if (slot >= d.nworld) return;
  // (G = 64: one world per wavefront, so the world id and its row counts are scalars: branches on them are s_cbranch)
  const int w = G == 64 ? __builtin_amdgcn_readfirstlane(d.ws_order[slot]) : d.ws_order[slot];
  float* S = smem + (size_t)gib * lay.total;
  float *Jl = S + lay.J, *Bl = S + lay.B, *rcv = S + lay.rc, *eforce = S + lay.force, *va = S + lay.va, *vb = S + lay.vb, *col = S + lay.col;
  int nefc = min(d.nefc[w], njmax);
  int ne = d.ne[w], nf = d.nf[w];
  if (G == 64) {
    nefc = __builtin_amdgcn_readfirstlane(nefc);
    ne = __builtin_amdgcn_readfirstlane(ne);
    nf = __builtin_amdgcn_readfirstlane(nf);
  }
  const size_t vo = (size_t)w * nv, eo = (size_t)w * njmax;
  const bool active = lig < nv;
  const int ligr = lig < NVR ? lig : NVR - 1;

  // ---- M row of this lane into registers, rows of M^-1 by Gauss-Jordan (as the CG kernel, solver.hpp) -----------
  float mrow[NVR], h[NVR];
  {
    for (int idx = lig; idx < NVR * JS; idx += G) Jl[idx] = 0.0f;
    gsync();
    const float* Mg = d.M + (size_t)w * nC;
    for (int i = lig; i < nv; i += G) {
      const int start = m.M_rowadr[i], n = m.M_rownnz[i];
      for (int a = 0; a < n; ++a) {
        const int j = m.M_colind[start + a];
        const float v = Mg[start + a];
        Jl[i * JS + j] = v;
        Jl[j * JS + i] = v;
      }
    }
    gsync();
#pragma unroll
    for (int c = 0; c < NVR; ++c) mrow[c] = active ? Jl[ligr * JS + c] : (c == lig ? 1.0f : 0.0f);
    gsync();
  }
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
  // x = M^-1 y for the lane's component, y given per lane; explicit inverse + one refinement step with the exact M
  auto minv = [&](float y) __attribute__((always_inline)) {
    va[lig] = active ? y : 0.0f;
    gsync();
    float x = mul_row(h, va);
    vb[lig] = x;
    gsync();
    const float res = y - mul_row(mrow, vb);
    va[lig] = active ? res : 0.0f;
    gsync();
    x += mul_row(h, va);
    gsync();
    return active ? x : 0.0f;
  };
  invert_rows<NVR, G>(mrow, h, col, lig);
  const float fs = active ? d.qfrc_smooth[vo + lig] : 0.0f;
  const float qs = minv(fs);
  if (active) d.qacc_smooth[vo + lig] = qs;

  if (nefc == 0) {  // unconstrained: qacc = qacc_smooth
    if (active) {
      d.qacc[vo + lig] = qs;
      d.qfrc_constraint[vo + lig] = 0.0f;
      d.efc_Ma[vo + lig] = fs;
    }
    if (lig == 0) d.solver_niter[w] = 0;
    return;
  }

  // ---- J into LDS ------------------------------------------------------------------------------------------------
  {
    const float* Jg = d.efc_J + (size_t)w * d.njmax_pad * nvp;
    for (int r = 0; r < nefc; ++r)
      for (int c = lig; c < JS; c += G) Jl[r * JS + c] = c < nvp ? Jg[(size_t)r * nvp + c] : 0.0f;
  }
  gsync();
  // ---- B_i = M^-1 J_i' (lane = dof), refined once with the exact M -------------------------------------------------
  for (int i = 0; i < nefc; ++i) {
    const float* Ji = Jl + i * JS;
    float x = mul_row(h, Ji);
    vb[lig] = x;
    gsync();
    const float res = (active ? Ji[ligr] : 0.0f) - mul_row(mrow, vb);
    va[lig] = active ? res : 0.0f;
    gsync();
    x += mul_row(h, va);
    for (int c = lig; c < JS; c += G) Bl[i * JS + c] = (c == lig && active) ? x : 0.0f;
  }
  // ---- per-row constants, warm-start forces (row per lane) ----------------------------------------------------------
  const bool warm = !(m.disableflags & DSBL_WARMSTART);
  va[lig] = active ? d.qacc_warmstart[vo + lig] : 0.0f;
  vb[lig] = qs;
  gsync();
  float cpart = 0.0f;  // this lane's share of the dual cost of the warm-start forces:  f'b + 0.5 f'R f
  for (int r = lig; r < nefc; r += G) {
    float sAR = 0.0f, sw = 0.0f, sb = 0.0f;
#pragma unroll
    for (int c4 = 0; c4 < NV4; ++c4) {
      const float4 j4 = *reinterpret_cast<const float4*>(Jl + r * JS + 4 * c4);
      const float4 b4 = *reinterpret_cast<const float4*>(Bl + r * JS + 4 * c4);
      const float4 w4 = *reinterpret_cast<const float4*>(va + 4 * c4);
      const float4 s4 = *reinterpret_cast<const float4*>(vb + 4 * c4);
      sAR += j4.x * b4.x + j4.y * b4.y + j4.z * b4.z + j4.w * b4.w;
      sw += j4.x * w4.x + j4.y * w4.y + j4.z * w4.z + j4.w * w4.w;
      sb += j4.x * s4.x + j4.y * s4.y + j4.z * s4.z + j4.w * s4.w;
    }
    const float D = d.efc_D[eo + r], aref = d.efc_aref[eo + r];
    const float R = 1.0f / D, AR = sAR + R;
    *reinterpret_cast<float4*>(rcv + 4 * r) = make_float4(aref, R, AR, 1.0f / AR);
    float f = 0.0f;
    if (warm) {  // primal constraint update at qacc_warmstart (solver.py:1698-1822)
      int state;
      const int kind = r >= ne + nf ? 2 : (r >= ne ? 1 : 0);
      row_force(kind, sw - aref, D, nf > 0, d.efc_frictionloss + eo + r, f, state);
    }
    eforce[r] = f;
    cpart += f * ((sb - aref) + 0.5f * R * f);
  }
  gsync();
  // q - qacc_smooth = B' f (lane = dof); the A part of the cost is 0.5 (J' f) . (B' f)
  float z = 0.0f, y = 0.0f;
  if (warm) {
    for (int r = 0; r < nefc; ++r) {
      const float f = eforce[r];
      z += f * Bl[r * JS + ligr];
      y += f * Jl[r * JS + ligr];
    }
    if (!active) z = y = 0.0f;
  }
  const float cost = gsumg<G>(cpart + 0.5f * y * z);
  const bool keep = warm && !(cost > 0.0f);
  float q = qs + (keep ? z : 0.0f);
  if (!keep)
    for (int r = lig; r < nefc; r += G) eforce[r] = 0.0f;
  gsync();

  // ---- sweeps ------------------------------------------------------------------------------------------------------
  const float tolerance = bf(m.opt_tolerance, m.opt_tolerance_nb, w, 1)[0];
  const float meaninertia = bf(m.stat_meaninertia, m.stat_meaninertia_nb, w, 1)[0];
  const float rscale = 1.0f / (meaninertia * (float)max(nv, 1));
  const int maxiter = m.iterations;
  int niter = 0;
  if (REG) {
    static_assert(!REG || G == 64, "the register-resident sweep maps the 64 rows onto the 64 lanes of one wavefront");
    // lane j owns row j: its row of A+R (64 VGPRs), residual, force, and the constants the refresh needs
    const bool has = lig < nefc;
    const int rr = has ? lig : 0;
    const float4 own = *reinterpret_cast<const float4*>(rcv + 4 * rr);  // aref, R, (A+R)_jj, 1/(A+R)_jj
    float ar[64];
    {
      float jr[NVR];
#pragma unroll
      for (int c4 = 0; c4 < NV4; ++c4) {
        const float4 j4 = *reinterpret_cast<const float4*>(Jl + rr * JS + 4 * c4);
        jr[4 * c4] = j4.x; jr[4 * c4 + 1] = j4.y; jr[4 * c4 + 2] = j4.z; jr[4 * c4 + 3] = j4.w;
      }
#pragma unroll
      for (int c0 = 0; c0 < 64; c0 += 8) {
        if (c0 < nefc) {
#pragma unroll
          for (int c = c0; c < c0 + 8; ++c) {
            float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
            for (int c4 = 0; c4 < NV4; ++c4) {
              const float4 b4 = *reinterpret_cast<const float4*>(Bl + c * JS + 4 * c4);  // (rows >= nefc: discarded below)
              s0 += jr[4 * c4] * b4.x + jr[4 * c4 + 2] * b4.z;
              s1 += jr[4 * c4 + 1] * b4.y + jr[4 * c4 + 3] * b4.w;
            }
            ar[c] = (has && c < nefc) ? s0 + s1 + (c == lig ? own.y : 0.0f) : 0.0f;
          }
        } else {
#pragma unroll
          for (int c = c0; c < c0 + 8; ++c) ar[c] = 0.0f;
        }
      }
    }
    gsync();
    // per-row constants of the sweep: (lower bound, upper bound, (A+R)_ii, 1/(A+R)_ii); the projection is a clamp for
    // every row kind, and a padding row (all zeros, zero matrix column) is an exact no-op of the unrolled sweep
    if (lig < ((njmax + 15) / 16) * 16) {
      const float inf = __builtin_huge_valf();
      float lo = 0.0f, hi = has ? inf : 0.0f;
      if (has && lig < ne) lo = -inf;
      else if (has && lig < ne + nf) {
        hi = d.efc_frictionloss[eo + lig];
        lo = -hi;
      }
      *reinterpret_cast<float4*>(rcv + 4 * lig) = make_float4(lo, hi, has ? own.z : 0.0f, has ? own.w : 0.0f);
    }
    float rf = has ? eforce[rr] : 0.0f, rres = 0.0f;
    gsync();
    // residual of the owned row from scratch: r_j = J_j (qacc_smooth + B'f) - aref_j + R_j f_j.  The incremental updates
    // of the sweep accumulate round-off that the ill-conditioned A+R (redundant pyramid rows, R ~ 1e-4) amplifies, so
    // the residual is rebuilt at the start of every `refresh`-th sweep (default 8; measured: parity over 120 steps is identical
    // for 1, 4 and never, a rebuild costs 15 % of a sweep).
    auto rebuild = [&]() __attribute__((always_inline)) {
      if (has) eforce[lig] = rf;
      gsync();
      float zz = 0.0f;
      for (int r = 0; r < nefc; ++r) zz += eforce[r] * Bl[r * JS + ligr];
      vb[lig] = active ? qs + zz : 0.0f;
      gsync();
      float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
      for (int c4 = 0; c4 < NV4; ++c4) {
        const float4 j4 = *reinterpret_cast<const float4*>(Jl + rr * JS + 4 * c4);
        const float4 q4 = *reinterpret_cast<const float4*>(vb + 4 * c4);
        s0 += j4.x * q4.x + j4.z * q4.z;
        s1 += j4.y * q4.y + j4.w * q4.w;
      }
      rres = has ? (s0 + s1) - own.x + own.y * rf : 0.0f;
      gsync();
    };
    while (niter < maxiter) {
      if (refresh > 0 ? niter % refresh == 0 : niter == 0) rebuild();
      float improvement = 0.0f;
#pragma unroll
      for (int i0 = 0; i0 < 64; i0 += 8) {
        if (i0 < nefc) {
          // opaque copy of the lane id: keeps the 64 `lane == i` masks from being hoisted out of the sweep loop (they
          // would be spilled to VGPR lanes and read back with two v_readlane per row)
          int lv = lig;
          asm volatile("" : "+v"(lv));
#pragma unroll
          for (int i = i0; i < i0 + 8; ++i) {
            const float res = bcastg<64>(rres, i), fold = bcastg<64>(rf, i);
            const float4 rc = *reinterpret_cast<const float4*>(rcv + 4 * i);
            const float fn = __builtin_amdgcn_fmed3f(fold - res * rc.w, rc.x, rc.y);  // clamp to [lo, hi]: one v_med3_f32
            float delta = fn - fold;
            const float change = delta * (0.5f * delta * rc.z + res);
            const bool bad = change > 1e-10f;  // never accept a cost increase (round-off)
            delta = bad ? 0.0f : delta;
            improvement -= bad ? 0.0f : change;
            rres += ar[i] * delta;
            rf = lv == i ? fold + delta : rf;
          }
        }
      }
      ++niter;
      if (improvement * rscale < tolerance) break;
    }
    if (has) eforce[lig] = rf;
    gsync();
  } else
  while (niter < maxiter) {
    float improvement = 0.0f;
    for (int i = 0; i < nefc; ++i) {
      const float4 rc = *reinterpret_cast<const float4*>(rcv + 4 * i);
      const float fold = eforce[i];
      const float part = active ? Jl[i * JS + ligr] * q : 0.0f;
      const float res = gsumg<G>(part) - rc.x + rc.y * fold;
      float fn = fold - res * rc.w;
      if (i >= ne + nf) {
        fn = fmaxf(fn, 0.0f);
      } else if (i >= ne) {
        const float fl = d.efc_frictionloss[eo + i];
        fn = fminf(fmaxf(fn, -fl), fl);
      }
      float delta = fn - fold;
      float change = delta * (0.5f * delta * rc.z + res);
      if (change > 1e-10f) {  // never accept a cost increase (round-off)
        delta = 0.0f;
        change = 0.0f;
        fn = fold;
      }
      improvement -= change;
      q += delta * Bl[i * JS + ligr];
      if (lig == 0) eforce[i] = fn;
    }
    gsync();
    ++niter;
    if (improvement * rscale < tolerance) break;
  }

  // ---- finish: qfrc_constraint = J' f, qacc = qacc_smooth + B' f (B rows are refined against the exact M) ---------------
  float qc = 0.0f, dq = 0.0f;
  for (int r = 0; r < nefc; ++r) {
    const float f = eforce[r];
    qc += f * Jl[r * JS + ligr];
    dq += f * Bl[r * JS + ligr];
  }
  if (!active) qc = dq = 0.0f;
  if (active) {
    d.qacc[vo + lig] = qs + dq;
    d.qfrc_constraint[vo + lig] = qc;
    d.efc_Ma[vo + lig] = fs + qc;
  }
  for (int r = lig; r < nefc; r += G) {  // dual state (engine_solver.c dualState)
    const float f = eforce[r];
    int state;
    if (r < ne) state = ST_QUADRATIC;
    else if (r < ne + nf) {
      const float fl = d.efc_frictionloss[eo + r];
      state = f <= -fl ? ST_LINEARPOS : (f >= fl ? ST_LINEARNEG : ST_QUADRATIC);
    } else state = f <= 0.0f ? ST_SATISFIED : ST_QUADRATIC;
    d.efc_force[eo + r] = f;
    d.efc_state[eo + r] = state;
  }
  if (lig == 0) d.solver_niter[w] = niter;
}
