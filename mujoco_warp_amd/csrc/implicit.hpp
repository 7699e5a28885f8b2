// implicit.hpp -- the fully implicit-in-velocity integrator (IntegratorType.IMPLICIT; reference forward.py:578-600 with
// derivative.py:514-586 deriv_rne_vel and 1117 deriv_smooth_vel; MuJoCo C mj_implicit / mjd_smooth_vel / mjd_rne_vel).
//
//   (M - h dF/dv) qacc' = M qacc,   F = qfrc_smooth = qfrc_passive - qfrc_bias + qfrc_actuator
// i.e. the matrix of implicitfast (M + h damping - h gear^2 (bias_vel + gain_vel ctrl) on the diagonal) PLUS h d(qfrc_bias)/dv, the
// velocity derivative of the Coriolis / centrifugal forces, which is dense over each kinematic tree and not symmetric.  (Sign: MuJoCo C
// accumulates qDeriv -= d(bias)/dv and factors M - h qDeriv; oracle/mjref.c:ref_deriv_rne_vel is checked against finite differences of
// qfrc_bias, tests/test_oracle.py.)
//
// MI355X mapping (a correctness path for models of at most 64 dofs: one 32-lane group per world).  The reference materialises four
// [nbody, nv] arrays of spatial vectors per world in global memory and walks the tree with one launch per level and atomics.  Here the
// derivative is built column by column with LANE = COLUMN k: d(cvel_b)/dv_k is cdof[k] exactly on the bodies dof k moves (bit test on
// body_dofmask, nothing stored), d(cacc)/dv_k and d(cfrc_body)/dv_k run root -> leaf and leaf -> root in the lane's own LDS column
// (bodies are depth-first numbered: parents before children), and the projection on cdof[i] fills column k of the dense matrix.  The
// (nv x nv) system is then solved in LDS by LU without pivoting (M plus an O(h) perturbation), forward substitution fused.
// Output: Data.ws_iacc = qacc', which the integrator launch (integrate_body, mode 2) advances the state with.
#pragma once
#include "dev_common.hpp"
#include "smooth.hpp"

struct ImpLayout {
  int cdof, cdot, cvel, cinert, qvel, rhs, A, ws, total, AS;
};
__host__ __device__ inline ImpLayout imp_layout(int nv, int nbody, int G) {
  ImpLayout p;
  int o = 0;
  p.cdof = o; o += 6 * nv;
  p.cdot = o; o += 6 * nv;
  p.cvel = o; o += 6 * nbody;
  p.cinert = o; o += 10 * nbody;
  p.qvel = o; o += nv;
  p.rhs = o; o += nv;
  p.AS = nv | 1;  // odd row stride
  p.A = o; o += nv * p.AS;
  p.ws = o; o += 12 * nbody * G;  // per lane: d(cacc)/dv_k and d(cfrc)/dv_k of every body, [body * 12 + c][lane]
  p.total = ((o + 3) / 4) * 4 + 1;
  return p;
}

template <int G>
__global__ void __launch_bounds__(64) k_implicit_solve(MjhModel m, MjhData d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int nv = m.nv, nb = m.nbody, nu = m.nu, nC = m.nC;
  const ImpLayout lay = imp_layout(nv, nb, G);
  const int lig = threadIdx.x & (G - 1), gib = threadIdx.x / G;
  const int w = blockIdx.x * (blockDim.x / G) + gib;
  if (w >= d.nworld) return;
  float* S = smem + (size_t)gib * lay.total;
  float *cdof = S + lay.cdof, *cdot = S + lay.cdot, *cvel = S + lay.cvel, *cinert = S + lay.cinert, *qvel = S + lay.qvel, *rhs = S + lay.rhs,
        *A = S + lay.A, *ws = S + lay.ws;
  const int AS = lay.AS;
  const float h = bf(m.opt_timestep, m.opt_timestep_nb, w, 1)[0];
  const size_t vo = (size_t)w * nv;
  const int dsbl = m.disableflags;
  gcopy<G>(cdof, d.cdof + 6 * vo, 6 * nv, lig);
  gcopy<G>(cdot, d.cdof_dot + 6 * vo, 6 * nv, lig);
  gcopy<G>(cvel, d.cvel + (size_t)w * 6 * nb, 6 * nb, lig);
  gcopy<G>(cinert, d.cinert + (size_t)w * 10 * nb, 10 * nb, lig);
  gcopy<G>(qvel, d.qvel + vo, nv, lig);
  gcopy<G>(rhs, d.efc_Ma + vo, nv, lig);
  for (int idx = lig; idx < nv * AS; idx += G) A[idx] = 0.0f;
  gsync();
  // ---- A = M + h damping - h d(qfrc_actuator)/dv (the implicitfast matrix, dense) -------------------------------------------------
  {
    const float* Mg = d.M + (size_t)w * nC;
    const float* damp = bf(m.dof_damping, m.dof_damping_nb, w, nv);
    const float* gear = bf(m.actuator_gear, m.actuator_gear_nb, w, 6 * nu);
    for (int i = lig; i < nv; i += G) {
      const int start = m.M_rowadr[i], n = m.M_rownnz[i];
      for (int a = 0; a < n; ++a) {
        const int j = m.M_colind[start + a];
        const float v = Mg[start + a];
        A[i * AS + j] = v;
        A[j * AS + i] = v;  // (row i of the lower triangle is written by lane i only, column i of the upper triangle too: no race)
      }
    }
    gsync();
    for (int i = lig; i < nv; i += G) {
      float dg = 0.0f;
      if (!(dsbl & DSBL_DAMPER)) dg += h * damp[i];
      if (!(dsbl & DSBL_ACTUATION))
        for (int u = 0; u < nu; ++u) {
          if (m.jnt_dofadr[m.actuator_trnid[2 * u]] != i) continue;
          const float bias_vel = m.actuator_biastype[u] == 1 ? bf(m.actuator_biasprm, m.actuator_biasprm_nb, w, 10 * nu)[10 * u + 2] : 0.0f;
          const float gain_vel = m.actuator_gaintype[u] == 1 ? bf(m.actuator_gainprm, m.actuator_gainprm_nb, w, 10 * nu)[10 * u + 2] : 0.0f;
          float ctrl = d.ctrl[(size_t)w * nu + u];  // the RAW control (derivative.py:159-161)
          if (m.actuator_dyntype[u] != 0) ctrl = d.act[(size_t)w * m.na + m.actuator_actadr[u]];
          const float dv = bias_vel + gain_vel * ctrl;
          if (dv == 0.0f) continue;
          if (m.actuator_forcelimited[u]) {
            const float* fr = bf(m.actuator_forcerange, m.actuator_forcerange_nb, w, 2 * nu) + 2 * u;
            const float f = d.actuator_force[(size_t)w * nu + u];
            if (f <= fr[0] || f >= fr[1]) continue;
          }
          dg -= h * gear[6 * u] * gear[6 * u] * dv;
        }
      A[i * AS + i] += dg;
    }
    gsync();
  }
  // ---- + h d(qfrc_bias)/dv, column k per lane ---------------------------------------------------------------------------------------
  const int nw = (nv + 31) / 32;
  for (int k0 = 0; k0 < nv; k0 += G) {
    const int k = k0 + lig;
    const bool has = k < nv;
    const int kk = has ? k : 0;
    float ck[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) ck[c] = cdof[6 * kk + c];
    auto moved = [&](int b) __attribute__((always_inline)) { return (m.body_dofmask[b * nw + (kk >> 5)] >> (kk & 31)) & 1u; };
    // world body: zero derivatives
    for (int c = 0; c < 12; ++c) ws[c * G + lig] = 0.0f;
    for (int b = 1; b < nb; ++b) {
      const int pid = m.body_parentid[b];
      float ca[6], cv[6];
      const bool pm = moved(pid);
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        ca[c] = ws[(12 * pid + c) * G + lig];
        cv[c] = pm ? ck[c] : 0.0f;
      }
      int dof = m.body_dofadr[b];
      for (int j = m.body_jntadr[b]; j < m.body_jntadr[b] + m.body_jntnum[b]; ++j) {
        const int t = m.jnt_type[j];
        const int ngrp = t == JNT_FREE ? 6 : (t == JNT_BALL ? 3 : 1);
        int first = 0;
        if (t == JNT_FREE) {  // dofs 0..2: cdof_dot identically zero; they enter the velocity before the rotational group
          if (kk >= dof && kk < dof + 3)
            for (int c = 0; c < 6; ++c) cv[c] += ck[c];
          first = 3;
        }
        for (int q = first; q < ngrp; ++q) {  // d(cdof_dot_j)/dv_k = d(cvel before the group)/dv_k x cdof_j
          float dd[6];
          motion_cross(cv, cdof + 6 * (dof + q), dd);
          const float qv = qvel[dof + q];
          const bool self = dof + q == kk;
#pragma unroll
          for (int c = 0; c < 6; ++c) ca[c] += dd[c] * qv + (self ? cdot[6 * kk + c] : 0.0f);
        }
        if (kk >= dof + first && kk < dof + ngrp)
          for (int c = 0; c < 6; ++c) cv[c] += ck[c];
        dof += ngrp;
      }
      // d(cfrc_body)/dv_k = I d(cacc) + d(cvel) x* (I cvel) + cvel x* (I d(cvel))
      float t1[6], icv[6], idcv[6], x1[6], x2[6];
      inert_vec(cinert + 10 * b, ca, t1);
      inert_vec(cinert + 10 * b, cvel + 6 * b, icv);
      inert_vec(cinert + 10 * b, cv, idcv);
      motion_cross_force(cv, icv, x1);
      motion_cross_force(cvel + 6 * b, idcv, x2);
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        ws[(12 * b + c) * G + lig] = ca[c];
        ws[(12 * b + 6 + c) * G + lig] = t1[c] + x1[c] + x2[c];
      }
    }
    for (int b = nb - 1; b > 0; --b) {  // subtree sums (children have larger ids than their parents)
      const int pid = m.body_parentid[b];
      for (int c = 0; c < 6; ++c) ws[(12 * pid + 6 + c) * G + lig] += ws[(12 * b + 6 + c) * G + lig];
    }
    if (has)
      for (int i = 0; i < nv; ++i) {
        const int bi = m.dof_bodyid[i];
        float s = 0.0f;
#pragma unroll
        for (int c = 0; c < 6; ++c) s += cdof[6 * i + c] * ws[(12 * bi + 6 + c) * G + lig];
        A[i * AS + k] += h * s;
      }
    gsync();
  }
  // ---- LU without pivoting, forward substitution fused (lane = row below the pivot) ----------------------------------------------------
  for (int k = 0; k < nv; ++k) {
    const float piv = A[k * AS + k];
    const float rk = rhs[k];
    for (int i = k + 1 + lig; i < nv; i += G) {
      const float f = A[i * AS + k] / piv;
      if (f != 0.0f) {
        for (int j = k + 1; j < nv; ++j) A[i * AS + j] -= f * A[k * AS + j];
        rhs[i] -= f * rk;
      }
    }
    gsync();
  }
  for (int i = nv - 1; i >= 0; --i) {  // back substitution: x_i = (rhs_i - sum_{j > i} U_ij x_j) / U_ii, the sum spread over the lanes
    float part = 0.0f;
    for (int j = i + 1 + lig; j < nv; j += G) part += A[i * AS + j] * rhs[j];
    const float s = gsum<G>(part);
    if (lig == 0) rhs[i] = (rhs[i] - s) / A[i * AS + i];
    gsync();
  }
  for (int i = lig; i < nv; i += G) d.ws_iacc[vo + i] = rhs[i];
}
