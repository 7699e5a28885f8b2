// smooth.hpp -- position- and velocity-dependent smooth dynamics, one lane group per world.
//
//   k_fwd_pos : kinematics -> com_pos -> crb -> (factor_m)      reference smooth.py:46-226, 686-855, 1029-1098, 1183-1232
//   k_fwd_vel : com_vel -> passive -> rne -> actuation -> fwd_acceleration (factor + solve)
//               reference smooth.py:2179-2285, 1353-1515, passive.py:74-306, forward.py:680-702, 756-1149, 1255-1324
//
// MI355X mapping: the whole per-world pipeline runs out of one LDS slice (qpos row staged with a
// coalesced load, all intermediates LDS-resident, every API-visible array written back with
// group-contiguous 128 B stores).  Tree recursions are restated so that no level-by-level launches and
// no atomics are needed: subtree sums use the depth-first contiguity of MuJoCo body ids
// (body_subtreenum), chain sums (cvel, cacc) walk the dof-ancestor row of the CSR M-structure.
#pragma once
#include "dev_common.hpp"

// block-shared copy of the M-structure (read by every world of the block in factor/solve loops)
struct MStruct {
  const int* rowadr;
  const int* rownnz;
  const int* colind;
  const int* dof_tree;      // dof ids sorted by depth
  const int* dof_leveladr;  // [ndoflevel + 1]
};

template <int G>
DEV MStruct load_mstruct(const MjhModel& m, int* sh, int nthreads, bool sync = true) {
  // cooperative (whole block) copy of M_rowadr | M_rownnz | M_colind into LDS; ends with __syncthreads
  int nv = m.nv, nC = m.nC;
  for (int i = threadIdx.x; i < nv; i += nthreads) {
    sh[i] = m.M_rowadr[i];
    sh[nv + i] = m.M_rownnz[i];
  }
  for (int i = threadIdx.x; i < nC; i += nthreads) sh[2 * nv + i] = m.M_colind[i];
  int* lv = sh + 2 * nv + nC;
  for (int i = threadIdx.x; i < nv; i += nthreads) lv[i] = m.dof_tree[i];
  for (int i = threadIdx.x; i <= m.ndoflevel; i += nthreads) lv[nv + i] = m.dof_leveladr[i];
  if (sync) __syncthreads();
  return MStruct{sh, sh + nv, sh + 2 * nv, lv, lv + nv};
}
// M_rowadr | M_rownnz | M_colind | dof_tree | dof_leveladr (ndoflevel <= nv)
__host__ __device__ inline int mstruct_ints(int nv, int nC) { return ((4 * nv + 1 + nC + 3) / 4) * 4; }

// Block-shared kinematics table (fwd_pos fast path): everything the level loop of the forward kinematics looks up per body and
// per joint, staged once per workgroup so that the loop chases LDS (64 cycles a hop) instead of model tables in global memory
// (five dependent L1/L2 hops per tree level).  Valid when the staged fields are not batched per world.
//   per body  (FKB words): parent | jntadr | jntnum | mocapid | pos[3] | quat[4]
//   per joint (FKJ words): type | qposadr | pos[3] | axis[3] | qpos0[qposadr]
//   then body_tree[nbody] and body_leveladr[nbodylevel + 1]
#define FKB 11
#define FKJ 9
__host__ __device__ inline int fk_table_words(int nbody, int njnt, int nlevel) { return ((FKB * nbody + FKJ * njnt + nbody + nlevel + 1 + 3) / 4) * 4; }
DEV bool fk_table_ok(const MjhModel& m) {
  return m.body_pos_nb <= 1 && m.body_quat_nb <= 1 && m.jnt_pos_nb <= 1 && m.jnt_axis_nb <= 1 && m.qpos0_nb <= 1;
}
// cooperative (whole workgroup) fill; ends with __syncthreads
DEV void load_fk_table(const MjhModel& m, float* T, int nthreads, bool sync = true) {
  const int nbody = m.nbody, njnt = m.njnt;
  int* Ti = reinterpret_cast<int*>(T);
  for (int b = threadIdx.x; b < nbody; b += nthreads) {
    float* r = T + FKB * b;
    int* ri = Ti + FKB * b;
    ri[0] = m.body_parentid[b];
    ri[1] = m.body_jntadr[b];
    ri[2] = m.body_jntnum[b];
    ri[3] = m.nmocap ? m.body_mocapid[b] : -1;
    for (int k = 0; k < 3; ++k) r[4 + k] = m.body_pos[3 * b + k];
    for (int k = 0; k < 4; ++k) r[7 + k] = m.body_quat[4 * b + k];
  }
  float* J = T + FKB * nbody;
  for (int j = threadIdx.x; j < njnt; j += nthreads) {
    float* r = J + FKJ * j;
    int* ri = reinterpret_cast<int*>(r);
    const int qa = m.jnt_qposadr[j];
    ri[0] = m.jnt_type[j];
    ri[1] = qa;
    for (int k = 0; k < 3; ++k) r[2 + k] = m.jnt_pos[3 * j + k];
    for (int k = 0; k < 3; ++k) r[5 + k] = m.jnt_axis[3 * j + k];
    r[8] = m.qpos0[qa];
  }
  int* L = reinterpret_cast<int*>(J + FKJ * njnt);
  for (int i = threadIdx.x; i < nbody; i += nthreads) L[i] = m.body_tree[i];
  for (int i = threadIdx.x; i <= m.nbodylevel; i += nthreads) L[nbody + i] = m.body_leveladr[i];
  if (sync) __syncthreads();
}

// Block-shared model table of the stages AFTER the kinematics (round 3): every model constant com_pos, crb and the geom / site poses
// look up, staged with the kinematics table and the M-structure in ONE batch of loads at the start of the kernel.  Before, each stage
// began with its own round trip to L2 (body_ipos, geom_bodyid -> ..., body_subtreenum -> body_mass[c] in a serial loop of up to nbody
// loads, the dof_parentid chain of the mass-matrix rows ...): about twenty dependent trips of 500+ cycles in a kernel whose arithmetic
// is 2,100 instructions per wavefront -- k_fwd_pos took 30 us for 1,024 worlds (one wavefront per SIMD) and 40 us for 8,192.
// Structure of arrays, float / int words; valid when none of the staged fields is batched per world.
struct PosTab {
  const float *mass, *submass, *inertia, *ipos, *iquat, *armature, *gpos, *gquat, *spos, *squat;
  const int *subnum, *rootid, *jbody, *jdof, *jtype, *dbody, *dparent, *gbody, *sbody;
};
__host__ __device__ inline int pos_tab_words(int nbody, int njnt, int nv, int ngeom, int nsite) {
  return ((14 * nbody + 3 * njnt + 3 * nv + 8 * ngeom + 8 * nsite + 3) / 4) * 4;
}
DEV bool pos_tab_ok(const MjhModel& m) {
  return fk_table_ok(m) && m.body_mass_nb <= 1 && m.body_subtreemass_nb <= 1 && m.body_inertia_nb <= 1 && m.body_ipos_nb <= 1 && m.body_iquat_nb <= 1 &&
         m.dof_armature_nb <= 1 && m.geom_pos_nb <= 1 && m.geom_quat_nb <= 1 && m.site_pos_nb <= 1 && m.site_quat_nb <= 1;
}
// cooperative (whole workgroup) fill, no barrier: thread t stages body t, joint t, dof t, geom t, site t -- all loads independent
DEV PosTab load_pos_tab(const MjhModel& m, float* T, int nthreads) {
  const int nbody = m.nbody, njnt = m.njnt, nv = m.nv, ngeom = m.ngeom, nsite = m.nsite;
  float* mass = T;
  float* submass = mass + nbody;
  float* inertia = submass + nbody;
  float* ipos = inertia + 3 * nbody;
  float* iquat = ipos + 3 * nbody;
  int* subnum = reinterpret_cast<int*>(iquat + 4 * nbody);
  int* rootid = subnum + nbody;
  int* jbody = rootid + nbody;
  int* jdof = jbody + njnt;
  int* jtype = jdof + njnt;
  float* armature = reinterpret_cast<float*>(jtype + njnt);
  int* dbody = reinterpret_cast<int*>(armature + nv);
  int* dparent = dbody + nv;
  int* gbody = dparent + nv;
  float* gpos = reinterpret_cast<float*>(gbody + ngeom);
  float* gquat = gpos + 3 * ngeom;
  int* sbody = reinterpret_cast<int*>(gquat + 4 * ngeom);
  float* spos = reinterpret_cast<float*>(sbody + nsite);
  float* squat = spos + 3 * nsite;
  for (int b = threadIdx.x; b < nbody; b += nthreads) {
    mass[b] = m.body_mass[b];
    submass[b] = m.body_subtreemass[b];
    subnum[b] = m.body_subtreenum[b];
    rootid[b] = m.body_rootid[b];
    for (int k = 0; k < 3; ++k) inertia[3 * b + k] = m.body_inertia[3 * b + k];
    for (int k = 0; k < 3; ++k) ipos[3 * b + k] = m.body_ipos[3 * b + k];
    for (int k = 0; k < 4; ++k) iquat[4 * b + k] = m.body_iquat[4 * b + k];
  }
  for (int j = threadIdx.x; j < njnt; j += nthreads) {
    jbody[j] = m.jnt_bodyid[j];
    jdof[j] = m.jnt_dofadr[j];
    jtype[j] = m.jnt_type[j];
  }
  for (int i = threadIdx.x; i < nv; i += nthreads) {
    armature[i] = m.dof_armature[i];
    dbody[i] = m.dof_bodyid[i];
    dparent[i] = m.dof_parentid[i];
  }
  for (int g = threadIdx.x; g < ngeom; g += nthreads) {
    gbody[g] = m.geom_bodyid[g];
    for (int k = 0; k < 3; ++k) gpos[3 * g + k] = m.geom_pos[3 * g + k];
    for (int k = 0; k < 4; ++k) gquat[4 * g + k] = m.geom_quat[4 * g + k];
  }
  for (int g = threadIdx.x; g < nsite; g += nthreads) {
    sbody[g] = m.site_bodyid[g];
    for (int k = 0; k < 3; ++k) spos[3 * g + k] = m.site_pos[3 * g + k];
    for (int k = 0; k < 4; ++k) squat[4 * g + k] = m.site_quat[4 * g + k];
  }
  return PosTab{mass, submass, inertia, ipos, iquat, armature, gpos, gquat, spos, squat, subnum, rootid, jbody, jdof, jtype, dbody, dparent, gbody, sbody};
}

// sparse L'DL factorisation in LDS (reference smooth.py:1183-1232 _qLD_acc/_qLDiag_div == MuJoCo mj_factorI).
// L holds a copy of M on entry.  Row k is eliminated sequentially (leaf to root); the updates of its
// ancestor rows are spread over the lanes (one ancestor row per lane, no write conflicts).
// (Tried in round 2: one lane per (ancestor, column) pair, and a level-wise gather for the transposed solve -- the per-row chain of
// dependent LDS round trips is what costs, ~1,800 cycles per row on the G1 either way; the gather was twice as slow.)
// Several kinematic trees (round 3): M is block diagonal over them, so the s-th row from the end of EVERY tree is eliminated in the same
// step -- the chain is tree_nvmax steps long instead of nv (three humanoids: 27 instead of 81; clutter_synth: 8 instead of 136).  Inside a
// tree the order is the sequential one and no entry is shared between trees: bit-identical factors.
// It pays when the trees are shallow: a step of the sequential version costs one row (its ancestors spread over the lanes), a step of the
// tree-parallel one ceil(ntree (tree_nvmax - 1) / G) rows -- three humanoids (3 trees of 27 dofs, rows of up to 26 ancestors on 32 lanes): 81
// steps against 27 x 3, no gain (measured: integrator launch 148 -> 174 us); clutter_synth (22 trees of at most 8 dofs): 136 against 8 x 5.
template <int G>
DEV bool ld_by_tree(const MjhModel& m) {
  if (m.ntree <= 1) return false;
  const int manc = max(m.tree_nvmax - 1, 1);
  return m.nv >= 2 * m.tree_nvmax * ((m.ntree * manc + G - 1) / G);
}
template <int G>
DEV void factor_ld(const MStruct& ms, float* L, float* dinv, int nv, int lig, const MjhModel* mt = nullptr) {
  if (mt && ld_by_tree<G>(*mt)) {
    const int nt = mt->ntree, maxd = mt->tree_nvmax, manc = max(maxd - 1, 1), nitem = nt * manc;
    for (int s = 0; s < maxd; ++s) {
      for (int item = lig; item < nitem; item += G) {  // item = (tree, ancestor slot of its current row)
        const int t = item / manc, a = item - t * manc, nd = mt->tree_dofnum[t];
        if (s >= nd) continue;
        const int k = mt->tree_dofadr[t] + nd - 1 - s, start = ms.rowadr[k], n = ms.rownnz[k];
        if (a >= n - 1) continue;
        const float tt = L[start + a] / L[start + n - 1];
        const int ai = ms.rowadr[ms.colind[start + a]];
        for (int j = 0; j <= a; ++j) L[ai + j] -= L[start + j] * tt;
      }
      gsync();
      for (int item = lig; item < nitem; item += G) {
        const int t = item / manc, a = item - t * manc, nd = mt->tree_dofnum[t];
        if (s >= nd) continue;
        const int k = mt->tree_dofadr[t] + nd - 1 - s, start = ms.rowadr[k], n = ms.rownnz[k];
        const float dk = L[start + n - 1];
        if (a == 0) dinv[k] = 1.0f / dk;
        if (a < n - 1) L[start + a] = L[start + a] / dk;
      }
    }
    gsync();
    return;
  }
  for (int k = nv - 1; k >= 0; --k) {
    const int start = ms.rowadr[k], n = ms.rownnz[k], diag = start + n - 1;
    const float dk = L[diag];
    for (int a = lig; a < n - 1; a += G) {
      const float t = L[start + a] / dk;
      const int ai = ms.rowadr[ms.colind[start + a]];
      for (int j = 0; j <= a; ++j) L[ai + j] -= L[start + j] * t;
    }
    gsync();
    for (int a = lig; a < n - 1; a += G) L[start + a] = L[start + a] / dk;
    if (lig == 0) dinv[k] = 1.0f / dk;
  }
  gsync();
}

// x <- (L' D L)^-1 x in LDS (reference solve_LD smooth.py:3187 == MuJoCo mj_solveLD)
template <int G>
DEV void solve_ld(const MjhModel& m, const MStruct& ms, const float* L, const float* dinv, float* x, int nv, int lig) {
  if (ld_by_tree<G>(m)) {  // x <- L^-T x, one row of every kinematic tree per step (see factor_ld)
    const int nt = m.ntree, maxd = m.tree_nvmax, manc = max(maxd - 1, 1), nitem = nt * manc;
    for (int s = 0; s < maxd; ++s) {
      for (int item = lig; item < nitem; item += G) {
        const int t = item / manc, a = item - t * manc, nd = m.tree_dofnum[t];
        if (s >= nd) continue;
        const int k = m.tree_dofadr[t] + nd - 1 - s, start = ms.rowadr[k], n = ms.rownnz[k];
        if (a < n - 1) x[ms.colind[start + a]] -= L[start + a] * x[k];
      }
      gsync();
    }
  } else
  for (int k = nv - 1; k >= 0; --k) {  // x <- L^-T x
    const int start = ms.rowadr[k], n = ms.rownnz[k];
    const float xk = x[k];
    for (int a = lig; a < n - 1; a += G) x[ms.colind[start + a]] -= L[start + a] * xk;
    gsync();
  }
  for (int i = lig; i < nv; i += G) x[i] *= dinv[i];
  gsync();
  for (int l = 1; l < m.ndoflevel; ++l) {  // x <- L^-1 x, level by level down the dof tree
    const int beg = ms.dof_leveladr[l], end = ms.dof_leveladr[l + 1];
    for (int idx = beg + lig; idx < end; idx += G) {
      const int k = ms.dof_tree[idx];
      const int start = ms.rowadr[k], n = ms.rownnz[k];
      float s = x[k];
      for (int a = 0; a < n - 1; ++a) s -= L[start + a] * x[ms.colind[start + a]];
      x[k] = s;
    }
    gsync();
  }
}

// res = M * vec (support.py:154 mul_m); M CSR in LDS or global, vec/res in LDS
template <int G>
DEV void mul_m_ld(const MStruct& ms, const float* M, const float* vec, float* res, int nv, int lig) {
  // row part (ancestors + diagonal) ...
  for (int i = lig; i < nv; i += G) {
    const int start = ms.rowadr[i], n = ms.rownnz[i];
    float s = 0.0f;
    for (int a = 0; a < n; ++a) s += M[start + a] * vec[ms.colind[start + a]];
    res[i] = s;
  }
  gsync();
  // ... plus the symmetric column part, scattered sequentially per descendant row (deterministic)
  for (int k = 0; k < nv; ++k) {
    const int start = ms.rowadr[k], n = ms.rownnz[k];
    const float vk = vec[k];
    for (int a = lig; a < n - 1; a += G) res[ms.colind[start + a]] += M[start + a] * vk;
    gsync();
  }
}

// -h d(qfrc_actuator)/d(qvel), the diagonal of implicitfast's system matrix beside the damping (derivative.py:1117 deriv_smooth_vel,
// joint transmissions: gear^2 (bias_vel + gain_vel ctrl)), accumulated into `diag` -- an LDS line of >= nv floats, zeroed by the caller
// and fenced (gsync) before and after.  One lane per ACTUATOR (the dof-major double loop costs nv x nu dependent table loads: measured
// 40 % of the integrator kernel on the G1); contributions meet with LDS atomic adds, which is order-independent -- hence
// deterministic -- as long as no dof has more than two actuators (float addition commutes); models beyond that keep the dof-major loop.
template <int G>
DEV void actuator_vel_diag(const MjhModel& m, const MjhData& d, int w, int lig, float h, float* diag) {
  const int nu = m.nu, nv = m.nv;
  const float* gear = bf(m.actuator_gear, m.actuator_gear_nb, w, 6 * nu);
  auto act_dv = [&](int u, float& val) __attribute__((always_inline)) {
    const float bias_vel = m.actuator_biastype[u] == 1 ? bf(m.actuator_biasprm, m.actuator_biasprm_nb, w, 10 * nu)[10 * u + 2] : 0.0f;
    const float gain_vel = m.actuator_gaintype[u] == 1 ? bf(m.actuator_gainprm, m.actuator_gainprm_nb, w, 10 * nu)[10 * u + 2] : 0.0f;
    // the RAW control, not the clamped one, multiplies the velocity gain (derivative.py:159-161, mjd_actuator_vel)
    float ctrl = d.ctrl[(size_t)w * nu + u];
    if (m.actuator_dyntype[u] != 0) ctrl = d.act[(size_t)w * m.na + m.actuator_actadr[u]];
    const float dv = bias_vel + gain_vel * ctrl;
    if (dv == 0.0f) return false;
    if (m.actuator_forcelimited[u]) {
      const float* fr = bf(m.actuator_forcerange, m.actuator_forcerange_nb, w, 2 * nu) + 2 * u;
      const float f = d.actuator_force[(size_t)w * nu + u];
      if (f <= fr[0] || f >= fr[1]) return false;
    }
    val = gear[6 * u] * gear[6 * u] * dv;
    return true;
  };
  if (m.act_dof_max <= 2) {
    for (int u = lig; u < nu; u += G) {
      float val;
      if (!act_dv(u, val)) continue;
      atomicAdd(&diag[m.jnt_dofadr[m.actuator_trnid[2 * u]]], -h * val);
    }
  } else {
    for (int i = lig; i < nv; i += G) {
      float acc = 0.0f;
      for (int u = 0; u < nu; ++u) {
        float val;
        if (m.jnt_dofadr[m.actuator_trnid[2 * u]] != i || !act_dv(u, val)) continue;
        acc += val;
      }
      diag[i] -= h * acc;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
struct PosLayout {
  int qpos, xpos, xquat, xmat, xipos, ximat, xanchor, xaxis, scom, cinert, cdof, crb, M, L, dinv, total;
};
// with_factor: the POS_FACTOR phase (factor_m stage only) needs L and dinv.  M lives on top of xmat/ximat when it fits:
// both are dead (already written back) by the time crb runs.  LDS per world decides how many worlds a CU holds, i.e.
// how many dependent tree levels overlap: 1,703 -> 1,190 words for the humanoid (20 -> 28 worlds per CU).
__host__ __device__ inline PosLayout pos_layout(int nq, int nv, int nbody, int njnt, int nC, bool with_factor) {
  PosLayout p;
  int o = 0;
  p.qpos = o; o += nq;
  // xpos | xquat | xanchor | xaxis are dead (written back, last read by cdof) when crb starts: crb lives on top of them when it fits
  const int pose0 = o;
  p.xpos = o; o += 3 * nbody;
  p.xquat = o; o += 4 * nbody;
  p.xanchor = o; o += 3 * njnt;
  p.xaxis = o; o += 3 * njnt;
  const int pose_words = o - pose0;
  p.xmat = o; o += 9 * nbody;
  p.xipos = o; o += 3 * nbody;
  p.ximat = o; o += 9 * nbody;
  p.scom = o; o += 3 * nbody;
  p.cinert = o; o += 10 * nbody;
  p.cdof = o; o += 6 * nv;
  if (10 * nbody <= pose_words) {
    p.crb = pose0;
  } else {
    p.crb = o; o += 10 * nbody;
  }
  if (nC <= 21 * nbody) {
    p.M = p.xmat;  // xmat, xipos, ximat are adjacent (21 nbody words) and no longer read once com_pos is done
  } else {
    p.M = o; o += nC;
  }
  p.L = o; o += with_factor ? nC : 0;
  p.dinv = o; o += with_factor ? nv : 0;
  p.total = ((o + 3) / 4) * 4 + 1;  // odd-ish stride keeps worlds of one wave on different banks
  return p;
}

// block-shared words in front of the per-world slices: kinematics table | M-structure | table of the later stages
__host__ __device__ inline int pos_shared_words(int nv, int nC, int nbody, int njnt, int nlevel, int ngeom, int nsite) {
  return fk_table_words(nbody, njnt, nlevel) + mstruct_ints(nv, nC) + pos_tab_words(nbody, njnt, nv, ngeom, nsite);
}

enum { POS_KINEMATICS = 0, POS_COM = 1, POS_CRB = 2, POS_FACTOR = 3 };

// TAB: the model tables of every stage are staged in LDS (pos_tab_ok); otherwise the stages read the (per-world batched) model arrays
template <int G, bool TAB>
DEV void fwd_pos_impl(const MjhModel& m, const MjhData& d, int first, int last, float* smem, const Blk& b) {
  if ((int)threadIdx.x >= b.nthreads) return;
  const int nq = m.nq, nv = m.nv, nbody = m.nbody, njnt = m.njnt, nC = m.nC;
  const PosLayout lay = pos_layout(nq, nv, nbody, njnt, nC, last >= POS_FACTOR);
  const int fkw = fk_table_words(nbody, njnt, m.nbodylevel), msw = mstruct_ints(nv, nC);
  const int shared_words = pos_shared_words(nv, nC, nbody, njnt, m.nbodylevel, m.ngeom, m.nsite);
  const bool fk_fast = TAB && first <= POS_KINEMATICS;
  const int lig = threadIdx.x & (G - 1), gib = threadIdx.x / G;
  const int w = b.w0 + gib;
  const bool valid = w < d.nworld;  // (no early return: the workgroup meets again at the barriers below)
  float* S = smem + shared_words + (size_t)gib * lay.total;
  // the block-shared region: kinematics table | M-structure | table of the later stages, filled by ONE batch of loads together with
  // the world's qpos row, one barrier
  MStruct ms = MStruct{nullptr, nullptr, nullptr, nullptr, nullptr};
  PosTab pt = PosTab{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (TAB) {
    if (fk_fast) load_fk_table(m, smem, b.nthreads, false);
    ms = load_mstruct<G>(m, reinterpret_cast<int*>(smem + fkw), b.nthreads, false);
    pt = load_pos_tab(m, smem + fkw + msw, b.nthreads);
    if (fk_fast && valid) gcopy<G>(S + lay.qpos, d.qpos + (size_t)w * nq, nq, lig);
    __syncthreads();
  } else {
    ms = load_mstruct<G>(m, reinterpret_cast<int*>(smem + fkw), b.nthreads);
  }
  PhaseClock pc(1, lig);
  float *qpos = S + lay.qpos, *xpos = S + lay.xpos, *xquat = S + lay.xquat, *xmat = S + lay.xmat, *xipos = S + lay.xipos,
        *ximat = S + lay.ximat, *xanchor = S + lay.xanchor, *xaxis = S + lay.xaxis, *scom = S + lay.scom,
        *cinert = S + lay.cinert, *cdof = S + lay.cdof, *crb = S + lay.crb, *M = S + lay.M, *L = S + lay.L,
        *dinv = S + lay.dinv;

  // ---- kinematics (smooth.py:46-226) ---------------------------------------------------------------
  if (first <= POS_KINEMATICS && valid && fk_fast) {
    const float* TB = smem;
    const float* TJ = smem + FKB * nbody;
    const int* TL = reinterpret_cast<const int*>(TJ + FKJ * njnt);
    // joint-local transforms, one lane per joint (all trigonometry up front, off the level chain): lq = rotation of the joint
    // about its own axis (hinge) / the ball quaternion / identity, disp = slide displacement.  Parked in the cinert slice.
    float* lq = cinert;
    for (int j = lig; j < njnt; j += G) {
      const float* r = TJ + FKJ * j;
      const int t = __float_as_int(r[0]), qa = __float_as_int(r[1]);
      Q4 q = Q4{1, 0, 0, 0};
      float disp = 0.0f;
      if (t == JNT_HINGE) q = axis_angle_to_quat(ld3(r + 5), qpos[qa] - r[8]);
      else if (t == JNT_BALL) q = quat_normalize(ld4(qpos + qa));
      else if (t == JNT_SLIDE) disp = qpos[qa] - r[8];
      st4(lq + 5 * j, q);
      lq[5 * j + 4] = disp;
    }
    gsync();
    pc.mark(4);
  }
  if (fk_fast && first <= POS_KINEMATICS) {
    // ---- level loop, REPACKED: a tree level holds a handful of bodies (at most 4 for the humanoid), so with one 32-lane group
    // per world 7/8 of every instruction's lanes idle -- and the loop is issue-bound (four wavefronts per SIMD).  Instead LW
    // lanes per world (the widest level, rounded up to a power of two) walk the levels, for 64 / LW worlds per wavefront: the
    // same arithmetic in an eighth of the wavefront-instructions; the other wavefronts of the workgroup wait at the barrier.
    __syncthreads();  // every world's joint-local transforms are in LDS
    const float* TB = smem;
    const float* TJ = smem + FKB * nbody;
    const int* TL = reinterpret_cast<const int*>(TJ + FKJ * njnt);
    int maxw = 1;
    for (int l = 0; l < m.nbodylevel; ++l) maxw = max(maxw, TL[nbody + l + 1] - TL[nbody + l]);
    int LW = 1;
    while (LW < maxw) LW <<= 1;
    if (LW > G) LW = G;  // (wider levels: the lanes stride, as before)
    const int t = (int)threadIdx.x, g2 = t / LW, slot = t - g2 * LW;
    const int w2 = b.w0 + g2;
    if (g2 < b.nw && w2 < d.nworld) {
      float* S2 = smem + shared_words + (size_t)g2 * lay.total;
      float *qpos = S2 + lay.qpos, *xpos = S2 + lay.xpos, *xquat = S2 + lay.xquat, *xanchor = S2 + lay.xanchor, *xaxis = S2 + lay.xaxis;
      const float* lq = S2 + lay.cinert;
      for (int l = 0; l < m.nbodylevel; ++l) {
        const int beg = TL[nbody + l], end = TL[nbody + l + 1];
        for (int idx = beg + slot; idx < end; idx += LW) {
          const int b = TL[idx];
          if (b == 0) {
            st3(xpos, V3{0, 0, 0});
            st4(xquat, Q4{1, 0, 0, 0});
            continue;
          }
          const float* rb = TB + FKB * b;
          const int pid = __float_as_int(rb[0]), jntadr = __float_as_int(rb[1]), jntnum = __float_as_int(rb[2]), mid = __float_as_int(rb[3]);
          if (jntnum == 1 && __float_as_int(TJ[FKJ * jntadr]) == JNT_FREE) {
            const int qa = __float_as_int(TJ[FKJ * jntadr + 1]);
            V3 p = ld3(qpos + qa);
            Q4 q = quat_normalize(ld4(qpos + qa + 3));
            st3(xpos + 3 * b, p);
            st4(xquat + 4 * b, q);
            st3(xanchor + 3 * jntadr, p);
            st3(xaxis + 3 * jntadr, ld3(TJ + FKJ * jntadr + 5));
            continue;
          }
          Q4 pq = ld4(xquat + 4 * pid);
          // mocap bodies (children of the world without joints) take their pose from Data.mocap_* (smooth.py:104-108)
          const V3 bp = mid >= 0 ? ld3(d.mocap_pos + ((size_t)w2 * m.nmocap + mid) * 3) : ld3(rb + 4);
          const Q4 bq = mid >= 0 ? ld4(d.mocap_quat + ((size_t)w2 * m.nmocap + mid) * 4) : ld4(rb + 7);
          V3 pos = rot_vec_quat(bp, pq) + ld3(xpos + 3 * pid);
          Q4 quat = mul_quat(pq, bq);
          for (int j = jntadr; j < jntadr + jntnum; ++j) {
            const float* r = TJ + FKJ * j;
            const int tj = __float_as_int(r[0]);
            const V3 jp = ld3(r + 2), ja = ld3(r + 5);
            V3 anchor = rot_vec_quat(jp, quat) + pos;
            V3 axis = rot_vec_quat(ja, quat);
            if (tj == JNT_SLIDE) {
              pos = pos + axis * lq[5 * j + 4];
            } else if (tj == JNT_BALL || tj == JNT_HINGE) {
              quat = mul_quat(quat, ld4(lq + 5 * j));
              pos = anchor - rot_vec_quat(jp, quat);
            }
            st3(xanchor + 3 * j, anchor);
            st3(xaxis + 3 * j, axis);
          }
          st3(xpos + 3 * b, pos);
          st4(xquat + 4 * b, quat_normalize(quat));
        }
        gsync();  // (the LW lanes of a world sit in one wavefront: 64 % LW == 0)
      }
    }
  } else if (first <= POS_KINEMATICS && valid) {
    gcopy<G>(qpos, d.qpos + (size_t)w * nq, nq, lig);
    gsync();
    const float* qpos0 = bf(m.qpos0, m.qpos0_nb, w, nq);
    const float* body_pos = bf(m.body_pos, m.body_pos_nb, w, 3 * nbody);
    const float* body_quat = bf(m.body_quat, m.body_quat_nb, w, 4 * nbody);
    const float* jnt_pos = bf(m.jnt_pos, m.jnt_pos_nb, w, 3 * njnt);
    const float* jnt_axis = bf(m.jnt_axis, m.jnt_axis_nb, w, 3 * njnt);
    for (int l = 0; l < m.nbodylevel; ++l) {
      const int beg = m.body_leveladr[l], end = m.body_leveladr[l + 1];
      for (int idx = beg + lig; idx < end; idx += G) {
        const int b = m.body_tree[idx];
        if (b == 0) {
          st3(xpos, V3{0, 0, 0});
          st4(xquat, Q4{1, 0, 0, 0});
          continue;
        }
        const int pid = m.body_parentid[b], jntadr = m.body_jntadr[b], jntnum = m.body_jntnum[b];
        if (jntnum == 1 && m.jnt_type[jntadr] == JNT_FREE) {
          const int qa = m.jnt_qposadr[jntadr];
          V3 p = ld3(qpos + qa);
          Q4 q = quat_normalize(ld4(qpos + qa + 3));
          st3(xpos + 3 * b, p);
          st4(xquat + 4 * b, q);
          st3(xanchor + 3 * jntadr, p);
          st3(xaxis + 3 * jntadr, ld3(jnt_axis + 3 * jntadr));
          continue;
        }
        Q4 pq = ld4(xquat + 4 * pid);
        // mocap bodies (children of the world without joints) take their pose from Data.mocap_* (smooth.py:104-108)
        const int mid = m.nmocap ? m.body_mocapid[b] : -1;
        const V3 bp = mid >= 0 ? ld3(d.mocap_pos + ((size_t)w * m.nmocap + mid) * 3) : ld3(body_pos + 3 * b);
        const Q4 bq = mid >= 0 ? ld4(d.mocap_quat + ((size_t)w * m.nmocap + mid) * 4) : ld4(body_quat + 4 * b);
        V3 pos = rot_vec_quat(bp, pq) + ld3(xpos + 3 * pid);
        Q4 quat = mul_quat(pq, bq);
        for (int j = jntadr; j < jntadr + jntnum; ++j) {
          const int qa = m.jnt_qposadr[j], t = m.jnt_type[j];
          V3 jp = ld3(jnt_pos + 3 * j), ja = ld3(jnt_axis + 3 * j);
          V3 anchor = rot_vec_quat(jp, quat) + pos;
          V3 axis = rot_vec_quat(ja, quat);
          if (t == JNT_BALL) {
            quat = mul_quat(quat, quat_normalize(ld4(qpos + qa)));
            pos = anchor - rot_vec_quat(jp, quat);
          } else if (t == JNT_SLIDE) {
            pos = pos + axis * (qpos[qa] - qpos0[qa]);
          } else if (t == JNT_HINGE) {
            quat = mul_quat(quat, axis_angle_to_quat(ja, qpos[qa] - qpos0[qa]));
            pos = anchor - rot_vec_quat(jp, quat);
          }
          st3(xanchor + 3 * j, anchor);
          st3(xaxis + 3 * j, axis);
        }
        st3(xpos + 3 * b, pos);
        st4(xquat + 4 * b, quat_normalize(quat));
      }
      gsync();
    }
  }
  if (fk_fast) __syncthreads();  // (the level loop wrote poses into the slices of other wavefronts' worlds)
  if (!valid) return;
  pc.mark(5);
  // model constants: from the LDS table (TAB) or from the model arrays (batched fields)
  const float* body_ipos = TAB ? pt.ipos : bf(m.body_ipos, m.body_ipos_nb, w, 3 * nbody);
  const float* body_iquat = TAB ? pt.iquat : bf(m.body_iquat, m.body_iquat_nb, w, 4 * nbody);
  const float* geom_pos = TAB ? pt.gpos : bf(m.geom_pos, m.geom_pos_nb, w, 3 * m.ngeom);
  const float* geom_quat = TAB ? pt.gquat : bf(m.geom_quat, m.geom_quat_nb, w, 4 * m.ngeom);
  const float* site_pos = TAB ? pt.spos : bf(m.site_pos, m.site_pos_nb, w, 3 * m.nsite);
  const float* site_quat = TAB ? pt.squat : bf(m.site_quat, m.site_quat_nb, w, 4 * m.nsite);
  const float* body_mass = TAB ? pt.mass : bf(m.body_mass, m.body_mass_nb, w, nbody);
  const float* body_subtreemass = TAB ? pt.submass : bf(m.body_subtreemass, m.body_subtreemass_nb, w, nbody);
  const float* body_inertia = TAB ? pt.inertia : bf(m.body_inertia, m.body_inertia_nb, w, 3 * nbody);
  const float* armature = TAB ? pt.armature : bf(m.dof_armature, m.dof_armature_nb, w, nv);
  const int* geom_bodyid = TAB ? pt.gbody : m.geom_bodyid;
  const int* site_bodyid = TAB ? pt.sbody : m.site_bodyid;
  const int* body_subtreenum = TAB ? pt.subnum : m.body_subtreenum;
  const int* body_rootid = TAB ? pt.rootid : m.body_rootid;
  const int* jnt_bodyid = TAB ? pt.jbody : m.jnt_bodyid;
  const int* jnt_dofadr = TAB ? pt.jdof : m.jnt_dofadr;
  const int* jnt_type = TAB ? pt.jtype : m.jnt_type;
  const int* dof_bodyid = TAB ? pt.dbody : m.dof_bodyid;
  const int* dof_parentid = TAB ? pt.dparent : m.dof_parentid;
  if (first <= POS_KINEMATICS) {
    for (int b = lig; b < nbody; b += G) {
      Q4 q = ld4(xquat + 4 * b);
      quat_to_mat(q, xmat + 9 * b);
      st3(xipos + 3 * b, ld3(xpos + 3 * b) + rot_vec_quat(ld3(body_ipos + 3 * b), q));
      quat_to_mat(mul_quat(q, ld4(body_iquat + 4 * b)), ximat + 9 * b);
    }
    pc.mark(6);
    {  // geoms and sites go straight to HBM (consumed by the collision kernel)
      for (int g = lig; g < m.ngeom; g += G) {
        const int b = geom_bodyid[g];
        Q4 q = ld4(xquat + 4 * b);
        float mat[9];
        st3(d.geom_xpos + ((size_t)w * m.ngeom + g) * 3, ld3(xpos + 3 * b) + rot_vec_quat(ld3(geom_pos + 3 * g), q));
        quat_to_mat(mul_quat(q, ld4(geom_quat + 4 * g)), mat);
        float* out = d.geom_xmat + ((size_t)w * m.ngeom + g) * 9;
        for (int k = 0; k < 9; ++k) out[k] = mat[k];
      }
      for (int s = lig; s < m.nsite; s += G) {
        const int b = site_bodyid[s];
        Q4 q = ld4(xquat + 4 * b);
        float mat[9];
        st3(d.site_xpos + ((size_t)w * m.nsite + s) * 3, ld3(xpos + 3 * b) + rot_vec_quat(ld3(site_pos + 3 * s), q));
        quat_to_mat(mul_quat(q, ld4(site_quat + 4 * s)), mat);
        float* out = d.site_xmat + ((size_t)w * m.nsite + s) * 9;
        for (int k = 0; k < 9; ++k) out[k] = mat[k];
      }
    }
    gsync();
    pc.mark(7);
    gcopy<G>(d.xpos + (size_t)w * 3 * nbody, xpos, 3 * nbody, lig);
    gcopy<G>(d.xquat + (size_t)w * 4 * nbody, xquat, 4 * nbody, lig);
    gcopy<G>(d.xmat + (size_t)w * 9 * nbody, xmat, 9 * nbody, lig);
    gcopy<G>(d.xipos + (size_t)w * 3 * nbody, xipos, 3 * nbody, lig);
    gcopy<G>(d.ximat + (size_t)w * 9 * nbody, ximat, 9 * nbody, lig);
    gcopy<G>(d.xanchor + (size_t)w * 3 * njnt, xanchor, 3 * njnt, lig);
    gcopy<G>(d.xaxis + (size_t)w * 3 * njnt, xaxis, 3 * njnt, lig);
  }
  pc.mark(0);
  if (last < POS_COM) return;

  // ---- com_pos (smooth.py:686-822) -------------------------------------------------------------------
  if (first <= POS_COM) {
    if (first == POS_COM) {
      gcopy<G>(xmat, d.xmat + (size_t)w * 9 * nbody, 9 * nbody, lig);
      gcopy<G>(xipos, d.xipos + (size_t)w * 3 * nbody, 3 * nbody, lig);
      gcopy<G>(ximat, d.ximat + (size_t)w * 9 * nbody, 9 * nbody, lig);
      gcopy<G>(xanchor, d.xanchor + (size_t)w * 3 * njnt, 3 * njnt, lig);
      gcopy<G>(xaxis, d.xaxis + (size_t)w * 3 * njnt, 3 * njnt, lig);
      gsync();
    }
    for (int b = lig; b < nbody; b += G) {  // subtree = contiguous id range (depth-first numbering)
      V3 s = V3{0, 0, 0};
      const int end = b + body_subtreenum[b];
      for (int c = b; c < end; ++c) s = s + ld3(xipos + 3 * c) * body_mass[c];
      const float mass = body_subtreemass[b];
      if (mass != 0.0f) s = s * (1.0f / mass);
      st3(scom + 3 * b, s);
    }
    gsync();
    for (int b = lig; b < nbody; b += G) {  // _cinert smooth.py:733
      const float* mat = ximat + 9 * b;
      V3 in = ld3(body_inertia + 3 * b);
      const float mass = body_mass[b];
      V3 dif = ld3(xipos + 3 * b) - ld3(scom + 3 * body_rootid[b]);
      float t00 = mat[0] * in.x * mat[0] + mat[1] * in.y * mat[1] + mat[2] * in.z * mat[2];
      float t11 = mat[3] * in.x * mat[3] + mat[4] * in.y * mat[4] + mat[5] * in.z * mat[5];
      float t22 = mat[6] * in.x * mat[6] + mat[7] * in.y * mat[7] + mat[8] * in.z * mat[8];
      float t01 = mat[0] * in.x * mat[3] + mat[1] * in.y * mat[4] + mat[2] * in.z * mat[5];
      float t02 = mat[0] * in.x * mat[6] + mat[1] * in.y * mat[7] + mat[2] * in.z * mat[8];
      float t12 = mat[3] * in.x * mat[6] + mat[4] * in.y * mat[7] + mat[5] * in.z * mat[8];
      float* r = cinert + 10 * b;
      r[0] = t00 + mass * (dif.y * dif.y + dif.z * dif.z);
      r[1] = t11 + mass * (dif.x * dif.x + dif.z * dif.z);
      r[2] = t22 + mass * (dif.x * dif.x + dif.y * dif.y);
      r[3] = t01 - mass * dif.x * dif.y;
      r[4] = t02 - mass * dif.x * dif.z;
      r[5] = t12 - mass * dif.y * dif.z;
      r[6] = mass * dif.x;
      r[7] = mass * dif.y;
      r[8] = mass * dif.z;
      r[9] = mass;
    }
    for (int j = lig; j < njnt; j += G) {  // _cdof smooth.py:779
      const int b = jnt_bodyid[j], t = jnt_type[j];
      int dof = jnt_dofadr[j];
      const float* xm = xmat + 9 * b;
      V3 off = ld3(scom + 3 * body_rootid[b]) - ld3(xanchor + 3 * j);
      if (t == JNT_FREE || t == JNT_BALL) {
        if (t == JNT_FREE) {
          for (int k = 0; k < 3; ++k) {
            float* c = cdof + 6 * (dof + k);
            c[0] = c[1] = c[2] = 0.0f;
            c[3] = k == 0 ? 1.0f : 0.0f;
            c[4] = k == 1 ? 1.0f : 0.0f;
            c[5] = k == 2 ? 1.0f : 0.0f;
          }
          dof += 3;
        }
        for (int k = 0; k < 3; ++k) {
          V3 ax = V3{xm[k], xm[3 + k], xm[6 + k]};
          st3(cdof + 6 * (dof + k), ax);
          st3(cdof + 6 * (dof + k) + 3, cross(ax, off));
        }
      } else if (t == JNT_SLIDE) {
        st3(cdof + 6 * dof, V3{0, 0, 0});
        st3(cdof + 6 * dof + 3, ld3(xaxis + 3 * j));
      } else {
        V3 ax = ld3(xaxis + 3 * j);
        st3(cdof + 6 * dof, ax);
        st3(cdof + 6 * dof + 3, cross(ax, off));
      }
    }
    gsync();
    gcopy<G>(d.subtree_com + (size_t)w * 3 * nbody, scom, 3 * nbody, lig);
    gcopy<G>(d.cinert + (size_t)w * 10 * nbody, cinert, 10 * nbody, lig);
    gcopy<G>(d.cdof + (size_t)w * 6 * nv, cdof, 6 * nv, lig);
  }
  pc.mark(1);
  if (last < POS_CRB) return;

  // ---- crb (smooth.py:1029-1098) -----------------------------------------------------------------------
  if (first <= POS_CRB) {
    if (first == POS_CRB) {
      gcopy<G>(cinert, d.cinert + (size_t)w * 10 * nbody, 10 * nbody, lig);
      gcopy<G>(cdof, d.cdof + (size_t)w * 6 * nv, 6 * nv, lig);
      gsync();
    }
    for (int b = lig; b < nbody; b += G) {
      float acc[10];
      for (int k = 0; k < 10; ++k) acc[k] = cinert[10 * b + k];
      if (b > 0) {
        const int end = b + body_subtreenum[b];
        for (int c = b + 1; c < end; ++c)
          for (int k = 0; k < 10; ++k) acc[k] += cinert[10 * c + k];
      }
      for (int k = 0; k < 10; ++k) crb[10 * b + k] = acc[k];
    }
    gsync();
    for (int i = lig; i < nv; i += G) {  // _M smooth.py:1048
      int adr = ms.rowadr[i] + ms.rownnz[i] - 1;
      float buf[6], ci[6];
      for (int k = 0; k < 6; ++k) ci[k] = cdof[6 * i + k];
      inert_vec(crb + 10 * dof_bodyid[i], ci, buf);
      int j = i;
      float arm = armature[i];
      while (j >= 0) {
        float s = 0.0f;
        for (int k = 0; k < 6; ++k) s += cdof[6 * j + k] * buf[k];
        M[adr] = s + arm;
        arm = 0.0f;
        --adr;
        j = dof_parentid[j];
      }
    }
    gsync();
    gcopy<G>(d.crb + (size_t)w * 10 * nbody, crb, 10 * nbody, lig);
    gcopy<G>(d.M + (size_t)w * nC, M, nC, lig);
  }
  pc.mark(2);
  if (last < POS_FACTOR) return;

  // ---- factor_m (smooth.py:1183-1232) -----------------------------------------------------------------
  if (first == POS_FACTOR) gcopy<G>(L, d.M + (size_t)w * nC, nC, lig);
  else gcopy<G>(L, M, nC, lig);
  gsync();
  factor_ld<G>(ms, L, dinv, nv, lig, &m);
  gcopy<G>(d.qLD + (size_t)w * nC, L, nC, lig);
  gcopy<G>(d.qLDiagInv + (size_t)w * nv, dinv, nv, lig);
  pc.mark(3);
}
template <int G>
DEV void fwd_pos_body(const MjhModel& m, const MjhData& d, int first, int last, float* smem, const Blk& b) {
  if (pos_tab_ok(m)) fwd_pos_impl<G, true>(m, d, first, last, smem, b);
  else fwd_pos_impl<G, false>(m, d, first, last, smem, b);
}

// ---------------------------------------------------------------------------------------------------
struct VelLayout {
  int qpos, qvel, cdof, cinert, cvel, cdof_dot, cacc, cfrc, cfi, uforce, fspring, fdamper, fgrav, fpassive, fbias, factuator, x, total;
};
__host__ __device__ inline VelLayout vel_layout(int nq, int nv, int nbody, int nC, int nu) {
  VelLayout p;
  int o = 0;
  p.qpos = o; o += nq;
  p.qvel = o; o += nv;
  p.cdof = o; o += 6 * nv;
  p.cinert = o; o += 10 * nbody;
  p.cvel = o; o += 6 * nbody;
  p.cdof_dot = o; o += 6 * nv;
  p.cacc = o; o += 6 * nbody;
  p.cfrc = o; o += 6 * nbody;
  p.cfi = o; o += 6 * nbody;
  p.uforce = o; o += nu;
  p.fspring = o; o += nv;
  p.fdamper = o; o += nv;
  p.fgrav = o; o += nv;
  p.fpassive = o; o += nv;
  p.fbias = o; o += nv;
  p.factuator = o; o += nv;
  p.x = o; o += nv;
  p.total = ((o + 3) / 4) * 4 + 1;
  return p;
}

enum { VEL_COMVEL = 0, VEL_PASSIVE = 1, VEL_RNE = 2, VEL_ACTUATION = 3, VEL_ACCEL = 4 };

// J^T f contribution of a Cartesian wrench applied at `point` on body b to dof i (support.py:259-322, 488-533)
DEV float jac_dot(const MjhModel& m, const float* cdof_i, V3 offset, V3 force, V3 torque) {
  V3 ang = ld3(cdof_i), lin = ld3(cdof_i + 3);
  return dot(lin + cross(ang, offset), force) + dot(ang, torque);
}

template <int G>
DEV void fwd_vel_body(const MjhModel& m, const MjhData& d, int first, int last, float* smem, const Blk& b) {
  if ((int)threadIdx.x >= b.nthreads) return;
  const int nq = m.nq, nv = m.nv, nbody = m.nbody, njnt = m.njnt, nC = m.nC, nu = m.nu;
  const VelLayout lay = vel_layout(nq, nv, nbody, nC, nu);
  int* shi = reinterpret_cast<int*>(smem);
  const MStruct ms = load_mstruct<G>(m, shi, b.nthreads);
  const int lig = threadIdx.x & (G - 1), gib = threadIdx.x / G;
  const int w = b.w0 + gib;
  if (w >= d.nworld) return;
  float* S = smem + mstruct_ints(nv, nC) + (size_t)gib * lay.total;
  float *qpos = S + lay.qpos, *qvel = S + lay.qvel, *cdof = S + lay.cdof, *cinert = S + lay.cinert, *cvel = S + lay.cvel,
        *cdof_dot = S + lay.cdof_dot, *cacc = S + lay.cacc, *cfrc = S + lay.cfrc, *cfi = S + lay.cfi, *uforce = S + lay.uforce,
        *fspring = S + lay.fspring,
        *fdamper = S + lay.fdamper, *fgrav = S + lay.fgrav, *fpassive = S + lay.fpassive, *fbias = S + lay.fbias,
        *factuator = S + lay.factuator, *x = S + lay.x;
  const int dsbl = m.disableflags;
  PhaseClock pc(4, lig);

  gcopy<G>(qpos, d.qpos + (size_t)w * nq, nq, lig);
  gcopy<G>(qvel, d.qvel + (size_t)w * nv, nv, lig);
  gcopy<G>(cdof, d.cdof + (size_t)w * 6 * nv, 6 * nv, lig);
  if (first <= VEL_RNE && last >= VEL_RNE) gcopy<G>(cinert, d.cinert + (size_t)w * 10 * nbody, 10 * nbody, lig);
  gsync();
  pc.mark(0);

  // ---- com_vel (smooth.py:2179-2258) + actuator_velocity (forward.py:680-702) ---------------------------
  if (first <= VEL_COMVEL) {
    // cdof_dot[i] = cvel_before(i) x cdof[i]; cvel_before = sum over dof-ancestors outside i's own
    // rotational triple, accumulated root -> leaf exactly like the sequential recursion
    for (int i = lig; i < nv; i += G) {
      const int start = ms.rowadr[i], n = ms.rownnz[i];
      const int grp = m.dof_grpadr[i];
      const int jt = m.jnt_type[m.dof_jntid[i]];
      float cv[6] = {0, 0, 0, 0, 0, 0};
      for (int a = 0; a < n - 1; ++a) {
        const int j = ms.colind[start + a];
        if (j >= grp) break;
        const float qv = qvel[j];
        for (int k = 0; k < 6; ++k) cv[k] += cdof[6 * j + k] * qv;
      }
      float r[6];
      const bool free_trans = (jt == JNT_FREE) && (i - m.jnt_dofadr[m.dof_jntid[i]] < 3);
      if (free_trans) {
        for (int k = 0; k < 6; ++k) r[k] = 0.0f;
      } else {
        float ci[6];
        for (int k = 0; k < 6; ++k) ci[k] = cdof[6 * i + k];
        motion_cross(cv, ci, r);
      }
      for (int k = 0; k < 6; ++k) cdof_dot[6 * i + k] = r[k];
    }
    for (int b = lig; b < nbody; b += G) {
      float cv[6] = {0, 0, 0, 0, 0, 0};
      const int ld = m.body_lastdof[b];
      if (ld >= 0) {
        const int start = ms.rowadr[ld], n = ms.rownnz[ld];
        for (int a = 0; a < n; ++a) {
          const int j = ms.colind[start + a];
          const float qv = qvel[j];
          for (int k = 0; k < 6; ++k) cv[k] += cdof[6 * j + k] * qv;
        }
      }
      for (int k = 0; k < 6; ++k) cvel[6 * b + k] = cv[k];
    }
    const float* gear = bf(m.actuator_gear, m.actuator_gear_nb, w, 6 * nu);
    for (int u = lig; u < nu; u += G)
      d.actuator_velocity[(size_t)w * nu + u] = gear[6 * u] * qvel[m.jnt_dofadr[m.actuator_trnid[2 * u]]];
    gsync();
    gcopy<G>(d.cvel + (size_t)w * 6 * nbody, cvel, 6 * nbody, lig);
    gcopy<G>(d.cdof_dot + (size_t)w * 6 * nv, cdof_dot, 6 * nv, lig);
  }
  pc.mark(1);
  if (last < VEL_PASSIVE) return;

  // ---- passive (passive.py:74-210 spring/damper, 275-306 gravcomp, 631-668 sum) --------------------------
  if (first <= VEL_PASSIVE) {
    const float* stiff = bf(m.jnt_stiffness, m.jnt_stiffness_nb, w, njnt);
    const float* damp = bf(m.dof_damping, m.dof_damping_nb, w, nv);
    const float* qspring = bf(m.qpos_spring, m.qpos_spring_nb, w, nq);
    for (int i = lig; i < nv; i += G) {
      fspring[i] = 0.0f;
      fgrav[i] = 0.0f;
      fdamper[i] = (dsbl & DSBL_DAMPER) ? 0.0f : -damp[i] * qvel[i];
    }
    gsync();
    if (!(dsbl & DSBL_SPRING)) {
      for (int j = lig; j < njnt; j += G) {
        const float k = stiff[j];
        if (k == 0.0f) continue;
        const int dof = m.jnt_dofadr[j], qa = m.jnt_qposadr[j], t = m.jnt_type[j];
        if (t == JNT_FREE) {
          for (int c = 0; c < 3; ++c) fspring[dof + c] = -k * (qpos[qa + c] - qspring[qa + c]);
          V3 dif = quat_sub(quat_normalize(ld4(qpos + qa + 3)), ld4(qspring + qa + 3));
          st3(fspring + dof + 3, dif * (-k));
        } else if (t == JNT_BALL) {
          V3 dif = quat_sub(quat_normalize(ld4(qpos + qa)), ld4(qspring + qa));
          st3(fspring + dof, dif * (-k));
        } else {
          fspring[dof] = -k * (qpos[qa] - qspring[qa]);
        }
      }
    }
    if (!(dsbl & DSBL_GRAVITY)) {
      const float* gcomp = bf(m.body_gravcomp, m.body_gravcomp_nb, w, nbody);
      const float* mass = bf(m.body_mass, m.body_mass_nb, w, nbody);
      const float* grav = bf(m.opt_gravity, m.opt_gravity_nb, w, 3);
      const int nw = (nv + 31) / 32;
      for (int b = 1; b < nbody; ++b) {
        const float gc = gcomp[b];
        if (gc == 0.0f) continue;
        V3 force = ld3(grav) * (-mass[b] * gc);
        V3 off = ld3(d.xipos + ((size_t)w * nbody + b) * 3) - ld3(d.subtree_com + ((size_t)w * nbody + m.body_rootid[b]) * 3);
        for (int i = lig; i < nv; i += G)
          if (m.body_dofmask[b * nw + (i >> 5)] & (1u << (i & 31))) fgrav[i] += jac_dot(m, cdof + 6 * i, off, force, V3{0, 0, 0});
      }
    }
    gsync();
    // (passive.py:631-668: gravity compensation is passive unless the joint routes it through its actuators)
    for (int i = lig; i < nv; i += G) fpassive[i] = fspring[i] + fdamper[i] + (m.jnt_actgravcomp[m.dof_jntid[i]] ? 0.0f : fgrav[i]);
    gsync();
    gcopy<G>(d.qfrc_spring + (size_t)w * nv, fspring, nv, lig);
    gcopy<G>(d.qfrc_damper + (size_t)w * nv, fdamper, nv, lig);
    gcopy<G>(d.qfrc_gravcomp + (size_t)w * nv, fgrav, nv, lig);
    gcopy<G>(d.qfrc_passive + (size_t)w * nv, fpassive, nv, lig);
  }
  pc.mark(2);
  if (last < VEL_RNE) return;

  // ---- rne (smooth.py:1353-1515) -------------------------------------------------------------------------
  if (first <= VEL_RNE) {
    if (first == VEL_RNE) {
      gcopy<G>(cvel, d.cvel + (size_t)w * 6 * nbody, 6 * nbody, lig);
      gcopy<G>(cdof_dot, d.cdof_dot + (size_t)w * 6 * nv, 6 * nv, lig);
      gsync();
    }
    const float* grav = bf(m.opt_gravity, m.opt_gravity_nb, w, 3);
    for (int b = lig; b < nbody; b += G) {
      float ca[6] = {0, 0, 0, 0, 0, 0};
      if (!(dsbl & DSBL_GRAVITY)) {
        ca[3] = -grav[0];
        ca[4] = -grav[1];
        ca[5] = -grav[2];
      }
      const int ld = m.body_lastdof[b];
      if (ld >= 0) {
        const int start = ms.rowadr[ld], n = ms.rownnz[ld];
        for (int a = 0; a < n; ++a) {
          const int j = ms.colind[start + a];
          const float qv = qvel[j];
          for (int k = 0; k < 6; ++k) ca[k] += cdof_dot[6 * j + k] * qv;
        }
      }
      for (int k = 0; k < 6; ++k) cacc[6 * b + k] = ca[k];
      float f[6] = {0, 0, 0, 0, 0, 0};
      if (b > 0) {
        float cv[6], iv[6], f1[6], f2[6];
        for (int k = 0; k < 6; ++k) cv[k] = cvel[6 * b + k];
        inert_vec(cinert + 10 * b, ca, f1);
        inert_vec(cinert + 10 * b, cv, iv);
        motion_cross_force(cv, iv, f2);
        for (int k = 0; k < 6; ++k) f[k] = f1[k] + f2[k];
      }
      for (int k = 0; k < 6; ++k) cfrc[6 * b + k] = f[k];
    }
    gsync();
    // backward accumulation (smooth.py:1459) restated as a subtree-range sum over depth-first body ids
    for (int b = lig; b < nbody; b += G) {
      float acc[6] = {0, 0, 0, 0, 0, 0};
      const int end = b + m.body_subtreenum[b];
      for (int c = (b == 0 ? 1 : b); c < end; ++c)
        for (int k = 0; k < 6; ++k) acc[k] += cfrc[6 * c + k];
      for (int k = 0; k < 6; ++k) cfi[6 * b + k] = acc[k];
    }
    gsync();
    for (int i = lig; i < nv; i += G) {
      const int b = m.dof_bodyid[i];
      float s = 0.0f;
      for (int k = 0; k < 6; ++k) s += cdof[6 * i + k] * cfi[6 * b + k];
      fbias[i] = s;
    }
    gsync();
    gcopy<G>(d.cfrc_int + (size_t)w * 6 * nbody, cfi, 6 * nbody, lig);
    gcopy<G>(d.cacc + (size_t)w * 6 * nbody, cacc, 6 * nbody, lig);
    gcopy<G>(d.qfrc_bias + (size_t)w * nv, fbias, nv, lig);
  }
  pc.mark(3);
  if (last < VEL_ACTUATION) return;

  // ---- fwd_actuation (forward.py:756-1149; NONE/INTEGRATOR/FILTER dyn, FIXED/AFFINE gain, NONE/AFFINE bias) --
  if (first <= VEL_ACTUATION) {
    const float* gear = bf(m.actuator_gear, m.actuator_gear_nb, w, 6 * nu);
    for (int u = lig; u < nu; u += G) {
      float force = 0.0f;
      if (!(dsbl & DSBL_ACTUATION)) {
        float ctrl = d.ctrl[(size_t)w * nu + u];
        if (m.actuator_ctrllimited[u] && !(dsbl & DSBL_CLAMPCTRL)) {
          const float* cr = bf(m.actuator_ctrlrange, m.actuator_ctrlrange_nb, w, 2 * nu) + 2 * u;
          ctrl = clampf(ctrl, cr[0], cr[1]);
        }
        float ctrl_act = ctrl;
        const int dyn = m.actuator_dyntype[u];
        if (dyn != 0) {
          const int adr = m.actuator_actadr[u];
          const float act = d.act[(size_t)w * m.na + adr];
          float act_dot = 0.0f;
          if (dyn == 1) act_dot = ctrl;
          else if (dyn == 2 || dyn == 3) act_dot = (ctrl - act) / fmaxf(MJ_MINVAL, bf(m.actuator_dynprm, m.actuator_dynprm_nb, w, 10 * nu)[10 * u]);
          d.act_dot[(size_t)w * m.na + adr] = act_dot;
          ctrl_act = act;
        }
        const int jid = m.actuator_trnid[2 * u];
        const float length = qpos[m.jnt_qposadr[jid]] * gear[6 * u];
        const float velocity = gear[6 * u] * qvel[m.jnt_dofadr[jid]];
        const float* gp = bf(m.actuator_gainprm, m.actuator_gainprm_nb, w, 10 * nu) + 10 * u;
        const float* bp = bf(m.actuator_biasprm, m.actuator_biasprm_nb, w, 10 * nu) + 10 * u;
        const float gain = m.actuator_gaintype[u] == 0 ? gp[0] : gp[0] + gp[1] * length + gp[2] * velocity;
        const float bias = m.actuator_biastype[u] == 0 ? 0.0f : bp[0] + bp[1] * length + bp[2] * velocity;
        force = gain * ctrl_act + bias;
        if (m.actuator_forcelimited[u]) {
          const float* fr = bf(m.actuator_forcerange, m.actuator_forcerange_nb, w, 2 * nu) + 2 * u;
          force = clampf(force, fr[0], fr[1]);
        }
        d.actuator_length[(size_t)w * nu + u] = length;
      }
      d.actuator_force[(size_t)w * nu + u] = force;
      uforce[u] = force;
    }
    // qfrc_actuator = moment^T force.  One lane per actuator adds its joint force into the dof's LDS slot: order-independent, hence
    // deterministic, while no dof has more than two actuators (float addition commutes); beyond that, a gather per dof over all
    // actuators (nv x nu dependent table loads: it was 28 % of this kernel on the humanoid, 48 % on three of them)
    for (int i = lig; i < nv; i += G) factuator[i] = 0.0f;
    gsync();
    if (!(dsbl & DSBL_ACTUATION)) {
      if (m.act_dof_max <= 2) {
        for (int u = lig; u < nu; u += G) atomicAdd(&factuator[m.jnt_dofadr[m.actuator_trnid[2 * u]]], gear[6 * u] * uforce[u]);
      } else {
        for (int i = lig; i < nv; i += G) {
          float s = 0.0f;
          for (int u = 0; u < nu; ++u)
            if (m.jnt_dofadr[m.actuator_trnid[2 * u]] == i) s += gear[6 * u] * uforce[u];
          factuator[i] = s;
        }
      }
    }
    gsync();
    if (nu > 0 && !(dsbl & DSBL_ACTUATION)) {  // forward.py:1121-1150 _qfrc_actuator_gravcomp_limits: actuator-level gravity compensation, then the
      // joint's actuatorfrcrange -- not without actuators / with actuation disabled: the reference returns with qfrc_actuator = 0 (forward.py:1155-1159)
      const float* afr = bf(m.jnt_actfrcrange, m.jnt_actfrcrange_nb, w, 2 * njnt);
      for (int i = lig; i < nv; i += G) {
        const int j = m.dof_jntid[i];
        float q = factuator[i];
        if (m.jnt_actgravcomp[j] && !(dsbl & DSBL_GRAVITY)) q += (first <= VEL_PASSIVE) ? fgrav[i] : d.qfrc_gravcomp[(size_t)w * nv + i];
        if (m.jnt_actfrclimited[j]) q = clampf(q, afr[2 * j], afr[2 * j + 1]);
        factuator[i] = q;
      }
    }
    gsync();
    gcopy<G>(d.qfrc_actuator + (size_t)w * nv, factuator, nv, lig);
  }
  pc.mark(4);
  if (last < VEL_ACCEL) return;

  // ---- fwd_acceleration (forward.py:1255-1324): qfrc_smooth, factor(M), qacc_smooth = M^-1 qfrc_smooth -------
  {
    if (first == VEL_ACCEL) {
      gcopy<G>(fpassive, d.qfrc_passive + (size_t)w * nv, nv, lig);
      gcopy<G>(fbias, d.qfrc_bias + (size_t)w * nv, nv, lig);
      gcopy<G>(factuator, d.qfrc_actuator + (size_t)w * nv, nv, lig);
    }
    gsync();
    for (int i = lig; i < nv; i += G) x[i] = fpassive[i] - fbias[i] + factuator[i] + d.qfrc_applied[(size_t)w * nv + i];
    // xfrc_applied (support.py:259-322): wrench (force, torque) at xipos of each body
    {
      const int nw = (nv + 31) / 32;
      // which bodies carry a wrench: one parallel probe (lane = body) instead of nbody dependent per-world loads
      for (int b0 = 0; b0 < nbody; b0 += G) {
        bool any = false;
        if (b0 + lig < nbody && b0 + lig > 0) {
          const float* f = d.xfrc_applied + ((size_t)w * nbody + b0 + lig) * 6;
          any = f[0] != 0.0f || f[1] != 0.0f || f[2] != 0.0f || f[3] != 0.0f || f[4] != 0.0f || f[5] != 0.0f;
        }
        unsigned mask = (unsigned)(gballot<G>(any));
        while (mask) {
        const int b = b0 + __ffs(mask) - 1;
        mask &= mask - 1;
        const float* f = d.xfrc_applied + ((size_t)w * nbody + b) * 6;
        V3 force = ld3(f), torque = ld3(f + 3);
        V3 off = ld3(d.xipos + ((size_t)w * nbody + b) * 3) - ld3(d.subtree_com + ((size_t)w * nbody + m.body_rootid[b]) * 3);
        for (int i = lig; i < nv; i += G)
          if (m.body_dofmask[b * nw + (i >> 5)] & (1u << (i & 31))) x[i] += jac_dot(m, cdof + 6 * i, off, force, torque);
        }
      }
    }
    gsync();
    // qLD / qLDiagInv / qacc_smooth are produced by k_factor_smooth beside the solver (k_solve takes M^-1 qfrc_smooth
    // from its own register-resident factor), so the L'DL latency chain is off the critical path of the step
    gcopy<G>(d.qfrc_smooth + (size_t)w * nv, x, nv, lig);
    pc.mark(5);
  }
}

// L'DL factor of M and (optionally) qacc_smooth = M^-1 qfrc_smooth: the public outputs of fwd_acceleration
// (forward.py:1255-1324, smooth.py:1183-1232 factor_m, 3187 solve_LD).
struct FacLayout {
  int L, dinv, x, total;
};
__host__ __device__ inline FacLayout fac_layout(int nv, int nC) {
  FacLayout p;
  int o = 0;
  p.L = o; o += nC;
  p.dinv = o; o += nv;
  p.x = o; o += nv;
  p.total = ((o + 3) / 4) * 4 + 1;
  return p;
}
template <int G>
DEV void factor_smooth_body(const MjhModel& m, const MjhData& d, int write_qacc, float* smem, const Blk& b) {
  if ((int)threadIdx.x >= b.nthreads) return;
  const int nv = m.nv, nC = m.nC;
  const FacLayout lay = fac_layout(nv, nC);
  int* shi = reinterpret_cast<int*>(smem);
  const MStruct ms = load_mstruct<G>(m, shi, b.nthreads);
  const int lig = threadIdx.x & (G - 1), gib = threadIdx.x / G;
  const int w = b.w0 + gib;
  if (w >= d.nworld) return;
  float* S = smem + mstruct_ints(nv, nC) + (size_t)gib * lay.total;
  float *L = S + lay.L, *dinv = S + lay.dinv, *x = S + lay.x;
  gcopy<G>(L, d.M + (size_t)w * nC, nC, lig);
  if (write_qacc) gcopy<G>(x, d.qfrc_smooth + (size_t)w * nv, nv, lig);
  gsync();
  factor_ld<G>(ms, L, dinv, nv, lig, &m);
  if (write_qacc) solve_ld<G>(m, ms, L, dinv, x, nv, lig);
  gcopy<G>(d.qLD + (size_t)w * nC, L, nC, lig);
  gcopy<G>(d.qLDiagInv + (size_t)w * nv, dinv, nv, lig);
  if (write_qacc) gcopy<G>(d.qacc_smooth + (size_t)w * nv, x, nv, lig);
}

template <int G>
__global__ void __launch_bounds__(256) k_fwd_pos(MjhModel m, MjhData d, int first, int last) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  fwd_pos_body<G>(m, d, first, last, smem, blk_of_launch<G>());
}
template <int G>
__global__ void __launch_bounds__(256) k_fwd_vel(MjhModel m, MjhData d, int first, int last) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  fwd_vel_body<G>(m, d, first, last, smem, blk_of_launch<G>());
}
template <int G>
__global__ void __launch_bounds__(256) k_factor_smooth(MjhModel m, MjhData d, int write_qacc) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  factor_smooth_body<G>(m, d, write_qacc, smem, blk_of_launch<G>());
}

