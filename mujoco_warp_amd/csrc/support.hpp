// Support functions of the reference's public API (support.py): contact_force (:445, 6D force of selected contacts from the constraint
// forces) and jac (:581, translational / rotational Jacobian of a point on a body).  Small helper kernels, one thread per output.
#pragma once
#include "dev_common.hpp"

// support.py:311-397: pyramidal rows decode to normal / friction components; elliptic rows are the components themselves
DEV void contact_force_of(const MjhModel& m, const MjhData& d, int cid, int to_world_frame, float (&f)[6]);
__global__ void __launch_bounds__(256) k_contact_force(MjhModel m, MjhData d, const int* contact_ids, int n, int to_world_frame, float* out) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  if (tid >= n) return;
  const int cid = contact_ids[tid];
  float f[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  // (-1 is the usual "no contact" marker: a zero wrench, and nothing is read through the id)
  if (cid >= 0 && cid < d.nacon[0]) contact_force_of(m, d, cid, to_world_frame, f);
  for (int k = 0; k < 6; ++k) out[(size_t)tid * 6 + k] = f[k];
}
DEV void contact_force_of(const MjhModel& m, const MjhData& d, int cid, int to_world_frame, float (&f)[6]) {
  for (int k = 0; k < 6; ++k) f[k] = 0.0f;
  const int w = d.contact_worldid[cid], condim = d.contact_dim[cid], npyr = d.nmaxpyramid;
  const int adr0 = d.contact_efc_address[(size_t)cid * npyr];
  if (cid >= 0 && adr0 >= 0) {
    const float* force = d.efc_force + (size_t)w * d.njmax;
    if (m.cone == CONE_PYRAMIDAL) {
      if (condim == 1) f[0] = force[adr0];
      else
        for (int i = 0; i < condim - 1; ++i) {
          const int adr = adr0 + 2 * i;
          const float d1 = adr < d.njmax ? force[adr] : 0.0f, d2 = adr + 1 < d.njmax ? force[adr + 1] : 0.0f;
          f[0] += d1 + d2;
          f[i + 1] = (d1 - d2) * d.contact_friction[(size_t)cid * 5 + i];
        }
    } else {
      for (int i = 0; i < condim; ++i) {
        const int adr = d.contact_efc_address[(size_t)cid * npyr + i];
        if (adr >= 0 && adr < d.njmax) f[i] = force[adr];
      }
    }
  }
  if (to_world_frame) {  // row vector times the contact frame (rows = normal, tangent 1, tangent 2)
    const float* R = d.contact_frame + (size_t)cid * 9;
    const float t[3] = {f[0], f[1], f[2]}, b[3] = {f[3], f[4], f[5]};
    for (int k = 0; k < 3; ++k) {
      f[k] = t[0] * R[k] + t[1] * R[3 + k] + t[2] * R[6 + k];
      f[3 + k] = b[0] * R[k] + b[1] * R[3 + k] + b[2] * R[6 + k];
    }
  }
}

// support.py:505-578: column dof of the Jacobians of `point` (world coordinates) moving with body `body[w]`; either output may be null
__global__ void __launch_bounds__(256) k_jac(MjhModel m, MjhData d, float* jacp, float* jacr, const float* point, const int* body) {
  const int idx = blockIdx.x * 256 + threadIdx.x, nv = m.nv;
  if (idx >= d.nworld * nv) return;
  const int w = idx / nv, dof = idx - w * nv, b = body[w], nw32 = (nv + 31) / 32;
  V3 jp = V3{0, 0, 0}, jr = V3{0, 0, 0};
  if (m.body_dofmask[b * nw32 + (dof >> 5)] & (1u << (dof & 31))) {
    const V3 off = ld3(point + 3 * w) - ld3(d.subtree_com + ((size_t)w * m.nbody + m.body_rootid[b]) * 3);
    const float* cd = d.cdof + ((size_t)w * nv + dof) * 6;
    jr = ld3(cd);
    jp = ld3(cd + 3) + cross(jr, off);
  }
  if (jacp) {
    jacp[((size_t)w * 3 + 0) * nv + dof] = jp.x;
    jacp[((size_t)w * 3 + 1) * nv + dof] = jp.y;
    jacp[((size_t)w * 3 + 2) * nv + dof] = jp.z;
  }
  if (jacr) {
    jacr[((size_t)w * 3 + 0) * nv + dof] = jr.x;
    jacr[((size_t)w * 3 + 1) * nv + dof] = jr.y;
    jacr[((size_t)w * 3 + 2) * nv + dof] = jr.z;
  }
}

// sensor.energy_pos / energy_vel (sensor.py:2773-3018; EnableBit.ENERGY): Data.energy = (potential, kinetic).  Potential = -sum m g . xipos
// + joint springs 0.5 k r^2 (r = displacement from qpos_spring; quaternion joints: the rotation vector); kinetic = 0.5 qvel . M qvel from
// the sparse M (row i = the dof's ancestor chain, diagonal last).  One 32-lane group per world: bodies / joints / dofs over the lanes, one
// cross-lane sum per term (round 3; was one thread per world).
template <int G>
__global__ void __launch_bounds__(256) k_energy(MjhModel m, MjhData d) {
  const int lig = threadIdx.x & (G - 1);
  const int w = blockIdx.x * (blockDim.x / G) + threadIdx.x / G;
  if (w >= d.nworld) return;
  const V3 g = ld3(bf(m.opt_gravity, m.opt_gravity_nb, w, 3));
  const float* mass = bf(m.body_mass, m.body_mass_nb, w, m.nbody);
  float pot = 0.0f;
  if (!(m.disableflags & DSBL_GRAVITY))
    for (int b = 1 + lig; b < m.nbody; b += G) pot -= mass[b] * dot(g, ld3(d.xipos + ((size_t)w * m.nbody + b) * 3));
  if (!(m.disableflags & DSBL_SPRING)) {
    const float* stiff = bf(m.jnt_stiffness, m.jnt_stiffness_nb, w, m.njnt);
    const float* qs = bf(m.qpos_spring, m.qpos_spring_nb, w, m.nq);
    const float* qpos = d.qpos + (size_t)w * m.nq;
    for (int j = lig; j < m.njnt; j += G) {
      const float kk = stiff[j];
      if (kk == 0.0f) continue;
      const int qa = m.jnt_qposadr[j], t = m.jnt_type[j];
      if (t == JNT_FREE) {
        const V3 d0 = ld3(qpos + qa) - ld3(qs + qa);
        const V3 d1 = quat_sub(quat_normalize(ld4(qpos + qa + 3)), ld4(qs + qa + 3));
        pot += 0.5f * kk * (dot(d0, d0) + dot(d1, d1));
      } else if (t == JNT_BALL) {
        const V3 d1 = quat_sub(quat_normalize(ld4(qpos + qa)), ld4(qs + qa));
        pot += 0.5f * kk * dot(d1, d1);
      } else {
        const float r = qpos[qa] - qs[qa];
        pot += 0.5f * kk * r * r;
      }
    }
  }
  const float* M = d.M + (size_t)w * m.nC;
  const float* v = d.qvel + (size_t)w * m.nv;
  float kin = 0.0f;
  for (int i = lig; i < m.nv; i += G) {
    const int adr = m.M_rowadr[i], nnz = m.M_rownnz[i];
    float acc = 0.5f * M[adr + nnz - 1] * v[i];  // diagonal is the last entry of the row
    for (int q = 0; q < nnz - 1; ++q) acc += M[adr + q] * v[m.M_colind[adr + q]];
    kin += v[i] * acc;
  }
  pot = gsum<G>(pot);
  kin = gsum<G>(kin);
  if (lig == 0) {
    d.energy[2 * w] = pot;
    d.energy[2 * w + 1] = kin;
  }
}

// smooth.subtree_vel (smooth.py:3502-3662): linear velocity of every subtree's centre of mass and angular momentum of every subtree about
// it.  The reference accumulates momenta towards the root level by level; body ids are depth-first, so a subtree is the id range
// [b, b + body_subtreenum[b]) and both results are range sums of per-body terms that are independent of each other:
//   linvel_b = sum_{c in subtree(b)} m_c v_c / subtreemass_b,                       v_c = velocity of the body's centre of mass
//   angmom_b = sum_{c in subtree(b)} (A_c + T_c) - T_b,   A_c = R I R' w + (xipos_c - subtree_com_c) x m_c (v_c - linvel_c),
//                                                           T_c = (subtree_com_c - subtree_com_parent) x subtreemass_c (linvel_c - linvel_parent)
// (A_c: the body's own angular momentum about its subtree's centre; T_c: what moving subtree c's momentum to the parent's centre adds, once
// per body whatever the number of descendants -- the unrolled form of the reference's child-to-parent recursion.)
// One 32-lane group per world, one lane per body, terms in LDS (3 x 3 nbody floats per world).  Round 3: the first version ran one THREAD
// per world over strided rows -- 63 us per step for the G1 at 4096 worlds, 13 % of its step.
DEV V3 body_com_linvel(const MjhModel& m, const MjhData& d, int w, int b) {
  const float* cv = d.cvel + ((size_t)w * m.nbody + b) * 6;
  const V3 off = ld3(d.xipos + ((size_t)w * m.nbody + b) * 3) - ld3(d.subtree_com + ((size_t)w * m.nbody + m.body_rootid[b]) * 3);
  return ld3(cv + 3) - cross(off, ld3(cv));
}
template <int G>
__global__ void __launch_bounds__(256) k_subtree_vel(MjhModel m, MjhData d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lig = threadIdx.x & (G - 1), gib = threadIdx.x / G, nb = m.nbody;
  const int w = blockIdx.x * (blockDim.x / G) + gib;
  if (w >= d.nworld) return;
  float* lin = smem + (size_t)gib * 9 * nb;  // m v per body, then T
  float* lv = lin + 3 * nb;                  // subtree linear velocity
  float* acc = lv + 3 * nb;                  // A + T per body
  const float* mass = bf(m.body_mass, m.body_mass_nb, w, nb);
  const float* inertia = bf(m.body_inertia, m.body_inertia_nb, w, 3 * nb);
  const float* stm = bf(m.body_subtreemass, m.body_subtreemass_nb, w, nb);
  float* linvel = d.subtree_linvel + (size_t)w * nb * 3;
  float* angmom = d.subtree_angmom + (size_t)w * nb * 3;
  const float* scom = d.subtree_com + (size_t)w * nb * 3;
  for (int b = lig; b < nb; b += G) st3(lin + 3 * b, mass[b] * body_com_linvel(m, d, w, b));
  gsync();
  for (int b = lig; b < nb; b += G) {
    V3 s = ld3(lin + 3 * b);
    const int e = b + m.body_subtreenum[b];
    for (int c = b + 1; c < e; ++c) s = s + ld3(lin + 3 * c);
    s = s * (1.0f / fmaxf(MJ_MINVAL, stm[b]));
    st3(lv + 3 * b, s);
    st3(linvel + 3 * b, s);
  }
  gsync();
  for (int b = lig; b < nb; b += G) {
    const float* R = d.ximat + ((size_t)w * nb + b) * 9;
    V3 dv = matT_mul(R, ld3(d.cvel + ((size_t)w * nb + b) * 6));
    dv = V3{dv.x * inertia[3 * b], dv.y * inertia[3 * b + 1], dv.z * inertia[3 * b + 2]};
    V3 A = mat_mul(R, dv), T = V3{0, 0, 0};
    if (b > 0) {
      const int p = m.body_parentid[b];
      const V3 lb = ld3(lv + 3 * b);
      A = A + cross(ld3(d.xipos + ((size_t)w * nb + b) * 3) - ld3(scom + 3 * b), (body_com_linvel(m, d, w, b) - lb) * mass[b]);
      T = cross(ld3(scom + 3 * b) - ld3(scom + 3 * p), (lb - ld3(lv + 3 * p)) * stm[b]);
    }
    st3(acc + 3 * b, A + T);
    st3(lin + 3 * b, T);  // (every lane is past its reads of lin: the gsync above)
  }
  gsync();
  for (int b = lig; b < nb; b += G) {
    V3 s = ld3(acc + 3 * b) - ld3(lin + 3 * b);
    const int e = b + m.body_subtreenum[b];
    for (int c = b + 1; c < e; ++c) s = s + ld3(acc + 3 * c);
    st3(angmom + 3 * b, s);
  }
}
// worlds per 256-thread workgroup of k_subtree_vel<32> so that 9 nbody floats per world fit in 64 KB of LDS (0: the model is too big)
static inline int subtree_vel_wpb(int nbody) {
  int wpb = 8;
  while (wpb > 0 && (size_t)wpb * 9 * nbody * sizeof(float) > 65536) wpb >>= 1;
  return wpb;
}

// smooth.rne_postconstraint (smooth.py:1519-1826): cacc, cfrc_ext, cfrc_int with the constraint forces in.  cfrc_ext = applied Cartesian
// forces + contact forces moved to the tree's centre of mass (spatial vectors: torque first); cacc from qacc down the tree; cfrc_int =
// I cacc + v x* (I v) - cfrc_ext, summed towards the root (equality rows: joint equalities exert no cfrc_ext).
// One 32-lane group per world, one lane per body (round 3; was one thread per world): every lane scans the world's contacts in order and keeps
// those of its body (deterministic sums); cacc of a body = the world's -g + the per-body terms of its ancestor chain; cfrc_int = the sum of the
// bodies' own terms over the depth-first id range of the subtree.  LDS: 18 nbody floats per world.
template <int G>
__global__ void __launch_bounds__(256) k_rne_postconstraint(MjhModel m, MjhData d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lig = threadIdx.x & (G - 1), gib = threadIdx.x / G, nb = m.nbody;
  const int w = blockIdx.x * (blockDim.x / G) + gib;
  if (w >= d.nworld) return;
  float* cext = smem + (size_t)gib * 18 * nb;  // cfrc_ext per body
  float* loc = cext + 6 * nb;                 // per-body acceleration term, then cacc
  float* own = loc + 6 * nb;                  // the body's own I cacc + v x* (I v) - cfrc_ext
  const float* scom = d.subtree_com + (size_t)w * nb * 3;
  const int ncon = min(d.ws_ncon[w], d.concap);
  const float* efc_force = d.efc_force + (size_t)w * d.njmax;
  const float* qvel = d.qvel + (size_t)w * m.nv;
  const float* qacc = d.qacc + (size_t)w * m.nv;
  for (int b = lig; b < nb; b += G) {
    float c[6] = {0, 0, 0, 0, 0, 0};
    auto add_force = [&](V3 force, V3 torque, V3 offset, float sgn) {  // support.transform_force: (torque - offset x force, force)
      const V3 t = torque - cross(offset, force);
      c[0] += sgn * t.x; c[1] += sgn * t.y; c[2] += sgn * t.z;
      c[3] += sgn * force.x; c[4] += sgn * force.y; c[5] += sgn * force.z;
    };
    if (b > 0) {
      const float* xf = d.xfrc_applied + ((size_t)w * nb + b) * 6;
      add_force(ld3(xf), ld3(xf + 3), ld3(scom + 3 * m.body_rootid[b]) - ld3(d.xipos + ((size_t)w * nb + b) * 3), 1.0f);
      // contacts: from the world's records (collide.hpp; the public contact arrays are published off the critical path and may not be
      // there yet when the acceleration-stage sensors run); same decoding as k_contact_force
      for (int cc = 0; cc < ncon; ++cc) {
        const float* rec = d.ws_contact + ((size_t)w * d.concap + cc) * CON_STRIDE;
        const int* reci = reinterpret_cast<const int*>(rec);
        const int b1 = m.geom_bodyid[reci[25]], b2 = m.geom_bodyid[reci[26]], adr0 = reci[28], condim = reci[24];
        if ((b1 != b && b2 != b) || adr0 < 0) continue;
        float f[6] = {0, 0, 0, 0, 0, 0};
        if (m.cone == CONE_PYRAMIDAL) {
          if (condim == 1) f[0] = adr0 < d.njmax ? efc_force[adr0] : 0.0f;
          else
            for (int i = 0; i < condim - 1; ++i) {
              const int a = adr0 + 2 * i;
              const float d1 = a < d.njmax ? efc_force[a] : 0.0f, d2 = a + 1 < d.njmax ? efc_force[a + 1] : 0.0f;
              f[0] += d1 + d2;
              f[i + 1] = (d1 - d2) * rec[CON_FRICTION_WORD(i)];
            }
        } else {
          for (int i = 0; i < condim; ++i)
            if (adr0 + i < d.njmax) f[i] = efc_force[adr0 + i];
        }
        const float* R = rec + 4;  // contact frame rows: normal, tangent 1, tangent 2
        const V3 force = V3{f[0] * R[0] + f[1] * R[3] + f[2] * R[6], f[0] * R[1] + f[1] * R[4] + f[2] * R[7], f[0] * R[2] + f[1] * R[5] + f[2] * R[8]};
        const V3 torque = V3{f[3] * R[0] + f[4] * R[3] + f[5] * R[6], f[3] * R[1] + f[4] * R[4] + f[5] * R[7], f[3] * R[2] + f[4] * R[5] + f[5] * R[8]};
        const V3 off = ld3(scom + 3 * m.body_rootid[b]) - ld3(rec + 1);
        if (b1 == b) add_force(force, torque, off, -1.0f);
        if (b2 == b) add_force(force, torque, off, 1.0f);
      }
    }
    for (int k = 0; k < 6; ++k) cext[6 * b + k] = c[k];
    float a[6] = {0, 0, 0, 0, 0, 0};  // this body's own term of cacc
    for (int j = 0; b > 0 && j < m.body_dofnum[b]; ++j) {
      const int dof = m.body_dofadr[b] + j;
      const float* cd = d.cdof + ((size_t)w * m.nv + dof) * 6;
      const float* cdd = d.cdof_dot + ((size_t)w * m.nv + dof) * 6;
      for (int k = 0; k < 6; ++k) a[k] += cdd[k] * qvel[dof] + cd[k] * qacc[dof];
    }
    for (int k = 0; k < 6; ++k) loc[6 * b + k] = a[k];
  }
  gsync();
  const V3 g = ld3(bf(m.opt_gravity, m.opt_gravity_nb, w, 3));
  const bool grav = !(m.disableflags & DSBL_GRAVITY);
  for (int b = lig; b < nb; b += G) {
    float a[6] = {0, 0, 0, grav ? -g.x : 0.0f, grav ? -g.y : 0.0f, grav ? -g.z : 0.0f};
    for (int p = b; p > 0; p = m.body_parentid[p])
      for (int k = 0; k < 6; ++k) a[k] += loc[6 * p + k];
    float f1[6] = {0, 0, 0, 0, 0, 0}, iv[6], f2[6] = {0, 0, 0, 0, 0, 0};
    if (b > 0) {
      const float* ci = d.cinert + ((size_t)w * nb + b) * 10;
      const float* cv = d.cvel + ((size_t)w * nb + b) * 6;
      inert_vec(ci, a, f1);
      inert_vec(ci, cv, iv);
      motion_cross_force(cv, iv, f2);
    }
    for (int k = 0; k < 6; ++k) {
      own[6 * b + k] = b ? f1[k] + f2[k] - cext[6 * b + k] : 0.0f;
      d.cacc[((size_t)w * nb + b) * 6 + k] = a[k];
      d.cfrc_ext[((size_t)w * nb + b) * 6 + k] = cext[6 * b + k];
    }
  }
  gsync();
  for (int b = lig; b < nb; b += G) {
    float s[6] = {0, 0, 0, 0, 0, 0};
    const int e = b + m.body_subtreenum[b];
    for (int c = b; c < e; ++c)
      for (int k = 0; k < 6; ++k) s[k] += own[6 * c + k];
    for (int k = 0; k < 6; ++k) d.cfrc_int[((size_t)w * nb + b) * 6 + k] = s[k];
  }
}
// worlds per 256-thread workgroup of a 32-lane-per-world kernel with `words` floats of LDS per world (0: does not fit in 64 KB)
static inline int lds_wpb32(size_t words) {
  int wpb = 8;
  while (wpb > 0 && (size_t)wpb * words * sizeof(float) > 65536) wpb >>= 1;
  return wpb;
}
