// constraint.hpp -- constraint assembly fused per world (one lane group per world).
//
// Reference: constraint.py:84-153 (_efc_row), 1766-1865 (_friction_dof), 1991-2105 (_limit_slide_hinge),
// 2107-2240 (_limit_ball), 2641-2757 (_efc_contact_init), 3751-3879 (_efc_contact_jac_dense),
// 4197-4343 (_efc_contact_update); support.py:488-533 (jac helpers).
//
// MI355X mapping: the reference allocates rows with per-world atomics from five separate launches (row order
// nondeterministic).  Here rows are allocated with ballot/prefix compaction in MuJoCo's canonical order
// (dof friction, joint limits, contacts in contact order), J rows are written with lanes mapped to dofs
// (coalesced 128 B row segments), and J*qvel comes from DPP/permute group reductions -- no atomics at all.
#pragma once
#include "dev_common.hpp"

struct EfcRowOut {
  float D, aref, pos;
};

// _efc_row constraint.py:84-153
DEV EfcRowOut efc_row(int dsbl, float timestep, float pos_aref, float pos_imp, float invweight, const float* solref,
                      const float* solimp, float margin, float vel) {
  float timeconst = solref[0];
  const float dampratio = solref[1];
  float dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
  if (!(dsbl & DSBL_REFSAFE)) timeconst = fmaxf(timeconst, 2.0f * timestep);
  dmin = clampf(dmin, MJ_MINIMP, MJ_MAXIMP);
  dmax = clampf(dmax, MJ_MINIMP, MJ_MAXIMP);
  width = fmaxf(MJ_MINVAL, width);
  mid = clampf(mid, MJ_MINIMP, MJ_MAXIMP);
  power = fmaxf(1.0f, power);
  const float dmax_sq = dmax * dmax;
  float k = 1.0f / (dmax_sq * timeconst * timeconst * dampratio * dampratio);
  float b = 2.0f / (dmax * timeconst);
  if (solref[0] <= 0.0f) k = -solref[0] / dmax_sq;
  if (solref[1] <= 0.0f) b = -solref[1] / dmax;
  const float imp_x = fabsf(pos_imp) / width;
  float imp_a, imp_b;
  if (power == 2.0f) {  // MuJoCo's default: x^2 / mid and 1 - (1-x)^2 / (1-mid) without four powf expansions (~300 VALU ops)
    imp_a = imp_x * imp_x / mid;
    imp_b = 1.0f - (1.0f - imp_x) * (1.0f - imp_x) / (1.0f - mid);
  } else {
    imp_a = (1.0f / powf(mid, power - 1.0f)) * powf(imp_x, power);
    imp_b = 1.0f - (1.0f / powf(1.0f - mid, power - 1.0f)) * powf(1.0f - imp_x, power);
  }
  const float imp_y = imp_x < mid ? imp_a : imp_b;
  float imp = dmin + imp_y * (dmax - dmin);
  imp = clampf(imp, dmin, dmax);
  if (imp_x > 1.0f) imp = dmax;
  EfcRowOut o;
  o.D = 1.0f / fmaxf(invweight * (1.0f - imp) / imp, MJ_MINVAL);
  o.aref = -k * imp * pos_aref - b * vel;
  o.pos = pos_aref + margin;
  return o;
}

struct ConLayout {
  int cdof, qvel, rowdof, rowval, row2con, clist, cwin, scom, gbody, broot, bmask, total;
};
__host__ __device__ inline ConLayout con_layout(int nv, int njmax, int ncap, int nbody, int ngeom) {
  ConLayout p;
  int o = 0;
  p.cdof = o; o += 6 * nv;
  p.qvel = o; o += nv;
  p.rowdof = o; o += njmax;
  p.rowval = o; o += njmax;
  p.row2con = o; o += njmax;
  p.clist = o; o += 3 * ncap;
  p.cwin = o; o += CON_WINDOW * CON_STRIDE;  // staged contact records (see collide.hpp)
  // model/kinematic tables of the contact Jacobian: the contact loop must not issue global loads, because a wave's
  // loads wait behind its earlier J stores (one vmcnt counter for both on gfx9)
  p.scom = o; o += 3 * nbody;
  p.gbody = o; o += ngeom;
  p.broot = o; o += nbody;
  p.bmask = o; o += nbody * ((nv + 31) / 32);
  p.total = ((o + 3) / 4) * 4 + 1;
  return p;
}

// sums of N values over the G lanes of a group, the result in every lane (VALU only: DPP row shifts, then the row sums are combined)
template <int G, int N>
DEV void csum_n(float (&v)[N]) {
  static_assert(G == 16 || G == 32 || G == 64, "lane groups of 16, 32 or 64");
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = dpp_add_f<0x111, 0xf, 0xf>(v[i]);  // row_shr:1
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = dpp_add_f<0x112, 0xf, 0xf>(v[i]);  // row_shr:2
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = dpp_add_f<0x114, 0xf, 0xf>(v[i]);  // row_shr:4
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = dpp_add_f<0x118, 0xf, 0xf>(v[i]);  // row_shr:8: lane 15 of every 16-lane row holds the row's sum
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (G == 16) {
      v[i] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[i]), 0x15F, 0xf, 0xf, true));  // row_newbcast:15
    } else if (G == 32) {
      const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i]), false, false);
      const float lo = __int_as_float(__builtin_amdgcn_update_dpp(0, (int)sw[0], 0x15F, 0xf, 0xf, true));
      const float hi = __int_as_float(__builtin_amdgcn_update_dpp(0, (int)sw[1], 0x15F, 0xf, 0xf, true));
      v[i] = lo + hi;
    } else {
      float x = dpp_add_f<0x142, 0xa, 0xf>(v[i]);  // row_bcast:15 into rows 1 and 3
      x = dpp_add_f<0x143, 0xc, 0xf>(x);           // row_bcast:31 into rows 2 and 3: lane 63 holds the wavefront's sum
      v[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
    }
  }
}

template <int G>
DEV void make_constraint_body(const MjhModel& m, const MjhData& d, float* smem, const Blk& b, int stride_words = 0) {
  const int lig = threadIdx.x & (G - 1), gib = threadIdx.x / G;
  const int w = b.w0 + gib;
  if ((int)threadIdx.x >= b.nthreads || w >= d.nworld) return;
  const int nv = m.nv, njnt = m.njnt, nbody = m.nbody, njmax = d.njmax, nvp = d.nv_pad, ncap = d.concap;
  const ConLayout lay = con_layout(nv, njmax, ncap, nbody, m.ngeom);
  float* S = smem + (size_t)gib * (stride_words ? stride_words : lay.total);
  float *cdof = S + lay.cdof, *qvel = S + lay.qvel, *rowval = S + lay.rowval;
  int *rowdof = reinterpret_cast<int*>(S + lay.rowdof), *row2con = reinterpret_cast<int*>(S + lay.row2con),
      *clist = reinterpret_cast<int*>(S + lay.clist);
  float* cwin = S + lay.cwin;
  float* scom = S + lay.scom;
  int *gbody = reinterpret_cast<int*>(S + lay.gbody), *broot = reinterpret_cast<int*>(S + lay.broot);
  unsigned* bmask = reinterpret_cast<unsigned*>(S + lay.bmask);
  const int dsbl = m.disableflags;
  float* J = d.efc_J + (size_t)w * d.njmax_pad * nvp;
  const size_t eo = (size_t)w * njmax;

  if (dsbl & DSBL_CONSTRAINT) {
    if (lig == 0) d.ne[w] = d.nf[w] = d.nl[w] = d.nefc[w] = 0;
    return;
  }
  PhaseClock pc(3, lig);
  gcopy<G>(cdof, d.cdof + (size_t)w * 6 * nv, 6 * nv, lig);
  gcopy<G>(qvel, d.qvel + (size_t)w * nv, nv, lig);
  gcopy<G>(scom, d.subtree_com + (size_t)w * 3 * nbody, 3 * nbody, lig);
  gcopyi<G>(gbody, m.geom_bodyid, m.ngeom, lig);
  gcopyi<G>(broot, m.body_rootid, nbody, lig);
  for (int i = lig; i < nbody * ((nv + 31) / 32); i += G) bmask[i] = m.body_dofmask[i];
  const float timestep = bf(m.opt_timestep, m.opt_timestep_nb, w, 1)[0];
  const float* qpos = d.qpos + (size_t)w * m.nq;
  const float* invw = bf(m.dof_invweight0, m.dof_invweight0_nb, w, nv);
  gsync();
  pc.mark(0);

  int nefc = 0, ne = 0, nf = 0, nl = 0;
  // ---- equality constraints: joint couplings (constraint.py:500-640); rows in equality-id order -------------
  if (!(dsbl & DSBL_EQUALITY)) {
    const int neq = m.neq;
    const float* qpos0 = bf(m.qpos0, m.qpos0_nb, w, m.nq);
    for (int base = 0; base < neq; base += G) {
      const int e = base + lig;
      const bool act = e < neq && d.eq_active[(size_t)w * neq + e] != 0;
      int tot;
      const int r = nefc + grank<G>(act, lig, tot);
      if (act && r < njmax) {
        const int j1 = m.eq_obj1id[e], j2 = m.eq_obj2id[e];
        const float* data = bf(m.eq_data, m.eq_data_nb, w, 11 * neq) + 11 * e;
        const int dof1 = m.jnt_dofadr[j1], qa1 = m.jnt_qposadr[j1];
        for (int c = 0; c < nvp; ++c) J[(size_t)r * nvp + c] = 0.0f;
        J[(size_t)r * nvp + dof1] = 1.0f;
        float pos, vel, invweight;
        if (j2 >= 0) {
          const int dof2 = m.jnt_dofadr[j2], qa2 = m.jnt_qposadr[j2];
          const float dif = qpos[qa2] - qpos0[qa2];
          const float rhs = data[0] + dif * (data[1] + dif * (data[2] + dif * (data[3] + dif * data[4])));
          const float deriv = data[1] + dif * (2.0f * data[2] + dif * (3.0f * data[3] + dif * 4.0f * data[4]));
          pos = qpos[qa1] - qpos0[qa1] - rhs;
          vel = qvel[dof1] - qvel[dof2] * deriv;
          invweight = invw[dof1] + invw[dof2];
          J[(size_t)r * nvp + dof2] = -deriv;
        } else {
          pos = qpos[qa1] - qpos0[qa1] - data[0];
          vel = qvel[dof1];
          invweight = invw[dof1];
        }
        EfcRowOut o = efc_row(dsbl, timestep, pos, pos, invweight, bf(m.eq_solref, m.eq_solref_nb, w, 2 * neq) + 2 * e,
                              bf(m.eq_solimp, m.eq_solimp_nb, w, 5 * neq) + 5 * e, 0.0f, vel);
        d.efc_D[eo + r] = o.D;
        d.efc_aref[eo + r] = o.aref;
        d.efc_pos[eo + r] = o.pos;
        d.efc_margin[eo + r] = 0.0f;
        d.efc_vel[eo + r] = vel;
        d.efc_frictionloss[eo + r] = 0.0f;
        d.efc_type[eo + r] = CT_EQUALITY;
        d.efc_id[eo + r] = e;
      }
      nefc += tot;
      ne += tot;
    }
  }
  const int nrow_equality = nefc;
  // ---- dof friction loss (constraint.py:1766-1865) ---------------------------------------------------------
  if (!(dsbl & DSBL_FRICTIONLOSS)) {
    const float* fl = bf(m.dof_frictionloss, m.dof_frictionloss_nb, w, nv);
    for (int base = 0; base < nv; base += G) {
      const int i = base + lig;
      const bool act = i < nv && fl[i] > 0.0f;
      int tot;
      const int r = nefc + grank<G>(act, lig, tot);
      if (act && r < njmax) {
        rowdof[r] = i;
        rowval[r] = 1.0f;
        const float vel = qvel[i];
        EfcRowOut o = efc_row(dsbl, timestep, 0.0f, 0.0f, invw[i], bf(m.dof_solref, m.dof_solref_nb, w, 2 * nv) + 2 * i,
                              bf(m.dof_solimp, m.dof_solimp_nb, w, 5 * nv) + 5 * i, 0.0f, vel);
        d.efc_D[eo + r] = o.D;
        d.efc_aref[eo + r] = o.aref;
        d.efc_pos[eo + r] = o.pos;
        d.efc_margin[eo + r] = 0.0f;
        d.efc_vel[eo + r] = vel;
        d.efc_frictionloss[eo + r] = fl[i];
        d.efc_type[eo + r] = CT_FRICTION_DOF;
        d.efc_id[eo + r] = i;
      }
      nefc += tot;
      nf += tot;
    }
  }
  // ---- joint limits (constraint.py:1991-2240) ---------------------------------------------------------------
  int nball = 0;
  if (!(dsbl & DSBL_LIMIT)) {
    const float* jrange = bf(m.jnt_range, m.jnt_range_nb, w, 2 * njnt);
    const float* jmargin = bf(m.jnt_margin, m.jnt_margin_nb, w, njnt);
    for (int base = 0; base < njnt; base += G) {
      const int j = base + lig;
      bool act = false;
      float pos = 0.0f, jval = 0.0f, margin = 0.0f;
      V3 axis = V3{0, 0, 0};
      int t = -1, dof = 0;
      if (j < njnt && m.jnt_limited[j]) {
        t = m.jnt_type[j];
        dof = m.jnt_dofadr[j];
        margin = jmargin[j];
        const int qa = m.jnt_qposadr[j];
        if (t == JNT_SLIDE || t == JNT_HINGE) {
          const float q = qpos[qa];
          const float dmin = q - jrange[2 * j], dmax = jrange[2 * j + 1] - q;
          pos = fminf(dmin, dmax) - margin;
          jval = dmin < dmax ? 1.0f : -1.0f;
          act = pos < 0.0f;
        } else if (t == JNT_BALL) {
          float angle;
          axis = normalize_with_norm(quat_to_vel(quat_normalize(ld4(qpos + qa))), angle);
          pos = fmaxf(jrange[2 * j], jrange[2 * j + 1]) - angle - margin;
          act = pos < 0.0f;
        }
      }
      int tot;
      const int r = nefc + grank<G>(act, lig, tot);
      if (act && r < njmax) {
        float vel;
        if (t == JNT_BALL) {
          rowdof[r] = -(dof + 1);  // ball row: three entries written below
          rowval[r] = 0.0f;
          vel = -(axis.x * qvel[dof] + axis.y * qvel[dof + 1] + axis.z * qvel[dof + 2]);
        } else {
          rowdof[r] = dof;
          rowval[r] = jval;
          vel = jval * qvel[dof];
        }
        EfcRowOut o = efc_row(dsbl, timestep, pos, pos, invw[dof], bf(m.jnt_solref, m.jnt_solref_nb, w, 2 * njnt) + 2 * j,
                              bf(m.jnt_solimp, m.jnt_solimp_nb, w, 5 * njnt) + 5 * j, margin, vel);
        d.efc_D[eo + r] = o.D;
        d.efc_aref[eo + r] = o.aref;
        d.efc_pos[eo + r] = o.pos;
        d.efc_margin[eo + r] = margin;
        d.efc_vel[eo + r] = vel;
        d.efc_frictionloss[eo + r] = 0.0f;
        d.efc_type[eo + r] = CT_LIMIT_JOINT;
        d.efc_id[eo + r] = j;
      }
      nball += gsumi<G>((act && t == JNT_BALL) ? 1 : 0);
      nefc += tot;
      nl += tot;
    }
  }
  gsync();
  pc.mark(1);
  // cooperative, coalesced write of the (one-hot) friction/limit rows of J
  {
    const int nrow = min(nefc, njmax);
    for (int idx = nrow_equality * nvp + lig; idx < nrow * nvp; idx += G) {  // equality rows were written above
      const int r = idx / nvp, c = idx - r * nvp;
      J[idx] = (c == rowdof[r]) ? rowval[r] : 0.0f;
    }
    if (nball > 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      for (int r = nrow_equality + lig; r < nrow; r += G)
        if (rowdof[r] < 0) {
          const int dof = -rowdof[r] - 1;
          const int j = m.dof_jntid[dof];
          float angle;
          V3 axis = normalize_with_norm(quat_to_vel(quat_normalize(ld4(qpos + m.jnt_qposadr[j]))), angle);
          J[r * nvp + dof] = -axis.x;
          J[r * nvp + dof + 1] = -axis.y;
          J[r * nvp + dof + 2] = -axis.z;
        }
    }
  }
  const int nrow_noncontact = nefc;
  pc.mark(2);

  // ---- contacts (constraint.py:2641-2757 init, 3751-3879 jac, 4197-4343 update) ---------------------------
  // The world's contact records come from d.ws_contact (k_collision); the public contact_* arrays are not read.
  int nactive = 0;
  const int ncon = (dsbl & DSBL_CONTACT) ? 0 : min(d.ws_ncon[w], ncap);
  float* crec = d.ws_contact + (size_t)w * ncap * CON_STRIDE;
  int* creci = reinterpret_cast<int*>(crec);
  for (int base = 0; base < ncon; base += G) {
    const int c = base + lig;
    bool act = false;
    int ndim = 0;
    if (c < ncon) {
      const float pos = crec[c * CON_STRIDE] - crec[c * CON_STRIDE + 13];
      act = pos < 0.0f;
      const int condim = creci[c * CON_STRIDE + 24];
      // pyramidal: two rows per friction dimension; elliptic: one row per contact dimension (constraint.py:2698-2704)
      ndim = act ? min(condim == 1 ? 1 : (m.cone == CONE_ELLIPTIC ? condim : 2 * (condim - 1)), d.nmaxpyramid) : 0;
    }
    int incl = ndim;
    for (int off = 1; off < G; off <<= 1) {
      int v = __shfl_up(incl, off, G);
      if (lig >= off) incl += v;
    }
    const int rbase = nefc + incl - ndim;
    int tot;
    const int arank = nactive + grank<G>(act, lig, tot);
    if (c < ncon) {
      creci[c * CON_STRIDE + 28] = act ? rbase : -1;  // contact.efc_address / efc.id are published from these
      creci[c * CON_STRIDE + 29] = ndim;
    }
    if (act) {
      for (int k = 0; k < ndim; ++k)
        if (rbase + k < njmax) row2con[rbase + k] = c * 16 + k;
      clist[3 * arank] = c;
      clist[3 * arank + 1] = rbase;
      clist[3 * arank + 2] = ndim;
    }
    nefc += __shfl(incl, G - 1, G);
    nactive += tot;
  }
  gsync();
  pc.mark(3);
  // Contacts, a window of CON_WINDOW records at a time (round 5; the round-4 loop walked the active contacts one after the other through
  // five dependent LDS hops each -- list -> record -> geom body -> root body -> subtree com -- and a later pass re-read the finished J rows
  // from global memory for J qvel: 2.8 k cycles per contact, 36 % + 29 % of this kernel):
  //   1. the window's raw records are staged in LDS with consecutive addresses (`cwin`: one memory round trip);
  //   2. dof loop over the window's active contacts, lane = dof: the record is read from `cwin` (broadcast reads), the six Jacobian components
  //      give the rows (coalesced 128-byte stores) and, multiplied with qvel and group-reduced, the contact's basis velocities v_n, v_t1, v_t2
  //      (and the three spin / roll ones for condim > 3), which OVERWRITE words 17 .. 22 of the staged record -- its solref / solimp words:
  //      from here on those are only valid in the global record `crec`;
  //   3. the window's rows, lane = row: D, aref, pos, margin, type, id -- and vel as a combination of the contact's basis velocities
  //      (nothing of J is read back); solref / solimp come from `crec`.
  // (An earlier design note here described a pre-pass that rewrote compact records in active order; it was never what the code does.)
  const int nw = (nv + 31) / 32;
  const float impr2 = bf(m.opt_impratio_invsqrt, m.opt_impratio_invsqrt_nb, w, 1)[0];
  const float* biw = bf(m.body_invweight0, m.body_invweight0_nb, w, 2 * nbody);
  int a = 0;
  for (int wbase = 0; wbase < ncon && a < nactive; wbase += CON_WINDOW) {
    const int wend = min(wbase + CON_WINDOW, ncon);
    if (clist[3 * a] >= wend) continue;  // no active contact in this window
    for (int idx = lig; idx < (wend - wbase) * CON_STRIDE; idx += G) cwin[idx] = crec[(size_t)wbase * CON_STRIDE + idx];
    gsync();
    // the active contacts of this window: a0 .. a1 - 1 (at most CON_WINDOW <= G of them)
    const int a0 = a;
    int a1;
    bool cut;  // the row budget ended inside this window: the rest of the contacts get no rows
    {
      const int la = a0 + lig;
      const bool inwin = lig < CON_WINDOW && la < nactive && clist[3 * la] < wend;
      const unsigned long long bw = gballot<G>(inwin), bo = gballot<G>(inwin && clist[3 * la + 1] < njmax);
      const int nin = __popcll(bw), nok = __popcll(bo);  // (contacts are in order and their first rows increase: both sets are prefixes)
      a1 = a0 + nok;
      cut = nok < nin;
    }
    if (a1 == a0) {
      if (cut) a = nactive;
      gsync();
      continue;
    }
    // dof loop
    for (a = a0; a < a1; ++a) {
      const int c = clist[3 * a], rbase = clist[3 * a + 1], ndim = clist[3 * a + 2];
      float* cr = cwin + (c - wbase) * CON_STRIDE;
      const int* cri = reinterpret_cast<const int*>(cr);
      const int b1 = gbody[cri[25]], b2 = gbody[cri[26]];
      const V3 cpos = ld3(cr + 1);
      const V3 off1 = cpos - ld3(scom + 3 * broot[b1]);
      const V3 off2 = cpos - ld3(scom + 3 * broot[b2]);
      const V3 f0 = ld3(cr + 4), f1 = ld3(cr + 7), f2 = ld3(cr + 10);
      const int condim = cri[24];
      const float mu[6] = {0.0f, cr[14], cr[30], cr[15], cr[16], cr[31]};
      float vs[3] = {0.0f, 0.0f, 0.0f}, vr[3] = {0.0f, 0.0f, 0.0f};  // this lane's share of the basis velocities (vr: spin / roll, condim > 3 only)
      for (int i0 = 0; i0 < nvp; i0 += G) {
        const int i = i0 + lig;
        V3 jp = V3{0, 0, 0}, jr = V3{0, 0, 0};
        float qv = 0.0f;
        if (i < nv) {
          const bool a1_ = bmask[b1 * nw + (i >> 5)] & (1u << (i & 31));
          const bool a2_ = bmask[b2 * nw + (i >> 5)] & (1u << (i & 31));
          const V3 ang = ld3(cdof + 6 * i), lin = ld3(cdof + 6 * i + 3);
          qv = qvel[i];
          if (a2_) {
            jp = jp + lin + cross(ang, off2);
            jr = jr + ang;
          }
          if (a1_) {
            jp = jp - (lin + cross(ang, off1));
            jr = jr - ang;
          }
        }
        const float comp[6] = {dot(f0, jp), dot(f1, jp), dot(f2, jp), dot(f0, jr), dot(f1, jr), dot(f2, jr)};
        vs[0] += comp[0] * qv; vs[1] += comp[1] * qv; vs[2] += comp[2] * qv;
        if (condim > 3) { vr[0] += comp[3] * qv; vr[1] += comp[4] * qv; vr[2] += comp[5] * qv; }
        if (i < nvp) {
          if (condim == 1) {
            if (rbase < njmax) J[(size_t)rbase * nvp + i] = comp[0];
          } else if (m.cone == CONE_ELLIPTIC) {  // rows = normal, tangent 1, tangent 2, spin, roll 1, roll 2 (constraint.py:3836-3847)
#pragma unroll
            for (int q = 0; q < 6; ++q)
              if (q < ndim && rbase + q < njmax) J[(size_t)(rbase + q) * nvp + i] = comp[q];
          } else {
            // pyramid rows 2(q-1), 2(q-1)+1 = normal component +- mu_q * component q (q = 1..condim-1): tangent 1, tangent 2, spin, roll 1, roll 2
#pragma unroll
            for (int q = 1; q < 6; ++q) {
              if (q < condim && 2 * (q - 1) + 1 < ndim) {
                const int r = rbase + 2 * (q - 1);
                if (r < njmax) J[(size_t)r * nvp + i] = comp[0] + mu[q] * comp[q];
                if (r + 1 < njmax) J[(size_t)(r + 1) * nvp + i] = comp[0] - mu[q] * comp[q];
              }
            }
          }
        }
      }
      // basis velocities of this contact: sums over the dofs (every lane of the group takes part; lanes past nv hold zeros)
      csum_n<G, 3>(vs);
      if (lig < 3) cr[17 + lig] = lig == 0 ? vs[0] : (lig == 1 ? vs[1] : vs[2]);  // (the record's solref / solimp words: the rows below read those from global memory)
      if (condim > 3) {
        csum_n<G, 3>(vr);
        if (lig < 3) cr[20 + lig] = lig == 0 ? vr[0] : (lig == 1 ? vr[1] : vr[2]);
      }
    }
    gsync();
    pc.mark(4);
    // the window's rows (lane per row)
    {
      const int r_lo = clist[3 * a0 + 1], r_hi = min(clist[3 * (a1 - 1) + 1] + clist[3 * (a1 - 1) + 2], njmax);
      for (int r = r_lo + lig; r < r_hi; r += G) {
        const int c = row2con[r] >> 4, dimid = row2con[r] & 15;
        const float* cr = crec + c * CON_STRIDE;
        const int* cri = creci + c * CON_STRIDE;
        const float* pr = cwin + (c - wbase) * CON_STRIDE;
        const float includemargin = cr[13];
        const float pos = cr[0] - includemargin;
        const int condim = cri[24];
        const int b1 = gbody[cri[25]], b2 = gbody[cri[26]];
        float invweight = biw[2 * b1] + biw[2 * b2];
        const bool elliptic = m.cone == CONE_ELLIPTIC && condim > 1;
        float vel;
        if (elliptic) {
          // friction rows of an elliptic contact (constraint.py:4277-4294): regularisation scaled by 1 / impratio and by
          // (mu_1 / mu_dim)^2, no position term in aref
          if (dimid > 0) invweight *= impr2 * impr2;
          if (dimid > 1) {
            const float frii = cr[CON_FRICTION_WORD(dimid - 1)];
            invweight *= cr[14] * cr[14] / (frii * frii);
          }
          vel = pr[17 + dimid];
        } else if (condim > 1) {
          const float fri0 = cr[14];
          invweight = invweight + fri0 * fri0 * invweight;
          invweight = invweight * 2.0f * fri0 * fri0 * impr2 * impr2;
          const int q = 1 + (dimid >> 1);  // the friction component of this pyramid row
          const float muq = cr[CON_FRICTION_WORD(q - 1)], vt = pr[17 + q];
          vel = (dimid & 1) ? pr[17] - muq * vt : pr[17] + muq * vt;
        } else {
          vel = pr[17];
        }
        // friction rows of an elliptic contact take the pair's solreffriction when it is set (constraint.py:4277-4283; explicit pairs only)
        const float* ref = cr + 17;
        if (elliptic && dimid > 0 && m.nexplicit) {
          const int pid = (cri[27] >> 8) - 1;
          if (pid >= 0 && (m.pair_solreffriction[2 * pid] != 0.0f || m.pair_solreffriction[2 * pid + 1] != 0.0f)) ref = m.pair_solreffriction + 2 * pid;
        }
        EfcRowOut eo_ = efc_row(dsbl, timestep, elliptic && dimid > 0 ? 0.0f : pos, pos, invweight, ref, cr + 19, includemargin, vel);
        d.efc_D[eo + r] = eo_.D;
        d.efc_aref[eo + r] = eo_.aref;
        d.efc_pos[eo + r] = eo_.pos;
        d.efc_margin[eo + r] = includemargin;
        d.efc_vel[eo + r] = vel;
        d.efc_frictionloss[eo + r] = 0.0f;
        d.efc_type[eo + r] = condim == 1 ? CT_CONTACT_FRICTIONLESS : (elliptic ? CT_CONTACT_ELLIPTIC : CT_CONTACT_PYRAMIDAL);
        d.efc_id[eo + r] = c;  // world-local; k_publish_contacts rewrites it with the public contact id
        if (m.cone == CONE_ELLIPTIC || m.tree_solve) d.ws_efc_con[eo + r] = row2con[r];  // the solver groups the rows of a contact / of a tree
      }
    }
    gsync();  // (the next window overwrites the compact records)
    a = a1;
    if (cut) a = nactive;
  }
  pc.mark(5);
  if (lig == 0) {
    d.ne[w] = ne;
    d.nf[w] = nf;
    d.nl[w] = nl;
    d.nefc[w] = nefc;
    if (nefc > njmax) atomicOr(d.overflow + w, OVF_NEFC);
  }
  pc.mark(5);
}

template <int G>
__global__ void __launch_bounds__(256) k_make_constraint(MjhModel m, MjhData d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  make_constraint_body<G>(m, d, smem, blk_of_launch<G>());
}


// ---- CSR view of efc.J (opt-in, mjh_efc_j_sparse) --------------------------------------------------------------------------------
// one 64-lane wavefront per world: rows in order, each row's non-zeros compacted with a ballot prefix over 64-column chunks
__global__ void __launch_bounds__(64) k_efc_j_sparse(MjhData d, int nv, int njmax_nnz, int* rownnz, int* rowadr, int* colind, float* values) {
  const int w = blockIdx.x, lane = threadIdx.x;
  if (w >= d.nworld) return;
  const int nefc = min(d.nefc[w], d.njmax), nvp = d.nv_pad;
  const float* J = d.efc_J + (size_t)w * d.njmax_pad * nvp;
  int adr = 0;
  bool ovf = false;
  for (int r = 0; r < d.njmax; ++r) {
    int nnz = 0;
    if (r < nefc) {
      for (int c0 = 0; c0 < nv; c0 += 64) {
        const int c = c0 + lane;
        const float v = c < nv ? J[(size_t)r * nvp + c] : 0.0f;
        const unsigned long long bits = __ballot(v != 0.0f);
        const int rank = __popcll(bits & ((1ull << lane) - 1ull)), o = adr + nnz + rank;
        if (v != 0.0f) {
          if (o < njmax_nnz) {
            colind[(size_t)w * njmax_nnz + o] = c;
            values[(size_t)w * njmax_nnz + o] = v;
          } else {
            ovf = true;
          }
        }
        nnz += __popcll(bits);
      }
    }
    if (lane == 0) {
      rownnz[(size_t)w * d.njmax + r] = min(nnz, max(njmax_nnz - adr, 0));
      rowadr[(size_t)w * d.njmax + r] = min(adr, njmax_nnz);
    }
    adr += nnz;
  }
  if (__ballot(ovf) && lane == 0) atomicOr(d.overflow + w, OVF_NJMAX_NNZ);
}


// ---- constraint islands for nv > 64 (MjhModel.tree_solve) ---------------------------------------------------------------------------
// M is block diagonal over kinematic trees, so the constraint problem of a world separates over the connected components of the graph
// whose nodes are trees and whose edges are the rows touching two trees (a contact between bodies of two trees, a joint equality across
// them) -- MuJoCo's constraint islands at tree granularity (reference island.py).  One 64-lane wavefront per world:
//   1. tree(s) of every row from its type: equalities and limits through their joint, dof friction through its dof, contacts through
//      the bodies of their geoms (a static body has no tree);
//   2. connected components by min-label propagation over the coupling rows (LDS atomics, at most ntree sweeps);
//   3. islands numbered by their smallest tree; their dofs (tree ranges concatenated in tree order) listed in ws_isl_dofmap with the
//      inverse in ws_isl_dofinv, their rows grouped stably in ws_tree_rowmap; ws_isl_dofadr / ws_tree_rowadr delimit island k.
// ws_separable = 1 when every island has at most 64 dofs: the world is then solved per island by the register-resident kernels
// (solve_body TREE), otherwise by the generic solver.
// clears the list counters of k_tree_rows (a kernel, not hipMemsetAsync: a memset node inside the captured step graph faulted on ROCm 7.2)
__global__ void __launch_bounds__(64) k_isl_clear(MjhData d) {
  if (threadIdx.x < 4) d.ws_isl_count[threadIdx.x] = 0;
}
__global__ void __launch_bounds__(64) k_tree_rows(MjhModel m, MjhData d) {
  extern __shared__ int sh_isl[];  // [njmax] first tree of each row | [njmax] second tree or -1 | [ntree] label | [ntree] island of tree
  const int w = blockIdx.x, lane = threadIdx.x;
  if (w >= d.nworld) return;
  const int njmax = d.njmax, nefc = min(d.nefc[w], njmax), ntree = m.ntree, nv = m.nv;
  int *rt1 = sh_isl, *rt2 = sh_isl + njmax, *label = sh_isl + 2 * njmax, *isl_of = label + ntree;
  const size_t eo = (size_t)w * njmax;
  int* rowadr = d.ws_tree_rowadr + (size_t)w * (ntree + 1);
  int* dofadr = d.ws_isl_dofadr + (size_t)w * (ntree + 1);
  int* rowmap = d.ws_tree_rowmap + eo;
  int* dmap = d.ws_isl_dofmap + (size_t)w * nv;
  int* dinv = d.ws_isl_dofinv + (size_t)w * nv;
  for (int t = lane; t < ntree; t += 64) label[t] = t;
  for (int r = lane; r < nefc; r += 64) {
    const int type = d.efc_type[eo + r];
    int t1 = 0, t2 = -1;
    if (type == CT_EQUALITY) {
      const int e = d.efc_id[eo + r], j1 = m.eq_obj1id[e], j2 = m.eq_obj2id[e];
      t1 = m.dof_treeid[m.jnt_dofadr[j1]];
      if (j2 >= 0) t2 = m.dof_treeid[m.jnt_dofadr[j2]];
    } else if (type == CT_FRICTION_DOF) {
      t1 = m.dof_treeid[d.efc_id[eo + r]];
    } else if (type == CT_LIMIT_JOINT) {
      t1 = m.dof_treeid[m.jnt_dofadr[d.efc_id[eo + r]]];
    } else {  // contact rows (efc.id is being rewritten by the contact publication: the row -> contact map is ws_efc_con)
      const int c = d.ws_efc_con[eo + r] >> 4;
      const int* rec = reinterpret_cast<const int*>(d.ws_contact + ((size_t)w * d.concap + c) * CON_STRIDE);
      const int a = m.body_treeid[m.geom_bodyid[rec[25]]], b = m.body_treeid[m.geom_bodyid[rec[26]]];
      t1 = a >= 0 ? a : (b >= 0 ? b : 0);  // (two static bodies: an all-zero row, any tree will do)
      t2 = (a >= 0 && b >= 0) ? b : -1;
    }
    if (t2 == t1) t2 = -1;
    rt1[r] = t1;
    rt2[r] = t2;
  }
  __syncthreads();
  for (int sweep = 0; sweep < ntree; ++sweep) {  // labels only decrease; a path of k trees needs at most k sweeps
    for (int r = lane; r < nefc; r += 64)
      if (rt2[r] >= 0) {
        const int lo = min(label[rt1[r]], label[rt2[r]]);
        atomicMin(&label[rt1[r]], lo);
        atomicMin(&label[rt2[r]], lo);
      }
    __syncthreads();
    for (int t = lane; t < ntree; t += 64) label[t] = label[label[t]];
    __syncthreads();
  }
  // islands in the order of their smallest tree (its label is itself): lane 0 walks the (few) trees and records for every tree its
  // island and the first island-local dof, then all lanes fill the dof maps
  int nisland = 0;
  bool fits = true;
  if (lane == 0) {
    int adr = 0;
    for (int t = 0; t < ntree; ++t) {
      if (label[t] != t) continue;
      dofadr[nisland] = adr;
      for (int u = t; u < ntree; ++u)
        if (label[u] == t) {
          isl_of[u] = nisland;
          label[u] = -1 - adr;  // (the label has served: keep the tree's first slot in the dof map, tagged negative)
          adr += m.tree_dofnum[u];
        }
      if (adr - dofadr[nisland] > 64) fits = false;
      ++nisland;
    }
    for (int k = nisland; k <= ntree; ++k) dofadr[k] = adr;
  }
  __syncthreads();
  for (int i = lane; i < nv; i += 64) {
    const int t = m.dof_treeid[i], slot0 = -1 - label[t], pos = slot0 + i - m.tree_dofadr[t];
    dmap[pos] = i;
    dinv[i] = pos - dofadr[isl_of[t]];
  }
  nisland = __builtin_amdgcn_readfirstlane(nisland);
  __syncthreads();
  int adr = 0;
  for (int k = 0; k < nisland; ++k) {
    if (lane == 0) rowadr[k] = adr;
    for (int r0 = 0; r0 < nefc; r0 += 64) {
      const int r = r0 + lane;
      const bool mine = r < nefc && isl_of[rt1[r]] == k;
      const unsigned long long bits = __ballot(mine);
      if (mine) rowmap[adr + __popcll(bits & ((1ull << lane) - 1ull))] = r;
      adr += __popcll(bits);
    }
  }
  if (lane == 0) {
    for (int k = nisland; k <= ntree; ++k) rowadr[k] = adr;
    rowadr[nisland] = adr;
    int flags = 0;  // which rare island classes this world holds (the LOOPED launches of solve_tree.hpp test them)
    for (int k = 0; k < nisland; ++k) {
      const int nd = dofadr[k + 1] - dofadr[k], nr = rowadr[k + 1] - rowadr[k];
      if (nd > 32) flags |= ISL_WIDE;
      else if (nr > 64) flags |= ISL_MANYROWS;
      if (nr > 192) fits = false;  // (the register-resident kernels end at 192 rows per island: 6 x 32 / 3 x 64 lanes -- the generic solver takes the world)
    }
    d.ws_isl_flags[w] = fits ? flags : 0;
    // worlds of the rare classes are listed for the looped launches (one atomic each; the counters are cleared before this kernel)
    if (!fits) d.ws_isl_list[(size_t)ISL_LIST_GENERIC * d.nworld + atomicAdd(d.ws_isl_count + ISL_LIST_GENERIC, 1)] = w;
    else {
      if (flags & ISL_WIDE) d.ws_isl_list[(size_t)ISL_LIST_WIDE * d.nworld + atomicAdd(d.ws_isl_count + ISL_LIST_WIDE, 1)] = w;
      if (flags & ISL_MANYROWS) d.ws_isl_list[(size_t)ISL_LIST_MANYROWS * d.nworld + atomicAdd(d.ws_isl_count + ISL_LIST_MANYROWS, 1)] = w;
    }
    d.ws_nisland[w] = nisland;
    d.ws_separable[w] = fits ? 1 : 0;
    d.solver_niter[w] = 0;
  }
}


// ---- opt-in: dense upper Cholesky factors of M's tree blocks in the reference's packed qLD layout (mjh_qld_dense) --------------------
// one 64-lane wavefront per (world, tree), n <= 64: the block is densified in LDS, factored column by column (lane = column of the
// trailing update), and stored as U (M = U^T U) row-major with zeros below the diagonal
__global__ void __launch_bounds__(64) k_qld_dense(MjhModel m, MjhData d, float* out, int stride) {
  __shared__ float A[64 * 65];
  const int w = blockIdx.x / m.ntree, t = blockIdx.x % m.ntree, lane = threadIdx.x;
  if (w >= d.nworld) return;
  const int d0 = m.tree_dofadr[t], n = m.tree_dofnum[t];
  if (n > 64) return;
  int off = 0;
  for (int u = 0; u < t; ++u) off += m.tree_dofnum[u] <= 64 ? m.tree_dofnum[u] * m.tree_dofnum[u] : 0;
  const float* Mg = d.M + (size_t)w * m.nC;
  for (int idx = lane; idx < n * 65; idx += 64) A[idx] = 0.0f;
  __syncthreads();
  for (int i = lane; i < n; i += 64) {
    const int start = m.M_rowadr[d0 + i], nn = m.M_rownnz[d0 + i];
    for (int a = 0; a < nn; ++a) {
      const int j = m.M_colind[start + a] - d0;
      A[i * 65 + j] = Mg[start + a];
      A[j * 65 + i] = Mg[start + a];
    }
  }
  __syncthreads();
  for (int k = 0; k < n; ++k) {  // U[k][k..] = A[k][k..] / sqrt(A[k][k]); A[i][j] -= U[k][i] U[k][j]
    const float piv = sqrtf(fmaxf(A[k * 65 + k], MJ_MINVAL));
    __syncthreads();
    if (lane >= k && lane < n) A[k * 65 + lane] = A[k * 65 + lane] / piv;
    __syncthreads();
    if (lane > k && lane < n) {
      const float ukj = A[k * 65 + lane];
      for (int i = k + 1; i <= lane; ++i) A[i * 65 + lane] -= A[k * 65 + i] * ukj;
    }
    __syncthreads();
  }
  float* o = out + (size_t)w * stride + off;
  for (int idx = lane; idx < n * n; idx += 64) {
    const int i = idx / n, j = idx % n;
    o[idx] = j >= i ? A[i * 65 + j] : 0.0f;
  }
}
