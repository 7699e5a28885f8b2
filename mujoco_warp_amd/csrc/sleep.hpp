// Sleeping / waking of kinematic trees and tree-level constraint islands (reference sleep.py, island.py:28-310, the sleep hooks of
// forward.py:345-349, 652-675, 1273-1278, 1344-1347 and the compacted solve solver.py:3899-4135).
//
// The reference keeps an active-dof set, gathers the awake dofs into dense nvmax-wide arrays and runs its solver there.  This engine
// solves one world per lane group with the whole model resident, so "compaction" would save nothing; instead the sleeping dofs are
// frozen in place: M is block diagonal over kinematic trees, hence the compacted solve equals the full-size solve with the sleeping
// dofs' columns of J, their warm start and their smooth force zeroed (the gradient then vanishes identically on those dofs and they
// never move).  k_sleep_mask writes that masked copy of efc.J / qacc_warmstart for the solver; the public efc.J stays complete.
//
// The bookkeeping itself (wake / sleep cycles / island labels: a few integer tables per world, loops over ntree and over rows) is serial
// per world -- union-finds and cycle walks whose result the tests compare table for table with the oracle -- so one lane walks it.  Round 3:
// one WAVEFRONT per world instead of one thread per world: the 64 lanes stage everything the walk touches in LDS with coalesced loads (the
// world's sleep tables, velocities, applied forces, row types / ids, the geoms and first rows of its contacts), lane 0 runs the unchanged
// serial code on that world-local view (`MjhData` rebased to the world, w = 0), the awake lists are rebuilt by all lanes (ballot ranks)
// and the lanes write the tables back.  The first version (one thread per world on world-major rows: every load a cache line of its own)
// cost 119 us per call, four calls per step -- 15 % of the clutter_synth step.
// The step of such models is the staged launch sequence in mjhip.hip (run_sleep_step), not the fused four-launch step.
#pragma once
#include "dev_common.hpp"

#define SLEEP_STATIC (-1)
#define SLEEP_ASLEEP 0
#define SLEEP_AWAKE 1
#define SLEEP_POLICY_AUTO_NEVER 1
#define MJ_MINAWAKE 10                       /* mjMINAWAKE (types.py:29) */
#define SLEEP_AWAKE_VAL (-(1 + MJ_MINAWAKE)) /* sleep.py:28 */
enum { SLP_WAKE = 0, SLP_WAKE_COLLISION = 1, SLP_POST_CONSTRAINT = 2, SLP_SLEEP = 3, SLP_UPDATE = 4, SLP_WAKE_EQUALITY = 5, SLP_ISLAND = 6 };

// sleep.py:171-215 update_sleep (flg_staticawake = 0)
DEV void sleep_update(const MjhModel& m, const MjhData& d, int w) {
  const int nt = m.ntree, nb = m.nbody, nv = m.nv;
  const int* asleep = d.tree_asleep + (size_t)w * nt;
  int* tawake = d.tree_awake + (size_t)w * nt;
  int* bawake = d.body_awake + (size_t)w * nb;
  int* bind = d.body_awake_ind + (size_t)w * nb;
  int* dind = d.dof_awake_ind + (size_t)w * nv;
  int ntree_awake = 0, nbody_awake = 0, nv_awake = 0;
  for (int t = 0; t < nt; ++t) {
    const int a = asleep[t] < 0;
    tawake[t] = a;
    ntree_awake += a;
  }
  for (int b = 0; b < nb; ++b) {
    const int tree = m.body_treeid[b];
    int state;
    if (tree < 0) state = m.body_mocapid[m.body_rootid[b]] >= 0 ? SLEEP_AWAKE : SLEEP_STATIC;
    else state = tawake[tree] ? SLEEP_AWAKE : SLEEP_ASLEEP;
    bawake[b] = state;
    if (state != SLEEP_ASLEEP) bind[nbody_awake++] = b;
  }
  for (int i = 0; i < nv; ++i) {
    const int b = m.dof_bodyid[i];
    if (m.body_treeid[b] >= 0 && bawake[b] == SLEEP_AWAKE) dind[nv_awake++] = i;
  }
  d.ntree_awake[w] = ntree_awake;
  d.nbody_awake[w] = nbody_awake;
  d.nv_awake[w] = nv_awake;
}
// sleep.py:33: smallest tree of the sleep cycle through treeid, -1 when awake
DEV int sleep_cycle(const int* asleep, int nt, int treeid) {
  if (treeid < 0 || treeid >= nt) return -1;
  int smallest = treeid, current = treeid;
  for (int step = 0; step < nt + 1; ++step) {
    const int next = asleep[current];
    if (next < 0 || next >= nt) return -1;
    if (next < smallest) smallest = next;
    current = next;
    if (current == treeid) break;
  }
  return smallest;
}
// sleep.py:236: wakes the tree and the rest of its sleep cycle
DEV void sleep_wake_tree(int* asleep, int nt, int treeid, int wakeval) {
  if (treeid < 0 || treeid >= nt) return;
  const int val = asleep[treeid];
  if (val < 0) {
    if (wakeval < val) asleep[treeid] = wakeval;
    return;
  }
  int current = treeid;
  for (int step = 0; step < nt + 1; ++step) {
    const int next = asleep[current];
    if (next < 0 || next >= nt) break;
    asleep[current] = wakeval;
    current = next;
    if (current == treeid) break;
  }
}
// (sleep.py:273 tree_can_sleep: policy check in sleep_phase, the force / velocity scans by all lanes in k_sleep)
DEV int sleep_find(const int* parent, int x) {
  while (parent[x] != x) x = parent[x];
  return x;
}
DEV void sleep_edge(int* parent, int t0, int t1) {  // island.py:120-134: self edge or cross-tree edge
  if (t0 < 0 && t1 >= 0) {
    t0 = t1;
    t1 = -1;
  }
  if (t0 < 0) return;
  if (parent[t0] == -1) parent[t0] = t0;
  if (t1 < 0 || t1 == t0) return;
  if (parent[t1] == -1) parent[t1] = t1;
  const int r0 = sleep_find(parent, t0), r1 = sleep_find(parent, t1);
  if (r0 < r1) parent[r1] = r0;
  else if (r1 < r0) parent[r0] = r1;
}
// island.py:28-310: trees joined by constraint rows; islands numbered in the order of their smallest tree (what the reference's
// flood fill over ascending trees yields), -1 for trees without rows.  Union-find with the smaller root winning, so that a root IS
// the smallest tree of its island.  (Equality rows: the reference scans the Jacobian row; for joint equalities -- the only type on
// this path -- the non-zeros are the dofs of the two joints.)
// cg: per contact (tree of geom 1, tree of geom 2, first constraint row), staged by k_sleep
DEV void sleep_island(const MjhModel& m, const MjhData& d, int w, const int* cg) {
  const int nt = m.ntree, njmax = d.njmax;
  int* isl = d.tree_island + (size_t)w * nt;
  for (int t = 0; t < nt; ++t) isl[t] = -1;
  const size_t eo = (size_t)w * njmax;
  const int nefc = min(d.nefc[w], njmax);
  for (int r = 0; r < nefc; ++r) {
    const int type = d.efc_type[eo + r];
    if (type == CT_EQUALITY) {
      const int e = d.efc_id[eo + r], j1 = m.eq_obj1id[e], j2 = m.eq_obj2id[e];
      sleep_edge(isl, m.dof_treeid[m.jnt_dofadr[j1]], j2 >= 0 ? m.dof_treeid[m.jnt_dofadr[j2]] : -1);
    } else if (type == CT_FRICTION_DOF) {
      sleep_edge(isl, m.dof_treeid[d.efc_id[eo + r]], -1);
    } else if (type == CT_LIMIT_JOINT) {
      sleep_edge(isl, m.dof_treeid[m.jnt_dofadr[d.efc_id[eo + r]]], -1);
    }
  }
  const int ncon = min(d.ws_ncon[w], d.concap);
  for (int c = 0; c < ncon; ++c) {  // contacts that own constraint rows (record words 28-29 are filled by make_constraint)
    if (cg[3 * c + 2] < 0 || cg[3 * c + 2] >= njmax) continue;
    sleep_edge(isl, cg[3 * c], cg[3 * c + 1]);
  }
  int nisland = 0;
  for (int t = 0; t < nt; ++t) {  // roots precede their members: number the root, members copy the (negative) code
    if (isl[t] == -1) continue;
    if (isl[t] == t) isl[t] = -(2 + nisland++);
    else {
      int r = isl[t];
      while (r >= 0 && isl[r] >= 0 && isl[r] != r) r = isl[r];  // (walk to the root: either still a tree id or already coded)
      isl[t] = isl[r] < -1 ? isl[r] : isl[t];
    }
  }
  for (int t = 0; t < nt; ++t)
    if (isl[t] < -1) isl[t] = -(isl[t] + 2);
  d.nisland[w] = nisland;
}

// the serial part of one phase on a world-local view of Data (see k_sleep)
// busy[t] != 0: tree t fails the force / velocity part of tree_can_sleep (computed by all lanes in k_sleep for the phase's tolerance)
DEV void sleep_phase(const MjhModel& m, const MjhData& d, int w, int phase, const int* cg, const int* busy) {
  auto can_sleep = [&](int t) { return m.tree_sleep_policy[t] != SLEEP_POLICY_AUTO_NEVER && !busy[t]; };
  const int nt = m.ntree;
  int* asleep = d.tree_asleep + (size_t)w * nt;
  const int* tawake = d.tree_awake + (size_t)w * nt;
  if (phase == SLP_WAKE) {  // sleep.py:325, 721: user changes (velocity, applied forces, a stale awake table) wake a sleeping tree
    for (int t = 0; t < nt; ++t) {
      if (asleep[t] < 0) continue;
      if (tawake[t] == 1 || !can_sleep(t)) sleep_wake_tree(asleep, nt, t, SLEEP_AWAKE_VAL);
    }
  } else if (phase == SLP_WAKE_COLLISION) {  // sleep.py:367, 744: a contact between an awake and a sleeping tree wakes the latter
    bool woke = false;
    const int ncon = min(d.ws_ncon[w], d.concap);
    for (int c = 0; c < ncon; ++c) {
      const int t1 = cg[3 * c], t2 = cg[3 * c + 1];
      if (t1 < 0 || t2 < 0) continue;
      const int a1 = tawake[t1], a2 = tawake[t2];
      if (a1 == a2) continue;
      sleep_wake_tree(asleep, nt, a1 == 1 ? t2 : t1, a1 == 1 ? asleep[t1] : asleep[t2]);
      woke = true;
    }
    d.ws_sleep_flag[w] = woke ? 1 : 0;
  } else if (phase == SLP_POST_CONSTRAINT || phase == SLP_WAKE_EQUALITY || phase == SLP_ISLAND) {  // sleep.py:579, 793 wake_equality (joint equalities), then island.island
    if (phase != SLP_ISLAND && !(m.disableflags & (DSBL_CONSTRAINT | DSBL_EQUALITY)))
      for (int e = 0; e < m.neq; ++e) {
        if (!d.eq_active[(size_t)w * m.neq + e]) continue;
        const int id1 = m.eq_obj1id[e], id2 = m.eq_obj2id[e];
        const int t1 = id1 >= 0 ? m.body_treeid[m.jnt_bodyid[id1]] : -1, t2 = id2 >= 0 ? m.body_treeid[m.jnt_bodyid[id2]] : -1;
        const int s1 = t1 >= 0 ? tawake[t1] : SLEEP_STATIC, s2 = t2 >= 0 ? tawake[t2] : SLEEP_STATIC;
        if (s1 != SLEEP_ASLEEP && s2 != SLEEP_ASLEEP) continue;
        if (s1 == SLEEP_STATIC || s2 == SLEEP_STATIC) continue;
        if (t1 == t2) continue;
        if (s1 == SLEEP_ASLEEP && s2 == SLEEP_ASLEEP) {
          if (sleep_cycle(asleep, nt, t1) != sleep_cycle(asleep, nt, t2)) {
            sleep_wake_tree(asleep, nt, t1, SLEEP_AWAKE_VAL);
            sleep_wake_tree(asleep, nt, t2, SLEEP_AWAKE_VAL);
          }
        } else {
          sleep_wake_tree(asleep, nt, s1 == SLEEP_ASLEEP ? t1 : t2, SLEEP_AWAKE_VAL);
        }
      }
    if (phase != SLP_WAKE_EQUALITY) sleep_island(m, d, w, cg);
  } else if (phase == SLP_SLEEP) {  // sleep.py:824-999
    float* qvel = d.qvel + (size_t)w * m.nv;
    float* qacc = d.qacc + (size_t)w * m.nv;
    const int* isl = d.tree_island + (size_t)w * nt;
    const int nisland = d.nisland[w];
    for (int t = 0; t < nt; ++t) {  // 1. awake trees count towards sleep while they could sleep
      const int val = asleep[t];
      if (val >= 0) continue;
      if (can_sleep(t)) {
        if (val < -1) asleep[t] = val + 1;
      } else {
        asleep[t] = SLEEP_AWAKE_VAL;
      }
    }
    for (int k = 0; k < nisland; ++k) {  // 2 + 3. an island sleeps when every tree in it is ready: link its trees into a cycle
      bool can = true;
      for (int t = 0; t < nt; ++t)
        if (isl[t] == k && asleep[t] < -1) can = false;
      if (!can) continue;
      int first = -1, prev = -1;
      for (int t = 0; t < nt; ++t) {
        if (isl[t] != k) continue;
        if (first == -1) first = t;
        if (prev != -1) asleep[prev] = t;
        prev = t;
        for (int j = 0; j < m.tree_dofnum[t]; ++j) qvel[m.tree_dofadr[t] + j] = qacc[m.tree_dofadr[t] + j] = 0.0f;
      }
      if (first != -1) asleep[prev] = first;
    }
    for (int t = 0; t < nt; ++t) {  // unconstrained trees sleep on their own
      if (isl[t] >= 0 && isl[t] < nisland) continue;
      if (asleep[t] == -1) asleep[t] = t;
      if (asleep[t] >= 0)
        for (int j = 0; j < m.tree_dofnum[t]; ++j) qvel[m.tree_dofadr[t] + j] = qacc[m.tree_dofadr[t] + j] = 0.0f;
    }
  }
}

// LDS of one world in k_sleep (ints): the sleep tables, the float vectors the walk reads, row types / ids, (tree 1, tree 2, first row) per contact
struct SleepLds {
  int asleep, tawake, bawake, bind, dind, isl, busy, qvel, qacc, qfrc, xfrc, etype, eid, cg, total;
};
__host__ __device__ inline SleepLds sleep_lds(int nt, int nb, int nv, int njmax, int concap) {
  SleepLds p;
  int o = 0;
  p.asleep = o; o += nt;
  p.tawake = o; o += nt;
  p.isl = o; o += nt;
  p.busy = o; o += nt;
  p.bawake = o; o += nb;
  p.bind = o; o += nb;
  p.dind = o; o += nv;
  p.qvel = o; o += nv;
  p.qacc = o; o += nv;
  p.qfrc = o; o += nv;
  p.xfrc = o; o += 6 * nb;
  p.etype = o; o += njmax;
  p.eid = o; o += njmax;
  p.cg = o; o += 3 * concap;
  p.total = o + 4;
  return p;
}
__global__ void __launch_bounds__(64) k_sleep(MjhModel m, MjhData d, int phase) {
  extern __shared__ int sl[];
  const int w = blockIdx.x, lane = threadIdx.x;
  if (w >= d.nworld) return;
  const int nt = m.ntree, nb = m.nbody, nv = m.nv, njmax = d.njmax;
  const SleepLds L = sleep_lds(nt, nb, nv, njmax, d.concap);
  int *asleep = sl + L.asleep, *tawake = sl + L.tawake, *isl = sl + L.isl, *bawake = sl + L.bawake, *bind = sl + L.bind, *dind = sl + L.dind,
      *etype = sl + L.etype, *eid = sl + L.eid, *cg = sl + L.cg, *busy = sl + L.busy;
  float *qvel = reinterpret_cast<float*>(sl + L.qvel), *qacc = reinterpret_cast<float*>(sl + L.qacc), *qfrc = reinterpret_cast<float*>(sl + L.qfrc),
        *xfrc = reinterpret_cast<float*>(sl + L.xfrc);
  const int nefc = min(d.nefc[w], njmax), ncon = min(d.ws_ncon[w], d.concap);
  const bool rows = phase == SLP_POST_CONSTRAINT || phase == SLP_ISLAND, cons = rows || phase == SLP_WAKE_COLLISION;
  const bool vel = phase == SLP_WAKE || phase == SLP_SLEEP;
  // ---- stage (coalesced) ----
  for (int i = lane; i < nt; i += 64) {
    asleep[i] = d.tree_asleep[(size_t)w * nt + i];
    tawake[i] = d.tree_awake[(size_t)w * nt + i];
    isl[i] = d.tree_island[(size_t)w * nt + i];
  }
  if (vel) {
    for (int i = lane; i < nv; i += 64) {
      qvel[i] = d.qvel[(size_t)w * nv + i];
      qacc[i] = d.qacc[(size_t)w * nv + i];
      qfrc[i] = d.qfrc_applied[(size_t)w * nv + i];
    }
    for (int i = lane; i < 6 * nb; i += 64) xfrc[i] = d.xfrc_applied[(size_t)w * 6 * nb + i];
  }
  if (rows)
    for (int i = lane; i < nefc; i += 64) {
      etype[i] = d.efc_type[(size_t)w * njmax + i];
      eid[i] = d.efc_id[(size_t)w * njmax + i];
    }
  if (cons)
    for (int c = lane; c < ncon; c += 64) {
      const int* rec = reinterpret_cast<const int*>(d.ws_contact + ((size_t)w * d.concap + c) * CON_STRIDE);
      cg[3 * c] = m.body_treeid[m.geom_bodyid[rec[25]]];
      cg[3 * c + 1] = m.body_treeid[m.geom_bodyid[rec[26]]];
      cg[3 * c + 2] = rec[28];
    }
  gsync();
  // ---- sleep.py:273 tree_can_sleep, the part that scans bodies and dofs: applied forces and velocities per tree, all lanes (the serial walk
  // looked at every body for every tree: ntree x nbody loads, 100 us for 22 trees x 45 bodies) ----
  if (vel) {
    for (int t = lane; t < nt; t += 64) busy[t] = 0;
    gsync();
    const float tol = phase == SLP_WAKE ? 0.0f : m.opt_sleep_tolerance;
    for (int b = lane; b < nb; b += 64) {
      const int t = m.body_treeid[b];
      bool f = false;
      for (int i = 0; i < 6; ++i) f = f || xfrc[6 * b + i] != 0.0f;
      if (t >= 0 && f) busy[t] = 1;  // (every writer stores the same value)
    }
    for (int i = lane; i < nv; i += 64) {
      const float v = qvel[i];
      const bool f = qfrc[i] != 0.0f || (tol > 0.0f ? fabsf(m.dof_length[i] * v) >= tol : v != 0.0f);
      if (f) busy[m.dof_treeid[i]] = 1;
    }
    gsync();
  }
  // ---- the serial walk on the world-local view (every per-world array it touches rebased to this world: used with w = 0) ----
  if (lane == 0) {
    MjhData v = d;
    v.tree_asleep = asleep; v.tree_awake = tawake; v.tree_island = isl; v.body_awake = bawake; v.body_awake_ind = bind; v.dof_awake_ind = dind;
    v.qvel = qvel; v.qacc = qacc; v.qfrc_applied = qfrc; v.xfrc_applied = xfrc; v.efc_type = etype; v.efc_id = eid;
    v.nefc = d.nefc + w; v.ws_ncon = d.ws_ncon + w; v.nisland = d.nisland + w; v.ws_sleep_flag = d.ws_sleep_flag + w;
    v.eq_active = d.eq_active + (size_t)w * m.neq;
    sleep_phase(m, v, 0, phase, cg, busy);
  }
  gsync();
  // ---- sleep.py:171-215 update_sleep (flg_staticawake = 0) by all lanes: ordered compaction with ballot ranks ----
  int ntree_awake = 0, nbody_awake = 0, nv_awake = 0;
  for (int t0 = 0; t0 < nt; t0 += 64) {
    const int t = t0 + lane;
    const bool a = t < nt && asleep[t] < 0;
    if (t < nt) tawake[t] = a ? 1 : 0;
    ntree_awake += __popcll(__ballot(a));
  }
  gsync();
  for (int b0 = 0; b0 < nb; b0 += 64) {
    const int b = b0 + lane;
    int state = SLEEP_ASLEEP;
    if (b < nb) {
      const int tree = m.body_treeid[b];
      if (tree < 0) state = m.body_mocapid[m.body_rootid[b]] >= 0 ? SLEEP_AWAKE : SLEEP_STATIC;
      else state = tawake[tree] ? SLEEP_AWAKE : SLEEP_ASLEEP;
      bawake[b] = state;
    }
    const bool keep = b < nb && state != SLEEP_ASLEEP;
    const unsigned long long bits = __ballot(keep);
    if (keep) bind[nbody_awake + __popcll(bits & ((1ull << lane) - 1ull))] = b;
    nbody_awake += __popcll(bits);
  }
  gsync();
  for (int i0 = 0; i0 < nv; i0 += 64) {
    const int i = i0 + lane;
    bool keep = false;
    if (i < nv) {
      const int b = m.dof_bodyid[i];
      keep = m.body_treeid[b] >= 0 && bawake[b] == SLEEP_AWAKE;
    }
    const unsigned long long bits = __ballot(keep);
    if (keep) dind[nv_awake + __popcll(bits & ((1ull << lane) - 1ull))] = i;
    nv_awake += __popcll(bits);
  }
  gsync();
  // ---- write back (entries of the index lists past the counts keep their old values, as before) ----
  for (int i = lane; i < nt; i += 64) {
    d.tree_asleep[(size_t)w * nt + i] = asleep[i];
    d.tree_awake[(size_t)w * nt + i] = tawake[i];
    if (rows) d.tree_island[(size_t)w * nt + i] = isl[i];
  }
  for (int i = lane; i < nb; i += 64) d.body_awake[(size_t)w * nb + i] = bawake[i];
  for (int i = lane; i < nbody_awake; i += 64) d.body_awake_ind[(size_t)w * nb + i] = bind[i];
  for (int i = lane; i < nv_awake; i += 64) d.dof_awake_ind[(size_t)w * nv + i] = dind[i];
  if (phase == SLP_SLEEP)
    for (int i = lane; i < nv; i += 64) {
      d.qvel[(size_t)w * nv + i] = qvel[i];
      d.qacc[(size_t)w * nv + i] = qacc[i];
    }
  if (lane == 0) {
    d.ntree_awake[w] = ntree_awake;
    d.nbody_awake[w] = nbody_awake;
    d.nv_awake[w] = nv_awake;
  }
}

// the solver's view of a world with sleeping trees: efc.J with the sleeping dofs' columns zeroed, qacc_warmstart likewise
__global__ void __launch_bounds__(256) k_sleep_mask(MjhModel m, MjhData d) {
  const int w = blockIdx.x;
  const int nvp = d.nv_pad, nv = m.nv, nefc = min(d.nefc[w], d.njmax);
  const int* tawake = d.tree_awake + (size_t)w * m.ntree;
  const size_t jo = (size_t)w * d.njmax_pad * nvp;
  for (int idx = threadIdx.x; idx < nefc * nvp; idx += 256) {
    const int c = idx % nvp;
    const bool awake = c < nv ? tawake[m.dof_treeid[c]] != 0 : true;
    d.ws_sleep_J[jo + idx] = awake ? d.efc_J[jo + idx] : 0.0f;
  }
  for (int i = threadIdx.x; i < nv; i += 256) d.ws_sleep_warm[(size_t)w * nv + i] = tawake[m.dof_treeid[i]] ? d.qacc_warmstart[(size_t)w * nv + i] : 0.0f;
  // the reference compacts the awake dofs into nvmax-wide arrays and flags a world that does not fit (island.py:1010-1019, "behavior
  // undefined"); here nothing is compacted, so the world is solved in full and only the flag is raised
  if (threadIdx.x == 0 && d.nv_awake[w] > d.nvmax) atomicOr(d.overflow + w, OVF_NVMAX);
}
// forward.py:1273-1278: no smooth force on the dofs of a sleeping tree (the L'DL solve of a zero right-hand side then leaves their
// qacc_smooth at exactly 0, the reference's "frozen inactive DOF", solver.py:3896)
__global__ void __launch_bounds__(256) k_sleep_qfrc(MjhModel m, MjhData d) {
  const int idx = blockIdx.x * 256 + threadIdx.x, nv = m.nv;
  if (idx >= d.nworld * nv) return;
  const int w = idx / nv, i = idx - w * nv;
  if (!d.tree_awake[(size_t)w * m.ntree + m.dof_treeid[i]]) d.qfrc_smooth[idx] = 0.0f;
}
