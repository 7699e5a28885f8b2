// collide.hpp -- broadphase + primitive narrowphase fused per world (one lane group per world).
//
// Reference: collision_driver.py:98-120 (plane/sphere bounding filters), 338-372 (_add_geom_pair), 684-770
// (_nxn_broadphase); collision_core.py:214-294 (write_contact), 297-414 (margin/gap, priority/solmix mixing);
// collision_primitive_core.py:48-530, 1044-1096 (analytic colliders); collision_primitive.py:280-836 (wrappers).
//
// MI355X mapping: the reference appends candidate pairs and contacts to *global* atomic queues (two kernels,
// one global atomic per pair and per contact).  Here a lane group filters its world's pre-filtered geom pairs,
// compacts the survivors in pair order with ballot/popcount, runs the colliders lane-parallel over survivors,
// compacts the resulting contacts (again ordered), and reserves a contiguous block of the public contact
// arrays with ONE atomic per world.  Contact order inside a world is deterministic (pair order, then the
// collider's own contact index); only the order of the per-world blocks depends on scheduling.
#pragma once
#include "dev_common.hpp"

#define MJH_MAXCON_PER_PAIR 8

struct ConGeom {
  float dist;
  V3 pos;
  float frame[9];
};

DEV void plane_sphere(V3 n, V3 ppos, V3 spos, float r, float& dist, V3& pos) {  // core:48
  dist = dot(spos - ppos, n) - r;
  pos = spos - n * (r + 0.5f * dist);
}
DEV void sphere_sphere(V3 p1, float r1, V3 p2, float r2, float& dist, V3& pos, V3& n) {  // core:56
  V3 dir = p2 - p1;
  float dd = length(dir);
  n = dd == 0.0f ? V3{1, 0, 0} : dir * (1.0f / dd);
  dist = dd - (r1 + r2);
  pos = p1 + n * (r1 + 0.5f * dist);
}
DEV V3 closest_segment_point(V3 a, V3 b, V3 pt) {  // math.py:270
  V3 ab = b - a;
  float t = dot(pt - a, ab) / (dot(ab, ab) + 1e-6f);
  return a + ab * clampf(t, 0.0f, 1.0f);
}

// runs the collider for geoms (g1,g2) with type1 <= type2; returns number of candidate contacts
DEV int collide_pair(int t1, int t2, V3 p1, const float* R1, V3 s1, V3 p2, const float* R2, V3 s2, float margin, ConGeom* out) {
  V3 ax1 = V3{R1[2], R1[5], R1[8]}, ax2 = V3{R2[2], R2[5], R2[8]};
  int n = 0;
  if (t1 == G_PLANE && t2 == G_SPHERE) {
    plane_sphere(ax1, p1, p2, s2.x, out[0].dist, out[0].pos);
    make_frame(ax1, out[0].frame);
    n = 1;
  } else if (t1 == G_PLANE && t2 == G_CAPSULE) {  // core:253
    float bn;
    V3 b = normalize_with_norm(ax2 - ax1 * dot(ax1, ax2), bn);
    if (bn < 0.5f) b = (-0.5f < ax1.y && ax1.y < 0.5f) ? V3{0, 1, 0} : V3{0, 0, 1};
    V3 c = cross(ax1, b);
    V3 seg = ax2 * s2.y;
    for (int k = 0; k < 2; ++k) {
      plane_sphere(ax1, p1, k == 0 ? p2 + seg : p2 - seg, s2.x, out[k].dist, out[k].pos);
      st3(out[k].frame, ax1);
      st3(out[k].frame + 3, b);
      st3(out[k].frame + 6, c);
    }
    n = 2;
  } else if (t1 == G_PLANE && t2 == G_BOX) {  // core:337
    float cd = dot(p2 - p1, ax1);
    float fr[9];
    make_frame(ax1, fr);
    for (int i = 0; i < 8; ++i) {
      V3 corner = V3{(i & 1) ? s2.x : -s2.x, (i & 2) ? s2.y : -s2.y, (i & 4) ? s2.z : -s2.z};
      V3 cw = mat_mul(R2, corner);
      float cdist = cd + dot(ax1, cw);
      out[i].dist = cdist;
      out[i].pos = cw + p2 - ax1 * (0.5f * cdist);
      for (int k = 0; k < 9; ++k) out[i].frame[k] = fr[k];
    }
    n = 8;
  } else if (t1 == G_PLANE && t2 == G_ELLIPSOID) {  // core:306
    V3 loc = matT_mul(R2, ax1);
    V3 sup = normalize(V3{loc.x * s2.x, loc.y * s2.y, loc.z * s2.z});
    sup = V3{-sup.x * s2.x, -sup.y * s2.y, -sup.z * s2.z};
    V3 pw = mat_mul(R2, sup) + p2;
    float dist = dot(ax1, pw - p1);
    out[0].dist = dist;
    out[0].pos = pw - ax1 * (0.5f * dist);
    make_frame(ax1, out[0].frame);
    n = 1;
  } else if (t1 == G_PLANE && t2 == G_CYLINDER) {  // core:460
    V3 axis = ax2;
    const float r = s2.x, hh = s2.y;
    float prjaxis = dot(ax1, axis);
    if (prjaxis > 0.0f) {
      axis = -axis;
      prjaxis = -prjaxis;
    }
    const float dist0 = dot(p2 - p1, ax1);
    V3 vec = axis * prjaxis - ax1;
    const float len_sqr = dot(vec, vec);
    vec = len_sqr >= 1e-12f ? vec * safe_div(r, sqrtf(len_sqr)) : V3{r, 0, 0};
    const float prjvec = dot(vec, ax1);
    axis = axis * hh;
    prjaxis *= hh;
    const float d1 = dist0 + prjaxis + prjvec, d2 = dist0 - prjaxis + prjvec;
    out[0].dist = d1;
    out[0].pos = p2 + vec + axis - ax1 * (d1 * 0.5f);
    out[1].dist = d2;
    out[1].pos = p2 + vec - axis - ax1 * (d2 * 0.5f);
    const float d3 = dist0 + prjaxis - 0.5f * prjvec;
    V3 vec1 = normalize(cross(vec, axis)) * (r * sqrtf(3.0f) * 0.5f);
    out[2].dist = d3;
    out[2].pos = p2 + vec1 + axis - vec * 0.5f - ax1 * (d3 * 0.5f);
    out[3].dist = d3;
    out[3].pos = p2 - vec1 + axis - vec * 0.5f - ax1 * (d3 * 0.5f);
    float fr[9];
    make_frame(ax1, fr);
    for (int i = 0; i < 4; ++i)
      for (int k = 0; k < 9; ++k) out[i].frame[k] = fr[k];
    n = 4;
  } else if (t1 == G_SPHERE && t2 == G_SPHERE) {
    V3 nn;
    sphere_sphere(p1, s1.x, p2, s2.x, out[0].dist, out[0].pos, nn);
    make_frame(nn, out[0].frame);
    n = 1;
  } else if (t1 == G_SPHERE && t2 == G_CAPSULE) {  // core:88
    V3 seg = ax2 * s2.y, nn;
    V3 pt = closest_segment_point(p2 - seg, p2 + seg, p1);
    sphere_sphere(p1, s1.x, pt, s2.x, out[0].dist, out[0].pos, nn);
    make_frame(nn, out[0].frame);
    n = 1;
  } else if (t1 == G_CAPSULE && t2 == G_CAPSULE) {  // core:123
    V3 axis1 = ax1 * s1.y, axis2 = ax2 * s2.y, dif = p1 - p2, nn, pos;
    const float ma = dot(axis1, axis1), mb = -dot(axis1, axis2), mc = dot(axis2, axis2);
    const float u = -dot(axis1, dif), v = dot(axis2, dif);
    const float det = ma * mc - mb * mb;
    float dist;
    if (fabsf(det) >= MJ_MINVAL) {
      float x1 = (mc * u - mb * v) / det, x2 = (ma * v - mb * u) / det;
      if (x1 > 1.0f) {
        x1 = 1.0f;
        x2 = (v - mb) / mc;
      } else if (x1 < -1.0f) {
        x1 = -1.0f;
        x2 = (v + mb) / mc;
      }
      if (x2 > 1.0f) {
        x2 = 1.0f;
        x1 = clampf((u - mb) / ma, -1.0f, 1.0f);
      } else if (x2 < -1.0f) {
        x2 = -1.0f;
        x1 = clampf((u + mb) / ma, -1.0f, 1.0f);
      }
      sphere_sphere(p1 + axis1 * x1, s1.x, p2 + axis2 * x2, s2.x, dist, pos, nn);
      if (dist <= margin) {
        out[0].dist = dist;
        out[0].pos = pos;
        make_frame(nn, out[0].frame);
        n = 1;
      }
    } else {  // parallel axes: up to 2 contacts from the 4 endpoint tests
      for (int e = 0; e < 4 && n < 2; ++e) {
        V3 v1, v2;
        if (e == 0) {
          v1 = p1 + axis1;
          v2 = p2 + axis2 * clampf((v - mb) / mc, -1.0f, 1.0f);
        } else if (e == 1) {
          v1 = p1 - axis1;
          v2 = p2 + axis2 * clampf((v + mb) / mc, -1.0f, 1.0f);
        } else if (e == 2) {
          v2 = p2 + axis2;
          v1 = p1 + axis1 * clampf((u - mb) / ma, -1.0f, 1.0f);
        } else {
          v2 = p2 - axis2;
          v1 = p1 + axis1 * clampf((u + mb) / ma, -1.0f, 1.0f);
        }
        sphere_sphere(v1, s1.x, v2, s2.x, dist, pos, nn);
        if (dist <= margin) {
          out[n].dist = dist;
          out[n].pos = pos;
          make_frame(nn, out[n].frame);
          ++n;
        }
      }
    }
  } else if (t1 == G_SPHERE && t2 == G_BOX) {  // core:1044
    V3 center = matT_mul(R2, p1 - p2);
    V3 clamped = V3{fmaxf(-s2.x, fminf(s2.x, center.x)), fmaxf(-s2.y, fminf(s2.y, center.y)), fmaxf(-s2.z, fminf(s2.z, center.z))};
    float dist;
    V3 cdir = normalize_with_norm(clamped - center, dist);
    V3 pos, nn;
    if (dist <= MJ_MINVAL) {
      const float sz[3] = {s2.x, s2.y, s2.z}, ce[3] = {center.x, center.y, center.z};
      float closest = 2.0f * (s2.x + s2.y + s2.z);
      int kk = 0;
      for (int i = 0; i < 6; ++i) {
        float fd = fabsf(((i % 2) ? 1.0f : -1.0f) * sz[i / 2] - ce[i / 2]);
        if (closest > fd) {
          closest = fd;
          kk = i;
        }
      }
      float ne[3] = {0, 0, 0};
      ne[kk / 2] = (kk % 2) ? -1.0f : 1.0f;
      V3 nearest = V3{ne[0], ne[1], ne[2]};
      pos = center + nearest * ((s1.x - closest) * 0.5f);
      nn = mat_mul(R2, nearest);
      out[0].dist = -closest - s1.x;
    } else {
      pos = (clamped + center + cdir * s1.x) * 0.5f;
      nn = mat_mul(R2, cdir);
      out[0].dist = dist - s1.x;
    }
    out[0].pos = mat_mul(R2, pos) + p2;
    make_frame(nn, out[0].frame);
    n = 1;
  }
  return n;
}

struct PairParams {
  int condim;
  float friction[5], solref[2], solreffriction[2], solimp[5], margin, gap;
};

// collision_core.py:297-414
DEV PairParams contact_params(const MjhModel& m, int w, int g1, int g2) {
  PairParams p;
  const int ng = m.ngeom;
  const float* gm = bf(m.geom_margin, m.geom_margin_nb, w, ng);
  const float* gg = bf(m.geom_gap, m.geom_gap_nb, w, ng);
  const float* gsm = bf(m.geom_solmix, m.geom_solmix_nb, w, ng);
  const float* gfr = bf(m.geom_friction, m.geom_friction_nb, w, 3 * ng);
  const float* gsr = bf(m.geom_solref, m.geom_solref_nb, w, 2 * ng);
  const float* gsi = bf(m.geom_solimp, m.geom_solimp_nb, w, 5 * ng);
  p.margin = gm[g1] + gm[g2];
  p.gap = gg[g1] + gg[g2];
  const float s1 = gsm[g1], s2 = gsm[g2];
  const int p1 = m.geom_priority[g1], p2 = m.geom_priority[g2];
  float mix, f0, f1, f2;
  if (p1 > p2) {
    mix = 1.0f;
    p.condim = m.geom_condim[g1];
    f0 = gfr[3 * g1]; f1 = gfr[3 * g1 + 1]; f2 = gfr[3 * g1 + 2];
  } else if (p2 > p1) {
    mix = 0.0f;
    p.condim = m.geom_condim[g2];
    f0 = gfr[3 * g2]; f1 = gfr[3 * g2 + 1]; f2 = gfr[3 * g2 + 2];
  } else {
    mix = safe_div(s1, s1 + s2);
    if (s1 < MJ_MINVAL && s2 < MJ_MINVAL) mix = 0.5f;
    if (s1 < MJ_MINVAL && s2 >= MJ_MINVAL) mix = 0.0f;
    if (s1 >= MJ_MINVAL && s2 < MJ_MINVAL) mix = 1.0f;
    p.condim = max(m.geom_condim[g1], m.geom_condim[g2]);
    f0 = fmaxf(gfr[3 * g1], gfr[3 * g2]);
    f1 = fmaxf(gfr[3 * g1 + 1], gfr[3 * g2 + 1]);
    f2 = fmaxf(gfr[3 * g1 + 2], gfr[3 * g2 + 2]);
  }
  p.friction[0] = p.friction[1] = fmaxf(MJ_MINMU, f0);
  p.friction[2] = fmaxf(MJ_MINMU, f1);
  p.friction[3] = p.friction[4] = fmaxf(MJ_MINMU, f2);
  const float *r1 = gsr + 2 * g1, *r2 = gsr + 2 * g2;
  if (r1[0] > 0.0f && r2[0] > 0.0f) {
    p.solref[0] = mix * r1[0] + (1.0f - mix) * r2[0];
    p.solref[1] = mix * r1[1] + (1.0f - mix) * r2[1];
  } else {
    p.solref[0] = fminf(r1[0], r2[0]);
    p.solref[1] = fminf(r1[1], r2[1]);
  }
  p.solreffriction[0] = p.solreffriction[1] = 0.0f;
  for (int k = 0; k < 5; ++k) p.solimp[k] = mix * gsi[5 * g1 + k] + (1.0f - mix) * gsi[5 * g2 + k];
  return p;
}

// per-world LDS: candidate list (ints) | staged contacts: per contact 16 words (dist,pos3,frame9,pairidx,cid,pad)
#define CON_STAGE_WORDS 16
__host__ __device__ inline int collide_lds_words(int npair, int ncap) { return ((npair + 3) / 4) * 4 + ncap * CON_STAGE_WORDS + 1; }

template <int G>
__global__ void __launch_bounds__(256) k_collision(MjhModel m, MjhData d, int ncap) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lig = threadIdx.x & (G - 1), gib = threadIdx.x / G;
  const int w = blockIdx.x * (blockDim.x / G) + gib;
  if (w >= d.nworld) return;
  const int npair = m.npair, ng = m.ngeom;
  float* S = smem + (size_t)gib * collide_lds_words(npair, ncap);
  int* cand = reinterpret_cast<int*>(S);
  float* stage = S + ((npair + 3) / 4) * 4;

  if (m.disableflags & (DSBL_CONSTRAINT | DSBL_CONTACT)) {
    if (lig == 0) {
      d.ws_ncon[w] = 0;
      d.ws_conadr[w] = 0;
      d.ws_ncollision[w] = 0;
    }
    return;
  }
  const float* gxpos = d.geom_xpos + (size_t)w * 3 * ng;
  const float* gxmat = d.geom_xmat + (size_t)w * 9 * ng;
  const float* rbound = bf(m.geom_rbound, m.geom_rbound_nb, w, ng);
  const float* gmargin = bf(m.geom_margin, m.geom_margin_nb, w, ng);
  const float* ggap = bf(m.geom_gap, m.geom_gap_nb, w, ng);
  const float* gsize = bf(m.geom_size, m.geom_size_nb, w, 3 * ng);

  // ---- broadphase: plane / bounding-sphere filter, ordered compaction -----------------------------------
  int ncand = 0;
  for (int base = 0; base < npair; base += G) {
    const int p = base + lig;
    bool pass = false;
    if (p < npair) {
      const int g1 = m.nxn_geom_pair[2 * p], g2 = m.nxn_geom_pair[2 * p + 1];
      const float rb1 = rbound[g1], rb2 = rbound[g2];
      const float mg = gmargin[g1] + ggap[g1] + gmargin[g2] + ggap[g2];
      V3 x1 = ld3(gxpos + 3 * g1), x2 = ld3(gxpos + 3 * g2);
      if (rb1 == 0.0f || rb2 == 0.0f) {
        if (rb1 == 0.0f) {
          const float* R = gxmat + 9 * g1;
          pass = dot(x2 - x1, V3{R[2], R[5], R[8]}) <= rb2 + mg;
        } else {
          const float* R = gxmat + 9 * g2;
          pass = dot(x1 - x2, V3{R[2], R[5], R[8]}) <= rb1 + mg;
        }
      } else {
        const float bound = rb1 + rb2 + mg;
        V3 dif = x2 - x1;
        pass = dot(dif, dif) <= bound * bound;
      }
    }
    int tot;
    const int rank = grank<G>(pass, lig, tot);
    if (pass) cand[ncand + rank] = p;
    ncand += tot;
  }
  gsync();

  // ---- narrowphase over candidates, ordered compaction of detected contacts ------------------------------
  int ncon = 0;
  for (int base = 0; base < ncand; base += G) {
    const int ci = base + lig;
    ConGeom out[MJH_MAXCON_PER_PAIR];
    unsigned keep = 0;
    int p = -1, nk = 0;
    if (ci < ncand) {
      p = cand[ci];
      int g1 = m.nxn_geom_pair[2 * p], g2 = m.nxn_geom_pair[2 * p + 1];
      int t1 = m.geom_type[g1], t2 = m.geom_type[g2];
      if (t1 > t2) {
        int t = g1; g1 = g2; g2 = t;
        t = t1; t1 = t2; t2 = t;
      }
      const float margin = gmargin[g1] + gmargin[g2], gap = ggap[g1] + ggap[g2];
      const int n = collide_pair(t1, t2, ld3(gxpos + 3 * g1), gxmat + 9 * g1, ld3(gsize + 3 * g1), ld3(gxpos + 3 * g2),
                                 gxmat + 9 * g2, ld3(gsize + 3 * g2), margin, out);
      for (int k = 0; k < n; ++k)
        if (out[k].dist < margin + gap) {
          keep |= 1u << k;
          ++nk;
        }
    }
    // exclusive prefix of nk over the group (ordered by candidate index)
    int incl = nk;
    for (int off = 1; off < G; off <<= 1) {
      int v = __shfl_up(incl, off, G);
      if (lig >= off) incl += v;
    }
    const int excl = incl - nk;
    const int tot = __shfl(incl, G - 1, G);
    int slot = ncon + excl;
    for (int k = 0; k < MJH_MAXCON_PER_PAIR; ++k) {
      if (!(keep & (1u << k))) continue;
      if (slot < ncap) {
        float* s = stage + slot * CON_STAGE_WORDS;
        s[0] = out[k].dist;
        st3(s + 1, out[k].pos);
        for (int q = 0; q < 9; ++q) s[4 + q] = out[k].frame[q];
        reinterpret_cast<int*>(s)[13] = p;
        reinterpret_cast<int*>(s)[14] = k;
      }
      ++slot;
    }
    ncon += tot;
  }
  gsync();

  // ---- reserve a block of the public contact arrays (one atomic per world) and publish ------------------
  const int nfound = ncon;
  if (ncon > ncap) ncon = ncap;
  int adr = 0;
  if (lig == 0) {
    adr = atomicAdd(d.nacon, ncon);
    atomicAdd(d.ncollision, ncand);
  }
  adr = __shfl(adr, 0, G);
  int nwrite = ncon;
  if (adr + nwrite > d.naconmax) nwrite = max(0, d.naconmax - adr);
  if (lig == 0) {
    d.ws_ncon[w] = nwrite;
    d.ws_conadr[w] = adr;
    d.ws_ncollision[w] = ncand;
    if (nwrite < nfound) atomicOr(d.overflow + w, OVF_NARROWPHASE);
  }
  for (int c = lig; c < nwrite; c += G) {
    const float* s = stage + c * CON_STAGE_WORDS;
    const int p = reinterpret_cast<const int*>(s)[13], cid = reinterpret_cast<const int*>(s)[14];
    int g1 = m.nxn_geom_pair[2 * p], g2 = m.nxn_geom_pair[2 * p + 1];
    if (m.geom_type[g1] > m.geom_type[g2]) {
      int t = g1; g1 = g2; g2 = t;
    }
    const PairParams pp = contact_params(m, w, g1, g2);
    const size_t o = (size_t)(adr + c);
    d.contact_dist[o] = s[0];
    for (int q = 0; q < 3; ++q) d.contact_pos[3 * o + q] = s[1 + q];
    for (int q = 0; q < 9; ++q) d.contact_frame[9 * o + q] = s[4 + q];
    d.contact_includemargin[o] = pp.margin;
    for (int q = 0; q < 5; ++q) d.contact_friction[5 * o + q] = pp.friction[q];
    for (int q = 0; q < 2; ++q) d.contact_solref[2 * o + q] = pp.solref[q];
    for (int q = 0; q < 2; ++q) d.contact_solreffriction[2 * o + q] = pp.solreffriction[q];
    for (int q = 0; q < 5; ++q) d.contact_solimp[5 * o + q] = pp.solimp[q];
    d.contact_dim[o] = pp.condim;
    d.contact_geom[2 * o] = g1;
    d.contact_geom[2 * o + 1] = g2;
    d.contact_worldid[o] = w;
    d.contact_type[o] = CONTACT_TYPE_CONSTRAINT;
    d.contact_geomcollisionid[o] = cid;
    for (int q = 0; q < d.nmaxpyramid; ++q) d.contact_efc_address[o * d.nmaxpyramid + q] = -1;
  }
}
