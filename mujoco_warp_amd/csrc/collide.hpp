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

// A collider reports each candidate contact through `emit(index, dist, pos, frame_row0, frame_row1, frame_row2)`.
// The kernel runs every collider twice with different emitters (count, then write): nothing is buffered per
// lane, so the kernel needs no scratch memory and no dynamically indexed register arrays.
struct Frame {
  V3 a, b, c;
};
DEV Frame make_frame3(V3 n) {  // math.py:203-257
  float f[9];
  make_frame(n, f);
  return Frame{ld3(f), ld3(f + 3), ld3(f + 6)};
}

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

// sphere_box core:1044 -- contact of a sphere (centre sp, radius r) with a box; results in world coordinates
DEV void sphere_box(V3 sp, float r, V3 bp, const float* R, V3 bs, float& dist, V3& wpos, V3& nn) {
  V3 center = matT_mul(R, sp - bp);
  V3 clamped = V3{fmaxf(-bs.x, fminf(bs.x, center.x)), fmaxf(-bs.y, fminf(bs.y, center.y)), fmaxf(-bs.z, fminf(bs.z, center.z))};
  V3 pos;
  V3 cdir = normalize_with_norm(clamped - center, dist);
  if (dist <= MJ_MINVAL) {  // centre inside the box: push out through the nearest face
    const float sz[3] = {bs.x, bs.y, bs.z}, ce[3] = {center.x, center.y, center.z};
    float closest = 2.0f * (bs.x + bs.y + bs.z);
    int kk = 0;
    for (int i = 0; i < 6; ++i) {
      float fd = fabsf(((i % 2) ? 1.0f : -1.0f) * sz[i / 2] - ce[i / 2]);
      if (closest > fd) {
        closest = fd;
        kk = i;
      }
    }
    const float sg = (kk % 2) ? -1.0f : 1.0f;
    V3 nearest = V3{kk / 2 == 0 ? sg : 0.0f, kk / 2 == 1 ? sg : 0.0f, kk / 2 == 2 ? sg : 0.0f};
    pos = center + nearest * ((r - closest) * 0.5f);
    nn = mat_mul(R, nearest);
    dist = -closest - r;
  } else {
    pos = (clamped + center + cdir * r) * 0.5f;
    nn = mat_mul(R, cdir);
    dist = dist - r;
  }
  wpos = mat_mul(R, pos) + bp;
}

DEV float pick3(V3 v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }

// capsule_box core:1099 (MuJoCo's mjc_CapsuleBox).  Returns the segment parameters (point = pos + t * halfaxis, in [-1, 1])
// of the contact spheres: t1 always (unless the configuration is degenerate: returns 0), t2 when the capsule lies along a
// face or an edge of the box.  Same decision tree as oracle/mjref.c:capsule_box.
DEV int capsule_box_params(V3 pos, V3 axis, float hl, V3 bs, float& t1, float& t2) {
  const V3 ha = axis * hl;
  const float p[3] = {pos.x, pos.y, pos.z}, h[3] = {ha.x, ha.y, ha.z}, s[3] = {bs.x, bs.y, bs.z};
  const int axisdir = (ha.x > 0.0f ? 1 : 0) + (ha.y > 0.0f ? 2 : 0) + (ha.z > 0.0f ? 4 : 0);
  float bestdist = 1e32f, bestseg = -12.0f, bestbox = 0.0f, second = -4.0f;
  int cltype = -4, clface = -12, clcorner = -123, cledge = -123;
  // (1) a capsule end over a face (or inside): clamped in at most one coordinate
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const float sgn = e == 0 ? -1.0f : 1.0f;
    float d2 = 0.0f;
    int nout = 0, axout = -1;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float tip = p[k] + sgn * h[k];
      float c = tip;
      if (c < -s[k]) { ++nout; axout = k; c = -s[k]; }
      else if (c > s[k]) { ++nout; axout = k; c = s[k]; }
      d2 += (c - tip) * (c - tip);
    }
    if (nout <= 1 && d2 < bestdist) {
      bestdist = d2;
      bestseg = sgn;
      cltype = e == 0 ? -3 : -1;
      clface = axout;
    }
  }
  // (2) the segment against each of the 12 box edges (corner i, edge direction j with bit j of i clear)
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (i & (1 << j)) continue;
      float dif[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) dif[k] = (k == j ? 0.0f : (((i >> k) & 1) ? s[k] : -s[k])) - p[k];
      const float u = -s[j] * dif[j], v = h[0] * dif[0] + h[1] * dif[1] + h[2] * dif[2];
      const float ma = s[j] * s[j], mb = -s[j] * h[j], mc = hl * hl;
      const float det = ma * mc - mb * mb;
      if (fabsf(det) < MJ_MINVAL) continue;
      float x1 = (mc * u - mb * v) / det, x2 = (ma * v - mb * u) / det;  // x1 along the edge, x2 along the capsule
      int s1 = 1, s2 = 1;                                                // 1: interior, 0 / 2: lower / upper end
      if (x1 > 1.0f) { x1 = 1.0f; s1 = 2; x2 = safe_div(v - mb, mc); }
      else if (x1 < -1.0f) { x1 = -1.0f; s1 = 0; x2 = safe_div(v + mb, mc); }
      if (x2 > 1.0f || x2 < -1.0f) {
        if (x2 > 1.0f) { x2 = 1.0f; s2 = 2; x1 = safe_div(u - mb, ma); }
        else { x2 = -1.0f; s2 = 0; x1 = safe_div(u + mb, ma); }
        if (x1 > 1.0f) { x1 = 1.0f; s1 = 2; }
        else if (x1 < -1.0f) { x1 = -1.0f; s1 = 0; }
      }
      float d2 = 0.0f;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float dk = dif[k] - h[k] * x2 + (k == j ? s[j] * x1 : 0.0f);
        d2 += dk * dk;
      }
      if (d2 < bestdist - MJ_MINVAL) {
        const int ct = s1 * 3 + s2;
        bestdist = d2;
        bestseg = x2;
        bestbox = x1;
        clcorner = i + (1 << j) * (ct / 6);  // the box corner nearest to the closest point
        cledge = j;
        cltype = ct;
      }
    }
  }
  if (cltype == -4) return 0;
  // (3) a second contact further along the capsule
  if (cltype >= 0 && cltype / 3 != 1) {  // closest box point is a corner
    int c1 = axisdir ^ clcorner;
    if (c1 != 0 && c1 != 7) {  // the capsule does not point straight at / away from the corner
      float mul = 1.0f;
      if (!(c1 == 1 || c1 == 2 || c1 == 4)) {
        mul = -1.0f;
        c1 = 7 - c1;
      }
      const int ax = c1 == 1 ? 0 : (c1 == 2 ? 1 : 2), ax1 = (ax + 1) % 3, ax2 = (ax + 2) % 3;
      const float aa = pick3(axis, ax);
      if (aa * aa > 0.5f) {  // along the edge
        second = fminf(1.0f - mul * bestseg, 2.0f * safe_div(pick3(bs, ax), fabsf(pick3(ha, ax))));
      } else {  // across a face
        const float mlim = 2.0f * fminf(safe_div(pick3(bs, ax1), fabsf(pick3(ha, ax1))), safe_div(pick3(bs, ax2), fabsf(pick3(ha, ax2))));
        second = -fminf(1.0f + mul * bestseg, mlim);
      }
      second *= mul;
    }
  } else if (cltype >= 0) {  // closest box point is inside an edge
    const int c1 = (axisdir ^ clcorner) & (7 - (1 << cledge));
    if (c1 == 1 || c1 == 2 || c1 == 4) {  // X configuration (a T configuration has no second contact)
      const int ax = cledge;
      int ax1 = (ax + 1) % 3, ax2 = (ax + 2) % 3;
      if (fabsf(pick3(axis, ax1)) > fabsf(pick3(axis, ax2))) ax1 = ax2;  // the face the capsule makes the smaller angle with
      ax2 = 3 - ax - ax1;
      float mul;
      if (c1 & (1 << ax2)) { mul = 1.0f; second = 1.0f - bestseg; }
      else { mul = -1.0f; second = 1.0f + bestseg; }
      second = fminf(2.0f * safe_div(pick3(bs, ax2), fabsf(pick3(ha, ax2))), second);
      const float e2 = (((axisdir & (1 << ax)) != 0) == ((c1 & (1 << ax2)) != 0)) ? 1.0f - bestbox : 1.0f + bestbox;
      second = fminf(pick3(bs, ax) * safe_div(e2, fabsf(pick3(ha, ax))), second);
      second *= mul;
    }
  } else if (clface != -1) {  // an end is closest to a face: follow the capsule until it leaves the box footprint
    const float mul = cltype == -3 ? 1.0f : -1.0f;
    second = 2.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (k == clface) continue;
      const float tmp = p[k] - h[k] * mul, har = safe_div(mul, h[k]);
      float e1 = (s[k] - tmp) * har;
      if (0.0f < e1 && e1 < second) second = e1;
      e1 = (-s[k] - tmp) * har;
      if (0.0f < e1 && e1 < second) second = e1;
    }
    second *= mul;
  }
  t1 = bestseg;
  t2 = second + bestseg;
  return second > -3.0f ? 2 : 1;
}

// runs the collider for geoms (g1,g2) with type1 <= type2
template <bool HEAVY, class Emit>
DEV void collide_pair(int t1, int t2, V3 p1, const float* R1, V3 s1, V3 p2, const float* R2, V3 s2, float margin, Emit&& emit) {
  V3 ax1 = V3{R1[2], R1[5], R1[8]}, ax2 = V3{R2[2], R2[5], R2[8]};
  float dist;
  V3 pos, nn;
  if (t1 == G_PLANE && t2 == G_SPHERE) {
    plane_sphere(ax1, p1, p2, s2.x, dist, pos);
    const Frame f = make_frame3(ax1);
    emit(0, dist, pos, f.a, f.b, f.c);
  } else if (t1 == G_PLANE && t2 == G_CAPSULE) {  // core:253
    float bn;
    V3 b = normalize_with_norm(ax2 - ax1 * dot(ax1, ax2), bn);
    if (bn < 0.5f) b = (-0.5f < ax1.y && ax1.y < 0.5f) ? V3{0, 1, 0} : V3{0, 0, 1};
    V3 c = cross(ax1, b);
    V3 seg = ax2 * s2.y;
    plane_sphere(ax1, p1, p2 + seg, s2.x, dist, pos);
    emit(0, dist, pos, ax1, b, c);
    plane_sphere(ax1, p1, p2 - seg, s2.x, dist, pos);
    emit(1, dist, pos, ax1, b, c);
  } else if (t1 == G_PLANE && t2 == G_BOX) {  // core:337
    const float cd = dot(p2 - p1, ax1);
    const Frame f = make_frame3(ax1);
    for (int i = 0; i < 8; ++i) {
      V3 corner = V3{(i & 1) ? s2.x : -s2.x, (i & 2) ? s2.y : -s2.y, (i & 4) ? s2.z : -s2.z};
      V3 cw = mat_mul(R2, corner);
      const float cdist = cd + dot(ax1, cw);
      emit(i, cdist, cw + p2 - ax1 * (0.5f * cdist), f.a, f.b, f.c);
    }
  } else if (t1 == G_PLANE && t2 == G_ELLIPSOID) {  // core:306
    V3 loc = matT_mul(R2, ax1);
    V3 sup = normalize(V3{loc.x * s2.x, loc.y * s2.y, loc.z * s2.z});
    sup = V3{-sup.x * s2.x, -sup.y * s2.y, -sup.z * s2.z};
    V3 pw = mat_mul(R2, sup) + p2;
    dist = dot(ax1, pw - p1);
    const Frame f = make_frame3(ax1);
    emit(0, dist, pw - ax1 * (0.5f * dist), f.a, f.b, f.c);
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
    const Frame f = make_frame3(ax1);
    emit(0, d1, p2 + vec + axis - ax1 * (d1 * 0.5f), f.a, f.b, f.c);
    emit(1, d2, p2 + vec - axis - ax1 * (d2 * 0.5f), f.a, f.b, f.c);
    const float d3 = dist0 + prjaxis - 0.5f * prjvec;
    V3 vec1 = normalize(cross(vec, axis)) * (r * sqrtf(3.0f) * 0.5f);
    emit(2, d3, p2 + vec1 + axis - vec * 0.5f - ax1 * (d3 * 0.5f), f.a, f.b, f.c);
    emit(3, d3, p2 - vec1 + axis - vec * 0.5f - ax1 * (d3 * 0.5f), f.a, f.b, f.c);
  } else if (t1 == G_SPHERE && t2 == G_SPHERE) {
    sphere_sphere(p1, s1.x, p2, s2.x, dist, pos, nn);
    const Frame f = make_frame3(nn);
    emit(0, dist, pos, f.a, f.b, f.c);
  } else if (t1 == G_SPHERE && t2 == G_CAPSULE) {  // core:88
    V3 seg = ax2 * s2.y;
    V3 pt = closest_segment_point(p2 - seg, p2 + seg, p1);
    sphere_sphere(p1, s1.x, pt, s2.x, dist, pos, nn);
    const Frame f = make_frame3(nn);
    emit(0, dist, pos, f.a, f.b, f.c);
  } else if (t1 == G_CAPSULE && t2 == G_CAPSULE) {  // core:123
    V3 axis1 = ax1 * s1.y, axis2 = ax2 * s2.y, dif = p1 - p2;
    const float ma = dot(axis1, axis1), mb = -dot(axis1, axis2), mc = dot(axis2, axis2);
    const float u = -dot(axis1, dif), v = dot(axis2, dif);
    const float det = ma * mc - mb * mb;
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
        const Frame f = make_frame3(nn);
        emit(0, dist, pos, f.a, f.b, f.c);
      }
    } else {  // parallel axes: up to 2 contacts from the 4 endpoint tests
      int n = 0;
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
          const Frame f = make_frame3(nn);
          emit(n, dist, pos, f.a, f.b, f.c);
          ++n;
        }
      }
    }
  } else if (t1 == G_SPHERE && t2 == G_BOX) {  // core:1044
    sphere_box(p1, s1.x, p2, R2, s2, dist, pos, nn);
    const Frame f = make_frame3(nn);
    emit(0, dist, pos, f.a, f.b, f.c);
  } else if (HEAVY && t1 == G_CAPSULE && t2 == G_BOX) {  // core:1099
    const V3 lp = matT_mul(R2, p1 - p2), la = matT_mul(R2, ax1);
    float tt[2];
    const int n = capsule_box_params(lp, la, s1.y, s2, tt[0], tt[1]);
    for (int c = 0; c < n; ++c) {
      sphere_box(mat_mul(R2, lp + la * (s1.y * tt[c])) + p2, s1.x, p2, R2, s2, dist, pos, nn);
      const Frame f = make_frame3(nn);
      emit(c, dist, pos, f.a, f.b, f.c);
    }
  } else if (t1 == G_SPHERE && t2 == G_CYLINDER) {  // core:388
    const float r = s2.x, hh = s2.y;
    const V3 vec = p1 - p2;
    const float x = dot(vec, ax2);
    const V3 aproj = ax2 * x, pproj = vec - aproj;
    const float psq = dot(pproj, pproj);
    bool side = fabsf(x) < hh, cap = psq < r * r;
    if (side && cap) {  // centre inside the cylinder: the nearer surface wins
      if (hh - fabsf(x) < r - sqrtf(psq)) side = false;
      else cap = false;
    }
    if (side) {
      sphere_sphere(p1, s1.x, p2 + aproj, r, dist, pos, nn);
    } else if (cap) {
      const V3 pn = ax2 * (x > 0.0f ? 1.0f : -1.0f);
      plane_sphere(pn, p2 + pn * hh, p1, s1.x, dist, pos);
      nn = pn * -1.0f;
    } else {  // rim
      const float sg = x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f);
      sphere_sphere(p1, s1.x, p2 + ax2 * (sg * hh) + pproj * (r * safe_div(1.0f, sqrtf(psq))), 0.0f, dist, pos, nn);
    }
    const Frame f = make_frame3(nn);
    emit(0, dist, pos, f.a, f.b, f.c);
  }
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

// Contact record (CON_REC words; CON_STRIDE in the per-world hand-off buffer d.ws_contact, so that a record is one
// aligned 128-byte line):
//   0 dist | 1-3 pos | 4-12 frame | 13 includemargin | 14-16 friction (slide, spin, roll) | 17-18 solref |
//   19-23 solimp | 24 condim | 25-26 geoms | 27 collider contact id | 28 first efc row or -1 | 29 number of rows
// (28-29 are filled by k_make_constraint).  k_collision hands the contacts of a world to k_make_constraint through
// d.ws_contact[w]; the public, compact contact_* arrays are produced from the same records by publish_body
// (appended to the integrator launch or run as k_publish_contacts).  The reference reserves public slots with one global atomic per
// contact (collision_core.py write_contact); on MI355X one same-address device atomic per WORLD already cost 25-50 us
// per launch (they resolve at the memory side, ~6 ns each, and every later load of the wave waits behind them).
#define CON_WINDOW 16
#define CON_REC 30
#define CON_LDS 31  /* odd LDS stride: lane-per-contact reads are bank-conflict free */
#define CON_STRIDE 32
// per-world LDS: geom poses (12 words per geom) | candidate pair list | first contact slot << 8 | contact mask per
// candidate | staging window of CON_WINDOW records
__host__ __device__ inline int collide_lds_words(int ngeom, int npair) {
  return 12 * ngeom + ((npair + 3) / 4) * 4 + ((npair + 3) / 4) * 4 + CON_WINDOW * CON_LDS;
}

// HEAVY: the instantiation that also carries the large colliders (capsule-box); models without such pairs run the light one,
// whose register footprint (and with it the occupancy of k_mid) is unchanged
template <int G, bool HEAVY = false>
DEV void collision_body(const MjhModel& m, const MjhData& d, float* smem, const Blk& b, int stride_words = 0) {
  const int lig = threadIdx.x & (G - 1), gib = threadIdx.x / G;
  const int w = b.w0 + gib;
  if ((int)threadIdx.x >= b.nthreads || w >= d.nworld) return;
  const int npair = m.npair, ng = m.ngeom, ncap = d.concap;
  float* S = smem + (size_t)gib * (stride_words ? stride_words : collide_lds_words(ng, npair));
  float* gxpos = S;
  float* gxmat = S + 3 * ng;
  int* cand = reinterpret_cast<int*>(S + 12 * ng);
  int* cslot = cand + ((npair + 3) / 4) * 4;
  float* rec = reinterpret_cast<float*>(cslot + ((npair + 3) / 4) * 4);

  if (m.disableflags & (DSBL_CONSTRAINT | DSBL_CONTACT)) {
    if (lig == 0) {
      d.ws_ncon[w] = 0;
      d.ws_ncollision[w] = 0;
    }
    return;
  }
  PhaseClock pc(2, lig);
  gcopy<G>(gxpos, d.geom_xpos + (size_t)w * 3 * ng, 3 * ng, lig);
  gcopy<G>(gxmat, d.geom_xmat + (size_t)w * 9 * ng, 9 * ng, lig);
  const float* rbound = bf(m.geom_rbound, m.geom_rbound_nb, w, ng);
  const float* gmargin = bf(m.geom_margin, m.geom_margin_nb, w, ng);
  const float* ggap = bf(m.geom_gap, m.geom_gap_nb, w, ng);
  const float* gsize = bf(m.geom_size, m.geom_size_nb, w, 3 * ng);
  gsync();
  pc.mark(0);

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
  pc.mark(1);

  // ---- narrowphase, pass 1: contacts per candidate and their exclusive prefix (contacts stay in pair order) ----
  auto load_pair = [&](int p, int& g1, int& g2, int& t1, int& t2) {
    g1 = m.nxn_geom_pair[2 * p];
    g2 = m.nxn_geom_pair[2 * p + 1];
    t1 = m.geom_type[g1];
    t2 = m.geom_type[g2];
    if (t1 > t2) {
      int t = g1; g1 = g2; g2 = t;
      t = t1; t1 = t2; t2 = t;
    }
  };
  int ncon = 0;
  for (int base = 0; base < ncand; base += G) {
    const int ci = base + lig;
    // which of the collider's (at most 8) contacts pass the margin test: pass 2 replays this mask instead of
    // re-testing, so the two passes agree even if the compiler contracts the distance arithmetic differently
    unsigned mask = 0u;
    if (ci < ncand) {
      int g1, g2, t1, t2;
      load_pair(cand[ci], g1, g2, t1, t2);
      const float margin = gmargin[g1] + gmargin[g2];
      const float lim = margin + ggap[g1] + ggap[g2];
      collide_pair<HEAVY>(t1, t2, ld3(gxpos + 3 * g1), gxmat + 9 * g1, ld3(gsize + 3 * g1), ld3(gxpos + 3 * g2), gxmat + 9 * g2,
                   ld3(gsize + 3 * g2), margin, [&](int k, float dist, V3, V3, V3, V3) { mask |= dist < lim ? (1u << (k & 7)) : 0u; });
    }
    const int nk = __popc(mask);
    int incl = nk;
    for (int off = 1; off < G; off <<= 1) {
      int v = __shfl_up(incl, off, G);
      if (lig >= off) incl += v;
    }
    if (ci < ncand) cslot[ci] = ((ncon + incl - nk) << 8) | (int)mask;
    ncon += __shfl(incl, G - 1, G);
  }
  gsync();
  pc.mark(2);

  // ---- pass 2, CON_WINDOW contacts at a time: contacting candidates recompute their contacts into the LDS window,
  // then the group copies the window to the world's slice of d.ws_contact with consecutive addresses -------------
  const int nfound = ncon;
  if (ncon > ncap) ncon = ncap;
  float* out = d.ws_contact + (size_t)w * ncap * CON_STRIDE;
  for (int wbase = 0; wbase < ncon; wbase += CON_WINDOW) {
    const int wend = min(wbase + CON_WINDOW, ncon);
    // a slot that pass 2 fails to reproduce (see the mask above) becomes a harmless inactive contact
    for (int idx = lig; idx < CON_WINDOW * CON_LDS; idx += G) {
      const int f = idx % CON_LDS;
      rec[idx] = f == 0 ? 1e10f : (f == 24 ? __int_as_float(1) : (f == 28 ? __int_as_float(-1) : 0.0f));
    }
    gsync();
    pc.mark(3);
    for (int base = 0; base < ncand; base += G) {
      const int ci = base + lig;
      if (ci >= ncand) continue;
      const unsigned mask = (unsigned)cslot[ci] & 0xffu;
      int slot = cslot[ci] >> 8;
      const int send = slot + __popc(mask);
      if (mask == 0u || slot >= wend || send <= wbase) continue;
      int g1, g2, t1, t2;
      load_pair(cand[ci], g1, g2, t1, t2);
      const float margin = gmargin[g1] + gmargin[g2];
      const PairParams pp = contact_params(m, w, g1, g2);
      collide_pair<HEAVY>(t1, t2, ld3(gxpos + 3 * g1), gxmat + 9 * g1, ld3(gsize + 3 * g1), ld3(gxpos + 3 * g2), gxmat + 9 * g2,
                   ld3(gsize + 3 * g2), margin, [&](int cid, float dist, V3 pos, V3 fa, V3 fb, V3 fc) {
                     if (!((mask >> (cid & 7)) & 1u)) return;
                     if (slot >= wbase && slot < wend) {
                       float* r = rec + (slot - wbase) * CON_LDS;
                       r[0] = dist;
                       st3(r + 1, pos);
                       st3(r + 4, fa);
                       st3(r + 7, fb);
                       st3(r + 10, fc);
                       r[13] = pp.margin;
                       r[14] = pp.friction[0];
                       r[15] = pp.friction[2];
                       r[16] = pp.friction[3];
                       r[17] = pp.solref[0];
                       r[18] = pp.solref[1];
                       for (int q = 0; q < 5; ++q) r[19 + q] = pp.solimp[q];
                       int* ri = reinterpret_cast<int*>(r);
                       ri[24] = pp.condim;
                       ri[25] = g1;
                       ri[26] = g2;
                       ri[27] = cid;
                     }
                     ++slot;
                   });
    }
    gsync();
    pc.mark(4);
    const int nw = (wend - wbase) * CON_STRIDE;
    for (int idx = lig; idx < nw; idx += G) {
      const int f = idx & (CON_STRIDE - 1);
      out[(size_t)wbase * CON_STRIDE + idx] = f < CON_LDS ? rec[(idx / CON_STRIDE) * CON_LDS + f] : 0.0f;
    }
    gsync();
    pc.mark(5);
  }
  if (lig == 0) {
    d.ws_ncon[w] = ncon;
    d.ws_ncollision[w] = ncand;
    if (ncon < nfound) atomicOr(d.overflow + w, OVF_NARROWPHASE);
  }
}

template <int G, bool HEAVY>
__global__ void __launch_bounds__(256) k_collision(MjhModel m, MjhData d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  collision_body<G, HEAVY>(m, d, smem, blk_of_launch<G>());
}

// ---- publication of the compact public contact arrays (off the critical path) ---------------------------------
// publish_body: one group per world copies its records to the public SoA arrays (consecutive addresses per array),
// fills contact.efc_address and the contact rows of efc.id.  Worlds are published in world order, so the public
// arrays are deterministic (the reference's order depends on atomic arrival).
// self_prefix: the workgroup first sums ws_ncon over all earlier worlds itself (the counts are L2-resident: 32 KB at
// 8192 worlds), so that no scan kernel has to run before it; the workgroup of the last world also writes the totals.
// `sh` needs 64 ints of LDS.
template <int G>
DEV void publish_body(const MjhData& d, int self_prefix, int* sh, const Blk& b) {
  if ((int)threadIdx.x >= b.nthreads) return;
  const int lig = threadIdx.x & (G - 1), gib = threadIdx.x / G;
  const int w = b.w0 + gib;
  int adr_self = 0;
  if (self_prefix) {
    const int t = threadIdx.x, nwave = (b.nthreads + 63) / 64;
    const bool last = b.w0 + b.nw >= d.nworld;
    int s = 0, s2 = 0;
    {  // 16-byte loads, all in flight before the first add (a dependent scalar loop costs ~0.5 us per trip)
      const int4* p4 = reinterpret_cast<const int4*>(d.ws_ncon);
      const int n4 = b.w0 >> 2;
#pragma unroll 8
      for (int i = t; i < n4; i += b.nthreads) {
        const int4 v = p4[i];
        s += v.x + v.y + v.z + v.w;
      }
      for (int i = (n4 << 2) + t; i < b.w0; i += b.nthreads) s += d.ws_ncon[i];
    }
    if (last) {
      const int4* p4 = reinterpret_cast<const int4*>(d.ws_ncollision);
      const int n4 = d.nworld >> 2;
#pragma unroll 8
      for (int i = t; i < n4; i += b.nthreads) {
        const int4 v = p4[i];
        s2 += v.x + v.y + v.z + v.w;
      }
      for (int i = (n4 << 2) + t; i < d.nworld; i += b.nthreads) s2 += d.ws_ncollision[i];
    }
    for (int off = 32; off > 0; off >>= 1) {
      s += __shfl_xor(s, off, 64);
      s2 += __shfl_xor(s2, off, 64);
    }
    if ((t & 63) == 0) {
      sh[t >> 6] = s;
      sh[16 + (t >> 6)] = s2;
    }
    __syncthreads();
    int base = 0, tot2 = 0;
    for (int k = 0; k < nwave; ++k) {
      base += sh[k];
      tot2 += sh[16 + k];
    }
    if (w < d.nworld) {
      // exclusive prefix inside the workgroup: worlds of a workgroup are consecutive
      int adr = base;
      for (int i = b.w0; i < w; ++i) adr += d.ws_ncon[i];
      adr_self = adr;
      if (lig == 0) d.ws_conadr[w] = adr;
      if (last && w == d.nworld - 1 && lig == 0) {
        d.nacon[0] = adr + d.ws_ncon[w];
        d.ncollision[0] = tot2;
      }
    }
  }
  if (w >= d.nworld) return;
  const int ncon = d.ws_ncon[w], adr = self_prefix ? adr_self : d.ws_conadr[w], njmax = d.njmax, npyr = d.nmaxpyramid;
  int n = ncon;
  if (adr + n > d.naconmax) n = max(0, d.naconmax - adr);
  if (n < ncon && lig == 0) atomicOr(d.overflow + w, OVF_NARROWPHASE);
  const float* rec = d.ws_contact + (size_t)w * d.concap * CON_STRIDE;
  const int* reci = reinterpret_cast<const int*>(rec);
  const size_t o0 = (size_t)adr;
  for (int c = lig; c < n; c += G) {  // one contact per lane: the scalar-per-contact arrays
    const float* r = rec + c * CON_STRIDE;
    const int* ri = reci + c * CON_STRIDE;
    const size_t o = o0 + c;
    d.contact_dist[o] = r[0];
    d.contact_includemargin[o] = r[13];
    *reinterpret_cast<float2*>(d.contact_solref + 2 * o) = float2{r[17], r[18]};
    *reinterpret_cast<float2*>(d.contact_solreffriction + 2 * o) = float2{0.0f, 0.0f};
    d.contact_dim[o] = ri[24];
    *reinterpret_cast<int2*>(d.contact_geom + 2 * o) = int2{ri[25], ri[26]};
    d.contact_worldid[o] = w;
    d.contact_type[o] = CONTACT_TYPE_CONSTRAINT;
    d.contact_geomcollisionid[o] = ri[27];
    const int rbase = ri[28], ndim = ri[29];
    for (int k = 0; k < ndim; ++k)
      if (rbase >= 0 && rbase + k < njmax) d.efc_id[(size_t)w * njmax + rbase + k] = adr + c;
  }
  for (int idx = lig; idx < 3 * n; idx += G) d.contact_pos[3 * o0 + idx] = rec[(idx / 3) * CON_STRIDE + 1 + idx % 3];
  for (int idx = lig; idx < 9 * n; idx += G) d.contact_frame[9 * o0 + idx] = rec[(idx / 9) * CON_STRIDE + 4 + idx % 9];
  for (int idx = lig; idx < 5 * n; idx += G) {
    const int q = idx % 5;
    d.contact_friction[5 * o0 + idx] = rec[(idx / 5) * CON_STRIDE + 14 + (q < 2 ? 0 : q - 1 - (q == 4 ? 1 : 0))];
    d.contact_solimp[5 * o0 + idx] = rec[(idx / 5) * CON_STRIDE + 19 + q];
  }
  for (int idx = lig; idx < npyr * n; idx += G) {
    const int c = idx / npyr, k = idx % npyr;
    const int rbase = reci[c * CON_STRIDE + 28], ndim = reci[c * CON_STRIDE + 29];
    d.contact_efc_address[npyr * o0 + idx] = (rbase >= 0 && k < ndim && rbase + k < njmax) ? rbase + k : -1;
  }
}

template <int G>
__global__ void __launch_bounds__(256) k_publish_contacts(MjhData d, int self_prefix) {
  __shared__ int sh[64];
  publish_body<G>(d, self_prefix, sh, blk_of_launch<G>());
}
