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
#include "convex.hpp"
#include "contact_rec.hpp"

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

// ---- box_box core:589 (MuJoCo's mjc_BoxBox); same structure as oracle/mjref.c:box_box ------------------------------------
DEV void bb_face_rot(int f, float* r) {  // rotation taking face f of a box to +z (core:557)
  const float T[6][9] = {{0, 0, -1, 0, 1, 0, 1, 0, 0}, {1, 0, 0, 0, 0, -1, 0, 1, 0}, {1, 0, 0, 0, 1, 0, 0, 0, 1},
                         {0, 0, 1, 0, 1, 0, -1, 0, 0}, {1, 0, 0, 0, 0, 1, 0, -1, 0}, {-1, 0, 0, 0, 1, 0, 0, 0, -1}};
  for (int k = 0; k < 9; ++k) r[k] = T[f][k];
}
DEV void bb_mul(float* r, const float* a, const float* b) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}
DEV void bb_T(float* r, const float* a) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r[3 * i + j] = a[3 * j + i];
}
DEV void bb_mv(float* r, const float* m, const float* v) {
  for (int i = 0; i < 3; ++i) r[i] = m[3 * i] * v[0] + m[3 * i + 1] * v[1] + m[3 * i + 2] * v[2];
}
// returns the number of contacts; pts (world) / depth / normal (world) are filled
DEV int box_box(V3 P1, const float* R1, V3 S1, V3 P2, const float* R2, V3 S2, float margin, float (*wp)[3], float* depth, float* normal) {
  const float p1[3] = {P1.x, P1.y, P1.z}, p2[3] = {P2.x, P2.y, P2.z}, s1[3] = {S1.x, S1.y, S1.z}, s2[3] = {S2.x, S2.y, S2.z};
  float pos21[3], pos12[3], R1T[9], R2T[9], rot21[9], rot12[9], a21[9], a12[9], plen1[3], plen2[3], dv[3];
  bb_T(R1T, R1);
  bb_T(R2T, R2);
  for (int k = 0; k < 3; ++k) dv[k] = p2[k] - p1[k];
  bb_mv(pos21, R1T, dv);
  for (int k = 0; k < 3; ++k) dv[k] = -dv[k];
  bb_mv(pos12, R2T, dv);
  bb_mul(rot21, R1T, R2);
  bb_T(rot12, rot21);
  for (int k = 0; k < 9; ++k) {
    a21[k] = fabsf(rot21[k]);
    a12[k] = fabsf(rot12[k]);
  }
  bb_mv(plen2, a21, s2);
  bb_mv(plen1, a12, s1);
  float separation = margin + 3.0f * (s1[0] + s2[0]) + 3.0f * (s1[1] + s2[1]) + 3.0f * (s1[2] + s2[2]);
  int code = -1;
  const float tie = 5e-7f * (s1[0] + s1[1] + s1[2] + s2[0] + s2[1] + s2[2]);  // ~3x the float32 round-off of the overlap sums
  for (int i = 0; i < 3; ++i) {  // face normals
    const float c1 = -fabsf(pos21[i]) + s1[i] + plen2[i], c2 = -fabsf(pos12[i]) + s2[i] + plen1[i];
    if (c1 < -margin || c2 < -margin) return 0;
    // ties go to the earlier candidate, not to round-off (see the edge axes below)
    if (c1 < separation - tie) { separation = c1; code = i + 3 * (pos21[i] < 0.0f ? 1 : 0); }
    if (c2 < separation - tie) { separation = c2; code = i + 3 * (pos12[i] < 0.0f ? 1 : 0) + 6; }
  }
  float clnorm[3] = {0, 0, 0};
  int inv = 0, cle1 = 0, cle2 = 0;
  for (int i = 0; i < 3; ++i) {  // edge i of box 1 x edge j of box 2
    for (int j = 0; j < 3; ++j) {
      const float* a = rot12 + 3 * j;  // axis j of box 2 in the frame of box 1
      float cr[3];
      if (i == 0) { cr[0] = 0.0f; cr[1] = -a[2]; cr[2] = a[1]; }
      else if (i == 1) { cr[0] = a[2]; cr[1] = 0.0f; cr[2] = -a[0]; }
      else { cr[0] = -a[1]; cr[1] = a[0]; cr[2] = 0.0f; }
      const float len = sqrtf(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);
      if (len < MJ_MINVAL) continue;
      for (int k = 0; k < 3; ++k) cr[k] /= len;
      const float bd = pos21[0] * cr[0] + pos21[1] * cr[1] + pos21[2] * cr[2];
      float c3 = 0.0f;
      for (int k = 0; k < 3; ++k) {
        if (k != i) c3 += s1[k] * fabsf(cr[k]);
        if (k != j) c3 += s2[k] * a21[3 * i + (3 - k - j)] / len;
      }
      c3 -= fabsf(bd);
      if (c3 < -margin) return 0;
      // An edge axis only replaces a face axis when it is clearly less penetrated.  MuJoCo C guards the tie of flat face-face
      // contact (every edge cross product is then parallel to the face normal and has the same overlap) with a relative
      // 1e-12, which float32 cannot represent (the reference's float32 kernel keeps the constant, so its guard is a no-op);
      // here the guard is scaled to the float32 round-off of the overlap sums (`tie`, sub-micrometre for decimetre boxes).
      if (c3 < separation - tie) {
        separation = c3;
        cle1 = cle2 = 0;
        for (int k = 0; k < 3; ++k) {
          if (k != i && ((cr[k] > 0.0f) != (bd < 0.0f))) cle1 += 1 << k;
          if (k != j && (((rot21[3 * i + (3 - k - j)] > 0.0f) != (bd < 0.0f)) != (((k - j + 3) % 3) == 1))) cle2 += 1 << k;
        }
        code = 12 + 3 * i + j;
        for (int k = 0; k < 3; ++k) clnorm[k] = cr[k];
        inv = bd < 0.0f;
      }
    }
  }
  if (code == -1) return 0;
  float pts[8][3], rw[9], pw[3], hz;
  int n = 0;
  if (code < 12) {  // a face of one box against vertices / edges of the other
    const int f = code % 6, bi = code / 6;
    float rm[9], rmT[9], r[9], rt[9], pp[3], ss[3], tmp[3];
    bb_face_rot(f, rm);
    bb_T(rmT, rm);
    bb_mul(r, rm, bi ? rot12 : rot21);
    bb_mv(pp, rm, bi ? pos12 : pos21);
    bb_mv(tmp, rm, bi ? s2 : s1);
    for (int k = 0; k < 3; ++k) ss[k] = fabsf(tmp[k]);
    const float* so = bi ? s1 : s2;  // half sizes of the other box
    bb_T(rt, r);
    const float lx = ss[0], ly = ss[1];
    hz = ss[2];
    pp[2] -= hz;
    int corner = 0;
    for (int i = 0; i < 3; ++i)
      if (r[6 + i] < 0.0f) corner += 1 << i;
    float lp[3], cn1[3] = {0, 0, 0}, cn2[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k) lp[k] = pp[k];
    for (int i = 0; i < 3; ++i)
      for (int k = 0; k < 3; ++k) lp[k] += rt[3 * i + k] * so[i] * (((corner >> i) & 1) ? 1.0f : -1.0f);
    int dirs = 0;
    for (int i = 0; i < 3; ++i) {
      if (fabsf(r[6 + i]) < 0.5f) {
        const float sc = so[i] * (((corner >> i) & 1) ? -2.0f : 2.0f);
        for (int k = 0; k < 3; ++k) (dirs ? cn2 : cn1)[k] = rt[3 * i + k] * sc;
        ++dirs;
      }
    }
    // candidates are filtered (z <= margin) as they are produced; the reference compacts afterwards, same order
    auto push = [&](float x, float y, float z) {
      if (z > margin || n >= 8) return;
      pts[n][0] = x;
      pts[n][1] = y;
      pts[n][2] = 0.5f * z;
      depth[n] = z;
      ++n;
    };
    for (int i = 0; i < dirs * dirs; ++i) {  // edges of the other box's lowest face against the face rectangle
      for (int q = 0; q < 2; ++q) {
        float lav[3], lbv[3];
        for (int k = 0; k < 3; ++k) {
          lav[k] = lp[k] + (i < 2 ? 0.0f : (i == 2 ? cn1[k] : cn2[k]));
          lbv[k] = (i == 0 || i == 3) ? cn1[k] : cn2[k];
        }
        if (fabsf(lbv[q]) > MJ_MINVAL) {
          const float br = 1.0f / lbv[q];
          for (int j = -1; j <= 1; j += 2) {
            const float l = ss[q] * (float)j, c1 = (l - lav[q]) * br;
            if (c1 < 0.0f || c1 > 1.0f) continue;
            const float c2 = lav[1 - q] + lbv[1 - q] * c1;
            if (fabsf(c2) > ss[1 - q]) continue;
            push(lav[0] + c1 * lbv[0], lav[1] + c1 * lbv[1], lav[2] + c1 * lbv[2]);
          }
        }
      }
    }
    if (dirs == 2) {  // rectangle corners inside the other face's parallelogram
      const float ax = cn1[0], bx = cn2[0], ay = cn1[1], by = cn2[1], C = safe_div(1.0f, ax * by - bx * ay);
      for (int i = 0; i < 4; ++i) {
        const float llx = (i / 2) ? lx : -lx, lly = (i % 2) ? ly : -ly, x = llx - lp[0], y = lly - lp[1];
        const float u = (x * by - y * bx) * C, v = (y * ax - x * ay) * C;
        if (u > 0.0f && v > 0.0f && u < 1.0f && v < 1.0f) push(llx, lly, lp[2] + u * cn1[2] + v * cn2[2]);
      }
    }
    for (int i = 0; i < (1 << dirs); ++i) {  // the other box's corners above the rectangle
      float t[3];
      for (int k = 0; k < 3; ++k) t[k] = lp[k] + (float)(i & 1) * cn1[k] + ((i & 2) ? cn2[k] : 0.0f);
      if (t[0] > -lx && t[0] < lx && t[1] > -ly && t[1] < ly) push(t[0], t[1], t[2]);
    }
    bb_mul(rw, bi ? R2 : R1, rmT);
    for (int k = 0; k < 3; ++k) {
      pw[k] = bi ? p2[k] : p1[k];
      normal[k] = (bi ? -1.0f : 1.0f) * rw[3 * k + 2];
    }
  } else {  // edge against edge
    const int e1 = (code - 12) / 3, e2 = (code - 12) % 3;
    int ax1 = 1 - (e2 & 1), ax2 = 2 - (e2 & 2), pax1 = 1 - (e1 & 1), pax2 = 2 - (e1 & 2);
    if (a21[3 * e1 + ax1] < a21[3 * e1 + ax2]) { const int t = ax1; ax1 = ax2; ax2 = t; }
    if (a12[3 * e2 + pax1] < a12[3 * e2 + pax2]) { const int t = pax1; pax1 = pax2; pax2 = t; }
    float rm[9], rmT[9], pp[3], rnorm[3], r[9], rt[9], tmp[3], sz[3];
    bb_face_rot((cle1 & (1 << pax2)) ? pax2 : pax2 + 3, rm);
    bb_T(rmT, rm);
    bb_mv(pp, rm, pos21);
    bb_mv(rnorm, rm, clnorm);
    bb_mul(r, rm, rot21);
    bb_T(rt, r);
    bb_mv(tmp, rmT, s1);
    for (int k = 0; k < 3; ++k) sz[k] = fabsf(tmp[k]);
    const float lx = sz[0], ly = sz[1];
    hz = sz[2];
    pp[2] -= hz;
    float q4[4][3];  // the face of box 2 nearest to box 1, as 4 points
    for (int k = 0; k < 3; ++k) {
      const float b1 = rt[3 * ax1 + k] * s2[ax1], b2 = rt[3 * ax2 + k] * s2[ax2], be = rt[3 * e2 + k] * s2[e2];
      const float sg1 = (cle2 & (1 << ax1)) ? 1.0f : -1.0f, sg2 = (cle2 & (1 << ax2)) ? 1.0f : -1.0f;
      const float base0 = pp[k] + b1 * sg1 + b2 * sg2, base2 = pp[k] - b1 * sg1 + b2 * sg2;
      q4[0][k] = base0 + be;
      q4[1][k] = base0 - be;
      q4[2][k] = base2 + be;
      q4[3][k] = base2 - be;
    }
    float axi_lp[3], axi_cn1[3], axi_cn2[3];
    for (int k = 0; k < 3; ++k) {
      axi_lp[k] = q4[0][k];
      axi_cn1[k] = q4[1][k] - q4[0][k];
      axi_cn2[k] = q4[2][k] - q4[0][k];
    }
    if (fabsf(rnorm[2]) < MJ_MINVAL) return 0;
    const float sgn = inv ? -1.0f : 1.0f, innorm = sgn / rnorm[2];
    float pu[4][3];
    for (int i = 0; i < 4; ++i) {  // project along the contact normal onto the plane z = 0
      const float c = q4[i][2] * sgn * innorm;
      for (int k = 0; k < 3; ++k) {
        pu[i][k] = q4[i][k];
        q4[i][k] -= rnorm[k] * c;
      }
    }
    float lp[3], cn1[3], cn2[3];
    for (int k = 0; k < 3; ++k) {
      lp[k] = q4[0][k];
      cn1[k] = q4[1][k] - q4[0][k];
      cn2[k] = q4[2][k] - q4[0][k];
    }
    for (int i = 0; i < 4; ++i) {
      for (int q = 0; q < 2; ++q) {
        const float la = lp[q] + (i < 2 ? 0.0f : (i == 2 ? cn1[q] : cn2[q])), lb = (i == 0 || i == 3) ? cn1[q] : cn2[q];
        const float lc = lp[1 - q] + (i < 2 ? 0.0f : (i == 2 ? cn1[1 - q] : cn2[1 - q])), ld = (i == 0 || i == 3) ? cn1[1 - q] : cn2[1 - q];
        float lua[3], lub[3];
        for (int k = 0; k < 3; ++k) {
          lua[k] = axi_lp[k] + (i < 2 ? 0.0f : (i == 2 ? axi_cn1[k] : axi_cn2[k]));
          lub[k] = (i == 0 || i == 3) ? axi_cn1[k] : axi_cn2[k];
        }
        if (fabsf(lb) > MJ_MINVAL) {
          const float br = 1.0f / lb;
          for (int j = -1; j <= 1; j += 2) {
            if (n == 8) break;
            const float l = sz[q] * (float)j, c1 = (l - la) * br;
            if (c1 < 0.0f || c1 > 1.0f) continue;
            const float c2 = lc + ld * c1;
            if (fabsf(c2) > sz[1 - q]) continue;
            if ((lua[2] + lub[2] * c1) * innorm > margin) continue;
            for (int k = 0; k < 3; ++k) pts[n][k] = lua[k] * 0.5f + c1 * lub[k] * 0.5f;
            pts[n][q] += 0.5f * l;
            pts[n][1 - q] += 0.5f * c2;
            depth[n] = pts[n][2] * innorm * 2.0f;
            ++n;
          }
        }
      }
    }
    const int nl = n;
    const float ax = cn1[0], bx = cn2[0], ay = cn1[1], by = cn2[1], C = safe_div(1.0f, ax * by - bx * ay);
    for (int i = 0; i < 4; ++i) {
      if (n == 8) break;
      const float llx = (i / 2) ? lx : -lx, lly = (i % 2) ? ly : -ly, x = llx - lp[0], y = lly - lp[1];
      float u = (x * by - y * bx) * C, v = (y * ax - x * ay) * C;
      if (nl == 0) {
        if ((u < 0.0f || u > 1.0f) && (v < 0.0f || v > 1.0f)) continue;
      } else if (u < 0.0f || v < 0.0f || u > 1.0f || v > 1.0f) continue;
      u = clampf(u, 0.0f, 1.0f);
      v = clampf(v, 0.0f, 1.0f);
      const float w = 1.0f - u - v;
      float vt[3], tc1 = 0.0f;
      const float corner3[3] = {llx, lly, 0.0f};
      for (int k = 0; k < 3; ++k) {
        vt[k] = pu[0][k] * w + pu[1][k] * u + pu[2][k] * v;
        tc1 += (corner3[k] - vt[k]) * (corner3[k] - vt[k]);
      }
      if (vt[2] > 0.0f && tc1 > margin * margin) continue;
      for (int k = 0; k < 3; ++k) pts[n][k] = 0.5f * (corner3[k] + vt[k]);
      depth[n] = sqrtf(tc1) * (vt[2] < 0.0f ? -1.0f : 1.0f);
      ++n;
    }
    const int nf = n;
    for (int i = 0; i < 4; ++i) {
      if (n >= 8) break;
      const float x = pu[i][0], y = pu[i][1];
      if (nl == 0 && nf != 0) {
        if ((x < -lx || x > lx) && (y < -ly || y > ly)) continue;
      } else if (x < -lx || x > lx || y < -ly || y > ly) continue;
      float c1 = 0.0f;
      for (int j = 0; j < 2; ++j) {
        if (pu[i][j] < -sz[j]) c1 += (pu[i][j] + sz[j]) * (pu[i][j] + sz[j]);
        else if (pu[i][j] > sz[j]) c1 += (pu[i][j] - sz[j]) * (pu[i][j] - sz[j]);
      }
      c1 += pu[i][2] * innorm * pu[i][2] * innorm;
      if (pu[i][2] > 0.0f && c1 > margin * margin) continue;
      float tp[3] = {pu[i][0], pu[i][1], 0.0f};
      for (int j = 0; j < 2; ++j) {
        if (pu[i][j] < -sz[j]) tp[j] = -sz[j] * 0.5f;
        else if (pu[i][j] > sz[j]) tp[j] = sz[j] * 0.5f;
      }
      for (int k = 0; k < 3; ++k) pts[n][k] = 0.5f * (tp[k] + pu[i][k]);
      depth[n] = sqrtf(c1) * (pu[i][2] < 0.0f ? -1.0f : 1.0f);
      ++n;
    }
    bb_mul(rw, R1, rmT);
    float nn[3];
    bb_mv(nn, rw, rnorm);
    for (int k = 0; k < 3; ++k) {
      pw[k] = p1[k];
      normal[k] = sgn * nn[k];
    }
  }
  for (int i = 0; i < n; ++i) {
    pts[i][2] += hz;
    bb_mv(wp[i], rw, pts[i]);
    for (int k = 0; k < 3; ++k) wp[i][k] += pw[k];
  }
  return n;
}

// convex pairs go through GJK / EPA (collision_convex.py:747-977): both geoms carry the pair's margin (support points are inflated by
// half of it), the GJK cutoff is the gap, the reported distance is un-inflated, the contact sits midway between the witness points
// ---- the convex pair in two launches (round 4; convex.hpp header) -----------------------------------------------------------------------
// GJK by ONE lane (CG = 0) or by the CG lanes of a group together (identical arguments in every lane; the mesh support function spreads the
// neighbours of a hill-climbing step over the lanes).  Returns 0: no contact, 1: one contact, emitted (no penetration deeper than the tolerance: EPA not needed), 2: EPA
// needed -- `res` holds the simplex, idx1 / idx2 the mesh vertex caches the support function left
template <int CG = 0, class Emit>
DEV int convex_gjk_lane(float tolerance, int iterations, int t1, int t2, V3 p1, const float* R1, V3 s1, V3 p2, const float* R2, V3 s2, float margin, float gap, Emit&& emit,
                        const float* vert1, int nvert1, const float* vert2, int nvert2, const MjhModel& mm, int mesh1, int mesh2, GjkOut& res, int& idx1, int& idx2, int lig = 0) {
  auto graph_of = [&](int meshid) -> const int* { return (meshid >= 0 && mm.mesh_graphadr[meshid] >= 0) ? mm.mesh_graph + mm.mesh_graphadr[meshid] : nullptr; };
  CcdGeom a = CcdGeom{t1, p1, R1, s1, margin, vert1, nvert1, -1, mesh1, graph_of(mesh1), -1, nullptr}, b = CcdGeom{t2, p2, R2, s2, margin, vert2, nvert2, -1, mesh2, graph_of(mesh2), -1, nullptr};
  float dist;
  V3 w1, w2;
  const int r = ccd_gjk_phase<CG>(tolerance, gap, iterations, a, b, dist, w1, w2, res, lig);
  idx1 = a.index;
  idx2 = b.index;
  if (r == 2) return 2;
  if (dist >= gap) return 0;
  dist += margin;
  const Frame f = make_frame3(dist <= margin ? w1 - w2 : w2 - w1);
  emit(0, dist, 0.5f * (w1 + w2), f.a, f.b, f.c);
  return 1;
}
// EPA + multi-contact recovery by the CG lanes of a group TOGETHER (identical arguments in every lane): from the simplex in `res`;
// `poly` = the group's LDS polytope (stride 1), `mcws` = the group's multi-contact words (stride 1)
template <int CG, class Emit>
DEV void convex_epa_group(float tolerance, int epa_iterations, int t1, int t2, V3 p1, const float* R1, V3 s1, V3 p2, const float* R2, V3 s2, float margin, float gap, float* poly,
                          int& overflow, Emit&& emit, const float* vert1, int nvert1, const float* vert2, int nvert2, const MjhModel& mm, int mesh1, int mesh2, GjkOut& res,
                          int idx1, int idx2, int lig, float* mcws) {
  auto graph_of = [&](int meshid) -> const int* { return (meshid >= 0 && mm.mesh_graphadr[meshid] >= 0) ? mm.mesh_graph + mm.mesh_graphadr[meshid] : nullptr; };
  const CcdGeom a = CcdGeom{t1, p1, R1, s1, margin, vert1, nvert1, idx1, mesh1, graph_of(mesh1), idx1, nullptr}, b = CcdGeom{t2, p2, R2, s2, margin, vert2, nvert2, idx2, mesh2, graph_of(mesh2), idx2, nullptr};
  float dist = res.dist;
  V3 w1 = res.x1, w2 = res.x2;
  int face;
  Poly pt;
#ifdef MJH_DBG_EPA_SKIP  // (profiling variants, tools/build_variant_fast.py: the launch without EPA / without the multi-contact recovery)
  return;
#endif
  DBG_TICK_START();
  int n = ccd_epa_phase<CG>(tolerance, epa_iterations, a, b, res, poly, dist, w1, w2, overflow, face, pt, lig, 1);
  DBG_TICK(2);  // EPA proper
  if (n == 0 || dist >= gap) return;
  dist += margin;
  const bool anymesh = t1 == G_MESH || t2 == G_MESH;
  // (nmeshdegmax == 0: the model was put with MULTICCD disabled -- no multi-contact scratch was sized, whatever the flag says now)
  if (face >= 0 && anymesh && ((mm.disableflags & DSBL_MULTICCD) || mm.nmeshpoly == 0 || mm.nmeshdegmax == 0)) face = -1;
#ifdef MJH_DBG_EPA_NOMC
  face = -1;
#endif
  if (face >= 0) {  // zero margin: up to four contacts from the EPA face, same distance and frame (collision_convex.py:888-960)
    V3 m1[4], m2[4];
    n = anymesh ? ccd_multicontact_mesh_inl(mm, pt, face, w1, w2, a, b, m1, m2, mcws, 1, poly, ccd_coop_face_offset(max(mm.ccd_iterations, mm.epa_iterations), mm.npolygonmax, mm.nmeshdegmax), ccd_coop_scratch_offset(max(mm.ccd_iterations, mm.epa_iterations)), lig, CG)
                : ccd_multicontact_box(pt, face, w1, w2, a, b, m1, m2);
    if (n == 0) return;
    const Frame f = make_frame3(dist <= margin ? m1[0] - m2[0] : m2[0] - m1[0]);
    for (int i = 0; i < n; ++i) emit(i, dist, 0.5f * (m1[i] + m2[i]), f.a, f.b, f.c);
    return;
  }
  const Frame f = make_frame3(dist <= margin ? w1 - w2 : w2 - w1);
  emit(0, dist, 0.5f * (w1 + w2), f.a, f.b, f.c);
}
// candidate capacity per world: every filtered pair for ordinary models; capped for big scenes (thousands of pairs of which few
// survive the broadphase), where running out sets OverflowType.BROADPHASE like the reference's full pair queue
__host__ __device__ inline int collide_ccap(int npair, int concap) {
  const int cap = 8 * concap > 512 ? 8 * concap : 512;
  return ((npair < cap ? npair : cap) + 3) / 4 * 4;
}
// EPA entries a Data can hold (Data.nccdhand): 32 per world on average, and never fewer than 4,096 (or every candidate of every world)
// for small batches; beyond it OverflowType.CCD and the pair is dropped
__host__ __device__ inline int ccd_handcap(int nworld, int ccap) {
  const long long all = (long long)nworld * ccap, avg = (long long)nworld * (ccap < 32 ? ccap : 32), lo = all < 4096 ? all : 4096;
  return (int)(avg > lo ? avg : lo);
}
DEV CcdLayout ccd_layout_of(const MjhModel& m, const MjhData& d) {
  return ccd_layout(d.nworld, max(m.ccd_iterations, m.epa_iterations), m.nhfield, m.npolygonmax, m.nmeshdegmax, collide_ccap(m.npair, d.concap), d.nccdhand, m.npair);
}
DEV bool is_ccd_pair(const MjhModel& m, int t1, int t2) {  // the pairs served by k_ccd_gjk / k_ccd_epa (height fields stay in the contact kernel)
  return t1 != G_HFIELD && (is_convex_pair(t1, t2) || (!(m.disableflags & DSBL_NATIVECCD) && t1 == G_BOX && t2 == G_BOX));
}
// the cache entry of a convex candidate: count | distance | frame | up to four positions
DEV void ccd_cache_store(float* cache, int& nem, int k, float dist, V3 pos, V3 fa, V3 fb, V3 fc, bool store) {
  if (k < 4) {
    if (store) {
      if (k == 0) {
        cache[1] = dist;
        st3(cache + 2, fa);
        st3(cache + 5, fb);
        st3(cache + 8, fc);
      }
      st3(cache + 11 + 3 * k, pos);
    }
    nem = k + 1;
  }
}

// plane - convex mesh (collision_primitive.py:52-274 plane_convex, the exhaustive branch: meshes without a hill-climbing graph or with
// fewer than 10 vertices): the deepest vertex a, then among the vertices within 1 mm of it the one furthest from a, the one furthest
// from the line a-b and the one furthest from the other two edges; vertices that were picked once become contacts (at most 4)
template <class Emit>
DEV void plane_mesh(V3 pn, V3 pp, V3 mp, const float* R, const float* vert, int nvert, Emit&& emit, const int* graph = nullptr) {
  const float HUGE_V = 1e6f;
  const V3 pl = matT_mul(R, pp - mp), nl = matT_mul(R, pn);
  int idx[4] = {-1, -1, -1, -1};
  float max_support = -HUGE_V;
  V3 a = V3{0, 0, 0}, b = a, c = a;
  if (graph && nvert >= 10) {
    // collision_primitive.py:131-243: the same four picks by hill climbing on the hull's vertex graph, each climb starting where the
    // previous one ended (local maxima along the graph; no early exit for a separated mesh, unlike the exhaustive branch)
    const int numvert = graph[0];
    const int *edgeadr = graph + 2, *globalid = graph + 2 + numvert, *edge = graph + 2 + 2 * numvert;
    int imax = 0, prev;
    float threshold = 0.0f;
    V3 ab = a, ac = a, bc = a;
    auto climb = [&](auto&& score) {  // score(vertex) -> value to maximise; the running best persists across the sweeps of one climb
      float best = -HUGE_V;
      do {
        prev = imax;
        for (int i = edgeadr[imax]; edge[i] >= 0; ++i) {
          const float dd = score(ld3(vert + 3 * globalid[edge[i]]));
          if (dd > best) {
            best = dd;
            imax = edge[i];
          }
        }
      } while (imax != prev);
      return best;
    };
    max_support = climb([&](V3 v) { return dot(pl - v, nl); });
    threshold = fmaxf(0.0f, max_support - 1e-3f);
    auto mask = [&](V3 v) { return dot(pl - v, nl) > threshold ? 0.0f : -HUGE_V; };
    climb([&](V3 v) {
      const float sup = dot(pl - v, nl);
      return sup > threshold ? sup : -HUGE_V;
    });
    idx[0] = globalid[imax];
    a = ld3(vert + 3 * idx[0]);
    climb([&](V3 v) { return dot(a - v, a - v) + mask(v); });
    idx[1] = globalid[imax];
    b = ld3(vert + 3 * idx[1]);
    ab = cross(nl, a - b);
    climb([&](V3 v) { return fabsf(dot(a - v, ab)) + mask(v); });
    idx[2] = globalid[imax];
    c = ld3(vert + 3 * idx[2]);
    ac = cross(nl, a - c);
    bc = cross(nl, b - c);
    climb([&](V3 v) { return (fabsf(dot(a - v, ac)) + mask(v)) + (fabsf(dot(b - v, bc)) + mask(v)); });
    idx[3] = globalid[imax];
  } else {
  for (int i = 0; i < nvert; ++i) {
    const V3 v = ld3(vert + 3 * i);
    const float sup = dot(pl - v, nl);
    if (sup > max_support) {
      max_support = sup;
      idx[0] = i;
      a = v;
    }
  }
  if (max_support < 0.0f) return;
  const float threshold = max_support - 1e-3f;
  float best = -HUGE_V;
  for (int i = 0; i < nvert; ++i) {
    const V3 v = ld3(vert + 3 * i);
    const float mask = dot(pl - v, nl) > threshold ? 0.0f : -HUGE_V;
    const float dd = dot(a - v, a - v) + mask;
    if (dd > best) {
      idx[1] = i;
      best = dd;
      b = v;
    }
  }
  const V3 ab = cross(nl, a - b);
  best = -HUGE_V;
  for (int i = 0; i < nvert; ++i) {
    const V3 v = ld3(vert + 3 * i);
    const float mask = dot(pl - v, nl) > threshold ? 0.0f : -HUGE_V;
    const float dd = fabsf(dot(a - v, ab)) + mask;
    if (dd > best) {
      idx[2] = i;
      best = dd;
      c = v;
    }
  }
  const V3 ac = cross(nl, a - c), bc = cross(nl, b - c);
  best = -HUGE_V;
  for (int i = 0; i < nvert; ++i) {
    const V3 v = ld3(vert + 3 * i);
    const float mask = dot(pl - v, nl) > threshold ? 0.0f : -HUGE_V;
    const float dd = (fabsf(dot(a - v, ac)) + mask) + (fabsf(dot(b - v, bc)) + mask);
    if (dd > best) {
      idx[3] = i;
      best = dd;
    }
  }
  }
  const Frame f = make_frame3(pn);
  int n = 0;
  for (int i = 3; i >= 0; --i) {
    int count = 0;
    for (int j = 0; j <= i; ++j) count += idx[j] == idx[i];
    if (count != 1) continue;
    const V3 v = ld3(vert + 3 * idx[i]);
    const float dist = -dot(pl - v, nl);
    emit(n++, dist, mp + mat_mul(R, v) - (0.5f * dist) * pn, f.a, f.b, f.c);
  }
}

// height field against a convex geom (collision_convex.py:60-161 _hfield_filter, 164-730; MuJoCo mjc_ConvexHField): in the height field's
// frame, every triangular prism of the cells under the geom's bounding box runs GJK / EPA against the geom; of the (at most 50) results up
// to four are kept: the deepest, the one furthest from it, the one furthest from that line, the one furthest from the other two edges.
// The G lanes of the world evaluate G prisms of one pair at a time (hfield_fill: every lane has its own EPA workspace) and rank the kept
// results by ballot into the owner lane's table (7 words per prism + a count, lane-interleaved) in the reference's row / column / triangle
// order; the owner lane then selects and emits (hfield_select).
template <int G>
__device__ __noinline__ void hfield_fill(const MjhModel& m, float tolerance, int iterations, int epa_iterations, int g1, int t2, V3 pos1, const float* mat1, V3 pos2,
                                         const float* mat2, V3 size2, float rbound2, float fmargin, float margin, float* scratch, float* hf, int& overflow, int lig,
                                         bool owner, const float* vert2, int nvert2, int mesh2) {
  int* hcount = reinterpret_cast<int*>(hf + (size_t)(7 * CCD_HF_MAXCONPAIR) * CCD_LANES);
  if (owner) *hcount = 0;
  const int hid = m.geom_dataid[g1];
  const float* size1 = m.hfield_size + 4 * hid;
  const V3 pos = matT_mul(mat1, pos2 - pos1);
  if (size1[0] < pos.x - rbound2 - fmargin || -size1[0] > pos.x + rbound2 + fmargin) return;
  if (size1[1] < pos.y - rbound2 - fmargin || -size1[1] > pos.y + rbound2 + fmargin) return;
  if (size1[2] < pos.z - rbound2 - fmargin) return;
  if (-size1[3] > pos.z + rbound2 + fmargin) return;
  float R[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[3 * i + j] = mat1[i] * mat2[j] + mat1[3 + i] * mat2[3 + j] + mat1[6 + i] * mat2[6 + j];
  const int* graph2 = (mesh2 >= 0 && m.mesh_graphadr[mesh2] >= 0) ? m.mesh_graph + m.mesh_graphadr[mesh2] : nullptr;
  CcdGeom b = CcdGeom{t2, pos, R, size2, 0.0f, vert2, nvert2, -1, mesh2, graph2, -1, nullptr};
  int vid;
  const float xmax = ccd_support(b, V3{1, 0, 0}, vid).x, xmin = ccd_support(b, V3{-1, 0, 0}, vid).x;
  const float ymax = ccd_support(b, V3{0, 1, 0}, vid).y, ymin = ccd_support(b, V3{0, -1, 0}, vid).y;
  const float zmax = ccd_support(b, V3{0, 0, 1}, vid).z, zmin = ccd_support(b, V3{0, 0, -1}, vid).z;
  if (xmin - fmargin > size1[0] || xmax + fmargin < -size1[0] || ymin - fmargin > size1[1] || ymax + fmargin < -size1[1] || zmin - fmargin > size1[2] ||
      zmax + fmargin < -size1[3])
    return;
  const int nrow = m.hfield_nrow[hid], ncol = m.hfield_ncol[hid], adr = m.hfield_adr[hid];
  const float x_scale = 0.5f * (float)(ncol - 1) / size1[0], y_scale = 0.5f * (float)(nrow - 1) / size1[1];
  const int cmin = max(0, (int)floorf((xmin + size1[0]) * x_scale)), cmax = min(ncol - 1, (int)ceilf((xmax + size1[0]) * x_scale));
  const int rmin = max(0, (int)floorf((ymin + size1[1]) * y_scale)), rmax = min(nrow - 1, (int)ceilf((ymax + size1[1]) * y_scale));
  const float dx = 2.0f * size1[0] / (float)(ncol - 1), dy = 2.0f * size1[1] / (float)(nrow - 1);
  b.margin = margin;  // the geom is inflated by half the margin, the prism tops are raised by the whole of it
  const float ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  // prism n of the strip of row r uses the strip's vertices n-2, n-1, n; vertex j sits at column cmin + j / 2, row r + 1 (j even) or r (j odd);
  // the reference visits c = cmin + 1 .. cmax, k = 0, 1, i.e. n = 2 (c - cmin) + k
  const int per_row = 2 * max(cmax - cmin, 0), total = max(rmax - rmin, 0) * per_row;
  auto vertex = [&](int r, int j, float& x, float& y, float& z) {
    const int col = cmin + (j >> 1), row = r + ((j & 1) ? 0 : 1);
    x = dx * (float)col - size1[0];
    y = dy * (float)row - size1[1];
    z = m.hfield_data[adr + row * ncol + col] * size1[2] + margin;
  };
  int count = 0, last_kept = -1;
  for (int b0 = 0; b0 < total && count < CCD_HF_MAXCONPAIR; b0 += G) {
    const int idx = b0 + lig;
    bool kept = false;
    float dist = 0.0f;
    V3 w1 = V3{0, 0, 0}, w2 = w1;
    if (idx < total) {
      const int r = rmin + idx / per_row, n = 2 + idx % per_row;
      V3 prism[6];
      for (int q = 0; q < 3; ++q) {
        float x, y, z;
        vertex(r, n - 2 + q, x, y, z);
        prism[q] = V3{x, y, -size1[3]};
        prism[3 + q] = V3{x, y, z};
      }
      if (!(prism[3].z < zmin && prism[4].z < zmin && prism[5].z < zmin)) {
        const V3 centre = (prism[0] + prism[1] + prism[2] + prism[3] + prism[4] + prism[5]) * (1.0f / 6.0f);
        const CcdGeom a = CcdGeom{G_HFIELD, centre, ident, V3{0, 0, 0}, 0.0f, nullptr, 0, -1, -1, nullptr, -1, prism};
        CcdGeom bb = b;  // (the reference passes its geom structs by value: every prism starts from the uncached geom)
        int face;
        Poly pt;
        kept = ccd_run(tolerance, 0.0f, iterations, epa_iterations, a, bb, scratch, dist, w1, w2, overflow, face, pt) != 0;
      }
    }
    const unsigned long long bits = gballot<G>(kept);
    const int slot = count + __popcll(bits & ((1ull << lig) - 1ull));
    if (kept) {
      if (slot < CCD_HF_MAXCONPAIR) {
        const V3 p = mat_mul(mat1, 0.5f * (w1 + w2)) + pos1;
        const V3 nrm = mat_mul(mat1, make_frame3(w1 - w2).a);
        float* e = hf + (size_t)(7 * slot) * CCD_LANES;
        e[0] = dist;
        e[1 * CCD_LANES] = p.x; e[2 * CCD_LANES] = p.y; e[3 * CCD_LANES] = p.z;
        e[4 * CCD_LANES] = nrm.x; e[5 * CCD_LANES] = nrm.y; e[6 * CCD_LANES] = nrm.z;
      } else {
        overflow |= OVF_HFIELD;
      }
    }
    if (bits) {
      // index of the prism that filled the table's last slot (if any): prisms after it overflow like in the reference's serial walk
      unsigned long long upto = bits;
      int need = CCD_HF_MAXCONPAIR - count;  // kept results still fitting
      if (__popcll(bits) >= need) {
        for (int q = 1; q < need; ++q) upto &= upto - 1;  // drop the need-1 lowest set bits: the lowest remaining is the filler
        last_kept = b0 + (__ffsll((long long)upto) - 1);
      }
    }
    count += __popcll(bits);
  }
  if (count >= CCD_HF_MAXCONPAIR && last_kept >= 0 && last_kept < total - 1) overflow |= OVF_HFIELD;
  __threadfence_block();
  if (owner) *hcount = min(count, CCD_HF_MAXCONPAIR);
}
// the owner lane's part: up to four of the table's entries (collision_convex.py:505-730)
template <class Emit>
DEV void hfield_select(float* hf, Emit&& emit) {
  const int count = *reinterpret_cast<const int*>(hf + (size_t)(7 * CCD_HF_MAXCONPAIR) * CCD_LANES);
  auto hfget = [&](int i, int q) -> float { return hf[(size_t)(7 * i + q) * CCD_LANES]; };
  auto hfpos = [&](int i) { return V3{hfget(i, 1), hfget(i, 2), hfget(i, 3)}; };
  auto hfnrm = [&](int i) { return V3{hfget(i, 4), hfget(i, 5), hfget(i, 6)}; };
  float min_dist = MJ_MAXVAL;
  V3 min_pos = V3{MJ_MAXVAL, MJ_MAXVAL, MJ_MAXVAL}, min_nrm = min_pos;
  int min_id = -1;
  for (int i = 0; i < count; ++i)
    if (hfget(i, 0) < min_dist) {
      min_dist = hfget(i, 0);
      min_pos = hfpos(i);
      min_nrm = hfnrm(i);
      min_id = i;
    }
  int nout = 0;
  auto put = [&](float dist, V3 p, V3 nrm) {
    const Frame f = make_frame3(nrm);
    emit(nout++, dist, p, f.a, f.b, f.c);
  };
  put(min_dist, min_pos, min_nrm);  // (written unconditionally; the caller's margin test drops an empty one)
  const float MIN_NEXT = 1.0e-3f;
  int id1 = -1, id2 = -1, id3 = -1;
  float best = -MJ_MAXVAL;
  for (int i = 0; i < count; ++i) {
    if (i == min_id) continue;
    const float dd = length(hfpos(i) - min_pos);
    if (dd > best) {
      id1 = i;
      best = dd;
    }
  }
  if (id1 == -1 || (0.0f < best && best < MIN_NEXT)) return;
  const V3 pos_1 = hfpos(id1);
  put(hfget(id1, 0), pos_1, hfnrm(id1));
  const V3 dmin1 = cross(min_nrm, min_pos - pos_1);
  best = -MJ_MAXVAL;
  for (int i = 0; i < count; ++i) {
    if (i == min_id || i == id1) continue;
    const float dd = fabsf(dot(hfpos(i) - min_pos, dmin1));
    if (dd > best) {
      id2 = i;
      best = dd;
    }
  }
  if (id2 == -1 || (0.0f < best && best < MIN_NEXT)) return;
  const V3 pos_2 = hfpos(id2);
  put(hfget(id2, 0), pos_2, hfnrm(id2));
  const V3 vmin2 = cross(min_nrm, min_pos - pos_2), v12 = cross(min_nrm, pos_1 - pos_2);
  best = -MJ_MAXVAL;
  for (int i = 0; i < count; ++i) {
    if (i == min_id || i == id1 || i == id2) continue;
    const V3 p = hfpos(i);
    const float dd = fabsf(dot(p - min_pos, vmin2)) + fabsf(dot(pos_1 - p, v12));
    if (dd > best) {
      id3 = i;
      best = dd;
    }
  }
  if (id3 == -1 || (0.0f < best && best < MIN_NEXT)) return;
  put(hfget(id3, 0), hfpos(id3), hfnrm(id3));
}

template <bool HEAVY, class Emit>
DEV void collide_pair(int t1, int t2, V3 p1, const float* R1, V3 s1, V3 p2, const float* R2, V3 s2, float margin, Emit&& emit, const float* vert2 = nullptr,
                      int nvert2 = 0, const int* graph2 = nullptr) {
  V3 ax1 = V3{R1[2], R1[5], R1[8]}, ax2 = V3{R2[2], R2[5], R2[8]};
  float dist;
  V3 pos, nn;
  if (HEAVY && t1 == G_PLANE && t2 == G_MESH) {
    plane_mesh(ax1, p1, p2, R2, vert2, nvert2, emit, graph2);
  } else if (t1 == G_PLANE && t2 == G_SPHERE) {
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
  } else if (HEAVY && t1 == G_BOX && t2 == G_BOX) {  // core:589
    float wpts[8][3], dep[8], nrm[3];
    const int n = box_box(p1, R1, s1, p2, R2, s2, margin, wpts, dep, nrm);
    const Frame f = make_frame3(V3{nrm[0], nrm[1], nrm[2]});
    for (int c = 0; c < n; ++c) emit(c, dep[c], V3{wpts[c][0], wpts[c][1], wpts[c][2]}, f.a, f.b, f.c);
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
DEV PairParams contact_params(const MjhModel& m, int w, int g1, int g2, int pid = -1) {
  PairParams p;
  if (pid >= 0) {  // explicit <contact><pair>: its own parameters (collision_core.py contact_params, pairid >= 0)
    p.condim = m.pair_dim[pid];
    for (int k = 0; k < 5; ++k) {
      p.friction[k] = fmaxf(MJ_MINMU, m.pair_friction[5 * pid + k]);
      p.solimp[k] = m.pair_solimp[5 * pid + k];
    }
    for (int k = 0; k < 2; ++k) {
      p.solref[k] = m.pair_solref[2 * pid + k];
      p.solreffriction[k] = m.pair_solreffriction[2 * pid + k];
    }
    p.margin = m.pair_margin[pid];
    p.gap = m.pair_gap[pid];
    return p;
  }
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

// (contact record layout and publish_body: contact_rec.hpp)
// per-world LDS: geom poses (12 words per geom) | candidate pair list | first contact slot << 8 | contact mask per
// candidate | staging window of CON_WINDOW records
// SAP broadphase (sap != 0) adds: projection bounds (2 words per geom, padded to a power of two for the bitonic sort) | sorted
// geom ids | one mark bit per filtered pair
__host__ __device__ inline int sap_pow2(int n) {
  int p = 1;
  while (p < n) p <<= 1;
  return p;
}
// k_ccd_broad's slice: position | radius per geom (4 words), candidate list, slots, queue (+ the SAP arrays)
__host__ __device__ inline int broad_lds_words(int ngeom, int npair, int concap, int sap = 0) {
  if (!sap) return 2 * collide_ccap(npair, concap);  // NXN: the filters ran in k_broad_mask -- candidate list and slots only
  const int base = 4 * ngeom + 2 * collide_ccap(npair, concap) + CON_WINDOW * CON_LDS;
  return ((base + 3 * sap_pow2(ngeom) + ((npair + 31) / 32 + 3) / 4 * 4) + 3) / 4 * 4;
}
// pre: the candidate list and the convex results come from the launches in front of the kernel (models with GJK pairs) -- no broadphase here,
// and the poses of the few candidates' geoms are read from global memory instead of staging all of them (ALOHA scene: 204 geoms = 9.8 of 16 KB
// per world for some 14 candidates)
__host__ __device__ inline int collide_lds_words(int ngeom, int npair, int concap, int sap = 0, bool pre = false) {
  if (pre) return 2 * collide_ccap(npair, concap) + CON_WINDOW * CON_LDS;
  const int base = 12 * ngeom + 2 * collide_ccap(npair, concap) + CON_WINDOW * CON_LDS;
  return base + (sap ? 3 * sap_pow2(ngeom) + ((npair + 31) / 32 + 3) / 4 * 4 : 0);
}

// ---- broadphase filters beyond plane / bounding sphere (collision_driver.py:124-275), HEAVY instantiation only ----------
// world-aligned boxes around the two rotated local boxes (_aabb_filter)
DEV bool aabb_filter(const float* a1, const float* a2, float margin, V3 x1, V3 x2, const float* R1, const float* R2) {
  float ctr[2][3], mx[2][3], mn[2][3];
  const float* as[2] = {a1, a2};
  const float* Rs[2] = {R1, R2};
  const V3 xs[2] = {x1, x2};
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const V3 c = mat_mul(Rs[g], ld3(as[g])) + xs[g];
    ctr[g][0] = c.x; ctr[g][1] = c.y; ctr[g][2] = c.z;
    const float sx = as[g][3], sy = as[g][4], sz = as[g][5];
#pragma unroll
    for (int k = 0; k < 3; ++k) {  // extreme of R (+-sx, +-sy, +-sz) along world axis k = sum of absolute terms
      const float e = fabsf(Rs[g][3 * k] * sx) + fabsf(Rs[g][3 * k + 1] * sy) + fabsf(Rs[g][3 * k + 2] * sz);
      mx[g][k] = e;
      mn[g][k] = -e;
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (ctr[0][k] + mx[0][k] + margin < ctr[1][k] + mn[1][k]) return false;
    if (ctr[1][k] + mx[1][k] + margin < ctr[0][k] + mn[0][k]) return false;
  }
  return true;
}
// separating-axis test on the six face normals of the two oriented boxes (_obb_filter, mj_collideOBB)
DEV bool obb_filter(const float* a1, const float* a2, float margin, V3 x1, V3 x2, const float* R1, const float* R2) {
  const V3 c1 = mat_mul(R1, ld3(a1)) + x1, c2 = mat_mul(R2, ld3(a2)) + x2;
  V3 nrm[6];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    nrm[a] = V3{R1[a], R1[3 + a], R1[6 + a]};
    nrm[3 + a] = V3{R2[a], R2[3 + a], R2[6 + a]};
  }
  const float* ss[2] = {a1 + 3, a2 + 3};
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    float radius = 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      radius += fabsf(ss[i][0] * dot(nrm[3 * i], nrm[j])) + fabsf(ss[i][1] * dot(nrm[3 * i + 1], nrm[j])) + fabsf(ss[i][2] * dot(nrm[3 * i + 2], nrm[j]));
    if (radius + margin < fabsf(dot(c2, nrm[j]) - dot(c1, nrm[j]))) return false;
  }
  return true;
}

// HEAVY: the instantiation that also carries the large colliders (capsule-box, box-box) and the explicit <contact><pair>
// parameter tables; models without them run the light one, whose register footprint (and with it the occupancy of k_mid: 125
// VGPRs, exactly four waves per SIMD) is unchanged.  (Measured: the pair-id lookups alone cost the light k_mid 20 us.)
// MODE 0: the contact kernel (broadphase + both narrowphase passes; for models with GJK pairs -- Data.ws_ccd -- the candidate list and the
// convex results come from the three launches in front of it).  MODE 1: k_ccd_broad -- the broadphase alone; the candidate list goes to
// the world's slice of Data.ws_ccd (k_ccd_gjk finds the convex candidates by their rank in it).
// HFT = false: the heavy instantiation without the height-field colliders (per-lane GJK / EPA on prisms: the largest register consumer left in
// the contact kernel) for models without height fields
template <int G, bool HEAVY = false, int MODE = 0, bool HFT = true>
DEV void collision_body(const MjhModel& m, const MjhData& d, float* smem, const Blk& b, int stride_words = 0) {
  constexpr bool HFC = HEAVY && HFT && MODE == 0;
  const int lig = threadIdx.x & (G - 1), gib = threadIdx.x / G;
  const int w = b.w0 + gib;
  // k_ccd_broad: the per-geom model tables the pair loop reads (bounding radius, margin, gap, local box) staged ONCE per workgroup in LDS
  // behind the worlds' slices (unless a field is batched per world): the loop's only global loads are then the coalesced pair ids -- with
  // the tables in L2 every trip was two dependent round trips at one wavefront per SIMD (ALOHA scene, 9,154 pairs: 785 us per launch)
  const float* tbl = nullptr;
  bool zero_mg = false;
  if constexpr (MODE == 1) {
    if (m.broadphase != 0 && m.geom_rbound_nb <= 1 && m.geom_margin_nb <= 1 && m.geom_gap_nb <= 1 && m.geom_aabb_nb <= 1) {  // (NXN: the filters ran in k_broad_mask)
      float* t = smem + (size_t)(blockDim.x / G) * broad_lds_words(m.ngeom, m.npair, d.concap, m.broadphase);
      const int n = m.ngeom;
      for (int i = threadIdx.x; i < n; i += blockDim.x) {
        t[i] = m.geom_rbound[i];
        t[n + i] = m.geom_margin[i];
        t[2 * n + i] = m.geom_gap[i];
      }
      for (int i = threadIdx.x; i < 6 * n; i += blockDim.x) t[3 * n + i] = m.geom_aabb[i];
      tbl = t;
      bool nz = false;
      for (int i = threadIdx.x; i < n; i += blockDim.x) nz |= m.geom_margin[i] != 0.0f || m.geom_gap[i] != 0.0f;
      zero_mg = !__syncthreads_or(nz);  // (also the barrier behind the staging) no margins, no gaps: the pair loop skips four table reads per pair
    }
  }
  if ((int)threadIdx.x >= b.nthreads || w >= d.nworld) return;
  // sleeping, second collision pass (forward.py:652-666): only worlds where a contact of pass 1 woke a tree can gain pairs
  if (HEAVY && d.sleep_pass == 2 && !d.ws_sleep_flag[w]) return;
  const int npair = m.npair, ng = m.ngeom, ncap = d.concap;
  // k_ccd_broad keeps only what its pair loop reads in LDS -- position | bounding radius of every geom as ONE 16-byte read, the candidate
  // list, the queue -- and reads the rotation matrices of the few pairs that reach the plane test / the box filters from global memory:
  // 9.5 instead of 19 KB per world on the ALOHA scene (12 instead of 8 worlds per CU)
  const bool pre = HEAVY && d.ws_ccd != nullptr;  // models with GJK pairs: k_ccd_broad / k_ccd_gjk / k_ccd_epa ran (or: this IS k_ccd_broad)
  const bool geoms_global = MODE == 1 || pre;     // geom poses from global memory (MODE 0 without the launches in front: staged in LDS)
  const int slice_words = MODE == 1 ? broad_lds_words(ng, npair, ncap, m.broadphase) : collide_lds_words(ng, npair, ncap, HEAVY ? m.broadphase : 0, pre);
  float* S = smem + (size_t)gib * (stride_words ? stride_words : slice_words);
  float* gx4 = S;  // (MODE 1 only)
  const float* gxpos = geoms_global ? d.geom_xpos + (size_t)w * 3 * m.ngeom : S;
  const float* gxmat = geoms_global ? d.geom_xmat + (size_t)w * 9 * m.ngeom : S + 3 * ng;
  int* cand = reinterpret_cast<int*>(S + (MODE == 1 ? (m.broadphase != 0 ? 4 * ng : 0) : (pre ? 0 : 12 * ng)));
  const int ccap = collide_ccap(npair, ncap);
  int* cslot = cand + ccap;
  float* rec = reinterpret_cast<float*>(cslot + ccap);

  if (m.disableflags & (DSBL_CONSTRAINT | DSBL_CONTACT)) {
    if (lig == 0 && MODE == 0) {
      d.ws_ncon[w] = 0;
      d.ws_ncollision[w] = 0;
    }
    return;
  }
  CcdLayout CL;
  float* ccd_world = nullptr;
  if (HEAVY && pre) {
    CL = ccd_layout_of(m, d);
    ccd_world = d.ws_ccd + (size_t)w * CL.world_stride;
  }
  int* gcand = ccd_world ? reinterpret_cast<int*>(ccd_world + CL.cand) : nullptr;  // ccap candidates | ncand, nbroad, nconvex
  PhaseClock pc(2, lig);
  if (!geoms_global) {
    gcopy<G>(S, d.geom_xpos + (size_t)w * 3 * ng, 3 * ng, lig);
    gcopy<G>(S + 3 * ng, d.geom_xmat + (size_t)w * 9 * ng, 9 * ng, lig);
  }
  const float* rbound = tbl ? tbl : bf(m.geom_rbound, m.geom_rbound_nb, w, ng);
  const float* gmargin = tbl ? tbl + ng : bf(m.geom_margin, m.geom_margin_nb, w, ng);
  const float* ggap = tbl ? tbl + 2 * ng : bf(m.geom_gap, m.geom_gap_nb, w, ng);
  const float* gsize = bf(m.geom_size, m.geom_size_nb, w, 3 * ng);
  gsync();
  if (MODE == 1 && m.broadphase != 0) {  // (the NXN tests ran in k_broad_mask)
    for (int g = lig; g < ng; g += G) {
      gx4[4 * g] = gxpos[3 * g];
      gx4[4 * g + 1] = gxpos[3 * g + 1];
      gx4[4 * g + 2] = gxpos[3 * g + 2];
      gx4[4 * g + 3] = rbound[g];
    }
    gsync();
  }
  pc.mark(0);

  // ---- broadphase (collision_driver.py:278-334 filters, 567-682 SAP, 684-770 NXN), ordered compaction --------------------
  // The candidates always leave in the canonical pair order, whatever produced them, so contacts do not depend on the
  // broadphase.  Plane / bounding-sphere filters live in both instantiations; AABB / OBB filters and the sweep-and-prune
  // candidate generator only in the HEAVY one (the light k_mid sits exactly at its 4-waves-per-SIMD register budget).
  const int filt = HEAVY ? m.broadphase_filter : 3;
  const float* gaabb = HEAVY ? (tbl ? tbl + 3 * ng : bf(m.geom_aabb, m.geom_aabb_nb, w, 6 * ng)) : nullptr;
  unsigned* sapmark = nullptr;
  int ncand = 0, nbroad = 0;
  if (HEAVY && MODE == 0 && pre) {  // k_ccd_broad's list (the convex results are keyed by the rank in exactly this list)
    nbroad = gcand[ccap + 1];
    ncand = gcand[ccap];
    for (int i = lig; i < ncand; i += G) cand[i] = gcand[i];
    gsync();
  } else {
  if (HEAVY && m.broadphase != 0) {
    // sweep and prune: project the bounding spheres on a fixed direction, sort by the lower bound (bitonic, in LDS), sweep
    const int np2 = sap_pow2(ng);
    float* plo = rec + CON_WINDOW * CON_LDS;
    float* phi = plo + np2;
    int* sidx = reinterpret_cast<int*>(phi + np2);
    sapmark = reinterpret_cast<unsigned*>(sidx + np2);
    const float dn = 1.0f / sqrtf(0.5935f * 0.5935f + 0.7790f * 0.7790f + 0.1235f * 0.1235f);
    const V3 dir = V3{0.5935f * dn, 0.7790f * dn, 0.1235f * dn};
    for (int g = lig; g < np2; g += G) {
      float lo = 3.0e38f, hi = 3.0e38f;  // padding sorts last
      if (g < ng) {
        float rb = rbound[g];
        if (rb == 0.0f) rb = MJ_MAXVAL;
        const float radius = rb + gmargin[g] + ggap[g], center = dot(dir, ld3(gxpos + 3 * g));
        const bool ok = center == center;
        lo = ok ? center - radius : MJ_MAXVAL;
        hi = ok ? center + radius : MJ_MAXVAL;
      }
      plo[g] = lo;
      phi[g] = hi;
      sidx[g] = g;
    }
    for (int i = lig; i < (npair + 31) / 32; i += G) sapmark[i] = 0u;
    gsync();
    for (int k = 2; k <= np2; k <<= 1)  // bitonic sort of (lower bound, geom id); ties broken by geom id: deterministic
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int t = lig; t < np2; t += G) {
          const int u = t ^ j;
          if (u > t) {
            const int a = sidx[t], b2 = sidx[u];
            const float la = plo[a < ng ? a : 0], lb = plo[b2 < ng ? b2 : 0];
            const float ka = a < ng ? la : 3.0e38f, kb = b2 < ng ? lb : 3.0e38f;
            const bool up = (t & k) == 0;
            const bool gt = ka > kb || (ka == kb && a > b2);
            if (gt == up) {
              sidx[t] = b2;
              sidx[u] = a;
            }
          }
        }
        gsync();
      }
    // sap_range (collision_core.py:501) + sweep: sorted element i against i+1 .. limit (the first element whose lower bound
    // exceeds i's upper bound, included as in the reference)
    for (int i = lig; i < ng; i += G) {
      const int gi = sidx[i];
      const float upper = phi[gi];
      int lo_i = i + 1, hi_i = ng;
      while (lo_i < hi_i) {
        const int mid = (lo_i + hi_i) >> 1;
        if (plo[sidx[mid]] > upper) hi_i = mid;
        else lo_i = mid + 1;
      }
      const int limit = hi_i < ng - 1 ? hi_i : ng - 1;
      for (int j = i + 1; j <= limit; ++j) {
        int g1 = gi, g2 = sidx[j];
        if (g2 < g1) {
          const int t = g1;
          g1 = g2;
          g2 = t;
        }
        const int p = m.nxn_pairindex[(g1 * (2 * ng - g1 - 3)) / 2 + g2 - 1];  // upper_tri_index math.py:323
        if (p >= 0) atomicOr(sapmark + (p >> 5), 1u << (p & 31));
      }
    }
    gsync();
  }
  // Pair loop in two stages (round 4).  Stage A: the cheap tests (plane distance / bounding spheres, the SAP mark, the sleep filter) on
  // every filtered pair, PU per lane and trip.  Stage B (heavy instantiation): the box filters on the SURVIVORS only, which queue up in
  // LDS in pair order and are served a full lane group at a time -- on the ALOHA scene 760 of 9,154 pairs survive stage A, scattered over
  // nearly every trip: with the box filters inside the trip (~300 instructions) every trip paid them for one or two live lanes.
  constexpr int PU = MODE == 1 ? (G > 32 ? 4 : 8) : (HEAVY ? 4 : 1);  // (64-lane groups: fewer pairs per lane and trip, the queue below is sized for G (PU + 1) entries)
  constexpr int QCAP = HEAVY ? CON_WINDOW * CON_LDS : 0;  // the staging window's LDS is free until the narrowphase's second pass
  static_assert(!HEAVY || QCAP >= (PU + 1) * G || CON_WINDOW < 16, "queue: a trip's survivors behind a partial group");  // (CON_WINDOW < 16: the round-5 occupancy experiment, light instantiation only)
  int* queue = reinterpret_cast<int*>(rec);
  int nq = 0;
  const bool box_filters = HEAVY && (filt & 12) != 0;
  auto stage_b = [&](bool all) __attribute__((always_inline)) {  // entries: pair | 1 << 30 if the pair skips the box filters (plane pairs)
    int qh = 0;  // the queue is consumed from its head; what stays behind (fewer than G entries) moves to the front at the end
    while (nq >= G || (all && nq > 0)) {
      const int n = nq < G ? nq : G;
      bool pass = false;
      int p = 0;
      if (lig < n) {
        const int e = queue[qh + lig];
        p = e & 0x3fffffff;
        pass = true;
        if (!(e >> 30)) {
          const int g1 = m.nxn_geom_pair[2 * p], g2 = m.nxn_geom_pair[2 * p + 1];
          const int pid = m.nexplicit ? m.nxn_pairid[p] : -1;
          const float mg = pid >= 0 ? m.pair_margin[pid] + m.pair_gap[pid] : gmargin[g1] + ggap[g1] + gmargin[g2] + ggap[g2];
          const V3 x1 = ld3(gxpos + 3 * g1), x2 = ld3(gxpos + 3 * g2);
          if (filt & 4) pass = aabb_filter(gaabb + 6 * g1, gaabb + 6 * g2, mg, x1, x2, gxmat + 9 * g1, gxmat + 9 * g2);
          if (pass && (filt & 8)) pass = obb_filter(gaabb + 6 * g1, gaabb + 6 * g2, mg, x1, x2, gxmat + 9 * g1, gxmat + 9 * g2);
        }
      }
      int tot;
      const int rank = grank<G>(pass, lig, tot);
      if (pass && ncand + rank < ccap) cand[ncand + rank] = p;
      ncand += tot;
      qh += n;
      nq -= n;
    }
    if (qh > 0 && nq > 0) {
      gsync();
      const int v = lig < nq ? queue[qh + lig] : 0;
      gsync();
      if (lig < nq) queue[lig] = v;
    }
    gsync();
  };
#ifdef MJH_DBG_BROAD_SKIP  // (profiling variant, tools/build_variant_fast.py: the launch without its pair loop)
  if (MODE == 1) nq = 0; else
#endif
  // (the pair ids of the NEXT trip are loaded before this trip's tests: at one wavefront per SIMD the table's L2 round trip -- 73 KB read by
  // every world -- was the chain: 72 trips x ~2.5 us)
  const int2* pairs2 = reinterpret_cast<const int2*>(m.nxn_geom_pair);
  int2 cur[PU];
#pragma unroll
  for (int u = 0; u < PU; ++u) cur[u] = (u * G + lig < npair) ? pairs2[u * G + lig] : make_int2(0, 0);
  for (int base = 0; base < npair; base += PU * G) {
    bool passu[PU];
    bool plainu[PU];
    int2 nxt[PU];
#pragma unroll
    for (int u = 0; u < PU; ++u) {
      const int pn = base + PU * G + u * G + lig;
      nxt[u] = pn < npair ? pairs2[pn] : make_int2(0, 0);
    }
#pragma unroll
    for (int u = 0; u < PU; ++u) {
    const int p = base + u * G + lig;
    bool pass = false, plain = false;
    if (p < npair && (!sapmark || ((sapmark[p >> 5] >> (p & 31)) & 1u))) {
      const int2 gg = cur[u];
      const int g1 = gg.x, g2 = gg.y;
      float rb1, rb2;
      V3 x1, x2;
      if constexpr (MODE == 1) {
        const float4 a = *reinterpret_cast<const float4*>(gx4 + 4 * g1), c = *reinterpret_cast<const float4*>(gx4 + 4 * g2);
        x1 = V3{a.x, a.y, a.z};
        x2 = V3{c.x, c.y, c.z};
        rb1 = a.w;
        rb2 = c.w;
      } else {
        rb1 = rbound[g1];
        rb2 = rbound[g2];
        x1 = ld3(gxpos + 3 * g1);
        x2 = ld3(gxpos + 3 * g2);
      }
      const int pid = (HEAVY && m.nexplicit) ? m.nxn_pairid[p] : -1;
      const float mg = pid >= 0 ? m.pair_margin[pid] + m.pair_gap[pid] : (zero_mg ? 0.0f : gmargin[g1] + ggap[g1] + gmargin[g2] + ggap[g2]);
      if (rb1 == 0.0f || rb2 == 0.0f) {
        plain = true;
        if (!(filt & 1)) {
          pass = true;
        } else if (rb1 == 0.0f) {
          const float* R = gxmat + 9 * g1;
          pass = dot(x2 - x1, V3{R[2], R[5], R[8]}) <= rb2 + mg;
        } else {
          const float* R = gxmat + 9 * g2;
          pass = dot(x1 - x2, V3{R[2], R[5], R[8]}) <= rb1 + mg;
        }
      } else {
        pass = true;
        if (filt & 2) {
          const float bound = rb1 + rb2 + mg;
          V3 dif = x2 - x1;
          pass = dot(dif, dif) <= bound * bound;
        }
      }
      if (HEAVY && m.sleep_enabled && pass) {  // collision_driver.py:494-503: no pair of two sleeping bodies, or of a sleeping and a static one
        const int* bawake = d.body_awake + (size_t)w * m.nbody;
        const int s1 = bawake[m.geom_bodyid[g1]], s2 = bawake[m.geom_bodyid[g2]];
        if ((s1 == 0 && s2 == 0) || (s1 == 0 && s2 == -1) || (s2 == 0 && s1 == -1)) pass = false;
      }
    }
    passu[u] = pass;
    plainu[u] = plain;
    }
#pragma unroll
    for (int u = 0; u < PU; ++u) {
      int tot;
      const int rank = grank<G>(passu[u], lig, tot);
      if (!box_filters) {  // no stage B: the survivors are the candidates
        if (passu[u] && ncand + rank < ccap) cand[ncand + rank] = base + u * G + lig;
        ncand += tot;
      } else {
        if (passu[u]) queue[nq + rank] = (base + u * G + lig) | (plainu[u] ? (1 << 30) : 0);
        nq += tot;
      }
    }
    if (box_filters) {
      gsync();
      stage_b(false);
    }
#pragma unroll
    for (int u = 0; u < PU; ++u) cur[u] = nxt[u];
  }
  if (box_filters) stage_b(true);
  nbroad = ncand;  // candidates found (Data.ncollision); the capacity bounds what the narrowphase sees
  if (ncand > ccap) ncand = ccap;
  gsync();
  }
  pc.mark(1);
  if constexpr (MODE == 1) {
    // ---- k_ccd_broad: publish the candidate list; a convex candidate's slot = its rank among the world's convex candidates (its cache
    // entry starts as "no contact" and carries the pair id for k_ccd_gjk, which walks (slot, world))
    int* cnt = reinterpret_cast<int*>(d.ws_ccd + CL.cnt);
    int ncvx = 0;
    for (int base = 0; base < ncand; base += G) {
      const int ci = base + lig;
      bool cvx = false;
      int p = 0;
      if (ci < ncand) {
        p = cand[ci];
        gcand[ci] = p;
        const int t1 = m.geom_type[m.nxn_geom_pair[2 * p]], t2 = m.geom_type[m.nxn_geom_pair[2 * p + 1]];
        cvx = is_ccd_pair(m, min(t1, t2), max(t1, t2));
      }
      int tot;
      const int rank = grank<G>(cvx, lig, tot);
      if (cvx) {
        int* ce = reinterpret_cast<int*>(ccd_world + CL.cache + (size_t)(ncvx + rank) * CCD_CACHE_WORDS);
        ce[0] = 0;
        ce[CCD_CACHE_WORDS - 1] = p;  // (the entry's spare word)
      }
      ncvx += tot;
    }
    if (lig == 0) {
      gcand[ccap] = ncand;
      gcand[ccap + 1] = nbroad;
      gcand[ccap + 2] = ncvx;
      // the longest convex list of any world (one atomic per world on one address costs ~50 us at 8192 worlds: only the worlds that raise
      // the maximum issue it -- a stale read only means an atomic that changes nothing)
      if (ncvx > __atomic_load_n(cnt, __ATOMIC_RELAXED)) atomicMax(cnt, ncvx);
    }
    return;
  }

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
  auto mesh_of = [&](int g, int t, const float*& vert, int& nvert) {  // mesh geoms: their vertices in the geom frame
    vert = nullptr;
    nvert = 0;
    if (HEAVY && t == G_MESH) {
      const int id = m.geom_dataid[g];
      vert = m.mesh_vert + 3 * m.mesh_vertadr[id];
      nvert = m.mesh_vertnum[id];
    }
  };
  auto graph_of_geom = [&](int g, int t) -> const int* {
    if (!HEAVY || t != G_MESH) return nullptr;
    const int adr = m.mesh_graphadr[m.geom_dataid[g]];
    return adr >= 0 ? m.mesh_graph + adr : nullptr;
  };
  // EPA polytope of this lane (convex.hpp): word k of lane l at k * 32 + l inside the world's slice of d.ws_ccd
  float* ccd_scratch = (HFC && ccd_world && m.nhfield > 0) ? ccd_world + CL.hf + (lig & (CCD_LANES - 1)) : nullptr;  // height-field prisms only
  const float* ccd_cache = ccd_world ? ccd_world + CL.cache : nullptr;  // results of k_ccd_gjk / k_ccd_epa, one entry per convex candidate
  const int hf0 = ccd_words(max(m.ccd_iterations, m.epa_iterations), 0);  // first word of the lane's height-field result table
  const float ccd_tol = HEAVY ? bf(m.opt_ccd_tolerance, m.opt_ccd_tolerance_nb, w, 1)[0] : 0.0f;
  const int ccd_it = min(m.ccd_iterations, CCD_MAX_ITER), epa_it = min(m.epa_iterations, CCD_MAX_ITER);
  const int ccd_cache0 = ccd_poly_words(max(m.ccd_iterations, m.epa_iterations));  // first word of the lane's contact cache
  const bool box_ccd = !(m.disableflags & DSBL_NATIVECCD);  // box-box: CCD + multi-contact unless the flag asks for mjc_BoxBox
  int ccd_overflow = 0;
  // height-field candidates: the lanes whose candidate is one (`want`) are served in turn by the whole group (hfield_fill)
  auto hf_pair = [&](int p) {
    int g1, g2, t1, t2;
    load_pair(p, g1, g2, t1, t2);
    return t1 == G_HFIELD && t2 >= G_SPHERE;
  };
  int hf_have = -1;  // the candidate whose results this lane's table holds (pass 2 refills it only if a later candidate replaced them)
  auto hf_coop = [&](bool want, int mycand, int myci) {
    unsigned long long bits = gballot<G>(want);
    while (bits) {
      const int owner = __ffsll((long long)bits) - 1;
      bits &= bits - 1;
      int g1, g2, t1, t2;
      load_pair(__shfl(mycand, owner, G), g1, g2, t1, t2);
      const int pid = m.nexplicit ? m.nxn_pairid[__shfl(mycand, owner, G)] : -1;
      const float margin = pid >= 0 ? m.pair_margin[pid] : gmargin[g1] + gmargin[g2];
      const float* mv2;
      int mn2;
      mesh_of(g2, t2, mv2, mn2);
      float* table = ccd_scratch - (lig & (CCD_LANES - 1)) + (owner & (CCD_LANES - 1)) + (size_t)hf0 * CCD_LANES;
      if constexpr (HFC) {
        hfield_fill<G>(m, ccd_tol, ccd_it, epa_it, g1, t2, ld3(gxpos + 3 * g1), gxmat + 9 * g1, ld3(gxpos + 3 * g2), gxmat + 9 * g2, ld3(gsize + 3 * g2), rbound[g2],
                       gmargin[g1] + gmargin[g2], margin, ccd_scratch, table, ccd_overflow, lig, lig == owner, mv2, mn2, t2 == G_MESH ? m.geom_dataid[g2] : -1);
      }
      if (lig == owner) hf_have = myci;
    }
  };
  int ncon = 0;
  int ncvx = 0;  // convex candidates seen so far: the slot of the next one's cached result
  for (int base = 0; base < ncand; base += G) {
    const int ci = base + lig;
    // which of the collider's (at most 8) contacts pass the margin test: pass 2 replays this mask instead of
    // re-testing, so the two passes agree even if the compiler contracts the distance arithmetic differently
    unsigned mask = 0u;
    bool cvx = false;
    float cvx_lim = 0.0f;
    if (HFC && m.nhfield > 0 && ccd_scratch) hf_coop(ci < ncand && hf_pair(cand[ci]), ci < ncand ? cand[ci] : 0, ci);
    if (ci < ncand) {
      int g1, g2, t1, t2;
      load_pair(cand[ci], g1, g2, t1, t2);
      const int pid = (HEAVY && m.nexplicit) ? m.nxn_pairid[cand[ci]] : -1;
      const float margin = pid >= 0 ? m.pair_margin[pid] : gmargin[g1] + gmargin[g2];
      const float gap = pid >= 0 ? m.pair_gap[pid] : ggap[g1] + ggap[g2];
      const float lim = margin + gap;
      cvx_lim = lim;
      auto count = [&](int k, float dist, V3, V3, V3, V3) { mask |= dist < lim ? (1u << (k & 7)) : 0u; };
      const float *mv1, *mv2;
      int mn1, mn2;
      mesh_of(g1, t1, mv1, mn1);
      mesh_of(g2, t2, mv2, mn2);
      if (HEAVY && t1 == G_HFIELD) {  // (the group filled this lane's table just above; pass 2 reads it again)
        if constexpr (HFC) if (t2 >= G_SPHERE && ccd_scratch) hfield_select(ccd_scratch + (size_t)hf0 * CCD_LANES, count);
      } else if (HEAVY && is_ccd_pair(m, t1, t2)) {
        cvx = true;  // (its result was computed by k_ccd_gjk / k_ccd_epa: read below, once the lane knows its slot)
      }
      else
        collide_pair<HEAVY>(t1, t2, ld3(gxpos + 3 * g1), gxmat + 9 * g1, ld3(gsize + 3 * g1), ld3(gxpos + 3 * g2), gxmat + 9 * g2,
                            ld3(gsize + 3 * g2), margin, count, mv2, mn2, graph_of_geom(g2, t2));
    }
    if constexpr (HEAVY) if (ccd_cache) {
      int tot;
      const int slot = ncvx + grank<G>(cvx, lig, tot);
      ncvx += tot;
      if (cvx) {
        const float* cache = ccd_cache + (size_t)slot * CCD_CACHE_WORDS;
        const int nem = reinterpret_cast<const int*>(cache)[0];
        mask = (nem > 0 && cache[1] < cvx_lim) ? (1u << nem) - 1u : 0u;  // (the contacts of a pair share one distance)
      }
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
    int ncvx2 = 0;
    for (int base = 0; base < ncand; base += G) {
      const int ci = base + lig;
      int myslot = -1;  // convex candidates: the slot of the cached result
      if constexpr (HEAVY) if (ccd_cache) {
        bool cvx = false;
        if (ci < ncand) {
          int g1, g2, t1, t2;
          load_pair(cand[ci], g1, g2, t1, t2);
          cvx = is_ccd_pair(m, t1, t2);
        }
        int tot;
        const int rank = grank<G>(cvx, lig, tot);
        if (cvx) myslot = ncvx2 + rank;
        ncvx2 += tot;
      }
      if (HFC && m.nhfield > 0 && ccd_scratch) {  // height-field candidates with contacts in this window: the group refills their tables
        bool want = false;
        if (ci < ncand && hf_pair(cand[ci])) {
          const unsigned mk = (unsigned)cslot[ci] & 0xffu;
          const int s0 = cslot[ci] >> 8;
          want = mk != 0u && s0 < wend && s0 + __popc(mk) > wbase && hf_have != ci;
        }
        hf_coop(want, ci < ncand ? cand[ci] : 0, ci);
      }
      if (ci >= ncand) continue;
      const unsigned mask = (unsigned)cslot[ci] & 0xffu;
      int slot = cslot[ci] >> 8;
      const int send = slot + __popc(mask);
      if (mask == 0u || slot >= wend || send <= wbase) continue;
      int g1, g2, t1, t2;
      load_pair(cand[ci], g1, g2, t1, t2);
      const int pid = (HEAVY && m.nexplicit) ? m.nxn_pairid[cand[ci]] : -1;
      const PairParams pp = contact_params(m, w, g1, g2, pid);
      const float margin = pp.margin;
      auto write = [&](int cid, float dist, V3 pos, V3 fa, V3 fb, V3 fc) {
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
                       ri[27] = cid | ((pid + 1) << 8);
                       r[30] = pp.friction[1];
                       r[31] = pp.friction[4];
                     }
                     ++slot;
                   };
      if (HEAVY && t1 == G_HFIELD) {
        if constexpr (HFC) if (t2 >= G_SPHERE && ccd_scratch) hfield_select(ccd_scratch + (size_t)hf0 * CCD_LANES, write);
      } else if (HEAVY && is_ccd_pair(m, t1, t2)) {
        if (myslot >= 0 && ccd_cache) {  // replay the contacts k_ccd_gjk / k_ccd_epa found
          const float* cache = ccd_cache + (size_t)myslot * CCD_CACHE_WORDS;
          const int nem = reinterpret_cast<const int*>(cache)[0];
          for (int k = 0; k < nem; ++k) write(k, cache[1], ld3(cache + 11 + 3 * k), ld3(cache + 2), ld3(cache + 5), ld3(cache + 8));
        }
      }
      else {
        const float* mv2;
        int mn2;
        mesh_of(g2, t2, mv2, mn2);
        collide_pair<HEAVY>(t1, t2, ld3(gxpos + 3 * g1), gxmat + 9 * g1, ld3(gsize + 3 * g1), ld3(gxpos + 3 * g2), gxmat + 9 * g2,
                            ld3(gsize + 3 * g2), margin, write, mv2, mn2, graph_of_geom(g2, t2));
      }
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
    d.ws_ncollision[w] = nbroad;
    if (ncon < nfound) atomicOr(d.overflow + w, OVF_NARROWPHASE);
    if (ncand < nbroad) atomicOr(d.overflow + w, OVF_BROADPHASE);
  }
  if (HEAVY && ccd_overflow) atomicOr(d.overflow + w, ccd_overflow);
}

// ---- the three launches of the convex narrowphase (convex.hpp header) -------------------------------------------------------------------
// ---- k_broad_mask: the broadphase FILTERS of one world per workgroup, results as a bit mask over the pair list (rounds 4 and 5) ---------
// k_ccd_broad used to run the filters inside its per-world lane group: 32 lanes walking the world's whole pair list (ALOHA scene: 9,154
// pairs) with the candidate list and a queue in LDS -- at most 1.5 wavefronts per SIMD, 480 of the launch's 505 us.  The tests of different
// pairs do not depend on each other and their results are ordered by the pair index alone, so they left that kernel (round 4: every pair
// tested, 64 per wavefront trip: 201 us).  Round 5 tests GROUPS of geoms first:
//   * the host regroups the pair list by pairs of geom groups (io.py cull_tables: the colliding geoms among the consecutive geoms of one
//     moving body; a static geom is its own group) into ROWS of at most 16 pairs;
//   * a workgroup (four wavefronts) takes one world; it stages, per geom, position | bounding radius, margin, gap, with the AABB filter on the
//     world-aligned box (centre +- extent, the pair-independent half of _aabb_filter), with the OBB filter on rotation and local box;
//   * a lane per colliding geom measures its group's bounding sphere around the group's centre geom (LDS atomicMax on the radius bits; the
//     radius carries a 1e-4 relative slack for the float32 sums; a group holding a plane -- rbound 0: the plane filter decides -- is open, as
//     is every group when the sphere filter is off; rows of explicit pairs are always tested);
//   * a lane per row tests the two group spheres; surviving rows are compacted (BMASK_ROWCHUNK rows at a time);
//   * a quarter wavefront per surviving row, four rows in flight, runs the pair filters: plane / bounding sphere, sleep state, the box
//     overlap as six compares; passing pairs set their bit of the world's mask (LDS atomicOr); the OBB filter (separating axes, ~250
//     instructions) runs on the pairs that got that far, queued per wavefront and served 64 at a time;
//   * the mask is expanded in pair order by the whole workgroup (prefix sums over its four wavefronts): the candidate list is the serial
//     loop's, and the launch publishes it like k_ccd_broad does for the sweep-and-prune broadphase.
// A pair whose spheres overlap lies in two groups whose spheres overlap (triangle inequality), so the mask is the one every pair's test gives.
// ALOHA scene: 201 -> 160 us (staging 30, group spheres 4, rows 18, pair filters 76 -- two thirds of it the OBB arithmetic on the ~900 pairs
// that pass the sphere test --, expansion 30).
// Filters: collision_driver.py:124-275 (_aabb_filter, _obb_filter), 278-334 (_plane_filter, _sphere_filter), 494-503 (sleep).
// rows of cull_pair a workgroup takes at a time (the survivors' list in LDS is this long at most: a scene of hundreds of one-geom bodies has
// as many rows as pairs)
#define BMASK_ROWCHUNK 4096
struct BmaskLayout {  // word offsets of a workgroup's LDS (the box tables only with their filters on)
  int gx4, hl, rl, ab, gm, gp, fm, queue, gc4, surv, pre, cnt, cgl, total;
};
__host__ __device__ inline BmaskLayout bmask_layout(int ngeom, int npair, int filt, int ncullgeom, int ngroup, int ncullpair) {
  BmaskLayout L;
  int o = 0;
  L.gx4 = o; o += 4 * ngeom;                            // x y z rbound
  L.hl = o; o += (filt & 4) ? 8 * ngeom : 0;            // hi.x hi.y hi.z lo.x | lo.y lo.z - -   (world-aligned box of the geom's local box)
  L.rl = o; o += (filt & 8) ? 15 * ngeom : 0;           // rotation matrix (9 words) | local box (6 words): what the OBB filter reads
  L.ab = L.rl + 9;
  o = (o + 3) & ~3;
  L.gm = o; o += ngeom;                                 // margin
  L.gp = o; o += ngeom;                                 // gap
  L.fm = o; o += (2 * ((npair + 63) / 64) + 1) & ~1;    // the world's mask over the pair list
  o = (o + 1) & ~1;
  L.queue = o; o += 4 * 256;                            // a wavefront's OBB queue: 128 x (pair, g1 | g2 << 16)
  o = (o + 3) & ~3;
  L.gc4 = o; o += ngroup;                               // group radii (float bits: atomicMax target; +inf: open)
  L.surv = o; o += ncullpair < BMASK_ROWCHUNK ? ncullpair : BMASK_ROWCHUNK;  // surviving rows of cull_pair (of one chunk of rows): first entry of cull_list | count << 24
  L.pre = o; o += 16;                                   // per-wavefront totals of the expansion's prefix sums
  L.cnt = o; o += 2;                                    // survivors
  L.cgl = o;
  L.total = o + 4;
  return L;
}
__global__ void __launch_bounds__(256) k_broad_mask(MjhModel m, MjhData d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (m.disableflags & (DSBL_CONSTRAINT | DSBL_CONTACT)) return;
  const int ng = m.ngeom, npair = m.npair, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int ngran = (npair + 63) / 64;
  const CcdLayout CL = ccd_layout_of(m, d);
  const int filt = m.broadphase_filter;
  const bool use_aabb = (filt & 4) != 0, use_obb = (filt & 8) != 0;
  const int ngrp = m.ncullgroup, ncp = m.ncullpair;
  const BmaskLayout L = bmask_layout(ng, npair, filt, m.ncullgeom, ngrp, ncp);
  float* gx4 = smem + L.gx4;
  float* hl = smem + L.hl;
  float* rl = smem + L.rl;
  float* ab = smem + L.ab;
  float* gm = smem + L.gm;
  float* gp = smem + L.gp;
  unsigned* fm = reinterpret_cast<unsigned*>(smem + L.fm);
  int2* queue = reinterpret_cast<int2*>(smem + L.queue) + wv * 128;  // this wavefront's OBB queue
  int* gw = reinterpret_cast<int*>(smem + L.gc4);
  int* surv = reinterpret_cast<int*>(smem + L.surv);
  int* wtot = reinterpret_cast<int*>(smem + L.pre);
  int* nsurv_p = reinterpret_cast<int*>(smem + L.cnt);
  const int2* cgeom = reinterpret_cast<const int2*>(m.cull_geom);
  const int4* cpair = reinterpret_cast<const int4*>(m.cull_pair);
  const int2* clist = reinterpret_cast<const int2*>(m.cull_list);
  const unsigned* cmask = reinterpret_cast<const unsigned*>(d.ws_ccd + CL.cmask);
  const int inf_bits = 0x7f800000;
  for (int w = blockIdx.x; w < d.nworld; w += gridDim.x) {
    if (d.sleep_pass == 2 && !d.ws_sleep_flag[w]) continue;  // (uniform over the workgroup; k_ccd_broad skips these worlds too)
    const float* gxpos = d.geom_xpos + (size_t)w * 3 * ng;
    const float* gxmat = d.geom_xmat + (size_t)w * 9 * ng;
    const float* gaabb = bf(m.geom_aabb, m.geom_aabb_nb, w, 6 * ng);
    __syncthreads();  // the previous world's readers are done
    bool nz = false;  // any margin or gap in this world's tables
    {
      const float* rb = bf(m.geom_rbound, m.geom_rbound_nb, w, ng);
      const float* mg = bf(m.geom_margin, m.geom_margin_nb, w, ng);
      const float* ga = bf(m.geom_gap, m.geom_gap_nb, w, ng);
      for (int g = tid; g < ng; g += 256) {
        const V3 x = ld3(gxpos + 3 * g);
        *reinterpret_cast<float4*>(gx4 + 4 * g) = make_float4(x.x, x.y, x.z, rb[g]);
        gm[g] = mg[g];
        gp[g] = ga[g];
        nz |= mg[g] != 0.0f || ga[g] != 0.0f;
        if (use_aabb || use_obb) {
          const float* R = gxmat + 9 * g;
          const float* a = gaabb + 6 * g;
          if (use_obb) {
#pragma unroll
            for (int k = 0; k < 9; ++k) rl[15 * g + k] = R[k];
#pragma unroll
            for (int k = 0; k < 6; ++k) ab[15 * g + k] = a[k];
          }
          if (use_aabb) {  // as aabb_filter: centre R a + x, extent along world axis k = sum of the absolute terms
            const V3 c = mat_mul(R, ld3(a)) + x;
            const float sx = a[3], sy = a[4], sz = a[5];
            const float ex = fabsf(R[0] * sx) + fabsf(R[1] * sy) + fabsf(R[2] * sz);
            const float ey = fabsf(R[3] * sx) + fabsf(R[4] * sy) + fabsf(R[5] * sz);
            const float ez = fabsf(R[6] * sx) + fabsf(R[7] * sy) + fabsf(R[8] * sz);
            *reinterpret_cast<float4*>(hl + 8 * g) = make_float4(c.x + ex, c.y + ey, c.z + ez, c.x + -ex);
            *reinterpret_cast<float2*>(hl + 8 * g + 4) = make_float2(c.y + -ey, c.z + -ez);
          }
        }
      }
      for (int i = tid; i < 2 * ngran; i += 256) fm[i] = 0u;
      for (int i = tid; i < ngrp; i += 256) gw[i] = (filt & 2) ? 0 : inf_bits;  // (sphere filter off: every group open)
      if (tid < 2) nsurv_p[tid] = 0;
    }
    const bool zero_mg = !__syncthreads_or(nz);  // (also the barrier behind the staging) no margins, no gaps: four table reads per pair less
    const int* bawake = m.sleep_enabled ? d.body_awake + (size_t)w * m.nbody : nullptr;
    // the groups' bounding spheres: centred on the group's centre geom (the host's choice), radius to the farthest of its colliding geoms --
    // a lane per geom, the maximum through an LDS atomic on the float's bits (radii are >= 0)
    for (int j = tid; j < m.ncullgeom; j += 256) {
      const int2 e = cgeom[j];
      const int k = e.x & 0xffff, g = (e.x >> 16) & 0xffff;
      const float4 a = *reinterpret_cast<const float4*>(gx4 + 4 * k), c = *reinterpret_cast<const float4*>(gx4 + 4 * e.y);
      const V3 dc = V3{a.x - c.x, a.y - c.y, a.z - c.z};
      const float R = (sqrtf(dot(dc, dc)) + a.w + (zero_mg ? 0.0f : gm[k] + gp[k])) * 1.0001f + 1e-6f;
      atomicMax(gw + g, a.w == 0.0f ? inf_bits : __float_as_int(fmaxf(R, 0.0f)));
    }
    __syncthreads();
    int nq = 0;
    auto obb_round = [&](int n) __attribute__((always_inline)) {  // the first n queue entries (n <= 64)
      if (lane < n) {
        const int2 e = queue[lane];
        const int p = e.x, g1 = e.y & 0xffff, g2 = (e.y >> 16) & 0xffff;
        const int pid = m.nexplicit ? m.nxn_pairid[p] : -1;
        const float mgn = pid >= 0 ? m.pair_margin[pid] + m.pair_gap[pid] : (zero_mg ? 0.0f : gm[g1] + gp[g1] + gm[g2] + gp[g2]);
        const float4 a = *reinterpret_cast<const float4*>(gx4 + 4 * g1), b = *reinterpret_cast<const float4*>(gx4 + 4 * g2);
        if (obb_filter(ab + 15 * g1, ab + 15 * g2, mgn, V3{a.x, a.y, a.z}, V3{b.x, b.y, b.z}, rl + 15 * g1, rl + 15 * g2))
          atomicOr(fm + (p >> 5), 1u << (p & 31));
      }
    };
    // the rows of cull_pair whose group spheres overlap (or are open), in any order: the mask orders the result
    for (int r0 = 0; r0 < ncp; r0 += BMASK_ROWCHUNK) {  // (one pass for up to BMASK_ROWCHUNK rows)
    const int r1 = min(ncp, r0 + BMASK_ROWCHUNK);
    if (r0 > 0) {  // the previous chunk's list was read to its end
      __syncthreads();
      if (tid == 0) nsurv_p[0] = 0;
      __syncthreads();
    }
    for (int base = r0; base < r1; base += 1024) {  // (four trips' table loads in flight together)
      int4 e4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = base + 256 * j + tid;
        e4[j] = k < r1 ? cpair[k] : make_int4(-2, 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int4 e = e4[j];
        bool keep = e.x != -2;
        if (e.x >= 0) {
          const float4 a = *reinterpret_cast<const float4*>(gx4 + 4 * (e.z & 0xffff)), b = *reinterpret_cast<const float4*>(gx4 + 4 * ((e.z >> 16) & 0xffff));
          const float bound = __int_as_float(gw[e.x]) + __int_as_float(gw[e.y]);
          const V3 dif = V3{b.x - a.x, b.y - a.y, b.z - a.z};
          keep = !(dot(dif, dif) > bound * bound);
        }
        const unsigned long long bb = __ballot(keep);
        int at = 0;
        if (lane == 0 && bb) at = atomicAdd(nsurv_p, __popcll(bb));
        at = __shfl(at, 0, 64);
        if (keep) surv[at + __popcll(bb & ((1ull << lane) - 1ull))] = e.w;
      }
    }
    __syncthreads();
    const int nsurv = nsurv_p[0];
    // the pairs of the surviving rows (at most 16 entries each): a 16-lane quarter wavefront takes four rows per trip, their entries loaded
    // together; the loops are uniform over the workgroup
    const int sg = tid >> 4, sl = tid & 15;
    for (int s0 = 0; s0 < nsurv; s0 += 64) {
      int2 c4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int idx = s0 + 16 * j + sg;
        const int sv = idx < nsurv ? surv[idx] : 0;
        c4[j] = sl < ((sv >> 24) & 0xff) ? clist[(sv & 0xffffff) + sl] : make_int2(-1, 0);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
      const int2 cur = c4[j];
      bool pass = false, plain = false;
      const int p = cur.x, g1 = cur.y & 0xffff, g2 = (cur.y >> 16) & 0xffff;
      if (p >= 0) {
        const float4 a = *reinterpret_cast<const float4*>(gx4 + 4 * g1), b = *reinterpret_cast<const float4*>(gx4 + 4 * g2);
        const V3 x1 = V3{a.x, a.y, a.z}, x2 = V3{b.x, b.y, b.z};
        const float rb1 = a.w, rb2 = b.w;
        const int pid = m.nexplicit ? m.nxn_pairid[p] : -1;
        const float mgn = pid >= 0 ? m.pair_margin[pid] + m.pair_gap[pid] : (zero_mg ? 0.0f : gm[g1] + gp[g1] + gm[g2] + gp[g2]);
        if (rb1 == 0.0f || rb2 == 0.0f) {
          plain = true;
          if (!(filt & 1)) {
            pass = true;
          } else if (rb1 == 0.0f) {
            const float* R = gxmat + 9 * g1;
            pass = dot(x2 - x1, V3{R[2], R[5], R[8]}) <= rb2 + mgn;
          } else {
            const float* R = gxmat + 9 * g2;
            pass = dot(x1 - x2, V3{R[2], R[5], R[8]}) <= rb1 + mgn;
          }
        } else {
          pass = true;
          if (filt & 2) {
            const float bound = rb1 + rb2 + mgn;
            const V3 dif = x2 - x1;
            pass = dot(dif, dif) <= bound * bound;
          }
        }
        if (bawake && pass) {  // collision_driver.py:494-503: no pair of two sleeping bodies, or of a sleeping and a static one
          const int s1 = bawake[m.geom_bodyid[g1]], s2 = bawake[m.geom_bodyid[g2]];
          if ((s1 == 0 && s2 == 0) || (s1 == 0 && s2 == -1) || (s2 == 0 && s1 == -1)) pass = false;
        }
        if (use_aabb && pass && !plain) {
          const float4 h1 = *reinterpret_cast<const float4*>(hl + 8 * g1), h2 = *reinterpret_cast<const float4*>(hl + 8 * g2);
          const float2 l1 = *reinterpret_cast<const float2*>(hl + 8 * g1 + 4), l2 = *reinterpret_cast<const float2*>(hl + 8 * g2 + 4);
          if (h1.x + mgn < h2.w || h2.x + mgn < h1.w || h1.y + mgn < l2.x || h2.y + mgn < l1.x || h1.z + mgn < l2.y || h2.z + mgn < l1.y) pass = false;
        }
      }
      const bool need = use_obb && pass && !plain;  // still to pass the OBB filter
      if (pass && !need) atomicOr(fm + (p >> 5), 1u << (p & 31));
      if (use_obb) {
        const unsigned long long bn = __ballot(need);
        if (need) queue[nq + __popcll(bn & ((1ull << lane) - 1ull))] = make_int2(p, g1 | (g2 << 16));
        nq += __popcll(bn);
        gsync();
        if (nq >= 64) {
          obb_round(64);
          gsync();
          const int2 v = (64 + lane < nq) ? queue[64 + lane] : make_int2(0, 0);
          gsync();
          if (64 + lane < nq) queue[lane] = v;
          nq -= 64;
          gsync();
        }
      }
      }
    }
    }
    if (use_obb && nq > 0) obb_round(nq);
    __syncthreads();
    // the mask expanded in pair order = the world's candidate list (what k_ccd_broad publishes for the sweep-and-prune broadphase):
    // candidates, counts, and per convex candidate a cache entry "no contact" that carries the pair id.  A lane per mask word; a candidate's
    // position and a convex candidate's slot are prefix sums over the workgroup (wavefront scans + the wavefronts' totals through LDS)
    {
      float* ccd_world = d.ws_ccd + (size_t)w * CL.world_stride;
      int* gcand = reinterpret_cast<int*>(ccd_world + CL.cand);
      int* cnt = reinterpret_cast<int*>(d.ws_ccd + CL.cnt);
      const int ccap = CL.ccap;
      int ncand = 0, ncvx = 0;  // (uniform over the workgroup)
      for (int base = 0; base < 2 * ngran; base += 256) {
        const int wi = base + tid;
        const unsigned bits = wi < 2 * ngran ? fm[wi] : 0u;
        // convex pairs among them (k_ccd_reset's mask)
        const unsigned cbits = wi < 2 * ngran ? bits & cmask[wi] : 0u;
        const int c1 = __popc(bits), c2 = __popc(cbits);
        int o1 = c1, o2 = c2;  // inclusive prefixes over the wavefront
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
          const int u1 = __shfl_up(o1, sft, 64), u2 = __shfl_up(o2, sft, 64);
          if (lane >= sft) {
            o1 += u1;
            o2 += u2;
          }
        }
        if (lane == 63) {
          wtot[2 * wv] = o1;
          wtot[2 * wv + 1] = o2;
        }
        __syncthreads();
        int a = ncand + o1 - c1, c = ncvx + o2 - c2, t1 = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (k < wv) {
            a += wtot[2 * k];
            c += wtot[2 * k + 1];
          }
          t1 += wtot[2 * k];
        }
        int kc = 0;  // convex candidates of this lane inside the capacity
        for (unsigned b = bits; b; b &= b - 1) {
          const int bit = __ffs(b) - 1, p = 32 * wi + bit;
          const bool cv = (cbits >> bit) & 1u;
          if (a < ccap) {
            gcand[a] = p;
            if (cv) {
              int* ce = reinterpret_cast<int*>(ccd_world + CL.cache + (size_t)c * CCD_CACHE_WORDS);
              ce[0] = 0;
              ce[CCD_CACHE_WORDS - 1] = p;
              ++kc;
            }
          }
          if (cv) ++c;
          ++a;
        }
        // (candidates beyond the capacity are dropped, convex ones among them: only the kept convex candidates count)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) kc += __shfl_xor(kc, off, 64);
        if (lane == 0) wtot[8 + wv] = kc;
        __syncthreads();
        ncvx += wtot[8] + wtot[9] + wtot[10] + wtot[11];
        ncand += t1;
      }
      if (tid == 0) {
        gcand[ccap] = min(ncand, ccap);
        gcand[ccap + 1] = ncand;
        gcand[ccap + 2] = ncvx;
        if (ncvx > __atomic_load_n(cnt, __ATOMIC_RELAXED)) atomicMax(cnt, ncvx);  // (see k_ccd_broad)
      }
    }
  }
}
// counters to zero; and which pairs of the filtered list are convex pairs (a wavefront per 64 pairs; the pair types are the model's, box-box
// depends on DisableBit.NATIVECCD of the step) -- k_broad_mask ranks a world's convex candidates with it
__global__ void __launch_bounds__(64) k_ccd_reset(MjhModel m, MjhData d) {
  const CcdLayout CL = ccd_layout_of(m, d);
  if (blockIdx.x == 0 && threadIdx.x < 8) reinterpret_cast<int*>(d.ws_ccd + CL.cnt)[threadIdx.x] = 0;
  const int p = 64 * (int)blockIdx.x + (int)threadIdx.x;
  bool cvx = false;
  if (p < m.npair) {
    const int t1 = m.geom_type[m.nxn_geom_pair[2 * p]], t2 = m.geom_type[m.nxn_geom_pair[2 * p + 1]];
    cvx = is_ccd_pair(m, min(t1, t2), max(t1, t2));
  }
  const unsigned long long b = __ballot(cvx);
  if (threadIdx.x == 0 && 2 * (int)blockIdx.x + 1 < CL.nbw) {
    unsigned* cm = reinterpret_cast<unsigned*>(d.ws_ccd + CL.cmask);
    cm[2 * blockIdx.x] = (unsigned)b;
    cm[2 * blockIdx.x + 1] = (unsigned)(b >> 32);
  }
}
template <int G>
__global__ void __launch_bounds__(256) k_ccd_broad(MjhModel m, MjhData d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  collision_body<G, true, 1>(m, d, smem, blk_of_launch<G>());
}
// GJK for every convex candidate of every world; the result to the candidate's cache entry, penetrating pairs to the EPA list.
// CGJ = 1: one LANE per entry.  CGJ = 8 / 16 / 32: a group of CGJ lanes per entry -- chosen by the launch when the list is too short to fill the
// device with one lane per entry: on the ALOHA scene a world has 3-14 convex candidates (65 k lanes = ONE wavefront per SIMD) and a pair
// evaluates 80-360 hill-climbing neighbours (the pot's vertices have up to 43) one after the other, three dependent loads each: 400 us of
// latency on an idle device.  The group spreads a step's neighbours over its lanes (ccd_support_c: the serial scan's result, ties included).
// The number of work items is only known on the device: the launch enqueues the instantiations back to back and each returns at once
// unless the number lies in its range [lo, hi).
#ifndef MJH_GJK_WAVES  // wavefronts per SIMD the register allocation of k_ccd_gjk aims at (developer knob)
#define MJH_GJK_WAVES 2
#endif
template <int CGJ>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MJH_GJK_WAVES, 8))) k_ccd_gjk(MjhModel m, MjhData d, int lo, int hi) {
  const CcdLayout CL = ccd_layout_of(m, d);
  int* cnt = reinterpret_cast<int*>(d.ws_ccd + CL.cnt);
#ifdef MJH_DBG_GJK_STATS
  g_dbg_cnt = cnt;
#endif
#ifdef MJH_DBG_GJK_CLOCK
  g_dbg_gclk = cnt;
  unsigned long long kt0_ = __builtin_amdgcn_s_memtime();
#endif
  const int nrank = min(cnt[0], CL.ccap), ntask = nrank * d.nworld;  // (slot, world) work items, empty ones included
  if (ntask < lo || ntask >= hi) return;
  // (the launch is sized for the device, not for the list's capacity: whole wavefronts walk the list with the grid's stride)
  const int lig = threadIdx.x & (CGJ - 1);
  // Work item t = (rank r among a world's convex candidates, world w), worlds fastest: the lanes of a wavefront hold the SAME rank of
  // consecutive worlds -- in a batch of similar worlds the same geom pair, hence one path through the support functions and the simplex
  // code and the same mesh tables, where the flat list's order (a world's candidates side by side) made every wavefront run the union
  // of eight different pairs' paths.  (The number of work items picks the instantiation.)
  for (int t0 = blockIdx.x * (blockDim.x / CGJ); t0 < ntask; t0 += gridDim.x * (blockDim.x / CGJ)) {
  const int t = t0 + (int)threadIdx.x / CGJ;
  int st = 0, w = 0, p = 0, slot = 0, idx1 = -1, idx2 = -1;
  GjkOut res;
  bool have = false;
  if (t < ntask) {
    slot = t / d.nworld;
    w = t - slot * d.nworld;
    const float* cw = d.ws_ccd + (size_t)w * CL.world_stride;
    have = slot < reinterpret_cast<const int*>(cw + CL.cand)[CL.ccap + 2];
    // sleeping, second collision pass (forward.py:652-666): a world where no contact of pass 1 woke a tree keeps its candidate list, its result
    // cache and its contacts from pass 1 (k_broad_mask and k_collision skip it too) -- without this gate pass 2 re-ran GJK and EPA for every
    // world: 230 us of the 1,430 us clutter_synth step (round 6, dispatch timeline)
    if (have && d.sleep_pass == 2 && !d.ws_sleep_flag[w]) have = false;
    if (have) p = reinterpret_cast<const int*>(cw + CL.cache + (size_t)slot * CCD_CACHE_WORDS)[CCD_CACHE_WORDS - 1];
  }
  if (have) {
    int g1 = m.nxn_geom_pair[2 * p], g2 = m.nxn_geom_pair[2 * p + 1];
    int t1 = m.geom_type[g1], t2 = m.geom_type[g2];
    if (t1 > t2) {
      int x = g1; g1 = g2; g2 = x;
      x = t1; t1 = t2; t2 = x;
    }
    const int ng = m.ngeom;
    const int pid = m.nexplicit ? m.nxn_pairid[p] : -1;
    const float* gmargin = bf(m.geom_margin, m.geom_margin_nb, w, ng);
    const float* ggap = bf(m.geom_gap, m.geom_gap_nb, w, ng);
    const float* gsize = bf(m.geom_size, m.geom_size_nb, w, 3 * ng);
    const float margin = pid >= 0 ? m.pair_margin[pid] : gmargin[g1] + gmargin[g2];
    const float gap = pid >= 0 ? m.pair_gap[pid] : ggap[g1] + ggap[g2];
    const float* gxpos = d.geom_xpos + (size_t)w * 3 * ng;
    const float* gxmat = d.geom_xmat + (size_t)w * 9 * ng;
    const float *mv1 = nullptr, *mv2 = nullptr;
    int mn1 = 0, mn2 = 0, me1 = -1, me2 = -1;
    if (t1 == G_MESH) { me1 = m.geom_dataid[g1]; mv1 = m.mesh_vert + 3 * m.mesh_vertadr[me1]; mn1 = m.mesh_vertnum[me1]; }
    if (t2 == G_MESH) { me2 = m.geom_dataid[g2]; mv2 = m.mesh_vert + 3 * m.mesh_vertadr[me2]; mn2 = m.mesh_vertnum[me2]; }
    float* cache = d.ws_ccd + (size_t)w * CL.world_stride + CL.cache + (size_t)slot * CCD_CACHE_WORDS;
    int nem = 0;
    const float ccd_tol = bf(m.opt_ccd_tolerance, m.opt_ccd_tolerance_nb, w, 1)[0];
#ifdef MJH_DBG_GJK_CLOCK
    if (lig == 0) atomicAdd(cnt + 5, (int)((__builtin_amdgcn_s_memtime() - kt0_) >> 10));  // pair set-up
    kt0_ = __builtin_amdgcn_s_memtime();
#endif
#ifdef MJH_DBG_GJK_ITER  // (profiling variants: 0 = the launch without GJK, n = at most n iterations)
    const int gjk_cap = MJH_DBG_GJK_ITER;
    if (gjk_cap > 0)
#else
    const int gjk_cap = CCD_MAX_ITER;
#endif
    st = convex_gjk_lane<(CGJ > 1 ? CGJ : 0)>(ccd_tol, min(m.ccd_iterations, gjk_cap), t1, t2, ld3(gxpos + 3 * g1), gxmat + 9 * g1, ld3(gsize + 3 * g1), ld3(gxpos + 3 * g2), gxmat + 9 * g2,
                         ld3(gsize + 3 * g2), margin, gap, [&](int k, float dist, V3 pos, V3 fa, V3 fb, V3 fc) { ccd_cache_store(cache, nem, k, dist, pos, fa, fb, fc, lig == 0); },
                         mv1, mn1, mv2, mn2, m, me1, me2, res, idx1, idx2, lig);
    if (lig == 0) reinterpret_cast<int*>(cache)[0] = nem;
#ifdef MJH_DBG_GJK_CLOCK
    if (lig == 0) atomicAdd(cnt + 6, (int)((__builtin_amdgcn_s_memtime() - kt0_) >> 10));  // the whole GJK phase
    kt0_ = __builtin_amdgcn_s_memtime();
#endif
    if (lig != 0) st = 0;  // (the group's first lane hands the pair over)
  }
  // penetrating pairs: one reservation per wavefront in the EPA list, then every such lane writes its hand-over record
  const unsigned long long em = __ballot(st == 2);
  if (em) {
    const int lane = threadIdx.x & 63;
    int base0 = 0;
    if (lane == __ffsll((long long)em) - 1) base0 = atomicAdd(cnt + 1, __popcll(em));
    base0 = __shfl(base0, __ffsll((long long)em) - 1, 64);
    if (st == 2) {
      const int h = base0 + __popcll(em & ((1ull << lane) - 1ull));
      if (h < CL.handcap) {
        float* hand = d.ws_ccd + CL.hand + (size_t)h * CCD_HAND_WORDS;
        int* hi = reinterpret_cast<int*>(hand);
        hi[0] = w; hi[1] = p; hi[2] = slot; hi[3] = idx1; hi[4] = idx2; hi[5] = res.dim; hi[6] = res.separated ? 1 : 0;
        hand[7] = res.dist;
        st3(hand + 8, res.x1);
        st3(hand + 11, res.x2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          st3(hand + 14 + 9 * i, res.s[i]);
          st3(hand + 17 + 9 * i, res.s1[i]);
          st3(hand + 20 + 9 * i, res.s2[i]);
          hi[50 + i] = res.i1[i];
          hi[54 + i] = res.i2[i];
        }
      } else {
        atomicOr(d.overflow + w, OVF_CCD);  // more penetrating pairs than Data.nccdhand: dropped
      }
    }
  }
  }
}
// one lane GROUP per entry of the EPA list: polytope in the group's LDS, then the multi-contact recovery; lane 0 stores the result
#ifndef MJH_EPA_WAVES  // wavefronts per SIMD the register allocation of k_ccd_epa aims at (developer knob)
// (round 5: 2 -- 237 VGPRs, no spills -- instead of 3 -- 168 VGPRs, 60 spilled: with the multi-contact recovery inlined the spills sit in its
// serial loops; ALOHA scene, 8192 worlds, same box: 941 -> 911 us per step)
#define MJH_EPA_WAVES 2
#endif
template <int G>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MJH_EPA_WAVES, 8))) k_ccd_epa(MjhModel m, MjhData d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const CcdLayout CL = ccd_layout_of(m, d);
  const int npend = min(reinterpret_cast<const int*>(d.ws_ccd + CL.cnt)[1], CL.handcap);
#ifdef MJH_DBG_EPA_CLOCK
  g_dbg_clk = reinterpret_cast<int*>(d.ws_ccd + CL.cnt);
#endif
  const int lig = threadIdx.x & (G - 1), gib = threadIdx.x / G;
  const int it = max(m.ccd_iterations, m.epa_iterations);
  float* poly = smem + (size_t)gib * ccd_coop_words(it, m.npolygonmax, m.nmeshdegmax);
  for (int h = blockIdx.x * (blockDim.x / G) + gib; h < npend; h += gridDim.x * (blockDim.x / G)) {
  const float* hand = d.ws_ccd + CL.hand + (size_t)h * CCD_HAND_WORDS;
  const int* hi = reinterpret_cast<const int*>(hand);
  const int w = hi[0], p = hi[1], slot = hi[2], idx1 = hi[3], idx2 = hi[4];
  GjkOut res;
  res.dim = hi[5];
  res.separated = hi[6] != 0;
  res.dist = hand[7];
  res.x1 = ld3(hand + 8);
  res.x2 = ld3(hand + 11);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    res.s[i] = ld3(hand + 14 + 9 * i);
    res.s1[i] = ld3(hand + 17 + 9 * i);
    res.s2[i] = ld3(hand + 20 + 9 * i);
    res.i1[i] = hi[50 + i];
    res.i2[i] = hi[54 + i];
  }
  int g1 = m.nxn_geom_pair[2 * p], g2 = m.nxn_geom_pair[2 * p + 1];
  int t1 = m.geom_type[g1], t2 = m.geom_type[g2];
  if (t1 > t2) {
    int x = g1; g1 = g2; g2 = x;
    x = t1; t1 = t2; t2 = x;
  }
  const int ng = m.ngeom;
  const int pid = m.nexplicit ? m.nxn_pairid[p] : -1;
  const float* gmargin = bf(m.geom_margin, m.geom_margin_nb, w, ng);
  const float* ggap = bf(m.geom_gap, m.geom_gap_nb, w, ng);
  const float* gsize = bf(m.geom_size, m.geom_size_nb, w, 3 * ng);
  const float margin = pid >= 0 ? m.pair_margin[pid] : gmargin[g1] + gmargin[g2];
  const float gap = pid >= 0 ? m.pair_gap[pid] : ggap[g1] + ggap[g2];
  const float* gxpos = d.geom_xpos + (size_t)w * 3 * ng;
  const float* gxmat = d.geom_xmat + (size_t)w * 9 * ng;
  const float *mv1 = nullptr, *mv2 = nullptr;
  int mn1 = 0, mn2 = 0, me1 = -1, me2 = -1;
  if (t1 == G_MESH) { me1 = m.geom_dataid[g1]; mv1 = m.mesh_vert + 3 * m.mesh_vertadr[me1]; mn1 = m.mesh_vertnum[me1]; }
  if (t2 == G_MESH) { me2 = m.geom_dataid[g2]; mv2 = m.mesh_vert + 3 * m.mesh_vertadr[me2]; mn2 = m.mesh_vertnum[me2]; }
  float* cache = d.ws_ccd + (size_t)w * CL.world_stride + CL.cache + (size_t)slot * CCD_CACHE_WORDS;
  float* mcws = nullptr;  // (the multi-contact recovery's lists live in the group's LDS)
  int nem = 0, overflow = 0;
  const float ccd_tol = bf(m.opt_ccd_tolerance, m.opt_ccd_tolerance_nb, w, 1)[0];
  convex_epa_group<G>(ccd_tol, min(m.epa_iterations, CCD_MAX_ITER), t1, t2, ld3(gxpos + 3 * g1), gxmat + 9 * g1, ld3(gsize + 3 * g1), ld3(gxpos + 3 * g2), gxmat + 9 * g2, ld3(gsize + 3 * g2),
                      margin, gap, poly, overflow, [&](int k, float dist, V3 pos, V3 fa, V3 fb, V3 fc) { ccd_cache_store(cache, nem, k, dist, pos, fa, fb, fc, lig == 0); },
                      mv1, mn1, mv2, mn2, m, me1, me2, res, idx1, idx2, lig, mcws);
  if (lig == 0) {
    reinterpret_cast<int*>(cache)[0] = nem;
    if (overflow) atomicOr(d.overflow + w, overflow);
  }
  gsync();
  }
}

template <int G, bool HEAVY, bool HFT = true>
__global__ void __launch_bounds__(256) k_collision(MjhModel m, MjhData d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  collision_body<G, HEAVY, 0, HFT>(m, d, smem, blk_of_launch<G>());
}

template <int G>
__global__ void __launch_bounds__(256) k_publish_contacts(MjhData d, int self_prefix, const float* pair_solreffriction) {
  __shared__ int sh[64];
  publish_body<G>(d, self_prefix, sh, blk_of_launch<G>(), pair_solreffriction);
}
