// Ray casting against the primitive geoms (reference ray.py): ray() / rays() of the public API and the rangefinder sensor.
// One thread per (world, ray) walks the world's geoms -- a model has tens of them and a caller of rays() brings hundreds of rays per
// world, so the rays are the parallel axis (the reference spends a block per ray and its threads on the geoms).
// Mesh and height-field geoms are not intersected (they report no hit).
#pragma once
#include "dev_common.hpp"

// smallest non-negative root of a x^2 + 2 b x + c = 0 (ray.py:105-125 _ray_quad); both roots in x0, x1 (-1 when there is none)
DEV float ray_quad(float a, float b, float c, float& x0, float& x1) {
  x0 = x1 = -1.0f;
  float det = b * b - a * c;
  if (det < MJ_MINVAL) return -1.0f;
  det = sqrtf(det);
  const float den = safe_div(1.0f, a);
  x0 = (-b - det) * den;
  x1 = (-b + det) * den;
  return x0 >= 0.0f ? x0 : (x1 >= 0.0f ? x1 : -1.0f);
}
DEV float ray_sphere(V3 pos, float dist_sqr, V3 pnt, V3 vec, V3& normal) {  // ray.py:237-251
  const V3 dif = pnt - pos;
  float x0, x1;
  const float sol = ray_quad(dot(vec, vec), dot(vec, dif), dot(dif, dif) - dist_sqr, x0, x1);
  normal = sol >= 0.0f ? normalize(pnt + vec * sol - pos) : V3{0, 0, 0};
  return sol;
}
DEV float ray_plane(V3 pos, const float* mat, V3 size, V3 pnt, V3 vec, V3& normal) {  // ray.py:213-234: front face only, inside the rendered rectangle
  normal = V3{0, 0, 0};
  const V3 lp = matT_mul(mat, pnt - pos), lv = matT_mul(mat, vec);
  if (lv.z > -MJ_MINVAL) return -1.0f;
  const float x = -lp.z / lv.z;
  if (x < 0.0f) return -1.0f;
  const float px = lp.x + x * lv.x, py = lp.y + x * lv.y;
  if ((size.x <= 0.0f || fabsf(px) <= size.x) && (size.y <= 0.0f || fabsf(py) <= size.y)) {
    normal = V3{mat[2], mat[5], mat[8]};
    return x;
  }
  return -1.0f;
}
DEV float ray_capsule(V3 pos, const float* mat, V3 size, V3 pnt, V3 vec, V3& normal) {  // ray.py:254-325
  const float ssz = size.x + size.y;
  if (ray_sphere(pos, ssz * ssz, pnt, vec, normal) < 0.0f) {
    normal = V3{0, 0, 0};
    return -1.0f;
  }
  const V3 lp = matT_mul(mat, pnt - pos), lv = matT_mul(mat, vec);
  float x = -1.0f, x0, x1;
  const float r2 = size.x * size.x;
  float a = lv.x * lv.x + lv.y * lv.y;
  float sol = ray_quad(a, lv.x * lp.x + lv.y * lp.y, lp.x * lp.x + lp.y * lp.y - r2, x0, x1);
  int part = 0;  // -1 bottom cap, 0 side, 1 top cap
  if (sol >= 0.0f && fabsf(lp.z + sol * lv.z) <= size.y) x = sol;
  a += lv.z * lv.z;
  for (int cap = 1; cap >= -1; cap -= 2) {  // top first, then bottom: only the outer half of each sphere
    const V3 ld = V3{lp.x, lp.y, lp.z - (float)cap * size.y};
    ray_quad(a, dot(lv, ld), dot(ld, ld) - r2, x0, x1);
    for (int i = 0; i < 2; ++i) {
      const float xi = i ? x1 : x0;
      if (xi >= 0.0f && (float)cap * (lp.z + xi * lv.z) >= size.y && (x < 0.0f || xi < x)) {
        x = xi;
        part = cap;
      }
    }
  }
  normal = V3{0, 0, 0};
  if (x >= 0.0f) {
    const V3 n = V3{lp.x + lv.x * x, lp.y + lv.y * x, part == 0 ? 0.0f : lp.z + lv.z * x - size.y * (float)part};
    normal = mat_mul(mat, normalize(n));
  }
  return x;
}
DEV float ray_ellipsoid(V3 pos, const float* mat, V3 size, V3 pnt, V3 vec, V3& normal) {  // ray.py:328-356
  const V3 lp = matT_mul(mat, pnt - pos), lv = matT_mul(mat, vec);
  const V3 s = V3{safe_div(1.0f, size.x * size.x), safe_div(1.0f, size.y * size.y), safe_div(1.0f, size.z * size.z)};
  const V3 slv = V3{s.x * lv.x, s.y * lv.y, s.z * lv.z}, slp = V3{s.x * lp.x, s.y * lp.y, s.z * lp.z};
  float x0, x1;
  const float sol = ray_quad(dot(slv, lv), dot(slv, lp), dot(slp, lp) - 1.0f, x0, x1);
  normal = V3{0, 0, 0};
  if (sol >= 0.0f) {
    const V3 l = lp + lv * sol;
    normal = mat_mul(mat, normalize(V3{s.x * l.x, s.y * l.y, s.z * l.z}));
  }
  return sol;
}
DEV float ray_cylinder(V3 pos, const float* mat, V3 size, V3 pnt, V3 vec, V3& normal) {  // ray.py:359-417
  if (ray_sphere(pos, size.x * size.x + size.y * size.y, pnt, vec, normal) < 0.0f) {
    normal = V3{0, 0, 0};
    return -1.0f;
  }
  const V3 lp = matT_mul(mat, pnt - pos), lv = matT_mul(mat, vec);
  float x = -1.0f;
  int part = 0;
  if (fabsf(lv.z) > MJ_MINVAL)
    for (int side = -1; side <= 1; side += 2) {
      const float sol = ((float)side * size.y - lp.z) / lv.z;
      if (sol >= 0.0f) {
        const float px = lp.x + sol * lv.x, py = lp.y + sol * lv.y;
        if (px * px + py * py <= size.x * size.x && (x < 0.0f || sol < x)) {
          x = sol;
          part = side;
        }
      }
    }
  float x0, x1;
  const float sol = ray_quad(lv.x * lv.x + lv.y * lv.y, lv.x * lp.x + lv.y * lp.y, lp.x * lp.x + lp.y * lp.y - size.x * size.x, x0, x1);
  if (sol >= 0.0f && fabsf(lp.z + sol * lv.z) <= size.y && (x < 0.0f || sol < x)) {
    x = sol;
    part = 0;
  }
  normal = V3{0, 0, 0};
  if (x >= 0.0f) {
    const V3 l = lp + lv * x;
    normal = mat_mul(mat, part == 0 ? normalize(V3{l.x, l.y, 0.0f}) : V3{0, 0, (float)part});
  }
  return x;
}
DEV float ray_box(V3 pos, const float* mat, V3 size, V3 pnt, V3 vec, V3& normal) {  // ray.py:420-471
  if (ray_sphere(pos, dot(size, size), pnt, vec, normal) < 0.0f) {
    normal = V3{0, 0, 0};
    return -1.0f;
  }
  const V3 lpv = matT_mul(mat, pnt - pos), lvv = matT_mul(mat, vec);
  const float lp[3] = {lpv.x, lpv.y, lpv.z}, lv[3] = {lvv.x, lvv.y, lvv.z}, sz[3] = {size.x, size.y, size.z};
  float x = -1.0f;
  int face_axis = -1, face_side = -1;
  for (int i = 0; i < 3; ++i) {
    if (!(fabsf(lv[i]) > MJ_MINVAL)) continue;
    for (int side = -1; side <= 1; side += 2) {
      const float sol = ((float)side * sz[i] - lp[i]) / lv[i];
      if (sol < 0.0f) continue;
      const int id0 = i == 0 ? 1 : 0, id1 = i == 2 ? 1 : 2;
      if (fabsf(lp[id0] + sol * lv[id0]) <= sz[id0] && fabsf(lp[id1] + sol * lv[id1]) <= sz[id1] && (x < 0.0f || sol < x)) {
        x = sol;
        face_axis = i;
        face_side = side;
      }
    }
  }
  normal = V3{0, 0, 0};
  if (x >= 0.0f) normal = V3{mat[face_axis], mat[3 + face_axis], mat[6 + face_axis]} * (float)face_side;
  return x;
}
DEV float ray_geom(int type, V3 pos, const float* mat, V3 size, V3 pnt, V3 vec, V3& normal) {  // ray.py:798-819
  normal = V3{0, 0, 0};
  switch (type) {
    case G_PLANE: return ray_plane(pos, mat, size, pnt, vec, normal);
    case G_SPHERE: return ray_sphere(pos, size.x * size.x, pnt, vec, normal);
    case G_CAPSULE: return ray_capsule(pos, mat, size, pnt, vec, normal);
    case G_ELLIPSOID: return ray_ellipsoid(pos, mat, size, pnt, vec, normal);
    case G_CYLINDER: return ray_cylinder(pos, mat, size, pnt, vec, normal);
    case G_BOX: return ray_box(pos, mat, size, pnt, vec, normal);
    default: return -1.0f;
  }
}
struct RayGroup {
  float g[6];
};
// geoms a ray ignores (ray.py:52-102 _ray_eliminate): the excluded body's, invisible ones (alpha 0 on the geom or on its material),
// static ones unless flg_static, and those outside the group mask (a mask of six -1 includes every group)
DEV bool ray_eliminate(const MjhModel& m, int g, const RayGroup& gg, int flg_static, int bodyexclude) {
  const int b = m.geom_bodyid[g], mat = m.geom_matid[g];
  if (b == bodyexclude) return true;
  if (mat < 0 && m.geom_rgba[4 * g + 3] == 0.0f) return true;
  if (mat >= 0 && m.mat_rgba[4 * mat + 3] == 0.0f) return true;
  if (!flg_static && m.body_weldid[b] == 0) return true;
  bool none = true;
  for (int i = 0; i < 6; ++i) none = none && gg.g[i] == -1.0f;
  if (none) return false;
  return gg.g[min(5, max(0, m.geom_group[g]))] == 0.0f;
}
// nearest hit of one ray in world w (ray.py:907-1011 _ray): distance (-1: none), the geom and the surface normal there
DEV float ray_world(const MjhModel& m, const MjhData& d, int w, V3 pnt, V3 vec, const RayGroup& gg, int flg_static, int bodyexclude, int& geomid, V3& normal) {
  float best = MJ_MAXVAL;
  geomid = -1;
  normal = V3{0, 0, 0};
  for (int g = 0; g < m.ngeom; ++g) {
    if (ray_eliminate(m, g, gg, flg_static, bodyexclude)) continue;
    V3 n;
    const float dist = ray_geom(m.geom_type[g], ld3(d.geom_xpos + ((size_t)w * m.ngeom + g) * 3), d.geom_xmat + ((size_t)w * m.ngeom + g) * 9,
                                ld3(bf(m.geom_size, m.geom_size_nb, w, 3 * m.ngeom) + 3 * g), pnt, vec, n);
    if (dist >= 0.0f && dist < best) {
      best = dist;
      geomid = g;
      normal = n;
    }
  }
  return best >= MJ_MAXVAL ? -1.0f : best;
}
// rays (ray.py:1219-1325): pnt / vec [pnt_nworld (1 or nworld), nray, 3]; bodyexclude [nray]; outputs [nworld, nray]
__global__ void __launch_bounds__(256) k_rays(MjhModel m, MjhData d, const float* pnt, const float* vec, int pnt_nworld, int nray, RayGroup gg, int flg_static,
                                              const int* bodyexclude, float* dist, int* geomid, float* normal) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= d.nworld * nray) return;
  const int w = idx / nray, r = idx - w * nray;
  const size_t src = ((size_t)(w % pnt_nworld) * nray + r) * 3;
  int g;
  V3 n;
  const float x = ray_world(m, d, w, ld3(pnt + src), ld3(vec + src), gg, flg_static, bodyexclude ? bodyexclude[r] : -1, g, n);
  dist[idx] = x;
  if (geomid) geomid[idx] = g;
  if (normal) st3(normal + (size_t)idx * 3, n);
}
