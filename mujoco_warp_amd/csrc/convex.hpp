// convex.hpp -- general convex narrowphase (GJK distance + EPA penetration) for the primitive convex shapes.
//
// Reference: collision_gjk.py support 116-223, signed-volume sub-distance 281-593, gjk 635-770, polytope seeds 1021-1286,
// EPA 1319-1454, witness points 947-1018, gjk_phase / epa_phase / ccd 2303-2575; collision_convex.py 747-977
// (eval_ccd_write_contact: margins, cutoff, un-inflated distance, frame, midpoint).  Pair types routed here
// (MJ_COLLISION_TABLE collision_driver.py:47-80 minus meshes / height fields): sphere-ellipsoid, capsule-ellipsoid,
// capsule-cylinder, ellipsoid-ellipsoid / -cylinder / -box, cylinder-cylinder / -box.
//
// MI355X mapping: the reference runs one thread per candidate pair with the EPA polytope in a global scratch row per pair.
// Here a candidate pair is still one lane (the collision kernel's lane-per-candidate narrowphase); the GJK simplex lives in
// registers, the EPA polytope of a lane in a per-world global workspace interleaved by lane (word k of lane l at k * 32 + l), so the
// 32 lanes of a world that walk their polytopes in step touch consecutive addresses.  Only the HEAVY collision instantiation
// carries this code.
#pragma once
#include "dev_common.hpp"

#define CCD_FLOAT_MAX 1e30f
#define CCD_MINVAL 1e-15f
#define CCD_MINVAL2 1e-30f
#define CCD_MIN_DIST2 1e-10f
#define CCD_MIN_DIST3 1e-10f
#define CCD_MIN_DIST4 1e-17f
#define CCD_MIN_EPATOL 1e-7f
#define CCD_MAX_HORIZON 24  // types.py:31
#define CCD_EPAFACES 5      // types.py:33
#define CCD_MAX_ITER 64     // cap of opt.ccd_iterations on this engine (workspace words per lane: ccd_words)
#define CCD_LANES 32        // polytope slots per world = lanes of a world's group

// The narrowphase visits every candidate twice (pass 1 counts contacts, pass 2 writes records): a lane keeps the contacts of its
// first CCD_CACHE_SLOTS convex candidates from pass 1 -- count, distance, frame, up to four positions -- so that pass 2 replays them
// instead of running GJK / EPA again (measured: half of the collision time of a box scene)
#define CCD_CACHE_SLOTS 4
#define CCD_CACHE_WORDS 24
// workspace words of one lane: polytope = vertex pairs (6 floats + 2 ids) | faces (packed verts, projection, norm^2) | horizon; contact cache
__host__ __device__ inline int ccd_poly_words(int iterations) {
  const int it = iterations < CCD_MAX_ITER ? iterations : CCD_MAX_ITER;
  return 8 * (5 + it) + 5 * (6 + CCD_EPAFACES * it) + CCD_MAX_HORIZON;
}
#define CCD_HF_MAXCONPAIR 50  // mjMAXCONPAIR (types.py:27): prisms of one height-field pair whose result is kept
#define CCD_HF_WORDS (7 * CCD_HF_MAXCONPAIR + 1)  // distance, position, normal of each + their count (models with height fields only)
__host__ __device__ inline int ccd_words(int iterations, int hfield = 0) {
  return ccd_poly_words(iterations) + CCD_CACHE_SLOTS * CCD_CACHE_WORDS + (hfield ? CCD_HF_WORDS : 0);
}
// Round 4: the convex narrowphase as its own LAUNCHES in front of the contact kernel (models with GJK pairs; csrc/collide.hpp).
//   k_broad_mask  a workgroup per world: the broadphase filters of the whole pair list, results as a bit mask (NXN broadphase)
//   k_ccd_broad   one lane group per world: the mask expanded in pair order (or the sweep-and-prune broadphase), candidate list to the
//                 world's slice of Data.ws_ccd; a convex candidate's cache slot = its rank among the world's convex candidates
//   k_ccd_gjk     GJK with nothing but registers, work items ordered (slot, world): a wavefront holds the same slot of consecutive
//                 worlds -- in a batch of similar worlds one geom pair, one path through the code, the same mesh tables.  One lane per
//                 item, or 8 / 32 lanes (the mesh support function spreads a hill-climbing step's neighbours over them) when there are
//                 too few items to fill the device.  Its cost is the chain of dependent table loads of the support function (hill
//                 climbing on a mesh graph: edge list -> vertex id -> vertex), which only concurrency hides -- as one role of the
//                 heavy contact kernel (509 VGPRs, one wavefront per SIMD) every round trip stalled the SIMD.  Results (separated /
//                 one shallow contact) go to the candidate's cache entry; penetrating pairs append their simplex to the EPA list
//                 (one reservation per wavefront)
//   k_ccd_epa     one lane GROUP per EPA entry: the polytope in ONE copy in the group's LDS (a lane used to walk its slice of global
//                 memory at one round trip per face: 2 scans x 181 faces x 35 iterations), the nearest-face and visible-face scans and
//                 the attachment of the horizon's faces spread over the lanes, the mesh support function spreading the neighbours of a
//                 hill-climbing step over the lanes, the rest redundantly in every lane (uniform control flow, broadcast loads); then
//                 the multi-contact recovery.  Arg-max / arg-min ties go to the smallest index like the serial loops: results are
//                 identical to the one-lane code (the 17 reference-held GJK poses stay digit for digit).
// The contact kernel (k_mid / k_collision) then only replays the cached results (count | distance | frame | up to four positions) in
// its two passes.  (A first cut that ran whole pairs cooperatively inside the contact kernel, one after the other, was 30 % SLOWER on the
// ALOHA scene: it serialised the GJK chains that the lanes had at least run side by side.)
//
// Layout of Data.ws_ccd (floats; csrc/collide.hpp and io.py size it with ccd_layout):
//   per world  [hf]     per-lane polytope + result table of the height-field prisms, interleaved by lane (models with height fields)
//              [cache]  ccap x CCD_CACHE_WORDS: one entry per convex candidate, slot = its rank among the world's convex candidates
//              [cand]   ccap + 4 ints: the world's candidate pairs in canonical order | ncand, nbroad, nconvex
//   tail       [cnt]    8 ints: longest convex candidate list of a world, EPA entries (zeroed by k_ccd_reset in front of k_ccd_broad)
//              [cmask]  2 ceil(npair / 64) words: which pairs of the list are convex pairs (k_ccd_reset)
//              [hand]   handcap x CCD_HAND_WORDS: list entry | vertex caches | the GJK simplex, for the EPA launch
//              [mc]     handcap x ccd_mc_words: multi-contact buffers of the EPA groups (stride 1)
#define CCD_HAND_WORDS 64
struct CcdLayout {
  size_t world_stride;  // floats per world
  size_t hf, cache, cand, bmask;  // offsets inside a world's slice
  size_t tail, cnt, cmask, hand, mc, total;  // offsets from the start of ws_ccd
  int ccap, handcap, mcw, nbw;
};
__host__ __device__ inline CcdLayout ccd_layout(int nworld, int iterations, int nhfield, int npolygonmax, int nmeshdegmax, int ccap, int handcap, int npair) {
  CcdLayout L;
  L.ccap = ccap;
  L.hf = 0;
  L.cache = nhfield ? (size_t)(ccd_poly_words(iterations) + CCD_CACHE_SLOTS * CCD_CACHE_WORDS + CCD_HF_WORDS) * CCD_LANES : 0;
  L.cand = L.cache + (size_t)ccap * CCD_CACHE_WORDS;
  // k_broad_mask's bit mask over the filtered pair list (bit p % 32 of word p / 32): the pairs that passed the broadphase filters; 64-pair
  // granules, one per wavefront trip of that launch
  L.nbw = 2 * ((npair + 63) / 64);
  L.bmask = ((L.cand + ccap + 4 + 3) / 4) * 4;
  L.world_stride = ((L.bmask + (size_t)L.nbw + 3) / 4) * 4;
  L.tail = L.world_stride * (size_t)nworld;
  L.cnt = L.tail;
  L.handcap = handcap;
  L.cmask = L.cnt + 8;  // bit mask over the pair list: the pairs the convex launches serve (k_ccd_reset writes it: the pair types with the step's flags)
  L.hand = ((L.cmask + (size_t)L.nbw + 3) / 4) * 4;
  // (the multi-contact recovery of k_ccd_epa works in the lane group's LDS: round 4 kept 11 D + 22 P words of global scratch per entry --
  // 2.25 of the ALOHA scene's 2.75 GB)
  L.mcw = 0;
  L.mc = L.hand + (size_t)handcap * CCD_HAND_WORDS;
  L.total = L.mc;
  return L;
}
// LDS words of an EPA group: the polytope; for models with multi-contact recovery on mesh faces (nmeshdegmax > 0) also the polygon buffers of
// that recovery (P = npolygonmax): the two clip buffers (12 P words) from word 0 -- over the polytope, which is dead by the time they are
// written -- and the two faces (6 P words) behind them, but not before the polytope's face records end (vertices, vertex ids and face
// records are still read while the faces are gathered; the face normals / distances behind them are not)
// the multi-contact recovery's feature lists (vertex-normal indices and normals of the two features, edge end points: 11 D words, D =
// nmeshdegmax) sit right behind the polytope's live records (round 5; global memory before: the serial loops over them -- up to 43 x 43
// normal pairs on the ALOHA pot -- were three quarters of k_ccd_epa); the faces start behind them; the clip buffers, written when the lists
// are dead, cover them.
__host__ __device__ inline int ccd_coop_scratch_offset(int iterations) {
  const int it = iterations < CCD_MAX_ITER ? iterations : CCD_MAX_ITER;
  return (8 * (5 + it) + (6 + CCD_EPAFACES * it) + 3) & ~3;  // vertices (6 words) + vertex ids (2) per vertex, one word per face
}
__host__ __device__ inline int ccd_coop_face_offset(int iterations, int npolygonmax, int nmeshdegmax) {
  const int P = npolygonmax > 4 ? npolygonmax : 4, D = nmeshdegmax > 3 ? nmeshdegmax : 3;
  const int lists_end = ccd_coop_scratch_offset(iterations) + 11 * D;
  return 12 * P > lists_end ? 12 * P : lists_end;
}
__host__ __device__ inline int ccd_coop_words(int iterations, int npolygonmax, int nmeshdegmax) {
  const int pw = ccd_poly_words(iterations), P = npolygonmax > 4 ? npolygonmax : 4;
  const int mc = nmeshdegmax > 0 ? ccd_coop_face_offset(iterations, npolygonmax, nmeshdegmax) + 6 * P : 0;
  return (((mc > pw ? mc : pw) + 3) / 4) * 4;
}
// (models with multi-contact recovery on mesh faces append ccd_mc_words(npolygonmax, nmeshdegmax) words per lane: further below)

#ifdef MJH_DBG_GJK_STATS  // (profiling variant: GJK iterations / hill-climbing steps / neighbour evaluations summed into the list counters 2..6)
__device__ int* g_dbg_cnt;
#define DBG_COUNT(k, n) atomicAdd(g_dbg_cnt + (k), (n))
#define DBG_MAX(k, n) atomicMax(g_dbg_cnt + (k), (n))
#else
#define DBG_COUNT(k, n)
#define DBG_MAX(k, n)
#endif
#ifdef MJH_DBG_GJK_CLOCK  // (profiling variant: shader-clock ticks of k_ccd_gjk's phases, lane 0 of every group, summed into the counters 2..7 in units of 1024 ticks)
__device__ int* g_dbg_gclk;
#define GJK_TICK(k)                                                                 \
  do {                                                                              \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();                   \
    if (lig == 0) atomicAdd(g_dbg_gclk + (k), (int)((now_ - gjk_t0_) >> 10));        \
    gjk_t0_ = __builtin_amdgcn_s_memtime();                                         \
  } while (0)
#define GJK_TICK_START() unsigned long long gjk_t0_ = __builtin_amdgcn_s_memtime()
#else
#define GJK_TICK(k)
#define GJK_TICK_START()
#endif
#ifdef MJH_DBG_EPA_CLOCK  // (profiling variant: shader-clock ticks of the EPA kernel's phases, lane 0 of every group, summed into the counters 2..7 in units of 16 ticks)
__device__ int* g_dbg_clk;
#define DBG_TICK(k)                                                                      \
  do {                                                                                   \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();                        \
    if ((threadIdx.x & 31) == 0) atomicAdd(g_dbg_clk + (k), (int)((now_ - dbg_t0_) >> 4)); \
    dbg_t0_ = __builtin_amdgcn_s_memtime();                                              \
  } while (0)
#define DBG_TICK_START() unsigned long long dbg_t0_ = __builtin_amdgcn_s_memtime()
#else
#define DBG_TICK(k)
#define DBG_TICK_START()
#endif
struct CcdGeom {
  int type;
  V3 pos;
  const float* rot;  // 3x3 row-major
  V3 size;
  float margin;
  const float* vert;  // mesh: vertices in the geom frame (Model.mesh_vert)
  int nvert;
  int index;  // mesh: vertex of the last support call (warm start: wins ties), -1 at the start (reference Geom.index)
  int meshid;  // mesh: Model.geom_dataid (polygon tables for the multi-contact recovery), -1 otherwise
  const int* graph;  // mesh: its block of Model.mesh_graph (hill climbing for meshes of 10 or more vertices) or nullptr
  int cache;  // out of ccd_support: SupportPoint.cached_index (a vertex id on the exhaustive path, a graph-local id when hill climbing)
  const V3* prism;  // height-field prism (type G_HFIELD): six vertices in the frame the pair is solved in (collision_gjk.py:197-206)
};
struct GjkOut {
  bool separated;
  int dim;
  float dist;
  V3 x1, x2;
  V3 s[4], s1[4], s2[4];
  int i1[4], i2[4];
};

DEV float ccd_sign(float x) { return x < 0.0f ? -1.0f : 1.0f; }

DEV V3 ccd_support(const CcdGeom& g, V3 dir, int& vid) {
  vid = -1;
  if (g.type == G_SPHERE) return g.pos + (g.size.x + 0.5f * g.margin) * dir;
  if (g.type == G_HFIELD) {  // the furthest of the prism's six vertices
    float best = -CCD_FLOAT_MAX;
    V3 p = V3{0.0f, 0.0f, 0.0f};
    vid = dir.z < 0.0f ? -2 : -3;
    for (int i = 0; i < 6; ++i) {
      const float dd = dot(g.prism[i], dir);
      if (dd > best) {
        best = dd;
        p = g.prism[i];
      }
    }
    if (g.margin > 0.0f) p = p + dir * (0.5f * g.margin);
    return p;
  }
  const V3 l = matT_mul(g.rot, dir);
  V3 r = V3{0.0f, 0.0f, 0.0f};
  if (g.type == G_BOX) {
    const float sx = ccd_sign(l.x), sy = ccd_sign(l.y), sz = ccd_sign(l.z);
    r = V3{sx * g.size.x, sy * g.size.y, sz * g.size.z};
    vid = (sx > 0.0f ? 1 : 0) + (sy > 0.0f ? 2 : 0) + (sz > 0.0f ? 4 : 0);
  } else if (g.type == G_CAPSULE) {
    r = l * g.size.x;
    r.z += ccd_sign(l.z) * g.size.y;
  } else if (g.type == G_ELLIPSOID) {
    r = normalize(V3{l.x * g.size.x, l.y * g.size.y, l.z * g.size.z});
    r = V3{r.x * g.size.x, r.y * g.size.y, r.z * g.size.z};
  } else if (g.type == G_MESH && (!g.graph || g.nvert < 10)) {  // collision_gjk.py:154-169: exhaustive vertex search, the cached vertex first
    float best = -CCD_FLOAT_MAX;
    if (g.index > -1) {
      vid = g.index;
      best = dot(ld3(g.vert + 3 * vid), l);
    }
    for (int i = 0; i < g.nvert; ++i) {
      const float dd = dot(ld3(g.vert + 3 * i), l);
      if (dd > best) {
        best = dd;
        vid = i;
      }
    }
    const_cast<CcdGeom&>(g).cache = vid;
    r = ld3(g.vert + 3 * vid);
  } else if (g.type == G_MESH) {  // collision_gjk.py:170-196: hill climbing on the hull's vertex graph from the cached vertex
    const int numvert = g.graph[0];
    const int *edgeadr = g.graph + 2, *globalid = g.graph + 2 + numvert, *edge = g.graph + 2 + 2 * numvert;
    int prev = -1, imax = g.index > -1 ? g.index : 0;
    float best = dot(l, ld3(g.vert + 3 * globalid[imax]));
    while (imax != prev) {
      prev = imax;
      DBG_COUNT(4, 1);
      for (int i = edgeadr[imax]; edge[i] >= 0; ++i) {
        DBG_COUNT(5, 1);
        const float dd = dot(l, ld3(g.vert + 3 * globalid[edge[i]]));
        if (dd > best) {
          best = dd;
          imax = edge[i];
        }
      }
    }
    const_cast<CcdGeom&>(g).cache = imax;
    vid = globalid[imax];
    r = ld3(g.vert + 3 * vid);
  } else if (g.type == G_CYLINDER) {
    const float dd = sqrtf(l.x * l.x + l.y * l.y);
    if (dd > CCD_MINVAL) {
      const float scl = g.size.x / dd;
      r.x = l.x * scl;
      r.y = l.y * scl;
    }
    r.z = ccd_sign(l.z) * g.size.y;
  }
  V3 out = mat_mul(g.rot, r) + g.pos;
  if (g.margin > 0.0f) out = out + dir * (0.5f * g.margin);
  return out;
}

// group-wide arg-max helpers (every lane receives the result)
template <int G>
DEV float gminf(float v) {
#pragma unroll
  for (int off = G / 2; off >= 1; off >>= 1) v = fminf(v, __shfl_xor(v, off, G));
  return v;
}
template <int G>
DEV int gmini(int v) {
#pragma unroll
  for (int off = G / 2; off >= 1; off >>= 1) v = min(v, __shfl_xor(v, off, G));
  return v;
}
// ccd_support by the G lanes of a group together (all lanes pass the same arguments and receive the same result).  Meshes: the
// neighbours of a hill-climbing step / the vertices of an exhaustive search are spread over the lanes; the serial loops take the FIRST
// index that attains the maximum (strict >), the cached vertex winning ties, and so does this.  Other shapes: closed forms, redundantly.
template <int G>
DEV V3 ccd_support_c(const CcdGeom& g, V3 dir, int& vid, int lig) {
  if (g.type != G_MESH) return ccd_support(g, dir, vid);
  const V3 l = matT_mul(g.rot, dir);
  if (!g.graph || g.nvert < 10) {
    float best = -CCD_FLOAT_MAX;
    int bi = 0x7fffffff;
    for (int i = lig; i < g.nvert; i += G) {
      const float dd = dot(ld3(g.vert + 3 * i), l);
      if (dd > best) {
        best = dd;
        bi = i;
      }
    }
    const float top = gmax<G>(best);
    vid = gmini<G>(best == top ? bi : 0x7fffffff);
    if (g.index > -1 && dot(ld3(g.vert + 3 * g.index), l) >= top) vid = g.index;
    const_cast<CcdGeom&>(g).cache = vid;
  } else {
    const int numvert = g.graph[0], nedge = numvert + 3 * g.graph[1];
    const int *edgeadr = g.graph + 2, *globalid = g.graph + 2 + numvert, *edge = g.graph + 2 + 2 * numvert;
    int prev = -1, imax = g.index > -1 ? g.index : 0;
    float best = dot(l, ld3(g.vert + 3 * globalid[imax]));
    while (imax != prev) {
      prev = imax;
      const int e0 = edgeadr[imax], deg = (imax + 1 < numvert ? edgeadr[imax + 1] : nedge) - e0 - 1;  // (each list ends with -1)
      for (int j0 = 0; j0 < deg; j0 += G) {
        const int j = j0 + lig;
        int nb = 0;
        float dd = -CCD_FLOAT_MAX;
        if (j < deg) {
          nb = edge[e0 + j];
          dd = dot(l, ld3(g.vert + 3 * globalid[nb]));
        }
        const float top = gmax<G>(dd);
        if (top > best) {  // the first neighbour that attains the chunk's maximum (the serial scan keeps the first of equals)
          const unsigned long long hit = gballot<G>(j < deg && dd == top);
          best = top;
          imax = __shfl(nb, __ffsll((long long)hit) - 1, G);
        }
      }
    }
    const_cast<CcdGeom&>(g).cache = imax;
    vid = globalid[imax];
  }
  V3 out = mat_mul(g.rot, ld3(g.vert + 3 * vid)) + g.pos;
  if (g.margin > 0.0f) out = out + dir * (0.5f * g.margin);
  return out;
}
// CG = 0: the one-lane code; CG = lanes of the cooperating group
template <int CG>
DEV V3 ccd_sup(const CcdGeom& g, V3 dir, int& vid, int lig) {
  if constexpr (CG > 0) return ccd_support_c<CG>(g, dir, vid, lig);
  else return ccd_support(g, dir, vid);
}

DEV float ccd_det3(V3 a, V3 b, V3 c) { return dot(a, cross(b, c)); }
DEV int ccd_same_sign(float a, float b) { return (a > 0.0f && b > 0.0f) ? 1 : ((a < 0.0f && b < 0.0f) ? -1 : 0); }
DEV float v3c(V3 a, int k) { return k == 0 ? a.x : (k == 1 ? a.y : a.z); }

DEV V3 ccd_origin_on_line(V3 v1, V3 v2) {
  const V3 df = v2 - v1;
  const float scl = -(dot(v2, df) / dot(df, df));
  return v2 + scl * df;
}
DEV bool ccd_origin_on_plane(V3 v1, V3 v2, V3 v3_, V3& out) {  // true: degenerate triangle
  const V3 d21 = v2 - v1, d31 = v3_ - v1, d32 = v3_ - v2;
  V3 n = cross(d32, d21);
  float nv = dot(n, v2), nn = dot(n, n);
  out = V3{0.0f, 0.0f, 0.0f};
  if (nn == 0.0f) return true;
  if (nv != 0.0f && nn > CCD_MINVAL) { out = (nv / nn) * n; return false; }
  n = cross(d21, d31);
  nv = dot(n, v1);
  nn = dot(n, n);
  if (nn == 0.0f) return true;
  if (nv != 0.0f && nn > CCD_MINVAL) { out = (nv / nn) * n; return false; }
  n = cross(d31, d32);
  nv = dot(n, v3_);
  nn = dot(n, n);
  out = (nv / nn) * n;
  return false;
}

// barycentric coordinates of the simplex point closest to the origin (signed volumes): 1-, 2-, 3-simplex
DEV void ccd_s1d(V3 s1, V3 s2, float& l0, float& l1) {
  const V3 po = ccd_origin_on_line(s1, s2);
  float mu_max = s1.x - s2.x;
  int idx = 0;
  float mu = s1.y - s2.y;
  if (fabsf(mu) >= fabsf(mu_max)) { mu_max = mu; idx = 1; }
  mu = s1.z - s2.z;
  if (fabsf(mu) >= fabsf(mu_max)) { mu_max = mu; idx = 2; }
  const float c1 = v3c(po, idx) - v3c(s2, idx), c2 = v3c(s1, idx) - v3c(po, idx);
  if (ccd_same_sign(mu_max, c1) && ccd_same_sign(mu_max, c2)) { l0 = c1 / mu_max; l1 = c2 / mu_max; }
  else { l0 = 0.0f; l1 = 1.0f; }
}
DEV void ccd_minors(V3 s1, V3 s2, V3 s3, float& mmax, int& x, int& y) {
  const float m14 = s2.y * s3.z - s2.z * s3.y - s1.y * s3.z + s1.z * s3.y + s1.y * s2.z - s1.z * s2.y;
  const float m24 = s2.x * s3.z - s2.z * s3.x - s1.x * s3.z + s1.z * s3.x + s1.x * s2.z - s1.z * s2.x;
  const float m34 = s2.x * s3.y - s2.y * s3.x - s1.x * s3.y + s1.y * s3.x + s1.x * s2.y - s1.y * s2.x;
  const float mu1 = fabsf(m14), mu2 = fabsf(m24), mu3 = fabsf(m34);
  if (mu1 >= mu2 && mu1 >= mu3) { mmax = m14; x = 1; y = 2; }
  else if (mu2 >= mu3) { mmax = m24; x = 0; y = 2; }
  else { mmax = m34; x = 0; y = 1; }
}
DEV void ccd_areas(V3 v1, V3 v2, V3 v3_, V3 p, int x, int y, float& c31, float& c32, float& c33) {
  const float ax = v3c(v1, x), ay = v3c(v1, y), bx = v3c(v2, x), by = v3c(v2, y), cx = v3c(v3_, x), cy = v3c(v3_, y), px = v3c(p, x), py = v3c(p, y);
  c31 = px * by + py * cx + bx * cy - px * cy - py * bx - cx * by;
  c32 = px * cy + py * ax + cx * ay - px * ay - py * cx - ax * cy;
  c33 = px * ay + py * bx + ax * by - px * by - py * ax - bx * ay;
}
DEV void ccd_s2d(V3 s1, V3 s2, V3 s3, float& l0, float& l1, float& l2) {
  V3 po;
  if (ccd_origin_on_plane(s1, s2, s3, po)) {
    ccd_s1d(s1, s2, l0, l1);
    l2 = 0.0f;
    return;
  }
  float mmax, c31, c32, c33;
  int x, y;
  ccd_minors(s1, s2, s3, mmax, x, y);
  ccd_areas(s1, s2, s3, po, x, y, c31, c32, c33);
  const int k1 = ccd_same_sign(mmax, c31), k2 = ccd_same_sign(mmax, c32), k3 = ccd_same_sign(mmax, c33);
  if (k1 && k2 && k3) { l0 = c31 / mmax; l1 = c32 / mmax; l2 = c33 / mmax; return; }
  float dmin = CCD_FLOAT_MAX, a, b;
  l0 = l1 = l2 = 0.0f;
  if (!k1) {
    ccd_s1d(s2, s3, a, b);
    const V3 xx = a * s2 + b * s3;
    l0 = 0.0f; l1 = a; l2 = b;
    dmin = dot(xx, xx);
  }
  if (!k2) {
    ccd_s1d(s1, s3, a, b);
    const V3 xx = a * s1 + b * s3;
    const float dd = dot(xx, xx);
    if (dd < dmin) { l0 = a; l1 = 0.0f; l2 = b; dmin = dd; }
  }
  if (!k3) {
    ccd_s1d(s1, s2, a, b);
    const V3 xx = a * s1 + b * s2;
    const float dd = dot(xx, xx);
    if (dd < dmin) { l0 = a; l1 = b; l2 = 0.0f; }
  }
}
DEV void ccd_s3d(V3 s1, V3 s2, V3 s3, V3 s4, float (&lam)[4]) {
  const float c41 = -ccd_det3(s2, s3, s4), c42 = ccd_det3(s1, s3, s4), c43 = -ccd_det3(s1, s2, s4), c44 = ccd_det3(s1, s2, s3);
  const float mdet = c41 + c42 + c43 + c44;
  const int k1 = ccd_same_sign(mdet, c41), k2 = ccd_same_sign(mdet, c42), k3 = ccd_same_sign(mdet, c43), k4 = ccd_same_sign(mdet, c44);
  if (k1 && k2 && k3 && k4) {
    lam[0] = c41 / mdet; lam[1] = c42 / mdet; lam[2] = c43 / mdet; lam[3] = c44 / mdet;
    return;
  }
  float dmin = CCD_FLOAT_MAX, a, b, c;
  lam[0] = lam[1] = lam[2] = lam[3] = 0.0f;
  if (!k1) {
    ccd_s2d(s2, s3, s4, a, b, c);
    const V3 xx = a * s2 + b * s3 + c * s4;
    const float dd = dot(xx, xx);
    if (dd < dmin) { lam[0] = 0.0f; lam[1] = a; lam[2] = b; lam[3] = c; dmin = dd; }
  }
  if (!k2) {
    ccd_s2d(s1, s3, s4, a, b, c);
    const V3 xx = a * s1 + b * s3 + c * s4;
    const float dd = dot(xx, xx);
    if (dd < dmin) { lam[0] = a; lam[1] = 0.0f; lam[2] = b; lam[3] = c; dmin = dd; }
  }
  if (!k3) {
    ccd_s2d(s1, s2, s4, a, b, c);
    const V3 xx = a * s1 + b * s2 + c * s4;
    const float dd = dot(xx, xx);
    if (dd < dmin) { lam[0] = a; lam[1] = b; lam[2] = 0.0f; lam[3] = c; dmin = dd; }
  }
  if (!k4) {
    ccd_s2d(s1, s2, s3, a, b, c);
    const V3 xx = a * s1 + b * s2 + c * s3;
    const float dd = dot(xx, xx);
    if (dd < dmin) { lam[0] = a; lam[1] = b; lam[2] = c; lam[3] = 0.0f; }
  }
}
DEV V3 ccd_combine(int n, const float (&lam)[4], const V3 (&m)[4]) {
  V3 o = V3{0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i < n) o = o + lam[i] * m[i];
  return o;
}

template <int CG = 0>
DEV void ccd_gjk(float tolerance, int iterations, const CcdGeom& g1, const CcdGeom& g2, V3 x1_0, V3 x2_0, float cutoff, bool is_discrete,
                 GjkOut& res, int lig = 0) {
  int n = 0;
  GJK_TICK_START();
  float lam[4] = {1.0f, 0.0f, 0.0f, 0.0f};
  const float epsilon = is_discrete ? 0.0f : 0.5f * tolerance * tolerance, min_norm = is_discrete ? CCD_MINVAL : tolerance;
  V3 xk = x1_0 - x2_0;
  float xnorm = sqrtf(dot(xk, xk)), xnorm_prev = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    res.s[i] = res.s1[i] = res.s2[i] = V3{0.0f, 0.0f, 0.0f};
    res.i1[i] = res.i2[i] = -1;
  }
  res.separated = false;
  res.dim = 0;
  res.dist = 0.0f;
  res.x1 = res.x2 = V3{0.0f, 0.0f, 0.0f};
  for (int it = 0; it < iterations; ++it) {
    if (xnorm < min_norm || fabsf(xnorm_prev - xnorm) < CCD_MINVAL) break;
    DBG_COUNT(2, 1);
    DBG_MAX(3, it + 1);
    V3 dneg = xk * (1.0f / xnorm);
    if (is_discrete && xnorm < 1e-4f) {
      if (n == 2) {
        const V3 e = res.s[1] - res.s[0];
        const float e2 = dot(e, e);
        if (e2 > CCD_MINVAL2) {
          dneg = dneg - (dot(dneg, e) / e2) * e;
          const float dn = length(dneg);
          if (dn > CCD_MINVAL) dneg = dneg * (1.0f / dn);
        }
      } else if (n == 3) {
        const V3 nr = cross(res.s[1] - res.s[0], res.s[2] - res.s[0]);
        const float nn = length(nr);
        if (nn > CCD_MINVAL) dneg = (ccd_sign(dot(dneg, nr)) / nn) * nr;
      }
    }
    int v1, v2;
    GJK_TICK(3);  // simplex arithmetic
    const V3 p1 = ccd_sup<CG>(g1, -dneg, v1, lig), p2 = ccd_sup<CG>(g2, dneg, v2, lig);
    GJK_TICK(2);  // supports
    const_cast<CcdGeom&>(g1).index = g1.cache;  // collision_gjk.py:675-680 (only meshes read it)
    const_cast<CcdGeom&>(g2).index = g2.cache;
    const V3 sn = p1 - p2;
    // slot n of the simplex (static indexing: the arrays stay in registers)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i == n) {
        res.s1[i] = p1;
        res.s2[i] = p2;
        res.s[i] = sn;
        res.i1[i] = v1;
        res.i2[i] = v2;
      }
    if (dot(xk, xk - sn) < epsilon) break;
    const float lower = dot(xk, sn);
    if ((cutoff == 0.0f && lower > 0.0f) || (cutoff != 0.0f && cutoff < CCD_FLOAT_MAX && lower > 0.0f && lower >= cutoff * xnorm)) {
      res.separated = true;
      res.dim = 0;
      res.dist = CCD_FLOAT_MAX;
      res.x1 = res.x2 = V3{0.0f, 0.0f, 0.0f};  // (an empty GJKResult: the height-field contact selection reads the witness points of separated prisms too)
      return;
    }
    lam[0] = 1.0f;
    lam[1] = lam[2] = lam[3] = 0.0f;
    if (n + 1 == 4) ccd_s3d(res.s[0], res.s[1], res.s[2], res.s[3], lam);
    else if (n + 1 == 3) ccd_s2d(res.s[0], res.s[1], res.s[2], lam[0], lam[1], lam[2]);
    else if (n + 1 == 2) ccd_s1d(res.s[0], res.s[1], lam[0], lam[1]);
    // compact the vertices that still carry weight (a stable, fully unrolled compaction)
    int mcount = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (lam[i] == 0.0f) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j == mcount && j <= i) {
          res.s[j] = res.s[i];
          res.s1[j] = res.s1[i];
          res.s2[j] = res.s2[i];
          res.i1[j] = res.i1[i];
          res.i2[j] = res.i2[i];
          lam[j] = lam[i];
        }
      ++mcount;
    }
    n = mcount;
    if (n < 1) break;
    xk = ccd_combine(n, lam, res.s);
    xnorm_prev = xnorm;
    xnorm = sqrtf(dot(xk, xk));
    if (n == 4) break;
  }
  GJK_TICK(3);
  res.separated = false;
  res.x1 = n == 0 ? x1_0 : ccd_combine(n, lam, res.s1);
  res.x2 = n == 0 ? x2_0 : ccd_combine(n, lam, res.s2);
  if (xnorm > 0.0f) {
    const V3 dir = xk * (1.0f / xnorm);
    int v;
    const V3 p1 = ccd_sup<CG>(g1, -dir, v, lig), p2 = ccd_sup<CG>(g2, dir, v, lig);
    res.separated = dot(xk, p1 - p2) > 0.0f;
  }
  GJK_TICK(4);  // closing support pair
  res.dist = (n == 4 && !res.separated) ? 0.0f : xnorm;
  res.dim = n;
  // The separation test above fires once the lower bound x_k . s reaches cutoff |x_k|; at convergence that bound is |x_k|^2, so a pair
  // at least `cutoff` apart is always reported as separated -- unless the duality-gap test, which comes first in the loop, stops that very
  // iteration, and in float32 the gap is rounding noise at that point: a coin flip.  Only the height-field collider can tell the two
  // outcomes apart (it keeps the witness points of separated prisms in its contact selection); normalise to the separation test's verdict
  // (oracle/ccd.c does the same).
  if (res.separated && xnorm > 0.0f && (cutoff == 0.0f || (cutoff < CCD_FLOAT_MAX && xnorm >= cutoff))) {
    res.dim = 0;
    res.dist = CCD_FLOAT_MAX;
    res.x1 = res.x2 = V3{0.0f, 0.0f, 0.0f};
  }
}

// ---- EPA polytope in the lane-interleaved workspace ------------------------------------------------------------------------------
struct Poly {
  float* base;  // word 0 of this lane
  int stride;   // words between consecutive entries: CCD_LANES in the lane-interleaved global workspace, 1 in the cooperative LDS copy
  int vcap, fcap, o_vidx, o_face, o_fpr, o_fn2, o_hor;
  int status, nvert, nface, nhorizon;
  V3 center;
  DEV float& F(int k) const { return base[(size_t)k * stride]; }
  DEV int& I(int k) const { return reinterpret_cast<int*>(base)[(size_t)k * stride]; }
  DEV V3 vert(int i) const { return V3{F(3 * i), F(3 * i + 1), F(3 * i + 2)}; }  // vertex pair v: 2v on geom 1, 2v + 1 on geom 2
  DEV void set_vert(int i, V3 p) const { F(3 * i) = p.x; F(3 * i + 1) = p.y; F(3 * i + 2) = p.z; }
  DEV V3 diff(int v) const { return vert(2 * v) - vert(2 * v + 1); }
  DEV int& vidx(int i) const { return I(o_vidx + i); }
  DEV int& face(int f) const { return I(o_face + f); }  // 10 bits per vertex | bit 31 deleted | bit 30 invalid
  DEV V3 fpr(int f) const { return V3{F(o_fpr + 3 * f), F(o_fpr + 3 * f + 1), F(o_fpr + 3 * f + 2)}; }
  DEV float& fn2(int f) const { return F(o_fn2 + f); }
  DEV int& hor(int i) const { return I(o_hor + i); }
};
DEV void poly_init(Poly& pt, float* base, int iterations, int stride = CCD_LANES) {
  pt.base = base;
  pt.stride = stride;
  pt.vcap = 5 + iterations;
  pt.fcap = 6 + CCD_EPAFACES * iterations;
  pt.o_vidx = 6 * pt.vcap;
  pt.o_face = pt.o_vidx + 2 * pt.vcap;
  pt.o_fpr = pt.o_face + pt.fcap;
  pt.o_fn2 = pt.o_fpr + 3 * pt.fcap;
  pt.o_hor = pt.o_fn2 + pt.fcap;
  pt.status = pt.nvert = pt.nface = pt.nhorizon = 0;
  pt.center = V3{0.0f, 0.0f, 0.0f};
}
DEV float poly_attach_face_at(Poly& pt, int idx, int v1, int v2, int v3_);
DEV float poly_attach_face(Poly& pt, int idx, int v1, int v2, int v3_) {
  if (pt.nface == pt.fcap) return 0.0f;
  return poly_attach_face_at(pt, idx, v1, v2, v3_);
}
DEV float poly_attach_face_at(Poly& pt, int idx, int v1, int v2, int v3_) {  // (no capacity check: the caller owns slot idx)
  const V3 p1 = pt.diff(v1), p2 = pt.diff(v2), p3 = pt.diff(v3_);
  V3 r;
  if (ccd_origin_on_plane(p3, p2, p1, r)) return 0.0f;
  if (dot(r, p1 - pt.center) < 0.0f) r = -r;
  pt.face(idx) = v1 | (v2 << 10) | (v3_ << 20);
  pt.F(pt.o_fpr + 3 * idx) = r.x;
  pt.F(pt.o_fpr + 3 * idx + 1) = r.y;
  pt.F(pt.o_fpr + 3 * idx + 2) = r.z;
  const float n2 = dot(r, r);
  pt.fn2(idx) = n2;
  return n2;
}
template <int CG = 0>
DEV void poly_support(Poly& pt, int idx, const CcdGeom& g1, const CcdGeom& g2, V3 dir, int lig = 0) {
  int v1, v2;
  pt.set_vert(2 * idx, ccd_sup<CG>(g1, dir, v1, lig));
  pt.set_vert(2 * idx + 1, ccd_sup<CG>(g2, -dir, v2, lig));
  pt.vidx(2 * idx) = v1;
  pt.vidx(2 * idx + 1) = v2;
}
DEV void poly_put(Poly& pt, int v, const GjkOut& res, int i) {  // simplex vertex i -> polytope vertex pair v (i static)
  pt.set_vert(2 * v, res.s1[i]);
  pt.set_vert(2 * v + 1, res.s2[i]);
  pt.vidx(2 * v) = res.i1[i];
  pt.vidx(2 * v + 1) = res.i2[i];
}
DEV void poly_replace_simplex3(const Poly& pt, int v1, int v2, int v3_, GjkOut& res) {
  const int v[3] = {v1, v2, v3_};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    res.s1[i] = pt.vert(2 * v[i]);
    res.s2[i] = pt.vert(2 * v[i] + 1);
    res.s[i] = res.s1[i] - res.s2[i];
    res.i1[i] = pt.vidx(2 * v[i]);
    res.i2[i] = pt.vidx(2 * v[i] + 1);
  }
  res.dim = 3;
}
DEV bool ccd_same_side(V3 p0, V3 p1, V3 p2, V3 p3) {
  const V3 n = cross(p1 - p0, p2 - p0);
  const float d1 = dot(n, p3 - p0), d2 = dot(n, -p0);
  return (d1 > 0.0f && d2 > 0.0f) || (d1 < 0.0f && d2 < 0.0f);
}
DEV bool ccd_test_tetra(V3 p0, V3 p1, V3 p2, V3 p3) {
  return ccd_same_side(p0, p1, p2, p3) && ccd_same_side(p1, p2, p3, p0) && ccd_same_side(p2, p3, p0, p1) && ccd_same_side(p3, p0, p1, p2);
}
DEV V3 ccd_tri_affine(V3 v1, V3 v2, V3 v3_, V3 p) {
  float mmax, c31, c32, c33;
  int x, y;
  ccd_minors(v1, v2, v3_, mmax, x, y);
  ccd_areas(v1, v2, v3_, p, x, y, c31, c32, c33);
  return V3{c31 / mmax, c32 / mmax, c33 / mmax};
}
DEV bool ccd_tri_point_intersect(V3 v1, V3 v2, V3 v3_, V3 p) {
  const V3 l = ccd_tri_affine(v1, v2, v3_, p);
  if (l.x < 0.0f || l.y < 0.0f || l.z < 0.0f) return false;
  const V3 pr = l.x * v1 + l.y * v2 + l.z * v3_;
  return length(pr - p) < CCD_MINVAL;
}
DEV int ccd_ray_triangle(V3 v1, V3 v2, V3 v3_, V3 v4, V3 v5) {
  const V3 e = v2 - v1;
  const float vol1 = ccd_det3(v3_ - v1, v4 - v1, e), vol2 = ccd_det3(v4 - v1, v5 - v1, e), vol3 = ccd_det3(v5 - v1, v3_ - v1, e);
  if (vol1 >= 0.0f && vol2 >= 0.0f && vol3 >= 0.0f) return 1;
  if (vol1 <= 0.0f && vol2 <= 0.0f && vol3 <= 0.0f) return -1;
  return 0;
}

// seed polytopes from GJK's 1-, 2-, 3-simplex; status 0: ready, -1: continue from the 2-simplex written into res, > 0: no depth
template <int CG = 0>
DEV void poly_seed2(Poly& pt, GjkOut& res, const CcdGeom& g1, const CcdGeom& g2, int lig = 0) {
  const V3 df = res.s[1] - res.s[0];
  pt.center = 0.5f * (res.s[0] + res.s[1]);
  int index = 0;
  float val = fabsf(df.x);
  if (fabsf(df.y) < val) { val = fabsf(df.y); index = 1; }
  if (fabsf(df.z) < val) index = 2;
  const V3 e = V3{index == 0 ? 1.0f : 0.0f, index == 1 ? 1.0f : 0.0f, index == 2 ? 1.0f : 0.0f};
  const V3 d1 = cross(e, df);
  const float n = length(df), u1 = df.x / n, u2 = df.y / n, u3 = df.z / n, sn = 0.86602540378f, cs = -0.5f;
  const float R[9] = {cs + u1 * u1 * (1 - cs),      u1 * u2 * (1 - cs) - u3 * sn, u1 * u3 * (1 - cs) + u2 * sn,
                      u2 * u1 * (1 - cs) + u3 * sn, cs + u2 * u2 * (1 - cs),      u2 * u3 * (1 - cs) - u1 * sn,
                      u1 * u3 * (1 - cs) - u2 * sn, u2 * u3 * (1 - cs) + u1 * sn, cs + u3 * u3 * (1 - cs)};
  const V3 d2 = mat_mul(R, d1), d3 = mat_mul(R, d2);
  poly_put(pt, 0, res, 0);
  poly_put(pt, 1, res, 1);
  poly_support<CG>(pt, 2, g1, g2, d1 * (1.0f / length(d1)), lig);
  poly_support<CG>(pt, 3, g1, g2, d2 * (1.0f / length(d2)), lig);
  poly_support<CG>(pt, 4, g1, g2, d3 * (1.0f / length(d3)), lig);
  const int Fc[6][3] = {{0, 2, 3}, {0, 4, 2}, {0, 3, 4}, {1, 3, 2}, {1, 2, 4}, {1, 4, 3}};
  for (int f = 0; f < 6; ++f)
    if (poly_attach_face(pt, f, Fc[f][0], Fc[f][1], Fc[f][2]) < CCD_MIN_DIST2) {
      pt.status = -1;
      poly_replace_simplex3(pt, Fc[f][0], Fc[f][1], Fc[f][2], res);
      return;
    }
  if (!ccd_ray_triangle(res.s[0], res.s[1], pt.diff(2), pt.diff(3), pt.diff(4))) {
    pt.status = 1;
    return;
  }
  pt.nvert = 5;
  pt.nface = 6;
  pt.status = 0;
}
template <int CG = 0>
DEV void poly_seed3(Poly& pt, const GjkOut& res, const CcdGeom& g1, const CcdGeom& g2, int lig = 0) {
  pt.center = (res.s[0] + res.s[1] + res.s[2]) * (1.0f / 3.0f);
  V3 n = cross(res.s[1] - res.s[0], res.s[2] - res.s[0]);
  const float norm = length(n);
  if (norm < CCD_MINVAL) { pt.status = 2; return; }
  n = n * (1.0f / norm);
  poly_put(pt, 0, res, 0);
  poly_put(pt, 1, res, 1);
  poly_put(pt, 2, res, 2);
  poly_support<CG>(pt, 3, g1, g2, -n, lig);
  poly_support<CG>(pt, 4, g1, g2, n, lig);
  const V3 v4 = pt.diff(3), v5 = pt.diff(4);
  if (ccd_tri_point_intersect(res.s[0], res.s[1], res.s[2], v4)) { pt.status = 3; return; }
  if (ccd_tri_point_intersect(res.s[0], res.s[1], res.s[2], v5)) { pt.status = 4; return; }
  if (res.dist > 1e-5f && !ccd_test_tetra(res.s[0], res.s[1], res.s[2], v4) && !ccd_test_tetra(res.s[0], res.s[1], res.s[2], v5)) {
    pt.status = 5;
    return;
  }
  const int Fc[6][3] = {{4, 0, 1}, {4, 2, 0}, {4, 1, 2}, {3, 1, 0}, {3, 0, 2}, {3, 2, 1}};
  for (int f = 0; f < 6; ++f)
    if (poly_attach_face(pt, f, Fc[f][0], Fc[f][1], Fc[f][2]) < CCD_MIN_DIST3) { pt.status = 6 + f; return; }
  pt.nvert = 5;
  pt.nface = 6;
  pt.status = 0;
}
DEV void poly_seed4(Poly& pt, GjkOut& res) {
  pt.center = 0.25f * (res.s[0] + res.s[1] + res.s[2] + res.s[3]);
  poly_put(pt, 0, res, 0);
  poly_put(pt, 1, res, 1);
  poly_put(pt, 2, res, 2);
  poly_put(pt, 3, res, 3);
  const int Fc[4][3] = {{0, 1, 2}, {0, 3, 1}, {0, 2, 3}, {3, 2, 1}};
  float dist[4];
  int idx = 0;
  for (int f = 0; f < 4; ++f) {
    dist[f] = poly_attach_face(pt, f, Fc[f][0], Fc[f][1], Fc[f][2]);
    if (dist[f] < CCD_MIN_DIST4) {
      pt.status = -1;
      poly_replace_simplex3(pt, Fc[f][0], Fc[f][1], Fc[f][2], res);
      return;
    }
    if (f > 0 && dist[f] < dist[idx]) idx = f;
  }
  if (!ccd_test_tetra(res.s[0], res.s[1], res.s[2], res.s[3])) {
    if (dist[idx] > CCD_MINVAL) { pt.status = 12; return; }
    pt.status = -1;
    poly_replace_simplex3(pt, Fc[idx][0], Fc[idx][1], Fc[idx][2], res);
    return;
  }
  pt.nvert = 4;
  pt.nface = 4;
  pt.status = 0;
}
DEV int poly_add_edge(Poly& pt, int e1, int e2) {
  const int n = pt.nhorizon;
  if (n < 0) return -1;
  const int edge = (min(e1, e2) << 10) | max(e1, e2);
  for (int i = 0; i < n; ++i)
    if (pt.hor(i) == edge) {
      pt.hor(i) = pt.hor(n - 1);
      return n - 1;
    }
  if (n == CCD_MAX_HORIZON) return -1;
  pt.hor(n) = edge;
  return n + 1;
}
#define CCD_FACE_DELETED 0x80000000u
#define CCD_FACE_INVALID 0x40000000u
DEV void poly_delete_face(Poly& pt, int f) {
  const unsigned fc = (unsigned)pt.face(f);
  pt.face(f) = (int)(fc | CCD_FACE_DELETED);
  pt.nhorizon = poly_add_edge(pt, fc & 0x3FF, (fc >> 10) & 0x3FF);
  pt.nhorizon = poly_add_edge(pt, (fc >> 10) & 0x3FF, (fc >> 20) & 0x3FF);
  pt.nhorizon = poly_add_edge(pt, (fc >> 20) & 0x3FF, fc & 0x3FF);
}
// returns the closest face (-1: no contact); dist <= 0 and the witness points of the penetration
template <int CG = 0>
DEV int ccd_epa(float tolerance, int iterations, Poly& pt, const CcdGeom& g1, const CcdGeom& g2, bool is_discrete, int& overflow, float& dist,
                V3& x1, V3& x2, int lig = 0) {
  static_assert(CG == 0 || CG >= CCD_MAX_HORIZON, "the horizon's faces are attached by one lane each");
  float upper = CCD_FLOAT_MAX, upper2 = CCD_FLOAT_MAX;
  const float epsilon = is_discrete ? CCD_MIN_EPATOL : tolerance;
  int idx = -1, nvalid = pt.nface;
  for (int it = 0; it < iterations; ++it) {
    const int pidx = idx;
    idx = -1;
    float lower2 = CCD_FLOAT_MAX;
    if constexpr (CG > 0) {  // nearest live face: lane-strided scan, then the smallest index among the lanes that hold the minimum
      int bi = 0x7fffffff;
      for (int i = lig; i < pt.nface; i += CG) {
        const float n2 = pt.fn2(i);
        if (!((unsigned)pt.face(i) & (CCD_FACE_DELETED | CCD_FACE_INVALID)) && n2 < lower2) {
          bi = i;
          lower2 = n2;
        }
      }
      const float low = gminf<CG>(lower2);
      const int first = gmini<CG>(lower2 == low ? bi : 0x7fffffff);
      lower2 = low;
      idx = first == 0x7fffffff ? -1 : first;
    } else {
    for (int i = 0; i < pt.nface; ++i) {
      const float n2 = pt.fn2(i);
      if (!((unsigned)pt.face(i) & (CCD_FACE_DELETED | CCD_FACE_INVALID)) && n2 < lower2) {
        idx = i;
        lower2 = n2;
      }
    }
    }
    if (lower2 > upper2 || idx < 0) {
      idx = pidx;
      break;
    }
    if (lower2 <= 0.0f) break;
    const float lower = sqrtf(lower2);
    const int wi = pt.nvert;
    const V3 fpr = pt.fpr(idx);
    poly_support<CG>(pt, wi, g1, g2, fpr * (1.0f / lower), lig);
    const_cast<CcdGeom&>(g1).index = g1.cache;  // collision_gjk.py:1370-1373
    const_cast<CcdGeom&>(g2).index = g2.cache;
    const V3 w = pt.diff(wi);
    pt.nvert++;
    const float upper_k = dot(fpr, w) / lower;
    if (upper_k < upper) {
      upper = upper_k;
      upper2 = upper * upper;
    }
    if (upper - lower < epsilon) break;
    if (is_discrete) {
      bool rep = false;
      const int a = pt.vidx(2 * wi), b = pt.vidx(2 * wi + 1);
      for (int i = 0; i < pt.nvert - 1 && !rep; ++i) rep = pt.vidx(2 * i) == a && pt.vidx(2 * i + 1) == b;
      if (rep) break;
    }
    nvalid--;
    poly_delete_face(pt, idx);
    if (pt.nhorizon == -1) {
      overflow |= OVF_EPA_HORIZON;
      idx = -1;
      break;
    }
    if constexpr (CG > 0) {
      // faces visible from the new vertex: the test is spread over the lanes, the deletions (horizon edges toggle in a 24-entry list)
      // follow in ascending face order like the serial scan
      gsync();
      for (int i0 = 0; i0 < pt.nface && idx != -1; i0 += CG) {
        const int i = i0 + lig;
        bool vis = false;
        if (i < pt.nface && !((unsigned)pt.face(i) & CCD_FACE_DELETED)) vis = dot(pt.fpr(i), w) - pt.fn2(i) > 1e-10f;
        unsigned long long vm = gballot<CG>(vis);
        while (vm) {
          const int f = i0 + __ffsll((long long)vm) - 1;
          vm &= vm - 1;
          if (!((unsigned)pt.face(f) & CCD_FACE_INVALID)) nvalid--;
          poly_delete_face(pt, f);
          if (pt.nhorizon == -1) {
            overflow |= OVF_EPA_HORIZON;
            idx = -1;
            break;
          }
        }
      }
      // one lane per horizon edge attaches its face (slot nface + lane); the serial loop stops at the first degenerate face
      const int nh = pt.nhorizon;
      gsync();
      float d2 = 1.0f;
      const bool has = lig < nh;
      if (has) {
        const int e = pt.hor(lig);
        d2 = pt.nface + lig < pt.fcap ? poly_attach_face_at(pt, pt.nface + lig, wi, e & 0x3FF, (e >> 10) & 0x3FF) : 0.0f;
      }
      const unsigned long long zm = gballot<CG>(has && d2 == 0.0f);
      const int nadd = nh <= 0 ? 0 : (zm ? __ffsll((long long)zm) - 1 : nh);
      const bool live = has && lig < nadd, ok = live && d2 >= lower2 && d2 <= upper2;
      if (live && !ok) pt.face(pt.nface + lig) = (int)((unsigned)pt.face(pt.nface + lig) | CCD_FACE_INVALID);
      nvalid += __popcll(gballot<CG>(ok));
      pt.nface += nadd;
      if (zm) idx = -1;
      gsync();
    } else {
    for (int i = 0; i < pt.nface; ++i) {
      const unsigned fc = (unsigned)pt.face(i);
      if (fc & CCD_FACE_DELETED) continue;
      if (dot(pt.fpr(i), w) - pt.fn2(i) > 1e-10f) {
        if (!(fc & CCD_FACE_INVALID)) nvalid--;
        poly_delete_face(pt, i);
        if (pt.nhorizon == -1) {
          overflow |= OVF_EPA_HORIZON;
          idx = -1;
          break;
        }
      }
    }
    for (int i = 0; i < pt.nhorizon; ++i) {
      const int e = pt.hor(i);
      const float d2 = poly_attach_face(pt, pt.nface, wi, e & 0x3FF, (e >> 10) & 0x3FF);
      if (d2 == 0.0f) {
        idx = -1;
        break;
      }
      pt.nface++;
      if (d2 >= lower2 && d2 <= upper2) nvalid++;
      else pt.face(pt.nface - 1) = (int)((unsigned)pt.face(pt.nface - 1) | CCD_FACE_INVALID);
    }
    }
    if (nvalid == 0 || idx == -1) break;
    pt.nhorizon = 0;
  }
  if (idx < 0) {
    dist = 0.0f;
    return -1;
  }
  const unsigned fc = (unsigned)pt.face(idx);
  const int f0 = fc & 0x3FF, f1 = (fc >> 10) & 0x3FF, f2 = (fc >> 20) & 0x3FF;
  const V3 l = ccd_tri_affine(pt.diff(f0), pt.diff(f1), pt.diff(f2), pt.fpr(idx));
  x1 = l.x * pt.vert(2 * f0) + l.y * pt.vert(2 * f1) + l.z * pt.vert(2 * f2);
  x2 = l.x * pt.vert(2 * f0 + 1) + l.y * pt.vert(2 * f1 + 1) + l.z * pt.vert(2 * f2 + 1);
  dist = -sqrtf(pt.fn2(idx));
  return idx;
}

// gjk_phase + epa_phase: number of contacts (0 / 1), distance between the margin-inflated shapes and the witness points
// face_out: the closest EPA face when the pair qualifies for multi-contact recovery (two boxes, zero margin), else -1
// collision_gjk.py:109 _discrete_geoms
DEV bool ccd_is_discrete(const CcdGeom& g1, const CcdGeom& g2) {
  return (g1.type == G_BOX || g1.type == G_MESH || g1.type == G_HFIELD) && (g2.type == G_BOX || g2.type == G_MESH || g2.type == G_HFIELD) && g1.margin == 0.0f && g2.margin == 0.0f;
}
// gjk_phase (collision_gjk.py:2350-2418).  Returns 1: done -- dist_out / x1 / x2 are the result (CCD_FLOAT_MAX: separated) --, 2: the
// pair penetrates and EPA must run from the simplex in `res`; g1 / g2 leave with their margins and sizes restored and the mesh vertex
// caches (`index`) of the last support calls: the state epa_phase continues from.
template <int CG = 0>
DEV int ccd_gjk_phase(float tolerance, float cutoff, int gjk_iterations, CcdGeom& g1, CcdGeom& g2, float& dist_out, V3& x1, V3& x2, GjkOut& res, int lig = 0) {
  const CcdGeom o1 = g1, o2 = g2;
  float full1 = 0.0f, full2 = 0.0f, size1 = 0.0f, size2 = 0.0f;
  const bool is_discrete = ccd_is_discrete(g1, g2);
  if (g1.type == G_SPHERE || g1.type == G_CAPSULE) {
    size1 = g1.size.x;
    full1 = size1 + 0.5f * g1.margin;
    g1.margin = 0.0f;
    g1.size.x = 0.0f;
  }
  if (g2.type == G_SPHERE || g2.type == G_CAPSULE) {
    size2 = g2.size.x;
    full2 = size2 + 0.5f * g2.margin;
    g2.margin = 0.0f;
    g2.size.x = 0.0f;
  }
  if (size1 + size2 > 0.0f) {
    cutoff += full1 + full2;
    ccd_gjk<CG>(tolerance, gjk_iterations, g1, g2, g1.pos, g2.pos, cutoff, is_discrete, res, lig);
    if (res.dist > tolerance) {
      dist_out = res.dist;
      x1 = res.x1;
      x2 = res.x2;
      if (res.dist == CCD_FLOAT_MAX) return 1;
      const V3 n = normalize(res.x2 - res.x1);
      if (full1 > 0.0f) x1 = x1 + full1 * n;
      if (full2 > 0.0f) x2 = x2 - full2 * n;
      dist_out = res.dist - (full1 + full2);
      return 1;
    }
    g1.margin = o1.margin;  /* (the cached mesh vertex of the first run stays: collision_gjk.py:2403-2406 restores margin and size only) */
    g1.size = o1.size;
    g2.margin = o2.margin;
    g2.size = o2.size;
    cutoff -= full1 + full2;
  }
  ccd_gjk<CG>(tolerance, gjk_iterations, g1, g2, g1.pos, g2.pos, cutoff, is_discrete, res, lig);
  dist_out = res.dist;
  x1 = res.x1;
  x2 = res.x2;
  if (res.dist > tolerance || res.dim < 2 || res.separated) return 1;
  return 2;
}
// epa_phase (collision_gjk.py:2421-2526) from the state ccd_gjk_phase left (g1, g2, res; dist_out / x1 / x2 hold GJK's result and are
// returned unchanged when the polytope cannot be seeded).  Returns the number of contacts (0 / 1); face_out: the closest EPA face when
// the pair qualifies for multi-contact recovery, else -1.  CG > 0: by the CG lanes of a group together (polytope stride 1 in LDS).
template <int CG = 0>
DEV int ccd_epa_phase(float tolerance, int epa_iterations, const CcdGeom& g1, const CcdGeom& g2, GjkOut& res, float* scratch, float& dist_out, V3& x1, V3& x2,
                      int& overflow, int& face_out, Poly& pt, int lig = 0, int pstride = CCD_LANES) {
  face_out = -1;
  const bool is_discrete = ccd_is_discrete(g1, g2);
  poly_init(pt, scratch, epa_iterations, pstride);
  if (res.dim == 2) poly_seed2<CG>(pt, res, g1, g2, lig);
  else if (res.dim == 4) poly_seed4(pt, res);
  if (res.dim == 3) {
    pt.status = 0;
    poly_seed3<CG>(pt, res, g1, g2, lig);
  }
  if (pt.status) return 1;
  float dist;
  const int idx = ccd_epa<CG>(tolerance, epa_iterations, pt, g1, g2, is_discrete, overflow, dist, x1, x2, lig);
  if (idx == -1) {
    dist_out = CCD_FLOAT_MAX;
    return 0;
  }
  dist_out = dist;
  if (g1.margin == 0.0f && g2.margin == 0.0f && (g1.type == G_BOX || g1.type == G_MESH) && (g2.type == G_BOX || g2.type == G_MESH)) face_out = idx;  // collision_gjk.py:2517-2523
  return 1;
}
// gjk_phase + epa_phase by one lane (the height-field prisms; every pair before round 4)
DEV int ccd_run(float tolerance, float cutoff, int gjk_iterations, int epa_iterations, CcdGeom g1, CcdGeom g2, float* scratch, float& dist_out,
                V3& x1, V3& x2, int& overflow, int& face_out, Poly& pt) {
  face_out = -1;
  GjkOut res;
  if (ccd_gjk_phase(tolerance, cutoff, gjk_iterations, g1, g2, dist_out, x1, x2, res) == 1) return 1;
  return ccd_epa_phase(tolerance, epa_iterations, g1, g2, res, scratch, dist_out, x1, x2, overflow, face_out, pt);
}

// ---- multi-contact recovery for box pairs (collision_gjk.py:2076-2300, box branches) -------------------------------------------------
// From the EPA face closest to the origin: the features (vertex / edge / face) of the two boxes it was built from, the box faces whose
// normals oppose each other within FACE_TOL (or an edge perpendicular to a face within EDGE_TOL), then the clipping of one face (or
// edge) against the side planes of the other, pruned to the quadrilateral of largest area.
#define CCD_FACE_TOL 0.99999872f
#define CCD_EDGE_TOL 0.0015999993f
#define CCD_INTERSECT_TOL 0.0000003f
DEV V3 mc_axis(int k, float v) { return V3{k == 0 ? v : 0.0f, k == 1 ? v : 0.0f, k == 2 ? v : 0.0f}; }
DEV V3 mc_face_normal(int i) { return mc_axis(i >> 1, (i & 1) ? -1.0f : 1.0f); }
DEV int mc_feature_dim(const Poly& pt, const int (&face)[3], int offset, int (&fi)[3], V3 (&fv)[3]) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    fi[k] = pt.vidx(2 * face[k] + offset);
    fv[k] = pt.vert(2 * face[k] + offset);
  }
  if (fi[0] != fi[1]) return (fi[2] == fi[0] || fi[2] == fi[1]) ? 2 : 3;
  fi[1] = fi[2];
  fv[1] = fv[2];
  return fi[0] != fi[2] ? 2 : 1;
}
DEV int mc_box_normals2(const float* mat, V3 n, V3 (&nout)[3], int (&iout)[3]) {
  const V3 ln = normalize(matT_mul(mat, n));
  for (int i = 0; i < 6; ++i)
    if (dot(ln, mc_face_normal(i)) > CCD_FACE_TOL) {
      nout[0] = mat_mul(mat, mc_face_normal(i));
      iout[0] = i;
      return 1;
    }
  return 0;
}
DEV int mc_box_normals(int dim, const int (&fi)[3], const float* mat, V3 dir, V3 (&nout)[3], int (&iout)[3]) {
  const int v1 = fi[0], v2 = fi[1], v3_ = fi[2];
  if (dim == 3) {
    int c = 0;
    float ax[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int b = 1 << k;
      ax[k] = (float)((v1 & b) && (v2 & b) && (v3_ & b)) - (float)(!(v1 & b) && !(v2 & b) && !(v3_ & b));
    }
    nout[0] = mat_mul(mat, V3{ax[0], ax[1], ax[2]});
    const float sgn = ax[0] + ax[1] + ax[2];
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (ax[k] != 0.0f) {
        if (c == 0) iout[0] = 2 * k;
        else if (c == 1) iout[1] = 2 * k;
        else iout[2] = 2 * k;
        ++c;
      }
    if (sgn == -1.0f) iout[0] = iout[0] + 1;
    if (c == 1) return 1;
    return mc_box_normals2(mat, dir, nout, iout);
  }
  if (dim == 2) {
    int c = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int b = 1 << k;
      const float a = (float)((v1 & b) && (v2 & b)) - (float)(!(v1 & b) && !(v2 & b));
      if (a != 0.0f) {
        const V3 nn = mat_mul(mat, mc_axis(k, a));
        const int id = a > 0.0f ? 2 * k : 2 * k + 1;
        if (c == 0) { nout[0] = nn; iout[0] = id; }
        else if (c == 1) { nout[1] = nn; iout[1] = id; }
        else { nout[2] = nn; iout[2] = id; }
        ++c;
      }
    }
    if (c == 1 || c == 2) return c;
    return mc_box_normals2(mat, dir, nout, iout);
  }
  if (dim == 1) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float e = (v1 & (1 << k)) ? 1.0f : -1.0f;
      nout[k] = mat_mul(mat, mc_axis(k, e));
      iout[k] = e > 0.0f ? 2 * k : 2 * k + 1;
    }
    return 3;
  }
  return 0;
}
DEV int mc_box_edge_normals(int dim, const CcdGeom& g, V3 v1, V3 v2, int v1i, V3 (&nout)[3], V3 (&endv)[3]) {
  if (dim == 2) {
    endv[0] = v2;
    nout[0] = normalize(v2 - v1);
    return 1;
  }
  if (dim == 1) {
    const float c[3] = {(v1i & 1) ? g.size.x : -g.size.x, (v1i & 2) ? g.size.y : -g.size.y, (v1i & 4) ? g.size.z : -g.size.z};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const V3 a = V3{k == 0 ? -c[0] : c[0], k == 1 ? -c[1] : c[1], k == 2 ? -c[2] : c[2]};
      endv[k] = mat_mul(g.rot, a) + g.pos;
      nout[k] = normalize(endv[k] - v1);
    }
    return 3;
  }
  return 0;
}
DEV int mc_box_face(const CcdGeom& g, int idx, V3 (&face)[4]) {
  if (idx < 0 || idx > 5) return 0;
  // corner signs of the six faces in the reference's vertex order (collision_gjk.py:1848-1888)
  const float S[6][4][3] = {{{1, 1, 1}, {1, 1, -1}, {1, -1, -1}, {1, -1, 1}},     {{-1, 1, -1}, {-1, 1, 1}, {-1, -1, 1}, {-1, -1, -1}},
                            {{-1, 1, -1}, {1, 1, -1}, {1, 1, 1}, {-1, 1, 1}},     {{-1, -1, 1}, {1, -1, 1}, {1, -1, -1}, {-1, -1, -1}},
                            {{-1, 1, 1}, {1, 1, 1}, {1, -1, 1}, {-1, -1, 1}},     {{1, 1, -1}, {-1, 1, -1}, {-1, -1, -1}, {1, -1, -1}}};
  for (int i = 0; i < 4; ++i) face[i] = mat_mul(g.rot, V3{S[idx][i][0] * g.size.x, S[idx][i][1] * g.size.y, S[idx][i][2] * g.size.z}) + g.pos;
  return 4;
}
DEV float mc_area4(V3 a, V3 b, V3 c, V3 d) { return 0.5f * length(cross(a - d, d - b) + cross(b - c, c - a)); }
template <class Get>
DEV void mc_polygon_quad_g(Get&& P, int n, int (&res)[4]) {
  int b = 1, c = 2, d = 3;
  res[0] = 0; res[1] = b; res[2] = c; res[3] = d;
  float m = mc_area4(P(0), P(b), P(c), P(d));
  for (int a = 0; a < n; ++a) {
    for (;;) {
      float mn = mc_area4(P(a), P(b), P(c), P((d + 1) % n));
      if (mn <= m) break;
      m = mn;
      d = (d + 1) % n;
      res[0] = a; res[1] = b; res[2] = c; res[3] = d;
      for (;;) {
        mn = mc_area4(P(a), P(b), P((c + 1) % n), P(d));
        if (mn <= m) break;
        m = mn;
        c = (c + 1) % n;
        res[0] = a; res[1] = b; res[2] = c; res[3] = d;
      }
      for (;;) {
        mn = mc_area4(P(a), P((b + 1) % n), P(c), P(d));
        if (mn <= m) break;
        m = mn;
        b = (b + 1) % n;
        res[0] = a; res[1] = b; res[2] = c; res[3] = d;
      }
    }
    if (b == a) {
      b = (b + 1) % n;
      if (c == b) {
        c = (c + 1) % n;
        if (d == c) d = (d + 1) % n;
      }
    }
  }
}
DEV void mc_polygon_quad(const V3* poly, int n, int (&res)[4]) {
  mc_polygon_quad_g([&](int i) { return poly[i]; }, n, res);
}
// clip polygon face2 against the side planes of face1 (normal n): w2 = clipped points, w1 = w2 - dir; returns the number of contacts
DEV int mc_polygon_clip(const V3 (&face1)[4], int nface1, const V3 (&face2)[4], int nface2, V3 n, V3 dir, V3 (&w1)[4], V3 (&w2)[4]) {
  if (nface1 < 3) return 0;
  V3 pn[4], bufa[8], bufb[8];
  float pd[4];
  V3* poly = bufa;
  V3* clip = bufb;
  for (int i = 0; i < nface1; ++i) {
    const V3 a = face1[i], b = face1[(i + 1) % nface1];
    pn[i] = cross(b - a, (a + n) - a);
    pd[i] = dot(pn[i], a);
  }
  int np = nface2, nc = 0;
  for (int i = 0; i < nface2; ++i) poly[i] = face2[i];
  for (int e = 0; e < nface1; ++e) {
    for (int i = 0; i < np; ++i) {
      const V3 P = poly[i], Q = poly[(i + 1) % np];
      const bool in1 = dot(P - face1[e], pn[e]) > -1e-10f, in2 = dot(Q - face1[e], pn[e]) > -1e-10f;
      if (!in1 && !in2) continue;
      if (in1 && in2) {
        if (nc < 8) clip[nc] = Q;
        ++nc;
        continue;
      }
      const V3 pq = Q - P;
      const float dt = dot(pn[e], pq);
      float t = fabsf(dt) < 1e-10f ? CCD_FLOAT_MAX : (pd[e] - dot(pn[e], P)) / dt;
      if (t > -CCD_INTERSECT_TOL && t < 1.0f + CCD_INTERSECT_TOL) {
        t = clampf(t, 0.0f, 1.0f);
        if (nc < 8) clip[nc] = P + t * pq;
        ++nc;
      }
      if (in2) {
        if (nc < 8) clip[nc] = Q;
        ++nc;
      }
    }
    if (nc > 8) nc = 8;
    V3* tmp = poly;
    poly = clip;
    clip = tmp;
    np = nc;
    nc = 0;
  }
  if (np < 1) return 0;
  if (nface2 == 2 && np > 2) {
    int b1 = 0, b2 = 1;
    float maxd = 0.0f;
    for (int i = 0; i < np; ++i)
      for (int j = i + 1; j < np; ++j) {
        const V3 df = poly[j] - poly[i];
        const float d2 = dot(df, df);
        if (d2 > maxd) {
          maxd = d2;
          b1 = i;
          b2 = j;
        }
      }
    w2[0] = poly[b1];
    w1[0] = w2[0] - dir;
    w2[1] = poly[b2];
    w1[1] = w2[1] - dir;
    return 2;
  }
  if (np > 4) {
    int q[4];
    mc_polygon_quad(poly, np, q);
    for (int i = 0; i < 4; ++i) {
      w2[i] = poly[q[i]];
      w1[i] = w2[i] - dir;
    }
    return 4;
  }
  for (int i = 0; i < np; ++i) {
    w2[i] = poly[i];
    w1[i] = w2[i] - dir;
  }
  return np;
}
DEV int ccd_multicontact_box(const Poly& pt, int epa_face, V3 x1, V3 x2, const CcdGeom& g1, const CcdGeom& g2, V3 (&w1)[4], V3 (&w2)[4]) {
  DBG_TICK_START();
  w1[0] = x1;
  w2[0] = x2;
  const unsigned fc = (unsigned)pt.face(epa_face);
  const int face[3] = {(int)(fc & 0x3FF), (int)((fc >> 10) & 0x3FF), (int)((fc >> 20) & 0x3FF)};
  int fi1[3], fi2[3], idx1[3] = {0, 0, 0}, idx2[3] = {0, 0, 0};
  V3 fv1[3], fv2[3], n1[3], n2[3], endv[3];
  const int nf1 = mc_feature_dim(pt, face, 0, fi1, fv1), nf2 = mc_feature_dim(pt, face, 1, fi2, fv2);
  const V3 dir = x2 - x1;
  int nn1 = mc_box_normals(nf1, fi1, g1.rot, -dir, n1, idx1), nn2 = mc_box_normals(nf2, fi2, g2.rot, dir, n2, idx2);
  bool edge1 = false, edge2 = false, found = false;
  int ri = 0, rj = 0;
  for (int i = 0; i < nn1 && !found; ++i)
    for (int j = 0; j < nn2 && !found; ++j)
      if (dot(n1[i], n2[j]) < -CCD_FACE_TOL) {
        ri = i;
        rj = j;
        found = true;
      }
  if (!found) {
    if (nf1 < 3 && nf1 <= nf2) {
      nn1 = mc_box_edge_normals(nf1, g1, fv1[0], fv1[1], fi1[0], n1, endv);
      for (int i = 0; i < nn2 && !found; ++i)
        for (int j = 0; j < nn1 && !found; ++j)
          if (fabsf(dot(n1[j], n2[i])) < CCD_EDGE_TOL) {
            ri = j;
            rj = i;
            found = true;
          }
      if (!found) return 1;
      edge1 = true;
    } else if (nf2 < 3) {
      nn2 = mc_box_edge_normals(nf2, g2, fv2[0], fv2[1], fi2[0], n2, endv);
      for (int i = 0; i < nn1 && !found; ++i)
        for (int j = 0; j < nn2 && !found; ++j)
          if (fabsf(dot(n2[j], n1[i])) < CCD_EDGE_TOL) {
            ri = j;
            rj = i;
            found = true;
          }
      if (!found) return 1;
      edge2 = true;
    } else {
      return 1;
    }
  }
  V3 face1[4], face2[4];
  int nface1, nface2;
  if (edge1) {
    face1[0] = pt.vert(2 * face[0]);
    face1[1] = endv[ri];
    nface1 = 2;
  } else {
    nface1 = mc_box_face(g1, edge2 ? idx1[rj] : idx1[ri], face1);
  }
  if (edge2) {
    face2[0] = pt.vert(2 * face[0] + 1);
    face2[1] = endv[ri];
    nface2 = 2;
  } else {
    nface2 = mc_box_face(g2, idx2[rj], face2);
  }
  const float dn = length(dir);
  if (edge1) return mc_polygon_clip(face2, nface2, face1, nface1, n2[rj], (-dn) * n2[rj], w2, w1);
  if (edge2) return mc_polygon_clip(face1, nface1, face2, nface2, n1[rj], (-dn) * n1[rj], w1, w2);
  return mc_polygon_clip(face1, nface1, face2, nface2, n1[ri], dn * n2[rj], w1, w2);
}

// ---- multi-contact recovery with mesh faces (collision_gjk.py:1556-1700, 1891-1913, 2076-2300) ------------------------------------------
// Pairs box-mesh / mesh-mesh, unless DisableBit.MULTICCD.  Same flow as ccd_multicontact_box with the mesh polygon tables of the Model
// (types.py:1707-1733) in place of the box's closed forms.  Like the reference (collision_convex.py:1346-1366) the feature buffers are
// sized from the MODEL -- npolygonmax vertices per polygon, nmeshdegmax polygons around a vertex (aloha_pot: 76 / 43) -- and live in
// the lane's slice of Data.ws_ccd behind the EPA polytope (word k of lane l at k * CCD_LANES + l, like the polytope itself); a separate
// non-inlined function keeps it out of the register budget of the narrowphase that calls it (it runs for the few pairs whose EPA face
// is a face-face / edge-face contact).
__host__ __device__ inline int ccd_mc_p(int npolygonmax) { return npolygonmax > 4 ? npolygonmax : 4; }
__host__ __device__ inline int ccd_mc_d(int nmeshdegmax) { return nmeshdegmax > 3 ? nmeshdegmax : 3; }
__host__ __device__ inline int ccd_mc_words(int npolygonmax, int nmeshdegmax) {  // idx1 idx2 | n1 n2 endv | face1 face2 pn | pd | bufa bufb
  return 11 * ccd_mc_d(nmeshdegmax) + 22 * ccd_mc_p(npolygonmax);
}
struct WsI {  // int array in the multi-contact workspace: entry i at p[i * st] (st = CCD_LANES in a lane-interleaved slice, 1 for an EPA group)
  int* p;
  int st;
  DEV int get(int i) const { return p[(size_t)i * st]; }
  DEV void set(int i, int v) const { p[(size_t)i * st] = v; }
};
struct WsF {
  float* p;
  int st;
  DEV float get(int i) const { return p[(size_t)i * st]; }
  DEV void set(int i, float v) const { p[(size_t)i * st] = v; }
};
struct WsV {  // V3 array
  float* p;
  int st;
  DEV V3 get(int i) const { return V3{p[(size_t)(3 * i) * st], p[(size_t)(3 * i + 1) * st], p[(size_t)(3 * i + 2) * st]}; }
  DEV void set(int i, V3 v) const {
    p[(size_t)(3 * i) * st] = v.x;
    p[(size_t)(3 * i + 1) * st] = v.y;
    p[(size_t)(3 * i + 2) * st] = v.z;
  }
};
DEV int mc_intersect(const int* a1, int n1, const int* a2, int n2, int (&res)[2]) {
  int count = 0;
  for (int i = 0; i < n1; ++i)
    for (int j = 0; j < n2; ++j)
      if (a1[i] == a2[j]) {
        res[count++] = a1[i];
        if (count == 2) return 2;
      }
  return count;
}
// the same by the cg lanes of a group together (identical arguments in every lane): a lane per (i, j) pair, matches taken in the serial
// loop's order -- the lists are table reads of two dependent loads per element, n1 x n2 of them one after the other for the serial loop
template <class A1, class A2>
DEV int mc_intersect_g(A1&& a1, int n1, A2&& a2, int n2, int (&res)[2], int lig, int cg) {
  int count = 0;
  const int total = n1 * n2, g0 = (int)(threadIdx.x & 63) - lig;  // first lane of the group in its wavefront
  const unsigned long long gmask = cg >= 64 ? ~0ull : ((1ull << cg) - 1ull);
  for (int base = 0; base < total && count < 2; base += cg) {
    const int k = base + lig;
    int val = 0;
    bool hit = false;
    if (k < total) {
      const int i = k / n2;
      val = a1(i);
      hit = val == a2(k - i * n2);
    }
    unsigned long long b = (__ballot(hit) >> g0) & gmask;
    while (b && count < 2) {
      res[count++] = __shfl(val, g0 + __ffsll((long long)b) - 1, 64);
      b &= b - 1ull;
    }
  }
  return count;
}
struct MeshTab {  // the polygon tables of one mesh, offset to it
  const float* vert;
  const float* polynormal;
  const int *polyvertadr, *polyvertnum, *polyvert, *polymapadr, *polymapnum, *polymap;
};
DEV MeshTab mesh_tab(const MjhModel& m, const CcdGeom& g) {
  const int pa = m.mesh_polyadr[g.meshid], va = m.mesh_vertadr[g.meshid];
  return MeshTab{g.vert, m.mesh_polynormal + 3 * pa, m.mesh_polyvertadr + pa, m.mesh_polyvertnum + pa, m.mesh_polyvert, m.mesh_polymapadr + va, m.mesh_polymapnum + va, m.mesh_polymap};
}
// (lig, cg: lane of a cooperating group and its width -- the lists of a high-degree vertex, one dependent pair of table loads per entry, are
// gathered one entry per lane; 0, 1: serial.  The caller fences before it reads.)
DEV int mc_mesh_normals(int dim, const int (&fi)[3], const MeshTab& t, const float* rot, int cap, const WsV& nout, const WsI& iout, int lig = 0, int cg = 1) {
  const int* m1 = t.polymap + t.polymapadr[fi[0]];
  const int n1 = t.polymapnum[fi[0]];
  if (dim == 3) {
    int edgeset[2] = {0, 0}, faceset[2] = {0, 0};
    const int *m2 = t.polymap + t.polymapadr[fi[1]], *m3 = t.polymap + t.polymapadr[fi[2]];
    const int n2 = t.polymapnum[fi[1]], n3 = t.polymapnum[fi[2]];
    int n;
    if (cg > 1) n = mc_intersect_g([&](int i) { return m1[i]; }, n1, [&](int j) { return m2[j]; }, n2, edgeset, lig, cg);
    else n = mc_intersect(m1, n1, m2, n2, edgeset);
    if (n == 0) return 0;
    if (cg > 1) n = mc_intersect_g([&](int i) { return i == 0 ? edgeset[0] : edgeset[1]; }, n, [&](int j) { return m3[j]; }, n3, faceset, lig, cg);
    else n = mc_intersect(edgeset, n, m3, n3, faceset);
    if (n == 0) return 0;
    nout.set(0, mat_mul(rot, ld3(t.polynormal + 3 * faceset[0])));
    iout.set(0, faceset[0]);
    return 1;
  }
  if (dim == 2) {
    int edgeset[2] = {0, 0};
    const int* m2 = t.polymap + t.polymapadr[fi[1]];
    const int n2 = t.polymapnum[fi[1]];
    const int n = cg > 1 ? mc_intersect_g([&](int i) { return m1[i]; }, n1, [&](int j) { return m2[j]; }, n2, edgeset, lig, cg) : mc_intersect(m1, n1, m2, n2, edgeset);
    for (int i = 0; i < n; ++i) {
      nout.set(i, mat_mul(rot, ld3(t.polynormal + 3 * edgeset[i])));
      iout.set(i, edgeset[i]);
    }
    return n;
  }
  if (dim == 1) {
    const int n = min(n1, cap);
    for (int i = lig; i < n; i += cg) {
      const int pi = m1[i];
      nout.set(i, mat_mul(rot, ld3(t.polynormal + 3 * pi)));
      iout.set(i, pi);
    }
    return n;
  }
  return 0;
}
DEV int mc_mesh_edge_normals(int dim, const MeshTab& t, const CcdGeom& g, V3 v1, V3 v2, int v1i, int cap, const WsV& nout, const WsV& endv, int lig = 0, int cg = 1) {
  if (dim == 2) {
    endv.set(0, v2);
    nout.set(0, normalize(v2 - v1));
    return 1;
  }
  if (dim == 1) {
    const int* pm = t.polymap + t.polymapadr[v1i];
    const int n = min(t.polymapnum[v1i], cap);
    for (int i = lig; i < n; i += cg) {
      const int adr = t.polyvertadr[pm[i]], nv = t.polyvertnum[pm[i]];
      for (int j = 0; j < nv; ++j)
        if (t.polyvert[adr + j] == v1i) {
          const int k = j == 0 ? nv - 1 : j - 1;
          const V3 e = mat_mul(g.rot, ld3(t.vert + 3 * t.polyvert[adr + k])) + g.pos;
          endv.set(i, e);
          nout.set(i, normalize(e - v1));
        }
    }
    return n;
  }
  return 0;
}
DEV int mc_mesh_face(const MeshTab& t, const CcdGeom& g, int idx, int cap, const WsV& face, int lig = 0, int cg = 1) {
  const int adr = t.polyvertadr[idx], nv = min(t.polyvertnum[idx], cap);
  for (int j = lig; j < nv; j += cg) face.set(j, mat_mul(g.rot, ld3(t.vert + 3 * t.polyvert[adr + nv - 1 - j])) + g.pos);  // (reversed order)
  return nv;
}
// mc_polygon_clip on workspace polygons; cap = 2 * npolygonmax slots per clip buffer (collision_convex.py:1346-1348)
DEV int mc_polygon_clip_ws(const WsV& face1, int nface1, const WsV& face2, int nface2, V3 n, V3 dir, int cap, WsV poly, WsV clip,
                           V3 (&w1)[4], V3 (&w2)[4]) {
  if (nface1 < 3) return 0;
#ifdef MJH_DBG_EPA_CLOCK
  unsigned long long dbg_c0_ = __builtin_amdgcn_s_memtime();
#endif
  // (the side plane of edge e -- normal cross(b - a, (a + n) - a), offset normal . a -- is computed where it is used instead of into a table
  // first: the same arithmetic, 4 P words of workspace less)
  int np = nface2, nc = 0;
  for (int i = 0; i < nface2; ++i) poly.set(i, face2.get(i));
  for (int e = 0; e < nface1; ++e) {
    const V3 fe = face1.get(e), fnx = face1.get((e + 1) % nface1);
    const V3 pne = cross(fnx - fe, (fe + n) - fe);
    const float pde = dot(pne, fe);
    V3 P = np > 0 ? poly.get(0) : V3{0, 0, 0};
    for (int i = 0; i < np; ++i) {
      const V3 Q = poly.get((i + 1) % np);
      const bool in1 = dot(P - fe, pne) > -1e-10f, in2 = dot(Q - fe, pne) > -1e-10f;
      if (in1 && in2) {
        if (nc < cap) clip.set(nc, Q);
        ++nc;
      } else if (in1 || in2) {
        const V3 pq = Q - P;
        const float dt = dot(pne, pq);
        float t = fabsf(dt) < 1e-10f ? CCD_FLOAT_MAX : (pde - dot(pne, P)) / dt;
        if (t > -CCD_INTERSECT_TOL && t < 1.0f + CCD_INTERSECT_TOL) {
          t = clampf(t, 0.0f, 1.0f);
          if (nc < cap) clip.set(nc, P + t * pq);
          ++nc;
        }
        if (in2) {
          if (nc < cap) clip.set(nc, Q);
          ++nc;
        }
      }
      P = Q;
    }
    if (nc > cap) nc = cap;
    const WsV tmp = poly;
    poly = clip;
    clip = tmp;
    np = nc;
    nc = 0;
  }
#ifdef MJH_DBG_EPA_CLOCK
  if ((threadIdx.x & 31) == 0) atomicAdd(g_dbg_clk + 5, (int)((__builtin_amdgcn_s_memtime() - dbg_c0_) >> 4));  // planes + clip loop
  dbg_c0_ = __builtin_amdgcn_s_memtime();
#endif
  if (np < 1) return 0;
  if (nface2 == 2 && np > 2) {
    int b1 = 0, b2 = 1;
    float maxd = 0.0f;
    for (int i = 0; i < np; ++i) {
      const V3 pi = poly.get(i);
      for (int j = i + 1; j < np; ++j) {
        const V3 df = poly.get(j) - pi;
        const float d2 = dot(df, df);
        if (d2 > maxd) {
          maxd = d2;
          b1 = i;
          b2 = j;
        }
      }
    }
    w2[0] = poly.get(b1);
    w1[0] = w2[0] - dir;
    w2[1] = poly.get(b2);
    w1[1] = w2[1] - dir;
    return 2;
  }
  if (np > 4) {
    int q[4];
    mc_polygon_quad_g([&](int i) { return poly.get(i); }, np, q);
    for (int i = 0; i < 4; ++i) {
      w2[i] = poly.get(q[i]);
      w1[i] = w2[i] - dir;
    }
#ifdef MJH_DBG_EPA_CLOCK
    if ((threadIdx.x & 31) == 0) atomicAdd(g_dbg_clk + 6, (int)((__builtin_amdgcn_s_memtime() - dbg_c0_) >> 4));  // pruning to four points
    if ((threadIdx.x & 31) == 0) atomicAdd(g_dbg_clk + 7, np);
#endif
    return 4;
  }
  for (int i = 0; i < np; ++i) {
    w2[i] = poly.get(i);
    w1[i] = w2[i] - dir;
  }
  return np;
}
// ws = the lane's multi-contact words (behind the polytope, the contact cache and the height-field table of Data.ws_ccd)
// (two entry points: inlined into k_ccd_epa -- behind a call every field of `m`, `pt`, `g1`, `g2` is a flat load from the caller's stack: the
// out-of-line copy ran at ~18 cycles per instruction, 70 % of that launch -- and out of line for the one-lane callers of the contact kernel,
// whose register budget it would otherwise raise)
DEV int ccd_multicontact_mesh_inl(const MjhModel& m, const Poly& pt, int epa_face, V3 x1, V3 x2, const CcdGeom& g1, const CcdGeom& g2,
                                  V3 (&w1)[4], V3 (&w2)[4], float* ws, int wst = CCD_LANES, float* lds = nullptr, int lds_face = 0, int lds_lists = -1, int lig = 0, int cg = 1) {
  if (lds && lds_lists >= 0) {  // the feature lists in the group's LDS (ccd_coop_scratch_offset)
    ws = lds + lds_lists;
    wst = 1;
  }
  DBG_TICK_START();
  w1[0] = x1;
  w2[0] = x2;
  const unsigned fc = (unsigned)pt.face(epa_face);
  const int face[3] = {(int)(fc & 0x3FF), (int)((fc >> 10) & 0x3FF), (int)((fc >> 20) & 0x3FF)};
  const bool mesh1 = g1.type == G_MESH, mesh2 = g2.type == G_MESH;
  const MeshTab t1 = mesh1 ? mesh_tab(m, g1) : MeshTab{}, t2 = mesh2 ? mesh_tab(m, g2) : MeshTab{};
  const int D = ccd_mc_d(m.nmeshdegmax), P = ccd_mc_p(m.npolygonmax);
  auto at = [&](int word) { return ws + (size_t)word * wst; };
  const WsI idx1{reinterpret_cast<int*>(at(0)), wst}, idx2{reinterpret_cast<int*>(at(D)), wst};
  const WsV n1{at(2 * D), wst}, n2{at(5 * D), wst}, endv{at(8 * D), wst};
  const int f0 = 11 * D;
  // an EPA group (lds != nullptr): the polygon buffers in the group's LDS (ccd_coop_words) -- the clip loop and the pruning of the clipped
  // polygon are serial chains of reads and writes of these buffers (ALOHA pot on the table: a 76-gon), in global memory 70 % of k_ccd_epa
  const WsV face1 = lds ? WsV{lds + lds_face, 1} : WsV{at(f0), wst}, face2 = lds ? WsV{lds + lds_face + 3 * P, 1} : WsV{at(f0 + 3 * P), wst};
  const WsV bufa = lds ? WsV{lds, 1} : WsV{at(f0 + 10 * P), wst}, bufb = lds ? WsV{lds + 6 * P, 1} : WsV{at(f0 + 16 * P), wst};
  int fi1[3], fi2[3];
  V3 fv1[3], fv2[3];
  const int nf1 = mc_feature_dim(pt, face, 0, fi1, fv1), nf2 = mc_feature_dim(pt, face, 1, fi2, fv2);
  const V3 dir = x2 - x1;
  auto box_normals = [&](int dim, const int (&fi)[3], const float* rot, V3 d, const WsV& nout, const WsI& iout) {
    V3 nb[3];
    int ib[3] = {0, 0, 0};
    const int n = mc_box_normals(dim, fi, rot, d, nb, ib);
    for (int k = 0; k < n; ++k) {
      nout.set(k, nb[k]);
      iout.set(k, ib[k]);
    }
    return n;
  };
  auto box_edge_normals = [&](int dim, const CcdGeom& g, V3 v1, V3 v2, int v1i, const WsV& nout, const WsV& ev) {
    V3 nb[3], eb[3];
    const int n = mc_box_edge_normals(dim, g, v1, v2, v1i, nb, eb);
    for (int k = 0; k < n; ++k) {
      nout.set(k, nb[k]);
      ev.set(k, eb[k]);
    }
    return n;
  };
  auto box_face = [&](const CcdGeom& g, int idx, const WsV& f) {
    V3 fb[4];
    const int n = mc_box_face(g, idx, fb);
    for (int k = 0; k < n; ++k) f.set(k, fb[k]);
    return n;
  };
  int nn1 = mesh1 ? mc_mesh_normals(nf1, fi1, t1, g1.rot, D, n1, idx1, lig, cg) : box_normals(nf1, fi1, g1.rot, -dir, n1, idx1);
  int nn2 = mesh2 ? mc_mesh_normals(nf2, fi2, t2, g2.rot, D, n2, idx2, lig, cg) : box_normals(nf2, fi2, g2.rot, dir, n2, idx2);
  gsync();  // (the lists were gathered a lane per entry)
  DBG_TICK(3);  // normals
  bool edge1 = false, edge2 = false, found = false;
  int ri = 0, rj = 0;
  for (int i = 0; i < nn1 && !found; ++i) {
    const V3 a = n1.get(i);
    for (int j = 0; j < nn2 && !found; ++j)
      if (dot(a, n2.get(j)) < -CCD_FACE_TOL) {
        ri = i;
        rj = j;
        found = true;
      }
  }
  if (!found) {
    if (nf1 < 3 && nf1 <= nf2) {
      nn1 = mesh1 ? mc_mesh_edge_normals(nf1, t1, g1, fv1[0], fv1[1], fi1[0], D, n1, endv, lig, cg) : box_edge_normals(nf1, g1, fv1[0], fv1[1], fi1[0], n1, endv);
      gsync();
      for (int i = 0; i < nn2 && !found; ++i) {
        const V3 b = n2.get(i);
        for (int j = 0; j < nn1 && !found; ++j)
          if (fabsf(dot(n1.get(j), b)) < CCD_EDGE_TOL) {
            ri = j;
            rj = i;
            found = true;
          }
      }
      if (!found) return 1;
      edge1 = true;
    } else if (nf2 < 3) {
      nn2 = mesh2 ? mc_mesh_edge_normals(nf2, t2, g2, fv2[0], fv2[1], fi2[0], D, n2, endv, lig, cg) : box_edge_normals(nf2, g2, fv2[0], fv2[1], fi2[0], n2, endv);
      gsync();
      for (int i = 0; i < nn1 && !found; ++i) {
        const V3 a = n1.get(i);
        for (int j = 0; j < nn2 && !found; ++j)
          if (fabsf(dot(n2.get(j), a)) < CCD_EDGE_TOL) {
            ri = j;
            rj = i;
            found = true;
          }
      }
      if (!found) return 1;
      edge2 = true;
    } else {
      return 1;
    }
  }
  int nface1, nface2;
  const V3 end_ri = (edge1 || edge2) ? endv.get(ri) : V3{0.0f, 0.0f, 0.0f};  // (read before a face is written: in LDS the first face covers the end of this list)
  if (edge1) {
    face1.set(0, pt.vert(2 * face[0]));
    face1.set(1, end_ri);
    nface1 = 2;
  } else {
    const int ind = edge2 ? idx1.get(rj) : idx1.get(ri);
    nface1 = mesh1 ? mc_mesh_face(t1, g1, ind, P, face1, lig, cg) : box_face(g1, ind, face1);
  }
  if (edge2) {
    face2.set(0, pt.vert(2 * face[0] + 1));
    face2.set(1, end_ri);
    nface2 = 2;
  } else {
    nface2 = mesh2 ? mc_mesh_face(t2, g2, idx2.get(rj), P, face2, lig, cg) : box_face(g2, idx2.get(rj), face2);
  }
  gsync();  // (the faces were gathered a lane per vertex)
  DBG_TICK(4);  // match + faces
  const float dn = length(dir);
  const int cap = 2 * P;
  if (edge1) {
    const V3 nn = n2.get(rj);
    return mc_polygon_clip_ws(face2, nface2, face1, nface1, nn, (-dn) * nn, cap, bufa, bufb, w2, w1);
  }
  if (edge2) {
    const V3 nn = n1.get(rj);
    return mc_polygon_clip_ws(face1, nface1, face2, nface2, nn, (-dn) * nn, cap, bufa, bufb, w1, w2);
  }
  return mc_polygon_clip_ws(face1, nface1, face2, nface2, n1.get(ri), dn * n2.get(rj), cap, bufa, bufb, w1, w2);
}

DEV bool is_convex_pair(int t1, int t2) {
  if (t2 == G_MESH && t1 >= G_SPHERE) return true;  // every mesh pair except plane-mesh (primitive) goes through GJK / EPA
  return (t1 == G_SPHERE && t2 == G_ELLIPSOID) || (t1 == G_CAPSULE && (t2 == G_ELLIPSOID || t2 == G_CYLINDER)) ||
         (t1 == G_ELLIPSOID && (t2 == G_ELLIPSOID || t2 == G_CYLINDER || t2 == G_BOX)) || (t1 == G_CYLINDER && (t2 == G_CYLINDER || t2 == G_BOX));
}
