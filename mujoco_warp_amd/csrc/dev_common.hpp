// dev_common.hpp -- device-side math and lane-group primitives shared by all kernels (gfx950, wave64).
//
// Execution model used by every kernel in this engine: one *lane group* of G lanes (G in {8,16,32,64},
// always inside one wavefront) owns one world.  A 256-thread workgroup therefore advances 256/G worlds.
// Per-world state lives in a private LDS slice; lanes of a group cooperate through it and through
// DPP/permute cross-lane ops.  Because a group never spans wavefronts no s_barrier is needed: LDS
// operations of one wave execute in program order, so a compiler-level fence suffices (gsync()).
//
// Math conventions follow the reference (/root/reference/mujoco_warp/_src/math.py): quaternions (w,x,y,z),
// row-major world-from-local 3x3, spatial vectors (angular[3], linear[3]), 10-vector inertias.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/mjhip.h"

#define MJ_MINVAL 1e-15f
#define MJ_MAXVAL 1e10f
#define MJ_MINIMP 0.0001f
#define MJ_MAXIMP 0.9999f
#define MJ_MINMU 1e-5f
#define MJ_PI 3.14159265358979323846f

enum { JNT_FREE = 0, JNT_BALL = 1, JNT_SLIDE = 2, JNT_HINGE = 3 };
enum { G_PLANE = 0, G_HFIELD = 1, G_SPHERE = 2, G_CAPSULE = 3, G_ELLIPSOID = 4, G_CYLINDER = 5, G_BOX = 6, G_MESH = 7 };
enum { ST_SATISFIED = 0, ST_QUADRATIC = 1, ST_LINEARNEG = 2, ST_LINEARPOS = 3, ST_CONE = 4 };
enum { CT_EQUALITY = 0, CT_FRICTION_DOF = 1, CT_FRICTION_TENDON = 2, CT_LIMIT_JOINT = 3, CT_LIMIT_TENDON = 4,
       CT_CONTACT_FRICTIONLESS = 5, CT_CONTACT_PYRAMIDAL = 6, CT_CONTACT_ELLIPTIC = 7 };
enum { DSBL_CONSTRAINT = 1 << 0, DSBL_EQUALITY = 1 << 1, DSBL_FRICTIONLOSS = 1 << 2, DSBL_LIMIT = 1 << 3,
       DSBL_CONTACT = 1 << 4, DSBL_SPRING = 1 << 5, DSBL_DAMPER = 1 << 6, DSBL_GRAVITY = 1 << 7,
       DSBL_CLAMPCTRL = 1 << 8, DSBL_WARMSTART = 1 << 9, DSBL_FILTERPARENT = 1 << 10, DSBL_ACTUATION = 1 << 11,
       DSBL_REFSAFE = 1 << 12, DSBL_SENSOR = 1 << 13, DSBL_EULERDAMP = 1 << 15, DSBL_NATIVECCD = 1 << 17, DSBL_MULTICCD = 1 << 19 };
enum { SOL_PGS = 0, SOL_CG = 1, SOL_NEWTON = 2 };
enum { CONE_PYRAMIDAL = 0, CONE_ELLIPTIC = 1 };
enum { ENBL_ENERGY = 1 << 1, ENBL_SLEEP = 1 << 5 };
enum { ISL_WIDE = 1, ISL_MANYROWS = 2 };
enum { ISL_LIST_MANYROWS = 0, ISL_LIST_WIDE = 1, ISL_LIST_GENERIC = 2 };  // Data.ws_isl_list / ws_isl_count classes  // Data.ws_isl_flags: a world holds an island of 33..64 dofs / of <= 32 dofs and > 64 rows
#define CON_STRIDE 32  /* words per contact record in d.ws_contact (layout: collide.hpp) */
// word of a contact record holding friction coefficient j of the reference's Contact.friction[5] (tangent 1, tangent 2, spin, roll 1, roll 2)
#define CON_FRICTION_WORD(j) ((j) == 0 ? 14 : (j) == 1 ? 30 : (j) == 2 ? 15 : (j) == 3 ? 16 : 31)
enum { INT_EULER = 0, INT_RK4 = 1, INT_IMPLICIT = 2, INT_IMPLICITFAST = 3 };
enum { OVF_NEFC = 1 << 0, OVF_NJMAX_NNZ = 1 << 1, OVF_BROADPHASE = 1 << 2, OVF_NARROWPHASE = 1 << 3, OVF_CCD = 1 << 4, OVF_HFIELD = 1 << 5, OVF_NVMAX = 1 << 7, OVF_EPA_HORIZON = 1 << 8, OVF_ITERATIONS = 1 << 9, OVF_LS_ITERATIONS = 1 << 10 };
enum { CONTACT_TYPE_CONSTRAINT = 1 };

#define DEV __device__ __forceinline__

// ---- lane-group primitives -------------------------------------------------------------------
// LDS ops of a single wavefront complete in issue order; this fence only stops the compiler from
// reordering / caching LDS values across the point where lanes exchange data.
DEV void gsync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// DPP reduction steps (VALU only, no LDS traffic)
template <int CTRL, int ROW_MASK, int BANK_MASK>
DEV float dpp_add_f(float v) {
  const int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, true);
  return v + __int_as_float(r);
}
// sum over a 32-lane group, valid in the LAST lane of the group only (lanes 31 / 63): five DPP adds, no broadcast
DEV float gsum_last32(float v) {
  v = dpp_add_f<0x111, 0xf, 0xf>(v);  // row_shr:1
  v = dpp_add_f<0x112, 0xf, 0xf>(v);  // row_shr:2
  v = dpp_add_f<0x114, 0xf, 0xf>(v);  // row_shr:4
  v = dpp_add_f<0x118, 0xf, 0xf>(v);  // row_shr:8
  return dpp_add_f<0x142, 0xa, 0xf>(v);  // row_bcast:15 into rows 1 and 3
}
template <int G>
DEV float gsum(float v) {
#pragma unroll
  for (int off = G / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, G);
  return v;
}
template <int G>
DEV float gmax(float v) {
#pragma unroll
  for (int off = G / 2; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, G));
  return v;
}
template <int G>
DEV int gsumi(int v) {
#pragma unroll
  for (int off = G / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, G);
  return v;
}
// bits of a wave-wide ballot belonging to this lane's group
template <int G>
DEV unsigned long long gballot(bool p) {
  unsigned long long b = __ballot(p);
  if (G == 64) return b;
  const int wl = threadIdx.x & 63;
  const int base = wl & ~(G - 1);
  return (b >> base) & ((1ull << (G & 63)) - 1ull);  // (G == 64 returned above)
}
// ordered compaction helper: rank of this lane among set lanes of its group, and group total
template <int G>
DEV int grank(bool p, int lig, int& total) {
  unsigned long long bits = gballot<G>(p);
  total = __popcll(bits);
  return __popcll(bits & ((1ull << lig) - 1ull));
}

// batched ("*") model field row: ptr + (w % nb) * stride   (reference types.py:1535-1808)
DEV const float* bf(const float* p, int nb, int w, int stride) { return p + (size_t)(nb > 1 ? (w % nb) : 0) * stride; }

// ---- small vector math ----------------------------------------------------------------------------
struct V3 {
  float x, y, z;
};
struct Q4 {
  float w, x, y, z;
};
DEV V3 v3(float x, float y, float z) { return V3{x, y, z}; }
DEV V3 ld3(const float* p) { return V3{p[0], p[1], p[2]}; }
DEV void st3(float* p, V3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
DEV Q4 ld4(const float* p) { return Q4{p[0], p[1], p[2], p[3]}; }
DEV void st4(float* p, Q4 a) { p[0] = a.w; p[1] = a.x; p[2] = a.y; p[3] = a.z; }
DEV V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
DEV V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
DEV V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
DEV V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
DEV V3 operator*(float s, V3 a) { return V3{a.x * s, a.y * s, a.z * s}; }
DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
DEV float length(V3 a) { return sqrtf(dot(a, a)); }
DEV V3 normalize(V3 a) {  // wp.normalize: zero stays zero
  float n = length(a);
  return n > 0.0f ? a * (1.0f / n) : a;
}
DEV V3 normalize_with_norm(V3 a, float& n) {
  n = length(a);
  return n == 0.0f ? a : a * (1.0f / n);
}
DEV float safe_div(float x, float y) { return x / (y != 0.0f ? y : MJ_MINVAL); }
DEV float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

DEV Q4 mul_quat(Q4 u, Q4 v) {  // math.py:24
  return Q4{u.w * v.w - u.x * v.x - u.y * v.y - u.z * v.z, u.w * v.x + u.x * v.w + u.y * v.z - u.z * v.y,
            u.w * v.y - u.x * v.z + u.y * v.w + u.z * v.x, u.w * v.z + u.x * v.y - u.y * v.x + u.z * v.w};
}
DEV V3 rot_vec_quat(V3 v, Q4 q) {  // math.py:46
  V3 u = V3{q.x, q.y, q.z};
  float s = q.w;
  V3 r = (2.0f * dot(u, v)) * u + (s * s - dot(u, u)) * v;
  return r + (2.0f * s) * cross(u, v);
}
DEV Q4 axis_angle_to_quat(V3 axis, float angle) {  // math.py:54
  float s, c;
  sincosf(angle * 0.5f, &s, &c);  // precise version: FK accuracy matters
  return Q4{c, axis.x * s, axis.y * s, axis.z * s};
}
DEV Q4 quat_normalize(Q4 q) {  // MuJoCo C mju_normalize4: a (near-)zero quaternion is the identity (keyframes of aloha_pot store 0 0 0 0)
  float n = sqrtf(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  if (n >= MJ_MINVAL) {
    float inv = 1.0f / n;
    return Q4{q.w * inv, q.x * inv, q.y * inv, q.z * inv};
  }
  return Q4{1.0f, 0.0f, 0.0f, 0.0f};
}
DEV void quat_to_mat(Q4 q, float* m) {  // math.py:61
  float q00 = q.w * q.w, q01 = q.w * q.x, q02 = q.w * q.y, q03 = q.w * q.z;
  float q11 = q.x * q.x, q12 = q.x * q.y, q13 = q.x * q.z, q22 = q.y * q.y, q23 = q.y * q.z, q33 = q.z * q.z;
  m[0] = q00 + q11 - q22 - q33; m[1] = 2.0f * (q12 - q03); m[2] = 2.0f * (q13 + q02);
  m[3] = 2.0f * (q12 + q03); m[4] = q00 - q11 + q22 - q33; m[5] = 2.0f * (q23 - q01);
  m[6] = 2.0f * (q13 - q02); m[7] = 2.0f * (q23 + q01); m[8] = q00 - q11 - q22 + q33;
}
DEV V3 quat_to_vel(Q4 q) {  // math.py:161
  V3 axis = V3{q.x, q.y, q.z};
  float s = length(axis);
  if (s == 0.0f) return V3{0, 0, 0};
  float speed = 2.0f * atan2f(s, q.w);
  if (speed > MJ_PI) speed -= 2.0f * MJ_PI;
  return axis * (speed / s);
}
DEV V3 quat_sub(Q4 qa, Q4 qb) {  // math.py:176
  return quat_to_vel(mul_quat(Q4{qb.w, -qb.x, -qb.y, -qb.z}, qa));
}
DEV Q4 quat_integrate(Q4 q, V3 v, float dt) {  // math.py:189
  float norm = length(v);
  V3 vn = normalize(v);
  Q4 qr = axis_angle_to_quat(vn, dt * norm);
  return quat_normalize(mul_quat(quat_normalize(q), qr));
}
DEV V3 mat_mul(const float* m, V3 v) {
  return V3{m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z};
}
DEV V3 matT_mul(const float* m, V3 v) {
  return V3{m[0] * v.x + m[3] * v.y + m[6] * v.z, m[1] * v.x + m[4] * v.y + m[7] * v.z, m[2] * v.x + m[5] * v.y + m[8] * v.z};
}
// 10-vector spatial inertia times motion 6-vector (math.py:121)
DEV void inert_vec(const float* i, const float* v, float* r) {
  r[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  r[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  r[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  r[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  r[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  r[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
DEV void motion_cross(const float* u, const float* v, float* r) {  // math.py:134
  V3 u0 = ld3(u), u1 = ld3(u + 3), v0 = ld3(v), v1 = ld3(v + 3);
  st3(r, cross(u0, v0));
  st3(r + 3, cross(u1, v0) + cross(u0, v1));
}
DEV void motion_cross_force(const float* v, const float* f, float* r) {  // math.py:148
  V3 v0 = ld3(v), v1 = ld3(v + 3), f0 = ld3(f), f1 = ld3(f + 3);
  st3(r, cross(v0, f0) + cross(v1, f1));
  st3(r + 3, cross(v0, f1));
}
DEV void make_frame(V3 a, float* frame) {  // math.py:203-257
  a = normalize(a);
  V3 y = V3{0, 1, 0}, z = V3{0, 0, 1};
  V3 b = (-0.5f < a.y && a.y < 0.5f) ? y : z;
  b = b - a * dot(a, b);
  b = normalize(b);
  if (length(a) == 0.0f) b = V3{0, 0, 0};
  V3 c = cross(a, b);
  st3(frame, a);
  st3(frame + 3, b);
  st3(frame + 6, c);
}

// cooperative copy of n floats between LDS and global by the G lanes of a group
template <int G>
DEV void gcopy(float* dst, const float* src, int n, int lig) {
  for (int i = lig; i < n; i += G) dst[i] = src[i];
}
template <int G>
DEV void gcopyi(int* dst, const int* src, int n, int lig) {
  for (int i = lig; i < n; i += G) dst[i] = src[i];
}

// The worlds handled by one workgroup.  Plain kernels derive it from blockIdx (blk_of_launch); composite launches
// (mjhip.hip: k_mid, k_solve_plus, k_integrate_plus) give every workgroup a role and a world range of its own.
struct Blk {
  int w0;        // first world (or first schedule slot) of this workgroup
  int nw;        // worlds in this workgroup
  int nthreads;  // = nw * G: threads that take part (block-wide loops and barriers); the rest return at once
};
template <int G>
DEV Blk blk_of_launch() {
  return Blk{(int)(blockIdx.x * (blockDim.x / G)), (int)(blockDim.x / G), (int)blockDim.x};
}

// Phase clock (profiling builds only: hipcc -DMJH_PHASE_CLOCK, tools/phase_clock.py).  Lane 0 of every group adds the
// shader-clock ticks between consecutive marks to g_phase_ticks[kernel][phase]; the product build compiles it away.
#ifdef MJH_PHASE_CLOCK
__device__ unsigned long long g_phase_ticks[64][8][16];  // [copy = block & 63] spreads the flush atomics
struct PhaseClock {
  long long t;
  unsigned acc[16];
  int k;
  bool on;
  DEV PhaseClock(int kernel, int lig) : t(clock64()), k(kernel), on(lig == 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0u;
  }
  DEV void mark(int phase) {  // phase must be a compile-time constant (register-resident accumulators)
    const long long n = clock64();
    acc[phase] += (unsigned)(n - t);
    t = clock64();
  }
  DEV ~PhaseClock() {
    if (!on) return;
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (acc[i]) atomicAdd(&g_phase_ticks[blockIdx.x & 63][k][i], (unsigned long long)acc[i]);
  }
};
#else
struct PhaseClock {
  DEV PhaseClock(int, int) {}
  DEV void mark(int) {}
};
#endif
