// host.hpp -- host-side helpers shared by the translation units of libmjhip.so.
//
// The library is built from several translation units compiled in parallel (the solver kernels are ~70 template
// instantiations and dominate the build time): mjhip.hip (entry points, launch sequencing, every non-solver kernel),
// solve_cg32 / solve_newton32 / solve_cg64 / solve_newton64 .hip and their elliptic-cone twins solve_ell_*.hip (k_solve_plus
// instantiations), pgs_tu.hip (k_solve_pgs) and
// solve_big.hip (k_solve_big).
// Device code is header-only and fully inlined per kernel, so no relocatable device code is needed.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/mjhip.h"

int mjh_fail(int code, const char* fmt, const char* a = "");  // records the message for mjh_last_error (mjhip.hip)
#define fail mjh_fail
#define HIPCHK(expr)                                                      \
  do {                                                                    \
    hipError_t e_ = (expr);                                               \
    if (e_ != hipSuccess) return fail(MJH_E_LAUNCH, #expr ": %s", hipGetErrorString(e_)); \
  } while (0)

static const int kLdsPerCU = 160 * 1024;

// Developer knobs (tuning / A-B switches; none changes results beyond what the parity tests bound).  The MJH_* environment variables are
// read ONCE, when the library is loaded, into a table (mjhip.hip); nothing on the launch path calls getenv.  mjh_dev_knob (include/mjhip.h)
// -- the one documented test hook -- overrides an entry of the table afterwards.  Returns nullptr when the knob is unset.
const char* mjh_knob(const char* name);

// pick threads per block in {256,128,64} maximising resident worlds per CU for the given LDS needs (mjhip.hip)
int pick_block(size_t shared_bytes, size_t per_world_bytes, int G, size_t* lds_out, bool prefer_small_arg = false);

// raise the dynamic-LDS cap of a kernel once (never during stream capture: mjh_graph_create warms up first)
template <typename K>
static hipError_t set_lds(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return hipSuccess;
  static std::mutex mu;
  static std::vector<std::pair<const void*, size_t>> done;
  const void* f = reinterpret_cast<const void*>(kernel);
  std::lock_guard<std::mutex> lock(mu);
  for (auto& p : done)
    if (p.first == f && p.second >= bytes) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) done.emplace_back(f, bytes);
  return e;
}

// developer knob MJH_DEBUG_OCC: print the resident workgroups per CU the runtime computes for a launch and the rounds its grid needs
template <typename K>
static void debug_occupancy(const char* name, K kernel, int grid, int threads, size_t lds) {
  static const bool on = mjh_knob("MJH_DEBUG_OCC") != nullptr;
  if (!on) return;
  int nb = -1;
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, threads, lds);
  fprintf(stderr, "%-22s grid %5d x %3d threads, LDS %6zu B: %2d workgroups per CU = %4.1f wavefronts per SIMD, %.2f rounds on 256 CUs\n", name, grid,
          threads, lds, nb, nb * (threads / 64) / 4.0, nb > 0 ? grid / (256.0 * nb) : 0.0);
}

// solver launches, one translation unit each (nr = rows per lane: 2 / 6 with 32 lanes per world, 1 / 2 / 3 with 64;
// (lo, hi] = row-count range of the worlds this launch solves; fuse_euler: explicit Euler step in the solver epilogue)
int launch_solve_32_cg(const MjhModel* m, const MjhData* d, int nr, bool with_factor, int fuse_euler, hipStream_t s, int lo, int hi);
// one row per lane (worlds of at most 32 rows): solve_ell_newton32_r1.hip
int launch_solve_32_newton_ell_r1(const MjhModel* m, const MjhData* d, bool with_factor, int fuse_euler, hipStream_t s, int lo, int hi);
// CG with one world per wavefront (solver_cgw.hpp): nv <= 32, worlds of at most 64 rows, pyramidal cones
int launch_solve_cgw(const MjhModel* m, const MjhData* d, bool with_factor, int fuse_euler, hipStream_t s, int lo, int hi);
// CG with contact-basis rows in one row pool per workgroup (solver_cgp.hpp): nv <= 32, njmax <= 64, MjhModel.cg_basis; worlds it cannot take
// are flagged solver_niter = -1 for the fallback launch (launch_solve_32_cg_deferred, solve_cg32.hip)
int launch_solve_32_cg_deferred(const MjhModel* m, const MjhData* d, int fuse_euler, hipStream_t s);
int launch_solve_cgp(const MjhModel* m, const MjhData* d, bool with_factor, int fuse_euler, hipStream_t s);
int launch_solve_32_newton(const MjhModel* m, const MjhData* d, int nr, bool with_factor, int fuse_euler, hipStream_t s, int lo, int hi);
int launch_solve_64_cg(const MjhModel* m, const MjhData* d, int nr, bool with_factor, int fuse_euler, hipStream_t s, int lo, int hi);
int launch_solve_64_newton(const MjhModel* m, const MjhData* d, int nr, bool with_factor, int fuse_euler, hipStream_t s, int lo, int hi);
// the same with elliptic friction cones (solve_ell_*.hip)
int launch_solve_32_cg_ell(const MjhModel* m, const MjhData* d, int nr, bool with_factor, int fuse_euler, hipStream_t s, int lo, int hi);
int launch_solve_32_newton_ell(const MjhModel* m, const MjhData* d, int nr, bool with_factor, int fuse_euler, hipStream_t s, int lo, int hi);
int launch_solve_64_cg_ell(const MjhModel* m, const MjhData* d, int nr, bool with_factor, int fuse_euler, hipStream_t s, int lo, int hi);
int launch_solve_64_newton_ell(const MjhModel* m, const MjhData* d, int nr, bool with_factor, int fuse_euler, hipStream_t s, int lo, int hi);
// per-(world, island) solves for nv > 64 (solve_tree_*.hip)
// (s: the common island class; sr, sr2, sr3: the rare classes -- 8..16 dofs, 16..32 dofs, many rows / 33..64 dofs -- which touch disjoint islands)
int launch_solve_tree_cg(const MjhModel* m, const MjhData* d, hipStream_t s, hipStream_t sr, hipStream_t sr2, hipStream_t sr3);
int launch_solve_tree_newton(const MjhModel* m, const MjhData* d, hipStream_t s, hipStream_t sr, hipStream_t sr2, hipStream_t sr3);
int launch_solve_tree_cg_ell(const MjhModel* m, const MjhData* d, hipStream_t s, hipStream_t sr, hipStream_t sr2, hipStream_t sr3);      // (elliptic cones: solve_tree_ell_*.hip)
int launch_solve_tree_newton_ell(const MjhModel* m, const MjhData* d, hipStream_t s, hipStream_t sr, hipStream_t sr2, hipStream_t sr3);
int launch_pgs(const MjhModel* m, const MjhData* d, hipStream_t s);
// generic LDS solver (solver_big.hpp): nv > 64, and the worlds of a small model with more than nefc_lo = 192 rows
int launch_solve_big(const MjhModel* m, const MjhData* d, hipStream_t s, int nefc_lo = -1);
