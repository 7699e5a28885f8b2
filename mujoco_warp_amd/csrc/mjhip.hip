// mjhip.hip -- extern "C" entry points of libmjhip.so (see include/mjhip.h) and launch sequencing.
//
// One `step` = 3-4 composite launches on the caller's stream (reference: ~90 fixed launches + ~12 per solver
// iteration, forward.py:1341-1380):
//   k_fwd_pos_plus (kinematics+com_pos+crb, solver schedule) -> k_mid ({collision -> make_constraint} and fwd_vel
//   workgroups) -> k_solve_plus / k_solve_pgs (the whole solve, riders) -> k_integrate_plus (integrator, riders);
// the stage API launches the same bodies as plain kernels.  See "composite launches" below and DESIGN.md section 3.
// Nothing here allocates or synchronises (hipGraph-capturable); workspace lives in MjhData.
#include "host.hpp"

#include "contact_rec.hpp"
// developer knobs of the k_mid occupancy experiment (round 5, LAB_NOTES.md R5): contacts per staging window, wavefronts per SIMD of the light k_mid
#ifdef MJH_CON_WINDOW
#undef CON_WINDOW
#define CON_WINDOW MJH_CON_WINDOW
#endif
#ifndef MJH_MID_WAVES
#define MJH_MID_WAVES 4
#endif

#include "collide.hpp"
#include "constraint.hpp"
#include "dev_common.hpp"
#include "integrate.hpp"
#include "implicit.hpp"
#include "sensor.hpp"
#include "sleep.hpp"
#include "smooth.hpp"
#include "support.hpp"

static thread_local char g_err[512] = "";
int mjh_fail(int code, const char* fmt, const char* a) {
  snprintf(g_err, sizeof(g_err), fmt, a);
  return code;
}

// ---- developer knobs: a table filled ONCE from the environment when the library is loaded; mjh_dev_knob overrides entries (host.hpp) ----
#include <atomic>
#include <map>
#include <string>
extern char** environ;
namespace {
struct KnobTable {
  std::mutex mu;
  std::map<std::string, std::string> kv;  // (node-based: a value's storage does not move when other keys are added)
  std::atomic<int> n{0};
  KnobTable() {
    for (char** e = environ; e && *e; ++e) {
      if (strncmp(*e, "MJH_", 4) != 0) continue;
      const char* eq = strchr(*e, '=');
      if (eq) kv[std::string(*e, eq - *e)] = std::string(eq + 1);
    }
    n.store((int)kv.size());
  }
};
KnobTable& knobs() {
  static KnobTable t;  // (constructed at first use or by the initialiser below, whichever comes first: before any launch)
  return t;
}
struct KnobInit {
  KnobInit() { knobs(); }
} g_knob_init;  // library load
}  // namespace
const char* mjh_knob(const char* name) {
  KnobTable& t = knobs();
  if (t.n.load(std::memory_order_acquire) == 0) return nullptr;  // the product path: no knob set, no lock, no lookup
  std::lock_guard<std::mutex> lock(t.mu);
  auto it = t.kv.find(name);
  return it == t.kv.end() ? nullptr : it->second.c_str();
}
#define TRY(x)               \
  do {                       \
    int rc_ = (x);           \
    if (rc_ != MJH_OK) return rc_; \
  } while (0)

// pick threads per block in {256,128,64} maximising resident worlds per CU for the given LDS needs
int pick_block(size_t shared_bytes, size_t per_world_bytes, int G, size_t* lds_out, bool prefer_small_arg) {
  static const int small_env = mjh_knob("MJH_SMALL_BLOCKS") ? atoi(mjh_knob("MJH_SMALL_BLOCKS")) : -1;  // developer knob
  const bool prefer_small = small_env >= 0 ? (small_env != 0) : prefer_small_arg;
  int best = 0, best_worlds = -1;
  // ties go to the first candidate: large blocks amortise the block-shared tables, small blocks retire as soon as
  // their own worlds converge (k_solve: per-world work varies by an order of magnitude)
  const int order_big[3] = {256, 128, 64}, order_small[3] = {64, 128, 256};
  for (int oi = 0; oi < 3; ++oi) {
    const int threads = prefer_small ? order_small[oi] : order_big[oi];
    const int wpb = threads / G;
    if (wpb < 1) continue;
    const size_t lds = shared_bytes + per_world_bytes * wpb;
    if (lds > (size_t)kLdsPerCU) continue;
    int blocks = (int)(kLdsPerCU / lds);
    const int wave_cap = 32 / (threads / 64);
    if (blocks > wave_cap) blocks = wave_cap;
    const int worlds = blocks * wpb;
    if (worlds > best_worlds) {
      best_worlds = worlds;
      best = threads;
    }
  }
  if (best) *lds_out = shared_bytes + per_world_bytes * (best / G);
  return best;
}

// Lanes per world of every kernel but the solvers (which choose their own): 32 -- two worlds per wavefront -- for models of at most 32 dofs
// and bodies, 64 beyond (round 3): these kernels are chains of dependent steps whose loops stride over dofs / bodies / geoms by the
// group size, so a model that needs two trips with 32 lanes needs one with 64 (three humanoids, nv 81: k_mid 690 -> 518 us, step + 22 %).
static inline bool lanes64(const MjhModel* m) {
  static const int force = mjh_knob("MJH_LANES") ? atoi(mjh_knob("MJH_LANES")) : 0;  // developer knob: 32 / 64
  if (force) return force == 64;
  // (models with GJK / EPA pairs stay at 32: Data.ws_ccd holds one polytope workspace per lane of a 32-lane group)
  return (m->nv > 32 || m->nbody > 32) && !m->heavy_colliders;
}
// 16 lanes per world -- four worlds per wavefront -- in k_mid for small models (at most 16 dofs and bodies, light colliders; round 3): half the
// wavefronts for the same worlds and still one trip per loop.  Same-box A/B: Panda 8192 worlds k_mid 62.0 -> 49.5 us, step 180 -> 164.6 us
// (- 8.7 %).  Not for larger models (humanoid, 27 dofs on 17 bodies, forced with MJH_LANES16_ANY: k_mid 84 -> 101 us -- every loop needs two
// trips) and not for k_fwd_pos (Panda 48.0 vs 49.1 us: its chain is the tree depth whatever the lane count; MJH_LANES16_POS turns it on).
static inline bool lanes16(const MjhModel* m) {
  static const int force = mjh_knob("MJH_LANES") ? atoi(mjh_knob("MJH_LANES")) : 0;  // developer knob: 16 / 32 / 64
  static const bool any = mjh_knob("MJH_LANES16_ANY") != nullptr;
  if (force && force != 16) return false;
  return ((m->nv <= 16 && m->nbody <= 16) || (force == 16 && any)) && !m->heavy_colliders;
}
// (k_fwd_pos: its loops run over bodies -- the G1, 35 dofs on 30 bodies, is 4 us slower with 64 lanes; three humanoids, 52 bodies, 25 us faster)

template <int G>
static int launch_pos_g(const MjhModel* m, const MjhData* d, int first, int last, hipStream_t s) {
  const PosLayout lay = pos_layout(m->nq, m->nv, m->nbody, m->njnt, m->nC, last >= POS_FACTOR);
  size_t lds;
  const int threads = pick_block(sizeof(int) * pos_shared_words(m->nv, m->nC, m->nbody, m->njnt, m->nbodylevel, m->ngeom, m->nsite), sizeof(float) * lay.total, G, &lds);
  if (!threads) return fail(MJH_E_UNSUPPORTED, "k_fwd_pos: model does not fit in LDS");
  HIPCHK(set_lds(k_fwd_pos<G>, lds));
  const int wpb = threads / G;
  hipLaunchKernelGGL(k_fwd_pos<G>, dim3((d->nworld + wpb - 1) / wpb), dim3(threads), lds, s, *m, *d, first, last);
  return MJH_OK;
}
static int launch_pos(const MjhModel* m, const MjhData* d, int first, int last, hipStream_t s) { return lanes64(m) && m->nbody > 32 ? launch_pos_g<64>(m, d, first, last, s) : launch_pos_g<32>(m, d, first, last, s); }
template <int G>
static int launch_vel_g(const MjhModel* m, const MjhData* d, int first, int last, hipStream_t s) {
  const VelLayout lay = vel_layout(m->nq, m->nv, m->nbody, m->nC, m->nu);
  size_t lds;
  const int threads = pick_block(sizeof(int) * mstruct_ints(m->nv, m->nC), sizeof(float) * lay.total, G, &lds);
  if (!threads) return fail(MJH_E_UNSUPPORTED, "k_fwd_vel: model does not fit in LDS");
  HIPCHK(set_lds(k_fwd_vel<G>, lds));
  const int wpb = threads / G;
  hipLaunchKernelGGL(k_fwd_vel<G>, dim3((d->nworld + wpb - 1) / wpb), dim3(threads), lds, s, *m, *d, first, last);
  return MJH_OK;
}
static int launch_vel(const MjhModel* m, const MjhData* d, int first, int last, hipStream_t s) { return lanes64(m) ? launch_vel_g<64>(m, d, first, last, s) : launch_vel_g<32>(m, d, first, last, s); }
// public L'DL factor (qLD, qLDiagInv) and, on request, qacc_smooth: outputs nobody inside the step waits for
template <int G>
static int launch_factor_smooth_g(const MjhModel* m, const MjhData* d, int write_qacc, hipStream_t s) {
  const FacLayout lay = fac_layout(m->nv, m->nC);
  size_t lds;
  const int threads = pick_block(sizeof(int) * mstruct_ints(m->nv, m->nC), sizeof(float) * lay.total, G, &lds);
  if (!threads) return fail(MJH_E_UNSUPPORTED, "k_factor_smooth: model does not fit in LDS");
  HIPCHK(set_lds(k_factor_smooth<G>, lds));
  const int wpb = threads / G;
  hipLaunchKernelGGL(k_factor_smooth<G>, dim3((d->nworld + wpb - 1) / wpb), dim3(threads), lds, s, *m, *d, write_qacc);
  return MJH_OK;
}
static int launch_factor_smooth(const MjhModel* m, const MjhData* d, int write_qacc, hipStream_t s) { return launch_factor_smooth_g<32>(m, d, write_qacc, s); }  // (chains over the sparse factor: more lanes per world only halve the worlds per wavefront)
// the three launches of the convex narrowphase in front of a contact kernel (models with GJK pairs; csrc/convex.hpp header)
template <int G>
static int launch_ccd_pre(const MjhModel* m, const MjhData* d, hipStream_t s) {
  const int it = std::max(m->ccd_iterations, m->epa_iterations);
  const CcdLayout CL = ccd_layout(d->nworld, it, m->nhfield, m->npolygonmax, m->nmeshdegmax, collide_ccap(m->npair, d->concap), d->nccdhand, m->npair);
  hipLaunchKernelGGL(k_ccd_reset, dim3(std::max((m->npair + 63) / 64, 1)), dim3(64), 0, s, *m, *d);  // counters, convex-pair mask (a kernel, not a memset node: replayed inside hipGraphs)
  if (m->broadphase == 0 && m->npair > 0) {  // NXN: the broadphase filters of every world as their own launch (a workgroup per world), results as bit masks
    const size_t lds_mask = sizeof(float) * (size_t)bmask_layout(m->ngeom, m->npair, m->broadphase_filter, m->ncullgeom, m->ncullgroup, m->ncullpair).total;
    if (lds_mask > 160 * 1024 || m->ngeom > 65535) return fail(MJH_E_UNSUPPORTED, "k_broad_mask: the geom tables do not fit in LDS");
    if (m->ncullpair <= 0 || !m->cull_pair || !m->cull_list) return fail(MJH_E_ARG, "k_broad_mask: Model.cull_pair / cull_list missing (io.py cull_tables; a binding must build them, INTEGRATION.md ABI v42)");
    HIPCHK(set_lds(k_broad_mask, lds_mask));
    hipLaunchKernelGGL(k_broad_mask, dim3((unsigned)std::min(d->nworld, 8192)), dim3(256), lds_mask, s, *m, *d);
  }
  if (m->broadphase != 0 || m->npair == 0) {  // sweep-and-prune: the broadphase of a world by its lane group (k_broad_mask publishes the NXN list itself)
    size_t lds;
    const int threads = pick_block(sizeof(float) * 9 * m->ngeom, sizeof(float) * broad_lds_words(m->ngeom, m->npair, d->concap, m->broadphase), G, &lds);  // (+ the staged model tables)
    if (!threads) return fail(MJH_E_UNSUPPORTED, "k_ccd_broad: pair list does not fit in LDS");
    HIPCHK(set_lds((k_ccd_broad<G>), lds));
    const int wpb = threads / G;
    hipLaunchKernelGGL((k_ccd_broad<G>), dim3((d->nworld + wpb - 1) / wpb), dim3(threads), lds, s, *m, *d);
  }
  // (grids sized for the device -- 256 CUs x 8 workgroups --, not for the lists' capacities: the kernels walk their lists with the grid's stride)
  {
    // lanes per pair by the length of the list (read on the device): one lane per pair needs >= 2 wavefronts per SIMD to hide its chains of
    // dependent table loads; shorter lists give a pair 8 or 32 lanes (MJH_GJK_LANES: developer knob, forces one instantiation)
    static const int force = mjh_knob("MJH_GJK_LANES") ? atoi(mjh_knob("MJH_GJK_LANES")) : 0;
    static const int t8_env = mjh_knob("MJH_GJK_T8") ? atoi(mjh_knob("MJH_GJK_T8")) : 16384, t1_env = mjh_knob("MJH_GJK_T1") ? atoi(mjh_knob("MJH_GJK_T1")) : 131072;  // developer knobs
    const int all = 0x7fffffff, t8 = force ? (force == 32 ? all : 0) : t8_env, t1 = force ? (force == 1 ? 0 : all) : std::max(t1_env, t8_env);
    const long long cap = (long long)d->nworld * CL.ccap;  // work items at most
    const int grid1 = (int)std::min<long long>((cap + 255) / 256, 2048);
    hipLaunchKernelGGL(k_ccd_gjk<32>, dim3((unsigned)std::min<long long>((cap + 7) / 8, 4096)), dim3(256), 0, s, *m, *d, 0, t8);
    hipLaunchKernelGGL(k_ccd_gjk<8>, dim3((unsigned)std::min<long long>((cap + 31) / 32, 4096)), dim3(256), 0, s, *m, *d, t8, t1);
    hipLaunchKernelGGL(k_ccd_gjk<1>, dim3(grid1), dim3(256), 0, s, *m, *d, t1, all);
  }
  // lane groups per workgroup: a group's LDS (polytope + the multi-contact polygon buffers: 11.7 KB on the ALOHA scene) decides how many are
  // resident on a CU -- eight groups in one 256-thread workgroup leave room for ONE workgroup there; smaller workgroups pack the LDS
  static const int epa_threads = mjh_knob("MJH_EPA_THREADS") ? atoi(mjh_knob("MJH_EPA_THREADS")) : 256;  // developer knob
  const size_t group_bytes = sizeof(float) * (size_t)ccd_coop_words(it, m->npolygonmax, m->nmeshdegmax);
  int gpb = std::max(epa_threads / G, 1);
  while (gpb > 1 && group_bytes * gpb > (size_t)kLdsPerCU) gpb >>= 1;  // (a mesh vertex of very high degree: long feature lists per group)
  if (group_bytes > (size_t)kLdsPerCU) return fail(MJH_E_UNSUPPORTED, "k_ccd_epa: a lane group's polytope and multi-contact lists do not fit in LDS (mesh vertex degree / polygon size too large)");
  const size_t lds_epa = group_bytes * gpb;
  HIPCHK(set_lds((k_ccd_epa<G>), lds_epa));
  hipLaunchKernelGGL((k_ccd_epa<G>), dim3(std::min((CL.handcap + gpb - 1) / gpb, 16384 / 8)), dim3(G * gpb), lds_epa, s, *m, *d);
  return MJH_OK;
}
template <int G>
static int launch_collision_g(const MjhModel* m, const MjhData* d, hipStream_t s) {
  if (m->heavy_colliders && d->ws_ccd) {
    const int rc = launch_ccd_pre<G>(m, d, s);
    if (rc != MJH_OK) return rc;
  }
  size_t lds;
  const int threads = pick_block(0, sizeof(float) * collide_lds_words(m->ngeom, m->npair, d->concap, m->heavy_colliders ? m->broadphase : 0, m->heavy_colliders && d->ws_ccd != nullptr), G, &lds);
  if (!threads) return fail(MJH_E_UNSUPPORTED, "k_collision: pair list does not fit in LDS");
  if (m->heavy_colliders) {
    const int wpb_h = threads / G;
    if (m->nhfield > 0) {
      HIPCHK(set_lds((k_collision<G, true, true>), lds));
      hipLaunchKernelGGL((k_collision<G, true, true>), dim3((d->nworld + wpb_h - 1) / wpb_h), dim3(threads), lds, s, *m, *d);
    } else {  // (without the height-field colliders: the registers of their per-lane GJK / EPA)
      HIPCHK(set_lds((k_collision<G, true, false>), lds));
      hipLaunchKernelGGL((k_collision<G, true, false>), dim3((d->nworld + wpb_h - 1) / wpb_h), dim3(threads), lds, s, *m, *d);
    }
    return MJH_OK;
  }
  HIPCHK(set_lds((k_collision<G, false>), lds));
  const int wpb = threads / G;
  hipLaunchKernelGGL((k_collision<G, false>), dim3((d->nworld + wpb - 1) / wpb), dim3(threads), lds, s, *m, *d);
  return MJH_OK;
}
static int launch_collision(const MjhModel* m, const MjhData* d, hipStream_t s) { return lanes64(m) ? launch_collision_g<64>(m, d, s) : launch_collision_g<32>(m, d, s); }
// compact public contact arrays, contact.efc_address and efc.id of contact rows from the per-world records
template <int G>
static int launch_publish_g(const MjhModel* m, const MjhData* d, hipStream_t s) {
  hipLaunchKernelGGL(k_publish_contacts<G>, dim3((d->nworld + 256 / G - 1) / (256 / G)), dim3(256), 0, s, *d, 1, m->nexplicit ? m->pair_solreffriction : nullptr);
  return MJH_OK;
}
static int launch_publish(const MjhModel* m, const MjhData* d, hipStream_t s) { return launch_publish_g<32>(m, d, s); }  // (chains over the sparse factor: more lanes per world only halve the worlds per wavefront)
template <int G>
static int launch_constraint_g(const MjhModel* m, const MjhData* d, hipStream_t s) {
  const ConLayout lay = con_layout(m->nv, d->njmax, d->concap, m->nbody, m->ngeom);
  size_t lds;
  const int threads = pick_block(0, sizeof(float) * lay.total, G, &lds);
  if (!threads) return fail(MJH_E_UNSUPPORTED, "k_make_constraint: does not fit in LDS");
  HIPCHK(set_lds(k_make_constraint<G>, lds));
  const int wpb = threads / G;
  hipLaunchKernelGGL(k_make_constraint<G>, dim3((d->nworld + wpb - 1) / wpb), dim3(threads), lds, s, *m, *d);
  return MJH_OK;
}
static int launch_constraint(const MjhModel* m, const MjhData* d, hipStream_t s) { return lanes64(m) ? launch_constraint_g<64>(m, d, s) : launch_constraint_g<32>(m, d, s); }
static int launch_rne_postconstraint(const MjhModel* m, const MjhData* d, hipStream_t s) {
  const int wpb = lds_wpb32((size_t)18 * m->nbody);
  if (!wpb) return fail(MJH_E_UNSUPPORTED, "k_rne_postconstraint: nbody does not fit in LDS");
  const size_t lds = sizeof(float) * 18 * m->nbody * wpb;
  HIPCHK(set_lds(k_rne_postconstraint<32>, lds));
  hipLaunchKernelGGL(k_rne_postconstraint<32>, dim3((d->nworld + wpb - 1) / wpb), dim3(32 * wpb), lds, s, *m, *d);
  return MJH_OK;
}
static int launch_subtree_vel(const MjhModel* m, const MjhData* d, hipStream_t s) {
  const int wpb = subtree_vel_wpb(m->nbody);
  if (!wpb) return fail(MJH_E_UNSUPPORTED, "k_subtree_vel: nbody does not fit in LDS");
  const size_t lds = sizeof(float) * 9 * m->nbody * wpb;
  HIPCHK(set_lds(k_subtree_vel<32>, lds));
  hipLaunchKernelGGL(k_subtree_vel<32>, dim3((d->nworld + wpb - 1) / wpb), dim3(32 * wpb), lds, s, *m, *d);
  return MJH_OK;
}
static int launch_sensor(const MjhModel* m, const MjhData* d, int stage, hipStream_t s) {  // stage 1: acceleration-stage sensors (after the solver)
  if (stage == 0 && ((m->enableflags & ENBL_ENERGY) || m->nsensor_energy > 0)) {  // Data.energy rides with the position / velocity stage sensors (forward.py:1326-1338)
    if (!d->energy) return fail(MJH_E_ARG, "Data.energy missing (allocate Data with make_data/put_data)");
    hipLaunchKernelGGL(k_energy<32>, dim3((d->nworld + 7) / 8), dim3(256), 0, s, *m, *d);
  }
  if (m->nsensor == 0 || (m->disableflags & DSBL_SENSOR) || (stage == 1 && m->nsensor_acc == 0)) return MJH_OK;
  if (stage == 1 && m->nsensor_frc > 0) {
    if (!d->cfrc_ext) return fail(MJH_E_ARG, "Data.cfrc_ext missing");
    TRY(launch_rne_postconstraint(m, d, s));
  }
  if (stage == 0 && m->nsensor_subtree > 0) {
    if (!d->subtree_linvel || !d->subtree_angmom) return fail(MJH_E_ARG, "Data.subtree_linvel / subtree_angmom missing");
    TRY(launch_subtree_vel(m, d, s));
  }
  if (!d->sensordata) return fail(MJH_E_ARG, "Data.sensordata missing (allocate Data with make_data/put_data)");
  hipLaunchKernelGGL(k_sensor, dim3((d->nworld * m->nsensor + 255) / 256), dim3(256), 0, s, *m, *d, stage);
  return MJH_OK;
}
static int launch_solve(const MjhModel* m, const MjhData* d, hipStream_t s);  // = the solver workgroups of k_solve_plus
// the linear system of the fully implicit integrator (csrc/implicit.hpp): Data.ws_iacc = (M - h dF/dv)^-1 M qacc
static int launch_implicit(const MjhModel* m, const MjhData* d, hipStream_t s) {
  const ImpLayout lay = imp_layout(m->nv, m->nbody, 32);
  int wpb = 2;
  if (sizeof(float) * lay.total * 2 > (size_t)kLdsPerCU) wpb = 1;
  const size_t lds = sizeof(float) * lay.total * wpb;
  if (lds > (size_t)kLdsPerCU) return fail(MJH_E_UNSUPPORTED, "k_implicit_solve: nv / nbody do not fit in LDS");
  HIPCHK(set_lds(k_implicit_solve<32>, lds));
  hipLaunchKernelGGL(k_implicit_solve<32>, dim3((d->nworld + wpb - 1) / wpb), dim3(32 * wpb), lds, s, *m, *d);
  return MJH_OK;
}
template <int G>
static int launch_integrate_g(const MjhModel* m, const MjhData* d, int mode, hipStream_t s) {
  if (mode == 2) TRY(launch_implicit(m, d, s));
  const IntLayout lay = int_layout(m->nv, m->nC);
  size_t lds;
  const int threads = pick_block(sizeof(int) * mstruct_ints(m->nv, m->nC), sizeof(float) * lay.total, G, &lds);
  if (!threads) return fail(MJH_E_UNSUPPORTED, "k_integrate: does not fit in LDS");
  HIPCHK(set_lds(k_integrate<G>, lds));
  const int wpb = threads / G;
  hipLaunchKernelGGL(k_integrate<G>, dim3((d->nworld + wpb - 1) / wpb), dim3(threads), lds, s, *m, *d, mode);
  return MJH_OK;
}
static int launch_integrate(const MjhModel* m, const MjhData* d, int mode, hipStream_t s) { return launch_integrate_g<32>(m, d, mode, s); }  // (chains over the sparse factor: more lanes per world only halve the worlds per wavefront)

// ---- composite launches of the fused step --------------------------------------------------------------------
// Cross-stream fork/join costs 5-15 us per hop on the critical path (event record -> barrier packet -> dispatch),
// while consecutive kernels of one stream start back to back.  The fused step therefore uses ONE stream and gives
// the workgroups of a launch different roles:
//   k_mid           : {collision -> make_constraint} workgroups, then the (shorter) fwd_vel workgroups (both only depend
//                     on k_fwd_pos; both are latency-bound, so they share the CUs)
//   k_solve_plus    : solver workgroups (longest expected solve first), then factor_smooth and publish workgroups: dispatched last,
//                     they fill the CUs that the solver's stragglers leave idle
//   k_fwd_pos_plus  : k_fwd_pos + one workgroup computing the solver schedule from the previous step's solver_niter
//   k_integrate_plus: integrator workgroups, then publish_contacts workgroups
template <int G, bool HEAVY, bool HFT = true>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((!HEAVY && G >= 32) ? MJH_MID_WAVES : 1, 8))) k_mid(MjhModel m, MjhData d, int ncc, int nvb, int nw_cc, int nw_v, int stride_cc, int sched) {  // (the light instantiation: four wavefronts per SIMD = 128 registers, its occupancy since round 3)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  (void)nvb;
  // sched: workgroup 0 sorts the solver schedule here instead of in the k_fwd_pos launch (models whose fwd_pos
  // workgroups are too small to do it quickly)
  // (LAST in the dispatch order, round 5: every launch of the step fills the device's workgroup slots exactly -- LDS-bound, four per CU --, so
  // one more workgroup in front pushes a real one into a second round: + 7 us here, + 8 us in k_fwd_pos_plus, measured.  Behind the
  // velocity-stage workgroups it takes the first slot one of them frees and ends inside the spread of their finish times.)
  if (sched && blockIdx.x == gridDim.x - 1) {
    schedule_body(d, reinterpret_cast<int*>(smem), blockDim.x, sched_cls(m, d));
    return;
  }
  const int bi = (int)blockIdx.x;
  const int cc_before = bi < ncc ? (int)bi : ncc;               // longest jobs first: all CC workgroups, then fwd_vel
  const bool is_cc = bi < ncc;
  if (is_cc) {
    const Blk b{cc_before * nw_cc, nw_cc, nw_cc * G};
    collision_body<G, HEAVY, 0, HFT>(m, d, smem, b, stride_cc);
    __threadfence_block();  // the world's contact records (global) are read back by the same lanes
    make_constraint_body<G>(m, d, smem, b, stride_cc);
  } else {
    const Blk b{((int)bi - cc_before) * nw_v, nw_v, nw_v * G};
    fwd_vel_body<G>(m, d, VEL_COMVEL, VEL_ACCEL, smem, b);
  }
}
template <int G>
__global__ void __launch_bounds__(256) k_integrate_plus(MjhModel m, MjhData d, int mode, int nint, int npub) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wpb = blockDim.x / G, bi = blockIdx.x;
  if (bi < nint) integrate_body<G>(m, d, mode, smem, Blk{bi * wpb, wpb, (int)blockDim.x});
  else if (bi < nint + npub) publish_body<G>(d, 1, reinterpret_cast<int*>(smem), Blk{(bi - nint) * wpb, wpb, (int)blockDim.x}, m.nexplicit ? m.pair_solreffriction : nullptr);
  else factor_smooth_body<G>(m, d, 1, smem, Blk{(bi - nint - npub) * wpb, wpb, (int)blockDim.x});  // Newton only, see k_solve_plus
}
// k_fwd_pos + one workgroup that sorts the worlds by the PREVIOUS step's solver_niter (the solver schedule of this step)
template <int G>
__global__ void __launch_bounds__(256) k_fwd_pos_plus(MjhModel m, MjhData d, int first, int last, int npos, NoiseArgs noise) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // the schedule workgroup goes FIRST: workgroups are dispatched in index order, so as the last one it would start
  // when the launch is nearly over and add its whole duration (~8 us) to it
  if (blockIdx.x == 0) {
    if (noise.sched) schedule_body(d, reinterpret_cast<int*>(smem), blockDim.x, sched_cls(m, d));
  } else if ((int)blockIdx.x <= npos) {
    const int wpb = blockDim.x / G;
    fwd_pos_body<G>(m, d, first, last, smem, Blk{((int)blockIdx.x - 1) * wpb, wpb, (int)blockDim.x});
  } else {
    // the benchmark loop's control noise rides here (mjh_timed_steps): nothing in this launch reads ctrl, the next launch
    // (fwd_vel's actuation) does; dispatched last, these few trivial workgroups run in the launch's tail
    const int idx = ((int)blockIdx.x - npos - 1) * blockDim.x + threadIdx.x;
    if (idx < noise.n) ctrl_noise_elem(m, d, noise.center, noise.step, noise.noise_std, noise.noise_rate, idx);
  }
}

template <int G>
static int launch_mid_g(const MjhModel* m, const MjhData* d, bool sched, hipStream_t s) {
  if constexpr (G != 16) {
    if (m->heavy_colliders && d->ws_ccd) {
      const int rc = launch_ccd_pre<G>(m, d, s);
      if (rc != MJH_OK) return rc;
    }
  }
  const ConLayout cl = con_layout(m->nv, d->njmax, d->concap, m->nbody, m->ngeom);
  const int stride_cc = std::max(cl.total, collide_lds_words(m->ngeom, m->npair, d->concap, m->heavy_colliders ? m->broadphase : 0, m->heavy_colliders && d->ws_ccd != nullptr) | 1);
  const VelLayout vl = vel_layout(m->nq, m->nv, m->nbody, m->nC, m->nu);
  const size_t ms_bytes = sizeof(int) * mstruct_ints(m->nv, m->nC);
  // 8 collision+constraint worlds per workgroup; as many fwd_vel worlds as fit in the same LDS footprint
  int nw_cc = 256 / G;
  while (nw_cc > 1 && sizeof(float) * stride_cc * nw_cc > (size_t)kLdsPerCU / 2) nw_cc >>= 1;
  size_t lds = sizeof(float) * stride_cc * nw_cc;
  int nw_v = (int)((lds > ms_bytes ? lds - ms_bytes : 0) / (sizeof(float) * vl.total));
  if (nw_v < 1) nw_v = 1;
  if (nw_v > 256 / G) nw_v = 256 / G;
  // an odd world count leaves half a wavefront slot empty: round up when the same number of workgroups still fits a CU
  if ((nw_v & 1) && nw_v < 256 / G) {
    const size_t up = ms_bytes + sizeof(float) * vl.total * (nw_v + 1);
    if ((size_t)kLdsPerCU / std::max(up, lds) == (size_t)kLdsPerCU / lds) ++nw_v;
    else --nw_v;
  }
  if (nw_v < 1) nw_v = 1;
  if (const char* e = mjh_knob("MJH_MID_W")) {  // tuning knob (developer only): worlds per workgroup of both roles
    nw_cc = nw_v = atoi(e);
    lds = sizeof(float) * stride_cc * nw_cc;
  }
  lds = std::max(lds, ms_bytes + sizeof(float) * vl.total * nw_v);
  if (lds > (size_t)kLdsPerCU) return fail(MJH_E_UNSUPPORTED, "k_mid: model does not fit in LDS");
  if constexpr (G == 16) {  // (light colliders only: lanes16)
    HIPCHK(set_lds((k_mid<G, false>), lds));
    const int ncc16 = (d->nworld + nw_cc - 1) / nw_cc, nvb16 = (d->nworld + nw_v - 1) / nw_v;
    hipLaunchKernelGGL((k_mid<G, false>), dim3(ncc16 + nvb16 + (sched ? 1 : 0)), dim3(G * std::max(nw_cc, nw_v)), lds, s, *m, *d, ncc16, nvb16, nw_cc, nw_v, stride_cc, sched ? 1 : 0);
    return MJH_OK;
  } else {
  const bool hf = m->nhfield > 0;  // (heavy colliders without height fields: the instantiation without their per-lane GJK / EPA)
  if (m->heavy_colliders && hf) HIPCHK(set_lds((k_mid<G, true, true>), lds));
  else if (m->heavy_colliders) HIPCHK(set_lds((k_mid<G, true, false>), lds));
  else HIPCHK(set_lds((k_mid<G, false>), lds));
  const int ncc = (d->nworld + nw_cc - 1) / nw_cc, nvb = (d->nworld + nw_v - 1) / nw_v;
  const dim3 grid(ncc + nvb + (sched ? 1 : 0)), block(G * std::max(nw_cc, nw_v));
  if (m->heavy_colliders) debug_occupancy("k_mid<heavy>", k_mid<G, true>, (int)grid.x, (int)block.x, lds);
  else debug_occupancy("k_mid", k_mid<G, false>, (int)grid.x, (int)block.x, lds);
  if (m->heavy_colliders && hf) hipLaunchKernelGGL((k_mid<G, true, true>), grid, block, lds, s, *m, *d, ncc, nvb, nw_cc, nw_v, stride_cc, sched ? 1 : 0);
  else if (m->heavy_colliders) hipLaunchKernelGGL((k_mid<G, true, false>), grid, block, lds, s, *m, *d, ncc, nvb, nw_cc, nw_v, stride_cc, sched ? 1 : 0);
  else hipLaunchKernelGGL((k_mid<G, false>), grid, block, lds, s, *m, *d, ncc, nvb, nw_cc, nw_v, stride_cc, sched ? 1 : 0);
  return MJH_OK;
  }
}
static int launch_mid(const MjhModel* m, const MjhData* d, bool sched, hipStream_t s) { return lanes16(m) ? launch_mid_g<16>(m, d, sched, s) : lanes64(m) ? launch_mid_g<64>(m, d, sched, s) : launch_mid_g<32>(m, d, sched, s); }
// set by the fused STEP path when the solver launch also integrates (see euler_fusable)
static thread_local int g_fuse_euler = 0;  // 1: explicit Euler, 2: implicitfast in the solver's epilogue
// set by the fused path when the Newton riders run on the side stream (see side_stream)
static thread_local bool g_riders_on_side = false;
static int solve_supported(const MjhModel* m, const MjhData* d) {
  if (m->cone != CONE_PYRAMIDAL && m->cone != CONE_ELLIPTIC) return fail(MJH_E_UNSUPPORTED, "unknown cone type");
  if (m->solver != SOL_NEWTON && m->solver != SOL_CG && m->solver != SOL_PGS) return fail(MJH_E_UNSUPPORTED, "unknown solver");
  if (m->nv <= 64 && d->njmax > 192 && m->solver == SOL_PGS && m->cone != CONE_ELLIPTIC)
    return fail(MJH_E_UNSUPPORTED, "njmax > 192 with the register / LDS resident PGS kernels is not supported (nv <= 64, pyramidal)");
  return MJH_OK;
}
// Auxiliary streams per host thread and device for the solver variants of the per-island dispatch (nv > 64): the rare island classes (one
// stream each: round 6) and the generic solver touch islands / worlds the common-class launch skips, so they run beside it instead of
// after it.  Released by mjh_release_thread_resources().
#define MJH_NAUX 4
#define MJH_NAUX_EAGER 4  // (see launch_solve_any: eager launches pay for the forks while few islands are awake, and win afterwards)
struct Aux {
  hipStream_t stream[MJH_NAUX];
  hipEvent_t fork, join[MJH_NAUX];
};
static thread_local Aux* g_aux_per_dev[16] = {nullptr};
static thread_local bool g_serial_solver = false;  // set while per-kernel instrumentation is on (one stream, one event pair per launch)
static Aux* aux_streams() {
  static const bool disabled = mjh_knob("MJH_NO_AUX") != nullptr;  // developer knob
  if (disabled || g_serial_solver) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!g_aux_per_dev[dev]) {
    Aux* a = new Aux();
    bool ok = hipEventCreateWithFlags(&a->fork, hipEventDisableTiming) == hipSuccess;
    for (int k = 0; k < MJH_NAUX && ok; ++k)
      ok = hipStreamCreateWithFlags(&a->stream[k], hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&a->join[k], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
      delete a;  // (a partially created set leaks a handle or two: only on an already failing device)
      return nullptr;
    }
    g_aux_per_dev[dev] = a;
  }
  return g_aux_per_dev[dev];
}
// The Newton riders as trailing workgroups of the MFMA solver launch (solver_newton.hpp) instead of a side-stream launch, for SMALL models
// (round 3).  Where the solve is short the side stream's fork / join hops and the riders' late start set the step: Panda (nv 9, njmax 5: solver
// 22 us) 135.0 -> 117.2 us per step (60.7 -> 69.9 M env-steps/s, two interleaved pairs on one box).  Where the solve is long the riders'
// workgroups in its tail cost more than the join they save: humanoid (nv 27) 0.2824 / 0.2844 -> 0.2894 / 0.2897 ms, so the side stream stays
// above 16 dofs.  Applies exactly when launch_solve_32_newton picks the MFMA kernel.  MJH_NEWTON_RIDERS=0 / 1 forces it off / on (developer knob).
static thread_local bool g_newton_inline = false;
static bool newton_inline_ok(const MjhModel* m, const MjhData* d) {
  static const int force = mjh_knob("MJH_NEWTON_RIDERS") ? atoi(mjh_knob("MJH_NEWTON_RIDERS")) : -1;
  static const bool old_path = mjh_knob("MJH_OLD_NEWTON") != nullptr;
  const bool ell = m->cone == CONE_ELLIPTIC && d->nmaxpyramid > 1;
  const bool want = force >= 0 ? force != 0 : m->nv <= 16;
  return want && !old_path && m->solver == SOL_NEWTON && !ell && m->nv <= 32 && d->njmax <= 64 && (double)d->nworld * std::max(m->nv, d->njmax) * 4.0 < 4.0e9;
}
// Which CG kernel serves the class nv <= 32, pyramidal cones (the ONE place that decides; mjh_solver_kernel reports it).
//   CG32_CGW  one world per wavefront (solver_cgw.hpp): SMALL batches.  A solve is a dependent chain of ~17 k instructions per wavefront of which
//             the line search's per-wavefront bookkeeping is the bulk, so pairing two worlds in a wavefront halves the instruction count per world
//             and wins whenever the SIMDs are issue bound (measured, humanoid: 8192 worlds 194 us one-per-wavefront vs 193 paired, 141 M vs 81 M
//             VALU instructions); with at most ~3 wavefronts per SIMD the chain's latency decides instead and the shorter chain of the unpaired
//             kernel wins (1024 worlds: 88 vs 112 us, 2048: 103 vs 120, 3072: 122 vs 124)
//   CG32_CGP  the pooled contact-basis kernel (solver_cgp.hpp): larger batches of models whose contacts are condim 1 / 3 and that have no
//             friction-loss rows (MjhModel.cg_basis), njmax <= 64
//   CG32_PAIR k_solve<cg> (solver.hpp): everything else
// MJH_CG_KERNEL = cgp / pair / cgw (developer knob; tests set it through mjh_dev_knob) forces one of the three where it applies.
enum { CG32_PAIR = 0, CG32_CGW = 1, CG32_CGP = 2 };
static int cg32_choice(const MjhModel* m, const MjhData* d, int fe) {
  const bool newton = m->solver == SOL_NEWTON, ell = m->cone == CONE_ELLIPTIC && d->nmaxpyramid > 1;
  if (newton || ell || m->solver == SOL_PGS || m->nv > 32) return CG32_PAIR;
  static const int wide_min_nv = mjh_knob("MJH_CGW_MIN_NV") ? atoi(mjh_knob("MJH_CGW_MIN_NV")) : 13;
  static const int wide_max_nworld = mjh_knob("MJH_CGW_MAX_NWORLD") ? atoi(mjh_knob("MJH_CGW_MAX_NWORLD")) : 3072;
  const bool wide = m->nv >= wide_min_nv && d->nworld <= wide_max_nworld && fe != 2;  // (its half rows of M do not serve the fused implicitfast update)
  const bool cgp_can = m->cg_basis && d->njmax <= 64 && m->nv >= 1;
  const char* force = mjh_knob("MJH_CG_KERNEL");
  if (force && !strcmp(force, "cgw")) return fe != 2 ? CG32_CGW : CG32_PAIR;
  if (force && !strcmp(force, "pair")) return CG32_PAIR;
  if (force && !strcmp(force, "cgp")) return cgp_can ? CG32_CGP : CG32_PAIR;
  return wide ? CG32_CGW : (cgp_can ? CG32_CGP : CG32_PAIR);
}
#ifndef MJH_SOLVE64_SPLIT_DEFAULT
#define MJH_SOLVE64_SPLIT_DEFAULT 0
#endif
static int launch_solve_any(const MjhModel* m, const MjhData* d, bool with_factor, hipStream_t s) {
  if (int rc = solve_supported(m, d)) return rc;
  if (m->solver == SOL_NEWTON) with_factor = with_factor && g_newton_inline;
  if (m->nv > 64) {  // no riders: they go with the integrator launch
    if (m->solver == SOL_PGS) return launch_pgs(m, d, s);  // (the generic PGS kernel: csrc/pgs_big.hpp)
    if (m->tree_solve) {
      // constraint islands (trees joined by coupling rows): worlds whose islands all have at most 64 dofs are solved per island by the
      // register-resident kernels, the others by the generic solver below
      hipLaunchKernelGGL(k_isl_clear, dim3(1), dim3(64), 0, s, *d);
      hipLaunchKernelGGL(k_tree_rows, dim3(d->nworld), dim3(64), sizeof(int) * (size_t)(2 * std::max(d->njmax, 1) + 2 * m->ntree), s, *m, *d);
      const bool ell_t = m->cone == CONE_ELLIPTIC && d->nmaxpyramid > 1;
      Aux* aux = aux_streams();
      // One stream per rare island class (round 6).  clutter_synth, 2048 worlds, steps 100-300 (tools/clutter_ab.py, bit-identical states, two
      // interleaved rounds): hipGraph replay -- the reference's own way to run a step, cli.py:262-290 -- none 1.27, one shared side stream 1.29,
      // one each 1.43 M env-steps/s; eager launches 1.28 / 1.28 / 1.40.  (Eager, steps 0-100 -- five trees awake, every class launch nearly
      // empty -- the forks are host API calls on the critical path: 1.74 / 1.48 / 1.51; the graph replay does not pay them: 1.75 / 1.73 / 1.78.)
      int naux = 0;
      if (aux) {
        hipStreamCaptureStatus cst = hipStreamCaptureStatusNone;
        naux = (hipStreamIsCapturing(s, &cst) == hipSuccess && cst == hipStreamCaptureStatusActive) ? MJH_NAUX : MJH_NAUX_EAGER;
        static const int cap = mjh_knob("MJH_NAUX") ? atoi(mjh_knob("MJH_NAUX")) : -1;  // developer knob (A/B): 2 = one side stream for the rare classes, 4 = one each
        if (cap >= 2 && cap <= MJH_NAUX) naux = cap;
      }
      // stream 0: islands of 8..16 dofs; 1: the generic solver (worlds with an island beyond 64 dofs); 2: 16..32 dofs; 3: many rows / 33..64 dofs.
      // MJH_BIG_SHARES=1 (developer knob, A/B): the generic solver behind the 16..32-dof class on stream 2 -- four branches instead of five (the
      // runtime has four hardware queues by default: R6.7).  Measured, two interleaved rounds: clutter_synth 1.55 / 1.54, three_humanoids
      // 5.18 / 5.17 M env-steps/s -- nothing; off.
      static const bool big_shares = mjh_knob("MJH_BIG_SHARES") && atoi(mjh_knob("MJH_BIG_SHARES")) != 0;
      const bool share = big_shares && naux > 2;
      bool used[MJH_NAUX] = {false, false, false, false};
      for (int k = 0; k < naux; ++k) used[k] = !(share && k == 1);
      hipStream_t s1 = aux ? aux->stream[0] : s, s3 = naux > 2 ? aux->stream[2] : s1, s4 = naux > 3 ? aux->stream[3] : s1;
      hipStream_t s2 = !aux ? s : (share ? s3 : aux->stream[1]);
      if (aux) {
        HIPCHK(hipEventRecord(aux->fork, s));
        for (int k = 0; k < naux; ++k)
          if (used[k]) HIPCHK(hipStreamWaitEvent(aux->stream[k], aux->fork, 0));
      }
      int rc = (m->solver == SOL_NEWTON ? (ell_t ? launch_solve_tree_newton_ell : launch_solve_tree_newton) : (ell_t ? launch_solve_tree_cg_ell : launch_solve_tree_cg))(m, d, s, s1, s3, s4);
      if (!rc) rc = launch_solve_big(m, d, s2);
      if (aux) {  // (every fork rejoins the caller's stream, also on the error path: the streams may be under capture)
        for (int k = 0; k < naux; ++k) {
          if (!used[k]) continue;
          HIPCHK(hipEventRecord(aux->join[k], aux->stream[k]));
          HIPCHK(hipStreamWaitEvent(s, aux->join[k], 0));
        }
      }
      return rc;
    }
    return launch_solve_big(m, d, s);
  }
  if (m->solver == SOL_PGS) return launch_pgs(m, d, s);
  const bool newton = m->solver == SOL_NEWTON;
  const int fe = g_fuse_euler;
  const int all = 0x7fffffff;
  // the k_solve_plus instantiations live in their own translation units (host.hpp): pick by lanes per world and solver
  const bool ell = m->cone == CONE_ELLIPTIC && d->nmaxpyramid > 1;  // (condim 1 everywhere: the cone type is moot)
  auto s32 = ell ? (newton ? launch_solve_32_newton_ell : launch_solve_32_cg_ell) : (newton ? launch_solve_32_newton : launch_solve_32_cg);
  auto s64 = ell ? (newton ? launch_solve_64_newton_ell : launch_solve_64_cg_ell) : (newton ? launch_solve_64_newton : launch_solve_64_cg);
  // njmax > 64: two launches over the same world list (see solve_body): a small-row instantiation for the worlds with
  // at most 64 rows, the big one (riders attached) for the rest
  // njmax > 192: the register-resident kernels end at 192 rows (6 x 32 / 3 x 64 lanes); the (rare) worlds beyond go to the generic
  // solver, which keeps J in HBM and is generic in njmax
  const int top = d->njmax > 192 ? 192 : all;
  if (m->nv <= 32) {
    const int cgk = cg32_choice(m, d, fe);  // (above)
    const bool wide_f = cgk == CG32_CGW, cgp = cgk == CG32_CGP;
    // rows per lane (32 lanes per world): 1 covers 32 rows, 2 covers 64 rows (humanoid, panda), 6 covers 192.  Newton with elliptic cones has
    // a one-row instantiation for the worlds of at most 32 rows (the ALOHA scene: nefc 24 on average, 27 at the 95th percentile): 155 instead
    // of 191 VGPRs and 6.7 instead of 12.2 KB of LDS per world -- 2.9 instead of 1.6 wavefronts per SIMD -- and half the row work per lane:
    // 333 -> 254 us per step there (MJH_SOLVE_R1=0: developer knob, off).  The launches follow one another on the caller's stream: a batch
    // with worlds in both classes pays the latency of one more solve (on a side stream beside the 64-row launch the ALOHA scene LOST 17 %:
    // 4.95 vs 5.97 M env-steps/s, the fork / join hops inside the step's graph).  (The same for CG with pyramidal cones was built and
    // measured on the humanoid -- bit-identical results, but no gain: behind one another 0.471 vs 0.369 ms per step at nefc 46 where a
    // quarter of the worlds are in the small class; side by side 0.370 vs 0.372 ms there and 0.349 vs 0.329 ms at nefc 32.  Not kept.)
    static const bool r1_on = !mjh_knob("MJH_SOLVE_R1") || atoi(mjh_knob("MJH_SOLVE_R1")) != 0;
    const bool r1 = r1_on && newton && ell && d->njmax > 32;
    int lo2 = -1;
    if (r1) {
      if (int rc = launch_solve_32_newton_ell_r1(m, d, false, fe, s, -1, 32)) return rc;
      lo2 = 32;
    }
    // CG, pyramidal, contacts of condim 1 / 3, njmax <= 64 (the headline class): contact-basis rows in one row pool per workgroup, three
    // wavefronts per SIMD (solver_cgp.hpp); the worlds it flags (a contact cut by njmax, a pool overflow) are solved by k_solve<cg> in the
    // launch behind it, which is empty in the common case.
    if (cgp) {
      if (int rc = launch_solve_cgp(m, d, with_factor, fe, s)) return rc;
      return launch_solve_32_cg_deferred(m, d, fe, s);
    }
    auto rest = [&, wide = wide_f]() -> int {
      if (d->njmax <= 64) return wide ? launch_solve_cgw(m, d, with_factor, fe, s, -1, all) : s32(m, d, 2, with_factor, fe, s, lo2, all);
      if (int rc = wide ? launch_solve_cgw(m, d, false, fe, s, -1, 64) : s32(m, d, 2, false, fe, s, lo2, 64)) return rc;
      if (int rc = s32(m, d, 6, with_factor, fe, s, 64, top)) return rc;
      return d->njmax > 192 ? launch_solve_big(m, d, s, 192) : MJH_OK;
    };
    return rest();
  }
  // 64 lanes per world: 1 / 2 / 3 rows per lane cover 64 / 128 / 192 rows.  The second launch of a pair runs after the
  // first on the same stream, so the split point is chosen to leave it (almost) empty: its real worlds would otherwise
  // be a serial tail on an idle GPU (G1: 6 % of the worlds exceed 64 rows, practically none exceed 128)
  if (d->njmax <= 64) return s64(m, d, 1, with_factor, fe, s, -1, all);
  // (developer knob MJH_SOLVE64_R1=1: the worlds of at most 64 rows by the one-row instantiation -- half the J tile -- in a launch of their own.
  // Measured on the G1 replay, 4096 worlds, nefc 68 on average: bit-identical states, 7.84 vs 9.14 M env-steps/s -- a second launch with real
  // worlds costs the latency of one more solve.  Off.)
  static const bool r1_64 = mjh_knob("MJH_SOLVE64_R1") && atoi(mjh_knob("MJH_SOLVE64_R1")) != 0;
  int lo64 = -1;
  if (r1_64) {
    if (int rc = s64(m, d, 1, false, fe, s, -1, 64)) return rc;
    lo64 = 64;
  }
  // Round 6: the same split with the two launches BESIDE one another (an auxiliary stream, fork / join through events).  The kernel is bound by
  // LDS per world (one-row instantiation 12.6 KB, two-row 23.7 KB: 8 against 6 worlds per CU on the G1), so the few-row worlds at their own
  // size shorten the batch by a partial round -- as long as they do not wait for the many-row launch.  MJH_SOLVE64_SPLIT=0 / 1 (developer knob).
  static const int split_knob = mjh_knob("MJH_SOLVE64_SPLIT") ? atoi(mjh_knob("MJH_SOLVE64_SPLIT")) : MJH_SOLVE64_SPLIT_DEFAULT;
  Aux* aux64 = (split_knob && !r1_64) ? aux_streams() : nullptr;
  if (aux64) {
    HIPCHK(hipEventRecord(aux64->fork, s));
    HIPCHK(hipStreamWaitEvent(aux64->stream[0], aux64->fork, 0));
    int rc = s64(m, d, 1, false, fe, aux64->stream[0], -1, 64);
    if (!rc) {
      if (d->njmax <= 128) rc = s64(m, d, 2, with_factor, fe, s, 64, all);
      else {
        rc = s64(m, d, 2, with_factor, fe, s, 64, 128);
        if (!rc) rc = s64(m, d, 3, false, fe, s, 128, top);
        if (!rc && d->njmax > 192) rc = launch_solve_big(m, d, s, 192);
      }
    }
    HIPCHK(hipEventRecord(aux64->join[0], aux64->stream[0]));  // (every fork rejoins the caller's stream, also on the error path)
    HIPCHK(hipStreamWaitEvent(s, aux64->join[0], 0));
    return rc;
  }
  if (d->njmax <= 128) return s64(m, d, 2, with_factor, fe, s, lo64, all);
  // Round 6: a launch's J tile is sized for the rows it can meet (solve_layout(min(njmax, hi))).  Lowering the two-row launch's bound to 112 rows
  // fits one more G1 world per CU (19.3 instead of 21.7 KB: 8 instead of 7) -- measured, two interleaved rounds, bit-identical states: 9.54 M
  // env-steps/s at 128, 9.47 at 112, 9.44 at 96: the launch is not bound by whole LDS rounds.  The bound stays 128; MJH_SOLVE64_HI2 (developer knob).
  static const int hi2_knob = mjh_knob("MJH_SOLVE64_HI2") ? atoi(mjh_knob("MJH_SOLVE64_HI2")) : 0;
  const int hi2 = (hi2_knob >= 80 && hi2_knob <= 128) ? (hi2_knob & ~15) : 128;
  if (int rc = s64(m, d, 2, with_factor, fe, s, lo64, hi2)) return rc;
  if (int rc = s64(m, d, 3, false, fe, s, hi2, top)) return rc;
  return d->njmax > 192 ? launch_solve_big(m, d, s, 192) : MJH_OK;
}
static int launch_solve(const MjhModel* m, const MjhData* d, hipStream_t s) { return launch_solve_any(m, d, false, s); }
static int launch_solve_plus(const MjhModel* m, const MjhData* d, hipStream_t s) { return launch_solve_any(m, d, true, s); }
// name of the solver mapping launch_solve_any picks for (m, d) in a fused step -- the decision functions above, nothing launched
static const char* solver_kernel_name(const MjhModel* m, const MjhData* d) {
  if (solve_supported(m, d)) return "unsupported";
  if (m->solver == SOL_PGS) return m->nv > 64 || (m->cone == CONE_ELLIPTIC && d->nmaxpyramid > 1) ? "pgs_big" : "pgs";
  if (m->nv > 64) return m->tree_solve ? "tree+big" : "big";
  const bool newton = m->solver == SOL_NEWTON, ell = m->cone == CONE_ELLIPTIC && d->nmaxpyramid > 1;
  if (m->nv > 32) return newton ? (ell ? "newton64_ell" : "newton64") : (ell ? "cg64_ell" : "cg64");
  if (newton) return ell ? "newton32_ell" : (d->njmax <= 64 ? "newton_mfma" : "newton32");
  if (ell) return "cg32_ell";
  const int fe = (m->integrator == INT_EULER || m->integrator == INT_IMPLICITFAST) ? 1 : 0;  // (only fe == 2 changes the choice, and only for cgw)
  switch (cg32_choice(m, d, fe)) {
    case CG32_CGW: return "cgw";
    case CG32_CGP: return "cgp";
    default: return "pair";
  }
}
// integrator (optional) + publication of the contact arrays + solver schedule
template <int G>
static int launch_integrate_plus_g(const MjhModel* m, const MjhData* d, int mode, bool integrate, hipStream_t s) {
  const IntLayout lay = int_layout(m->nv, m->nC);
  const FacLayout fl = fac_layout(m->nv, m->nC);
  const size_t ms_bytes = sizeof(int) * mstruct_ints(m->nv, m->nC);
  const bool with_factor = (m->solver != SOL_CG || m->nv > 64) && !g_riders_on_side;
  size_t lds = std::max(ms_bytes + sizeof(float) * std::max(lay.total, fl.total) * (256 / G), (size_t)2048);
  if (lds > (size_t)kLdsPerCU) return fail(MJH_E_UNSUPPORTED, "k_integrate: does not fit in LDS");
  HIPCHK(set_lds(k_integrate_plus<G>, lds));
  const int nb = (d->nworld + 256 / G - 1) / (256 / G);
  // Newton: publication and factor workgroups ride here; CG: they already rode with the solver launch
  const int nint = integrate ? nb : 0, npub = with_factor ? nb : 0, nfac = with_factor ? nb : 0;
  if (nint + npub + nfac == 0) return MJH_OK;
  if (integrate && mode == 2) TRY(launch_implicit(m, d, s));
  debug_occupancy("k_integrate_plus", k_integrate_plus<G>, nint + npub + nfac, 256, lds);
  hipLaunchKernelGGL(k_integrate_plus<G>, dim3(nint + npub + nfac), dim3(256), lds, s, *m, *d, mode, nint, npub);
  return MJH_OK;
}
static int launch_integrate_plus(const MjhModel* m, const MjhData* d, int mode, bool integrate, hipStream_t s) { return launch_integrate_plus_g<32>(m, d, mode, integrate, s); }  // (chains over the sparse factor: more lanes per world only halve the worlds per wavefront)
// *sched_done: whether the launch carried the schedule workgroup (it needs >= 128 threads to be quick; otherwise it
// rides with k_mid, whose workgroups always have 256)
// control noise queued by mjh_timed_steps for the next fused step: it rides with that step's first launch
static thread_local NoiseArgs g_noise = {0, 0, 0.0f, 0.0f, nullptr, 0};
template <int G>
static int launch_pos_plus_g(const MjhModel* m, const MjhData* d, int first, int last, bool* sched_done, hipStream_t s) {
  NoiseArgs noise = g_noise;
  g_noise.n = 0;
  const PosLayout lay = pos_layout(m->nq, m->nv, m->nbody, m->njnt, m->nC, last >= POS_FACTOR);
  size_t lds;
  const int threads = pick_block(sizeof(int) * pos_shared_words(m->nv, m->nC, m->nbody, m->njnt, m->nbodylevel, m->ngeom, m->nsite), sizeof(float) * lay.total, G, &lds);
  if (!threads) return fail(MJH_E_UNSUPPORTED, "k_fwd_pos: model does not fit in LDS");
  // The schedule workgroup is this launch's first workgroup.  Round 5: its sort was a chain of 32 dependent memory round trips (13 us: the tail of
  // whichever launch carried it -- fused k_fwd_pos 48 us against 39 us without it); with its loads in one batch (integrate.hpp schedule_body) it
  // costs this launch 2 us.  Same box, steady state / first steps, ms per step: here 0.2914 / 0.2698, as k_mid's last workgroup 0.2902 / 0.2745
  // (MJH_SCHED_IN_MID=1, developer knob).
  static const bool sched_mid = mjh_knob("MJH_SCHED_IN_MID") != nullptr;
  *sched_done = threads >= 128 && !sched_mid;
  if (threads < 128) {
    if (noise.n) hipLaunchKernelGGL(k_ctrl_noise, dim3((noise.n + 255) / 256), dim3(256), 0, s, *m, *d, noise.center, noise.step, noise.noise_std, noise.noise_rate);
    return launch_pos(m, d, first, last, s);
  }
  lds = std::max(lds, (size_t)2048);  // (the schedule workgroup: 512 ints)
  noise.sched = *sched_done ? 1 : 0;
  HIPCHK(set_lds(k_fwd_pos_plus<G>, lds));
  const int wpb = threads / G, npos = (d->nworld + wpb - 1) / wpb, nnoise = (noise.n + threads - 1) / threads;
  debug_occupancy("k_fwd_pos_plus", k_fwd_pos_plus<G>, npos + 1 + nnoise, threads, lds);
  hipLaunchKernelGGL(k_fwd_pos_plus<G>, dim3(npos + 1 + nnoise), dim3(threads), lds, s, *m, *d, first, last, npos, noise);
  return MJH_OK;
}
static int launch_pos_plus(const MjhModel* m, const MjhData* d, int first, int last, bool* sched_done, hipStream_t s) { static const bool pos16 = mjh_knob("MJH_LANES16_POS") != nullptr; return lanes16(m) && pos16 ? launch_pos_plus_g<16>(m, d, first, last, sched_done, s) : lanes64(m) && m->nbody > 32 ? launch_pos_plus_g<64>(m, d, first, last, sched_done, s) : launch_pos_plus_g<32>(m, d, first, last, sched_done, s); }


static int check(const MjhModel* m, const MjhData* d) {
  if (!m || !d) return fail(MJH_E_ARG, "null model/data");
  if (d->nworld <= 0) return fail(MJH_E_ARG, "nworld must be positive");
  if (d->concap <= 0 || !d->ws_contact) return fail(MJH_E_ARG, "Data.ws_contact / concap missing (allocate Data with make_data/put_data)");
  if (m->integrator != INT_EULER && m->integrator != INT_IMPLICITFAST && m->integrator != INT_RK4 && m->integrator != INT_IMPLICIT)
    return fail(MJH_E_UNSUPPORTED, "integrator must be Euler, RK4, implicit or implicitfast");
  if (m->integrator == INT_IMPLICIT && (m->nv > 64 || !d->ws_iacc)) return fail(MJH_E_UNSUPPORTED, "fully implicit integrator: at most 64 dofs, Data allocated for this model");
  if (m->integrator == INT_RK4 && !d->ws_rk) return fail(MJH_E_ARG, "Data.ws_rk missing (allocate Data with make_data/put_data)");
  return MJH_OK;
}

// optional per-kernel event instrumentation used by mjh_timed_steps
struct Instr {
  std::vector<hipEvent_t> ev;  // pairs
  std::vector<int> cls;
  hipStream_t s;
  bool on = false;
  bool plain = false;  // one plain kernel per stage instead of the fused launches
  hipError_t err = hipSuccess;  // first failing hipEvent* call of a Scope (reported by mjh_timed_steps)
};
static thread_local Instr* g_instr = nullptr;
struct Scope {
  int cls;
  hipEvent_t a, b;
  bool on;
  static void note(hipError_t e) {
    if (e != hipSuccess && g_instr && g_instr->err == hipSuccess) g_instr->err = e;
  }
  explicit Scope(int c) : cls(c), a(nullptr), b(nullptr), on(g_instr && g_instr->on) {
    if (on) {
      note(hipEventCreate(&a));
      note(hipEventCreate(&b));
      if (a && b) note(hipEventRecord(a, g_instr->s));
      else on = false;
    }
  }
  ~Scope() {
    if (on) {
      note(hipEventRecord(b, g_instr->s));
      g_instr->ev.push_back(a);
      g_instr->ev.push_back(b);
      g_instr->cls.push_back(cls);
    } else {
      if (a) hipEventDestroy(a);
      if (b) hipEventDestroy(b);
    }
  }
};
enum { K_NOISE = 0, K_POS = 1, K_COLLISION = 2, K_CONSTRAINT = 3, K_VEL = 4, K_SOLVE = 5, K_INTEGRATE = 6, K_OTHER = 7, K_MID = 8 };

// Newton only: the public-output riders (contact publication, L'DL factor + qacc_smooth) cannot ride with the solver
// launch (its 256 VGPRs throttle them) and cost 55 us at the end of the integrator launch; they run on a low-priority
// side stream beside the solver instead.  Two event hops (fork after k_mid, join after the integrator), neither on the
// solver's critical path; created on first use per host thread and device, released by mjh_release_thread_resources().
struct Side {
  hipStream_t stream;
  hipEvent_t fork, join;
  bool owns_stream;
};
static thread_local Side* g_side_per_dev[16] = {nullptr};
static Side* side_stream() {
  Side** per_dev = g_side_per_dev;
  static const bool disabled = mjh_knob("MJH_NO_SIDE") != nullptr;  // developer knob
  if (disabled) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!per_dev[dev]) {
    Side* sd = new Side();
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = greatest = 0;
    // developer knob (A/B): "high" / "normal"; default: the lowest priority.  Measured (round 3, two interleaved triples on one box): no
    // difference (0.2810 / 0.2803 / 0.2805 ms per Newton step) -- queue priority does not arbitrate CU slots.  The riders still end ~20 us
    // after the solver (rocprofv3 timeline, profiles/round3_newton_summary.json): their workgroups only get slots as the solver's retire
    const char* pe = mjh_knob("MJH_SIDE_PRIO");
    const int prio = pe && pe[0] == 'h' ? greatest : (pe && pe[0] == 'n' ? 0 : least);
    // Round 6: the side stream IS the first auxiliary stream of the per-island solver (aux_streams; the two are never used by the same step).
    // A stream of its own cost every later model of the process its solver concurrency: once it existed -- any Newton model of at most 32
    // dofs stepped earlier -- the four class launches of clutter_synth ran 25 % slower (1.43 -> 1.07 M env-steps/s, tools/interference_probe.py:
    // the runtime multiplexes user streams onto a few hardware queues, and streams that share one run in order).  MJH_SIDE_PRIO (a priority of
    // its own) therefore implies a stream of its own.
    Aux* shared = pe ? nullptr : aux_streams();
    sd->owns_stream = shared == nullptr;
    if (shared) sd->stream = shared->stream[0];
    if ((sd->owns_stream && hipStreamCreateWithPriority(&sd->stream, hipStreamNonBlocking, prio) != hipSuccess) ||
        hipEventCreateWithFlags(&sd->fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&sd->join, hipEventDisableTiming) != hipSuccess) {
      delete sd;
      return nullptr;
    }
    per_dev[dev] = sd;
  }
  return per_dev[dev];
}
// one wavefront per world (csrc/sleep.hpp k_sleep)
static int launch_sleep(const MjhModel* m, const MjhData* d, int phase, hipStream_t s) {
  const size_t lds = sizeof(int) * sleep_lds(m->ntree, m->nbody, m->nv, d->njmax, d->concap).total;
  if (lds > (size_t)kLdsPerCU) return fail(MJH_E_UNSUPPORTED, "k_sleep: the world's sleep tables do not fit in LDS");
  HIPCHK(set_lds(k_sleep, lds));
  hipLaunchKernelGGL(k_sleep, dim3(d->nworld), dim3(64), lds, s, *m, *d, phase);
  return MJH_OK;
}
// forward / step of a model with sleeping enabled (forward.py:345-349, 652-675, 1290-1324, 1341-1347): the staged launch sequence
// with the sleep bookkeeping between the stages (csrc/sleep.hpp)
static int run_sleep_step(const MjhModel* m, const MjhData* d, bool step, hipStream_t s) {
  if (!d->tree_asleep || !d->ws_sleep_J) return fail(MJH_E_ARG, "Data sleep tables missing (allocate Data with make_data/put_data)");
  if (m->solver != SOL_NEWTON) return fail(MJH_E_UNSUPPORTED, "sleeping requires the Newton solver (reference io.py:359)");
  if (step && m->integrator == INT_RK4) return fail(MJH_E_UNSUPPORTED, "sleeping with the RK4 integrator");
  const int mode = m->integrator == INT_IMPLICITFAST ? 1 : (m->integrator == INT_IMPLICIT ? 2 : 0);
  { Scope sc(K_OTHER); TRY(launch_sleep(m, d, (int)SLP_WAKE, s)); }
  { Scope sc(K_POS); TRY(launch_pos(m, d, POS_KINEMATICS, POS_CRB, s)); }
  { Scope sc(K_COLLISION); TRY(launch_collision(m, d, s)); }
  { Scope sc(K_OTHER); TRY(launch_sleep(m, d, (int)SLP_WAKE_COLLISION, s)); }
  {
    // pass 2: waking only ever lets more pairs through the sleep filter, so the pairs of pass 1 plus "the pairs pass 1 skipped that
    // involve a newly awakened body" are the pairs that pass the filter now: the woken worlds recompute their list
    Scope sc(K_COLLISION);
    MjhData d2 = *d;
    d2.sleep_pass = 2;
    TRY(launch_collision(m, &d2, s));
  }
  { Scope sc(K_CONSTRAINT); TRY(launch_constraint(m, d, s)); }
  { Scope sc(K_OTHER); TRY(launch_sleep(m, d, (int)SLP_POST_CONSTRAINT, s)); }
  { Scope sc(K_VEL); TRY(launch_vel(m, d, VEL_COMVEL, VEL_ACCEL, s)); }
  { Scope sc(K_OTHER); TRY(launch_sensor(m, d, 0, s)); }
  {
    Scope sc(K_OTHER);
    if (m->nv > 0) hipLaunchKernelGGL(k_sleep_qfrc, dim3((d->nworld * m->nv + 255) / 256), dim3(256), 0, s, *m, *d);
    hipLaunchKernelGGL(k_sleep_mask, dim3(d->nworld), dim3(256), 0, s, *m, *d);
  }
  {
    Scope sc(K_SOLVE);
    MjhData d3 = *d;  // the solver reads the masked copies (sleeping dofs frozen), and writes the public outputs
    d3.efc_J = d->ws_sleep_J;
    d3.qacc_warmstart = d->ws_sleep_warm;
    TRY(launch_solve(m, &d3, s));
  }
  { Scope sc(K_OTHER); TRY(launch_sensor(m, d, 1, s)); }
  if (step) { Scope sc(K_INTEGRATE); TRY(launch_integrate(m, d, mode, s)); }
  {
    // (round 6: these two public-output launches beside the solver on the side stream LOSE here -- clutter_synth 1.53 -> 1.36 M env-steps/s, two
    // interleaved pairs: their workgroups take LDS slots from the island solves, which are the step's critical path)
    Scope sc(K_OTHER);
    TRY(launch_publish(m, d, s));
    TRY(launch_factor_smooth(m, d, 1, s));
  }
  if (step) {
    { Scope sc(K_OTHER); TRY(launch_sleep(m, d, (int)SLP_SLEEP, s)); }
    { Scope sc(K_VEL); TRY(launch_vel(m, d, VEL_COMVEL, VEL_RNE, s)); }
  }
  return MJH_OK;
}

static int run_stage(const MjhModel* m, const MjhData* d, int stage, hipStream_t s) {
  switch (stage) {
    case MJH_STAGE_KINEMATICS: { Scope sc(K_POS); return launch_pos(m, d, POS_KINEMATICS, POS_KINEMATICS, s); }
    case MJH_STAGE_COM_POS: { Scope sc(K_POS); return launch_pos(m, d, POS_COM, POS_COM, s); }
    case MJH_STAGE_CRB: { Scope sc(K_POS); return launch_pos(m, d, POS_CRB, POS_CRB, s); }
    case MJH_STAGE_FACTOR_M: { Scope sc(K_POS); return launch_pos(m, d, POS_FACTOR, POS_FACTOR, s); }
    case MJH_STAGE_COLLISION:
      { Scope sc(K_COLLISION); TRY(launch_collision(m, d, s)); }
      { Scope sc(K_OTHER); return launch_publish(m, d, s); }
    case MJH_STAGE_MAKE_CONSTRAINT:
      { Scope sc(K_CONSTRAINT); TRY(launch_constraint(m, d, s)); }
      { Scope sc(K_OTHER); return launch_publish(m, d, s); }
    case MJH_STAGE_TRANSMISSION: return MJH_OK;  // joint transmissions: length/moment are produced by fwd_actuation
    case MJH_STAGE_COM_VEL: { Scope sc(K_VEL); return launch_vel(m, d, VEL_COMVEL, VEL_COMVEL, s); }
    case MJH_STAGE_PASSIVE: { Scope sc(K_VEL); return launch_vel(m, d, VEL_PASSIVE, VEL_PASSIVE, s); }
    case MJH_STAGE_RNE: { Scope sc(K_VEL); return launch_vel(m, d, VEL_RNE, VEL_RNE, s); }
    case MJH_STAGE_FWD_VELOCITY: { Scope sc(K_VEL); return launch_vel(m, d, VEL_COMVEL, VEL_RNE, s); }
    case MJH_STAGE_FWD_ACTUATION: { Scope sc(K_VEL); return launch_vel(m, d, VEL_ACTUATION, VEL_ACTUATION, s); }
    case MJH_STAGE_FWD_ACCELERATION:
      { Scope sc(K_VEL); TRY(launch_vel(m, d, VEL_ACCEL, VEL_ACCEL, s)); }
      { Scope sc(K_OTHER); return launch_factor_smooth(m, d, 1, s); }
    case MJH_STAGE_SOLVE: { Scope sc(K_SOLVE); return launch_solve(m, d, s); }
    case MJH_STAGE_EULER: { Scope sc(K_INTEGRATE); return launch_integrate(m, d, 0, s); }
    case MJH_STAGE_IMPLICIT: { Scope sc(K_INTEGRATE); return launch_integrate(m, d, m->integrator == INT_IMPLICIT ? 2 : 1, s); }
    case MJH_STAGE_FWD_POSITION:
      { Scope sc(K_POS); TRY(launch_pos(m, d, POS_KINEMATICS, POS_FACTOR, s)); }
      { Scope sc(K_COLLISION); TRY(launch_collision(m, d, s)); }
      { Scope sc(K_CONSTRAINT); TRY(launch_constraint(m, d, s)); }
      { Scope sc(K_OTHER); TRY(launch_publish(m, d, s)); }
      return MJH_OK;
    case MJH_STAGE_RNE_POSTCONSTRAINT: {
      if (!d->cfrc_ext) return fail(MJH_E_ARG, "Data.cfrc_ext missing");
      Scope sc(K_OTHER);
      TRY(launch_rne_postconstraint(m, d, s));
      return MJH_OK;
    }
    case MJH_STAGE_SUBTREE_VEL: {
      if (!d->subtree_linvel || !d->subtree_angmom) return fail(MJH_E_ARG, "Data.subtree_linvel / subtree_angmom missing");
      Scope sc(K_OTHER);
      TRY(launch_subtree_vel(m, d, s));
      return MJH_OK;
    }
    case MJH_STAGE_ENERGY: {
      if (!d->energy) return fail(MJH_E_ARG, "Data.energy missing (allocate Data with make_data/put_data)");
      Scope sc(K_OTHER);
      hipLaunchKernelGGL(k_energy<32>, dim3((d->nworld + 7) / 8), dim3(256), 0, s, *m, *d);
      return MJH_OK;
    }
    case MJH_STAGE_SENSOR: { Scope sc(K_OTHER); TRY(launch_sensor(m, d, 0, s)); return launch_sensor(m, d, 1, s); }
    case MJH_STAGE_SENSOR_POSVEL: { Scope sc(K_OTHER); return launch_sensor(m, d, 0, s); }
    case MJH_STAGE_SENSOR_ACC: { Scope sc(K_OTHER); return launch_sensor(m, d, 1, s); }
    case MJH_STAGE_UPDATE_SLEEP:
    case MJH_STAGE_WAKE:
    case MJH_STAGE_WAKE_COLLISION:
    case MJH_STAGE_WAKE_EQUALITY:
    case MJH_STAGE_ISLAND:
    case MJH_STAGE_SLEEP: {
      if (!d->tree_asleep) return fail(MJH_E_ARG, "Data sleep tables missing (allocate Data with make_data/put_data)");
      const int phase = stage == MJH_STAGE_UPDATE_SLEEP ? SLP_UPDATE : stage == MJH_STAGE_WAKE ? SLP_WAKE : stage == MJH_STAGE_WAKE_COLLISION ? SLP_WAKE_COLLISION
                        : stage == MJH_STAGE_WAKE_EQUALITY ? SLP_WAKE_EQUALITY : stage == MJH_STAGE_ISLAND ? SLP_ISLAND : SLP_SLEEP;
      Scope sc(K_OTHER);
      TRY(launch_sleep(m, d, phase, s));
      return MJH_OK;
    }
    case MJH_STAGE_RUNGEKUTTA4: {
      // forward.rungekutta4 (forward.py:524-557), called after a forward at t0: one k_rk4 launch after each evaluation
      // (accumulate + perturb; the last one restores t0 and advances); tableau A = (1/2, 1/2, 1), B = (1/6, 1/3, 1/3, 1/6)
      if (!d->ws_rk) return fail(MJH_E_ARG, "Data.ws_rk missing (allocate Data with make_data/put_data)");
      static const float A[4] = {0.5f, 0.5f, 1.0f, 0.0f}, B[4] = {1.0f / 6.0f, 1.0f / 3.0f, 1.0f / 3.0f, 1.0f / 6.0f};
      for (int k = 0; k < 4; ++k) {
        if (k) TRY(run_stage(m, d, MJH_STAGE_FORWARD, s));
        Scope sc(K_INTEGRATE);
        hipLaunchKernelGGL(k_rk4<32>, dim3((d->nworld + 7) / 8), dim3(8 * 32), sizeof(float) * 8 * (m->nv + 1), s, *m, *d, k, A[k], B[k]);
      }
      return MJH_OK;
    }
    case MJH_STAGE_FORWARD:
    case MJH_STAGE_STEP: {
      if (m->sleep_enabled) return run_sleep_step(m, d, stage == MJH_STAGE_STEP, s);
      if (stage == MJH_STAGE_STEP && m->integrator == INT_RK4) {
        TRY(run_stage(m, d, MJH_STAGE_FORWARD, s));
        return run_stage(m, d, MJH_STAGE_RUNGEKUTTA4, s);
      }
      const int mode = m->integrator == INT_IMPLICITFAST ? 1 : (m->integrator == INT_IMPLICIT ? 2 : 0);
      static const bool plain = mjh_knob("MJH_PLAIN") != nullptr;  // developer knob: one plain kernel per stage, serial
      if ((g_instr && g_instr->on && g_instr->plain) || plain) {
        // profiling pass: one plain kernel per stage, so that the event pairs time one kernel at a time
        { Scope sc(K_OTHER); hipLaunchKernelGGL(k_schedule_worlds, dim3(1), dim3(1024), 0, s, *d, m->nv > 32 ? 64 : ((m->solver == SOL_NEWTON && m->cone == CONE_ELLIPTIC && d->njmax > 32) ? 32 : 0)); /* = sched_cls, integrate.hpp */ }
        { Scope sc(K_POS); TRY(launch_pos(m, d, POS_KINEMATICS, POS_CRB, s)); }
        { Scope sc(K_COLLISION); TRY(launch_collision(m, d, s)); }
        { Scope sc(K_CONSTRAINT); TRY(launch_constraint(m, d, s)); }
        { Scope sc(K_VEL); TRY(launch_vel(m, d, VEL_COMVEL, VEL_ACCEL, s)); }
        { Scope sc(K_OTHER); TRY(launch_sensor(m, d, 0, s)); }
        { Scope sc(K_SOLVE); TRY(launch_solve(m, d, s)); }
        { Scope sc(K_OTHER); TRY(launch_sensor(m, d, 1, s)); }
        if (stage == MJH_STAGE_STEP) { Scope sc(K_INTEGRATE); TRY(launch_integrate(m, d, mode, s)); }
        Scope sc(K_OTHER);
        TRY(launch_publish(m, d, s));
        TRY(launch_factor_smooth(m, d, m->solver == SOL_NEWTON ? 1 : 0, s));  // CG and PGS write qacc_smooth themselves
        return MJH_OK;
      }
      // fused step: four launches on the caller's stream (see "composite launches" above)
      // (nv <= 32 only: beside the 64-lane solver of larger models the riders cost more than they save, G1 -3 %)
      static const int side_nv = mjh_knob("MJH_SIDE_NV") ? atoi(mjh_knob("MJH_SIDE_NV")) : 32;  // developer knob
      const bool inl = stage == MJH_STAGE_STEP && newton_inline_ok(m, d) && !(g_instr && g_instr->on);
      Side* side = (m->solver == SOL_NEWTON && m->nv <= side_nv && !(g_instr && g_instr->on) && !inl) ? side_stream() : nullptr;
      bool sched_done = false;
      { Scope sc(K_POS); TRY(launch_pos_plus(m, d, POS_KINEMATICS, POS_CRB, &sched_done, s)); }
      { Scope sc(K_MID); TRY(launch_mid(m, d, !sched_done, s)); }
      { Scope sc(K_OTHER); TRY(launch_sensor(m, d, 0, s)); }  // (no launch without sensors; before the solver, whose epilogue may integrate the state)
      if (side) {
        // (tried in round 3: forking the factor after k_fwd_pos, beside k_mid, with qacc_smooth as a separate solve after k_mid -- humanoid
        // Newton + 1.5 %, Panda - 5 %: the extra launch and the slower k_mid cost more than the shorter join wait saves)
        HIPCHK(hipEventRecord(side->fork, s));
        HIPCHK(hipStreamWaitEvent(side->stream, side->fork, 0));
        // both riders as roles of ONE launch (k_integrate_plus without integrator workgroups): their two chains overlap and one launch gap
        // goes (round 3, same-box A/B in two interleaved pairs: humanoid Newton 0.2842 -> 0.2829 / 0.2835 -> 0.2822 ms, Panda 142.9 -> 142.0 us)
        static const bool side_two = mjh_knob("MJH_SIDE_TWO") != nullptr;  // developer knob (A/B): the two plain kernels of round 2
        if (!side_two) {
          TRY(launch_integrate_plus(m, d, mode, false, side->stream));
        } else {
          TRY(launch_publish(m, d, side->stream));
          TRY(launch_factor_smooth(m, d, 1, side->stream));
        }
        HIPCHK(hipEventRecord(side->join, side->stream));
      }
      // explicit Euler without activations: the velocity/position update is a few loads and stores per dof, done by the
      // solver's own epilogue (saves a launch); every other case keeps the integrator workgroups
      // (Newton: only when its riders run on the side stream -- otherwise the integrator launch exists anyway, for them)
      const bool fusable = stage == MJH_STAGE_STEP && m->na == 0 && (m->solver == SOL_CG || side != nullptr || inl) && m->nv <= 64 &&
                           m->nsensor_acc == 0 &&  // (acceleration-stage sensors read qvel / qacc between the solver and the integrator)
                           d->njmax <= 192;        // (beyond: some worlds go to the generic solver, which does not integrate)
      // implicitfast without activations (round 3): the dense system (M + h D - h dA/dv) x = M qacc is solved from the M row the solver holds
      static const bool no_fuse_impfast = mjh_knob("MJH_NO_FUSE_IMPLICITFAST") != nullptr;  // developer knob (A/B)
      const int fuse_euler = !fusable ? 0
                             : (m->integrator == INT_EULER && (m->disableflags & (DSBL_EULERDAMP | DSBL_DAMPER)) != 0) ? 1
                             : (m->integrator == INT_IMPLICITFAST && !no_fuse_impfast && !m->act_velfeedback) ? 2 : 0;  // (positive velocity feedback: the matrix may be indefinite -- the integrator launch's L'DL handles that, the epilogue's Cholesky does not)
      g_fuse_euler = fuse_euler;
      g_newton_inline = inl;
      int rc;
      { Scope sc(K_SOLVE); rc = launch_solve_plus(m, d, s); }
      g_fuse_euler = 0;
      g_newton_inline = false;
      TRY(rc);
      { Scope sc(K_OTHER); TRY(launch_sensor(m, d, 1, s)); }
      g_riders_on_side = side != nullptr || inl;  // (either way the integrator launch carries no riders)
      { Scope sc(K_INTEGRATE); rc = launch_integrate_plus(m, d, mode, stage == MJH_STAGE_STEP && !fuse_euler, s); }
      g_riders_on_side = false;
      TRY(rc);
      if (side) HIPCHK(hipStreamWaitEvent(s, side->join, 0));  // every fork rejoins the caller's stream
      return MJH_OK;
    }
    default:
      return fail(MJH_E_ARG, "unknown stage");
  }
}

template <int G>
static int launch_solve_m_g(const MjhModel* m, const MjhData* d, float* x, const float* y, int mul, hipStream_t s) {
  const IntLayout lay = int_layout(m->nv, m->nC);
  size_t lds;
  const int threads = pick_block(sizeof(int) * mstruct_ints(m->nv, m->nC), sizeof(float) * lay.total, G, &lds);
  if (!threads) return fail(MJH_E_UNSUPPORTED, "k_solve_m: does not fit in LDS");
  HIPCHK(set_lds(k_solve_m<G>, lds));
  const int wpb = threads / G;
  hipLaunchKernelGGL(k_solve_m<G>, dim3((d->nworld + wpb - 1) / wpb), dim3(threads), lds, s, *m, *d, x, y, mul);
  HIPCHK(hipGetLastError());
  return MJH_OK;
}
static int launch_solve_m(const MjhModel* m, const MjhData* d, float* x, const float* y, int mul, hipStream_t s) { return launch_solve_m_g<32>(m, d, x, y, mul, s); }  // (chains over the sparse factor: more lanes per world only halve the worlds per wavefront)

// the C ABI is the only exported surface (the library is built with -fvisibility=hidden)
#pragma GCC visibility push(default)
extern "C" {

int mjh_abi_version(void) { return MJH_ABI_VERSION; }
int mjh_ws_ccd_floats(int nworld, int iterations, int nhfield, int npolygonmax, int nmeshdegmax, int npair, int concap, double* floats_out, int* nccdhand_out) {
  if (!floats_out || nworld < 0 || npair < 0 || concap < 0) return fail(MJH_E_ARG, "mjh_ws_ccd_floats: bad argument");
  const int ccap = collide_ccap(npair, concap), hand = ccd_handcap(nworld, ccap);
  *floats_out = (double)ccd_layout(nworld, iterations, nhfield, npolygonmax, nmeshdegmax, ccap, hand, npair).total;
  if (nccdhand_out) *nccdhand_out = hand;
  return MJH_OK;
}
const char* mjh_last_error(void) { return g_err; }
int mjh_dev_knob(const char* name, const char* value) {
  if (!name || strncmp(name, "MJH_", 4) != 0) return fail(MJH_E_ARG, "mjh_dev_knob: knob names start with MJH_");
  KnobTable& t = knobs();
  std::lock_guard<std::mutex> lock(t.mu);
  if (value) t.kv[name] = value;
  else t.kv.erase(name);
  t.n.store((int)t.kv.size(), std::memory_order_release);
  return MJH_OK;
}
const char* mjh_solver_kernel(const MjhModel* m, const MjhData* d) { return (m && d) ? solver_kernel_name(m, d) : "unsupported"; }

int mjh_stage(const MjhModel* m, const MjhData* d, int stage, void* stream) {
  TRY(check(m, d));
  TRY(run_stage(m, d, stage, (hipStream_t)stream));
  HIPCHK(hipGetLastError());
  return MJH_OK;
}
int mjh_step(const MjhModel* m, const MjhData* d, void* stream) { return mjh_stage(m, d, MJH_STAGE_STEP, stream); }
int mjh_forward(const MjhModel* m, const MjhData* d, void* stream) { return mjh_stage(m, d, MJH_STAGE_FORWARD, stream); }

int mjh_solve_m(const MjhModel* m, const MjhData* d, float* x, const float* y, void* stream) {
  TRY(check(m, d));
  return launch_solve_m(m, d, x, y, 0, (hipStream_t)stream);
}
int mjh_mul_m(const MjhModel* m, const MjhData* d, float* res, const float* vec, void* stream) {
  TRY(check(m, d));
  return launch_solve_m(m, d, res, vec, 1, (hipStream_t)stream);
}

int mjh_qld_dense(const MjhModel* m, const MjhData* d, float* qld_dense, int stride, void* stream) {
  TRY(check(m, d));
  if (!qld_dense || stride < 0) return fail(MJH_E_ARG, "mjh_qld_dense: null output");
  if (m->ntree == 0 || d->nworld == 0) return MJH_OK;
  hipLaunchKernelGGL(k_qld_dense, dim3(d->nworld * m->ntree), dim3(64), 0, (hipStream_t)stream, *m, *d, qld_dense, stride);
  HIPCHK(hipGetLastError());
  return MJH_OK;
}

int mjh_contact_force(const MjhModel* m, const MjhData* d, const int* contact_ids, int n, int to_world_frame, float* force, void* stream) {
  TRY(check(m, d));
  if (n < 0 || (n > 0 && (!contact_ids || !force))) return fail(MJH_E_ARG, "mjh_contact_force: null argument");
  if (n == 0) return MJH_OK;
  hipLaunchKernelGGL(k_contact_force, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, *m, *d, contact_ids, n, to_world_frame, force);
  HIPCHK(hipGetLastError());
  return MJH_OK;
}

int mjh_jac(const MjhModel* m, const MjhData* d, float* jacp, float* jacr, const float* point, const int* body, void* stream) {
  TRY(check(m, d));
  if (!point || !body) return fail(MJH_E_ARG, "mjh_jac: null point / body");
  if (m->nv == 0 || (!jacp && !jacr)) return MJH_OK;
  hipLaunchKernelGGL(k_jac, dim3((d->nworld * m->nv + 255) / 256), dim3(256), 0, (hipStream_t)stream, *m, *d, jacp, jacr, point, body);
  HIPCHK(hipGetLastError());
  return MJH_OK;
}

int mjh_rays(const MjhModel* m, const MjhData* d, const float* pnt, const float* vec, int pnt_nworld, int nray, const float* geomgroup, int flg_static,
             const int* bodyexclude, float* dist, int* geomid, float* normal, void* stream) {
  TRY(check(m, d));
  if (!pnt || !vec || !dist) return fail(MJH_E_ARG, "mjh_rays: null pnt / vec / dist");
  if (nray < 0 || (pnt_nworld != 1 && pnt_nworld != d->nworld)) return fail(MJH_E_ARG, "mjh_rays: pnt_nworld must be 1 or nworld");
  if (nray == 0) return MJH_OK;
  if ((long long)d->nworld * nray > 0x7fffffffLL) return fail(MJH_E_ARG, "mjh_rays: nworld * nray exceeds 2^31 - 1 (cast the rays in several calls)");
  RayGroup gg;
  for (int i = 0; i < 6; ++i) gg.g[i] = geomgroup ? geomgroup[i] : -1.0f;
  hipLaunchKernelGGL(k_rays, dim3((d->nworld * nray + 255) / 256), dim3(256), 0, (hipStream_t)stream, *m, *d, pnt, vec, pnt_nworld, nray, gg, flg_static, bodyexclude, dist,
                     geomid, normal);
  HIPCHK(hipGetLastError());
  return MJH_OK;
}

int mjh_efc_j_sparse(const MjhModel* m, const MjhData* d, int njmax_nnz, int* rownnz, int* rowadr, int* colind, float* values, void* stream) {
  TRY(check(m, d));
  if (njmax_nnz < 0 || !rownnz || !rowadr || (njmax_nnz > 0 && (!colind || !values))) return fail(MJH_E_ARG, "mjh_efc_j_sparse: null output or negative njmax_nnz");
  hipLaunchKernelGGL(k_efc_j_sparse, dim3(d->nworld), dim3(64), 0, (hipStream_t)stream, *d, m->nv, njmax_nnz, rownnz, rowadr, colind, values);
  HIPCHK(hipGetLastError());
  return MJH_OK;
}

int mjh_ctrl_noise(const MjhModel* m, const MjhData* d, const float* ctrl_center, int step, float noise_std, float noise_rate, void* stream) {
  TRY(check(m, d));
  const int n = d->nworld * m->nu;
  if (n == 0) return MJH_OK;
  Scope sc(K_NOISE);
  hipLaunchKernelGGL(k_ctrl_noise, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, *m, *d, ctrl_center, step, noise_std, noise_rate);
  HIPCHK(hipGetLastError());
  return MJH_OK;
}

int mjh_release_thread_resources(void) {
  // the calling thread's side streams and events (one set per device it stepped a Newton model on); the Newton side stream first: it may be
  // an alias of the first auxiliary stream
  int rc = MJH_OK;
  for (int dev = 0; dev < 16; ++dev) {
    Side* sd = g_side_per_dev[dev];
    if (!sd) continue;
    g_side_per_dev[dev] = nullptr;
    if (hipStreamSynchronize(sd->stream) != hipSuccess) rc = MJH_E_LAUNCH;
    if (hipEventDestroy(sd->fork) != hipSuccess) rc = MJH_E_LAUNCH;
    if (hipEventDestroy(sd->join) != hipSuccess) rc = MJH_E_LAUNCH;
    if (sd->owns_stream && hipStreamDestroy(sd->stream) != hipSuccess) rc = MJH_E_LAUNCH;
    delete sd;
  }
  for (int dev = 0; dev < 16; ++dev) {
    Aux* a = g_aux_per_dev[dev];
    if (a) {
      g_aux_per_dev[dev] = nullptr;
      for (int k = 0; k < MJH_NAUX; ++k) {
        if (hipStreamSynchronize(a->stream[k]) != hipSuccess) rc = MJH_E_LAUNCH;
        if (hipEventDestroy(a->join[k]) != hipSuccess) rc = MJH_E_LAUNCH;
        if (hipStreamDestroy(a->stream[k]) != hipSuccess) rc = MJH_E_LAUNCH;
      }
      if (hipEventDestroy(a->fork) != hipSuccess) rc = MJH_E_LAUNCH;
      delete a;
    }
  }
  return rc == MJH_OK ? MJH_OK : fail(rc, "mjh_release_thread_resources: %s", "a HIP call failed");
}

int mjh_graph_create(const MjhModel* m, const MjhData* d, void* stream, void** graph_exec_out) {
  TRY(check(m, d));
  hipStream_t s = (hipStream_t)stream;
  // warm the kernels (function attributes must be set outside capture)
  TRY(run_stage(m, d, MJH_STAGE_STEP, s));
  HIPCHK(hipStreamSynchronize(s));
  hipGraph_t graph;
  HIPCHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  int rc = run_stage(m, d, MJH_STAGE_STEP, s);
  hipError_t e = hipStreamEndCapture(s, &graph);  // always end the capture so the stream stays usable
  if (rc != MJH_OK) return rc;
  HIPCHK(e);
  hipGraphExec_t exec;
  HIPCHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  hipGraphDestroy(graph);
  *graph_exec_out = (void*)exec;
  return MJH_OK;
}
int mjh_graph_launch(void* graph_exec, void* stream) {
  HIPCHK(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
  return MJH_OK;
}
int mjh_graph_destroy(void* graph_exec) {
  HIPCHK(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
  return MJH_OK;
}

int mjh_timed_steps(const MjhModel* m, const MjhData* d, int nstep, int step0, float noise_std, float noise_rate,
                    void* stream, float* ms_out, float* per_kernel_ms, int plain_kernels) {
  TRY(check(m, d));
  hipStream_t s = (hipStream_t)stream;
  Instr instr;
  instr.s = s;
  instr.on = per_kernel_ms != nullptr;
  instr.plain = plain_kernels != 0;
  g_instr = &instr;
  g_serial_solver = instr.on;
  hipEvent_t t0, t1;
  HIPCHK(hipEventCreate(&t0));
  HIPCHK(hipEventCreate(&t1));
  HIPCHK(hipEventRecord(t0, s));
  int rc = MJH_OK;
  for (int i = 0; i < nstep && rc == MJH_OK; ++i) {
    // the noise of step i rides with that step's first launch (the fused path); the per-kernel profiling passes and the plain
    // path keep it as its own kernel
    static const bool plain_env = mjh_knob("MJH_PLAIN") != nullptr || mjh_knob("MJH_NO_NOISE_FUSION") != nullptr;
    if (noise_std >= 0.0f && m->nu > 0) {
      if (instr.on || plain_env) rc = mjh_ctrl_noise(m, d, nullptr, step0 + i, noise_std, noise_rate, stream);
      else g_noise = NoiseArgs{d->nworld * m->nu, step0 + i, noise_std, noise_rate, nullptr, 0};
    }
    if (rc == MJH_OK) rc = run_stage(m, d, MJH_STAGE_STEP, s);
    g_noise.n = 0;
  }
  hipError_t e = hipEventRecord(t1, s);
  if (e == hipSuccess) e = hipEventSynchronize(t1);
  if (e == hipSuccess) e = instr.err;
  g_instr = nullptr;
  g_serial_solver = false;
  float ms = 0.0f;
  hipEventElapsedTime(&ms, t0, t1);
  if (ms_out) *ms_out = ms;
  if (per_kernel_ms) {
    for (int k = 0; k < MJH_NKERNEL; ++k) per_kernel_ms[k] = 0.0f;
    for (size_t i = 0; i < instr.cls.size(); ++i) {
      float t = 0.0f;
      hipEventElapsedTime(&t, instr.ev[2 * i], instr.ev[2 * i + 1]);
      per_kernel_ms[instr.cls[i]] += t;
      hipEventDestroy(instr.ev[2 * i]);
      hipEventDestroy(instr.ev[2 * i + 1]);
    }
  }
  hipEventDestroy(t0);
  hipEventDestroy(t1);
  if (rc != MJH_OK) return rc;
  HIPCHK(e);
  HIPCHK(hipGetLastError());
  return MJH_OK;
}

#ifdef MJH_PHASE_CLOCK
// profiling builds only (tools/phase_clock.py): read (and optionally reset) the per-kernel, per-phase tick sums
int mjh_debug_phase_ticks(unsigned long long* out, int reset) {
  if (out) HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_ticks), sizeof(unsigned long long) * 64 * 8 * 16));
  if (reset) {
    static unsigned long long zeros[64 * 8 * 16] = {0};
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_ticks), zeros, sizeof(zeros)));
  }
  return MJH_OK;
}
#endif

}  // extern "C"
#pragma GCC visibility pop
