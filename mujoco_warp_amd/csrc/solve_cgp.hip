// solve_cgp.hip -- k_solve_cgp_plus: CG with every world of the batch resident at once (solver_cgp.hpp), with the fused step's riders
// (one translation unit of libmjhip.so, see host.hpp)
#include "host.hpp"

#include "contact_rec.hpp"
#include "smooth.hpp"
#include "solver_cgp.hpp"

#ifndef CGP_WAVES
#define CGP_WAVES 3
#endif
#define CGP_MAXT (256 * CGP_WAVES)
// CGP_WAVES wavefronts per SIMD: the register budget (168 at 3, 128 at 4) the design rests on
template <int NV4>
__global__ void __launch_bounds__(CGP_MAXT) __attribute__((amdgpu_waves_per_eu(CGP_WAVES, CGP_WAVES))) k_solve_cgp_plus(MjhModel m, MjhData d, int nsolve, int nfac, int fuse_euler, int pool_rows) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wpb = blockDim.x / 32;
  if ((int)blockIdx.x < nsolve) solve_cgp_body<NV4>(m, d, smem, (int)blockIdx.x * wpb, wpb, pool_rows, fuse_euler);
  else {  // riders of the fused step (see k_solve_plus): L'DL factor and contact publication
    const int bi = (int)blockIdx.x - nsolve;
    if (bi < nfac) factor_smooth_body<32>(m, d, 0, smem, Blk{bi * wpb, wpb, (int)blockDim.x});
    else publish_body<32>(d, 1, reinterpret_cast<int*>(smem), Blk{(bi - nfac) * wpb, wpb, (int)blockDim.x}, m.nexplicit ? m.pair_solreffriction : nullptr);
  }
}
template <int NV4>
static int launch_cgp_t(const MjhModel* m, const MjhData* d, bool with_factor, int fuse_euler, hipStream_t s) {
  // workgroup size: two wavefronts = four worlds sharing 26 KB (2 CGP_WAVES workgroups per CU).  Developer knob MJH_CGP_THREADS: any multiple
  // of 64 up to 256 CGP_WAVES -- larger workgroups pool more worlds but retire (and hand their LDS on) only when their last world is done:
  // measured at 8192 humanoids, steady state / first steps: 64 threads 168.3 / 172.9 us, 128: 166.9 / 168.9, 192: 170.2 / 173.3; 384 and
  // 768 (an earlier build): 255 and 202 against 182 for 64
  int threads = 128;
  if (const char* e = mjh_knob("MJH_CGP_THREADS")) threads = std::min(CGP_MAXT, std::max(64, (atoi(e) / 64) * 64));
  size_t lds = ((size_t)kLdsPerCU / (256 * CGP_WAVES / threads)) & ~(size_t)1023;  // (whole KB: the allocation granularity must not cost a workgroup)
  if (const char* e = mjh_knob("MJH_CGP_LDS")) lds = (size_t)atoi(e);  // developer knob: bytes of the header + pool of a workgroup
  const int wpb = threads / 32;
  const int pool_rows = cgp_pool_rows<NV4>(lds, wpb);
  if (pool_rows < cgp_min_rows<NV4>(fuse_euler)) return fail(MJH_E_UNSUPPORTED, "k_solve_cgp: pool too small");
  const FacLayout fl = fac_layout(m->nv, m->nC);
  if (with_factor) lds = std::max(lds, sizeof(int) * mstruct_ints(m->nv, m->nC) + sizeof(float) * fl.total * wpb);  // the riders' tables
  if (lds > (size_t)kLdsPerCU) return fail(MJH_E_UNSUPPORTED, "k_solve_cgp: the workgroup does not fit in LDS");
  HIPCHK(set_lds(k_solve_cgp_plus<NV4>, lds));
  const int nsolve = (d->nworld + wpb - 1) / wpb, nfac = with_factor ? nsolve : 0;
  debug_occupancy("k_solve_cgp_plus", k_solve_cgp_plus<NV4>, nsolve + 2 * nfac, threads, lds);
  hipLaunchKernelGGL(k_solve_cgp_plus<NV4>, dim3(nsolve + 2 * nfac), dim3(threads), lds, s, *m, *d, nsolve, nfac, fuse_euler, pool_rows);
  return MJH_OK;
}
int launch_solve_cgp(const MjhModel* m, const MjhData* d, bool with_factor, int fuse_euler, hipStream_t s) {
  switch ((m->nv + 3) / 4) {
    case 0:
    case 1: return launch_cgp_t<1>(m, d, with_factor, fuse_euler, s);
    case 2: return launch_cgp_t<2>(m, d, with_factor, fuse_euler, s);
    case 3: return launch_cgp_t<3>(m, d, with_factor, fuse_euler, s);
    case 4: return launch_cgp_t<4>(m, d, with_factor, fuse_euler, s);
    case 5: return launch_cgp_t<5>(m, d, with_factor, fuse_euler, s);
    case 6: return launch_cgp_t<6>(m, d, with_factor, fuse_euler, s);
    case 7: return launch_cgp_t<7>(m, d, with_factor, fuse_euler, s);
    default: return launch_cgp_t<8>(m, d, with_factor, fuse_euler, s);
  }
}

#ifdef MJH_PHASE_CLOCK
// profiling variant (tools/build_variant_fast.py clkp solve_cgp.hip -DMJH_PHASE_CLOCK; tools/phase_clock.py --lib ...): this unit's copy of the
// per-phase tick sums of solve_cgp_body
extern "C" __attribute__((visibility("default"))) int mjh_debug_phase_ticks(unsigned long long* out, int reset) {
  if (out) HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_ticks), sizeof(unsigned long long) * 64 * 8 * 16));
  if (reset) {
    static unsigned long long zeros[64 * 8 * 16] = {0};
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_ticks), zeros, sizeof(zeros)));
  }
  return MJH_OK;
}
#endif
