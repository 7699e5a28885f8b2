// solve_cgw.hip -- k_solve_cgw_plus: CG, one world per wavefront (solver_cgw.hpp), with the fused step's riders (one translation unit of
// libmjhip.so, see host.hpp)
#include "host.hpp"

#include "contact_rec.hpp"
#include "smooth.hpp"
#include "solver_cgw.hpp"

#ifndef MJH_CGW_WAVES
#define MJH_CGW_WAVES 4
#endif

template <int NV4>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MJH_CGW_WAVES, 8))) k_solve_cgw_plus(MjhModel m, MjhData d, int nsolve, int nfac, int nefc_lo, int nefc_hi, int fuse_euler) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wpb = blockDim.x / 64;
  if ((int)blockIdx.x < nsolve) solve_cgw_body<NV4>(m, d, smem, Blk{(int)blockIdx.x * wpb, wpb, (int)blockDim.x}, nefc_lo, nefc_hi, fuse_euler);
  else {  // riders of the fused step (see k_solve_plus): L'DL factor and contact publication
    const int wf = blockDim.x / 32, bi = (int)blockIdx.x - nsolve;
    if (bi < nfac) factor_smooth_body<32>(m, d, 0, smem, Blk{bi * wf, wf, (int)blockDim.x});
    else publish_body<32>(d, 1, reinterpret_cast<int*>(smem), Blk{(bi - nfac) * wf, wf, (int)blockDim.x}, m.nexplicit ? m.pair_solreffriction : nullptr);
  }
}
template <int NV4>
static int launch_cgw_t(const MjhModel* m, const MjhData* d, bool with_factor, int fuse_euler, hipStream_t s, int nefc_lo, int nefc_hi) {
  const CgwLayout lay = cgw_layout<NV4>(d->njmax);
  const FacLayout fl = fac_layout(m->nv, m->nC);
  const size_t ms_bytes = sizeof(int) * mstruct_ints(m->nv, m->nC);
  size_t lds;
  int threads = pick_block(0, sizeof(float) * lay.total, 64, &lds, true);
  if (!threads) return fail(MJH_E_UNSUPPORTED, "k_solve_cgw: njmax x nv does not fit in LDS");
  if (const char* e = mjh_knob("MJH_SOLVE_THREADS")) {  // tuning knob (developer only)
    threads = std::max(atoi(e), 64);
    lds = sizeof(float) * lay.total * (threads / 64);
  }
  const int wpb = threads / 64, wf = threads / 32;
  if (with_factor) lds = std::max(lds, ms_bytes + sizeof(float) * fl.total * wf);
  HIPCHK(set_lds(k_solve_cgw_plus<NV4>, lds));
  const int nsolve = (d->nworld + wpb - 1) / wpb, nfac = with_factor ? (d->nworld + wf - 1) / wf : 0;
  debug_occupancy("k_solve_cgw_plus", k_solve_cgw_plus<NV4>, nsolve + 2 * nfac, threads, lds);
  hipLaunchKernelGGL(k_solve_cgw_plus<NV4>, dim3(nsolve + 2 * nfac), dim3(threads), lds, s, *m, *d, nsolve, nfac, nefc_lo, nefc_hi, fuse_euler);
  return MJH_OK;
}
int launch_solve_cgw(const MjhModel* m, const MjhData* d, bool with_factor, int fuse_euler, hipStream_t s, int lo, int hi) {
  switch ((m->nv + 3) / 4) {
    case 0:
    case 1: return launch_cgw_t<1>(m, d, with_factor, fuse_euler, s, lo, hi);
    case 2: return launch_cgw_t<2>(m, d, with_factor, fuse_euler, s, lo, hi);
    case 3: return launch_cgw_t<3>(m, d, with_factor, fuse_euler, s, lo, hi);
    case 4: return launch_cgw_t<4>(m, d, with_factor, fuse_euler, s, lo, hi);
    case 5: return launch_cgw_t<5>(m, d, with_factor, fuse_euler, s, lo, hi);
    case 6: return launch_cgw_t<6>(m, d, with_factor, fuse_euler, s, lo, hi);
    case 7: return launch_cgw_t<7>(m, d, with_factor, fuse_euler, s, lo, hi);
    default: return launch_cgw_t<8>(m, d, with_factor, fuse_euler, s, lo, hi);
  }
}
