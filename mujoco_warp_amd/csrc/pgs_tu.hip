// pgs_tu.hip -- k_solve_pgs instantiations (one translation unit of libmjhip.so, see host.hpp)
#include "host.hpp"

#include "pgs.hpp"
#include "pgs_big.hpp"

// PGS (pgs.hpp): one launch, no riders (publish + factor ride with the integrator launch, as for Newton)
// REG: the register-resident sweep for njmax <= 64 (see pgs.hpp)
template <int NV4, int SG, bool REG>
__global__ void __launch_bounds__(256) k_solve_pgs(MjhModel m, MjhData d, int refresh) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wpb = blockDim.x / SG;
  pgs_body<NV4, SG, REG>(m, d, smem, Blk{(int)blockIdx.x * wpb, wpb, (int)blockDim.x}, refresh);
}
template <int NV4, int SG, bool REG>
static int launch_pgs_r(const MjhModel* m, const MjhData* d, hipStream_t s) {
  const PgsLayout lay = pgs_layout<NV4, SG>(d->njmax);
  size_t lds;
  const int threads = pick_block(0, sizeof(float) * lay.total, SG, &lds, true);
  if (!threads) return fail(MJH_E_UNSUPPORTED, "k_solve_pgs: njmax x nv does not fit in LDS");
  HIPCHK(set_lds((k_solve_pgs<NV4, SG, REG>), lds));
  const int wpb = threads / SG;
  static const int refresh = mjh_knob("MJH_PGS_REFRESH") ? atoi(mjh_knob("MJH_PGS_REFRESH")) : 8;  // developer knob (REG sweep): residual rebuild period
  hipLaunchKernelGGL((k_solve_pgs<NV4, SG, REG>), dim3((d->nworld + wpb - 1) / wpb), dim3(threads), lds, s, *m, *d, refresh);
  return MJH_OK;
}
template <int NV4, int SG>
static int launch_pgs_t(const MjhModel* m, const MjhData* d, hipStream_t s) {
  static const bool no_reg = mjh_knob("MJH_PGS_NOREG") != nullptr;  // developer knob: force the general (LDS) sweep
  if (d->njmax <= 64 && !no_reg) return launch_pgs_r<NV4, 64, true>(m, d, s);  // one world per wavefront
  return launch_pgs_r<NV4, SG, false>(m, d, s);
}
// generic PGS (pgs_big.hpp): more than 64 dofs, or elliptic friction cones -- one world per workgroup of up to 8 wavefronts (the sweep runs
// over the world's constraint islands in parallel; 128 VGPRs so that two workgroups share a CU)
__global__ void __launch_bounds__(64 * PGSB_MAXWAVES) __attribute__((amdgpu_waves_per_eu(4, 8))) k_solve_pgs_big(MjhModel m, MjhData d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  pgs_big_body<64>(m, d, smem, (int)blockIdx.x);  // (grid = nworld; block-wide barriers inside: no early exit here)
}
static int launch_pgs_big(const MjhModel* m, const MjhData* d, hipStream_t s) {
  if (!d->ws_pgsB) return fail(MJH_E_ARG, "Data.ws_pgsB missing (allocate Data with make_data / put_data for this model)");
  const PgsBigLayout lay = pgs_big_layout(m->nv, m->nC, d->njmax, m->cone == CONE_ELLIPTIC && d->nmaxpyramid > 1);
  const size_t lds = sizeof(int) * mstruct_ints(m->nv, m->nC) + sizeof(float) * lay.total;
  if (lds > (size_t)kLdsPerCU) return fail(MJH_E_UNSUPPORTED, "k_solve_pgs_big: nv / njmax do not fit in LDS");
  HIPCHK(set_lds(k_solve_pgs_big, lds));
  // wavefronts per world: islands only exist between kinematic trees (a single tree is one island: one wavefront)
  static const int waves_knob = mjh_knob("MJH_PGSB_WAVES") ? atoi(mjh_knob("MJH_PGSB_WAVES")) : PGSB_MAXWAVES;  // developer knob
  const int waves = (m->ntree > 1 && m->ntree <= 64) ? std::max(1, std::min(std::min(waves_knob, PGSB_MAXWAVES), m->ntree)) : 1;
  hipLaunchKernelGGL(k_solve_pgs_big, dim3(d->nworld), dim3(64 * waves), lds, s, *m, *d);
  return MJH_OK;
}
int launch_pgs(const MjhModel* m, const MjhData* d, hipStream_t s) {
  if (m->nv > 64 || (m->cone == CONE_ELLIPTIC && d->nmaxpyramid > 1)) return launch_pgs_big(m, d, s);
  const int nv4 = (m->nv + 3) / 4;
  if (m->nv <= 32) {
    switch (nv4) {
      case 0:
      case 1: return launch_pgs_t<1, 32>(m, d, s);
      case 2: return launch_pgs_t<2, 32>(m, d, s);
      case 3: return launch_pgs_t<3, 32>(m, d, s);
      case 4: return launch_pgs_t<4, 32>(m, d, s);
      case 5: return launch_pgs_t<5, 32>(m, d, s);
      case 6: return launch_pgs_t<6, 32>(m, d, s);
      case 7: return launch_pgs_t<7, 32>(m, d, s);
      default: return launch_pgs_t<8, 32>(m, d, s);
    }
  }
  if (nv4 <= 9) return launch_pgs_t<9, 64>(m, d, s);
  if (nv4 <= 10) return launch_pgs_t<10, 64>(m, d, s);
  if (nv4 <= 12) return launch_pgs_t<12, 64>(m, d, s);
  if (nv4 <= 14) return launch_pgs_t<14, 64>(m, d, s);
  return launch_pgs_t<16, 64>(m, d, s);
}
