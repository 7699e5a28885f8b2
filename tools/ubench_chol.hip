// ubench_chol.hip -- cycles of one chol_factor_solve<7> (blocked Cholesky + forward / backward substitution, solver_newton.hpp) per
// wavefront, alone on a SIMD and with 2 / 3 wavefronts per SIMD; also the old chol_factor_rows + chol_solve_rows (solver.hpp).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I mujoco_warp_amd/csrc -o ubench_chol tools/ubench_chol.hip
#include "solver_newton.hpp"

#include <cstdio>
#include <cstdlib>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(768) k(float* io, long long* ticks, int reps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NV4 = 7, NVR = 28;
  const int lig = threadIdx.x & 31, half = threadIdx.x >> 5;
  float* S = smem + half * 320;
  float h[NVR], h0[NVR];
#pragma unroll
  for (int c = 0; c < NVR; ++c) h0[c] = io[(threadIdx.x & 63) * NVR + c];
  float acc = 0.0f;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < reps; ++it) {
#pragma unroll
    for (int c = 0; c < NVR; ++c) h[c] = h0[c] + acc * 1e-20f;
    if (MODE == 0) {
      acc += chol_factor_solve<NV4>(h, 1.0f + lig, S, S + 128, S + 160, lig);
    } else {
      float lt[NVR], rd;
      chol_factor_rows<NVR, 28, 32>(h, lt, rd, S, lig);
      acc += chol_solve_rows<NVR, 32>(h, lt, rd, 1.0f + lig);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  io[64 * NVR + threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0) ticks[threadIdx.x >> 6] = t1 - t0;
}

int main() {
  const int NVR = 28, reps = 200;
  std::vector<float> H(64 * NVR);
  for (int l = 0; l < 64; ++l)
    for (int c = 0; c < NVR; ++c) {
      const int i = l & 31;
      H[l * NVR + c] = (i == c ? 30.0f : 0.0f) + 1.0f / (1.0f + (float)abs(i - c));  // SPD, diagonally dominant
    }
  float* io;
  long long* ticks;
  hipMalloc(&io, sizeof(float) * (64 * NVR + 1024));
  hipMalloc(&ticks, sizeof(long long) * 16);
  hipMemcpy(io, H.data(), sizeof(float) * 64 * NVR, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode)
    for (int w : {1, 2, 3}) {
      const int threads = 256 * w;
      const size_t lds = sizeof(float) * 320 * (threads / 32);
      for (int pass = 0; pass < 2; ++pass) {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(threads), lds, 0, io, ticks, reps);
        else hipLaunchKernelGGL(k<1>, dim3(1), dim3(threads), lds, 0, io, ticks, reps);
        hipDeviceSynchronize();
      }
      long long h[16];
      hipMemcpy(h, ticks, sizeof(long long) * (threads / 64), hipMemcpyDeviceToHost);
      long long mx = 0;
      for (int i = 0; i < threads / 64; ++i) mx = h[i] > mx ? h[i] : mx;
      printf("%s  waves/SIMD %d: %.0f cycles per factor+solve per wavefront (%.0f per SIMD)\n", mode == 0 ? "blocked fused (new)" : "rows + readlane (old)", w,
             (double)mx / reps, (double)mx / reps / w);
    }
  return 0;
}
