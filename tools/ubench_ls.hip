// ubench_ls.hip -- cycles of one line_search_rows<2, 32> call per wavefront as a function of the bracketing iterations it runs.
#include "solver_newton.hpp"

#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(512) k(float* io, long long* ticks, int reps, int ls_iterations, float gtol) {
  const int lig = threadIdx.x & 31;
  float rja[2], rjv[2], rD[2];
  int rkind[2];
  for (int k = 0; k < 2; ++k) {
    rja[k] = io[threadIdx.x * 8 + k];
    rjv[k] = io[threadIdx.x * 8 + 2 + k];
    rD[k] = io[threadIdx.x * 8 + 4 + k];
    rkind[k] = (lig + 32 * k) < 50 ? 2 : 3;
  }
  float acc = 0.0f;
  int its = 0;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < reps; ++it) {
    float alpha, imp;
    bool ok;
    line_search_rows<2, 32, false>(rja, rjv, rD, rkind, io, (-3.0f + acc * 1e-20f) / 32.0f, 2.0f / 32.0f, 0.0f, gtol, ls_iterations, alpha, imp, ok, &its);
    acc += alpha + imp;
  }
  const long long t1 = __builtin_readcyclecounter();
  io[4096 + threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0) {
    ticks[threadIdx.x >> 6] = t1 - t0;
    ticks[8 + (threadIdx.x >> 6)] = its;
  }
}

int main() {
  std::vector<float> h(4096 + 512);
  for (int t = 0; t < 512; ++t) {
    const int lig = t & 31;
    for (int k = 0; k < 2; ++k) {
      h[t * 8 + k] = 0.01f * (float)((lig * 7 + k * 13) % 11) - 0.04f;      // Jaref: mixed signs
      h[t * 8 + 2 + k] = 0.02f * (float)((lig * 5 + k * 3) % 9) - 0.08f;    // jv
      h[t * 8 + 4 + k] = 200.0f + 10.0f * lig;                              // D
    }
  }
  float* io;
  long long* ticks;
  hipMalloc(&io, sizeof(float) * h.size());
  hipMalloc(&ticks, sizeof(long long) * 16);
  hipMemcpy(io, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice);
  const int reps = 200;
  for (int w : {1, 2})
    for (int lsit : {0, 1, 2, 4, 8}) {
      for (int pass = 0; pass < 2; ++pass) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256 * w), 0, 0, io, ticks, reps, lsit, 1e-9f);  // tiny gtol: runs all allowed iterations
        hipDeviceSynchronize();
      }
      long long t[16];
      hipMemcpy(t, ticks, sizeof(t), hipMemcpyDeviceToHost);
      printf("waves/SIMD %d, ls_iterations cap %d: %.0f cycles per call (%.1f bracketing iterations per call)\n", w, lsit, (double)t[0] / reps, (double)t[8] / reps);
    }
  return 0;
}
