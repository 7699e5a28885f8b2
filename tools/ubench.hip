// ubench.hip -- gfx950 micro-measurements behind the solver design: VALU / packed-FMA / f32-MFMA issue cost
// per wave64 instruction at 1..4 waves per SIMD, v_readlane and LDS dependent-chain latency, and the register layout of
// v_mfma_f32_32x32x1_2b_f32 + v_permlane32_swap (checked against a host outer product).
//   hipcc --offload-arch=gfx950 -O3 -o ubench tools/ubench.hip && ./ubench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32v __attribute__((ext_vector_type(32)));
typedef float f2v __attribute__((ext_vector_type(2)));

#define CHK(x)                                                                  \
  do {                                                                          \
    hipError_t e = (x);                                                         \
    if (e != hipSuccess) {                                                      \
      printf("%s: %s\n", #x, hipGetErrorString(e));                             \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

__device__ inline long long now() { return __builtin_readcyclecounter(); }

// mode 0: v_fma_f32 (8 independent chains), 1: v_pk_fma_f32, 2: mfma 32x32x1_2b (1 acc), 3: mfma with 2 accs,
// 9: ONE dependent v_fma chain, 10: v_cmp -> v_cndmask dependent chain, 11: v_min / v_max dependent chain, 12: v_rcp chain,
// 4: readlane dependent chain, 5: LDS dependent chain (ds_read_b32), 6: dpp add chain, 7: ds_swizzle chain, 8: mfma 16x16x4
template <int MODE>
__global__ void k_issue(float* out, long long* ticks, int iters) {
  __shared__ int chase[1024];
  const int l = threadIdx.x;
  float a0 = l * 0.001f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float c = 1.0001f, dd = 0.5f;
  f2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = p0 + 1.0f, p5 = p1 + 1.0f, p6 = p2 + 1.0f, p7 = p3 + 1.0f;
  const f2v pc = {c, c}, pd = {dd, dd};
  f32v acc = {0}, acc2 = {0};
  typedef float f4v __attribute__((ext_vector_type(4)));
  f4v q0 = {0}, q1 = {0};
  for (int i = l; i < 1024; i += blockDim.x) chase[i] = (i * 17 + 5) & 1023;
  __syncthreads();
  int idx = l & 1023;
  const long long t0 = now();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        a0 = fmaf(a0, c, dd); a1 = fmaf(a1, c, dd); a2 = fmaf(a2, c, dd); a3 = fmaf(a3, c, dd);
        a4 = fmaf(a4, c, dd); a5 = fmaf(a5, c, dd); a6 = fmaf(a6, c, dd); a7 = fmaf(a7, c, dd);
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        p0 = __builtin_elementwise_fma(p0, pc, pd); p1 = __builtin_elementwise_fma(p1, pc, pd);
        p2 = __builtin_elementwise_fma(p2, pc, pd); p3 = __builtin_elementwise_fma(p3, pc, pd);
        p4 = __builtin_elementwise_fma(p4, pc, pd); p5 = __builtin_elementwise_fma(p5, pc, pd);
        p6 = __builtin_elementwise_fma(p6, pc, pd); p7 = __builtin_elementwise_fma(p7, pc, pd);
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x1f32(a0, a1, acc, 0, 0, 0);
    } else if (MODE == 3) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc = __builtin_amdgcn_mfma_f32_32x32x1f32(a0, a1, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x1f32(a2, a3, acc2, 0, 0, 0);
      }
    } else if (MODE == 4) {
#pragma unroll
      for (int u = 0; u < 16; ++u) a0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a0 * c), (u * 5) & 63)) + a1;
    } else if (MODE == 5) {
#pragma unroll
      for (int u = 0; u < 16; ++u) idx = chase[idx];
    } else if (MODE == 6) {
#pragma unroll
      for (int u = 0; u < 16; ++u) a0 = a0 + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a0), 0x111, 0xf, 0xf, true));
    } else if (MODE == 7) {
#pragma unroll
      for (int u = 0; u < 16; ++u) a0 = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(a0 * c), 0x3E0));
    } else if (MODE == 9) {
#pragma unroll
      for (int u = 0; u < 16; ++u) a0 = fmaf(a0, c, dd);
    } else if (MODE == 10) {
#pragma unroll
      for (int u = 0; u < 16; ++u) a0 = (a0 < a1) ? a0 + c : a0 * dd;
    } else if (MODE == 11) {
#pragma unroll
      for (int u = 0; u < 16; ++u) a0 = fminf(fmaxf(a0, a1), a2) + c;
    } else if (MODE == 12) {
#pragma unroll
      for (int u = 0; u < 16; ++u) a0 = __builtin_amdgcn_rcpf(a0) + c;
    } else if (MODE == 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        q0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, a1, q0, 0, 0, 0);
        q1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, a3, q1, 0, 0, 0);
      }
    }
  }
  const long long t1 = now();
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p5.y + p6.x + p7.y + (float)idx + q0.x + q1.y;
  for (int j = 0; j < 32; ++j) s += acc[j] + acc2[j];
  out[blockIdx.x * blockDim.x + l] = s;
  if ((l & 63) == 0) ticks[blockIdx.x * (blockDim.x / 64) + l / 64] = t1 - t0;
}

// layout dump: acc = a (x) b per block, then the permlane32_swap regrouping used by the solver
__global__ void k_layout(const float* a, const float* b, float* acc_out, float* row_out) {
  const int l = threadIdx.x;
  f32v acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x1f32(a[l], b[l], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x1f32(a[64 + l], b[64 + l], acc, 0, 0, 0);
  for (int j = 0; j < 32; ++j) acc_out[l * 32 + j] = acc[j];
  // after the swap lane (blk, c) holds all 32 rows of column c of its block: register j (< 16) of the first result holds
  // row 8*(j>>2) + (j&3), of the second result row 8*(j>>2) + 4 + (j&3)
  for (int j = 0; j < 16; ++j) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[j]), __float_as_uint(acc[16 + j]), false, false);
    row_out[l * 32 + 8 * (j >> 2) + (j & 3)] = __uint_as_float(r[0]);
    row_out[l * 32 + 8 * (j >> 2) + 4 + (j & 3)] = __uint_as_float(r[1]);
  }
}

template <int MODE>
static void run_issue(const char* name, int instr_per_iter, int waves_per_simd) {
  const int iters = 200, threads = 256 * waves_per_simd;  // one workgroup on one CU: waves_per_simd waves on each SIMD
  float* out;
  long long* ticks;
  CHK(hipMalloc(&out, sizeof(float) * threads));
  CHK(hipMalloc(&ticks, sizeof(long long) * 64));
  if (threads > 1024) {
    printf("%-34s skipped (block too large)\n", name);
    return;
  }
  hipLaunchKernelGGL(k_issue<MODE>, dim3(1), dim3(threads), 0, 0, out, ticks, iters);
  CHK(hipDeviceSynchronize());
  hipLaunchKernelGGL(k_issue<MODE>, dim3(1), dim3(threads), 0, 0, out, ticks, iters);
  CHK(hipDeviceSynchronize());
  std::vector<long long> h(threads / 64);
  CHK(hipMemcpy(h.data(), ticks, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
  long long mx = 0;
  for (auto v : h) mx = v > mx ? v : mx;
  // readcyclecounter ticks at a constant 100 MHz on gfx9 (s_memrealtime) or the shader clock (s_memtime): report raw ticks
  printf("%-34s waves/SIMD %d: %8lld ticks, %.2f ticks per wave-instr, %.2f per instr per SIMD\n", name, waves_per_simd, mx,
         (double)mx / (iters * instr_per_iter), (double)mx / (iters * instr_per_iter * waves_per_simd));
  CHK(hipFree(out));
  CHK(hipFree(ticks));
}

int main() {
  hipDeviceProp_t p;
  CHK(hipGetDeviceProperties(&p, 0));
  printf("%s, %d CUs, clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  for (int w : {1, 2, 4}) {
    run_issue<0>("v_fma_f32 x8 chains", 128, w);
    run_issue<1>("v_pk_fma_f32 x8 chains", 128, w);
    run_issue<2>("mfma_32x32x1_2b dependent", 8, w);
    run_issue<3>("mfma_32x32x1_2b 2 accs", 8, w);
    run_issue<8>("mfma_16x16x4 2 accs", 16, w);
    run_issue<4>("mul+readlane+add chain", 16, w);
    run_issue<5>("ds_read_b32 chase", 16, w);
    run_issue<6>("dpp row_shr add chain", 16, w);
    run_issue<7>("mul+ds_swizzle chain", 16, w);
    run_issue<9>("ONE dependent v_fma chain", 16, w);
    run_issue<10>("cmp -> 2 ops -> cndmask chain (4 instr/step)", 16, w);
    run_issue<11>("max, min, add chain (3 instr/step)", 16, w);
    run_issue<12>("rcp + add chain (2 instr/step)", 16, w);
  }
  // ---- layout check ----
  std::vector<float> a(128), b(128), acc(64 * 32), row(64 * 32);
  for (int i = 0; i < 128; ++i) {
    a[i] = 1.0f + 0.37f * i + 0.01f * (i % 7);
    b[i] = 2.0f - 0.11f * i + 0.003f * (i % 5) * i;
  }
  float *da, *db, *dacc, *drow;
  CHK(hipMalloc(&da, 512));
  CHK(hipMalloc(&db, 512));
  CHK(hipMalloc(&dacc, 64 * 32 * 4));
  CHK(hipMalloc(&drow, 64 * 32 * 4));
  CHK(hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice));
  CHK(hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, da, db, dacc, drow);
  CHK(hipDeviceSynchronize());
  CHK(hipMemcpy(acc.data(), dacc, 64 * 32 * 4, hipMemcpyDeviceToHost));
  CHK(hipMemcpy(row.data(), drow, 64 * 32 * 4, hipMemcpyDeviceToHost));
  auto C = [&](int blk, int i, int j) { return a[blk * 32 + i] * b[blk * 32 + j] + a[64 + blk * 32 + i] * b[64 + blk * 32 + j]; };
  int bad1 = 0, bad2 = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 32; ++r) {
      const int blk = r >> 4, rr = r & 15, col = l & 31, rw = (rr & 3) + 8 * (rr >> 2) + 4 * (l >> 5);
      const float e = C(blk, rw, col);
      if (fabsf(acc[l * 32 + r] - e) > 1e-3f * fabsf(e)) ++bad1;
    }
  for (int l = 0; l < 64; ++l)
    for (int i = 0; i < 32; ++i) {
      const float e = C(l >> 5, i, l & 31);  // lane (blk, c) holds column c: element i = C[i][c]
      if (fabsf(row[l * 32 + i] - e) > 1e-3f * fabsf(e)) ++bad2;
    }
  printf("layout: acc[reg r, lane l] = C_blk(r>>4)[ (r&3) + 8*((r&15)>>2) + 4*(l>>5) ][ l&31 ]: %s (%d mismatches)\n", bad1 ? "WRONG" : "ok", bad1);
  printf("layout: after permlane32_swap lane (blk, c) holds column c of C_blk: %s (%d mismatches)\n", bad2 ? "WRONG" : "ok", bad2);
  return 0;
}
