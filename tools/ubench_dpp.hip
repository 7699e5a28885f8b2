// ubench_dpp.hip -- what v_permlane16_swap, v_permlane32_swap and the DPP operand row_newbcast do on gfx950 (solver_cgw.hpp relies on it)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out) {
  const int l = threadIdx.x;
  const float x = (float)l;
  const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  out[l] = __uint_as_float(s16[0]);
  out[64 + l] = __uint_as_float(s16[1]);
  out[128 + l] = __uint_as_float(s32[0]);
  out[192 + l] = __uint_as_float(s32[1]);
  out[256 + l] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x150 + 5, 0xf, 0xf, true));
}
int main() {
  float* d;
  hipMalloc(&d, 320 * 4);
  k<<<1, 64>>>(d);
  float h[320];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[5] = {"permlane16_swap[0]", "permlane16_swap[1]", "permlane32_swap[0]", "permlane32_swap[1]", "row_newbcast:5"};
  for (int a = 0; a < 5; ++a) {
    printf("%-20s", names[a]);
    for (int l = 0; l < 64; ++l) printf(" %2.0f", h[64 * a + l]);
    printf("\n");
  }
  return 0;
}
