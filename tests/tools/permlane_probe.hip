// probe of v_permlane16_swap / v_permlane32_swap (gfx950) as cross-row reduction steps:
//   hipcc --offload-arch=gfx950 -O3 tests/tools/permlane_probe.hip -o gpurun_in/permlane_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k_builtin(float* o) {
  float z = o[threadIdx.x];
  u2 r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, z), __builtin_bit_cast(unsigned, z), false, false);
  z = __builtin_bit_cast(float, r.x) + __builtin_bit_cast(float, r.y);
  u2 q = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, z), __builtin_bit_cast(unsigned, z), false, false);
  z = __builtin_bit_cast(float, q.x) + __builtin_bit_cast(float, q.y);
  o[threadIdx.x] = z;
}
__global__ void k_asm(float* o) {
  float a = o[threadIdx.x], b = a;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  a += b;
  b = a;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  a += b;
  o[threadIdx.x] = a;
}
__global__ void k_four(float* o) {  // four values reduced across the rows at once: value v ends in row v
  float z0 = o[threadIdx.x], z1 = z0 + 100.f, z2 = z0 + 200.f, z3 = z0 + 300.f;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(z0), "+v"(z1));
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(z2), "+v"(z3));
  float s1 = z0 + z1, s2 = z2 + z3;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(s1), "+v"(s2));
  o[threadIdx.x] = s1 + s2;
}
int main() {
  float h[64], *d;
  hipMalloc(&d, 256);
  for (int v = 0; v < 3; ++v) {
    for (int i = 0; i < 64; ++i) h[i] = (float)i;
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    if (v == 0) k_builtin<<<1, 64>>>(d); else if (v == 1) k_asm<<<1, 64>>>(d); else k_four<<<1, 64>>>(d);
    hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    printf("%s:", v == 0 ? "builtin" : v == 1 ? "asm" : "four");
    for (int i = 0; i < 64; i += (v == 2 ? 5 : 7)) printf(" [%d]=%g", i, h[i]);
    printf("\n");
  }
  // expected builtin/asm: lane l -> l%16 * 4 + 96;  four: row r (lanes 16r..16r+15): 4 (l%16) + 96 + 400 r
  return 0;
}
