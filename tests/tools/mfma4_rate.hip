// Issue rate / dependent latency of v_mfma_f32_4x4x1_16B_f32 (and 16x16x4 for comparison) on gfx950.
// One wave per SIMD (256-thread block per CU would be 4 waves: one per SIMD); cycles from s_memtime-free clock64().
//   hipcc --offload-arch=gfx950 -O3 tests/tools/mfma4_rate.hip -o /tmp/mfma4_rate && /tmp/mfma4_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NACC, int KIND>
__global__ __launch_bounds__(64) void k(int iters, float* out, unsigned long long* cyc) {
  f4 acc[NACC];
  for (int j = 0; j < NACC; ++j) acc[j] = f4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = 1e-4f;
  unsigned long long w0 = wall_clock64();
  unsigned long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int j = 0; j < NACC; ++j)
        acc[j] = KIND == 0 ? __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[j], 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
  }
  unsigned long long c1 = clock64();
  unsigned long long w1 = wall_clock64();
  float r = 0;
  for (int j = 0; j < NACC; ++j) r += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[threadIdx.x] = r;
  if (threadIdx.x == 0) { cyc[0] = c1 - c0; cyc[1] = w1 - w0; }
}
template <int NACC, int KIND>
void run(const char* name) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 64 * 4); hipMalloc(&cyc, 16);
  const int iters = 2000;
  k<NACC, KIND><<<1, 64>>>(iters, out, cyc); hipDeviceSynchronize();
  k<NACC, KIND><<<1, 64>>>(iters, out, cyc); hipDeviceSynchronize();
  unsigned long long c[2]; hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
  printf("%s  %d independent accumulators: %.2f clock64 ticks, %.2f ns per MFMA (wall_clock64 at 100 MHz)\n", name, NACC, (double)c[0] / (iters * 8.0 * NACC), (double)c[1] * 10.0 / (iters * 8.0 * NACC));
  hipFree(out); hipFree(cyc);
}
int main() {
  run<1, 0>("4x4x1_16B "); run<2, 0>("4x4x1_16B "); run<4, 0>("4x4x1_16B "); run<8, 0>("4x4x1_16B ");
  run<1, 1>("16x16x4   "); run<2, 1>("16x16x4   "); run<4, 1>("16x16x4   ");
  // clock64 on gfx950 counts at a fixed 100 MHz: report the conversion too
  return 0;
}
