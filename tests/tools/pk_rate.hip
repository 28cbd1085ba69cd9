// pk_rate.hip - issue rate of v_fma_f32 against v_pk_fma_f32 on gfx950 (GPU box):
//   hipcc -O3 --offload-arch=gfx950 tests/tools/pk_rate.hip -o /tmp/pk_rate && /tmp/pk_rate
// Every wave runs ITER rounds of 16 independent accumulators (no dependency stalls at 4+ waves per SIMD); the grid fills every SIMD
// with W waves.  Prints cycles per instruction and wave (4 = one wave64 instruction per 4 clocks on a 16-lane SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITER = 4096;

template <bool PK>
__global__ __launch_bounds__(256) void rate_kernel(float* out, float s) {
  f2 a[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) a[k] = f2{(float)threadIdx.x + k, 1.0f + k};
  f2 m = {s, s * 0.5f}, c = {1e-3f, 2e-3f};
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (PK) {
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
      } else {
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k].x) : "v"(m.x), "v"(c.x));
      }
    }
  }
  float r = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) r += a[k].x + a[k].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <bool PK>
double run(int blocks, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(rate_kernel<PK>, dim3(blocks), dim3(256), 0, 0, out, 0.999f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(rate_kernel<PK>, dim3(blocks), dim3(256), 0, 0, out, 0.999f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const double ghz = p.clockRate * 1e-6;
  float* out;
  hipMalloc(&out, sizeof(float) * 256 * cus * 8);
  for (int wps = 1; wps <= 8; wps *= 2) {  // waves per SIMD (a 256-thread block = one wave per SIMD of a CU)
    const int blocks = cus * wps;
    const double ms0 = run<false>(blocks, out), ms1 = run<true>(blocks, out);
    const double inst_per_simd = (double)ITER * 16 * wps;
    printf("waves/SIMD %d: v_fma_f32 %.3f ms = %.2f clk per wave-instruction;  v_pk_fma_f32 %.3f ms = %.2f clk per wave-instruction (%.1f TFLOP/s)\n",
           wps, ms0, ms0 * 1e-3 * ghz * 1e9 / inst_per_simd, ms1, ms1 * 1e-3 * ghz * 1e9 / inst_per_simd,
           inst_per_simd * cus * 4 * 64 * 4 / (ms1 * 1e-3) / 1e12);
  }
  printf("device: %s, %d CUs, %.2f GHz (clockRate)\n", p.name, cus, ghz);
  return 0;
}
