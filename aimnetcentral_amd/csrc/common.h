// common.h - shared device helpers for the gfx950 AIMNet2 kernels (wave64 everywhere).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define AIMNET_WAVE 64

namespace aimnet {

// ---- error plumbing -------------------------------------------------------------------------
void set_last_error(const char* fmt, ...);

#define AIMNET_HIP_CHECK(expr)                                                            \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      ::aimnet::set_last_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return -2;                                                                          \
    }                                                                                     \
  } while (0)

#define AIMNET_LAUNCH_CHECK()                                                             \
  do {                                                                                    \
    hipError_t _e = hipGetLastError();                                                    \
    if (_e != hipSuccess) {                                                               \
      ::aimnet::set_last_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
      return -2;                                                                          \
    }                                                                                     \
  } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- wave-level reductions (DPP/shuffle; all 64 lanes participate) ---------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
  return v;
}

// exact-erf GELU and its derivative (torch.nn.GELU default, aimnet/modules/core.py:27)
__device__ __forceinline__ float gelu_f(float z) { return 0.5f * z * (1.0f + erff(z * 0.70710678118654752f)); }
__device__ __forceinline__ void gelu_and_grad(float z, float& h, float& d) {
  const float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752f));
  const float pdf = 0.39894228040143268f * expf(-0.5f * z * z);
  h = z * cdf;
  d = cdf + z * pdf;
}

// ---- packed integer lattice shifts: 3 x int8 in one int32 ------------------------------------
__device__ __forceinline__ int pack_shift(int sx, int sy, int sz) {
  return (sx & 0xff) | ((sy & 0xff) << 8) | ((sz & 0xff) << 16);
}
__device__ __forceinline__ void unpack_shift(int code, int& sx, int& sy, int& sz) {
  sx = (int)(int8_t)(code & 0xff);
  sy = (int)(int8_t)((code >> 8) & 0xff);
  sz = (int)(int8_t)((code >> 16) & 0xff);
}

}  // namespace aimnet
