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

// erf for the GELU epilogues: branch-free, 1e-7 absolute (<= 1.5 ulp) over the whole line, ~25 VALU instructions
// against ~55 with two divergent paths for the device-library erff - the GELU epilogue of a 10 080 x 512 layer was
// 6-7 us of pure VALU on top of a 60 us GEMM.  |x| < 0.921875: x * P5(x^2);  otherwise sign(x) (1 - 2^Q8(|x|)) with
// Q8 fitted to log2(erfc) on [0.92, 4] (erf(4) = 1 - 1.5e-8 rounds to 1 in fp32).  Least-squares fits on Chebyshev
// nodes, verified against scipy.special.erf on 10^5 fp32 arguments emulating fp32 FMA (tests/test_gpu_ops.py
// checks the compiled result through the GEMM epilogue).
__device__ __forceinline__ float erf_fast(float x) {
  const float a = fminf(fabsf(x), 4.0f);
  const float s = x * x;
  float p = -5.986208510e-04f;
  p = fmaf(p, s, 4.992181038e-03f);
  p = fmaf(p, s, -2.676586123e-02f);
  p = fmaf(p, s, 1.128179275e-01f);
  p = fmaf(p, s, -3.761249112e-01f);
  p = fmaf(p, s, 1.128379149e+00f);
  float q = 2.327214117e-06f;
  q = fmaf(q, a, -6.574532227e-05f);
  q = fmaf(q, a, 8.549435628e-04f);
  q = fmaf(q, a, -6.837534396e-03f);
  q = fmaf(q, a, 3.803287519e-02f);
  q = fmaf(q, a, -1.586590115e-01f);
  q = fmaf(q, a, -9.116990641e-01f);
  q = fmaf(q, a, -1.630481967e+00f);
  q = fmaf(q, a, 4.364755757e-04f);
  const float big = copysignf(1.0f - __builtin_amdgcn_exp2f(q), x);
  return a < 0.921875f ? p * x : big;
}

// exact-erf GELU and its derivative (torch.nn.GELU default, aimnet/modules/core.py:27)
__device__ __forceinline__ float gelu_f(float z) { return 0.5f * z * (1.0f + erf_fast(z * 0.70710678118654752f)); }
__device__ __forceinline__ void gelu_and_grad(float z, float& h, float& d) {
  const float cdf = 0.5f * (1.0f + erf_fast(z * 0.70710678118654752f));
  const float pdf = 0.39894228040143268f * __builtin_amdgcn_exp2f(-0.72134752044448170f * z * z);  // exp(-z^2/2)
  h = z * cdf;
  d = cdf + z * pdf;
}

// hipFuncSetAttribute is per DEVICE: one process may drive several engines (one per GPU), so "done once" flags for the
// > 64 KiB dynamic-LDS opt-in are kept per device ordinal.  Returns true the first time it is called on the current device.
struct PerDeviceOnce {
  bool done[64] = {};
  bool first() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;  // unknown device: just set the attribute again
    if (done[dev]) return false;
    done[dev] = true;
    return true;
  }
};

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
