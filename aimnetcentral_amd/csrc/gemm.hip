// gemm.hip - exact-fp32 MFMA GEMM for the AIMNet2 MLP stack on gfx950.
//
//   C[M,N] = A[M,K] . Bt[N,K]^T   (both operands K-contiguous: "NT"), fused epilogues.
//
// Replaces the torch addmm + GELU calls of aimnet/modules/core.py:11-46 (MLP builder) as used by
// aimnet/models/aimnet2.py:166 and the autograd replay of the same layers in the backward.
//
// Why fp32 MFMA: the reference's parity gate is |dE| < 1e-5 eV, |dF| < 1e-5 + 1e-4|F|
// (tests/test_calculator_gpu.py:445,464); bf16 inputs cannot hold it, and gfx950 has no
// xf32/TF32.  v_mfma_f32_32x32x2_f32 is bit-for-bit an fmaf chain at the fp32 vector peak
// (157 TFLOP/s, MI355X_MICROARCH.md) and leaves the VALU free for the GELU epilogue.
//
// Tiling: 128x128x32 block tile, 256 threads = 2x2 waves, each wave 64x64 = 2x2 MFMA tiles of
// 32x32 (64 accumulator VGPRs).  Both tiles are staged global -> VGPR -> LDS (padded rows, 36
// floats, so the ds_read_b128 fragment reads are conflict free) with a 2-deep LDS ring: the
// global loads of tile k+1 are in flight under the 64 MFMAs of tile k, one barrier per K tile.
// K mapping inside an 8-wide k block: lane half h = lane>>5 owns k = 4h..4h+3, so one b128 read
// feeds four consecutive MFMAs for A and for B alike (the MFMA only needs A and B to agree on
// which k a lane half carries).
#include "common.h"
#include "kernels.h"

namespace aimnet {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32, LDS_LD = BK + 4;

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const float* __restrict__ A, int lda,
                                                        const float* __restrict__ Bt, int ldb, int M, int N, int K,
                                                        const float* __restrict__ bias, float* __restrict__ C,
                                                        float* __restrict__ D, int ldc) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                         // [2][BM][LDS_LD]
  float* Bs = smem + 2 * BM * LDS_LD;       // [2][BN][LDS_LD]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int li = lane & 31, lh = lane >> 5;

  // XCD-aware tile order.  Hardware places block b on XCD b % 8 (observed, speed only); remap so
  // each XCD owns a contiguous run of logical tiles, and order tiles column-fastest: the tiles
  // that share one A row panel then run on one XCD and hit its private L2 (bijective remap,
  // cdna_hip_programming.md T1).
  const int tiles_n = (N + BN - 1) / BN;
  const int nwg = gridDim.x;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = blockIdx.x & 7;
  const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);
  const int tile_m = wg / tiles_n;
  const int tile_n = wg % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  float4 ra[4], rb[4];
  auto gload = [&](int k0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int idx = tid + 256 * r;
      const int row = idx >> 3, c4 = (idx & 7) << 2;
      const int gm = m0 + row, gn = n0 + row;
      ra[r] = (gm < M) ? *reinterpret_cast<const float4*>(A + (size_t)gm * lda + k0 + c4) : make_float4(0, 0, 0, 0);
      rb[r] = (gn < N) ? *reinterpret_cast<const float4*>(Bt + (size_t)gn * ldb + k0 + c4) : make_float4(0, 0, 0, 0);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int idx = tid + 256 * r;
      const int row = idx >> 3, c4 = (idx & 7) << 2;
      *reinterpret_cast<float4*>(As + (buf * BM + row) * LDS_LD + c4) = ra[r];
      *reinterpret_cast<float4*>(Bs + (buf * BN + row) * LDS_LD + c4) = rb[r];
    }
  };

  const int nk = K / BK;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * BK);
    const float* a_base = As + (buf * BM + wr * 64 + li) * LDS_LD + 4 * lh;
    const float* b_base = Bs + (buf * BN + wc * 64 + li) * LDS_LD + 4 * lh;
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      const float4 a0 = *reinterpret_cast<const float4*>(a_base + kk * 8);
      const float4 a1 = *reinterpret_cast<const float4*>(a_base + 32 * LDS_LD + kk * 8);
      const float4 b0 = *reinterpret_cast<const float4*>(b_base + kk * 8);
      const float4 b1 = *reinterpret_cast<const float4*>(b_base + 32 * LDS_LD + kk * 8);
      const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
      const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[t], bv0[t], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[t], bv1[t], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[t], bv0[t], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[t], bv1[t], acc[1][1], 0, 0, 0);
      }
    }
    if (kt + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }

  // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = n0 + wc * 64 + ni * 32 + li;
      if (col >= N) continue;
      float bv = 0.0f;
      if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) bv = bias[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row >= M) continue;
        const size_t o = (size_t)row * ldc + col;
        float v = acc[mi][ni][r];
        if (EPI == EPI_NONE) {
          C[o] = v;
        } else if (EPI == EPI_BIAS) {
          C[o] = v + bv;
        } else if (EPI == EPI_BIAS_GELU) {
          float h, d;
          gelu_and_grad(v + bv, h, d);
          C[o] = h;
          if (D) D[o] = d;
        } else {  // EPI_MUL: chain rule through the previous layer's GELU, D holds GELU'(z)
          C[o] = v * D[o];
        }
      }
    }
  }
}

int launch_gemm_nt(hipStream_t stream, int epi, const float* A, int lda, const float* Bt, int ldb, int M, int N,
                   int K, const float* bias, float* C, float* D, int ldc) {
  if (M <= 0) return 0;
  if (K % BK != 0 || (lda & 3) || (ldb & 3)) {
    set_last_error("gemm: K=%d must be a multiple of %d and lda/ldb multiples of 4", K, BK);
    return -1;
  }
  const int tiles = ceil_div(M, BM) * ceil_div(N, BN);
  const size_t lds = (size_t)2 * (BM + BN) * LDS_LD * sizeof(float);
  dim3 grid(tiles), block(256);
  switch (epi) {
    case EPI_NONE:
      hipLaunchKernelGGL(gemm_nt_kernel<EPI_NONE>, grid, block, lds, stream, A, lda, Bt, ldb, M, N, K, bias, C, D, ldc);
      break;
    case EPI_BIAS:
      hipLaunchKernelGGL(gemm_nt_kernel<EPI_BIAS>, grid, block, lds, stream, A, lda, Bt, ldb, M, N, K, bias, C, D, ldc);
      break;
    case EPI_BIAS_GELU:
      hipLaunchKernelGGL(gemm_nt_kernel<EPI_BIAS_GELU>, grid, block, lds, stream, A, lda, Bt, ldb, M, N, K, bias, C, D, ldc);
      break;
    case EPI_MUL:
      hipLaunchKernelGGL(gemm_nt_kernel<EPI_MUL>, grid, block, lds, stream, A, lda, Bt, ldb, M, N, K, bias, C, D, ldc);
      break;
    default:
      set_last_error("gemm: bad epilogue %d", epi);
      return -1;
  }
  AIMNET_LAUNCH_CHECK();
  return 0;
}

int gemm_set_attributes() {
  // 72 KiB of dynamic LDS per block exceeds the 64 KiB default cap.
  const int lds = 2 * (BM + BN) * LDS_LD * (int)sizeof(float);
  AIMNET_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt_kernel<EPI_NONE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  AIMNET_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt_kernel<EPI_BIAS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  AIMNET_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt_kernel<EPI_BIAS_GELU>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  AIMNET_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt_kernel<EPI_MUL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  return 0;
}

}  // namespace aimnet
