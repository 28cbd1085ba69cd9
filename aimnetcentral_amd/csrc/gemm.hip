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
// Tiling: BM x BN x 32 block tile (64x64 or 128x128, see the kernel), 256 threads = 2x2 waves of
// 32x32 MFMA tiles.  Both tiles are staged global -> VGPR -> LDS (padded rows, 36
// floats, so the ds_read_b128 fragment reads are conflict free) with a 2-deep LDS ring: the
// global loads of tile k+1 are in flight under the 64 MFMAs of tile k, one barrier per K tile.
// K mapping inside an 8-wide k block: lane half h = lane>>5 owns k = 4h..4h+3, so one b128 read
// feeds four consecutive MFMAs for A and for B alike (the MFMA only needs A and B to agree on
// which k a lane half carries).
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace aimnet {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK_DEFAULT = 32;

// BM x BN block tile, 256 threads = 2x2 waves, each wave (BM/2) x (BN/2) = MI x NI MFMA tiles of 32x32.
//   64 x 64  : 1 MFMA tile per wave, 36 KiB LDS -> 4 blocks/CU.  Fine-grained: with M ~ 10^4 rows a layer
//              has ~10^3 tiles, so the 256 CUs stay evenly loaded (a 128x128 grid of 316 tiles ran as
//              "2 rounds" at 62 % utilisation, profiles/r1a).
//   128 x 128: 2x2 MFMA tiles per wave, 72 KiB LDS -> 2 blocks/CU.  Half the L2->LDS traffic per FLOP;
//              chosen when there are enough tiles to fill the chip many times over.
template <int EPI, int BM, int BN, int ABL = 0, int BK = BK_DEFAULT>
__global__ __launch_bounds__(256, (BM * BN * BK >= 128 * 128 * 32) ? 2 : (BK <= 16 ? 6 : 4)) void gemm_nt_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ Bt, int ldb, int M, int N, int K,
    const float* __restrict__ bias, float* __restrict__ C, float* __restrict__ D, int ldc) {
  constexpr int LDS_LD = BK + 4;
  constexpr int MI = BM / 64, NI = BN / 64;   // MFMA tiles per wave along M / N
  constexpr int RPT = 256 / (BK / 4);         // rows staged per pass of the 256 threads
  constexpr int LA = BM / RPT, LB = BN / RPT; // float4 staging loads per thread per K tile
  constexpr int KB = BK / 8;                  // 8-wide k blocks per K tile
  static_assert(LA >= 1 && LA <= 4 && LB >= 1 && LB <= 4, "staging uses up to 4 named registers per operand");
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

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // staging: thread t moves float4 #t, #t+256, ... of each [rows][32] tile; rows past M / N are clamped
  // (loaded, never stored) so the loads issue branch-free.  The staging registers are NAMED scalars
  // (ra0..ra3 / rb0..rb3, guarded by if constexpr): as arrays hipcc kept them in scratch memory
  // (scratch_store right behind every global_load, -45 % throughput).
  const int srow = tid / (BK / 4), sc4 = (tid % (BK / 4)) << 2;
  const float* const ga0 = A + (size_t)min(m0 + srow, M - 1) * lda + sc4;
  const float* const ga1 = A + (size_t)min(m0 + srow + 1 * RPT, M - 1) * lda + sc4;
  const float* const ga2 = A + (size_t)min(m0 + srow + 2 * RPT, M - 1) * lda + sc4;
  const float* const ga3 = A + (size_t)min(m0 + srow + 3 * RPT, M - 1) * lda + sc4;
  const float* const gb0 = Bt + (size_t)min(n0 + srow, N - 1) * ldb + sc4;
  const float* const gb1 = Bt + (size_t)min(n0 + srow + 1 * RPT, N - 1) * ldb + sc4;
  const float* const gb2 = Bt + (size_t)min(n0 + srow + 2 * RPT, N - 1) * ldb + sc4;
  const float* const gb3 = Bt + (size_t)min(n0 + srow + 3 * RPT, N - 1) * ldb + sc4;
  float* const sa = As + srow * LDS_LD + sc4;
  float* const sb = Bs + srow * LDS_LD + sc4;
  float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
  ra0 = ra1 = ra2 = ra3 = rb0 = rb1 = rb2 = rb3 = make_float4(0.f, 0.f, 0.f, 0.f);

#define AIMNET_GLOAD(k0)                                                   \
  do {                                                                     \
    ra0 = *reinterpret_cast<const float4*>(ga0 + (k0));                    \
    if constexpr (LA > 1) ra1 = *reinterpret_cast<const float4*>(ga1 + (k0)); \
    if constexpr (LA > 2) {                                                \
      ra2 = *reinterpret_cast<const float4*>(ga2 + (k0));                  \
      ra3 = *reinterpret_cast<const float4*>(ga3 + (k0));                  \
    }                                                                      \
    rb0 = *reinterpret_cast<const float4*>(gb0 + (k0));                    \
    if constexpr (LB > 1) rb1 = *reinterpret_cast<const float4*>(gb1 + (k0)); \
    if constexpr (LB > 2) {                                                \
      rb2 = *reinterpret_cast<const float4*>(gb2 + (k0));                  \
      rb3 = *reinterpret_cast<const float4*>(gb3 + (k0));                  \
    }                                                                      \
  } while (0)
#define AIMNET_LSTORE(buf)                                                                  \
  do {                                                                                      \
    *reinterpret_cast<float4*>(sa + ((buf)*BM) * LDS_LD) = ra0;                             \
    if constexpr (LA > 1) *reinterpret_cast<float4*>(sa + ((buf)*BM + RPT) * LDS_LD) = ra1; \
    if constexpr (LA > 2) {                                                                 \
      *reinterpret_cast<float4*>(sa + ((buf)*BM + 2 * RPT) * LDS_LD) = ra2;                 \
      *reinterpret_cast<float4*>(sa + ((buf)*BM + 3 * RPT) * LDS_LD) = ra3;                 \
    }                                                                                       \
    *reinterpret_cast<float4*>(sb + ((buf)*BN) * LDS_LD) = rb0;                             \
    if constexpr (LB > 1) *reinterpret_cast<float4*>(sb + ((buf)*BN + RPT) * LDS_LD) = rb1; \
    if constexpr (LB > 2) {                                                                 \
      *reinterpret_cast<float4*>(sb + ((buf)*BN + 2 * RPT) * LDS_LD) = rb2;                 \
      *reinterpret_cast<float4*>(sb + ((buf)*BN + 3 * RPT) * LDS_LD) = rb3;                 \
    }                                                                                       \
  } while (0)

  AIMNET_GLOAD(0);
  AIMNET_LSTORE(0);
  __syncthreads();

  const int nk = K / BK;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const bool more = kt + 1 < nk;
    if (more && ABL < 1) AIMNET_GLOAD((kt + 1) * BK);
    const float* a_base = As + (buf * BM + wr * (BM / 2) + li) * LDS_LD + 4 * lh;
    const float* b_base = Bs + (buf * BN + wc * (BN / 2) + li) * LDS_LD + 4 * lh;
#pragma unroll
    for (int kk = 0; kk < KB; ++kk) {
      float4 af[MI], bf[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const float4*>(a_base + i * 32 * LDS_LD + kk * 8);
#pragma unroll
      for (int j = 0; j < NI; ++j) bf[j] = *reinterpret_cast<const float4*>(b_base + j * 32 * LDS_LD + kk * 8);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
        }
    }
    if (more && ABL < 1) AIMNET_LSTORE(buf ^ 1);
    if (ABL < 2) __syncthreads();
  }
#undef AIMNET_GLOAD
#undef AIMNET_LSTORE

  // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int col = n0 + wc * (BN / 2) + ni * 32 + li;
      if (col >= N) continue;
      float bvv = 0.0f;
      if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) bvv = bias[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * (BM / 2) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row >= M) continue;
        const size_t o = (size_t)row * ldc + col;
        float v = acc[mi][ni][r];
        if (EPI == EPI_NONE) {
          C[o] = v;
        } else if (EPI == EPI_BIAS) {
          C[o] = v + bvv;
        } else if (EPI == EPI_BIAS_GELU) {
          float h, d;
          gelu_and_grad(v + bvv, h, d);
          C[o] = h;
          if (D) D[o] = d;
        } else {  // EPI_MUL: chain rule through the previous layer's GELU, D holds GELU'(z)
          C[o] = v * D[o];
        }
      }
    }
  }
}

template <int BM, int BN>
static int launch_cfg(hipStream_t stream, int epi, const float* A, int lda, const float* Bt, int ldb, int M, int N, int K,
                      const float* bias, float* C, float* D, int ldc) {
  const int tiles = ceil_div(M, BM) * ceil_div(N, BN);
  static const size_t pad = getenv("AIMNET_GEMM_LDS_PAD") ? (size_t)atoi(getenv("AIMNET_GEMM_LDS_PAD")) : 0;  // occupancy probe
  const size_t lds = (size_t)2 * (BM + BN) * (BK_DEFAULT + 4) * sizeof(float) + pad;
  dim3 grid(tiles), block(256);
  switch (epi) {
    case EPI_NONE:
      hipLaunchKernelGGL((gemm_nt_kernel<EPI_NONE, BM, BN>), grid, block, lds, stream, A, lda, Bt, ldb, M, N, K, bias, C, D, ldc);
      break;
    case EPI_BIAS:
      hipLaunchKernelGGL((gemm_nt_kernel<EPI_BIAS, BM, BN>), grid, block, lds, stream, A, lda, Bt, ldb, M, N, K, bias, C, D, ldc);
      break;
    case EPI_BIAS_GELU:
      hipLaunchKernelGGL((gemm_nt_kernel<EPI_BIAS_GELU, BM, BN>), grid, block, lds, stream, A, lda, Bt, ldb, M, N, K, bias, C, D, ldc);
      break;
    case EPI_MUL:
      hipLaunchKernelGGL((gemm_nt_kernel<EPI_MUL, BM, BN>), grid, block, lds, stream, A, lda, Bt, ldb, M, N, K, bias, C, D, ldc);
      break;
    default:
      set_last_error("gemm: bad epilogue %d", epi);
      return -1;
  }
  AIMNET_LAUNCH_CHECK();
  return 0;
}

// timing-only ablations of the 64x64 kernel (wrong results): 1 = no global->LDS restaging, 2 = also no barrier
int launch_gemm_ablation(hipStream_t stream, int abl, const float* A, int lda, const float* Bt, int ldb, int M, int N, int K,
                         float* C, int ldc) {
  const int tiles = ceil_div(M, 64) * ceil_div(N, 64);
  const size_t lds = (size_t)2 * 128 * (BK_DEFAULT + 4) * sizeof(float);
  if (abl == 3) {  // 64x64 tile with BK = 64 (half the barriers, 2 blocks/CU)
    const size_t l64 = (size_t)2 * 128 * (64 + 4) * sizeof(float);
    static bool once = false;
    if (!once) {
      AIMNET_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt_kernel<EPI_NONE, 64, 64, 0, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      once = true;
    }
    hipLaunchKernelGGL((gemm_nt_kernel<EPI_NONE, 64, 64, 0, 64>), dim3(tiles), dim3(256), l64, stream, A, lda, Bt, ldb, M, N, K, nullptr, C, nullptr, ldc);
    AIMNET_LAUNCH_CHECK();
    return 0;
  }
  if (abl == 4) {  // 64x64 tile with BK = 16: 20 KiB LDS -> up to 8 blocks/CU, every tile resident at once
    const size_t l16 = (size_t)2 * 128 * (16 + 4) * sizeof(float);
    hipLaunchKernelGGL((gemm_nt_kernel<EPI_NONE, 64, 64, 0, 16>), dim3(tiles), dim3(256), l16, stream, A, lda, Bt, ldb, M, N, K, nullptr, C, nullptr, ldc);
    AIMNET_LAUNCH_CHECK();
    return 0;
  }
  if (abl == 1)
    hipLaunchKernelGGL((gemm_nt_kernel<EPI_NONE, 64, 64, 1>), dim3(tiles), dim3(256), lds, stream, A, lda, Bt, ldb, M, N, K, nullptr, C, nullptr, ldc);
  else
    hipLaunchKernelGGL((gemm_nt_kernel<EPI_NONE, 64, 64, 2>), dim3(tiles), dim3(256), lds, stream, A, lda, Bt, ldb, M, N, K, nullptr, C, nullptr, ldc);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

static int g_force_tile = 0;  // 0 auto; 64, 128, 12864 (128x64), 64128 (64x128) from AIMNET_GEMM_TILE for A/B runs

int launch_gemm_nt_cfg(hipStream_t stream, int cfg, int epi, const float* A, int lda, const float* Bt, int ldb, int M, int N,
                       int K, const float* bias, float* C, float* D, int ldc) {
  if (M <= 0) return 0;
  if (K % BK_DEFAULT != 0 || (lda & 3) || (ldb & 3)) {
    set_last_error("gemm: K=%d must be a multiple of %d and lda/ldb multiples of 4", K, BK_DEFAULT);
    return -1;
  }
  if (cfg == 0) cfg = g_force_tile;
  if (cfg == 0) {
    // 128x128 tiles only when they alone would fill the 256 CUs (2 blocks each) ~8 times over
    const long big_tiles = (long)ceil_div(M, 128) * ceil_div(N, 128);
    cfg = big_tiles >= 4096 ? 128 : 64;
  }
  switch (cfg) {
    case 128: return launch_cfg<128, 128>(stream, epi, A, lda, Bt, ldb, M, N, K, bias, C, D, ldc);
    case 12864: return launch_cfg<128, 64>(stream, epi, A, lda, Bt, ldb, M, N, K, bias, C, D, ldc);
    case 64128: return launch_cfg<64, 128>(stream, epi, A, lda, Bt, ldb, M, N, K, bias, C, D, ldc);
    default: return launch_cfg<64, 64>(stream, epi, A, lda, Bt, ldb, M, N, K, bias, C, D, ldc);
  }
}

int launch_gemm_nt(hipStream_t stream, int epi, const float* A, int lda, const float* Bt, int ldb, int M, int N,
                   int K, const float* bias, float* C, float* D, int ldc) {
  return launch_gemm_nt_cfg(stream, 0, epi, A, lda, Bt, ldb, M, N, K, bias, C, D, ldc);
}

template <int BM, int BN>
static int set_attr() {
  const int lds = 160 * 1024;
  AIMNET_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt_kernel<EPI_NONE, BM, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  AIMNET_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt_kernel<EPI_BIAS, BM, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  AIMNET_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt_kernel<EPI_BIAS_GELU, BM, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  AIMNET_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt_kernel<EPI_MUL, BM, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  return 0;
}

int gemm_set_attributes() {
  const char* env = getenv("AIMNET_GEMM_TILE");
  g_force_tile = env ? atoi(env) : 0;
  int rc;
  if ((rc = set_attr<64, 64>())) return rc;
  if ((rc = set_attr<128, 64>())) return rc;
  if ((rc = set_attr<64, 128>())) return rc;
  if ((rc = set_attr<128, 128>())) return rc;  // 72 KiB of dynamic LDS exceeds the 64 KiB default cap
  return 0;
}

}  // namespace aimnet
