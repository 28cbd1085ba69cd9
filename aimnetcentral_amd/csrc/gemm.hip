// gemm.hip - exact-fp32 MFMA GEMM for the AIMNet2 MLP stack on gfx950.
//
//   C[M,N] = A[M,K] . Bt[N,K]^T   (both operands K-contiguous: "NT"), fused epilogues.
//
// Replaces the torch addmm + GELU calls of aimnet/modules/core.py:11-46 (MLP builder) as used by
// aimnet/models/aimnet2.py:166 and the autograd replay of the same layers in the backward.
//
// Why fp32 MFMA: the reference's parity gate is |dE| < 1e-5 eV, |dF| < 1e-5 + 1e-4|F|
// (tests/test_calculator_gpu.py:445,464); bf16 inputs cannot hold it, and gfx950 has no
// xf32/TF32.  v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 multiply and accumulate in fp32 at the
// fp32 matrix peak (157 TFLOP/s, MI355X_MICROARCH.md) and leave the VALU free for the GELU epilogue.
//
// Two kernels, one data path (profiles/r1c_summary.md has the measurements behind every choice):
//   * operands are DMA'd global -> LDS with global_load_lds_dwordx4 (no VGPR staging, no ds_write) into a ring of
//     K = 16 stages; rows are unpadded and the 16-byte k-chunk index is XOR-swizzled on the SOURCE address (the DMA
//     writes lane-linear) so that the ds_read_b128 fragment reads are bank-conflict free;
//   * the fragment reads are inline asm: hipcc puts a conservative `s_waitcnt vmcnt(0)` in front of every C++ LDS
//     read that follows a global_load_lds, which would drain the ring on every K tile; with asm reads the only
//     vmcnt wait is ours ("tile kt has landed, the younger tiles may still be in flight");
//   * the panel kernel exists with 16-deep stages (ring of 4) and 32-deep stages (ring of 3: half the barriers, and the
//     second half's fragment reads overlap the first half's MFMAs); choose_tile() takes the deep form where its LDS
//     footprint does not cost a resident block;
//   * gemm_nt_panel_kernel (512 threads, 16x16x4 MFMA, block tiles from 48x128 to 160x192) carries the MLP layers of
//     large batches; gemm_nt_ring_kernel (256 threads, 32x32x2 MFMA, 64x64 tiles, scalar epilogue without alignment
//     requirements) serves small M and unaligned outputs.  choose_tile() picks per launch.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace aimnet {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK_DEFAULT = 32;  // every K handed to the GEMM is a multiple of 32 (weights are padded at upload)

__device__ __forceinline__ void glds16(const float* g, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// Ring kernel: 64x64x16 tiles, 2x2 waves of one 32x32 MFMA tile each, NST-deep LDS ring of 8 KiB stages
// (NST = 3 -> 24 KiB -> 6 blocks/CU, so the 1264 tiles of a 10 080 x 512 layer are all resident at once).
// LDS rows are 16 floats = 4 granules of 16 B; granule slot = k-chunk ^ ((row >> 2) & 3).
// The only vmcnt wait is vmcnt(2*(NST-2)): "tile kt has landed, the younger tiles may still be in flight".
template <int EPI, int NST>
__global__ __launch_bounds__(256, (NST == 2 ? 8 : 6)) void gemm_nt_ring_kernel(const float* __restrict__ A, int lda,
                                                                             const float* __restrict__ Bt, int ldb, int M,
                                                                             int N, int K, const float* __restrict__ bias,
                                                                             float* __restrict__ C, float* __restrict__ D,
                                                                             int ldc, const int* __restrict__ brow, int ldbias) {
  constexpr int BM = 64, BN = 64, BK = 16;
  constexpr int STAGE = (BM + BN) * BK;        // floats per stage: A tile then B tile, each [64 rows][4 granules of 16 B]
  constexpr int STAGE_BYTES = STAGE * 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  const int li = lane & 31, lh = lane >> 5;

  const int tiles_n = (N + BN - 1) / BN;
  const int nwg = gridDim.x;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = blockIdx.x & 7;
  const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);
  const int m0 = (wg / tiles_n) * BM, n0 = (wg % tiles_n) * BN;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;

  // DMA: wave w moves LDS granules G = w*64 + lane of each operand tile; granule G = row G>>2, k-chunk (G&3) ^ ((row>>2)&3)
  const int G = wid * 64 + lane;
  const int grow = G >> 2, gkc = (G & 3) ^ ((grow >> 2) & 3);
  const float* srcA = A + (size_t)min(m0 + grow, M - 1) * lda + gkc * 4;
  const float* srcB = Bt + (size_t)min(n0 + grow, N - 1) * ldb + gkc * 4;
  auto dma = [&](int stage, int k0) {
    float* base = smem + stage * STAGE + wid * 64 * 4;
    glds16(srcA + k0, base);
    glds16(srcB + k0, base + BM * BK);
  };
  // fragment byte addresses inside a stage: row r, k-chunk kc = 2*kk + lh -> (r*4 + (kc ^ ((r>>2)&3))) * 16
  const int ra = wr * 32 + li, rb = wc * 32 + li;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
  const unsigned aA0 = lds0 + (ra * 4 + ((0 + lh) ^ ((ra >> 2) & 3))) * 16;
  const unsigned aA1 = lds0 + (ra * 4 + ((2 + lh) ^ ((ra >> 2) & 3))) * 16;
  const unsigned aB0 = lds0 + BM * BK * 4 + (rb * 4 + ((0 + lh) ^ ((rb >> 2) & 3))) * 16;
  const unsigned aB1 = lds0 + BM * BK * 4 + (rb * 4 + ((2 + lh) ^ ((rb >> 2) & 3))) * 16;

  const int nk = K / BK;
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nk) dma(s, s * BK);
  int st = 0;          // stage of tile kt
  int sn = NST - 1;    // stage the next DMA goes to
  for (int kt = 0; kt < nk; ++kt) {
    if (NST > 2 && kt + NST - 2 < nk) __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * (NST - 2)));  // vmcnt(2*(NST-2))
    else __builtin_amdgcn_s_waitcnt(0x0F70);                                              // vmcnt(0)
    __builtin_amdgcn_s_barrier();  // tile kt visible to all waves; everybody is done reading tile kt-1 (stage sn)
    if (kt + NST - 1 < nk) dma(sn, (kt + NST - 1) * BK);
    f32x4 a0, a1, b0, b1;
    {
      const unsigned so = st * STAGE_BYTES;
      asm volatile(
          "ds_read_b128 %0, %4\n\tds_read_b128 %2, %6\n\tds_read_b128 %1, %5\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
          : "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(b1)
          : "v"(aA0 + so), "v"(aA1 + so), "v"(aB0 + so), "v"(aB1 + so)
          : "memory");
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[0], b0[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[1], b0[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[2], b0[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[3], b0[3], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[0], b1[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[1], b1[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[2], b1[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[3], b1[3], acc, 0, 0, 0);
    st = (st + 1 == NST) ? 0 : st + 1;
    sn = (sn + 1 == NST) ? 0 : sn + 1;
  }

  const int col = n0 + wc * 32 + li;
  if (col >= N) return;
  float bvv = 0.0f;
  if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) bvv = bias[col];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (row >= M) continue;
    const size_t o = (size_t)row * ldc + col;
    if (brow && (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU)) bvv = bias[(size_t)min(63, max(0, brow[row])) * ldbias + col];
    const float v = acc[r];
    if (EPI == EPI_NONE) {
      C[o] = v;
    } else if (EPI == EPI_BIAS) {
      C[o] = v + bvv;
    } else if (EPI == EPI_BIAS_GELU) {
      float h, d;
      gelu_and_grad(v + bvv, h, d);
      C[o] = h;
      if (D) D[o] = d;
    } else {
      C[o] = v * D[o];
    }
  }
}

template <int NST>
static int launch_ring(hipStream_t stream, int epi, const float* A, int lda, const float* Bt, int ldb, int M, int N, int K,
                       const float* bias, float* C, float* D, int ldc, const int* brow, int ldbias) {
  const int tiles = ceil_div(M, 64) * ceil_div(N, 64);
  const size_t lds = (size_t)NST * 128 * 16 * sizeof(float);
  dim3 grid(tiles), block(256);
  switch (epi) {
    case EPI_NONE:
      hipLaunchKernelGGL((gemm_nt_ring_kernel<EPI_NONE, NST>), grid, block, lds, stream, A, lda, Bt, ldb, M, N, K, bias, C, D, ldc, brow, ldbias);
      break;
    case EPI_BIAS:
      hipLaunchKernelGGL((gemm_nt_ring_kernel<EPI_BIAS, NST>), grid, block, lds, stream, A, lda, Bt, ldb, M, N, K, bias, C, D, ldc, brow, ldbias);
      break;
    case EPI_BIAS_GELU:
      hipLaunchKernelGGL((gemm_nt_ring_kernel<EPI_BIAS_GELU, NST>), grid, block, lds, stream, A, lda, Bt, ldb, M, N, K, bias, C, D, ldc, brow, ldbias);
      break;
    case EPI_MUL:
      hipLaunchKernelGGL((gemm_nt_ring_kernel<EPI_MUL, NST>), grid, block, lds, stream, A, lda, Bt, ldb, M, N, K, bias, C, D, ldc, brow, ldbias);
      break;
    default:
      set_last_error("gemm: bad epilogue %d", epi);
      return -1;
  }
  AIMNET_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Panel kernel: 512 threads = WM x WN waves, v_mfma_f32_16x16x4_f32, block tile
// TM x TN = (16*SM*WM) x (16*SN*WN) with SM*SN accumulator tiles per wave (160x128, 160x192, 128x96 ...).
// Why: a 64x64 tile streams (64+64)/(64*64) operand floats per MAC through the DMA and the LDS; measured
// (ablation runs, profiles/r1c_summary.md) that traffic - not L2 bandwidth, not the barrier - costs 30 % of the MFMA
// time.  A 160x128 tile moves 2.2x fewer bytes per MAC, and the strip counts are template parameters so the
// host can pick the shape whose tile count just fits a whole number of "rounds" of the 256 CUs
// (10 080 x 512 -> 63 x 4 = 252 tiles of 160x128: one tile per CU, 98 % balanced).
// Operands are swapped into the MFMA (weights = "A" operand, activations = "B"), which leaves each lane with
// 4 consecutive output COLUMNS of one row -> dwordx4 epilogue stores / bias / D loads.
// LDS: NST-deep ring of [TM rows | TN rows] x 16 floats, DMA'd like the ring kernel, slot = chunk ^ 3*((row>>3)&1):
// conflict-free for the 16x16x4 fragment reads, whose b128 lane groups are not contiguous - MI355X_MICROARCH.md LDS).
template <int OFF>
__device__ __forceinline__ f32x4 lds_read16(unsigned addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
// fragments f[I..N) of consecutive 16-row strips: strip i sits i * STRIP bytes further (16 rows x BK floats)
template <int I, int N, int STRIP>
__device__ __forceinline__ void read_frags(f32x4 (&f)[N], unsigned addr) {
  if constexpr (I < N) {
    f[I] = lds_read16<I * STRIP>(addr);
    read_frags<I + 1, N, STRIP>(f, addr);
  }
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_lgkmcnt() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
// granule slot of k-chunk c in row r of a stage with BK floats per row: an XOR swizzle that makes every lane group of a
// ds_read_b128 ({0-3, 12-15, 20-27}, ...) touch all 64 banks once.  64-byte rows: 4 rows per 256 B bank window;
// 128-byte rows: 2 rows per window, so the row pair index is folded in instead.
template <int BK>
__device__ __forceinline__ int swz(int row) {
  return BK == 16 ? 3 * ((row >> 3) & 1) : ((row >> 1) & 7);
}

// BK = 16: four stages of [TM|TN] x 16 floats.  BK = 32: three stages of twice the depth - one barrier per 32 k, and the
// LDS reads of the second 16-k half are in flight while the MFMAs of the first half run (the barrier re-aligns all eight
// waves every stage, so with BK = 16 the read latency of every stage is exposed on both waves of a SIMD at once).
template <int EPI, int SM, int SN, int WM, int WN, int BK = 16>
__global__ __launch_bounds__(512, ((SM * SN > 10 || (BK == 32 && 16 * (SM * WM + SN * WN) > 208)) ? 2 : 4)) void gemm_nt_panel_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ Bt, int ldb, int M, int N, int K,
    const float* __restrict__ bias, float* __restrict__ C, float* __restrict__ D, int ldc, const int* __restrict__ brow,
    int ldbias) {
  static_assert(WM * WN == 8, "8 waves");
  static_assert(BK == 16 || BK == 32, "BK");
  constexpr int NST = BK == 16 ? 4 : 3;
  constexpr int TM = 16 * SM * WM, TN = 16 * SN * WN;
  constexpr int GPR = BK / 4;                   // 16-byte granules per row
  constexpr int STAGE_BYTES = (TM + TN) * BK * 4;
  constexpr int NGRAN = (TM + TN) * GPR;        // 16-byte granules per stage
  constexpr int NPASS = (NGRAN + 511) / 512;    // DMA wave-instructions per wave per stage (last pass may be partial)
  constexpr int STRIP = 16 * BK * 4;            // bytes between consecutive 16-row strips
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;

  const int tiles_n = (N + TN - 1) / TN;
  const int nwg = gridDim.x;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = blockIdx.x & 7;
  const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);
  const int m0 = (wg / tiles_n) * TM, n0 = (wg % tiles_n) * TN;

  f32x4 acc[SM][SN];
#pragma unroll
  for (int i = 0; i < SM; ++i)
#pragma unroll
    for (int j = 0; j < SN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // DMA sources.  Granule G = p*512 + tid of a stage: rows [0,TM) are the activation tile, [TM,TM+TN) the weight tile;
  // TM*4 is a multiple of 64, so a wave-instruction never straddles the two.
  const float* src[NPASS];
#pragma unroll
  for (int p = 0; p < NPASS; ++p) {
    const int G = min(p * 512 + tid, NGRAN - 1);
    const int row = G / GPR;
    const int kc = (G % GPR) ^ swz<BK>(row);
    src[p] = (row < TM) ? A + (size_t)min(m0 + row, M - 1) * lda + kc * 4
                        : Bt + (size_t)min(n0 + row - TM, N - 1) * ldb + kc * 4;
  }
  // number of passes in which this wave has granules (wave-uniform): NPASS or NPASS-1
  const bool last_pass = ((NPASS - 1) * 512 + wid * 64) < NGRAN;
  auto dma = [&](int stage, int k0) {
    float* base = smem + stage * (STAGE_BYTES / 4) + wid * 64 * 4;
#pragma unroll
    for (int p = 0; p < NPASS - 1; ++p) glds16(src[p] + k0, base + p * 512 * 4);
    if (last_pass) glds16(src[NPASS - 1] + k0, base + (NPASS - 1) * 512 * 4);
  };

  // fragment byte addresses in stage 0: row r, k-chunk c = lane>>4 (+4 for the second half of a 32-deep stage)
  // -> (r*GPR + (c ^ swz(r))) * 16; strip i adds i*STRIP bytes
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
  const int l16 = lane & 15, lc = lane >> 4;
  const int rA = wm * 16 * SM + l16, rB = wn * 16 * SN + l16;
  const unsigned adA = lds0 + (rA * GPR + (lc ^ swz<BK>(rA))) * 16;
  const unsigned adB = lds0 + TM * BK * 4 + (rB * GPR + (lc ^ swz<BK>(rB))) * 16;
  const unsigned adA2 = lds0 + (rA * GPR + ((lc + 4) ^ swz<BK>(rA))) * 16;
  const unsigned adB2 = lds0 + TM * BK * 4 + (rB * GPR + ((lc + 4) ^ swz<BK>(rB))) * 16;

  const int nk = K / BK;
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nk) dma(s, s * BK);
  int st = 0, sn = NST - 1;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + NST - 2 < nk) {  // tiles kt+1 .. kt+NST-2 may stay in flight
      if (last_pass) wait_vmcnt<(NST - 2) * NPASS>();
      else wait_vmcnt<(NST - 2) * (NPASS - 1)>();
    } else {
      wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();  // tile kt visible to all waves; everybody has finished reading tile kt-1 (stage sn)
    if (kt + NST - 1 < nk) dma(sn, (kt + NST - 1) * BK);

    const unsigned so = st * STAGE_BYTES;
    f32x4 fa[SM], fb[SN];
    read_frags<0, SN, STRIP>(fb, adB + so);
    read_frags<0, SM, STRIP>(fa, adA + so);
    if constexpr (BK == 32) {
      f32x4 fa2[SM], fb2[SN];
      read_frags<0, SN, STRIP>(fb2, adB2 + so);
      read_frags<0, SM, STRIP>(fa2, adA2 + so);
      // LDS returns in order: the first half's fragments have landed when at most SM + SN reads are outstanding
      wait_lgkmcnt<(SM + SN < 15 ? SM + SN : 15)>();
#pragma unroll
      for (int i = 0; i < SM; ++i) asm volatile("" : "+v"(fa[i]));
#pragma unroll
      for (int j = 0; j < SN; ++j) asm volatile("" : "+v"(fb[j]));
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < SM; ++i)
#pragma unroll
          for (int j = 0; j < SN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[j][t], fa[i][t], acc[i][j], 0, 0, 0);
      wait_lgkmcnt<0>();
#pragma unroll
      for (int i = 0; i < SM; ++i) asm volatile("" : "+v"(fa2[i]));
#pragma unroll
      for (int j = 0; j < SN; ++j) asm volatile("" : "+v"(fb2[j]));
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < SM; ++i)
#pragma unroll
          for (int j = 0; j < SN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb2[j][t], fa2[i][t], acc[i][j], 0, 0, 0);
    } else {
      // one drain that "produces" every fragment (empty asm ties), so no MFMA can be scheduled above it
      wait_lgkmcnt<0>();
#pragma unroll
      for (int i = 0; i < SM; ++i) asm volatile("" : "+v"(fa[i]));
#pragma unroll
      for (int j = 0; j < SN; ++j) asm volatile("" : "+v"(fb[j]));
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < SM; ++i)
#pragma unroll
          for (int j = 0; j < SN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[j][t], fa[i][t], acc[i][j], 0, 0, 0);
    }
    st = (st + 1 == NST) ? 0 : st + 1;
    sn = (sn + 1 == NST) ? 0 : sn + 1;
  }

  // epilogue: acc[i][j][r] = C[m0 + wm*16*SM + 16 i + (lane&15)][n0 + wn*16*SN + 16 j + 4 (lane>>4) + r]
#pragma unroll
  for (int j = 0; j < SN; ++j) {
    const int col = n0 + wn * 16 * SN + 16 * j + 4 * lc;
    if (col >= N) continue;
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) bv = *reinterpret_cast<const f32x4*>(bias + col);
#pragma unroll
    for (int i = 0; i < SM; ++i) {
      const int row = m0 + wm * 16 * SM + 16 * i + l16;
      if (row >= M) continue;
      const size_t o = (size_t)row * ldc + col;
      // row-indexed bias table (brow: table row of every output row, e.g. the atom's element): pass 0 folds the constant
      // embedding block of its first layer into such a table
      if (brow && (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU))
        bv = *reinterpret_cast<const f32x4*>(bias + (size_t)min(63, max(0, brow[row])) * ldbias + col);
      f32x4 v = acc[i][j];
      if (EPI == EPI_NONE) {
        *reinterpret_cast<f32x4*>(C + o) = v;
      } else if (EPI == EPI_BIAS) {
        *reinterpret_cast<f32x4*>(C + o) = v + bv;
      } else if (EPI == EPI_BIAS_GELU) {
        f32x4 h, d;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float hh, dd;
          gelu_and_grad(v[r] + bv[r], hh, dd);
          h[r] = hh;
          d[r] = dd;
        }
        *reinterpret_cast<f32x4*>(C + o) = h;
        if (D) *reinterpret_cast<f32x4*>(D + o) = d;
      } else {
        const f32x4 dv = *reinterpret_cast<const f32x4*>(D + o);
        *reinterpret_cast<f32x4*>(C + o) = v * dv;
      }
    }
  }
}

template <int SM, int SN, int WM, int WN, int BK = 16>
static int launch_panel(hipStream_t stream, int epi, const float* A, int lda, const float* Bt, int ldb, int M, int N, int K,
                        const float* bias, float* C, float* D, int ldc, const int* brow, int ldbias) {
  constexpr int TM = 16 * SM * WM, TN = 16 * SN * WN;
  const int tiles = ceil_div(M, TM) * ceil_div(N, TN);
  const size_t lds = (size_t)(BK == 16 ? 4 : 3) * (TM + TN) * BK * 4;
  static PerDeviceOnce once;
  if (once.first()) {  // > 64 KiB of dynamic LDS needs the opt-in, once per instantiation and device
    AIMNET_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt_panel_kernel<EPI_NONE, SM, SN, WM, WN, BK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    AIMNET_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt_panel_kernel<EPI_BIAS, SM, SN, WM, WN, BK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    AIMNET_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt_panel_kernel<EPI_BIAS_GELU, SM, SN, WM, WN, BK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    AIMNET_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt_panel_kernel<EPI_MUL, SM, SN, WM, WN, BK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  }
  dim3 grid(tiles), block(512);
  switch (epi) {
    case EPI_NONE:
      hipLaunchKernelGGL((gemm_nt_panel_kernel<EPI_NONE, SM, SN, WM, WN, BK>), grid, block, lds, stream, A, lda, Bt, ldb, M, N, K, bias, C, D, ldc, brow, ldbias);
      break;
    case EPI_BIAS:
      hipLaunchKernelGGL((gemm_nt_panel_kernel<EPI_BIAS, SM, SN, WM, WN, BK>), grid, block, lds, stream, A, lda, Bt, ldb, M, N, K, bias, C, D, ldc, brow, ldbias);
      break;
    case EPI_BIAS_GELU:
      hipLaunchKernelGGL((gemm_nt_panel_kernel<EPI_BIAS_GELU, SM, SN, WM, WN, BK>), grid, block, lds, stream, A, lda, Bt, ldb, M, N, K, bias, C, D, ldc, brow, ldbias);
      break;
    case EPI_MUL:
      hipLaunchKernelGGL((gemm_nt_panel_kernel<EPI_MUL, SM, SN, WM, WN, BK>), grid, block, lds, stream, A, lda, Bt, ldb, M, N, K, bias, C, D, ldc, brow, ldbias);
      break;
    default:
      set_last_error("gemm: bad epilogue %d", epi);
      return -1;
  }
  AIMNET_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Skinny kernel for small M (single molecules: M ~ 100 rows).  There the MFMA work is ~1 us and the tiled kernels
// are pure latency: 46 dependent K steps of DMA -> barrier -> read.  Here every 16x16 output tile gets a block of
// SKINNY_KS waves that split K; a wave issues ALL of its operand loads (<= 3 + 3 dwordx4 straight from L2,
// no LDS) before its first MFMA, so the critical path is one memory round trip + ~12 MFMAs + a 16-way LDS
// reduction.  L2 traffic is M*N*K/2 bytes (each tile re-reads its panels), which is why this is only chosen for
// M <= 256.
constexpr int SKINNY_KS = 16;  // waves of a block = K slices of its tile.  4 -> 8 -> 16 slices: taxol 0.2890 -> 0.2849 -> 0.2745 ms (every wave's
                                // dependent chain of loads and MFMAs shrinks; the 16-way LDS reduction costs less than it saves)
template <int EPI>
__global__ __launch_bounds__(64 * SKINNY_KS) void gemm_nt_skinny_kernel(const float* __restrict__ A, int lda, const float* __restrict__ Bt,
                                                            int ldb, int M, int N, int K, const float* __restrict__ bias,
                                                            float* __restrict__ C, float* __restrict__ D, int ldc,
                                                            const int* __restrict__ brow, int ldbias) {
  constexpr int MAXC = 48 / SKINNY_KS;  // 16-wide k chunks in flight per wave (K <= 768 in one round)
  __shared__ f32x4 part[SKINNY_KS][64];
  const int lane = threadIdx.x & 63;
  const int ks = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tiles_n = (N + 15) >> 4;
  const int m0 = (blockIdx.x / tiles_n) * 16, n0 = (blockIdx.x % tiles_n) * 16;
  const int l16 = lane & 15, lc = lane >> 4;
  const int nch = K >> 4;                                   // chunks of 16 along K
  const int c_lo = (nch * ks) / SKINNY_KS, c_hi = (nch * (ks + 1)) / SKINNY_KS;  // this wave's chunk range
  const float* pa = A + (size_t)min(m0 + l16, M - 1) * lda + lc * 4;
  const float* pb = Bt + (size_t)min(n0 + l16, N - 1) * ldb + lc * 4;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int c0 = c_lo; c0 < c_hi; c0 += MAXC) {
    f32x4 fa[MAXC], fb[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int cc = min(c0 + c, c_hi - 1);  // clamped duplicates are masked below
      fa[c] = *reinterpret_cast<const f32x4*>(pa + cc * 16);
      fb[c] = *reinterpret_cast<const f32x4*>(pb + cc * 16);
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      if (c0 + c < c_hi) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[c][t], fa[c][t], acc, 0, 0, 0);
      }
    }
  }
  part[ks][lane] = acc;
  __syncthreads();
  if (ks != 0) return;
  f32x4 v = part[0][lane];
#pragma unroll
  for (int w = 1; w < SKINNY_KS; ++w) v = v + part[w][lane];
  // v[r] = C[m0 + (lane & 15)][n0 + 4 (lane >> 4) + r]
  const int row = m0 + l16, col = n0 + 4 * lc;
  if (row >= M) return;
  if (brow) bias += (size_t)min(63, max(0, brow[row])) * ldbias;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (col + r >= N) continue;
    const size_t o = (size_t)row * ldc + col + r;
    float x = v[r];
    if (EPI == EPI_NONE) {
      C[o] = x;
    } else if (EPI == EPI_BIAS) {
      C[o] = x + bias[col + r];
    } else if (EPI == EPI_BIAS_GELU) {
      float h, d;
      gelu_and_grad(x + bias[col + r], h, d);
      C[o] = h;
      if (D) D[o] = d;
    } else {
      C[o] = x * D[o];
    }
  }
}

static int launch_skinny(hipStream_t stream, int epi, const float* A, int lda, const float* Bt, int ldb, int M, int N, int K,
                         const float* bias, float* C, float* D, int ldc, const int* brow, int ldbias) {
  dim3 grid(ceil_div(M, 16) * ceil_div(N, 16)), block(64 * SKINNY_KS);
  switch (epi) {
    case EPI_NONE:
      hipLaunchKernelGGL(gemm_nt_skinny_kernel<EPI_NONE>, grid, block, 0, stream, A, lda, Bt, ldb, M, N, K, bias, C, D, ldc, brow, ldbias);
      break;
    case EPI_BIAS:
      hipLaunchKernelGGL(gemm_nt_skinny_kernel<EPI_BIAS>, grid, block, 0, stream, A, lda, Bt, ldb, M, N, K, bias, C, D, ldc, brow, ldbias);
      break;
    case EPI_BIAS_GELU:
      hipLaunchKernelGGL(gemm_nt_skinny_kernel<EPI_BIAS_GELU>, grid, block, 0, stream, A, lda, Bt, ldb, M, N, K, bias, C, D, ldc, brow, ldbias);
      break;
    case EPI_MUL:
      hipLaunchKernelGGL(gemm_nt_skinny_kernel<EPI_MUL>, grid, block, 0, stream, A, lda, Bt, ldb, M, N, K, bias, C, D, ldc, brow, ldbias);
      break;
    default:
      set_last_error("gemm: bad epilogue %d", epi);
      return -1;
  }
  AIMNET_LAUNCH_CHECK();
  return 0;
}

static int g_force_tile = 0;  // 0 = choose_tile(); a tile id from AIMNET_GEMM_TILE forces one configuration (A/B runs)
static int g_stage_k = 32;    // AIMNET_GEMM_BK=16 selects the 16-deep-stage form of the panel tiles (A/B runs)

// Tile choice.  Every candidate runs the same MFMA rate; what differs is (a) how evenly ceil(M/TM)*ceil(N/TN)
// tiles load the 256 CUs - the busiest CU carries ceil(tiles/256) tiles of TM*TN MACs per k, padding of M and N
// included - and (b) the operand bytes streamed per MAC, x = (TM+TN)/(TM*TN), which costs MFMA issue slots
// through the DMA + LDS-read path.  Fitted to tests/tools/tune_gemm.py on 10 080-row layers (profiles/r1c_summary.md):
// efficiency = 0.9 / (1 + 5x), times 0.88 when a CU holds a single block of <= 10 accumulator tiles per wave
// (nothing to overlap its barrier with).  The model reproduces the measured ranking on all six MLP shapes.
struct TileCand { int id, tm, tn, acc_tiles; };
static const TileCand kTileCands[] = {
    {152, 160, 128, 10}, {142, 128, 128, 8}, {132, 96, 128, 6}, {122, 64, 128, 4}, {153, 160, 192, 15}, {143, 128, 192, 12},
    {223, 128, 96, 6},   {213, 64, 96, 3},   {222, 128, 64, 4}, {351, 80, 128, 5}, {331, 48, 128, 3},   {5, 64, 64, 1},
    {381, 128, 128, 8},  {371, 112, 128, 7}, {361, 96, 128, 6}, {341, 64, 128, 4}, {321, 32, 128, 2},   {233, 192, 96, 9},
    {412, 192, 128, 12}, {411, 176, 128, 11}, {410, 160, 128, 10}, {409, 144, 128, 9}};
int device_cus() {  // compute units of the current device (256 on MI355X), queried once per device
  static int cus[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cus[dev] == 0) {
    int n = 0;
    cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
  }
  return cus[dev];
}

static int choose_tile(int M, int N, bool vec_ok) {
  if (M <= 256) return 7;  // latency regime: split-K skinny kernel (see gemm_nt_skinny_kernel)
  const long n_cu = device_cus();
  int best = 5, best_sum = 128;
  long best_per_cu = 1;
  double best_cost = 1e300;
  for (const TileCand& c : kTileCands) {
    if (c.id != 5 && !vec_ok) continue;
    const long tiles = (long)ceil_div(M, c.tm) * ceil_div(N, c.tn);
    const long per_cu = (tiles + n_cu - 1) / n_cu;
    const double x = (double)(c.tm + c.tn) / ((double)c.tm * c.tn);
    double eff = 0.9 / (1.0 + 5.0 * x);
    if (per_cu == 1 && c.acc_tiles <= 10) eff *= 0.88;
    if (c.id == 5) eff *= 0.9;  // 4-wave ring kernel: larger fixed cost per tile (profiles/r1c_summary.md)
    const double cost = (double)per_cu * c.tm * c.tn / eff;
    if (cost < best_cost) { best_cost = cost; best = c.id; best_sum = c.tm + c.tn; best_per_cu = per_cu; }
  }
  // 32-deep stages (id + 1000: one barrier and one exposed LDS round trip per 32 k instead of two) beat the 16-deep form
  // of the same tile by 0-5 % (tests/tools/tune_gemm_grid.py) - unless their 384 B x (TM + TN) of LDS evicts the second
  // resident block of a CU that has two or more tiles to run
  const bool deep = g_stage_k == 32 && best != 5 && (best_per_cu == 1 || best_sum <= 192);
  return deep ? 1000 + best : best;
}

int launch_gemm_nt_cfg(hipStream_t stream, int cfg, int epi, const float* A, int lda, const float* Bt, int ldb, int M, int N,
                       int K, const float* bias, float* C, float* D, int ldc, const int* brow, int ldbias) {
  if (M <= 0) return 0;
  if (K % BK_DEFAULT != 0 || (lda & 3) || (ldb & 3)) {
    set_last_error("gemm: K=%d must be a multiple of %d and lda/ldb multiples of 4", K, BK_DEFAULT);
    return -1;
  }
  if (cfg == 0) cfg = g_force_tile;
  if (cfg == 0) cfg = choose_tile(M, N, ((N | ldc) & 3) == 0 && (((size_t)bias | (size_t)C | (size_t)D) & 15) == 0);
  switch (cfg) {
#define AIMNET_PANEL_CASE(ID, SM_, SN_, WM_, WN_) \
    case ID: return launch_panel<SM_, SN_, WM_, WN_>(stream, epi, A, lda, Bt, ldb, M, N, K, bias, C, D, ldc, brow, ldbias);      \
    case 1000 + ID: return launch_panel<SM_, SN_, WM_, WN_, 32>(stream, epi, A, lda, Bt, ldb, M, N, K, bias, C, D, ldc, brow, ldbias);
    // panel kernels: id = 100*arrangement + 10*SM + SN; arrangement 1 = 2x4 waves, 2 = 4x2, 3 = 1x8
    AIMNET_PANEL_CASE(152, 5, 2, 2, 4)  // 160 x 128
    AIMNET_PANEL_CASE(142, 4, 2, 2, 4)  // 128 x 128
    AIMNET_PANEL_CASE(132, 3, 2, 2, 4)  //  96 x 128
    AIMNET_PANEL_CASE(122, 2, 2, 2, 4)  //  64 x 128
    AIMNET_PANEL_CASE(153, 5, 3, 2, 4)  // 160 x 192
    AIMNET_PANEL_CASE(143, 4, 3, 2, 4)  // 128 x 192
    AIMNET_PANEL_CASE(223, 2, 3, 4, 2)  // 128 x  96
    AIMNET_PANEL_CASE(213, 1, 3, 4, 2)  //  64 x  96
    AIMNET_PANEL_CASE(222, 2, 2, 4, 2)  // 128 x  64
    AIMNET_PANEL_CASE(412, 12, 1, 1, 8)  // 192 x 128 (1 x 8 waves; ids 409..412 = SM 9..12 of that family)
    AIMNET_PANEL_CASE(411, 11, 1, 1, 8)  // 176 x 128
    AIMNET_PANEL_CASE(410, 10, 1, 1, 8)  // 160 x 128
    AIMNET_PANEL_CASE(409, 9, 1, 1, 8)   // 144 x 128
    AIMNET_PANEL_CASE(381, 8, 1, 1, 8)  // 128 x 128 (1 x 8 waves)
    AIMNET_PANEL_CASE(371, 7, 1, 1, 8)  // 112 x 128
    AIMNET_PANEL_CASE(361, 6, 1, 1, 8)  //  96 x 128
    AIMNET_PANEL_CASE(351, 5, 1, 1, 8)  //  80 x 128
    AIMNET_PANEL_CASE(341, 4, 1, 1, 8)  //  64 x 128
    AIMNET_PANEL_CASE(331, 3, 1, 1, 8)  //  48 x 128
    AIMNET_PANEL_CASE(321, 2, 1, 1, 8)  //  32 x 128
    AIMNET_PANEL_CASE(233, 3, 3, 4, 2)  // 192 x  96
#undef AIMNET_PANEL_CASE
    case 5: return launch_ring<3>(stream, epi, A, lda, Bt, ldb, M, N, K, bias, C, D, ldc, brow, ldbias);
    case 7: return launch_skinny(stream, epi, A, lda, Bt, ldb, M, N, K, bias, C, D, ldc, brow, ldbias);
    default:
      set_last_error("gemm: unknown tile id %d", cfg);
      return -1;
  }
}

int launch_gemm_nt(hipStream_t stream, int epi, const float* A, int lda, const float* Bt, int ldb, int M, int N,
                   int K, const float* bias, float* C, float* D, int ldc, const int* brow, int ldbias) {
  return launch_gemm_nt_cfg(stream, 0, epi, A, lda, Bt, ldb, M, N, K, bias, C, D, ldc, brow, ldbias);
}

int gemm_set_attributes() {
  const char* env = getenv("AIMNET_GEMM_TILE");
  g_force_tile = env ? atoi(env) : 0;
  env = getenv("AIMNET_GEMM_BK");
  if (env) g_stage_k = atoi(env) == 16 ? 16 : 32;
  return 0;  // the panel kernels opt in to > 64 KiB of dynamic LDS at their first launch (launch_panel)
}

}  // namespace aimnet
