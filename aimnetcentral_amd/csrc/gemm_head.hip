// gemm_head.hip - the energy head, forward AND backward, as ONE launch.
//
//   e_i = w3 . GELU(W2 GELU(W1 aim_i + b1) + b2) + b3        (Output MLP 256 -> 128 -> 128 -> 1, aimnet/modules/core.py:114-132)
//   zbar_i = (dE/d aim_i) * GELU'(z_last,i)                   (E = sum_i e_i, so the adjoint seed of e_i is 1: the backward needs
//                                                              nothing that is not in this block - autograd's replay of the head)
// The four GEMMs of the head (two forward, two backward) have N = 128: one column tile, a quarter of the chip per launch, 9 - 13 us
// each for 2 % of the step's arithmetic.  Here a block owns 64 atoms and chains them on the bf16x3-split matrix path of
// gemm_bf3a.hip (same operand split, same six products per tile, same two interleaved accumulator sets):
//   1  Z1 = aim W1^T   (K = 256; aim arrives pre-split from the last MLP layer's epilogue, streamed through an LDS ring)
//      H1 = GELU(Z1 + b1) -> LDS in split form, D1 = GELU' stays in registers
//   2  Z2 = H1 W2^T    (K = 128, A operand resident in LDS); H2, D2; e_i = H2 . w3 + b3 (lane, wave, block reduction)
//   3  T = (w3 * D2) W2          -> LDS;   4  aim_bar = (T * D1) W1 in two column halves; zbar = aim_bar * D_last -> memory (split form)
// The weights are ONE stream of 24 tiles of 128 rows x 32 k through a 3-stage LDS ring (DMA two steps ahead, across the seams, so
// that the next GEMM's weights arrive under the epilogue in between).
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "gemm_h2_common.h"
#include "kernels.h"

namespace aimnet {

namespace {
constexpr int HT = 64;                 // atoms per block
constexpr int HA_ST = 16384;           // activation ring stage: 64 rows x 192 B, rounded up to two DMA passes of the block
constexpr int HB_ST = 24576;           // weight ring stage: 128 rows x 192 B = three DMA passes
constexpr int HM_KB = HT * ROWB;       // one k-block of the resident operand (aliases the activation ring)
constexpr int H_LDS = 3 * HA_ST + 3 * HB_ST + 4 * HT * 4;
static_assert(4 * HM_KB <= 3 * HA_ST, "the resident operand aliases the activation ring");
}  // namespace

__global__ __launch_bounds__(512) void head_fused_kernel(HeadFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;  // wave tile: rows 32 wm .. + 31, columns 32 wn .. + 31 of a 64 x 128 product
  const int l16 = lane & 15, lc = lane >> 4;
  const int m0 = blockIdx.x * HT;
  const int n_tiles = a.grad ? 24 : 12;

  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_h;
  const unsigned ldsB = lds0 + 3 * HA_ST;
  float* red = reinterpret_cast<float*>(smem_h + 3 * HA_ST + 3 * HB_ST);  // [4 wn][64 rows]

  // DMA granules (16 B): weights G = p * 512 + tid -> row G / 12, plane (G % 12) / 4, slot G % 4 holding k-chunk slot ^ swz(row)
  unsigned goff_w1[3], goff_w[3], goff_a[2];  // byte offsets: W1 rows are 1536 B apart, the other three matrices 768 B
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    const int G = p * 512 + tid, row = G / 12, g12 = G % 12;
    const unsigned in_row = (g12 >> 2) * 64 + (((g12 & 3) ^ swz192(row)) << 4);
    goff_w1[p] = row * 1536u + in_row;
    goff_w[p] = row * 768u + in_row;
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int G = min(p * 512 + tid, HT * 12 - 1), row = G / 12, g12 = G % 12;
    goff_a[p] = (unsigned)(min(m0 + row, a.M - 1) - m0) * 2u * (unsigned)a.lda3 + (g12 >> 2) * 64 + (((g12 & 3) ^ swz192(row)) << 4);
  }
  const unsigned char* abase = reinterpret_cast<const unsigned char*>(a.aim3 + (size_t)m0 * a.lda3);
  auto issue = [&](int t) __attribute__((always_inline)) {  // tile t of the stream (uniform): weights, and aim for t < 8
    const int st = t % 3;
    unsigned char* bdst = smem_h + 3 * HA_ST + st * HB_ST + wid * 1024;
    if (t < 8) {
      unsigned char* adst = smem_h + st * HA_ST + wid * 1024;
      const unsigned char* ga = abase + (size_t)t * ROWB;
#pragma unroll
      for (int p = 0; p < 2; ++p) glds16b(ga + goff_a[p], adst + p * 8192);
      const unsigned char* gb = reinterpret_cast<const unsigned char*>(a.w1) + (size_t)t * ROWB;
#pragma unroll
      for (int p = 0; p < 3; ++p) glds16b(gb + goff_w1[p], bdst + p * 8192);
    } else {
      const unsigned short* w = t < 12 ? a.w2 : t < 16 ? a.w2t : a.w1t;
      const unsigned char* gb = reinterpret_cast<const unsigned char*>(w) + (t >= 20 ? 128 * 768 : 0) + (size_t)(t & 3) * ROWB;
#pragma unroll
      for (int p = 0; p < 3; ++p) glds16b(gb + goff_w[p], bdst + p * 8192);
    }
  };

  // biases / last-layer weights of this lane's columns (loaded before the DMA stream starts)
  f32x4 b1v[2], b2v[2], w3v[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = wn * 32 + 16 * j + 4 * lc;
    b1v[j] = *reinterpret_cast<const f32x4*>(a.b1 + col);
    b2v[j] = *reinterpret_cast<const f32x4*>(a.b2 + col);
    w3v[j] = *reinterpret_cast<const f32x4*>(a.w3 + col);
  }
  const float b3 = a.b3[0];

  const int rA = wm * 32 + l16, rB = wn * 32 + l16;
  const unsigned adA = lds0 + rA * ROWB + ((lc ^ swz192(rA)) << 4);
  const unsigned adB = ldsB + rB * ROWB + ((lc ^ swz192(rB)) << 4);

  f32x4 acc[2][2][2];  // [parity of the k-step][i][j]: two interleaved accumulator sets, subtracted in the epilogues (gemm_bf3a.hip)
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto total = [&](int i, int j) __attribute__((always_inline)) -> f32x4 { return acc[0][i][j] - acc[1][i][j]; };
  zero_acc();
  bf16x8 fa[2][3], fb[2][3];
  // one 32-k step: tile t of the stream against the activation tile at LDS address a_lds; NW = stream operations that may stay
  // outstanding at its wait (the younger tile t + 1; the memory counter retires loads in order)
  auto step = [&](int t, unsigned a_lds, auto par_c, auto nw_c) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_c)::value;  // parity of the k-step inside its GEMM: the accumulator set
    wait_vm<decltype(nw_c)::value>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const unsigned ob = adB + (t % 3) * HB_ST;
    read_strips<0, 2, 0>(fb, ob);
    read_strips<0, 2, 0>(fa, a_lds);
    read_strips<0, 2, 1>(fb, ob);
    read_strips<0, 2, 1>(fa, a_lds);
    read_strips<0, 2, 2>(fb, ob);
    read_strips<0, 2, 2>(fa, a_lds);
    wait_lgkm<0>();
    __builtin_amdgcn_sched_barrier(0);
#define AIMNET_HEAD_PRODUCT(PA, PB)                                                                             \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[PAR][i][j] = \
      __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][PB], fa[i][PA], acc[PAR][i][j], 0, 0, 0);
    AIMNET_HEAD_PRODUCT(1, 1)
    // the DMA of tile t + 2 is issued among the matrix instructions (in front of the fragment reads it cost 100 - 185 cycles per
    // piece on the critical path); its ring stage was last read in step t - 1, which every wave left before this step's barrier
    __builtin_amdgcn_sched_barrier(0);
    if (t + 2 < n_tiles) issue(t + 2);
    __builtin_amdgcn_sched_barrier(0);
    AIMNET_HEAD_PRODUCT(0, 1)
    AIMNET_HEAD_PRODUCT(1, 0)
    AIMNET_HEAD_PRODUCT(0, 2)
    AIMNET_HEAD_PRODUCT(2, 0)
    AIMNET_HEAD_PRODUCT(0, 0)
#undef AIMNET_HEAD_PRODUCT
    __builtin_amdgcn_sched_barrier(0);
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
#define AIMNET_HEAD_STEP(T, A_LDS, NW) \
  if ((T) & 1) step(T, A_LDS, P1{}, NW{}); else step(T, A_LDS, P0{}, NW{})  // (all four GEMMs start on an even tile of the stream)
  using W0 = std::integral_constant<int, 0>;
  using W3 = std::integral_constant<int, 3>;
  using W5 = std::integral_constant<int, 5>;
  // the wave's 32 x 32 block of a 64 x 128 matrix -> the resident LDS operand (k-block wn), split form
  auto to_lds = [&](const f32x4 (&v)[2][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = wm * 32 + 16 * i + l16, kc = 2 * j + (lc >> 1);
        const unsigned ad = lds0 + wn * HM_KB + row * ROWB + ((kc ^ swz192(row)) << 4) + (lc & 1) * 8;
        unsigned lo0, lo1, lo2, hi0, hi1, hi2;
        split3_pair(v[i][j][0], v[i][j][1], lo0, lo1, lo2);
        split3_pair(v[i][j][2], v[i][j][3], hi0, hi1, hi2);
        lds_write8<0>(ad, lo0, hi0);
        lds_write8<64>(ad, lo1, hi1);
        lds_write8<128>(ad, lo2, hi2);
      }
    wait_lgkm<0>();
  };
  auto seam = [&]() __attribute__((always_inline)) {  // every wave has read the operand the next epilogue overwrites
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  issue(0);
  issue(1);
  // ---- 1: Z1 = aim W1^T ------------------------------------------------------------------------------------------------
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (t < 7) { AIMNET_HEAD_STEP(t, adA + (t % 3) * HA_ST, W5); }
    else { AIMNET_HEAD_STEP(t, adA + (t % 3) * HA_ST, W3); }
  }
  f32x4 D1[2][2], v[2][2];
  {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float h, d;
          gelu_and_grad(total(i, j)[r] + b1v[j][r], h, d);
          v[i][j][r] = h;
          D1[i][j][r] = d;
        }
  }
  seam();
  to_lds(v);
  zero_acc();
  // ---- 2: Z2 = H1 W2^T, e = H2 . w3 + b3 -------------------------------------------------------------------------------
#pragma unroll
  for (int t = 8; t < 12; ++t) {
    if (t < 11) { AIMNET_HEAD_STEP(t, adA + (t - 8) * HM_KB, W3); }
    else if (a.grad) { AIMNET_HEAD_STEP(t, adA + (t - 8) * HM_KB, W3); }
    else { AIMNET_HEAD_STEP(t, adA + (t - 8) * HM_KB, W0); }
  }
  {
    float pe[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float h, d;
          gelu_and_grad(total(i, j)[r] + b2v[j][r], h, d);
          pe[i] += h * w3v[j][r];
          v[i][j][r] = w3v[j][r] * d;  // adjoint seed of z2: dE/de = 1
        }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      pe[i] += __shfl_xor(pe[i], 16);
      pe[i] += __shfl_xor(pe[i], 32);
      if (lc == 0) red[wn * HT + wm * 32 + 16 * i + l16] = pe[i];
    }
  }
  seam();
  if (tid < HT && m0 + tid < a.M) a.e_atom[m0 + tid] = ((red[tid] + red[HT + tid]) + (red[2 * HT + tid] + red[3 * HT + tid])) + b3;
  if (!a.grad) return;
  to_lds(v);
  zero_acc();
  // ---- 3: T = (w3 * D2) W2 ------------------------------------------------------------------------------------------
#pragma unroll
  for (int t = 12; t < 16; ++t) { AIMNET_HEAD_STEP(t, adA + (t - 12) * HM_KB, W3); }
  {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) v[i][j] = total(i, j) * D1[i][j];
  }
  seam();
  to_lds(v);
  zero_acc();
  // ---- 4: aim_bar = (T * D1) W1, two halves of 128 columns; zbar = aim_bar * D_last -----------------------------------------
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int t = 16 + 4 * half; t < 20 + 4 * half; ++t) {
      const int kb = t - 16 - 4 * half;
      if (t < 23) { AIMNET_HEAD_STEP(t, adA + kb * HM_KB, W3); }
      else { AIMNET_HEAD_STEP(t, adA + kb * HM_KB, W0); }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = m0 + wm * 32 + 16 * i + l16;
      if (row >= a.M) continue;  // (the lanes of a row agree: the exchange inside store_bf3_tile_pair stays consistent)
      const int col0 = half * 128 + wn * 32;
      const f32x4 d0 = *reinterpret_cast<const f32x4*>(a.dlast + (size_t)row * a.ldd + col0 + 4 * lc);
      const f32x4 d1 = *reinterpret_cast<const f32x4*>(a.dlast + (size_t)row * a.ldd + col0 + 16 + 4 * lc);
      store_bf3_tile_pair(a.zbar3 + (size_t)row * a.ldz3, col0, lc, total(i, 0) * d0, total(i, 1) * d1);
    }
    zero_acc();
  }
#undef AIMNET_HEAD_STEP
}

// ---- the same head on fp16x2-split operands ("h2", gemm_h2_common.h / gemm_h2.hip): three products per tile and k-step (ah bh
// into the two interleaved sets, ah bl + al bh into a third that the epilogues scale by 1 / 4096), 128-byte rows, LDS tiles in 16-row
// strips of [hi 1 KiB][lo 1 KiB].  Same stream of 24 weight tiles, same seams; the resident operand has a buffer of its own.
namespace {
constexpr int H2A_ST = 8192;                 // activation ring stage: 64 rows x 128 B = one DMA pass of the block
constexpr int H2B_ST = 16384;                // weight ring stage: 128 rows x 128 B = two DMA passes
constexpr int H2M_KB = HT * H2_ROWB;         // one k-block of the resident operand (4 strips)
constexpr int H2_RES = 3 * H2A_ST + 3 * H2B_ST;  // byte offset of the resident operand
constexpr int H2_LDS = H2_RES + 4 * H2M_KB + 4 * HT * 4;
}  // namespace

__global__ __launch_bounds__(512) void head_fused_h2_kernel(HeadFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;  // wave tile: rows 32 wm .. + 31, columns 32 wn .. + 31 of a 64 x 128 product
  const int l16 = lane & 15, lc = lane >> 4;
  const int m0 = blockIdx.x * HT;
  const int n_tiles = a.grad ? 24 : 12;

  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_h;
  const unsigned ldsB = lds0 + 3 * H2A_ST, ldsR = lds0 + H2_RES;
  float* red = reinterpret_cast<float*>(smem_h + H2_RES + 4 * H2M_KB);  // [4 wn][64 rows]

  // DMA: pass p, wave wid -> KiB q = 8 p + wid of the stage = plane q & 1 of the 16-row strip q >> 1; lane -> row lane >> 2 of the
  // strip, slot lane & 3 holding k-chunk slot ^ swz(row)
  unsigned goff_w1[2], goff_w[2], goff_a;  // byte offsets: W1 rows are 1024 B apart, the other three matrices 512 B
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int q = p * 8 + wid, row = (q >> 1) * 16 + (lane >> 2);
    const unsigned in_row = (q & 1) * 64 + (((lane & 3) ^ swz_h2(row)) << 4);
    goff_w1[p] = row * 1024u + in_row;
    goff_w[p] = row * 512u + in_row;
  }
  {
    const int row = (wid >> 1) * 16 + (lane >> 2);
    goff_a = (unsigned)(min(m0 + row, a.M - 1) - m0) * 2u * (unsigned)a.lda3 + (wid & 1) * 64 + (((lane & 3) ^ swz_h2(row)) << 4);
  }
  const unsigned char* abase = reinterpret_cast<const unsigned char*>(a.aim3 + (size_t)m0 * a.lda3);
  auto issue = [&](int t) __attribute__((always_inline)) {  // tile t of the stream (uniform): weights, and aim for t < 8
    const int st = t % 3;
    unsigned char* bdst = smem_h + 3 * H2A_ST + st * H2B_ST + wid * 1024;
    if (t < 8) {
      glds16b(abase + (size_t)t * H2_ROWB + goff_a, smem_h + st * H2A_ST + wid * 1024);
      const unsigned char* gb = reinterpret_cast<const unsigned char*>(a.w1) + (size_t)t * H2_ROWB;
#pragma unroll
      for (int p = 0; p < 2; ++p) glds16b(gb + goff_w1[p], bdst + p * 8192);
    } else {
      const unsigned short* w = t < 12 ? a.w2 : t < 16 ? a.w2t : a.w1t;
      const unsigned char* gb = reinterpret_cast<const unsigned char*>(w) + (t >= 20 ? 128 * 512 : 0) + (size_t)(t & 3) * H2_ROWB;
#pragma unroll
      for (int p = 0; p < 2; ++p) glds16b(gb + goff_w[p], bdst + p * 8192);
    }
  };

  f32x4 b1v[2], b2v[2], w3v[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = wn * 32 + 16 * j + 4 * lc;
    b1v[j] = *reinterpret_cast<const f32x4*>(a.b1 + col);
    b2v[j] = *reinterpret_cast<const f32x4*>(a.b2 + col);
    w3v[j] = *reinterpret_cast<const f32x4*>(a.w3 + col);
  }
  const float b3 = a.b3[0];

  const unsigned frag = l16 * 64 + ((lc ^ swz_h2(l16)) << 4);
  const unsigned adA = lds0 + wm * 2 * H2_STRIP + frag;   // activation ring stage 0, the wave's two row strips
  const unsigned adR = ldsR + wm * 2 * H2_STRIP + frag;   // resident operand, k-block 0
  const unsigned adB = ldsB + wn * 2 * H2_STRIP + frag;

  f32x4 acc[3][2][2];  // [0], [1]: ah bh of the even / odd k-steps; [2]: cross terms (x 4096)
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < 3; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto total = [&](int i, int j) __attribute__((always_inline)) -> f32x4 {
    return (acc[0][i][j] - acc[1][i][j]) + acc[2][i][j] * H2_INV_SCALE;
  };
  zero_acc();
  f16x8 fa[2][2], fb[2][2];
  auto step = [&](int t, unsigned a_lds, auto par_c, auto nw_c) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_c)::value;
    wait_vm<decltype(nw_c)::value>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const unsigned ob = adB + (t % 3) * H2B_ST;
    read_strips_h<0, 2, 1>(fb, ob);
    read_strips_h<0, 2, 0>(fa, a_lds);
    read_strips_h<0, 2, 0>(fb, ob);
    read_strips_h<0, 2, 1>(fa, a_lds);
    wait_lgkm<0>();
    __builtin_amdgcn_sched_barrier(0);
#define AIMNET_HEAD_PRODUCT(SET, PA, PB)                                                                        \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[SET][i][j] = \
      __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j][PB], fa[i][PA], acc[SET][i][j], 0, 0, 0);
    AIMNET_HEAD_PRODUCT(2, 0, 1)
    __builtin_amdgcn_sched_barrier(0);
    if (t + 2 < n_tiles) issue(t + 2);  // among the matrix instructions; its ring stage was last read in step t - 1
    __builtin_amdgcn_sched_barrier(0);
    AIMNET_HEAD_PRODUCT(PAR, 0, 0)
    AIMNET_HEAD_PRODUCT(2, 1, 0)
#undef AIMNET_HEAD_PRODUCT
    __builtin_amdgcn_sched_barrier(0);
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
#define AIMNET_HEAD_STEP(T, A_LDS, NW) \
  if ((T) & 1) step(T, A_LDS, P1{}, NW{}); else step(T, A_LDS, P0{}, NW{})
  using W0 = std::integral_constant<int, 0>;
  using W2 = std::integral_constant<int, 2>;
  using W3 = std::integral_constant<int, 3>;
  // the wave's 32 x 32 block of a 64 x 128 matrix -> the resident LDS operand (k-block wn), activation form (lo of odd k-blocks negated)
  auto to_lds = [&](const f32x4 (&v)[2][2]) __attribute__((always_inline)) {
    const float sc = (wn & 1) ? -H2_SCALE : H2_SCALE;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = wm * 32 + 16 * i + l16, kc = 2 * j + (lc >> 1);
        const unsigned ad = ldsR + wn * H2M_KB + (row >> 4) * H2_STRIP + (row & 15) * 64 + ((kc ^ swz_h2(row)) << 4) + (lc & 1) * 8;
        unsigned h0, l0, h1, l1;
        split2_pair(v[i][j][0], v[i][j][1], sc, h0, l0);
        split2_pair(v[i][j][2], v[i][j][3], sc, h1, l1);
        lds_write8<0>(ad, h0, h1);
        lds_write8<1024>(ad, l0, l1);
      }
    wait_lgkm<0>();
  };
  auto seam = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  issue(0);
  issue(1);
  // ---- 1: Z1 = aim W1^T ------------------------------------------------------------------------------------------------
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (t < 7) { AIMNET_HEAD_STEP(t, adA + (t % 3) * H2A_ST, W3); }
    else { AIMNET_HEAD_STEP(t, adA + (t % 3) * H2A_ST, W2); }
  }
  f32x4 D1[2][2], v[2][2];
  {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const f32x4 z = total(i, j);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float h, d;
          gelu_and_grad(z[r] + b1v[j][r], h, d);
          v[i][j][r] = h;
          D1[i][j][r] = d;
        }
      }
  }
  to_lds(v);  // (its own buffer: no wave reads it before the next step's barrier)
  zero_acc();
  // ---- 2: Z2 = H1 W2^T, e = H2 . w3 + b3 -------------------------------------------------------------------------------
#pragma unroll
  for (int t = 8; t < 12; ++t) {
    if (t < 11) { AIMNET_HEAD_STEP(t, adR + (t - 8) * H2M_KB, W2); }
    else if (a.grad) { AIMNET_HEAD_STEP(t, adR + (t - 8) * H2M_KB, W2); }
    else { AIMNET_HEAD_STEP(t, adR + (t - 8) * H2M_KB, W0); }
  }
  {
    float pe[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const f32x4 z = total(i, j);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float h, d;
          gelu_and_grad(z[r] + b2v[j][r], h, d);
          pe[i] += h * w3v[j][r];
          v[i][j][r] = w3v[j][r] * d;  // adjoint seed of z2: dE/de = 1
        }
      }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      pe[i] += __shfl_xor(pe[i], 16);
      pe[i] += __shfl_xor(pe[i], 32);
      if (lc == 0) red[wn * HT + wm * 32 + 16 * i + l16] = pe[i];
    }
  }
  seam();  // every wave has read the resident operand the next epilogue overwrites; `red` is complete
  if (tid < HT && m0 + tid < a.M) a.e_atom[m0 + tid] = ((red[tid] + red[HT + tid]) + (red[2 * HT + tid] + red[3 * HT + tid])) + b3;
  if (!a.grad) return;
  to_lds(v);
  zero_acc();
  // ---- 3: T = (w3 * D2) W2 ------------------------------------------------------------------------------------------
#pragma unroll
  for (int t = 12; t < 16; ++t) { AIMNET_HEAD_STEP(t, adR + (t - 12) * H2M_KB, W2); }
  {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) v[i][j] = total(i, j) * D1[i][j];
  }
  seam();
  to_lds(v);
  zero_acc();
  // ---- 4: aim_bar = (T * D1) W1, two halves of 128 columns; zbar = aim_bar * D_last -----------------------------------------
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int t = 16 + 4 * half; t < 20 + 4 * half; ++t) {
      const int kb = t - 16 - 4 * half;
      if (t < 23) { AIMNET_HEAD_STEP(t, adR + kb * H2M_KB, W2); }
      else { AIMNET_HEAD_STEP(t, adR + kb * H2M_KB, W0); }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = m0 + wm * 32 + 16 * i + l16;
      if (row >= a.M) continue;  // (the lanes of a row agree: the exchange inside store_h2_tile_pair stays consistent)
      const int col0 = half * 128 + wn * 32;
      const f32x4 d0 = *reinterpret_cast<const f32x4*>(a.dlast + (size_t)row * a.ldd + col0 + 4 * lc);
      const f32x4 d1 = *reinterpret_cast<const f32x4*>(a.dlast + (size_t)row * a.ldd + col0 + 16 + 4 * lc);
      store_h2_tile_pair(a.zbar3 + (size_t)row * a.ldz3, col0, lc, total(i, 0) * d0, total(i, 1) * d1);
    }
    zero_acc();
  }
#undef AIMNET_HEAD_STEP
}

int launch_head_fused(hipStream_t s, const HeadFusedArgs& a) {
  if (a.M <= 0) return 0;
  const int blk = a.fmt == 2 ? 64 : 96;  // 16-bit elements per 32-k block of a row
  if ((a.lda3 % blk) || (a.ldz3 % blk) || (a.ldd & 3) ||
      (((size_t)a.aim3 | (size_t)a.w1 | (size_t)a.w2 | (size_t)a.w2t | (size_t)a.w1t | (size_t)a.b1 | (size_t)a.b2 | (size_t)a.w3 |
        (size_t)a.dlast | (size_t)a.zbar3) & 15)) {
    set_last_error("head_fused: operands must be 16-byte aligned, row strides whole 32-k blocks");
    return -1;
  }
  static PerDeviceOnce once;
  if (once.first()) {
    AIMNET_HIP_CHECK(hipFuncSetAttribute((const void*)head_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    AIMNET_HIP_CHECK(hipFuncSetAttribute((const void*)head_fused_h2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  }
  if (a.fmt == 2) hipLaunchKernelGGL(head_fused_h2_kernel, dim3(ceil_div(a.M, HT)), dim3(512), H2_LDS, s, a);
  else hipLaunchKernelGGL(head_fused_kernel, dim3(ceil_div(a.M, HT)), dim3(512), H_LDS, s, a);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

}  // namespace aimnet
