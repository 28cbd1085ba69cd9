// gemm_h2.hip - the MLP GEMM on fp16x2-split operands ("h2", gemm_h2_common.h): three matrix instructions per tile and k-step
// instead of the six of the bf16x3 split, 4 instead of 6 bytes per operand element.
//
//   C[M,N] = A[M,K] . Bt[N,K]^T, fused epilogues: the contract of gemm_bf3a.hip (which replaces the torch addmm + GELU calls of
//   aimnet/modules/core.py:11-46); A and Bt are handed over in the h2 layout (per row, K/32 blocks of [hi: 32 fp16][lo: 32 fp16]
//   = 128 B, fp32 == hi + lo / 4096 to 2^-24) and the epilogue can write C in the same layout for the next layer (OUT2).
//   Products per tile and k-step: ah bh into one of two interleaved accumulator sets (even / odd k-steps, the weights' hi planes
//   of the odd k-blocks negated: the one-signed truncation of the matrix pipe cancels in the difference, as in gemm_bf3a.hip),
//   ah bl and al bh into a third set that the epilogue scales by 1 / 4096.
//
// Schedule: the ping-pong of gemm_bf3a.hip - waves 0-3 (group 0) and waves 4-7 (group 1, same SIMDs) alternate LOAD and COMPUTE
// segments half a step apart, one s_barrier per segment; group 0 DMAs the whole activation tile of step j+1 (two LDS stages),
// group 1 the weight tile of step j+2 (three stages).  LDS tiles: 16-row strips of [hi 1 KiB][lo 1 KiB] (gemm_h2_common.h),
// one DMA wave-instruction per plane of a strip.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>

#include "common.h"
#include "gemm_h2_common.h"
#include "kernels.h"

namespace aimnet {

constexpr int h2_passes(int rows) { return (rows + 31) / 32; }  // DMA wave-instructions per wave of the issuing group (a pass = 4 KiB = 2 strips)
// NSA = activation ring depth (weights: h2_nsb).  2: the activation tile is requested ONE step ahead (weights: two).  3: two steps
// ahead as well; 4: both three steps ahead -
// for launches that leave CUs idle (a few hundred to ~2 000 rows): there a step is as long as the request's latency whatever the
// tile (~0.6 us; the GEMM family costs the same 0.25 ms from 384 to 2 304 atoms), and the second step of lead takes 8 % off it;
// on full grids the extra 20 KB of LDS cost 0.6 % (profiles/r5_size_sweep.jsonl).
constexpr int h2_nsb(int NSA) { return NSA == 4 ? 4 : 3; }  // weight ring depth that goes with an activation ring depth
constexpr int h2_lds_bytes(int TM, int TN, int NSA) { return NSA * h2_passes(TM) * 4096 + h2_nsb(NSA) * h2_passes(TN) * 4096; }

#ifdef AIMNET_BF3_TIMING
__device__ unsigned long long g_h2_stamps[1024];
#endif

template <int EPI, int SM, int SN, int WN, bool OUT3, int NSA>
__global__ __launch_bounds__(512, 2) void gemm_h2_kernel(const unsigned short* __restrict__ A3, int lda3,
                                                           const unsigned short* __restrict__ Bt, int ldb, int M, int N, int K,
                                                           const float* __restrict__ bias, float* __restrict__ C,
                                                           unsigned short* __restrict__ C3, int ldc3, float* __restrict__ D, int ldc,
                                                           const int* __restrict__ brow, int ldbias, int alt) {
  static_assert(WN == 8 || WN == 4 || WN == 2, "waves across N");
  constexpr int WM = 8 / WN;
  constexpr int TM = 16 * SM * WM, TN = 16 * SN * WN;
  constexpr int NPA = h2_passes(TM), NPB = h2_passes(TN);
  constexpr int SA_BYTES = NPA * 4096, SB_BYTES = NPB * 4096;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_a[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const bool late = wid >= 4;  // group 1 runs one segment behind group 0
  const int w4 = wid & 3;

  const int tiles_n = (N + TN - 1) / TN;
  const int nwg = gridDim.x;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = blockIdx.x & 7;
  const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);
  const int m0 = (wg / tiles_n) * TM, n0 = (wg % tiles_n) * TN;

  // Accumulation.  The matrix pipe TRUNCATES the aligned sum of its 32 products and the accumulator towards minus infinity
  // (gemm_bf3a.hip, "Accumulation"; profiles/r4_bf3_bias.txt): one-signed, it does not average out over atoms.  As there, the hi
  // planes of the weights' ODD k-blocks are stored negated and the hi x hi products of even / odd k-steps go to two accumulator
  // sets whose difference the epilogue takes.  The cross terms are 2^-12 of that sum: their own truncation is irrelevant and
  // they share one set (activations carry the lo planes of the odd k-blocks negated, so both cross products keep their sign).
  f32x4 acc[3][SM][SN];  // [0], [1]: ah bh of the even / odd k-steps; [2]: the cross terms ah bl + al bh (x 4096)
#pragma unroll
  for (int h = 0; h < 3; ++h)
#pragma unroll
    for (int i = 0; i < SM; ++i)
#pragma unroll
      for (int j = 0; j < SN; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_a;
  const unsigned ldsB = lds0 + NSA * SA_BYTES;

  // DMA of the issuing group: pass p, wave w4 -> KiB q = 4 p + w4 of the stage = plane q & 1 of the 16-row strip q >> 1; lane ->
  // row (lane >> 2) of the strip, slot lane & 3 holding k-chunk slot ^ swz(row).  Strips beyond the tile (padding of the last
  // pass) re-read the last row into the stage's padding; rows beyond the matrix re-read its last row.
  // Offsets are bytes relative to the tile's first row (32 bits: a tile spans < 200 rows).
  constexpr int NPMAX = NPA > NPB ? NPA : NPB;
  unsigned goff[NPMAX];
  {
    const int rmax = (late ? TN : TM) - 1;
    const int r0 = late ? n0 : m0, rlim = (late ? N : M) - 1;
    const unsigned ldbytes = 2u * (unsigned)(late ? ldb : lda3);
#pragma unroll
    for (int p = 0; p < NPMAX; ++p) {
      const int q = p * 4 + w4;
      const int row = min((q >> 1) * 16 + (lane >> 2), rmax), pl = q & 1;
      const int kcx = (lane & 3) ^ swz_h2(row);
      goff[p] = (unsigned)(min(r0 + row, rlim) - r0) * ldbytes + pl * 64 + kcx * 16;
    }
  }
  const unsigned char* abase = reinterpret_cast<const unsigned char*>(A3 + (size_t)m0 * lda3);
  const unsigned char* bbase = reinterpret_cast<const unsigned char*>(Bt + (size_t)n0 * ldb);
  // passes [P0, P1) of one tile
  auto dma_a = [&](int stage, int kt, auto p0_c, auto p1_c) __attribute__((always_inline)) {
    unsigned char* base = smem_a + stage * SA_BYTES + w4 * 1024;
    const unsigned char* g = abase + (size_t)kt * H2_ROWB;
#pragma unroll
    for (int p = decltype(p0_c)::value; p < decltype(p1_c)::value; ++p) glds16b(g + goff[p], base + p * 4096);
  };
  auto dma_b = [&](int stage, int kt, auto p0_c, auto p1_c) __attribute__((always_inline)) {
    unsigned char* base = smem_a + NSA * SA_BYTES + stage * SB_BYTES + w4 * 1024;
    const unsigned char* g = bbase + (size_t)kt * H2_ROWB;
#pragma unroll
    for (int p = decltype(p0_c)::value; p < decltype(p1_c)::value; ++p) glds16b(g + goff[p], base + p * 4096);
  };
  using PZ = std::integral_constant<int, 0>;
  using PAE = std::integral_constant<int, NPA>;
  using PBE = std::integral_constant<int, NPB>;

  // fragment addresses: row r, plane P, k-chunk c = lane >> 4 -> (r >> 4) * 2048 + P * 1024 + (r & 15) * 64 + (c ^ swz(r)) * 16
  const int l16 = lane & 15, lc = lane >> 4;
  const unsigned adA = lds0 + wm * SM * H2_STRIP + l16 * 64 + ((lc ^ swz_h2(l16)) << 4);
  const unsigned adB = ldsB + wn * SN * H2_STRIP + l16 * 64 + ((lc ^ swz_h2(l16)) << 4);

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  const int nk = K >> 5;
  // every step issues the same operations: k-steps past the end of K are clamped to the last one (redundant tiles nothing reads)
  auto kc = [&](int k) __attribute__((always_inline)) { return min(k, nk - 1); };

#ifdef AIMNET_BF3_TIMING
  int n_ts = 0;
  auto TS = [&]() __attribute__((always_inline)) {
    if (blockIdx.x == 0 && (wid & 3) == 0 && n_ts < 512) {
      const unsigned long long t = __builtin_readcyclecounter();
      if (lane == 0) g_h2_stamps[(wid >> 2) * 512 + n_ts] = t;
      ++n_ts;
    }
    __builtin_amdgcn_sched_barrier(0);
  };
#else
  auto TS = [&]() __attribute__((always_inline)) {};
#endif
  f16x8 fa[SM][2], fb[SN][2];
#define AIMNET_H2_PRODUCT(SET, PA, PB)                                                                                      \
  _Pragma("unroll") for (int i = 0; i < SM; ++i) _Pragma("unroll") for (int jj = 0; jj < SN; ++jj) acc[SET][i][jj] = \
      __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[jj][PB], fa[i][PA], acc[SET][i][jj], 0, 0, 0);
  TS();
  // ---- prologue: A(0) by group 0; B(0), B(1) by group 1
  constexpr int NSB = h2_nsb(NSA);  // lead of the requests: NSA - 1 steps for activation tiles, NSB - 1 for weight tiles
  if (!late) {
#pragma unroll
    for (int t = 0; t < NSA - 1; ++t) dma_a(t, kc(t), PZ{}, PAE{});
    wait_vm<(NSA - 2) * NPA>();  // A(0) has landed
  } else {
#pragma unroll
    for (int t = 0; t < NSB - 1; ++t) dma_b(t, kc(t), PZ{}, PBE{});
    wait_vm<(NSB - 2) * NPB>();  // B(0) has landed
  }
  __builtin_amdgcn_sched_barrier(0);
  TS();
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  // sa / sb: ring stages of this step's activation / weight tile
  auto seg_load = [&](int j, int sa, int sb, auto g_c) __attribute__((always_inline)) {
    constexpr int G = decltype(g_c)::value;
    const unsigned oa = adA + sa * SA_BYTES, ob = adB + sb * SB_BYTES;
    read_strips_h<0, SN, 1>(fb, ob);
    read_strips_h<0, SM, 0>(fa, oa);
    read_strips_h<0, SN, 0>(fb, ob);
    read_strips_h<0, SM, 1>(fa, oa);
    if constexpr (G == 0) {
      dma_a(sa == 0 ? NSA - 1 : sa - 1, kc(j + NSA - 1), PZ{}, PAE{});  // stage (sa + NSA - 1) % NSA held A(j - 1)
      wait_lgkm<0>();
    } else {
      dma_b(sb == 0 ? NSB - 1 : sb - 1, kc(j + NSB - 1), PZ{}, PBE{});  // stage (sb + NSB - 1) % NSB
      wait_vm<(NSB - 2) * NPB>();  // the weight tile of step j+1 has landed (the later requests may be outstanding)
      wait_lgkm<0>();
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto seg_compute = [&](auto par_c, auto g_c) __attribute__((always_inline)) {
    constexpr int G = decltype(g_c)::value, PAR = decltype(par_c)::value;  // PAR: parity of the k-step = accumulator set
    __builtin_amdgcn_sched_barrier(0);
    AIMNET_H2_PRODUCT(2, 0, 1)
    AIMNET_H2_PRODUCT(PAR, 0, 0)
    AIMNET_H2_PRODUCT(2, 1, 0)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (G == 0) wait_vm<(NSA - 2) * NPA>();  // the activation tile of step j+1 has landed
    __builtin_amdgcn_sched_barrier(0);
  };
  auto bar = [&]() __attribute__((always_inline)) {
    TS();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    TS();
  };
  auto run = [&](auto g_c) __attribute__((always_inline)) {
    int sa = 0, sb = 0, j = 0;
    auto next = [&]() __attribute__((always_inline)) {
      sa = sa == NSA - 1 ? 0 : sa + 1;
      sb = sb == NSB - 1 ? 0 : sb + 1;
    };
    for (; j + 1 < nk; j += 2) {
      seg_load(j, sa, sb, g_c);
      bar();
      seg_compute(I0{}, g_c);
      next();
      bar();
      seg_load(j + 1, sa, sb, g_c);
      bar();
      seg_compute(I1{}, g_c);
      next();
      if (j + 2 < nk) bar();
    }
    if (j < nk) {  // odd number of steps
      seg_load(j, sa, sb, g_c);
      bar();
      seg_compute(I0{}, g_c);
    }
  };
  if (late) {
    bar();
    run(I1{});
  } else {
    run(I0{});
    bar();  // group 0 has 2 nk segments, group 1 an empty one in front: both pass 2 nk barriers
  }
#undef AIMNET_H2_PRODUCT
  wait_vm<0>();  // the clamped look-ahead of the last steps: the wave must not end (LDS released) under its DMA
  __builtin_amdgcn_sched_barrier(0);
  TS();

  // epilogue: sfin * acc[i][j][r] = C[m0 + wm*16*SM + 16 i + (lane&15)][n0 + wn*16*SN + 16 j + 4 (lane>>4) + r]
  // even-step set +/- odd-step set: alt 0 = plain weights (sum), 1 = BF3_ALT weights from an even k-block (difference), 2 = from an odd one
  const float s0 = alt == 2 ? -1.0f : 1.0f, s1 = alt == 1 ? -1.0f : 1.0f;
  // value of tile (i, j) after the fused epilogue (GELU' / chain-rule factor through D); false: outside the matrix
  auto finish = [&](int i, int j, f32x4& v) __attribute__((always_inline)) -> bool {
    const int col = n0 + wn * 16 * SN + 16 * j + 4 * lc;
    const int row = m0 + wm * 16 * SM + 16 * i + l16;
    if (col >= N || row >= M) return false;
    const size_t o = (size_t)row * ldc + col;
    v = acc[0][i][j] * s0 + acc[1][i][j] * s1 + acc[2][i][j] * H2_INV_SCALE;
    if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) {
      const f32x4 bv = brow ? *reinterpret_cast<const f32x4*>(bias + (size_t)min(63, max(0, brow[row])) * ldbias + col)
                            : *reinterpret_cast<const f32x4*>(bias + col);
      v = v + bv;
    }
    if (EPI == EPI_BIAS_GELU) {
      f32x4 d;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float hh, dd;
        gelu_and_grad(v[r], hh, dd);
        v[r] = hh;
        d[r] = dd;
      }
      if (D) *reinterpret_cast<f32x4*>(D + o) = d;
    } else if (EPI == EPI_MUL) {
      v = v * *reinterpret_cast<const f32x4*>(D + o);
    }
    return true;
  };
  if constexpr (OUT3) {
    // h2 output: 16-byte stores per plane through store_h2_tile_pair
#pragma unroll
    for (int j = 0; j < SN; j += 2) {
#pragma unroll
      for (int i = 0; i < SM; ++i) {
        const int row = m0 + wm * 16 * SM + 16 * i + l16;
        unsigned short* crow = C3 + (size_t)row * ldc3;
        if (j + 1 < SN) {
          f32x4 v0, v1;
          const bool ok = finish(i, j, v0);
          finish(i, j + 1, v1);  // N % 32 == 0 and 32-aligned tile pairs: both tiles are inside or both outside
          if (!ok) continue;
          store_h2_tile_pair(crow, n0 + wn * 16 * SN + 16 * j, lc, v0, v1);
        } else {
          f32x4 v;
          if (!finish(i, j, v)) continue;
          store_h2_x4(crow, n0 + wn * 16 * SN + 16 * j + 4 * lc, v);
        }
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < SN; ++j) {
#pragma unroll
      for (int i = 0; i < SM; ++i) {
        f32x4 v;
        if (!finish(i, j, v)) continue;
        const int col = n0 + wn * 16 * SN + 16 * j + 4 * lc;
        const int row = m0 + wm * 16 * SM + 16 * i + l16;
        *reinterpret_cast<f32x4*>(C + (size_t)row * ldc + col) = v;
      }
    }
  }
#ifdef AIMNET_BF3_TIMING
  __builtin_amdgcn_sched_barrier(0);
  TS();
  wait_vm<0>();
  __builtin_amdgcn_sched_barrier(0);
  TS();
#endif
}

// (A one-instruction-stream-per-wave schedule - fragments of step j+1 fetched into the registers step j's products release, one
// barrier per step - was built and measured EQUAL, step 1.3452 vs 1.3456 ms: 60 matrix instructions per SIMD and step at the
// 16x16x32 shape's own rate are 1 164 of the ~1 500 cycles either schedule takes; profiles/r5_gemm_h2.md.  Removed; commit d9e6536.)
static int g_h2_deep = 4;  // AIMNET_H2_DEEP: ring depth (2, 3 or 4) of launches that fill at most half of the CUs
template <int SM, int SN, int WN>
static int launch_h2(hipStream_t stream, int epi, bool out3, const unsigned short* A3, int lda3, const unsigned short* Bt, int ldb,
                       int M, int N, int K, const float* bias, float* C, unsigned short* C3, int ldc3, float* D, int ldc,
                       const int* brow, int ldbias, int alt) {
  constexpr int WM = 8 / WN, TM = 16 * SM * WM, TN = 16 * SN * WN;
  const int tiles = ceil_div(M, TM) * ceil_div(N, TN);
  static_assert(h2_lds_bytes(TM, TN, 4) <= 160 * 1024, "LDS");
  const int deep = 2 * tiles <= device_cus() ? g_h2_deep : 2;  // at most half of the CUs busy: longer request lead (see NSA)
  const size_t lds = (size_t)(deep == 4 ? h2_lds_bytes(TM, TN, 4) : deep == 3 ? h2_lds_bytes(TM, TN, 3) : h2_lds_bytes(TM, TN, 2));
  dim3 grid(tiles), block(512);
#define AIMNET_H2_LAUNCH(E, O3)                                                                                            \
  {                                                                                                                          \
    static PerDeviceOnce once;                                                                                               \
    if (once.first()) {                                                                                                      \
      AIMNET_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_h2_kernel<E, SM, SN, WN, O3, 2>,                             \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                         \
      AIMNET_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_h2_kernel<E, SM, SN, WN, O3, 3>,                             \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                         \
      AIMNET_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_h2_kernel<E, SM, SN, WN, O3, 4>,                             \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                         \
    }                                                                                                                        \
    if (deep == 4)                                                                                                           \
      hipLaunchKernelGGL((gemm_h2_kernel<E, SM, SN, WN, O3, 4>), grid, block, lds, stream, A3, lda3, Bt, ldb, M, N, K,    \
                         bias, C, C3, ldc3, D, ldc, brow, ldbias, alt);                                                      \
    else if (deep == 3)                                                                                                      \
      hipLaunchKernelGGL((gemm_h2_kernel<E, SM, SN, WN, O3, 3>), grid, block, lds, stream, A3, lda3, Bt, ldb, M, N, K,    \
                         bias, C, C3, ldc3, D, ldc, brow, ldbias, alt);                                                      \
    else                                                                                                                     \
      hipLaunchKernelGGL((gemm_h2_kernel<E, SM, SN, WN, O3, 2>), grid, block, lds, stream, A3, lda3, Bt, ldb, M, N, K,    \
                         bias, C, C3, ldc3, D, ldc, brow, ldbias, alt);                                                      \
  }
  if (out3) {
    switch (epi) {
      case EPI_BIAS_GELU: AIMNET_H2_LAUNCH(EPI_BIAS_GELU, true) break;
      case EPI_MUL: AIMNET_H2_LAUNCH(EPI_MUL, true) break;
      default:
        set_last_error("gemm_h2: split output exists for the GELU and chain-rule epilogues only (got %d)", epi);
        return -1;
    }
  } else {
    switch (epi) {
      case EPI_NONE: AIMNET_H2_LAUNCH(EPI_NONE, false) break;
      case EPI_BIAS: AIMNET_H2_LAUNCH(EPI_BIAS, false) break;
      case EPI_BIAS_GELU: AIMNET_H2_LAUNCH(EPI_BIAS_GELU, false) break;
      case EPI_MUL: AIMNET_H2_LAUNCH(EPI_MUL, false) break;
      default:
        set_last_error("gemm_h2: bad epilogue %d", epi);
        return -1;
    }
  }
#undef AIMNET_H2_LAUNCH
  AIMNET_LAUNCH_CHECK();
  return 0;
}

static int g_h2_force_tile = 0;  // AIMNET_H2_TILE forces one configuration (A/B runs)

struct H2Cand { int id, tm, tn; };
// id = 100 * WN (waves across N; 8 / WN across M) + 10 * SM + SN; block tile (16 SM 8 / WN) x (16 SN WN)
static const H2Cand kH2Cands[] = {{452, 160, 128}, {224, 128, 128}, {432, 96, 128}, {422, 64, 128},
                                      {223, 128, 96},  {851, 80, 128},  {234, 192, 128}};

static int choose_h2_tile(int M, int N) {
  const long n_cu = device_cus();
  int best = kH2Cands[0].id;
  double best_cost = 1e300;
  for (const H2Cand& c : kH2Cands) {
    const long tiles = (long)ceil_div(M, c.tm) * ceil_div(N, c.tn);
    const long per_cu = (tiles + n_cu - 1) / n_cu;
    const double cost = (double)per_cu * ((double)c.tm * c.tn + 60.0 * (c.tm + c.tn) + 3000.0);
    if (cost < best_cost) { best_cost = cost; best = c.id; }
  }
  return best;
}

int launch_gemm_h2_cfg(hipStream_t stream, int cfg, int epi, bool out3, const unsigned short* A3, int lda3, const unsigned short* Bt,
                         int ldb, int M, int N, int K, const float* bias, float* C, unsigned short* C3, int ldc3, float* D, int ldc,
                         const int* brow, int ldbias, int alt) {
  if (M <= 0) return 0;
  if (K % 32 != 0 || (lda3 % 64) || (ldb % 64) || (N & 3) || (ldc & 3) || (out3 && (ldc3 % 64 || (N & 31) || ldc3 < 2 * N)) ||
      (((size_t)A3 | (size_t)Bt | (size_t)bias | (size_t)C | (size_t)C3 | (size_t)D) & 15)) {
    set_last_error("gemm_h2: K=%d must be a multiple of 32, ldc/N multiples of 4, pointers 16-byte aligned, lda3/ldb/ldc3 whole 128-byte blocks, N %% 32 == 0 for split output", K);
    return -1;
  }
  if (cfg == 0) cfg = g_h2_force_tile;
  if (cfg == 0) cfg = choose_h2_tile(M, N);
  switch (cfg) {
#define AIMNET_H2_CASE(ID, SM_, SN_, WN_)                                                                                    \
    case ID: return launch_h2<SM_, SN_, WN_>(stream, epi, out3, A3, lda3, Bt, ldb, M, N, K, bias, C, C3, ldc3, D, ldc, brow, \
                                             ldbias, alt);
    AIMNET_H2_CASE(452, 5, 2, 4)  // 160 x 128 (2 x 4 waves of 80 x 32; 136 KiB of LDS)
    AIMNET_H2_CASE(432, 3, 2, 4)  //  96 x 128
    AIMNET_H2_CASE(422, 2, 2, 4)  //  64 x 128
    AIMNET_H2_CASE(223, 2, 3, 2)  // 128 x  96 (4 x 2 waves of 32 x 48)
    AIMNET_H2_CASE(224, 2, 4, 2)  // 128 x 128 (4 x 2 waves of 32 x 64)
    AIMNET_H2_CASE(234, 3, 4, 2)  // 192 x 128 (4 x 2 waves of 48 x 64)
    AIMNET_H2_CASE(851, 5, 1, 8)  //  80 x 128 (1 x 8 waves of 80 x 16)
#undef AIMNET_H2_CASE
    default:
      set_last_error("gemm_h2: unknown tile id %d", cfg);
      return -1;
  }
}

#ifdef AIMNET_BF3_TIMING
int gemm_h2_read_stamps(unsigned long long* host1024) {
  AIMNET_HIP_CHECK(hipMemcpyFromSymbol(host1024, HIP_SYMBOL(g_h2_stamps), 1024 * sizeof(unsigned long long)));
  return 0;
}
#endif

// ---- fp32 [M][ld] (K columns) -> h2 [M][Kp/32][2][32]; columns >= K of the last block are zero; mode: H2_PLAIN / H2_ACT / H2_WEIGHT
__global__ __launch_bounds__(256) void split_h2_kernel(const float* __restrict__ src, int ld, int M, int K, int Kp,
                                                       unsigned short* __restrict__ dst, int ldd, int mode, int* __restrict__ ovf) {
  const int q = Kp >> 2;  // column quads per row
  const size_t n = (size_t)M * q;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const int m = (int)(e / q), col = (int)(e % q) * 4;
    f32x4 v;
    const float* s = src + (size_t)m * ld + col;
    if (col + 3 < K && (((size_t)s) & 15) == 0) {
      v = *reinterpret_cast<const f32x4*>(s);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = col + r < K ? s[r] : 0.0f;
    }
    h2_flag_overflow(ovf, h2_amax4(v));
    const bool odd = (col & 32) != 0;
    unsigned h0, l0, h1, l1;
    const float sc = (mode == H2_ACT && odd) ? -H2_SCALE : H2_SCALE;
    split2_pair(v[0], v[1], sc, h0, l0);
    split2_pair(v[2], v[3], sc, h1, l1);
    if (mode == H2_WEIGHT && odd) {
      h0 ^= 0x80008000u;
      h1 ^= 0x80008000u;
    }
    unsigned short* p = dst + (size_t)m * ldd + (col >> 5) * 64 + (col & 31);
    *reinterpret_cast<u32x2*>(p) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(p + 32) = u32x2{l0, l1};
  }
}

int launch_split_h2(hipStream_t s, const float* src, int ld, int M, int K, unsigned short* dst, int ldd, int mode, int* ovf) {
  if (M <= 0 || K <= 0) return 0;
  const int Kp = (K + 31) / 32 * 32;
  const size_t n = (size_t)M * (Kp >> 2);
  const int blocks = (int)std::min<size_t>((n + 255) / 256, 8192);
  hipLaunchKernelGGL(split_h2_kernel, dim3(blocks), dim3(256), 0, s, src, ld, M, K, Kp, dst, ldd, mode, ovf);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

// host-side split of a weight matrix [rows][K] (K % 32 == 0) into the h2 layout (round to nearest even, like the device)
bool split_h2_host(const float* w, int rows, int K, unsigned short* out, int mode) {
  bool fits = true;
  for (int r = 0; r < rows; ++r)
    for (int k = 0; k < K; ++k) {
      const float x = w[(size_t)r * K + k];
      if (!(fabsf(x) < H2_MAX)) fits = false;
      const bool odd = ((k >> 5) & 1) != 0;
      const _Float16 h = (_Float16)x;
      const float res = (x - (float)h) * ((mode == H2_ACT && odd) ? -H2_SCALE : H2_SCALE);
      const _Float16 l = (_Float16)res;
      unsigned short hb, lb;
      memcpy(&hb, &h, 2);
      memcpy(&lb, &l, 2);
      if (mode == H2_WEIGHT && odd) hb ^= 0x8000u;
      unsigned short* o = out + (size_t)r * 2 * K + (k >> 5) * 64 + (k & 31);
      o[0] = hb;
      o[32] = lb;
    }
  return fits;
}


int gemm_h2_set_attributes() {
  const char* env = getenv("AIMNET_H2_TILE");
  g_h2_force_tile = env ? atoi(env) : 0;
  env = getenv("AIMNET_H2_DEEP");
  if (env) g_h2_deep = atoi(env) == 3 ? 3 : atoi(env) == 2 ? 2 : 4;
  return 0;
}

}  // namespace aimnet
