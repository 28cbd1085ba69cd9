// gemm_chain.hip - a whole MLP (forward or backward) as ONE launch: row-panel-stationary chained GEMMs on fp16x2-split operands.
//
//   aimnet/modules/core.py:11-46 (MLP = Linear, GELU, Linear, GELU, ..., Linear; call site aimnet/models/aimnet2.py:166) acts on every
//   atom's row independently, and so does its adjoint.  A block therefore owns a PANEL of 48 rows (three 16-row matrix-instruction
//   strips) and the FULL width of every layer: the panel's hidden activations never leave the CU.
//     * the first layer's input rows (h2 form, gemm_h2_common.h) are DMA'd into LDS once (<= 23 k-blocks x 6 KiB = 138 KiB); every
//       layer's output is split to the h2 form in the epilogue and written back into the same LDS buffer as the next layer's A
//       operand (in place: a barrier on either side of the write);
//     * the weights have exactly ONE consuming wave per column tile, so they do not go through LDS at all: wave w streams the tiles
//       t = w, w + 8, ... of a layer from L2 straight into registers, in a host-packed FRAGMENT ORDER (one 1 KiB wave load = one
//       16-column x 32-k plane of one tile, already in the lane order of v_mfma_f32_16x16x32_f16), through a two-k-step register ring
//       that runs across the layer seams (the next layer's first tiles arrive under the epilogue);
//     * products and accumulation exactly as gemm_h2.hip (ah bh into two interleaved accumulator sets by k-step parity, ah bl + al bh
//       into a third, scaled by 1 / 4096 in the epilogue; same order per accumulator): the results are BITWISE those of the per-layer
//       launches this kernel replaces (tests/test_gpu_chain.py);
//     * GELU' (forward) is stored / (backward) read in fp32 as before; the hidden activations themselves are never stored.
//   Cost model (config 3, 10 080 rows = 210 panels): a panel streams the chain's weights once (2.1 - 3.3 MB from L2 at <= 64 B/clk per
//   CU) against ~12 000 matrix instructions (17 clk each on four SIMDs): both ~20 - 25 us per chain; the per-layer launches took
//   62 - 84 us per chain (fill, epilogue store of 41 MB per layer, write-back, dispatch ramp - profiles/r5_gemm_h2.md).
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <vector>

#include "common.h"
#include "gemm_h2_common.h"
#include "kernels.h"

namespace aimnet {

namespace {
constexpr int CH_MAX_SM = 3;               // 16-row strips per panel: 3 (48 rows) when the batch fills the chip, fewer below
#ifndef CH_RING8
#define CH_RING8 4
#endif

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---- chain shapes: k-steps and tile slots per wave of every pass (a pass = one layer, or one column range of a wide layer).
// Static, so that the register ring, the accumulators and the loop trip counts are: the launcher matches the engine's layer sizes
// against this list (chain_find_shape) and the engine falls back to the per-layer launches when nothing fits.
template <int ID>
struct Shape;
#define AIMNET_CHAIN_SHAPE(ID, NW_, BWD_, NP_, NH_, ...)                  \
  template <>                                                             \
  struct Shape<ID> {                                                      \
    static constexpr int NW = NW_; /* waves per block (two per SIMD, <= 256 registers each) */ \
    static constexpr int R = CH_RING8; /* ring positions (items in flight per wave) */ \
    static constexpr bool BWD = BWD_;                                     \
    static constexpr int NP = NP_, NH = NH_; /* passes; passes that write the LDS operand (the others write global fp32) */ \
    static constexpr int V[2][CHAIN_MAX_PASS] = {__VA_ARGS__};            \
    static constexpr int nk(int i) { return V[0][i]; }                    \
    static constexpr int nt(int i) { return V[1][i]; }                    \
  };
//                 id NW bwd NP NH   k-steps                 tile slots per wave
AIMNET_CHAIN_SHAPE(0, 8, false, 3, 2, {14, 16, 12, 0, 0}, {4, 3, 3, 0, 0})      // pass 0 forward, embedding block folded into the bias table: 448 -> 512 -> 384 -> 288
AIMNET_CHAIN_SHAPE(1, 8, false, 3, 2, {23, 16, 12, 0, 0}, {4, 3, 3, 0, 0})      // pass 1 forward: 736 -> 512 -> 384 -> 288
AIMNET_CHAIN_SHAPE(2, 8, false, 4, 3, {23, 16, 12, 12, 0}, {4, 3, 3, 2, 0})     // pass 2 forward: 736 -> 512 -> 384 -> 384 -> 256
AIMNET_CHAIN_SHAPE(3, 8, true, 3, 2, {9, 12, 16, 0, 0}, {3, 4, 4, 0, 0})        // pass 0 backward (conv columns only): 288 -> 384 -> 512 -> 448
AIMNET_CHAIN_SHAPE(4, 8, true, 4, 2, {9, 12, 16, 16, 0}, {3, 4, 3, 3, 0})       // pass 1 backward: 288 -> 384 -> 512 -> 736 (two column passes)
AIMNET_CHAIN_SHAPE(5, 8, true, 5, 3, {8, 12, 12, 16, 16}, {3, 3, 4, 3, 3})      // pass 2 backward: 256 -> 384 -> 384 -> 512 -> 736 (two column passes)
#undef AIMNET_CHAIN_SHAPE
constexpr int N_SHAPES = 6;
constexpr int CH_MAX_NT = 4, CH_NTG = 2;  // tile slots per wave: of a pass, of a column group

template <int J, int N, class F>
__host__ __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (J < N) {
    f(std::integral_constant<int, J>{});
    static_for<J + 1, N>(f);
  }
}

// A pass runs as one or two COLUMN GROUPS (group A = the first nt / 2 tile slots of every wave when nt >= 3, group B = the rest): each
// group is a k-loop of its own over the whole operand, and the epilogue work of group A (GELU, GELU' stores / loads, output stores)
// runs as side jobs between the matrix instructions of group B's k-loop.  A "unit" is one group of one pass; the weight stream and the
// register ring run over the units in order.
__host__ __device__ constexpr int chain_nta(int nt) { return chain_group_a(nt); }
template <class S>
constexpr int n_units() {
  int n = 0;
  for (int i = 0; i < S::NP; ++i) n += chain_nta(S::nt(i)) ? 2 : 1;
  return n;
}
template <class S>
constexpr int unit_pass(int u) {
  for (int i = 0; i < S::NP; ++i) {
    const int k = chain_nta(S::nt(i)) ? 2 : 1;
    if (u < k) return i;
    u -= k;
  }
  return S::NP;
}
template <class S>
constexpr int unit_group(int u) {  // 0 = group A, 1 = group B (a pass without a group A has only group B)
  for (int i = 0; i < S::NP; ++i) {
    const int k = chain_nta(S::nt(i)) ? 2 : 1;
    if (u < k) return k == 2 ? u : 1;
    u -= k;
  }
  return 0;
}
template <class S>
constexpr int unit_nt(int u) {
  if (u >= n_units<S>()) return 1;
  const int nt = S::nt(unit_pass<S>(u)), na = chain_nta(nt);
  return unit_group<S>(u) == 0 ? na : nt - na;
}
template <class S>
constexpr int unit_nk(int u) { return u >= n_units<S>() ? 0 : S::nk(unit_pass<S>(u)); }
template <class S>
constexpr int items_before(int u0, int u1) {  // items of the units u0 .. u1 - 1
  int n = 0;
  for (int q = u0; q < u1 && q < n_units<S>(); ++q) n += unit_nk<S>(q) * unit_nt<S>(q);
  return n;
}
template <class S>
constexpr int item_unit(int u, int qi) {  // unit that holds item qi counted from the first item of unit u (n_units: beyond the stream)
  while (u < n_units<S>() && qi >= unit_nk<S>(u) * unit_nt<S>(u)) {
    qi -= unit_nk<S>(u) * unit_nt<S>(u);
    ++u;
  }
  return u;
}
}  // namespace

#ifdef CH_TIMING
__device__ unsigned long long g_chain_stamps[4 * 64];  // [block 0 / last block][wave 0 / wave 4][stamp]
#endif

template <class S, int CH_SM>
__global__ __launch_bounds__(64 * S::NW, 1) void gemm_chain_kernel(ChainArgs a) {
  constexpr int NW = S::NW, NTH = 64 * NW, NU = n_units<S>();
  constexpr int CH_KB = CH_SM * H2_STRIP;  // LDS bytes per k-block (32 columns) of the resident operand
  constexpr int CHAIN_ROWS = 16 * CH_SM;   // rows of the panel
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_c[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, lc = lane >> 4;
  const int m0 = blockIdx.x * CHAIN_ROWS;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_c;
#ifdef CH_TIMING
  int n_ts = 0;
  auto TS = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    if ((blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && (wid & 3) == 0 && n_ts < 64) {
      const unsigned long long t = __builtin_readcyclecounter();
      if (lane == 0) g_chain_stamps[((blockIdx.x != 0) * 2 + (wid >> 2)) * 64 + n_ts] = t;
      ++n_ts;
    }
    __builtin_amdgcn_sched_barrier(0);
  };
#else
  auto TS = [&]() __attribute__((always_inline)) {};
#endif
  TS();

  // ---- input panel: k-blocks of rows m0 .. m0 + 16 CH_SM - 1 -> LDS.  Piece q = plane (q & 1) of strip (q >> 1) % CH_SM of k-block q / (2 CH_SM) = 16 rows
  // x 64 B; lane -> row lane >> 2 of the strip, 16-byte chunk lane & 3, which sits in slot chunk ^ swz(row).  Rows beyond the matrix
  // re-read its last row (their results are never stored).  The first XC0 k-blocks arrive by LDS-DMA in front of the first product;
  // the rest in chunks of XCH k-blocks through registers (ordinary loads: the compiler's wait counts stay exact - with an LDS-DMA request
  // pending beside ordinary loads it treats the memory counter as out of order and waits with vmcnt(0) for every ring refill), requested
  // XCH k-steps ahead and written to LDS, with a barrier, one k-step in front of their first use.
  constexpr int NK0 = S::nk(0), XC0 = NK0 < 7 ? NK0 : 7, XCH = 8, PPK = 2 * CH_SM /* pieces per k-block */, XP = (PPK * XCH + NW - 1) / NW;
  const unsigned char* xb = reinterpret_cast<const unsigned char*>(a.x);
  auto piece_src = [&](int q) __attribute__((always_inline)) {
    const int kb = q / PPK, st = (q % PPK) >> 1, pl = q & 1;
    const int row = min(m0 + st * 16 + (lane >> 2), a.M - 1);
    return xb + (size_t)row * 2u * (unsigned)a.ldx + kb * H2_ROWB + pl * 64;
  };
  {
    const int r16 = lane >> 2;
    const unsigned in_row = (((lane & 3) ^ swz_h2(r16)) << 4);
#pragma unroll
    for (int q0 = 0; q0 < XC0 * PPK; q0 += NW) {
      const int q = q0 + wid;
      if (q < XC0 * PPK) glds16b(piece_src(q) + in_row, smem_c + (q / PPK) * CH_KB + ((q % PPK) >> 1) * H2_STRIP + (q & 1) * 1024);
    }
  }
  u32x4 xr[XP];  // pieces of the chunk in flight (this wave's)
  auto x_chunk_load = [&](auto kb0_c) __attribute__((always_inline)) {
    constexpr int KB0 = decltype(kb0_c)::value, KB1 = KB0 + XCH < NK0 ? KB0 + XCH : NK0;
#pragma unroll
    for (int e = 0; e < XP; ++e) {
      const int q = KB0 * PPK + e * NW + wid;
      if (q < KB1 * PPK) xr[e] = *reinterpret_cast<const u32x4*>(piece_src(q) + (lane & 3) * 16);
    }
  };
  auto x_chunk_store = [&](auto kb0_c) __attribute__((always_inline)) {
    constexpr int KB0 = decltype(kb0_c)::value, KB1 = KB0 + XCH < NK0 ? KB0 + XCH : NK0;
    const int r16 = lane >> 2;
#pragma unroll
    for (int e = 0; e < XP; ++e) {
      const int q = KB0 * PPK + e * NW + wid;
      if (q < KB1 * PPK)
        *reinterpret_cast<u32x4*>(smem_c + (q / PPK) * CH_KB + ((q % PPK) >> 1) * H2_STRIP + (q & 1) * 1024 + r16 * 64 + (((lane & 3) ^ swz_h2(r16)) << 4)) = xr[e];
    }
  };

  // ---- weight ring.  The chain's weights are ONE stream of items (unit, k-step, slot) = two 1 KiB planes of a 16-column tile, consumed
  // in order; item q sits in ring position q % R and its position is refilled with item q + R as soon as its products are issued
  // (everything is unrolled: positions are static).  The record of k-step j of a unit is [NW waves][NTG slots][2 planes] x 1 KiB.
  constexpr int R = S::R;
  f16x8 ring[R][2];
  // item qi (counted from the first item of unit U; it may lie in a later unit) -> ring position POS
  auto load_item = [&](auto u_c, auto qi_c, auto pos_c) __attribute__((always_inline)) {
    constexpr int U = decltype(u_c)::value, QI = decltype(qi_c)::value, POS = decltype(pos_c)::value;
    constexpr int UU = item_unit<S>(U, QI);
    if constexpr (UU < NU) {
      constexpr int Q = QI - items_before<S>(U, UU), NTG = unit_nt<S>(UU);
      const f16x8* p = reinterpret_cast<const f16x8*>(a.p[unit_pass<S>(UU)].w[unit_group<S>(UU)]) + (size_t)wid * NTG * 128 + lane +
                       (size_t)(Q / NTG) * (NW * NTG * 128) + (Q % NTG) * 128;
      ring[POS][0] = p[0];
      ring[POS][1] = p[64];
    }
  };

  f32x4 acc[3][CH_SM][CH_NTG];  // [0], [1]: ah bh of the even / odd k-steps; [2]: the cross terms (x 4096); [strip][slot of the group]
  f32x4 PA[CH_SM][CH_NTG];      // group A of the current pass: pre-activation -> activation (until the LDS write); GELU' of the previous pass' group B
  f32x4 PB[CH_SM][CH_NTG];      // group B: GELU' of group A (forward) / prefetched GELU' (backward); then the pre-activations of group B, whose
                                // GELU runs under the next pass' first k-loop
  f32x4 bb[CH_NTG];             // bias of the group's slots (layers with ONE bias vector)
  f16x8 fa[2][CH_SM][2];        // the operand fragments of this k-step and of the next one (requested in front of this one's products)
  const unsigned adA = lds0 + l16 * 64 + ((lc ^ swz_h2(l16)) << 4);

  using I0 = std::integral_constant<int, 0>;

  // geometry of tile (strip i, slot s of the pass): columns c0 .. c0 + 3 of row `row`
  auto col0 = [&](int s) __attribute__((always_inline)) { return 16 * (NW * s + wid) + 4 * lc; };
  auto row_of = [&](int i) __attribute__((always_inline)) { return m0 + 16 * i + l16; };

  // ---- prologue of the ring: the first R items
  static_for<0, R>([&](auto q_c) __attribute__((always_inline)) { load_item(I0{}, q_c, q_c); });
  TS();
  wait_vm<0>();  // the first k-blocks of the panel have landed (the DMA requests are older than the ring's)
  __builtin_amdgcn_sched_barrier(0);
  TS();
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  TS();

  // tile of slot s of a pass (values v[.][sg]) -> the LDS operand of the next pass, h2 form: columns 16 t .. 16 t + 15, t = NW s + wid
  // = half of k-block t >> 1
  auto to_lds_tile = [&](int s, const f32x4 (&v)[CH_SM][CH_NTG], int sg, int ncols) __attribute__((always_inline)) {
    const int t = NW * s + wid;
    if (16 * t >= ncols) return;
    const float sc = (t & 2) ? -H2_SCALE : H2_SCALE;  // lo planes of the ODD k-blocks negated (H2_ACT)
    const int kc = 2 * (t & 1) + (lc >> 1);
#pragma unroll
    for (int i = 0; i < CH_SM; ++i) {
      const unsigned ad = lds0 + (unsigned)(t >> 1) * CH_KB + i * H2_STRIP + l16 * 64 + ((kc ^ swz_h2(l16)) << 4) + (lc & 1) * 8;
      unsigned h0, l0, h1, l1;
      split2_pair(v[i][sg][0], v[i][sg][1], sc, h0, l0);
      split2_pair(v[i][sg][2], v[i][sg][3], sc, h1, l1);
      lds_write8<0>(ad, h0, h1);
      lds_write8<1024>(ad, l0, l1);
    }
  };
  // tiles v[.][0 .. n - 1] = slots s0 .. of a pass -> rows of a [M][ld] fp32 matrix (one burst of stores)
  auto store_tiles = [&](float* base, int ld, int ncols, int s0, int n, const f32x4 (&v)[CH_SM][CH_NTG]) __attribute__((always_inline)) {
#pragma unroll
    for (int sg = 0; sg < CH_NTG; ++sg) {
      if (sg >= n) continue;
      const int c0 = col0(s0 + sg);
#pragma unroll
      for (int i = 0; i < CH_SM; ++i) {
        const int row = row_of(i);
        if (c0 < ncols && row < a.M) *reinterpret_cast<f32x4*>(base + (size_t)row * ld + c0) = v[i][sg];
      }
    }
  };

  // ---- one unit (column group of a pass) ---------------------------------------------------------------------------------------
  auto run_unit = [&](auto u_c) __attribute__((always_inline)) {
    constexpr int U = decltype(u_c)::value, I = unit_pass<S>(U), G = unit_group<S>(U);
    constexpr int NK = S::nk(I), NT = S::nt(I), NTA = chain_nta(NT), NTB = NT - NTA;
    constexpr int NTG = G == 0 ? NTA : NTB, SG = G == 0 ? 0 : NTA;      // slots of this group, first slot of the group
    constexpr int Q0 = items_before<S>(0, U);                           // stream index of this unit's first item
    constexpr bool FIRST = G == 0 || NTA == 0;                          // first unit of its pass
    constexpr bool TO_LDS = I < S::NH;
    const ChainPass& P = a.p[I];
    static_assert(NK >= 3 && NTG >= 1 && NTG <= CH_NTG && NT <= CH_MAX_NT, "pass shape");

#pragma unroll
    for (int h = 0; h < 3; ++h)
#pragma unroll
      for (int i = 0; i < CH_SM; ++i)
#pragma unroll
        for (int s = 0; s < NTG; ++s) acc[h][i][s] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- side jobs of this k-loop (static schedule: job q runs behind the products of position q * NPOS / NJOB).  Stores are NOT side
    // jobs: the memory counter is in order, so a store in the loop holds every later ring refill's wait until it is acknowledged
    // (~1 000 clocks each, measured); they go out in bursts in front of barriers.
    // kind 1 (forward, first unit of a pass behind a hidden pass): PB holds the PRE-activations of the previous pass' group B; GELU element
    //         by element (PB in place, GELU' into PA); in front of k-step KS - the first that reads their k-blocks - GELU' is stored and
    //         the activations enter the LDS operand (group A's were written at the end of the previous pass)
    // kind 2 (forward, group B's loop): GELU of group A element by element (PA in place, GELU' into PB)
    // (backward: no jobs - GELU' of a group is requested as ONE batch half a k-loop / a whole k-loop ahead of the group's end, into PB:
    // it was written by the forward sweep long ago and comes from HBM; the memory counter is in order, so every slow load holds the
    // ring's waits behind it once - one batch per group instead of one tile at a time)
    constexpr int NT_PREV = I > 0 ? S::nt(I > 0 ? I - 1 : 0) : 0, NTA_PREV = chain_nta(NT_PREV), NTB_PREV = NT_PREV - NTA_PREV;
    constexpr bool GELU_PREV = !S::BWD && FIRST && I > 0 && I - 1 < S::NH && NTA_PREV > 0;
    constexpr int KIND = GELU_PREV ? 1 : (G == 1 && NTA > 0 && !S::BWD) ? 2 : 0;
    constexpr int KS = GELU_PREV ? NW * NTA_PREV / 2 : 0;
    static_assert(!GELU_PREV || (KS >= 3 && KS < NK), "group B of the previous pass must start behind k-block 2");
    constexpr int NJOB = KIND == 1 ? CH_SM * NTB_PREV * 4 : KIND == 2 ? CH_SM * NTA * 4 : 0;
    constexpr int NPOS = (GELU_PREV ? KS - 1 : NK - 1) * NTG;  // positions that take jobs
    const bool gelu_here = TO_LDS || (!S::BWD && P.epi != CH_BIAS_F32);  // this pass' outputs go through GELU
    auto job = [&](auto q_c) __attribute__((always_inline)) {
      constexpr int Q = decltype(q_c)::value;
      if constexpr (KIND == 1) {
        constexpr int n = Q / 4, r = Q % 4, i = n % CH_SM, sb = n / CH_SM;
        float hh, dd;
#ifdef CH_PROBE_NO_EPI
        hh = PB[i][sb][r], dd = 1.0f;
#else
        gelu_and_grad(PB[i][sb][r], hh, dd);
#endif
        PB[i][sb][r] = hh;
        PA[i][sb][r] = dd;
      } else if constexpr (KIND == 2) {
        constexpr int n = Q / 4, r = Q % 4, i = n % CH_SM, s = n / CH_SM;
        if (gelu_here) {
          float hh, dd;
#ifdef CH_PROBE_NO_EPI
          hh = PA[i][s][r], dd = 1.0f;
#else
          gelu_and_grad(PA[i][s][r], hh, dd);
#endif
          PA[i][s][r] = hh;
          PB[i][s][r] = dd;
        }
      }
    };
    // loads that the end of this k-loop consumes, requested behind slot SL of the last k-step: the bias of the slot (forward, layers with
    // one bias vector), GELU' of group B (backward hidden: into PB)
    auto prefetch = [&](auto sl_c) __attribute__((always_inline)) {
      constexpr int SL = decltype(sl_c)::value;
      const int cc = min(col0(SG + SL), P.ncols - 4);
      if constexpr (!S::BWD) {
        if (!P.brow) bb[SL] = *reinterpret_cast<const f32x4*>(P.bias + cc);
      }
    };
    // backward hidden passes: GELU' of this group's tiles -> PB, one batch
    auto prefetch_gelu_grad = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int sl = 0; sl < NTG; ++sl) {
        const int cc = min(col0(SG + sl), P.ncols - 4);
#pragma unroll
        for (int i = 0; i < CH_SM; ++i)
          PB[i][sl] = *reinterpret_cast<const f32x4*>(P.D + (size_t)min(row_of(i), a.M - 1) * P.ldd + cc);
      }
    };
    constexpr int J_DGRAD = (S::BWD && TO_LDS) ? (G == 0 ? NK / 2 : 0) : -1;  // k-step in front of which that batch is requested
    // product sums of tile (i, s of the group) (+ bias, forward)
    auto combined = [&](int i, int s) __attribute__((always_inline)) -> f32x4 {
      f32x4 v = (acc[0][i][s] - acc[1][i][s]) + acc[2][i][s] * H2_INV_SCALE;
      if constexpr (!S::BWD) {
        if (P.brow) {  // the [64][ldbias] table of the embedding-bias layer: the row of this atom's element
          const int cc = min(col0(SG + s), P.ncols - 4), rowc = min(row_of(i), a.M - 1);
          v = v + *reinterpret_cast<const f32x4*>(P.bias + (size_t)min(63, max(0, P.brow[rowc])) * P.ldbias + cc);
        } else {
          v = v + bb[s];
        }
      }
      return v;
    };

    unsigned a_base = adA + (unsigned)P.kb0 * CH_KB;
    asm volatile("" : "+v"(a_base));  // (opaque per unit: the compiler otherwise keeps every k-step's address alive for the next unit - spills)
    // ---- the k-loop, fully unrolled (NK is static; in a rolled loop the compiler's wait-count pass also joins the back edge with a
    // vmcnt(0) in front of the first product - it would wait for the refills it has just issued)
    static_for<0, NK>([&](auto j_c) __attribute__((always_inline)) {
      constexpr int J = decltype(j_c)::value, PAR = J & 1;  // PAR: the accumulator set of this k-step
      if constexpr (J == 2 || J == NK / 2) TS();
      if constexpr (U == 0 && J + 1 >= XC0 && J + 1 < NK0 && (J + 1 - XC0) % XCH == 0) {  // the chunk that k-step J + 1 opens: registers -> LDS
        x_chunk_store(std::integral_constant<int, J + 1>{});
        wait_lgkm<0>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (U == 0 && XC0 < NK0 && J < NK0 - XC0 && (J == 0 || (J + 1 - XC0) % XCH == 0)) {  // request the next chunk
        constexpr int KBN = J == 0 ? XC0 : J + 1 + XCH;
        if constexpr (KBN < NK0) x_chunk_load(std::integral_constant<int, KBN>{});
      }
      if constexpr (GELU_PREV && J == KS - 1) {  // group B of the previous pass: GELU' out, activations into the operand (first read at k-step KS)
        const ChainPass& PP = a.p[I > 0 ? I - 1 : 0];
        if (PP.D) store_tiles(PP.D, PP.ldd, PP.ncols, NTA_PREV, NTB_PREV, PA);
#pragma unroll
        for (int sb = 0; sb < NTB_PREV; ++sb) to_lds_tile(NTA_PREV + sb, PB, sb, PP.ncols);
        wait_lgkm<0>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (J == J_DGRAD) {
        prefetch_gelu_grad();
        __builtin_amdgcn_sched_barrier(0);
      }
      // fragments: this k-step's were requested one step ahead; request the next step's in front of this step's products
#ifndef CH_PROBE_NO_LDSREAD
      if constexpr (J == 0) {
        read_strips_h<0, CH_SM, 0>(fa[0], a_base);
        read_strips_h<0, CH_SM, 1>(fa[0], a_base);
      }
      wait_lgkm<0>();
      if constexpr (J + 1 < NK) {
        const unsigned oa = a_base + (unsigned)(J + 1) * CH_KB;
        read_strips_h<0, CH_SM, 0>(fa[(J + 1) & 1], oa);
        read_strips_h<0, CH_SM, 1>(fa[(J + 1) & 1], oa);
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, NTG>([&](auto sl_c) __attribute__((always_inline)) {
        constexpr int SL = decltype(sl_c)::value, QI = J * NTG + SL, POS = (Q0 + QI) % R;
#pragma unroll
        for (int i = 0; i < CH_SM; ++i)
          acc[2][i][SL] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[POS][1], fa[J & 1][i][0], acc[2][i][SL], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < CH_SM; ++i)
          acc[PAR][i][SL] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[POS][0], fa[J & 1][i][0], acc[PAR][i][SL], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < CH_SM; ++i)
          acc[2][i][SL] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[POS][0], fa[J & 1][i][1], acc[2][i][SL], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#ifndef CH_PROBE_NO_LOAD
        load_item(u_c, std::integral_constant<int, QI + R>{}, std::integral_constant<int, POS>{});
#endif
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (J < NK - 1) {
          if constexpr (NJOB > 0 && QI < NPOS) {
            constexpr int Q_LO = QI * NJOB / NPOS, Q_HI = (QI + 1) * NJOB / NPOS;
            static_for<Q_LO, Q_HI>(job);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
          prefetch(sl_c);
          __builtin_amdgcn_sched_barrier(0);
        }
      });
    });
    TS();

    // ---- end of the k-loop.  Tile of slot s: t = NW s + wid, columns 16 t + 4 lc .. + 3 (relative to the pass), rows 16 i + l16.
    const int epi = P.epi;
    if constexpr (G == 0) {  // group A: the pre-activations wait in PA for the side jobs of group B's k-loop
#pragma unroll
      for (int s = 0; s < NTG; ++s)
#pragma unroll
        for (int i = 0; i < CH_SM; ++i) {
          PA[i][s] = combined(i, s);
          if constexpr (S::BWD && TO_LDS) PA[i][s] = PA[i][s] * PB[i][s];  // x GELU'(z) of the layer below (requested half a k-loop ago)
          asm volatile("" : "+v"(PA[i][s]));  // (formed HERE: the compiler otherwise sinks this into the side jobs and keeps - spills - the raw accumulators)
        }
    } else if constexpr (!S::BWD && TO_LDS && NTA > 0) {
      // forward hidden pass: GELU' of group A out; group B's pre-activations wait in PB for the next pass' first k-loop; group A's
      // activations enter the operand
      if (P.D) store_tiles(P.D, P.ldd, P.ncols, 0, NTA, PB);
#pragma unroll
      for (int s = 0; s < NTG; ++s)
#pragma unroll
        for (int i = 0; i < CH_SM; ++i) {
          PB[i][s] = combined(i, s);
          asm volatile("" : "+v"(PB[i][s]));
        }
      TS();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();  // every wave has read the operand this overwrites
      __builtin_amdgcn_sched_barrier(0);
      TS();
#pragma unroll
      for (int s = 0; s < NTA; ++s) to_lds_tile(s, PA, s, P.ncols);
      wait_lgkm<0>();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      TS();
    } else {
      f32x4 HB[CH_SM][CH_NTG];  // what group B hands to the LDS operand
      if constexpr (!S::BWD && NTA > 0) {  // last forward pass: group A's GELU' and outputs
        if (gelu_here && P.D) store_tiles(P.D, P.ldd, P.ncols, 0, NTA, PB);
        if (epi != CH_GELU_H2G) store_tiles(P.C, P.ldc, P.ncols, 0, NTA, PA);
      }
      if constexpr (S::BWD && !TO_LDS && NTA > 0) store_tiles(P.C, P.ldc, P.ncols, 0, NTA, PA);
#pragma unroll
      for (int s = 0; s < NTG; ++s) {
        const int c0 = col0(SG + s);
        const bool col_ok = c0 < P.ncols;
#pragma unroll
        for (int i = 0; i < CH_SM; ++i) {
          const int row = row_of(i);
          const bool ok = col_ok && row < a.M;
          f32x4 v = combined(i, s);
          if constexpr (!S::BWD) {
            if (gelu_here) {
              f32x4 d;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                float hh, dd;
#ifdef CH_PROBE_NO_EPI
                hh = v[r], dd = 1.0f;
#else
                gelu_and_grad(v[r], hh, dd);
#endif
                v[r] = hh;
                d[r] = dd;
              }
              if (ok && P.D) *reinterpret_cast<f32x4*>(P.D + (size_t)row * P.ldd + c0) = d;
            }
            if constexpr (!TO_LDS)
              if (ok && epi != CH_GELU_H2G) *reinterpret_cast<f32x4*>(P.C + (size_t)row * P.ldc + c0) = v;
            HB[i][s] = v;
          } else if constexpr (TO_LDS) {
            HB[i][s] = v * PB[i][s];
          } else {
            if (ok) *reinterpret_cast<f32x4*>(P.C + (size_t)row * P.ldc + c0) = v;
          }
        }
      }
      TS();
      const bool to_lds = TO_LDS || (!S::BWD && epi == CH_GELU_H2G);
      if (to_lds) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();  // every wave has read the operand this overwrites
        __builtin_amdgcn_sched_barrier(0);
        TS();
#pragma unroll
        for (int s = 0; s < NTA; ++s) to_lds_tile(s, PA, s, P.ncols);
#pragma unroll
        for (int s = 0; s < NTB; ++s) to_lds_tile(NTA + s, HB, s, P.ncols);
        wait_lgkm<0>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        TS();
      }
      if constexpr (!S::BWD && !TO_LDS) {
        if (epi == CH_GELU_H2G) {
          // the panel of the last layer's output, h2 form, LDS -> memory: a wave moves one row's 16-byte granules per instruction
          // (unit u of a row = k-block u >> 3, plane (u >> 2) & 1, chunk u & 3; a row is 4 * ncols contiguous bytes in memory)
          const int units = P.ncols >> 2;  // 16-byte granules per row
          for (int e = tid; e < CHAIN_ROWS * units; e += NTH) {
            const int r = e / units, u = e - r * units;
            if (m0 + r >= a.M) continue;
            const int kb = u >> 3, pl = (u >> 2) & 1, c = u & 3, r16 = r & 15;
            const u32x4 g = *reinterpret_cast<const u32x4*>(smem_c + kb * CH_KB + (r >> 4) * H2_STRIP + pl * 1024 + r16 * 64 + ((c ^ swz_h2(r16)) << 4));
            *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(P.C2) + (size_t)(m0 + r) * 2u * (unsigned)P.ldc2 + (size_t)u * 16) = g;
          }
        }
      }
    }
  };
  static_for<0, NU>([&](auto u_c) __attribute__((always_inline)) { run_unit(u_c); });
}

// ---- host side ----------------------------------------------------------------------------------------------------------------
int chain_find_shape(int n_waves, bool bwd, int n_pass, const int* nk, const int* nt, int n_hidden) {
  auto match = [&](auto s_c) -> bool {
    using S = Shape<decltype(s_c)::value>;
    if (S::NW != n_waves || S::BWD != bwd || S::NP != n_pass || S::NH != n_hidden) return false;
    for (int i = 0; i < n_pass; ++i)
      if (S::nk(i) != nk[i] || S::nt(i) != nt[i]) return false;
    return true;
  };
  int found = -1;
  static_for<0, N_SHAPES>([&](auto s_c) {
    if (found < 0 && match(s_c)) found = decltype(s_c)::value;
  });
  if (found >= 0) return found;
  return -1;
}

// Weight stream of one pass: output columns (= rows of the h2 weight matrix `w2`, [rows][2 * ldk] 16-bit elements, H2_WEIGHT form)
// n0 .. n0 + 16 * nw * nt - 1 (rows >= n_rows: zeros), k-blocks kb0 .. kb0 + nk - 1, in the order the kernel consumes it.
void chain_pack_weights(const unsigned short* w2, int n_rows, int ldk, int n0, int nw, int nt, int kb0, int nk, std::vector<unsigned short>& out) {
  out.assign((size_t)nk * nw * nt * 2 * 64 * 8, 0);
  for (int j = 0; j < nk; ++j)
    for (int w = 0; w < nw; ++w)
      for (int s = 0; s < nt; ++s)
        for (int pl = 0; pl < 2; ++pl)
          for (int lane = 0; lane < 64; ++lane) {
            const int n = n0 + 16 * (nw * s + w) + (lane & 15);
            if (n >= n_rows) continue;
            const unsigned short* src = w2 + (size_t)n * 2 * ldk + (size_t)(kb0 + j) * 64 + pl * 32 + (lane >> 4) * 8;
            unsigned short* dst = out.data() + ((((size_t)j * nw + w) * nt + s) * 2 + pl) * 512 + (size_t)lane * 8;
            memcpy(dst, src, 16);
          }
}

#ifdef CH_TIMING
int gemm_chain_read_stamps(unsigned long long* host256) {
  AIMNET_HIP_CHECK(hipMemcpyFromSymbol(host256, HIP_SYMBOL(g_chain_stamps), 256 * sizeof(unsigned long long)));
  return 0;
}
extern "C" int aimnet_debug_chain_stamps(unsigned long long* host256) { return gemm_chain_read_stamps(host256); }
#endif

// strips per panel: the fewest that keep the grid inside one round of the chip (a block streams the chain's weights once whatever its
// rows: below ~200 panels of 48 rows, narrower panels on more CUs are faster), 3 beyond
int chain_strips(int M) {
  const int cus = device_cus();
  for (int sm = 1; sm < CH_MAX_SM; ++sm)
    if (ceil_div(M, 16 * sm) <= cus) return sm;
  return CH_MAX_SM;
}
static int g_chain_sm = 0;  // AIMNET_CHAIN_SM forces the strip count (A/B runs)

int launch_gemm_chain(hipStream_t stream, int shape, const ChainArgs& a) {
  if (a.M <= 0) return 0;
  if (shape < 0 || shape >= N_SHAPES || (a.ldx % 64) || (((size_t)a.x) & 15)) {
    set_last_error("gemm_chain: bad shape id %d / operand alignment", shape);
    return -1;
  }
  static const bool env_read = [] {
    const char* env = getenv("AIMNET_CHAIN_SM");
    if (env) g_chain_sm = atoi(env);
    return true;
  }();
  (void)env_read;
  const int sm = (g_chain_sm >= 1 && g_chain_sm <= CH_MAX_SM) ? g_chain_sm : chain_strips(a.M);
  int rc = 0;
  static_for<0, N_SHAPES * CH_MAX_SM>([&](auto c_c) {
    constexpr int ID = decltype(c_c)::value / CH_MAX_SM, SM = decltype(c_c)::value % CH_MAX_SM + 1;
    if (ID != shape || SM != sm) return;
    static PerDeviceOnce once;
    if (once.first() && hipFuncSetAttribute((const void*)gemm_chain_kernel<Shape<ID>, SM>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            160 * 1024) != hipSuccess) {
      set_last_error("gemm_chain: cannot raise the dynamic LDS limit");
      rc = -2;
      return;
    }
    hipLaunchKernelGGL((gemm_chain_kernel<Shape<ID>, SM>), dim3(ceil_div(a.M, 16 * SM)), dim3(64 * Shape<ID>::NW), CHAIN_MAX_KB * SM * H2_STRIP,
                       stream, a);
  });
  if (rc) return rc;
  AIMNET_LAUNCH_CHECK();
  return 0;
}

}  // namespace aimnet
