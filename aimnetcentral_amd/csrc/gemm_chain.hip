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
constexpr int CH_SM = 3;                   // 16-row strips per panel
constexpr int CH_KB = CH_SM * H2_STRIP;    // LDS bytes per k-block (32 columns) of the resident operand
constexpr int CH_LDS = CHAIN_MAX_KB * CH_KB;
#ifndef CH_RING8
#define CH_RING8 6
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
    static constexpr int NW = NW_; /* waves per block: 8 (two per SIMD, <= 256 registers) or 4 (one per SIMD, <= 512) */ \
    static constexpr int R = NW_ == 8 ? CH_RING8 : 16; /* ring positions (items in flight per wave) */ \
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
constexpr int CH_MAX_NT = 4;

template <int J, int N, class F>
__host__ __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (J < N) {
    f(std::integral_constant<int, J>{});
    static_for<J + 1, N>(f);
  }
}

template <class S>
constexpr int items_before(int i0, int i1) {  // items of the passes i0 .. i1 - 1
  int n = 0;
  for (int q = i0; q < i1 && q < S::NP; ++q) n += S::nk(q) * S::nt(q);
  return n;
}
template <class S>
constexpr int item_pass(int i, int qi) {  // pass that holds item qi counted from the first item of pass i (NP: beyond the stream)
  while (i < S::NP && qi >= S::nk(i) * S::nt(i)) {
    qi -= S::nk(i) * S::nt(i);
    ++i;
  }
  return i;
}
}  // namespace

#ifdef CH_TIMING
__device__ unsigned long long g_chain_stamps[4 * 64];  // [block 0 / last block][wave 0 / wave 4][stamp]
#endif

template <class S>
__global__ __launch_bounds__(64 * S::NW, 1) void gemm_chain_kernel(ChainArgs a) {
  constexpr int NW = S::NW, NTH = 64 * NW;
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

  // ---- input panel: k-blocks 0 .. nk0 - 1 of rows m0 .. m0 + 47 -> LDS.  Piece q = plane (q & 1) of strip (q >> 1) % 3 of
  // k-block (q / 6); wave w takes the pieces q = w, w + NW, ...; lane -> row lane >> 2 of the strip, slot lane & 3 holding k-chunk
  // slot ^ swz(row).  Rows beyond the matrix re-read its last row (their results are never stored).
  {
    constexpr int NK0 = S::nk(0);
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(a.x);
    const int r16 = lane >> 2;
    const unsigned in_row = (((lane & 3) ^ swz_h2(r16)) << 4);
#pragma unroll
    for (int q0 = 0; q0 < NK0 * 6; q0 += NW) {
      const int q = q0 + wid;
      if (q < NK0 * 6) {
        const int kb = q / 6, st = (q % 6) >> 1, pl = q & 1;
        const int row = min(m0 + st * 16 + r16, a.M - 1);
        glds16b(xb + (size_t)row * 2u * (unsigned)a.ldx + kb * H2_ROWB + pl * 64 + in_row, smem_c + kb * CH_KB + st * H2_STRIP + pl * 1024);
      }
    }
  }

  // ---- weight ring.  The chain's weights are ONE stream of items (pass, k-step, slot) = two 1 KiB planes of a 16-column tile, consumed
  // in order; item q sits in ring position q % R and its position is refilled with item q + R as soon as its products are issued
  // (everything is unrolled: positions are static).  The record of k-step j of a pass is [NW waves][NT slots][2 planes] x 1 KiB.
  constexpr int R = S::R;
  f16x8 ring[R][2];
  auto w_base = [&](const void* w, int nt) __attribute__((always_inline)) {
    return reinterpret_cast<const f16x8*>(w) + (size_t)wid * nt * 128 + lane;
  };
  // item qi (counted from the first item of pass I; it may lie in a later pass) -> ring position POS
  auto load_item = [&](auto i_c, auto qi_c, auto pos_c) __attribute__((always_inline)) {
    constexpr int I = decltype(i_c)::value, QI = decltype(qi_c)::value, POS = decltype(pos_c)::value;
    constexpr int II = item_pass<S>(I, QI);
    if constexpr (II < S::NP) {
      constexpr int Q = QI - items_before<S>(I, II), NT = S::nt(II < S::NP ? II : 0);
      const f16x8* p = w_base(a.p[II < S::NP ? II : 0].w, NT) + (size_t)(Q / NT) * (NW * NT * 128) + (Q % NT) * 128;
      ring[POS][0] = p[0];
      ring[POS][1] = p[64];
    }
  };

  f32x4 acc[3][CH_SM][CH_MAX_NT];  // [0], [1]: ah bh of the even / odd k-steps; [2]: the cross terms (x 4096); [strip][slot]
  f16x8 fa[CH_SM][2];
  const unsigned adA = lds0 + l16 * 64 + ((lc ^ swz_h2(l16)) << 4);

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // ---- prologue of the ring: the first R items
  static_for<0, R>([&](auto q_c) __attribute__((always_inline)) { load_item(I0{}, q_c, q_c); });
  TS();
  wait_vm<0>();  // the panel has landed (the DMA requests are older than the ring's)
  __builtin_amdgcn_sched_barrier(0);
  TS();
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  TS();

  // ---- one pass ------------------------------------------------------------------------------------------------------------
  auto run_pass = [&](auto i_c) __attribute__((always_inline)) {
    constexpr int I = decltype(i_c)::value;
    constexpr int NK = S::nk(I), NT = S::nt(I), Q0 = items_before<S>(0, I);  // Q0: stream index of this pass' first item
    const ChainPass& P = a.p[I];
    static_assert(NK >= 2 && NT >= 1 && NT <= CH_MAX_NT, "pass shape");

#pragma unroll
    for (int h = 0; h < 3; ++h)
#pragma unroll
      for (int i = 0; i < CH_SM; ++i)
#pragma unroll
        for (int s = 0; s < NT; ++s) acc[h][i][s] = f32x4{0.f, 0.f, 0.f, 0.f};

    // the k-loop is fully unrolled (NK is static; in a rolled loop the compiler's wait-count pass also joins the back edge with a
    // vmcnt(0) in front of the first product - it would wait for the refills it has just issued)
    static_for<0, NK>([&](auto j_c) __attribute__((always_inline)) {
      constexpr int J = decltype(j_c)::value, PAR = J & 1;  // PAR: the accumulator set of this k-step
      if constexpr (J == 2 || J == NK / 2) TS();
      const unsigned oa = adA + (unsigned)(P.kb0 + J) * CH_KB;
#ifdef CH_PROBE_NO_LDSREAD
      if (J == 0) {
#endif
      read_strips_h<0, CH_SM, 0>(fa, oa);
      read_strips_h<0, CH_SM, 1>(fa, oa);
      wait_lgkm<0>();
#ifdef CH_PROBE_NO_LDSREAD
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, NT>([&](auto sl_c) __attribute__((always_inline)) {
        constexpr int SL = decltype(sl_c)::value, QI = J * NT + SL, POS = (Q0 + QI) % R;
#ifdef CH_PROBE_NO_MFMA
        acc[2][0][SL][0] += (float)ring[POS][0][0] + (float)ring[POS][1][0] + (float)fa[0][0][0] + (float)fa[1][1][0] + (float)fa[2][0][0];
#else
#pragma unroll
        for (int i = 0; i < CH_SM; ++i)
          acc[2][i][SL] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[POS][1], fa[i][0], acc[2][i][SL], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < CH_SM; ++i)
          acc[PAR][i][SL] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[POS][0], fa[i][0], acc[PAR][i][SL], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < CH_SM; ++i)
          acc[2][i][SL] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[POS][0], fa[i][1], acc[2][i][SL], 0, 0, 0);
#endif
        __builtin_amdgcn_sched_barrier(0);
#ifndef CH_PROBE_NO_LOAD
        load_item(i_c, std::integral_constant<int, QI + R>{}, std::integral_constant<int, POS>{});
#endif
        __builtin_amdgcn_sched_barrier(0);
      });
    });

    TS();
    // ---- epilogue.  Tile of slot s: t = NW s + wid, columns c0 = 16 t + 4 lc .. + 3 (relative to the pass), rows 16 i + l16.
    constexpr bool TO_LDS = I < S::NH;
    const int epi = P.epi;
    f32x4 val[CH_SM][CH_MAX_NT];
    // (every load of the epilogue is unconditional, at a clamped address, so that the compiler can issue them together and wait once)
    int rowc[CH_SM];
    const float* brp[CH_SM];  // forward: the bias row of each strip's row (the [64][ldbias] table of the embedding-bias layer, or the one bias)
#pragma unroll
    for (int i = 0; i < CH_SM; ++i) {
      rowc[i] = min(m0 + 16 * i + l16, a.M - 1);
      if constexpr (!S::BWD) brp[i] = P.bias + (P.brow ? (size_t)min(63, max(0, P.brow[rowc[i]])) * P.ldbias : (size_t)0);
    }
#pragma unroll
    for (int s = 0; s < NT; ++s) {
      const int c0 = 16 * (NW * s + wid) + 4 * lc;  // (ncols is a multiple of 16: the lanes of a wave agree on c0 < ncols)
      const int cc = min(c0, P.ncols - 4);
      const bool col_ok = c0 < P.ncols;
#pragma unroll
      for (int i = 0; i < CH_SM; ++i) {
        const int row = m0 + 16 * i + l16;
        const bool ok = col_ok && row < a.M;
        f32x4 v = (acc[0][i][s] - acc[1][i][s]) + acc[2][i][s] * H2_INV_SCALE;
        if constexpr (!S::BWD) {
          v = v + *reinterpret_cast<const f32x4*>(brp[i] + cc);
          if (TO_LDS || epi != CH_BIAS_F32) {
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
        } else if constexpr (TO_LDS) {
          v = v * *reinterpret_cast<const f32x4*>(P.D + (size_t)rowc[i] * P.ldd + cc);
        }
        if constexpr (!TO_LDS) {
          if (ok && (S::BWD || epi != CH_GELU_H2G)) *reinterpret_cast<f32x4*>(P.C + (size_t)row * P.ldc + c0) = v;
        }
        val[i][s] = v;
      }
    }
    TS();
    const bool to_lds = TO_LDS || (!S::BWD && epi == CH_GELU_H2G);
    if (to_lds) {
      // every wave has read the operand this overwrites
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      TS();
#pragma unroll
      for (int s = 0; s < NT; ++s) {
        const int t = NW * s + wid;  // tile = columns 16 t .. 16 t + 15 of the pass = half of k-block t >> 1 of the next operand
        if (16 * t >= P.ncols) continue;
        const float sc = (t & 2) ? -H2_SCALE : H2_SCALE;  // lo planes of the ODD k-blocks negated (H2_ACT)
        const int kc = 2 * (t & 1) + (lc >> 1);
#pragma unroll
        for (int i = 0; i < CH_SM; ++i) {
          const unsigned ad = lds0 + (unsigned)(t >> 1) * CH_KB + i * H2_STRIP + l16 * 64 + ((kc ^ swz_h2(l16)) << 4) + (lc & 1) * 8;
          unsigned h0, l0, h1, l1;
          split2_pair(val[i][s][0], val[i][s][1], sc, h0, l0);
          split2_pair(val[i][s][2], val[i][s][3], sc, h1, l1);
          lds_write8<0>(ad, h0, h1);
          lds_write8<1024>(ad, l0, l1);
        }
      }
      wait_lgkm<0>();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      TS();
    }
    if constexpr (!S::BWD) {
      if (TO_LDS ? (P.C2 != nullptr) : (epi == CH_GELU_H2G)) {  // (hidden passes: debug dump of the LDS operand, AIMNET_CHAIN_DUMP)
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
  };
  static_for<0, S::NP>([&](auto i_c) __attribute__((always_inline)) { run_pass(i_c); });
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

int launch_gemm_chain(hipStream_t stream, int shape, const ChainArgs& a) {
  if (a.M <= 0) return 0;
  if (shape < 0 || shape >= N_SHAPES || (a.ldx % 64) || (((size_t)a.x) & 15)) {
    set_last_error("gemm_chain: bad shape id %d / operand alignment", shape);
    return -1;
  }
  const dim3 grid(ceil_div(a.M, CHAIN_ROWS));
  int rc = 0;
  static_for<0, N_SHAPES>([&](auto s_c) {
    constexpr int ID = decltype(s_c)::value;
    if (ID != shape) return;
    static PerDeviceOnce once;
    if (once.first() && hipFuncSetAttribute((const void*)gemm_chain_kernel<Shape<ID>>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            160 * 1024) != hipSuccess) {
      set_last_error("gemm_chain: cannot raise the dynamic LDS limit");
      rc = -2;
      return;
    }
    hipLaunchKernelGGL((gemm_chain_kernel<Shape<ID>>), grid, dim3(64 * Shape<ID>::NW), CH_LDS, stream, a);
  });
  if (rc) return rc;
  AIMNET_LAUNCH_CHECK();
  return 0;
}

}  // namespace aimnet
