// gemm_h2_chain.hip - the layers of one MLP (forward, or backward) as ONE persistent launch of the fp16x2-split GEMM.
//
//   layer l:  C_l = epilogue_l(A_l . W_l^T),  A_{l+1} = C_l (h2 form)   - the chain the engine otherwise runs as one gemm_h2_kernel
//   launch per layer (aimnet/modules/core.py:11-46 forward; the input-gradient products of its autograd backward).
//
// Why (profiles/r5_gemm_h2.md): a large layer is ~15 us of main loop inside 26 - 32 us - cold first tiles, dispatch ramp, and the
// write-back of the 40 MB the layer leaves dirty when its grid retires; the loop itself runs at the matrix instruction's rate.
// Here the tiles of ALL layers of the chain are one list (layer-major, row-panel-major inside a layer) that the resident blocks
// draw from a device queue in order:
//   * a tile of layer l > 0 needs the row panel of layer l-1 complete: every finished tile adds 1 to its panel's arrival counter
//     AFTER its stores are acknowledged (the h2 activations leave with WRITE-THROUGH stores: sc0 sc1, cdna_hip_programming.md
//     Guideline 16 form R1), consumers poll that one word relaxed and read the panel with sc1 requests;
//   * tiles are drawn in list order and depend only on EARLIER tiles, which running blocks hold: whatever the residency (a grid
//     larger than the free CUs, another process on the device), some block can always finish - no co-residency assumption, no
//     grid-wide barrier; every spin is bounded all the same (error word);
//   * per tile: the matrix work of gemm_h2.hip's ping-pong schedule, instruction for instruction - results are BITWISE those of the
//     per-layer launches (tests/test_gpu_ops.py, tests/tools/h2_chain_bench.py).
// The sync words (queue head, arrival counters, error word) are zeroed by a memset node in front of the launch.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "common.h"
#include "gemm_h2_common.h"
#include "kernels.h"

namespace aimnet {

namespace {
constexpr int chain_passes(int rows) { return (rows + 31) / 32; }
constexpr int chain_ring_bytes(int TM, int TN) { return 2 * chain_passes(TM) * 4096 + 3 * chain_passes(TN) * 4096; }

__device__ __forceinline__ void glds16b_sc1(const void* g, void* lds_wave_base) {  // L2-served request (bypasses this CU's L1)
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 16);
}
}  // namespace

template <int SM, int SN, int WN>
__global__ __launch_bounds__(512, 2) void gemm_h2_chain_kernel(H2ChainArgs a) {
  static_assert(WN == 8 || WN == 4 || WN == 2, "waves across N");
  static_assert((SN & 1) == 0, "tile pairs (the split-output epilogue stores two column tiles at a time)");
  constexpr int WM = 8 / WN;
  constexpr int TM = 16 * SM * WM, TN = 16 * SN * WN;
  constexpr int NPA = chain_passes(TM), NPB = chain_passes(TN);
  constexpr int SA_BYTES = NPA * 4096, SB_BYTES = NPB * 4096;
  constexpr int RING = 2 * SA_BYTES + 3 * SB_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_a[];
  volatile int* s_ctl = reinterpret_cast<volatile int*>(smem_a + RING);  // [0] the next tile of this block

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const bool late = wid >= 4;  // group 1 runs one segment behind group 0
  const int w4 = wid & 3;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_a;
  const unsigned ldsB = lds0 + 2 * SA_BYTES;
  const int l16 = lane & 15, lc = lane >> 4;
  const unsigned adA = lds0 + wm * SM * H2_STRIP + l16 * 64 + ((lc ^ swz_h2(l16)) << 4);
  const unsigned adB = ldsB + wn * SN * H2_STRIP + l16 * 64 + ((lc ^ swz_h2(l16)) << 4);
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  if (tid == 0) s_ctl[0] = (int)__hip_atomic_fetch_add(a.queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  int t = __builtin_amdgcn_readfirstlane(s_ctl[0]);

  while (t < a.n_tiles) {
    int l = 0;
    for (int k = 1; k < a.n_layers; ++k)
      if (t >= a.L[k].tile0) l = k;
    l = __builtin_amdgcn_readfirstlane(l);
    const H2ChainLayer& Ly = a.L[l];
    const int tiles_n = Ly.tiles_n, N = Ly.N, K = Ly.K, M = a.M;
    const int r = t - Ly.tile0;
    const int pm = r / tiles_n, pn = r - pm * tiles_n;
    const int m0 = pm * TM, n0 = pn * TN;
    const int lda3 = Ly.lda, ldb = Ly.ldb, ldc = Ly.ldc, ldc3 = Ly.ldc2, alt = Ly.alt;

    // ---- the row panel of the previous layer must be complete (bounded spin; one lane polls one word, relaxed)
    if (Ly.dep) {
      if (tid == 0) {
        const unsigned need = (unsigned)a.L[l - 1].tiles_n;
        const unsigned* p = a.done + (size_t)(l - 1) * a.tiles_m + pm;
        unsigned spins = 0;
        while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
          __builtin_amdgcn_s_sleep(4);
          if (++spins > (1u << 24)) {
            __hip_atomic_fetch_or(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
      }
      __syncthreads();
    }

    f32x4 acc[3][SM][SN];
#pragma unroll
    for (int h = 0; h < 3; ++h)
#pragma unroll
      for (int i = 0; i < SM; ++i)
#pragma unroll
        for (int j = 0; j < SN; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

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
    const unsigned char* abase = reinterpret_cast<const unsigned char*>(Ly.A + (size_t)m0 * lda3);
    const unsigned char* bbase = reinterpret_cast<const unsigned char*>(Ly.W + (size_t)n0 * ldb);
    auto dma_a = [&](int stage, int kt) __attribute__((always_inline)) {  // (sc1: the panel may have been written in this launch)
      unsigned char* base = smem_a + stage * SA_BYTES + w4 * 1024;
      const unsigned char* g = abase + (size_t)kt * H2_ROWB;
#pragma unroll
      for (int p = 0; p < NPA; ++p) glds16b_sc1(g + goff[p], base + p * 4096);
    };
    auto dma_b = [&](int stage, int kt) __attribute__((always_inline)) {
      unsigned char* base = smem_a + 2 * SA_BYTES + stage * SB_BYTES + w4 * 1024;
      const unsigned char* g = bbase + (size_t)kt * H2_ROWB;
#pragma unroll
      for (int p = 0; p < NPB; ++p) glds16b(g + goff[p], base + p * 4096);
    };
    const int nk = K >> 5;
    auto kc = [&](int k) __attribute__((always_inline)) { return min(k, nk - 1); };

    f16x8 fa[SM][2], fb[SN][2];
#define AIMNET_H2_PRODUCT(SET, PA, PB)                                                                                      \
  _Pragma("unroll") for (int i = 0; i < SM; ++i) _Pragma("unroll") for (int jj = 0; jj < SN; ++jj) acc[SET][i][jj] = \
      __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[jj][PB], fa[i][PA], acc[SET][i][jj], 0, 0, 0);
    // ---- the ping-pong of gemm_h2.hip (SCHED 0): prologue A(0) by group 0; B(0), B(1) by group 1
    if (!late) {
      dma_a(0, 0);
      wait_vm<0>();
    } else {
      dma_b(0, 0);
      dma_b(1, kc(1));
      wait_vm<NPB>();
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    auto seg_load = [&](int j, int sa, int sb, auto g_c) __attribute__((always_inline)) {
      constexpr int G = decltype(g_c)::value;
      const unsigned oa = adA + sa * SA_BYTES, ob = adB + sb * SB_BYTES;
      read_strips_h<0, SN, 1>(fb, ob);
      read_strips_h<0, SM, 0>(fa, oa);
      read_strips_h<0, SN, 0>(fb, ob);
      read_strips_h<0, SM, 1>(fa, oa);
      if constexpr (G == 0) {
        dma_a(sa ^ 1, kc(j + 1));
        wait_lgkm<0>();
      } else {
        dma_b(sb == 0 ? 2 : sb - 1, kc(j + 2));  // (sb + 2) % 3
        wait_vm<NPB>();
        wait_lgkm<0>();
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    auto seg_compute = [&](auto par_c, auto g_c) __attribute__((always_inline)) {
      constexpr int G = decltype(g_c)::value, PAR = decltype(par_c)::value;
      __builtin_amdgcn_sched_barrier(0);
      AIMNET_H2_PRODUCT(2, 0, 1)
      AIMNET_H2_PRODUCT(PAR, 0, 0)
      AIMNET_H2_PRODUCT(2, 1, 0)
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (G == 0) wait_vm<0>();
      __builtin_amdgcn_sched_barrier(0);
    };
    auto bar = [&]() __attribute__((always_inline)) {
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    auto run = [&](auto g_c) __attribute__((always_inline)) {
      int sa = 0, sb = 0, j = 0;
      auto next = [&]() __attribute__((always_inline)) {
        sa ^= 1;
        sb = sb == 2 ? 0 : sb + 1;
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
      if (j < nk) {
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
      bar();
    }
#undef AIMNET_H2_PRODUCT
    wait_vm<0>();
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue (the arithmetic of gemm_h2_kernel's, selected at run time)
    const float s0 = alt == 2 ? -1.0f : 1.0f, s1 = alt == 1 ? -1.0f : 1.0f;
    const float* bias = Ly.bias;
    const int* brow = Ly.brow;
    const int ldbias = Ly.ldbias;
    float* C = Ly.C;
    float* D = Ly.D;
    unsigned short* C3 = Ly.C2;
    auto epilogue = [&](auto epi_c, auto out_c, auto wt_c) __attribute__((always_inline)) {
      constexpr int EPI = decltype(epi_c)::value;
      constexpr bool OUT3 = decltype(out_c)::value != 0, WT = decltype(wt_c)::value != 0;
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
          for (int rr = 0; rr < 4; ++rr) {
            float hh, dd;
            gelu_and_grad(v[rr], hh, dd);
            v[rr] = hh;
            d[rr] = dd;
          }
          if (D) *reinterpret_cast<f32x4*>(D + o) = d;
        } else if (EPI == EPI_MUL) {
          v = v * *reinterpret_cast<const f32x4*>(D + o);
        }
        return true;
      };
      if constexpr (OUT3) {
#pragma unroll
        for (int j = 0; j < SN; j += 2) {
#pragma unroll
          for (int i = 0; i < SM; ++i) {
            const int row = m0 + wm * 16 * SM + 16 * i + l16;
            unsigned short* crow = C3 + (size_t)row * ldc3;
            f32x4 v0, v1;
            const bool ok = finish(i, j, v0);
            finish(i, j + 1, v1);
            if (!ok) continue;
            store_h2_tile_pair<WT>(crow, n0 + wn * 16 * SN + 16 * j, lc, v0, v1);
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
    };
    using E0 = std::integral_constant<int, EPI_NONE>;
    using E1 = std::integral_constant<int, EPI_BIAS>;
    using E2 = std::integral_constant<int, EPI_BIAS_GELU>;
    using E3 = std::integral_constant<int, EPI_MUL>;
    const int epi = Ly.epi;
    if (Ly.out2) {  // h2 output: write-through when a later layer of this launch reads it
      if (Ly.publish) {
        if (epi == EPI_BIAS_GELU) epilogue(E2{}, I1{}, I1{});
        else epilogue(E3{}, I1{}, I1{});
      } else {
        if (epi == EPI_BIAS_GELU) epilogue(E2{}, I1{}, I0{});
        else epilogue(E3{}, I1{}, I0{});
      }
    } else if (epi == EPI_NONE) epilogue(E0{}, I0{}, I0{});
    else if (epi == EPI_BIAS) epilogue(E1{}, I0{}, I0{});
    else if (epi == EPI_BIAS_GELU) epilogue(E2{}, I0{}, I0{});
    else epilogue(E3{}, I0{}, I0{});

    // ---- publish: every storing wave drains its stores, then ONE lane adds to the panel's arrival counter; next tile from the queue
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // (also: every wave has left the LDS rings)
    if (tid == 0) {
      if (Ly.publish)
        __hip_atomic_fetch_add(a.done + (size_t)l * a.tiles_m + pm, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_ctl[0] = (int)__hip_atomic_fetch_add(a.queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    t = __builtin_amdgcn_readfirstlane(s_ctl[0]);
  }
}

size_t gemm_h2_chain_sync_words(int n_layers, int M) {
  const int tiles_m = ceil_div(M, 160);
  return 8 + (size_t)n_layers * tiles_m;  // [0] queue head, [1] error word, [8 ...] arrival counters
}

// layers: A / W / outputs / epilogue of each layer filled by the caller (A of layer l > 0 must be the C2 of layer l-1);
// sync: gemm_h2_chain_sync_words() unsigned words of device memory, zeroed here by a memset node in front of the launch
int launch_gemm_h2_chain(hipStream_t stream, const H2ChainLayer* layers, int n_layers, int M, unsigned* sync, size_t sync_words) {
  if (M <= 0 || n_layers <= 0) return 0;
  constexpr int SM = 5, SN = 2, WN = 4, TM = 160, TN = 128;
  if (n_layers > 4 || sync_words < gemm_h2_chain_sync_words(n_layers, M) || !sync) {
    set_last_error("gemm_h2_chain: 1 - 4 layers, a sync buffer of gemm_h2_chain_sync_words() words");
    return -1;
  }
  H2ChainArgs a{};
  a.n_layers = n_layers;
  a.M = M;
  a.tiles_m = ceil_div(M, TM);
  int tile0 = 0;
  for (int l = 0; l < n_layers; ++l) {
    H2ChainLayer L = layers[l];
    if (L.K % 32 != 0 || (L.lda % 64) || (L.ldb % 64) || (L.N & 3) || (L.ldc & 3) || (L.out2 && (L.ldc2 % 64 || (L.N & 31) || L.ldc2 < 2 * L.N)) ||
        (((size_t)L.A | (size_t)L.W | (size_t)L.bias | (size_t)L.C | (size_t)L.C2 | (size_t)L.D) & 15) || L.alt < 1 || L.alt > 2 ||
        (L.out2 && L.epi != EPI_BIAS_GELU && L.epi != EPI_MUL)) {
      set_last_error("gemm_h2_chain: layer %d: operand alignment / shape (the constraints of launch_gemm_h2_cfg)", l);
      return -1;
    }
    L.tiles_n = ceil_div(L.N, TN);
    L.tile0 = tile0;
    L.dep = l > 0 && (const void*)L.A == (const void*)layers[l - 1].C2 && layers[l - 1].out2;
    L.publish = 0;
    if (l > 0 && !L.dep) {
      set_last_error("gemm_h2_chain: layer %d does not read layer %d's split output", l, l - 1);
      return -1;
    }
    tile0 += a.tiles_m * L.tiles_n;
    a.L[l] = L;
  }
  for (int l = 0; l + 1 < n_layers; ++l) a.L[l].publish = 1;
  a.n_tiles = tile0;
  a.queue = sync;
  a.err = sync + 1;
  a.done = sync + 8;
  AIMNET_HIP_CHECK(hipMemsetAsync(sync, 0, gemm_h2_chain_sync_words(n_layers, M) * sizeof(unsigned), stream));
  constexpr size_t lds = (size_t)chain_ring_bytes(TM, TN) + 64;
  static PerDeviceOnce once;
  if (once.first())
    AIMNET_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_h2_chain_kernel<SM, SN, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int grid = std::min(a.n_tiles, device_cus());
  hipLaunchKernelGGL((gemm_h2_chain_kernel<SM, SN, WN>), dim3(grid), dim3(512), lds, stream, a);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

}  // namespace aimnet
