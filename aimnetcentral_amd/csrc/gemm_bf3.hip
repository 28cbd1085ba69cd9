// gemm_bf3.hip - the MLP GEMMs on the bf16 matrix pipe with fp32-equivalent operands ("bf16x3 split").
//
//   C[M,N] = A[M,K] . Bt[N,K]^T, fused epilogues - the same contract as gemm.hip, which stays the exact-fp32 form
//   (set_option("gemm_bf3", 0)).  Replaces the torch addmm + GELU calls of aimnet/modules/core.py:11-46.
//
// Why: v_mfma_f32_16x16x4_f32 runs at the fp32 VECTOR rate (157 TFLOP/s); tuned assembly reaches ~70 % of it at K <= 736
// (DESIGN.md section 7), so the MLP stack - 56 % of a 10 080-atom step - could not move.  gfx950 has no xf32/TF32, but an
// fp32 number is EXACTLY the sum of three bf16 numbers (x0 = rne(x), x1 = rne(x - x0), x2 = rne(x - x0 - x1): 3 x 8
// significand bits + the signs cover fp32's 24), and a bf16 product is exact in the fp32 accumulator.  Six
// v_mfma_f32_16x16x32_bf16 products per tile - a0b0, a0b1, a1b0, a1b1, a0b2, a2b0 - drop only a1b2 + a2b1 + a2b2
// <= 2^-26 |ab|, a quarter of the fp32 rounding of the product itself; the accumulation is fp32 as before.  Six bf16 MFMAs
// per 32 k at 16x the fp32 rate = 2.67x the fp32 matrix peak for the same arithmetic.
//
// Split-operand format ("bf3"): a matrix [rows][K], K % 32 == 0, is stored per row as K/32 blocks of 192 bytes:
// [plane 0: 32 bf16][plane 1: 32 bf16][plane 2: 32 bf16].  One 16-byte granule = 8 consecutive k of one plane = one lane's
// operand of the 16x16x32 MFMA.
//
// Where the split happens (measured on MI355X, tests/tools/bf3_bench.py, profiles/r3_gemm_bf3.md):
//   * weights: once, on the host, at engine creation; they reach LDS by global_load_lds DMA in bf3 form;
//   * activations: stay fp32 in HBM (every producer and consumer of the MLP rows is untouched, 4 instead of 6 bytes per element
//     of HBM and L2 traffic, and the exact-fp32 kernels of gemm.hip run on the very same buffers).  A block loads its fp32 row
//     panel global -> registers (full 128-byte lines per row), splits each element ONCE per block (4.5 VALU instructions) and
//     writes the three planes into a double-buffered LDS tile, two k-steps ahead of the MFMAs that consume it.
//   A first version kept the activations in bf3 form in HBM (split in the producing epilogue): its main loop was bound by the
//   L2 -> LDS stream (6 B per element, 192-byte row pieces that straddle 128-byte lines: 4/3 over-fetch) and its epilogue by the
//   wider stores - 52 us against 85 us (exact fp32) on the 10 080 x 512 x 736 layer, of which 25 us DMA floor and 15 us epilogue.
//
// Kernel: 512 threads = WM x WN waves, block tile (16 SM WM) x (16 SN WN), 32-k steps; lanes own 4 consecutive output columns
// (operands swapped into the MFMA like gemm_nt_panel_kernel), XCD-aware tile remap, the fused epilogues of gemm.hip.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>

#include "common.h"
#include "gemm_bf3_common.h"
#include "kernels.h"

namespace aimnet {

// ---- fp32 [M][ld] (columns c0 .. c0+K) -> bf3 [M][Kp/32][3][32]; columns >= K of the last block are zero ------------------
__global__ __launch_bounds__(256) void split_bf3_kernel(const float* __restrict__ src, int ld, int M, int K, int Kp,
                                                        unsigned short* __restrict__ dst, int ldd, int neg_from) {
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
    if (neg_from == -2 ? ((col >> 5) & 1) != 0 : col >= neg_from) v = -v;  // -2: every odd k-block (BF3_ALT)
    store_bf3_x4(dst + (size_t)m * ldd, col, v);
  }
}

int launch_split_bf3(hipStream_t s, const float* src, int ld, int M, int K, unsigned short* dst, int ldd, int neg_from_block) {
  if (M <= 0 || K <= 0) return 0;
  const int Kp = (K + 31) / 32 * 32;
  const size_t n = (size_t)M * (Kp >> 2);
  const int blocks = (int)std::min<size_t>((n + 255) / 256, 8192);
  const int neg_from = neg_from_block == BF3_ALT ? -2 : neg_from_block >= (1 << 24) ? (1 << 30) : neg_from_block * 32;
  hipLaunchKernelGGL(split_bf3_kernel, dim3(blocks), dim3(256), 0, s, src, ld, M, K, Kp, dst, ldd, neg_from);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

// host-side split of a weight matrix [rows][K] (K % 32 == 0) into the bf3 layout; round-to-nearest-even like the device
static inline unsigned short bf16_rne_host(float x, float* back) {
  unsigned u;
  memcpy(&u, &x, 4);
  unsigned short h;
  if ((u & 0x7fffffffu) > 0x7f800000u) h = (unsigned short)((u >> 16) | 0x40);  // NaN stays NaN
  else h = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
  const unsigned ub = (unsigned)h << 16;
  memcpy(back, &ub, 4);
  return h;
}
void split_bf3_host(const float* w, int rows, int K, unsigned short* out, int neg_from_block) {
  for (int r = 0; r < rows; ++r)
    for (int k = 0; k < K; ++k) {
      const bool neg = neg_from_block == BF3_ALT ? ((k >> 5) & 1) != 0 : (k >> 5) >= neg_from_block;
      const float x = neg ? -w[(size_t)r * K + k] : w[(size_t)r * K + k];
      float f0, f1, f2;
      const unsigned short h0 = bf16_rne_host(x, &f0);
      const float r1 = x - f0;
      const unsigned short h1 = bf16_rne_host(r1, &f1);
      const float r2 = r1 - f1;
      const unsigned short h2 = bf16_rne_host(r2, &f2);
      unsigned short* o = out + (size_t)r * 3 * K + (k >> 5) * 96 + (k & 31);
      o[0] = h0;
      o[32] = h1;
      o[64] = h2;
    }
}

// ---- activation registers ------------------------------------------------------------------------------------------------
// The fp32 quads of the activation tile travel global -> VGPR -> (split) -> LDS, and they are in flight across two barriers and
// a loop back-edge.  Held in C++ variables (asm-load outputs) they are unsafe: the register allocator is free to copy or re-home
// a variable right behind the load - before the data has arrived (observed: v_mov of the destination registers at the loop
// back-edge, garbage in the tile).  Compiler-tracked loads are safe but every loop-carried use becomes s_waitcnt vmcnt(0), which
// also drains the look-ahead DMA.  So the quads live in 24 FIXED registers, v232..v255, that the compiler never sees: the kernel
// is capped at 232 allocatable VGPRs (amdgpu_num_vgpr) and only the asm statements below name them (the clobber lists make the
// kernel descriptor count them).  Quad IDX = set * 3 + q.
template <int IDX>
struct AQuad;
#define AIMNET_BF3_AQUAD(IDX, R0, R1, R2, R3)                                                                                  \
  template <>                                                                                                                  \
  struct AQuad<IDX> {                                                                                                          \
    static __device__ __forceinline__ void load(const float* p) {                                                              \
      asm volatile("global_load_dwordx4 v[" #R0 ":" #R3 "], %0, off" ::"v"(p) : "memory", "v" #R0, "v" #R1, "v" #R2, "v" #R3); \
    }                                                                                                                          \
    static __device__ __forceinline__ unsigned cvt_lo() {                                                                      \
      unsigned h;                                                                                                              \
      asm volatile("v_cvt_pk_bf16_f32 %0, v" #R0 ", v" #R1 : "=v"(h));                                                         \
      return h;                                                                                                                \
    }                                                                                                                          \
    static __device__ __forceinline__ unsigned cvt_hi() {                                                                      \
      unsigned h;                                                                                                              \
      asm volatile("v_cvt_pk_bf16_f32 %0, v" #R2 ", v" #R3 : "=v"(h));                                                         \
      return h;                                                                                                                \
    }                                                                                                                          \
    template <int E>                                                                                                           \
    static __device__ __forceinline__ float minus(float f) { /* x[E] - f */                                                    \
      float r;                                                                                                                 \
      if constexpr (E == 0) asm volatile("v_sub_f32 %0, v" #R0 ", %1" : "=v"(r) : "v"(f));                                     \
      else if constexpr (E == 1) asm volatile("v_sub_f32 %0, v" #R1 ", %1" : "=v"(r) : "v"(f));                                \
      else if constexpr (E == 2) asm volatile("v_sub_f32 %0, v" #R2 ", %1" : "=v"(r) : "v"(f));                                \
      else asm volatile("v_sub_f32 %0, v" #R3 ", %1" : "=v"(r) : "v"(f));                                                      \
      return r;                                                                                                                \
    }                                                                                                                          \
  };
AIMNET_BF3_AQUAD(0, 232, 233, 234, 235)
AIMNET_BF3_AQUAD(1, 236, 237, 238, 239)
AIMNET_BF3_AQUAD(2, 240, 241, 242, 243)
AIMNET_BF3_AQUAD(3, 244, 245, 246, 247)
AIMNET_BF3_AQUAD(4, 248, 249, 250, 251)
AIMNET_BF3_AQUAD(5, 252, 253, 254, 255)
#undef AIMNET_BF3_AQUAD

// second and third plane of two fp32 residuals (first plane already removed)
__device__ __forceinline__ void split2_pair(float a, float b, unsigned& p1, unsigned& p2) {
  const f32x2 r1 = {a, b};
  const bf16x2 h1 = __builtin_convertvector(r1, bf16x2);
  const f32x2 r2 = r1 - __builtin_convertvector(h1, f32x2);
  const bf16x2 h2 = __builtin_convertvector(r2, bf16x2);
  p1 = __builtin_bit_cast(unsigned, h1);
  p2 = __builtin_bit_cast(unsigned, h2);
}

// weight stage: TN rows of 192 B, rounded up to whole DMA passes of the 8 waves (8 KiB each) so that every wave issues the
// same number of global_load_lds per stage (the vmcnt bookkeeping below is then one set of immediates)
constexpr int bf3_b_passes(int TN) { return (TN * 12 + 511) / 512; }
// activation tile: TM rows, rounded up to whole passes of the 512 threads (64 rows each) so that the last pass needs no predicate
constexpr int bf3_a_rows(int TM) { return (TM + 63) / 64 * 64; }
constexpr int bf3_lds_bytes(int SM, int SN, int WM, int WN, int NSTB) {
  return 2 * bf3_a_rows(16 * SM * WM) * ROWB + NSTB * bf3_b_passes(16 * SN * WN) * 8192;
}

#ifdef AIMNET_BF3_TIMING
__device__ unsigned long long g_bf3_stamps[1024];
#endif

// Accumulation bias.  v_mfma_f32_16x16x32_bf16 aligns its 32 products and the accumulator in a fixed-point adder and TRUNCATES
// what falls below its width - two's-complement truncation, i.e. always towards minus infinity.  One instruction loses ~2^-7.5 of
// an fp32 ulp that way; the 138 accumulations of a K = 736 layer add up to -0.8 ulp of |z| on EVERY output element (measured,
// tests/tools/bf3_bias.py: mean error -4.6e-8 |z| against -2e-10 for the fp32 MFMA chain, at a smaller rms).  A one-signed
// error does not average out over atoms or layers: config 5's molecule energies moved by 2e-4 eV (rms), twice the fp32 noise.
// Remedy without a second accumulator: a negated accumulation is truncated towards minus infinity as well, which is towards PLUS
// infinity of the quantity it stands for.  So the weight blocks of the last 44 % of K are stored negated (host split), the
// accumulators change sign once, at step `kneg`, and the epilogue undoes the sign: the two phases' biases cancel.  Measured mean
// error with the flip at 0.5 / 0.6 / 0.67 / 0.75 of K: +1.1e-8 / +1.8e-9 / -3.0e-9 / -1.3e-8 |z| on random-sign operands,
// -2.9e-9 / -7.1e-9 / -9.3e-9 / -1.4e-8 on all-positive ones (no flip: -4.7e-8 / -2.9e-8; fp32 MFMA chain: -2e-10); rms unchanged.
// The engine flips at 0.56 K, chosen on end-to-end energies against the fp64 oracle (engine.hip, upload_layer).

// C[M,N] = A[M,K] . B^T: A fp32 [M][lda], B the bf3 split of Bt [N][K] (ldb = bf16 elements per row = 3 x the padded K of the
// full weight matrix), C / D / bias / brow as in gemm_nt_panel_kernel.
//
// Schedule ("ping-pong").  A first version ran all eight waves in lock step - barrier, DMA issue, 21 fragment reads, 60 MFMAs -
// and the matrix pipe idled through everybody's issue and read phases: 3 750 cycles per 32-k step against 2 200 of MFMA work.
// Here waves 0-3 (group 0, the upper half of the block tile, one wave per SIMD) and waves 4-7 (group 1, the lower half; wave w + 4
// shares the SIMD of wave w) alternate between a LOAD segment and a COMPUTE segment, half a step out of phase, with one
// s_barrier between segments:
//      group 0:  L(0) | C(0) | L(1) | C(1) | ...            L(k): ds_read the fragments of step k into registers, issue the weight
//      group 1:   -   | L(0) | C(0) | L(1) | ...                  DMA of step k+2, split A(k+1) into the LDS tile SA[(k+1)&1], load A(k+3)
//                                                           C(k): the 6 x SM x SN MFMAs of step k, registers only
// so that on every SIMD one wave feeds the matrix pipe while its partner does the memory, LDS and VALU work of its next step.
// Buffers: the weight ring has three stages (stage k is read in segments 2k and 2k+1 and refilled for step k+3 from segment
// 2k+2 on), SA two (A(k+1) is written in segments 2k / 2k+1, read in 2k+2 / 2k+3).  Every wave issues, per L segment, its share
// of the weight DMA first and its activation loads last; the memory counter retires in order, so each wait is "at most n younger
// operations outstanding" with a constant n.
template <int EPI, int SM, int SN, int WN, int QC>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_num_vgpr(232))) void gemm_bf3_kernel(
    const float* __restrict__ A, int lda, const unsigned short* __restrict__ Bt, int ldb, int M, int N, int K,
    const float* __restrict__ bias, float* __restrict__ C, float* __restrict__ D, int ldc, const int* __restrict__ brow,
    int ldbias, int kneg) {
  static_assert(WN == 8 || WN == 4 || WN == 2, "waves across N");
  constexpr int WM = 8 / WN;
  constexpr int TM = 16 * SM * WM, TN = 16 * SN * WN;
  constexpr int SA_BYTES = bf3_a_rows(TM) * ROWB;  // rows >= TM: padding written by the last pass of quads, never read
  constexpr int NPB = bf3_b_passes(TN);          // DMA wave-instructions per wave per weight stage
  constexpr int BST_BYTES = NPB * 8192;
  constexpr int NGRAN_B = TN * 12;
  constexpr int NQ = (TM * 8 + 511) / 512;       // fp32 quads of the activation tile per thread and step
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];

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

  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_b;
  const unsigned ldsB = lds0 + 2 * SA_BYTES;

  // activation quads: Q = q * 512 + tid -> row Q >> 3, k-quad Q & 7 (8 consecutive lanes read one 128-byte line); the quad's
  // four bf16 of plane P go to row * 192 + P * 64 + ((kq >> 1) ^ swz(row)) * 16 + (kq & 1) * 8
  const float* asrc[NQ];
  unsigned adst[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int Q = q * 512 + tid;
    const int row = Q >> 3, kq = Q & 7;  // rows >= TM (last pass): a clamped source row, an LDS padding row
    asrc[q] = A + (size_t)min(m0 + min(row, TM - 1), M - 1) * lda + kq * 4;
    adst[q] = lds0 + row * ROWB + (((kq >> 1) ^ swz192(row)) << 4) + (kq & 1) * 8;
  }
  // weight granules: G = p * 512 + tid -> row G / 12, plane (G % 12) / 4, slot G % 4 holding k-chunk slot ^ swz(row); granules
  // beyond the tile (the padding of the last pass) re-read the last granule into the stage's padding
  const unsigned char* bsrc[NPB];
#pragma unroll
  for (int p = 0; p < NPB; ++p) {
    const int G = min(p * 512 + tid, NGRAN_B - 1);
    const int row = G / 12, g12 = G % 12;
    const int pl = g12 >> 2, kc = (g12 & 3) ^ swz192(row);
    bsrc[p] = reinterpret_cast<const unsigned char*>(Bt + (size_t)min(n0 + row, N - 1) * ldb) + pl * 64 + kc * 16;
  }
  auto dma_b = [&](int stage, int kt) __attribute__((always_inline)) {
    unsigned char* base = smem_b + 2 * SA_BYTES + stage * BST_BYTES + wid * 1024;
    const size_t go = (size_t)kt * ROWB;
#pragma unroll
    for (int p = 0; p < NPB; ++p) glds16b(bsrc[p] + go, base + p * 8192);
  };
  static_assert(NQ <= 3, "two sets of three activation quads (v232..v255)");
  // set SET (0 / 1) <- the quads of k-step kt
  // quads [Q0, Q1) of set SET (0 / 1) <- k-step kt
  static_assert(QC >= 0 && QC <= NQ, "quads split in the compute segment");
  constexpr int NL = NQ - QC;  // quads split in the load segment
  auto load_q = [&](auto set_c, auto q0_c, auto q1_c, int kt) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value, Q0 = decltype(q0_c)::value, Q1 = decltype(q1_c)::value;
    if constexpr (Q0 <= 0 && 0 < Q1) AQuad<SET * 3 + 0>::load(asrc[0] + kt * 32);
    if constexpr (Q0 <= 1 && 1 < Q1) AQuad<SET * 3 + 1>::load(asrc[NQ > 1 ? 1 : 0] + kt * 32);
    if constexpr (Q0 <= 2 && 2 < Q1) AQuad<SET * 3 + 2>::load(asrc[NQ > 2 ? 2 : 0] + kt * 32);
  };
  auto split_q = [&](auto idx_c, unsigned dst) __attribute__((always_inline)) {
    using Q = AQuad<decltype(idx_c)::value>;
    const unsigned h01 = Q::cvt_lo(), h23 = Q::cvt_hi();  // plane 0 of the four values, round to nearest even
    const float r0 = Q::template minus<0>(__builtin_bit_cast(float, h01 << 16));
    const float r1 = Q::template minus<1>(__builtin_bit_cast(float, h01 & 0xffff0000u));
    const float r2 = Q::template minus<2>(__builtin_bit_cast(float, h23 << 16));
    const float r3 = Q::template minus<3>(__builtin_bit_cast(float, h23 & 0xffff0000u));
    unsigned a1, a2, b1, b2;
    split2_pair(r0, r1, a1, a2);
    split2_pair(r2, r3, b1, b2);
    lds_write8<0>(dst, h01, h23);
    lds_write8<64>(dst, a1, b1);
    lds_write8<128>(dst, a2, b2);
  };
  auto split_r = [&](auto set_c, auto q0_c, auto q1_c, unsigned buf_off) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value, Q0 = decltype(q0_c)::value, Q1 = decltype(q1_c)::value;
    if constexpr (Q0 <= 0 && 0 < Q1) split_q(std::integral_constant<int, SET * 3 + 0>{}, adst[0] + buf_off);
    if constexpr (Q0 <= 1 && 1 < Q1) split_q(std::integral_constant<int, SET * 3 + 1>{}, adst[NQ > 1 ? 1 : 0] + buf_off);
    if constexpr (Q0 <= 2 && 2 < Q1) split_q(std::integral_constant<int, SET * 3 + 2>{}, adst[NQ > 2 ? 2 : 0] + buf_off);
  };
  using QL = std::integral_constant<int, NL>;  // quads [0, NL): load segment; [NL, NQ): compute segment
  using QE = std::integral_constant<int, NQ>;

  // fragment addresses: row r, plane P, k-chunk c = lane >> 4 -> r * 192 + P * 64 + (c ^ swz(r)) * 16
  const int l16 = lane & 15, lc = lane >> 4;
  const int rA = wm * 16 * SM + l16, rB = wn * 16 * SN + l16;
  const unsigned adA = lds0 + rA * ROWB + ((lc ^ swz192(rA)) << 4);
  const unsigned adB = ldsB + rB * ROWB + ((lc ^ swz192(rB)) << 4);

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  const int nk = K >> 5;
  const bool late = wid >= 4;  // group 1 runs one segment behind group 0
  // Every step issues the same operations: k-steps past the end of K are clamped to the last one (two redundant weight stages
  // and activation tiles per block, which nothing reads).  That keeps the wait counts constants and - the reason - the asm loads
  // unconditional: a conditional asm load makes the compiler merge "loaded" and "not loaded" registers with copies placed right
  // behind the load, i.e. before the data has arrived.
  auto kc = [&](int k) __attribute__((always_inline)) { return min(k, nk - 1); };
#ifdef AIMNET_BF3_TIMING
  // measurement build: waves 0 and 4 of block 0 stamp s_memtime at every segment boundary (g_bf3_stamps, 2 x 512 entries)
  int n_ts = 0;
  auto TS = [&]() __attribute__((always_inline)) {
    if (blockIdx.x == 0 && (wid & 3) == 0 && n_ts < 512) {
      const unsigned long long t = __builtin_readcyclecounter();
      if (lane == 0) g_bf3_stamps[(wid >> 2) * 512 + n_ts] = t;
      ++n_ts;
    }
    __builtin_amdgcn_sched_barrier(0);
  };
#else
  auto TS = [&]() __attribute__((always_inline)) {};
#endif
#ifdef AIMNET_BF3_TIMING_FINE
#define TSF() TS()
#else
#define TSF()
#endif
  TS();  // kernel entry
  // ---- prologue.  VMEM issue order of a wave in the steady state (g = group, A_L / A_C = the quads split in the load / compute
  // segment): ... B(j+1), A_L(j+2) [L(j-1)], A_C(j+2+g) [C(j-1)], B(j+2), A_L(j+3) [L(j)], A_C(j+3+g) [C(j)] ...; the prologue
  // continues that pattern backwards so that the constant wait counts hold from step 0 on.
  // Group 1 splits its compute-segment quads one step further ahead (its C(j) runs beside group 0's L(j+1), which reads step
  // j+1): A_C(j+2) in C(j), hence A_C(1) here.
  // (Round 4: requesting the step-1 loads in front of the first wait - one cold-cache latency instead of two - measured 1-2 % SLOWER on
  // every layer shape, profiles/r4_gemm.md: the larger first burst delays the step-0 data every wave is waiting for.)
  dma_b(0, 0);
  load_q(I0{}, I0{}, QE{}, 0);
  if (QC > 0 && late) load_q(I1{}, QL{}, QE{}, kc(1));
  wait_vm<0>();
  __builtin_amdgcn_sched_barrier(0);
  TS();
  split_r(I0{}, I0{}, QE{}, 0);  // SA[0]
  if (QC > 0 && late) split_r(I1{}, QL{}, QE{}, SA_BYTES);
  load_q(I1{}, I0{}, QL{}, kc(1));                  // "L(-2)": A_L(1)
  if (QC > 0) {                                     // "C(-2)": A_C(1 + g)
    if (!late) load_q(I1{}, QL{}, QE{}, kc(1));
    else load_q(I0{}, QL{}, QE{}, kc(2));
  }
  dma_b(1, kc(1));                                  // "L(-1)": B(1), A_L(2)
  load_q(I0{}, I0{}, QL{}, kc(2));
  if (QC > 0) {                                     // "C(-1)": A_C(2 + g)
    if (!late) load_q(I0{}, QL{}, QE{}, kc(2));
    else load_q(I1{}, QL{}, QE{}, kc(3));
  }
  wait_lgkm<0>();
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();  // SA[0] and weight stage 0 complete (every wave waited for its B(0) pieces above)
  __builtin_amdgcn_sched_barrier(0);

  bf16x8 fa[SM][3], fb[SN][3];
  // L(j): fragments of step j into registers; weight DMA of step j+2 into the ring stage of step j-1; split of the first NL quads
  // of A(j+1) (set (j+1) & 1, loaded in L(j-2)) into SA[(j+1) & 1], whose last readers passed two barriers ago; loads of the same
  // quads of A(j+3) into the same registers.
  // Where the split runs: vector work beside matrix work is slow on this part - the same 75 instructions cost ~450 cycles
  // among the wave's own MFMAs and ~950 beside the partner's - so the remaining QC quads are split in the compute segment, which
  // balances L (600 + 320 per quad) against C (1 100 + 150 per quad).
  auto seg_load = [&](int j, int st, auto par_c) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_c)::value;
    using NPAR = std::integral_constant<int, PAR ^ 1>;
    const unsigned oa = adA + PAR * SA_BYTES, ob = adB + st * BST_BYTES;
#ifdef AIMNET_BF3_PRIO_L  // measurement builds: issue priority of the load / split segment against the partner's matrix segment
    __builtin_amdgcn_s_setprio(AIMNET_BF3_PRIO_L);
#endif
    read_strips<0, SN, 0>(fb, ob);
    read_strips<0, SM, 0>(fa, oa);
    read_strips<0, SN, 1>(fb, ob);
    read_strips<0, SM, 1>(fa, oa);
    read_strips<0, SN, 2>(fb, ob);
    read_strips<0, SM, 2>(fa, oa);
    TSF();
    dma_b(st == 0 ? 2 : st - 1, kc(j + 2));
    TSF();
    if constexpr (NL > 0) {
      wait_vm<2 * NPB + NL + 2 * QC>();  // A_L(j+1); younger: A_C(j+1+g), B(j+1), A_L(j+2), A_C(j+2+g), B(j+2)
      __builtin_amdgcn_sched_barrier(0);
      TSF();
      split_r(NPAR{}, I0{}, QL{}, (PAR ^ 1) * SA_BYTES);
      TSF();
      load_q(NPAR{}, I0{}, QL{}, kc(j + 3));
    }
    wait_vm<2 * NL + QC + NPB>();  // this wave's pieces of weight stage j+1; younger: A_L(j+2), A_C(j+2+g), B(j+2), A_L(j+3)
    wait_lgkm<0>();                // fragments in registers, split planes written
    __builtin_amdgcn_sched_barrier(0);
    TSF();
  };
  // C(j): the 6 x SM x SN matrix instructions of step j on registers, and among them the split of the last QC quads of A(m),
  // m = j + 1 + g (set m & 1 = MPAR), into SA[m & 1], followed by the loads of the same quads of A(m + 2).  At step kneg the sign
  // of the accumulators flips: the weight blocks from there on are stored negated ("Accumulation bias" above).
  auto seg_compute = [&](int j, int m, auto mpar_c) __attribute__((always_inline)) {
    constexpr int MPAR = decltype(mpar_c)::value;
#ifdef AIMNET_BF3_PRIO_C
    __builtin_amdgcn_s_setprio(AIMNET_BF3_PRIO_C);
#endif
    if (j == kneg && j > 0) {
#pragma unroll
      for (int i = 0; i < SM; ++i)
#pragma unroll
        for (int jj = 0; jj < SN; ++jj) acc[i][jj] = -acc[i][jj];
    }
    if constexpr (QC > 0) wait_vm<2 * NPB + 2 * NL + QC>();  // A_C(m); younger: B(j+1), A_L(j+2), A_C(m+1), B(j+2), A_L(j+3)
    __builtin_amdgcn_sched_barrier(0);
#define AIMNET_BF3_PRODUCT(PA, PB)                                                                              \
  _Pragma("unroll") for (int i = 0; i < SM; ++i) _Pragma("unroll") for (int jj = 0; jj < SN; ++jj) acc[i][jj] = \
      __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[jj][PB], fa[i][PA], acc[i][jj], 0, 0, 0);
    AIMNET_BF3_PRODUCT(1, 1)
    AIMNET_BF3_PRODUCT(0, 1)
    AIMNET_BF3_PRODUCT(1, 0)
    AIMNET_BF3_PRODUCT(0, 2)
    if constexpr (QC > 0) split_r(mpar_c, QL{}, QE{}, MPAR * SA_BYTES);
    AIMNET_BF3_PRODUCT(2, 0)
    AIMNET_BF3_PRODUCT(0, 0)
#undef AIMNET_BF3_PRODUCT
    if constexpr (QC > 0) {
      // one matrix instruction, then up to two vector instructions of the split in its shadow
#pragma unroll
      for (int t = 0; t < 6 * SM * SN; ++t) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      load_q(mpar_c, QL{}, QE{}, kc(m + 2));
      wait_lgkm<0>();  // the split planes are in LDS before this wave passes the barrier
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto bar = [&]() __attribute__((always_inline)) {
    TS();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    TS();
  };
  // the two groups run the same step sequence, group 1 one barrier later; m = j + 1 + g: its parity is fixed per code path
  auto run = [&](auto g_c) __attribute__((always_inline)) {
    constexpr int G = decltype(g_c)::value;
    int st = 0, j = 0;
    for (; j + 1 < nk; j += 2) {
      seg_load(j, st, I0{});
      bar();
      seg_compute(j, j + 1 + G, std::integral_constant<int, (1 + G) & 1>{});
      st = st == 2 ? 0 : st + 1;
      bar();
      seg_load(j + 1, st, I1{});
      bar();
      seg_compute(j + 1, j + 2 + G, std::integral_constant<int, G & 1>{});
      st = st == 2 ? 0 : st + 1;
      if (j + 2 < nk) bar();
    }
    if (j < nk) {  // odd number of steps
      seg_load(j, st, I0{});
      bar();
      seg_compute(j, j + 1 + G, std::integral_constant<int, (1 + G) & 1>{});
    }
  };
  if (late) {
    bar();
    run(I1{});
  } else {
    run(I0{});
    bar();  // group 0 has 2 nk segments, group 1 an empty one in front: both pass 2 nk barriers
  }
  wait_vm<0>();  // the clamped look-ahead of the last steps is still in flight; the wave must not end (LDS released) under its DMA
  __builtin_amdgcn_sched_barrier(0);
  TS();  // main loop done

  // epilogue: sfin * acc[i][j][r] = C[m0 + wm*16*SM + 16 i + (lane&15)][n0 + wn*16*SN + 16 j + 4 (lane>>4) + r]
  const float sfin = kneg < nk ? -1.0f : 1.0f;  // the accumulators ended in the negated phase
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
      if (brow && (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU))
        bv = *reinterpret_cast<const f32x4*>(bias + (size_t)min(63, max(0, brow[row])) * ldbias + col);
      f32x4 v = acc[i][j] * sfin;
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
#ifdef AIMNET_BF3_TIMING
  __builtin_amdgcn_sched_barrier(0);
  TS();  // epilogue stores issued
  wait_vm<0>();
  __builtin_amdgcn_sched_barrier(0);
  TS();  // ... and acknowledged
#endif
}

template <int SM, int SN, int WN, int QC>
static int launch_bf3(hipStream_t stream, int epi, const float* A, int lda, const unsigned short* Bt, int ldb, int M, int N, int K,
                      const float* bias, float* C, float* D, int ldc, const int* brow, int ldbias, int kneg) {
  constexpr int WM = 8 / WN, TM = 16 * SM * WM, TN = 16 * SN * WN;
  const int tiles = ceil_div(M, TM) * ceil_div(N, TN);
  const size_t lds = (size_t)bf3_lds_bytes(SM, SN, WM, WN, 3);
  static_assert(bf3_lds_bytes(SM, SN, WM, WN, 3) <= 160 * 1024, "LDS");
  dim3 grid(tiles), block(512);
#define AIMNET_BF3_LAUNCH(E)                                                                                          \
  {                                                                                                                   \
    static PerDeviceOnce once;                                                                                        \
    if (once.first())                                                                                                 \
      AIMNET_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_bf3_kernel<E, SM, SN, WN, QC>,                     \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                  \
    hipLaunchKernelGGL((gemm_bf3_kernel<E, SM, SN, WN, QC>), grid, block, lds, stream, A, lda, Bt, ldb, M, N, K, \
                       bias, C, D, ldc, brow, ldbias, kneg);                                                          \
  }
  switch (epi) {
    case EPI_NONE: AIMNET_BF3_LAUNCH(EPI_NONE) break;
    case EPI_BIAS: AIMNET_BF3_LAUNCH(EPI_BIAS) break;
    case EPI_BIAS_GELU: AIMNET_BF3_LAUNCH(EPI_BIAS_GELU) break;
    case EPI_MUL: AIMNET_BF3_LAUNCH(EPI_MUL) break;
    default:
      set_last_error("gemm_bf3: bad epilogue %d", epi);
      return -1;
  }
#undef AIMNET_BF3_LAUNCH
  AIMNET_LAUNCH_CHECK();
  return 0;
}

static int g_bf3_force_tile = 0;  // AIMNET_BF3_TILE forces one configuration (A/B runs)

struct Bf3Cand { int id, tm, tn; };
// id = 100 * WN (waves across N; 8 / WN across M) + 10 * SM + SN; block tile (16 SM 8 / WN) x (16 SN WN)
static const Bf3Cand kBf3Cands[] = {{452, 160, 128}, {224, 128, 128}, {432, 96, 128}, {422, 64, 128}, {223, 128, 96},
                                    {851, 80, 128},  {234, 192, 128}, {871, 112, 128}, {861, 96, 128}, {891, 144, 128}};

// Tile choice: the busiest CU runs ceil(tiles / CUs) tiles one after the other (all these tiles hold one block per CU); a tile
// costs its MFMA work (tm x tn), the operand stream and split work per k-step (tm + tn) and a fixed prologue / epilogue part.
// Fitted to tests/tools/bf3_bench.py on the fourteen MLP layer shapes at 10 080 rows: it reproduces the measured best tile of
// every shape (profiles/r3_gemm_bf3.md).
static int choose_bf3_tile(int M, int N) {
  const long n_cu = device_cus();
  int best = kBf3Cands[0].id;
  double best_cost = 1e300;
  for (const Bf3Cand& c : kBf3Cands) {
    const long tiles = (long)ceil_div(M, c.tm) * ceil_div(N, c.tn);
    const long per_cu = (tiles + n_cu - 1) / n_cu;
    const double cost = (double)per_cu * ((double)c.tm * c.tn + 60.0 * (c.tm + c.tn) + 3000.0);
    if (cost < best_cost) { best_cost = cost; best = c.id; }
  }
  return best;
}

int launch_gemm_bf3_cfg(hipStream_t stream, int cfg, int epi, const float* A, int lda, const unsigned short* Bt, int ldb, int M,
                        int N, int K, const float* bias, float* C, float* D, int ldc, const int* brow, int ldbias, int kneg) {
  if (M <= 0) return 0;
  if (K % 32 != 0 || (lda & 3) || (ldb % 96) || (N & 3) || (ldc & 3) ||
      (((size_t)A | (size_t)Bt | (size_t)bias | (size_t)C | (size_t)D) & 15)) {
    set_last_error("gemm_bf3: K=%d must be a multiple of 32, lda/ldc/N multiples of 4, pointers 16-byte aligned, ldb whole 192-byte blocks", K);
    return -1;
  }
  if (cfg == 0) cfg = g_bf3_force_tile;
  if (cfg == 0) cfg = choose_bf3_tile(M, N);
  switch (cfg) {
#define AIMNET_BF3_CASE(ID, SM_, SN_, WN_)                                                                                        \
    case ID: return launch_bf3<SM_, SN_, WN_, 1>(stream, epi, A, lda, Bt, ldb, M, N, K, bias, C, D, ldc, brow, ldbias, kneg);      \
    case 1000 + ID: return launch_bf3<SM_, SN_, WN_, 0>(stream, epi, A, lda, Bt, ldb, M, N, K, bias, C, D, ldc, brow, ldbias, kneg);
    AIMNET_BF3_CASE(452, 5, 2, 4)  // 160 x 128 (2 x 4 waves of 80 x 32; 132 KiB of LDS)
    AIMNET_BF3_CASE(442, 4, 2, 4)  // 128 x 128
    AIMNET_BF3_CASE(432, 3, 2, 4)  //  96 x 128
    AIMNET_BF3_CASE(422, 2, 2, 4)  //  64 x 128
    AIMNET_BF3_CASE(223, 2, 3, 2)  // 128 x  96 (4 x 2 waves of 32 x 48)
    AIMNET_BF3_CASE(224, 2, 4, 2)  // 128 x 128 (4 x 2 waves of 32 x 64)
    AIMNET_BF3_CASE(234, 3, 4, 2)  // 192 x 128 (4 x 2 waves of 48 x 64)
    AIMNET_BF3_CASE(851, 5, 1, 8)  //  80 x 128 (1 x 8 waves of 80 x 16)
    AIMNET_BF3_CASE(861, 6, 1, 8)  //  96 x 128 (1 x 8 waves: row counts in steps of 16 for batches that are no multiple of 160 or 128)
    AIMNET_BF3_CASE(871, 7, 1, 8)  // 112 x 128
    AIMNET_BF3_CASE(891, 9, 1, 8)  // 144 x 128
#undef AIMNET_BF3_CASE
    default:
      set_last_error("gemm_bf3: unknown tile id %d", cfg);
      return -1;
  }
}

#ifdef AIMNET_BF3_TIMING
int gemm_bf3_read_stamps(unsigned long long* host1024) {
  AIMNET_HIP_CHECK(hipMemcpyFromSymbol(host1024, HIP_SYMBOL(g_bf3_stamps), 1024 * sizeof(unsigned long long)));
  return 0;
}
#endif

int gemm_bf3_set_attributes() {
  const char* env = getenv("AIMNET_BF3_TILE");
  g_bf3_force_tile = env ? atoi(env) : 0;
  return 0;
}

}  // namespace aimnet
