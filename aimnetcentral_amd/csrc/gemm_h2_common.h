// gemm_h2_common.h - the fp16x2-split operand form ("h2") of the MLP GEMMs (gemm_h2.hip, gemm_head.hip) and the device helpers
// its producers use (conv.hip row assembly, model.hip build_zbar, the GEMM epilogues).
//
// fp32 value x = hi + lo / 4096 with hi = fp16(x) and lo = fp16((x - hi) * 4096), both rounded to nearest even: |x - hi| <= 2^-12 |x|,
// so the scaled residual is no larger than |x| itself (no overflow beyond hi's own; gradual underflow only below 1e-11 absolute)
// and its own rounding leaves |x - hi - lo / 4096| <= 2^-24 |x| - the rounding of ONE fp32 operation.  A product of two such
// operands needs three matrix instructions instead of the six of the bf16x3 split (gemm_bf3a.hip):
//     a b = ah bh + (ah bl + al bh) / 4096 + al bl / 4096^2,   the last term <= 2^-24 |a b| (typically 2^-26) and dropped,
// with the cross terms summed in an accumulator of their own and scaled once in the epilogue.  Measured on the layer shapes
// (tests/tools/h2_bench.py, profiles/r5_gemm_h2.md): rms error against fp64 BELOW the fp32 MFMA chain and below the bf16x3 split.
//
// Memory layout: per row, K/32 blocks of [hi: 32 fp16][lo: 32 fp16] = 128 B (ld counts 16-bit elements = 2 x the padded K).
// Signs (the accumulation-bias cancellation of gemm_bf3a.hip, two interleaved accumulator sets for the hi x hi products):
//   weights     : hi of every ODD k-block negated, lo plain           (split_h2_host, H2_WEIGHT)
//   activations : hi plain, lo of every ODD k-block negated           (the producers below, H2_ACT)
// so that ah (-bh) lands in the odd set with a minus sign (epilogue: even - odd) and both cross terms ah bl, (-al)(-bh) keep
// their sign in the single cross accumulator.
// Range: |x| >= 65520 does not fit hi: it becomes inf, the products inf / NaN, and the energies / forces non-finite.  The kernels that
// write those raise status[6] bit 5 (kernels.h STATUS_NONFINITE, include/aimnet_hip.h) - visible to every consumer of the C ABI, also
// for an overflow in the adjoint sweep alone; the Python layer (engine.py) then repeats the evaluation with the bf16x3 operands
// (set_option("gemm_h2", 0)) and stays on them if that is finite.  Weights are checked at create time (h2_fits).
// (h2_flag_overflow below is the per-element form for the stand-alone split kernel of the tests.)
#pragma once

#include <hip/hip_runtime.h>

#include "gemm_bf3_common.h"
#include "kernels.h"

namespace aimnet {

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr float H2_SCALE = 4096.0f;
constexpr float H2_INV_SCALE = 1.0f / 4096.0f;
constexpr float H2_MAX = 65504.0f;

// two fp32 -> packed hi pair, packed (scaled) lo pair; lo_scale = +-4096 (the sign of the k-block for activations)
__device__ __forceinline__ void split2_pair(float a, float b, float lo_scale, unsigned& hi, unsigned& lo) {
  // (no contraction: with a producer `a = p * q` inlined in front, `a - hi` would become fma(p, q, -hi) - the residual of the UNROUNDED
  // product - in one kernel and not in another; the split is defined on the fp32 value, so that kernels that compute the same value
  // write the same two planes: gemm_chain.hip is bitwise gemm_h2.hip)
#pragma clang fp contract(off)
  const f32x2 x = {a, b};
  const f16x2 h = __builtin_convertvector(x, f16x2);
  const f32x2 r = (x - __builtin_convertvector(h, f32x2)) * lo_scale;
  const f16x2 l = __builtin_convertvector(r, f16x2);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}
// lo scale of an activation column (H2_ACT): odd k-blocks negated
__device__ __forceinline__ float h2_act_scale(int col) { return (col & 32) ? -H2_SCALE : H2_SCALE; }

// |x| that hi cannot hold -> one sticky flag (rare path: a compare per element, the atomic only when it fires)
__device__ __forceinline__ void h2_flag_overflow(int* __restrict__ flag, float amax) {
  if (flag && !(amax < H2_MAX)) atomicOr(flag, 1);
}
__device__ __forceinline__ float h2_amax4(f32x4 v) { return fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))); }

// four consecutive columns col..col+3 (col % 4 == 0) of one row; `row` points at the row's first block
__device__ __forceinline__ void store_h2_x4(unsigned short* __restrict__ row, int col, f32x4 v) {
  unsigned h0, l0, h1, l1;
  const float sc = h2_act_scale(col);
  split2_pair(v[0], v[1], sc, h0, l0);
  split2_pair(v[2], v[3], sc, h1, l1);
  unsigned short* p = row + (col >> 5) * 64 + (col & 31);
  *reinterpret_cast<u32x2*>(p) = u32x2{h0, h1};
  *reinterpret_cast<u32x2*>(p + 32) = u32x2{l0, l1};
}
// one element
__device__ __forceinline__ void store_h2_1(unsigned short* __restrict__ row, int col, float v) {
  unsigned h, l;
  split2_pair(v, 0.0f, h2_act_scale(col), h, l);
  unsigned short* p = row + (col >> 5) * 64 + (col & 31);
  p[0] = (unsigned short)h;
  p[32] = (unsigned short)l;
}
// Two horizontally adjacent 16x16 accumulator tiles of an MFMA epilogue -> h2 row (the contract of store_bf3_tile_pair: lane
// (l16, lc) holds columns 4 lc .. 4 lc + 3 of row l16 of each tile, col0 = first column of tile j, a multiple of 16; after the
// v_permlane16_swap exchange a lane owns 8 consecutive columns and stores 16 B per plane).  All 64 lanes must call it.
__device__ __forceinline__ void store_h2_tile_pair(unsigned short* __restrict__ crow, int col0, int lc, f32x4 v0, f32x4 v1) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  unsigned x[2][2], y[2][2];
  const float sc0 = h2_act_scale(col0), sc1 = h2_act_scale(col0 + 16);  // (col0 is a multiple of 16: each tile lies in one k-block)
  split2_pair(v0[0], v0[1], sc0, x[0][0], x[1][0]);
  split2_pair(v0[2], v0[3], sc0, x[0][1], x[1][1]);
  split2_pair(v1[0], v1[1], sc1, y[0][0], y[1][0]);
  split2_pair(v1[2], v1[3], sc1, y[0][1], y[1][1]);
  const int col = col0 + 16 * (lc & 1) + 8 * (lc >> 1);
  unsigned short* pc = crow + (col >> 5) * 64 + (col & 31);
#pragma unroll
  for (int pl = 0; pl < 2; ++pl) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 1"
                 : "+v"(x[pl][0]), "+v"(y[pl][0]), "+v"(x[pl][1]), "+v"(y[pl][1]));
    *reinterpret_cast<u32x4*>(pc + pl * 32) = u32x4{x[pl][0], x[pl][1], y[pl][0], y[pl][1]};
  }
}

// format-generic store of four consecutive columns: FMT 1 = bf3 (gemm_bf3_common.h), 2 = h2 activation form
template <int FMT>
__device__ __forceinline__ void store_split_x4(unsigned short* __restrict__ row, int col, f32x4 v) {
  if constexpr (FMT == 2) store_h2_x4(row, col, v);
  else store_bf3_x4(row, col, v);
}

// ---- LDS tile of one k-step: blocks of 16 rows, [hi: 16 rows x 64 B][lo: 16 rows x 64 B] = 2 KiB; the 16-byte granule of
// k-chunk c of row r sits in slot c ^ swz(r), swz(r) = (r >> 2) & 2.  A ds_read_b128 of one plane (lane = row & 15, chunk =
// lane >> 4) is serviced in the lane groups {0-3, 12-15, 20-27}, ...: with 64-byte rows the four rows r, r+4, r+8, r+12 share a
// bank window, and this XOR gives them four distinct slots in every group (MI355X_MICROARCH.md, LDS table).
__device__ __forceinline__ int swz_h2(int row) { return (row >> 2) & 2; }
constexpr int H2_ROWB = 128;    // bytes per row per 32-k step in memory
constexpr int H2_STRIP = 2048;  // LDS bytes of a 16-row strip (both planes)

template <int OFF>
__device__ __forceinline__ f16x8 lds_read_frag_h(unsigned addr) {
  f16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
// fragments of SMN consecutive 16-row strips, plane P
template <int I, int SMN, int P>
__device__ __forceinline__ void read_strips_h(f16x8 (&f)[SMN][2], unsigned addr) {
  if constexpr (I < SMN) {
    f[I][P] = lds_read_frag_h<I * H2_STRIP + P * 1024>(addr);
    read_strips_h<I + 1, SMN, P>(f, addr);
  }
}

}  // namespace aimnet
