// gemm_bf3_common.h - device helpers shared by the bf16x3-split GEMM kernels (gemm_bf3.hip: activations fp32 in memory, split
// in the kernel; gemm_bf3a.hip: activations pre-split by their producer).  "bf3" layout, LDS tile layout and swizzle: gemm_bf3.hip.
#pragma once

#include <hip/hip_runtime.h>

namespace aimnet {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// two fp32 -> three packed bf16 pairs (v_cvt_pk_bf16_f32 rounds to nearest even; the residuals are exact in fp32)
__device__ __forceinline__ void split3_pair(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
  const f32x2 x = {a, b};
  const bf16x2 h0 = __builtin_convertvector(x, bf16x2);
  const f32x2 r1 = x - __builtin_convertvector(h0, f32x2);
  const bf16x2 h1 = __builtin_convertvector(r1, bf16x2);
  const f32x2 r2 = r1 - __builtin_convertvector(h1, f32x2);
  const bf16x2 h2 = __builtin_convertvector(r2, bf16x2);
  p0 = __builtin_bit_cast(unsigned, h0);
  p1 = __builtin_bit_cast(unsigned, h1);
  p2 = __builtin_bit_cast(unsigned, h2);
}

// store four consecutive columns col..col+3 (col % 4 == 0) of one row in bf3 form; `row` points at the row's first block
__device__ __forceinline__ void store_bf3_x4(unsigned short* __restrict__ row, int col, f32x4 v) {
  unsigned a0, a1, a2, b0, b1, b2;
  split3_pair(v[0], v[1], a0, a1, a2);
  split3_pair(v[2], v[3], b0, b1, b2);
  unsigned short* p = row + (col >> 5) * 96 + (col & 31);
  *reinterpret_cast<u32x2*>(p) = u32x2{a0, b0};
  *reinterpret_cast<u32x2*>(p + 32) = u32x2{a1, b1};
  *reinterpret_cast<u32x2*>(p + 64) = u32x2{a2, b2};
}

// Two horizontally adjacent 16x16 accumulator tiles of an MFMA epilogue (lane (l16, lc) holds columns 4 lc .. 4 lc + 3 of row l16 of
// each: v0 = tile j, v1 = tile j + 1, col0 = first column of tile j, a multiple of 32) -> bf3 row `crow`.  A lane's own 8 bytes
// per plane would make 8-byte stores, which are issue-bound; v_permlane16_swap (odd 16-lane rows of the first operand <-> even
// rows of the second) leaves lane (l16, lc) with columns 8 (lc >> 1) .. + 7 of tile j + (lc & 1): one 16-byte store per plane,
// and the four lanes of a row write the 64 contiguous bytes of one plane segment of a k-block.  All 64 lanes must call it.
__device__ __forceinline__ void store_bf3_tile_pair(unsigned short* __restrict__ crow, int col0, int lc, f32x4 v0, f32x4 v1) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  unsigned x[3][2], y[3][2];
  split3_pair(v0[0], v0[1], x[0][0], x[1][0], x[2][0]);
  split3_pair(v0[2], v0[3], x[0][1], x[1][1], x[2][1]);
  split3_pair(v1[0], v1[1], y[0][0], y[1][0], y[2][0]);
  split3_pair(v1[2], v1[3], y[0][1], y[1][1], y[2][1]);
  const int col = col0 + 16 * (lc & 1) + 8 * (lc >> 1);
  unsigned short* pc = crow + (col >> 5) * 96 + (col & 31);
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 1"
                 : "+v"(x[pl][0]), "+v"(y[pl][0]), "+v"(x[pl][1]), "+v"(y[pl][1]));
    *reinterpret_cast<u32x4*>(pc + pl * 32) = u32x4{x[pl][0], x[pl][1], y[pl][0], y[pl][1]};
  }
}

// one element of a row in bf3 form
__device__ __forceinline__ void store_bf3_1(unsigned short* __restrict__ row, int col, float v) {
  unsigned p0, p1, p2;
  split3_pair(v, 0.0f, p0, p1, p2);
  unsigned short* p = row + (col >> 5) * 96 + (col & 31);
  p[0] = (unsigned short)p0;
  p[32] = (unsigned short)p1;
  p[64] = (unsigned short)p2;
}

// ------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void glds16b(const void* g, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
template <int OFF>
__device__ __forceinline__ bf16x8 lds_read_frag(unsigned addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF>
__device__ __forceinline__ void lds_write8(unsigned addr, unsigned lo, unsigned hi) {
  const u32x2 v = {lo, hi};
  asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}

// LDS tiles: [rows][3 planes][4 granules of 16 B] = 192 B per row; the granule of k-chunk c sits in slot c ^ swz(row),
// swz(row) = (-(row >> 2)) & 3.  With 192-byte rows, rows r and r + 4 start in the same bank window; this XOR makes every
// lane group of a ds_read_b128 ({0-3, 12-15, 20-27}, ...: lane = row & 15, chunk = lane >> 4) touch 16 distinct 16-byte slots
// of the 256-byte bank window (MI355X_MICROARCH.md, LDS table; SQ_LDS_BANK_CONFLICT = 0 measured).
__device__ __forceinline__ int swz192(int row) { return (-(row >> 2)) & 3; }

constexpr int ROWB = 192;  // bytes per row per 32-k step

// fragments of SMN consecutive 16-row strips (16 * 192 B apart), plane P
template <int I, int SMN, int P>
__device__ __forceinline__ void read_strips(bf16x8 (&f)[SMN][3], unsigned addr) {
  if constexpr (I < SMN) {
    f[I][P] = lds_read_frag<I * 16 * ROWB + P * 64>(addr);
    read_strips<I + 1, SMN, P>(f, addr);
  }
}

}  // namespace aimnet
