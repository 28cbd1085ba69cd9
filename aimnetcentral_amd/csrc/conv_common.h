// conv_common.h - constants and helpers shared by the gather-contract kernels (conv.hip, conv_mfma.hip).
#pragma once

#include "common.h"
#include "kernels.h"

namespace aimnet {

constexpr int A_ = 16, G_ = 16, H_ = 12;
constexpr int NF = A_ * G_;         // 256
constexpr int NV = A_ * H_;         // 192
constexpr int APB = 4;              // atoms (waves) per block
constexpr int CH = 64;              // neighbours staged in LDS per chunk (32 doubled the resident waves: no gain)
constexpr float PI_F = 3.14159265358979323846f;
constexpr int SPLIT_MAX_ATOMS = 1024;  // up to here a block per atom (4 waves share its neighbour row) still fits one wave of blocks

// 2-wide float vectors: LLVM lowers their arithmetic to v_pk_{mul,add,fma}_f32 (2 FMAs per issue slot)
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 mk2(float a, float b) {
  f2 r;
  r.x = a;
  r.y = b;
  return r;
}

// acc += t * s for a 2-vector t and a scalar s that sits in one half of a register PAIR p: the packed FMA broadcasts that half to
// both result lanes through its op_sel modifiers, so no v_mov builds the {s, s} splat (the compiler emits three moves per pair of
// conv_fwd for the three components of u: 17 % of the loop's vector instructions).  HI: s = p.y, else p.x.
template <bool HI>
__device__ __forceinline__ void pk_fma_bcast(f2& acc, const f2& t, const f2& p) {
  if (HI)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(t), "v"(p));
  else
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(t), "v"(p));
}

template <bool HI>
__device__ __forceinline__ f2 pk_mul_bcast(const f2& t, const f2& p) {  // t * s, s = one half of the register pair p
  f2 r;
  if (HI)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(t), "v"(p));
  else
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(r) : "v"(t), "v"(p));
  return r;
}

// Persistent-loop atom assignment.  Hardware places block b on XCD b % 8 (observed; speed only): give
// each XCD one CONTIGUOUS range of atoms, processed in order, so that when the input is spatially
// ordered (crystals, MD frames) the neighbour rows an XCD gathers were mostly produced / recently
// touched by the same XCD and hit its private 4 MiB L2 instead of the Infinity Cache.
struct AtomLoop {
  int first, last, step;  // atoms [first, last) in steps of `step`
};
__device__ __forceinline__ AtomLoop atom_loop(int n_atoms, int apb) {
  const int nb = gridDim.x, b = blockIdx.x;
  if (nb < 8) return AtomLoop{b * apb, n_atoms, nb * apb};
  const int xcd = b & 7, slot = b >> 3;
  const int per = (nb >> 3) + (xcd < (nb & 7) ? 1 : 0);          // blocks resident on this XCD: slot = 0..per-1
  const int nblk = (n_atoms + apb - 1) / apb;                    // atom blocks in total
  const int chunk = (nblk + 7) >> 3;                             // atom blocks per XCD
  const int lo = xcd * chunk, hi = min(nblk, lo + chunk);
  return AtomLoop{(lo + slot) * apb, hi * apb < n_atoms ? hi * apb : n_atoms, per * apb};
}

// cutoff envelope fc(d) = 0.5 (cos(pi d / rc) + 1) and its derivative (AEVSV._calc_aev, modules/aev.py:94-110), one call per pair.
// (sincosf on the fp32 product d * (pi / rc), as the reference forms it: a polynomial in t = d / rc is 3x cheaper and closer to
// the exact value, but its error no longer follows the fp32 reference's argument rounding and the tightest energy gate
// (batch5, 2e-5 eV) fails - measured, reverted.)
__device__ __forceinline__ float basis_fc(const BasisParams& bp, float d, float& dfc) {
  const float dc = fminf(fmaxf(d, 1e-6f), bp.rc);
  const float w = PI_F / bp.rc;
  float sn, cs;
  sincosf(dc * w, &sn, &cs);
  dfc = (d > 1e-6f && d < bp.rc) ? -0.5f * w * sn : 0.0f;
  return 0.5f * (cs + 1.0f);
}

// exp(x) for the radial-basis tables (x <= 0): one v_exp_f32 instead of the 15-instruction expf.  x * log2(e) is formed as an
// exact hi + lo pair (fma residual + the low word of the constant), exp2(hi) is the hardware instruction (1 ulp) and the lo
// part enters to first order, so the result keeps expf's accuracy: |x| reaches ~30 before the Gaussian underflows the sums
// it enters, where a single rounded product would already cost 2e-6 relative.
__device__ __forceinline__ float exp_neg(float x) {
  const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.925963033500011e-8f, LN2 = 0.6931471805599453f;
  const float p = x * L2E_HI;
  const float e = fmaf(x, L2E_HI, -p) + x * L2E_LO;
  const float r = __builtin_amdgcn_exp2f(p);
  return fmaf(r * e, LN2, r);
}

// hand-off of per-wave LDS data: a block barrier when waves share it, otherwise a wave-level fence (no s_barrier)
template <bool BLOCK>
__device__ __forceinline__ void lds_sync() {
  if (BLOCK) {
    __syncthreads();
  } else {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// v_mfma_f32_4x4x1_16B_f32: 16 independent 4x4 outer products, A: lane 4b + r, B: lane 4b + col, C: VGPR r of lane 4b + col
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }

template <int CTRL>
__device__ __forceinline__ float dpp0(float v) {  // DPP move, lanes without a source read 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}


}  // namespace aimnet
