// conv_mfma.hip - the ConvSV gather-contract kernels on the matrix pipe (v_mfma_f32_4x4x1_16B_f32).
//
// Same reference semantics as conv.hip (ConvSV.forward aev.py:156-189 == Warp kernel conv_sv_2d_sp_wp.py:90-112; backward
// kernels conv_sv_2d_sp_wp.py:115-164 fused with the AEV backward aev.py:94-110), same centre-major algebra
// (oracle/aimnet2_analytic.py), one wave per centre atom.  What changes is WHERE the per-pair contraction runs:
//
//   v_mfma_f32_4x4x1_16B_f32 is 16 independent 4x4 outer products C_b[r][col] += A_b[r] * B_b[col] (K = 1, exact fp32,
//   8 cycles, the fp32 vector rate) with the operand layouts  A: lane 4b + r,  B: lane 4b + col,  C: VGPR r, lane 4b + col.
//   The 16 blocks b are the 16 radial shifts g - the one index of the pair contractions that is neither summed nor shared
//   between the factors.  So lane l = (g = l >> 2, c = l & 3) throughout this file.
//
//   forward    S_i[a,g,c] = sum_m a_j[a,g] * w_m[g,c],  w_m[g,c] = gs_g(d_m) (1,u_m)_c:
//              A = w_m (row = c), B = a_j[4q + col][g] (q = 0..3 -> 4 MFMAs per pair), C_q[c] = S_i[4q + col, g, c];
//              K = the neighbour index = the instruction sequence.  25 VALU instructions per pair become ~8.
//   backward   X_m[g,c] = sum_a a_jm[a,g] Sbar_i[a,g,c]   (rows = 4 neighbours m at once, K = a: 16 MFMAs per 4 pairs)
//              Y_m[g,c] = sum_a a_i[a,g]  Sbar_jm[a,g,c]  (A = a_i replicated over the rows, K = a: 16 MFMAs per pair)
//              and the pair adjoints are  D = sum_g dgs_g ((1,u).X + (1,-u).Y),  U_k = sum_g gs_g (Y_k - X_k):
//              the 40 packed FMAs per pair and lane of conv_bwd_kernel collapse to ~6 VALU + a 16-lane DPP reduction,
//              abar_i += gs_g (1,-u)_c Sbar_jm[a,g,c] stays on the VALU (8 packed FMAs, the c-sum deferred to the epilogue),
//              and the force / virial tail runs once per CHUNK with lane = pair instead of once per pair in every lane.
//              Sbar rows are stored for this lane map ("T layout", written by unconcat_t_kernel): plane k = a >> 2 holds, for
//              lane (g,c), the four features a = 4k..4k+3 -> every row load is four contiguous 1 KiB wave loads.
#include <stdlib.h>

#include "conv_common.h"

namespace aimnet {


// ------------------------------------------------------------------------------------------------
// Neighbours staged per chunk: a multiple of the ring depth that covers a whole 5 A row of a molecular crystal, so the row
// pipeline below is filled once per centre atom; 6.7 KiB of LDS per wave keeps four blocks per CU.
constexpr int CHF = 72;
constexpr int RINGF = 8;  // neighbour rows in flight per wave (rolling register ring)
struct FwdMLds {
  float gs[CHF][G_];  // radial basis of the current chunk; reused as the epilogue scratch (768 + 48 NQ floats)
  float4 ud[CHF];
  int j[CHF];
  float fc[CHF];
  float qj[2][CHF];
};

// One wave per centre atom (the large-system form of conv_fwd_kernel; systems <= SPLIT_MAX_ATOMS keep the split VALU kernel).
template <int NQ>
__global__ __launch_bounds__(256, 4) void conv_fwd_mfma_kernel(const float* __restrict__ a, const float* __restrict__ a_t,
                                                               const int* __restrict__ row_of,
                                                               const float* __restrict__ q, const int* __restrict__ nb_idx,
                                                               const int* __restrict__ nb_cnt, const float4* __restrict__ pg, int cap,
                                                               const float* __restrict__ agh_a, const float* __restrict__ agh_q,
                                                               BasisParams bp, float* __restrict__ x, int ldx,
                                                               float* __restrict__ Vsave, float* __restrict__ Vqsave, int n_atoms,
                                                               const int* __restrict__ order) {
  __shared__ __attribute__((aligned(16))) FwdMLds wl[APB];
  __shared__ float s_agh[A_ * G_ * H_];
  constexpr bool HAS_Q = NQ > 0;
  constexpr int NQC = NQ > 0 ? NQ : 1;
  __shared__ float s_aghq[NQC * G_ * H_];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int k = threadIdx.x; k < A_ * G_ * H_; k += 256) s_agh[k] = agh_a[k];
  if (HAS_Q)
    for (int k = threadIdx.x; k < NQ * G_ * H_; k += 256) s_aghq[k] = agh_q[k];
  __syncthreads();
  FwdMLds& L = wl[wid];
  const int g = lane >> 2, c = lane & 3;
  const float fm0 = c == 0 ? 1.f : 0.f, fm1 = c == 1 ? 1.f : 0.f, fm2 = c == 2 ? 1.f : 0.f, fm3 = c == 3 ? 1.f : 0.f;
  // (B operands: float4 #lane of the neighbour's row in a_t = a_j transposed to [g][a] = features a = 4c..4c+3 of shift g)

  const AtomLoop al = atom_loop(n_atoms, APB);
  for (int i0 = al.first; i0 < al.last; i0 += al.step) {
    const bool live = i0 + wid < al.last;
    const int i = live ? (order ? order[i0 + wid] : i0 + wid) : 0;
    const int cnt = live ? nb_cnt[i] : 0;
    int cmax = cnt;
#pragma unroll
    for (int w = 0; w < APB; ++w) {
      const int iw = i0 + w;
      cmax = max(cmax, iw < al.last ? nb_cnt[order ? order[iw] : iw] : 0);
    }
    const int ri = row_of ? min(63, max(0, row_of[i])) : i;
    f32x4 acc[4];  // acc[e][c'] at lane (g, col) = S_i[a = 4 col + e, g, c']
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) acc[qq] = f32x4{0.f, 0.f, 0.f, 0.f};
    float accq[NQC];
#pragma unroll
    for (int ch = 0; ch < NQC; ++ch) accq[ch] = 0.0f;

    for (int c0 = 0; c0 < cmax; c0 += CHF) {
      const int nch = max(0, min(CHF, cnt - c0));
      __syncthreads();
      for (int sl = lane; sl < CHF; sl += 64) {
        if (sl < nch) {
          const size_t p = (size_t)i * cap + c0 + sl;
          const int j = nb_idx[p];
          L.j[sl] = row_of ? min(63, max(0, row_of[j])) : j;
          const float4 ud = pg[p];
          L.ud[sl] = ud;
          float dfc;
          L.fc[sl] = basis_fc(bp, ud.w, dfc);
#pragma unroll
          for (int ch = 0; ch < NQ; ++ch) L.qj[ch][sl] = q[(size_t)ch * n_atoms + j];
        } else {  // padding slots of the last ring round: the centre's own (valid, hot) row with zero weight
          L.j[sl] = ri;
          L.ud[sl] = make_float4(0.f, 0.f, 0.f, 1.f);
          L.fc[sl] = 0.f;
#pragma unroll
          for (int ch = 0; ch < NQ; ++ch) L.qj[ch][sl] = 0.f;
        }
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < CHF * G_ / 64; ++t) {
        const int e = lane + 64 * t;
        const int mm = e >> 4, gg = e & 15;
        float v = 0.0f;
        if (mm < nch) {
          const float dd = L.ud[mm].w - bp.shifts[gg];
          v = exp_neg(-bp.eta * dd * dd) * L.fc[mm];
        }
        L.gs[mm][gg] = v;
      }
      __syncthreads();
      // Rolling ring of RINGF neighbour rows in flight per wave, one coalesced dwordx4 wave load per row.  The kernel is bound
      // by memory-level parallelism, not by arithmetic: a pure gather of these 1 KiB rows runs in 26 us (tests/tools/
      // gather_probe.hip), the 4-rows-then-wait form of conv_fwd_kernel exposes one L2 latency per four pairs and takes 81.
      // Lane order must equal address order: the texture unit merges ADJACENT lanes only - a lane-permuted 1 KiB access (or four
      // dword loads per row) is 64 separate requests instead of 16 (+45 % on this kernel when it was tried).
      auto row = [&](int mm) {
        const int j = __builtin_amdgcn_readfirstlane(L.j[min(mm, CHF - 1)]);
        return reinterpret_cast<const float4*>(a_t + (size_t)j * NF)[lane];
      };
      auto use = [&](int mm, const float4& bv) {
        const float b[4] = {bv.x, bv.y, bv.z, bv.w};
        const float gv = L.gs[mm][g];
        const float4 u = L.ud[mm];
        const float w = gv * (fm0 + fm1 * u.x + fm2 * u.y + fm3 * u.z);
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) acc[qq] = mfma4(w, b[qq], acc[qq]);
        if (HAS_Q) {
#pragma unroll
          for (int ch = 0; ch < NQ; ++ch) accq[ch] += L.qj[ch][mm] * w;
        }
      };
      float4 r[RINGF];
#pragma unroll
      for (int sl = 0; sl < RINGF - 1; ++sl) {
        r[sl] = row(sl);
        __builtin_amdgcn_sched_barrier(0);  // keep the issue order: the loop's first wait is vmcnt(RINGF - 1) only if slot 0 went first
      }
      for (int m0 = 0; m0 < nch; m0 += RINGF) {
#pragma unroll
        for (int sl = 0; sl < RINGF; ++sl) {
          r[(sl + RINGF - 1) % RINGF] = row(m0 + sl + RINGF - 1);
          use(m0 + sl, r[sl]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    __syncthreads();
    // ---- epilogue: agh contraction + square-sum, assemble the MLP input row (as conv_fwd_kernel) -----------
    float* sv = &L.gs[0][0];  // sv[(a*16+g)*3 + k], 768 floats; svq at 768.. (48 floats per charge channel)
    if (live) {
      float* xr = x + (size_t)i * ldx;
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const int f = (4 * c + qq) * G_ + g;
        sv[f * 3 + 0] = acc[qq][1];
        sv[f * 3 + 1] = acc[qq][2];
        sv[f * 3 + 2] = acc[qq][3];
        xr[NF + f] = acc[qq][0];
      }
      if (HAS_Q && c != 0) {
#pragma unroll
        for (int ch = 0; ch < NQ; ++ch) sv[768 + ch * 48 + g * 3 + c - 1] = accq[ch];
      }
      reinterpret_cast<float4*>(xr)[lane] = reinterpret_cast<const float4*>(a + (size_t)ri * NF)[lane];
    }
    __syncthreads();
    if (live) {
      float* xr = x + (size_t)i * ldx;
#pragma unroll 1
      for (int t = 0; t < 3; ++t) {
        const int o = lane + 64 * t;  // (a, h) = (o / 12, o % 12)
        const int aa = o / H_, hh = o % H_;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
#pragma unroll 4
        for (int gg = 0; gg < G_; ++gg) {
          const float w = s_agh[(aa * G_ + gg) * H_ + hh];
          const float* s3 = &sv[(aa * G_ + gg) * 3];
          v0 += w * s3[0];
          v1 += w * s3[1];
          v2 += w * s3[2];
        }
        float* vs = Vsave + (size_t)i * (NV * 3) + o;
        vs[0] = v0; vs[NV] = v1; vs[2 * NV] = v2;
        xr[2 * NF + o] = v0 * v0 + v1 * v1 + v2 * v2;
      }
      if (HAS_Q) {
        const int c0 = 2 * NF + NV;  // 704
        if (lane < NQ) xr[c0 + lane] = q[(size_t)lane * n_atoms + i];
#pragma unroll
        for (int ch = 0; ch < NQ; ++ch) {
          if (c == 0) xr[c0 + NQ + ch * G_ + g] = accq[ch];
          if (lane < H_) {
            float v0 = 0.f, v1 = 0.f, v2 = 0.f;
#pragma unroll
            for (int gg = 0; gg < G_; ++gg) {
              const float w = s_aghq[(ch * G_ + gg) * H_ + lane];
              v0 += w * sv[768 + ch * 48 + gg * 3 + 0];
              v1 += w * sv[768 + ch * 48 + gg * 3 + 1];
              v2 += w * sv[768 + ch * 48 + gg * 3 + 2];
            }
            float* vs = Vqsave + ((size_t)i * NQ + ch) * (H_ * 3) + lane * 3;
            vs[0] = v0; vs[1] = v1; vs[2] = v2;
            xr[c0 + NQ + NQ * G_ + ch * H_ + lane] = v0 * v0 + v1 * v1 + v2 * v2;
          }
        }
        const int used = c0 + NQ * (1 + G_ + H_);
        if (lane < ldx - used) xr[used + lane] = 0.0f;
      } else {
        const int used = 2 * NF + NV;
        if (lane < ldx - used) xr[used + lane] = 0.0f;
      }
    }
  }
}

int launch_conv_fwd_mfma(hipStream_t s, int nq, const float* a, const float* a_t, const int* row_of, const float* q, const int* nb_idx,
                         const int* nb_cnt, const float4* pg, int cap, const float* agh_a, const float* agh_q, BasisParams bp,
                         float* x, int ldx, float* Vsave, float* Vqsave, int n_atoms, const int* order) {
  const int grid = min(ceil_div(n_atoms, APB), 256 * 8);
#define AIMNET_FWDM(HQ)                                                                                                        \
  hipLaunchKernelGGL((conv_fwd_mfma_kernel<HQ>), dim3(grid), dim3(256), 0, s, a, a_t, row_of, q, nb_idx, nb_cnt, pg, cap, agh_a, agh_q, \
                     bp, x, ldx, Vsave, Vqsave, n_atoms, order)
  if (nq == 2) AIMNET_FWDM(2);
  else if (nq == 1) AIMNET_FWDM(1);
  else AIMNET_FWDM(0);
#undef AIMNET_FWDM
  AIMNET_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// xbar -> Sbar in the T layout of conv_bwd_mfma_kernel: row i = 4 planes of 1 KiB, plane k, lane (g,c) = Sbar_i[4k..4k+3, g, c].
// Same arithmetic as unconcat_kernel (agh transposed contraction), mapped lane = (g, c): the c = 0 lanes copy the scalar block.
template <int NQ>
__global__ __launch_bounds__(256) void unconcat_t_kernel(const float* __restrict__ xbar, int ldx,
                                                        const float* __restrict__ Vsave, const float* __restrict__ Vqsave,
                                                        const float* __restrict__ agh_a, const float* __restrict__ agh_q,
                                                        float* __restrict__ SbarT, float* __restrict__ Sqbar, int n_atoms) {
  constexpr bool HAS_Q = NQ > 0;
  constexpr int NQC = NQ > 0 ? NQ : 1;
  __shared__ __attribute__((aligned(16))) float s_agh[A_ * G_ * H_];
  __shared__ float s_aghq[NQC * G_ * H_];
  __shared__ __attribute__((aligned(16))) float s_vb[APB][NV * 3 + NQC * H_ * 3];  // Vbar as three planes [k][a*12+h], then the q block
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int k = threadIdx.x; k < A_ * G_ * H_; k += 256) s_agh[k] = agh_a[k];
  if (HAS_Q)
    for (int k = threadIdx.x; k < NQ * G_ * H_; k += 256) s_aghq[k] = agh_q[k];
  float* vb = s_vb[wid];
  const int g = lane >> 2, c = lane & 3;
  const int cm = c > 0 ? c - 1 : 0;
  const AtomLoop al = atom_loop(n_atoms, APB);
  for (int i0 = al.first; i0 < al.last; i0 += al.step) {
    const int i = i0 + wid;
    const bool live = i < al.last;
    __syncthreads();
    if (live) {
      const float* xr = xbar + (size_t)i * ldx;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int o = lane + 64 * t;
        const float f = 2.0f * xr[2 * NF + o];
        const float* vs = Vsave + (size_t)i * (NV * 3) + o;
        vb[o] = f * vs[0];
        vb[NV + o] = f * vs[NV];
        vb[2 * NV + o] = f * vs[2 * NV];
      }
      if (HAS_Q && lane < H_) {
#pragma unroll
        for (int ch = 0; ch < NQ; ++ch) {
          const float f = 2.0f * xr[2 * NF + NV + NQ + NQ * G_ + ch * H_ + lane];
          const float* vs = Vqsave + ((size_t)i * NQ + ch) * (H_ * 3) + lane * 3;
          vb[NV * 3 + ch * (H_ * 3) + lane * 3 + 0] = f * vs[0];
          vb[NV * 3 + ch * (H_ * 3) + lane * 3 + 1] = f * vs[1];
          vb[NV * 3 + ch * (H_ * 3) + lane * 3 + 2] = f * vs[2];
        }
      }
    }
    __syncthreads();
    if (live) {
      const float* xr = xbar + (size_t)i * ldx;
      float4* out = reinterpret_cast<float4*>(SbarT + (size_t)i * (NF * 4)) + lane;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int aa = 4 * k + e;
          const float4* wp = reinterpret_cast<const float4*>(&s_agh[(aa * G_ + g) * H_]);
          const float4* vp = reinterpret_cast<const float4*>(&vb[cm * NV + aa * H_]);
          const float4 w0 = wp[0], w1 = wp[1], w2 = wp[2], v0 = vp[0], v1 = vp[1], v2 = vp[2];
          float d = w0.x * v0.x;
          d = fmaf(w0.y, v0.y, d); d = fmaf(w0.z, v0.z, d); d = fmaf(w0.w, v0.w, d);
          d = fmaf(w1.x, v1.x, d); d = fmaf(w1.y, v1.y, d); d = fmaf(w1.z, v1.z, d); d = fmaf(w1.w, v1.w, d);
          d = fmaf(w2.x, v2.x, d); d = fmaf(w2.y, v2.y, d); d = fmaf(w2.z, v2.z, d); d = fmaf(w2.w, v2.w, d);
          const float s0 = xr[NF + aa * G_ + g];
          o[e] = c == 0 ? s0 : d;
        }
        out[k * 64] = make_float4(o[0], o[1], o[2], o[3]);
      }
      if (HAS_Q) {
#pragma unroll
        for (int ch = 0; ch < NQ; ++ch) {
          float v;
          if (c == 0) {
            v = xr[2 * NF + NV + NQ + ch * G_ + g];
          } else {
            v = 0.f;
#pragma unroll
            for (int h = 0; h < H_; ++h) v += s_aghq[(ch * G_ + g) * H_ + h] * vb[NV * 3 + ch * (H_ * 3) + h * 3 + c - 1];
          }
          Sqbar[((size_t)i * NQ + ch) * (G_ * 4) + lane] = v;
        }
      }
    }
  }
}

int launch_unconcat_t(hipStream_t s, int nq, const float* xbar, int ldx, const float* Vsave, const float* Vqsave,
                      const float* agh_a, const float* agh_q, float* SbarT, float* Sqbar, int n_atoms) {
  const int grid = min(ceil_div(n_atoms, APB), 256 * 8);
  if (nq == 2)
    hipLaunchKernelGGL(unconcat_t_kernel<2>, dim3(grid), dim3(256), 0, s, xbar, ldx, Vsave, Vqsave, agh_a, agh_q, SbarT, Sqbar, n_atoms);
  else if (nq == 1)
    hipLaunchKernelGGL(unconcat_t_kernel<1>, dim3(grid), dim3(256), 0, s, xbar, ldx, Vsave, Vqsave, agh_a, agh_q, SbarT, Sqbar, n_atoms);
  else
    hipLaunchKernelGGL(unconcat_t_kernel<0>, dim3(grid), dim3(256), 0, s, xbar, ldx, Vsave, Vqsave, agh_a, agh_q, SbarT, Sqbar, n_atoms);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Neighbours staged per chunk (a multiple of 4: X is formed for 4 neighbours at a time).  80 covers a whole 5 A row of a
// molecular crystal, so the load pipeline below is filled once per centre atom; 17.9 KiB of LDS per wave -> two blocks per CU,
// which is also what the register budget wants: the kernel is bound by the BYTES IN FLIGHT towards L2 (Little's law: the
// VALU kernel keeps one 5 KiB row per wave in flight, 80 KiB per CU, and sits at 16 TB/s = 80 KiB x 256 / 1.2 us), so it runs
// 2 waves per SIMD with 256 VGPRs and a ring of four Sbar rows per wave instead of 4 waves with one.
constexpr int CHM = 80;
constexpr int RING = 4;
struct BwdMLds {
  float gs[CHM][G_];
  float dgs[CHM][G_];
  float4 ud[CHM];
  int j[CHM];      // neighbour atom
  int jr[CHM];     // its feature row (atomic number in pass 0)
  float qj[2][CHM];
  float fc[CHM], dfc[CHM];
  float4 red[CHM][4];  // per pair: the four 16-lane row partials of (D, U0, U1, U2)
};

template <int NQ, bool NEED_ABAR, bool STRESS>
__global__ __launch_bounds__(256, 2) void conv_bwd_mfma_kernel(const float* __restrict__ a_t, const int* __restrict__ row_of,
                                                               const float* __restrict__ q, const float* __restrict__ SbarT,
                                                               const float* __restrict__ Sqbar, const int* __restrict__ nb_idx,
                                                               const int* __restrict__ nb_cnt, const float4* __restrict__ pg, int cap,
                                                               BasisParams bp, const float* __restrict__ xbar, int ldx,
                                                               const float* __restrict__ abar_in, float* __restrict__ abar_out,
                                                               const float* __restrict__ qbar_in, float* __restrict__ qbar_out,
                                                               float* __restrict__ fgrad, float* __restrict__ virial_atom,
                                                               int n_atoms, const int* __restrict__ order) {
  constexpr bool HAS_Q = NQ > 0;
  constexpr int NQC = NQ > 0 ? NQ : 1;
  __shared__ __attribute__((aligned(16))) BwdMLds wl[APB];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  BwdMLds& L = wl[wid];
  const int g = lane >> 2, c = lane & 3;
  const float m0 = c == 0 ? 1.f : 0.f, m1 = c == 1 ? 1.f : 0.f, m2 = c == 2 ? 1.f : 0.f, m3 = c == 3 ? 1.f : 0.f;
  const float m0x2 = 2.0f * m0;
  const bool wr_lane = (lane & 12) == 12;  // lanes 12..15 of every 16-lane row hold that row's (D, U0, U1, U2) partials

  const AtomLoop al = atom_loop(n_atoms, APB);
  for (int i0 = al.first; i0 < al.last; i0 += al.step) {
    const bool live = i0 + wid < al.last;
    // wave-uniform scalars are forced into SGPRs: the branches below become scalar and the row addresses need no VGPRs
    const int i = __builtin_amdgcn_readfirstlane(live ? (order ? order[i0 + wid] : i0 + wid) : 0);
    const int cnt = __builtin_amdgcn_readfirstlane(live ? nb_cnt[i] : 0);
    int cmax = cnt;
#pragma unroll
    for (int w = 0; w < APB; ++w) {
      const int iw = i0 + w;
      cmax = max(cmax, iw < al.last ? nb_cnt[order ? order[iw] : iw] : 0);
    }
    cmax = __builtin_amdgcn_readfirstlane(cmax);
    // centre atom: a_i[a][g] (A operand of Y, the same for the four rows) and Sbar_i[a,g,c] (B operand of X)
    const int ri = __builtin_amdgcn_readfirstlane(row_of ? min(63, max(0, row_of[i])) : i);
    float ai[A_], Si[A_];
    {
      const float4* ap = reinterpret_cast<const float4*>(a_t + (size_t)ri * NF) + 4 * g;  // float4 4g + k = a_i[4k..4k+3][g]
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float4 v = live ? ap[k] : make_float4(0.f, 0.f, 0.f, 0.f);
        ai[4 * k] = v.x; ai[4 * k + 1] = v.y; ai[4 * k + 2] = v.z; ai[4 * k + 3] = v.w;
      }
      const float4* sp = reinterpret_cast<const float4*>(SbarT + (size_t)i * (NF * 4)) + lane;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float4 v = live ? sp[k * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
        Si[4 * k] = v.x; Si[4 * k + 1] = v.y; Si[4 * k + 2] = v.z; Si[4 * k + 3] = v.w;
      }
    }
    float qi[NQC], Sqi[NQC];
#pragma unroll
    for (int ch = 0; ch < NQC; ++ch) qi[ch] = Sqi[ch] = 0.0f;
    if (live) {
#pragma unroll
      for (int ch = 0; ch < NQ; ++ch) {
        qi[ch] = q[(size_t)ch * n_atoms + i];
        Sqi[ch] = Sqbar[((size_t)i * NQ + ch) * (G_ * 4) + lane];
      }
    }
    f2 ab[8];  // abar_i[a = 2h, 2h+1][g], the lane's c-term only (summed over the quad in the epilogue)
#pragma unroll
    for (int h = 0; h < 8; ++h) ab[h] = mk2(0.f, 0.f);
    float qacc[NQC];
#pragma unroll
    for (int ch = 0; ch < NQC; ++ch) qacc[ch] = 0.0f;
    float xa0 = 0.f, xa1 = 0.f, xa2 = 0.f;  // chunk tail, lane = pair
    float W[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) W[k] = 0.0f;

    for (int c0 = 0; c0 < cmax; c0 += CHM) {
      const int nch = max(0, min(CHM, cnt - c0));
      __syncthreads();
      for (int sl = lane; sl < CHM; sl += 64) {
        if (sl < nch) {
          const size_t p = (size_t)i * cap + c0 + sl;
          const int j = nb_idx[p];
          L.j[sl] = j;
          L.jr[sl] = row_of ? min(63, max(0, row_of[j])) : j;
          const float4 ud = pg[p];
          L.ud[sl] = ud;
          float dfc;
          L.fc[sl] = basis_fc(bp, ud.w, dfc);
          L.dfc[sl] = dfc;
#pragma unroll
          for (int ch = 0; ch < NQ; ++ch) L.qj[ch][sl] = q[(size_t)ch * n_atoms + j];
        } else {  // padding slots of the last group of 4: the centre's own (valid, cache-hot) rows with zero weight
          L.j[sl] = i;
          L.jr[sl] = ri;
          L.ud[sl] = make_float4(0.f, 0.f, 0.f, 1.f);
          L.fc[sl] = 0.f;
          L.dfc[sl] = 0.f;
#pragma unroll
          for (int ch = 0; ch < NQ; ++ch) L.qj[ch][sl] = 0.f;
        }
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < CHM * G_ / 64; ++t) {
        const int e = lane + 64 * t;
        const int mm = e >> 4, gg = e & 15;
        float v = 0.0f, dv = 0.0f;
        if (mm < nch) {
          const float fc = L.fc[mm], dfc = L.dfc[mm];
          const float dd = L.ud[mm].w - bp.shifts[gg];
          const float Gg = exp_neg(-bp.eta * dd * dd);
          v = Gg * fc;
          dv = Gg * (dfc - 2.0f * bp.eta * dd * fc);
        }
        L.gs[mm][gg] = v;
        L.dgs[mm][gg] = dv;
      }
      __syncthreads();

      const int ngrp = (nch + 3) >> 2;
      float aj[A_];  // A operand of X: lane (g, r) holds a_{j_r}[a][g] of the group's r-th neighbour
      auto load_aj = [&](int grp) {
        const int e = min(4 * grp + c, CHM - 1);
        const float4* p = reinterpret_cast<const float4*>(a_t + (size_t)L.jr[e] * NF) + 4 * g;  // the lane's 64 B: a_j[0..15][g]
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float4 v = p[k];
          aj[4 * k] = v.x; aj[4 * k + 1] = v.y; aj[4 * k + 2] = v.z; aj[4 * k + 3] = v.w;
        }
      };
      auto x_chain = [&]() {
        f32x4 x0 = f32x4{0.f, 0.f, 0.f, 0.f}, x1 = x0;
#pragma unroll
        for (int aa = 0; aa < A_; aa += 2) {
          x0 = mfma4(aj[aa], Si[aa], x0);
          x1 = mfma4(aj[aa + 1], Si[aa + 1], x1);
        }
        return x0 + x1;  // [r] at lane (g,c) = X_r[g,c]
      };
      float4 S[RING][4];
      float sq[RING][NQC];
      auto load_S = [&](int e, int slot) {
        // past the chunk's last group the index clamps to a padding slot = the centre's own (hot) row, weight zero
        const int jn = __builtin_amdgcn_readfirstlane(L.j[min(e, CHM - 1)]);
        const float4* sp = reinterpret_cast<const float4*>(SbarT + (size_t)jn * (NF * 4)) + lane;
        S[slot][0] = sp[0]; S[slot][1] = sp[64]; S[slot][2] = sp[128]; S[slot][3] = sp[192];
#pragma unroll
        for (int ch = 0; ch < NQC; ++ch) sq[slot][ch] = HAS_Q ? Sqbar[((size_t)jn * NQ + ch) * (G_ * 4) + lane] : 0.0f;
      };
      auto process = [&](int e, int slot, float Xe) {
        const float Sj[A_] = {S[slot][0].x, S[slot][0].y, S[slot][0].z, S[slot][0].w, S[slot][1].x, S[slot][1].y,
                              S[slot][1].z, S[slot][1].w, S[slot][2].x, S[slot][2].y, S[slot][2].z, S[slot][2].w,
                              S[slot][3].x, S[slot][3].y, S[slot][3].z, S[slot][3].w};
        f32x4 y0 = f32x4{0.f, 0.f, 0.f, 0.f}, y1 = y0;
#pragma unroll
        for (int aa = 0; aa < A_; aa += 2) {
          y0 = mfma4(ai[aa], Sj[aa], y0);
          y1 = mfma4(ai[aa + 1], Sj[aa + 1], y1);
        }
        float Y = y0[0] + y1[0];  // Y[g,c] (every row of the block holds the same value)
        const float gv = L.gs[e][g], dgv = L.dgs[e][g];
        const float4 u = L.ud[e];
        const float uc = m0 + m1 * u.x + m2 * u.y + m3 * u.z;  // (1, u)_c
        const float ucm = m0x2 - uc;                             // (1, -u)_c
        const float w = gv * ucm;
        if (NEED_ABAR) {
#pragma unroll
          for (int h = 0; h < 8; ++h) ab[h] += w * mk2(Sj[2 * h], Sj[2 * h + 1]);
        }
        float X = Xe;
        if (HAS_Q) {
#pragma unroll
          for (int ch = 0; ch < NQ; ++ch) {
            X += L.qj[ch][e] * Sqi[ch];
            Y += qi[ch] * sq[slot][ch];
            qacc[ch] += w * sq[slot][ch];
          }
        }
        float Dl = dgv * (uc * X + ucm * Y);
        const float Vl = gv * (Y - X);
        Dl += dpp0<0xB1>(Dl);  // quad sum: the pair's D gets a term from every component c
        Dl += dpp0<0x4E>(Dl);
        float Z = c == 0 ? Dl : Vl;  // lane (g,0): D of shift g;  lane (g,c>0): U_{c-1} of shift g
        Z += dpp0<0x114>(Z);         // row_shr:4, row_shr:8: lanes 12..15 of each row = sums over the row's four shifts
        Z += dpp0<0x118>(Z);
        if (wr_lane) reinterpret_cast<float*>(&L.red[e][lane >> 4])[c] = Z;
      };

      load_aj(0);
      f32x4 Xc = x_chain();
      load_aj(1);
      load_S(0, 0); load_S(1, 1); load_S(2, 2);
      for (int t = 0; t < ngrp; ++t) {
        const int e = 4 * t;
        // (scheduling fences: without them the compiler hoists the LDS reads and row loads of all four entries to the
        // top of the body and the live ranges no longer fit 256 VGPRs)
        load_S(e + 3, 3); process(e, 0, Xc[0]);
        __builtin_amdgcn_sched_barrier(0);
        load_S(e + 4, 0); process(e + 1, 1, Xc[1]);
        __builtin_amdgcn_sched_barrier(0);
        load_S(e + 5, 1); process(e + 2, 2, Xc[2]);
        __builtin_amdgcn_sched_barrier(0);
        load_S(e + 6, 2); process(e + 3, 3, Xc[3]);
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 Xn = x_chain();  // group t+1 (aj was loaded one group ahead)
        load_aj(t + 2);
        Xc = Xn;
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- chunk tail, lane = pair: both directions of the pair enter dE/dx_i through D = dbar_ij + dbar_ji and
      // U = ubar_ji - ubar_ij:  dE/dx_i += -D u + (U - (U.u) u) / d;  virial of the two ordered pairs: -1/2 r_ij (x) (that)
      __builtin_amdgcn_wave_barrier();
      for (int sl = lane; sl < nch; sl += 64) {
        const float4 r0 = L.red[sl][0], r1 = L.red[sl][1], r2 = L.red[sl][2], r3 = L.red[sl][3];
        const float D = (r0.x + r1.x) + (r2.x + r3.x), U0 = (r0.y + r1.y) + (r2.y + r3.y);
        const float U1 = (r0.z + r1.z) + (r2.z + r3.z), U2 = (r0.w + r1.w) + (r2.w + r3.w);
        const float4 u = L.ud[sl];
        const float inv_d = __builtin_amdgcn_rcpf(u.w);
        const float dot = U0 * u.x + U1 * u.y + U2 * u.z;
        const float f0 = (U0 - dot * u.x) * inv_d - D * u.x;
        const float f1 = (U1 - dot * u.y) * inv_d - D * u.y;
        const float f2v = (U2 - dot * u.z) * inv_d - D * u.z;
        xa0 += f0; xa1 += f1; xa2 += f2v;
        if (STRESS) {
          const float hx = -0.5f * u.x * u.w, hy = -0.5f * u.y * u.w, hz = -0.5f * u.z * u.w;
          W[0] += hx * f0; W[1] += hx * f1; W[2] += hx * f2v;
          W[3] += hy * f0; W[4] += hy * f1; W[5] += hy * f2v;
          W[6] += hz * f0; W[7] += hz * f1; W[8] += hz * f2v;
        }
      }
    }
    // ---- epilogue ---------------------------------------------------------------------------
    xa0 = wave_sum(xa0); xa1 = wave_sum(xa1); xa2 = wave_sum(xa2);
#pragma unroll
    for (int ch = 0; ch < NQ; ++ch) qacc[ch] = wave_sum(qacc[ch]);
    if (STRESS) {
#pragma unroll
      for (int k = 0; k < 9; ++k) W[k] = wave_sum(W[k]);
    }
    if (live) {
      if (NEED_ABAR) {
        float abs_[A_];
#pragma unroll
        for (int h = 0; h < 8; ++h) {
          float v0 = ab[h].x, v1 = ab[h].y;
          v0 += dpp0<0xB1>(v0); v0 += dpp0<0x4E>(v0);
          v1 += dpp0<0xB1>(v1); v1 += dpp0<0x4E>(v1);
          abs_[2 * h] = v0; abs_[2 * h + 1] = v1;
        }
        // lane (g,c) writes the features a = 4k + c (64 consecutive floats per k over the wave)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float v = c == 0 ? abs_[4 * k] : c == 1 ? abs_[4 * k + 1] : c == 2 ? abs_[4 * k + 2] : abs_[4 * k + 3];
          const int f = (4 * k + c) * G_ + g;
          float o = v + xbar[(size_t)i * ldx + f];
          if (abar_in) o += abar_in[(size_t)i * NF + f];
          abar_out[(size_t)i * NF + f] = o;
        }
      }
      if (lane == 0) {
        fgrad[3 * i + 0] += xa0;
        fgrad[3 * i + 1] += xa1;
        fgrad[3 * i + 2] += xa2;
#pragma unroll
        for (int ch = 0; ch < NQ; ++ch)
          qbar_out[(size_t)ch * n_atoms + i] = qbar_in[(size_t)ch * n_atoms + i] + xbar[(size_t)i * ldx + 2 * NF + NV + ch] + qacc[ch];
      }
      if (STRESS && lane < 9) {
        float v = W[0];
#pragma unroll
        for (int k = 1; k < 9; ++k) v = (lane == k) ? W[k] : v;
        virial_atom[(size_t)i * 9 + lane] += v;
      }
    }
  }
}

int launch_conv_bwd_mfma(hipStream_t s, int nq, bool need_abar, bool stress, const float* a_t, const int* row_of, const float* q,
                         const float* SbarT, const float* Sqbar, const int* nb_idx, const int* nb_cnt, const float4* pg, int cap,
                         BasisParams bp, const float* xbar, int ldx, const float* abar_in, float* abar_out, const float* qbar_in,
                         float* qbar_out, float* fgrad, float* virial_atom, int n_atoms, const int* order) {
  const int grid = min(ceil_div(n_atoms, APB), 256 * 8);
#define AIMNET_BWDM(HQ, NA, ST)                                                                                               \
  hipLaunchKernelGGL((conv_bwd_mfma_kernel<HQ, NA, ST>), dim3(grid), dim3(256), 0, s, a_t, row_of, q, SbarT, Sqbar, nb_idx, nb_cnt, \
                     pg, cap, bp, xbar, ldx, abar_in, abar_out, qbar_in, qbar_out, fgrad, virial_atom, n_atoms, order)
#define AIMNET_BWDM2(HQ)                                                 \
  do {                                                                   \
    if (need_abar) { if (stress) AIMNET_BWDM(HQ, true, true); else AIMNET_BWDM(HQ, true, false); } \
    else { if (stress) AIMNET_BWDM(HQ, false, true); else AIMNET_BWDM(HQ, false, false); }         \
  } while (0)
  if (nq == 2) AIMNET_BWDM2(2);
  else if (nq == 1) AIMNET_BWDM2(1);
  else AIMNET_BWDM2(0);
#undef AIMNET_BWDM2
#undef AIMNET_BWDM
  AIMNET_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Lane-layout probe of v_mfma_f32_4x4x1_16B_f32 (tests/test_gpu_ops.py): out[lb][r][l] = D_r[l] for A[l] = l + 1, B = one-hot(lb).
__global__ void mfma4_probe_kernel(float* __restrict__ out) {
  const int lane = threadIdx.x;
  for (int lb = 0; lb < 64; ++lb) {
    const f32x4 d = mfma4((float)(lane + 1), lane == lb ? 1.0f : 0.0f, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(lb * 4 + r) * 64 + lane] = d[r];
  }
}

int launch_mfma4_probe(hipStream_t s, float* out) {
  hipLaunchKernelGGL(mfma4_probe_kernel, dim3(1), dim3(64), 0, s, out);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

}  // namespace aimnet
