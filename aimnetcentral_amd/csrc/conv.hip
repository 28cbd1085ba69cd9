// conv.hip - AEV radial basis and the ConvSV gather-contract kernels (fwd + bwd).
//
// Reference semantics (paths relative to /root/reference/aimnet):
//   pair geometry (u, d) per list entry: written by the neighbour-list builders (nlist.hip, ops.calc_distances ops.py:37-66)
//   radial basis AEVSV._calc_aev modules/aev.py:94-110: gs_g = exp(-eta (d-s_g)^2) * 0.5 (cos(pi d/rc)+1)
//   conv_fwd    ConvSV.forward aev.py:156-189 == Warp kernel kernels/conv_sv_2d_sp_wp.py:90-112 plus the
//               agh contraction + square-sum, for conv_a (d2features) and conv_q in one launch;
//               also assembles the MLP input row of aimnet2.py:108-120,163-164
//   conv_bwd    Warp backward_a / backward_g kernels conv_sv_2d_sp_wp.py:115-164 fused with the AEV
//               backward, in CENTRE-MAJOR form: the neighbour matrix is full and exactly symmetric, so
//               every scatter into j is rewritten as a gather at i with u_ji = -u_ij (SURVEY.md App. A
//               step 4; oracle/aimnet2_analytic.py is the executable specification) - deterministic,
//               no atomics, and g_sv (N,M,16,4) is never materialised (recomputed from (u,d) in LDS).
//
// Mapping: one wave per centre atom, 4 atoms per 256-thread block, persistent grid-stride blocks.
// Lane l owns feature a = l>>2 and the four shifts g = 4*(l&3)..+3, i.e. the 16 B at a_j[4l..4l+3]:
// a neighbour's 1 KiB feature row is one coalesced dwordx4 wave load.  Neighbours are processed in
// chunks of 64 whose (j, u, d) and radial basis tables are staged in LDS once per chunk (16 exp
// per pair in total instead of per lane).
#include <stdlib.h>

#include "conv_common.h"
#include "gemm_h2_common.h"
#include "pairmap.h"

namespace aimnet {

// LDS hand-off between the lanes of ONE wave (the staging buffers of the one-wave-per-atom kernels are private to the wave): LDS
// operations of a wave execute in order, so only the compiler has to be kept from reordering - no s_barrier.  With block
// barriers the four waves of a block (four atoms with 60-80 neighbours each) ran every chunk in lock step: each block iteration
// cost the slowest atom, and the waves' memory phases could not drift apart (-2 to -4 % on the two kernels).  SPLIT kernels share
// data between waves and keep the real barrier.
// (lds_sync<BLOCK>() itself lives in conv_common.h)

// ------------------------------------------------------------------------------------------------
// per-wave LDS scratch of the forward kernel
struct FwdWaveLds {
  float gs[64][G_];   // radial basis of the current chunk (first CH rows); reused as 816-float epilogue scratch
  float4 ud[CH];      // (ux, uy, uz, d)
  int j[CH];
  float fc[CH];       // cutoff envelope fc(d) of the chunk (one sincos per pair)
  float qj[2][CH];    // neighbour charges, one row per charge channel
  float mt[64];       // P0M: one species' moments M[g][c], handed from the (g,c) lane map to the (a, 4 shifts) one
  int rz[CH], re[CH]; // P0M: the chunk's runs of equal elements: element, end slot
};

// `row_of` (may be NULL): feature row of atom j inside `a`.  Pass 0 gathers straight from the
// embedding table (a = afv, row_of = atomic numbers): the whole "feature table" is then a few KiB
// that live in L1 instead of a 1 KiB-per-pair trip to L2 / Infinity Cache.
// SPLIT (small systems, <= 1024 atoms): the four waves of a block share ONE centre atom and a quarter of its neighbour
// row each, partial sums meet in LDS and wave 0 runs the epilogue.  A 113-atom molecule otherwise occupies 113 waves
// that each walk ~31 dependent gathers; split four ways the chain is 8 long and 4x as many waves hide its latency.
// NQ: charge channels convolved alongside the features: 0 (pass 0), 1, or 2 (NSE models; q = planes [NQ][n_atoms],
// agh_q [NQ][G][H], row layout [q (NQ) | S^q_s (NQ x 16) | |V^q|^2 (NQ x 12)] as ConvSV(nchannel=NQ) emits it, aev.py:188).
// one wave per atom: four blocks per CU = 4 waves per SIMD (128 VGPRs; the kernel is latency-bound and gains 6 % over three);
// SPLIT holds three blocks' worth of LDS per CU anyway
// P0M (pass 0, one wave per atom, NQ = 0): the neighbour features are rows of the embedding table, a_j = afv[Z_j], so
//   S_i[a,g,c] = sum_z afv[z][a,g] M_i[z][g,c],   M_i[z][g,c] = sum_{j: Z_j = z} gs_g(d_ij) (1,u_ij)_c :
// the chunk's neighbours are ordered by element, the pair loop accumulates ONE number per lane (g,c) and pair (no row gather,
// no 16 FMAs per lane), and each element present costs one 16-FMA flush.  The forward twin of the species-moment backward.
#ifndef AIMNET_PROBE_FWD_OCC
#define AIMNET_PROBE_FWD_OCC 4
#endif
// X3: the MLP input row is written pre-split instead of fp32 - 1: for gemm_bf3a.hip ("bf3" layout: per 32 columns [plane 0][plane 1]
// [plane 2] x 32 bf16, fp32 == p0 + p1 + p2 exactly; `x` then points at bf16 elements, 3 * ldx per row); 2: for gemm_h2.hip ("h2"
// layout, activation form: per 32 columns [hi][lo] x 32 fp16, 2 * ldx elements per row; gemm_h2_common.h).
template <int NQ, bool SPLIT, bool P0M = false, int X3 = 0>
__global__ __launch_bounds__(256, SPLIT ? 3 : AIMNET_PROBE_FWD_OCC) void conv_fwd_kernel(const float* __restrict__ a, const int* __restrict__ row_of,
                                                      const float* __restrict__ q,
                                                      const int* __restrict__ nb_idx, const int* __restrict__ nb_cnt,
                                                      const float4* __restrict__ pg, int cap,
                                                      const float* __restrict__ agh_a, const float* __restrict__ agh_q,
                                                      BasisParams bp, float* __restrict__ x, int ldx,
                                                      float* __restrict__ Vsave, float* __restrict__ Vqsave, int n_atoms,
                                                      const int* __restrict__ order) {
  __shared__ __attribute__((aligned(16))) FwdWaveLds wl[APB];
  __shared__ float s_agh[A_ * G_ * H_];
  constexpr bool HAS_Q = NQ > 0;
  constexpr int NQC = NQ > 0 ? NQ : 1;
  constexpr int UNR_T = SPLIT ? 3 : 1, UNR_G = SPLIT ? 16 : 4;  // unroll factors of the agh contraction in the epilogue
  __shared__ float s_aghq[NQC * G_ * H_];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  // agh table transposed to [g][a * H + h]: the epilogue's lanes walk (a, h) at fixed g, so consecutive lanes read consecutive
  // words (in the [a][g][h] order of the weights a lane's address stride was 16 words per h-block: 3-6-way bank conflicts on 48 reads
  // per atom)
  for (int k = threadIdx.x; k < A_ * G_ * H_; k += 256) {
    const int aa = k / (G_ * H_), g = (k / H_) % G_, hh = k % H_;
    s_agh[g * (A_ * H_) + aa * H_ + hh] = agh_a[k];
  }
  if (HAS_Q)
    for (int k = threadIdx.x; k < NQ * G_ * H_; k += 256) s_aghq[k] = agh_q[k];
  __syncthreads();
  FwdWaveLds& L = wl[wid];
  const int g4 = (lane & 3) * 4;
  const float shift_l = bp.shifts[lane & 15];  // the table loops below give lane l the shift l & 15 in every iteration
  __shared__ float s_red[SPLIT ? 3 * (16 + NQC) * 64 : 1];  // SPLIT: accumulators of waves 1..3

  const AtomLoop al = atom_loop(n_atoms, SPLIT ? 1 : APB);
  // One-wave-per-atom form: the id, the row count and the first chunk's neighbour ids / pair geometry of the wave's NEXT atom
  // are requested before the epilogue of the current one, so the dependent chain order -> row -> neighbours (three L2 / HBM
  // latencies per atom that nothing else in the wave can hide) runs under the epilogue's ~700 instructions.
  int pf_i = 0, pf_cnt = 0, pf_j = 0;
  bool pf_live = false;
  float4 pf_ud = make_float4(0.f, 0.f, 0.f, 1.f);
  auto prefetch_atom = [&](int i0n) {
    pf_live = i0n < al.last && i0n + wid < al.last;
    pf_i = __builtin_amdgcn_readfirstlane(pf_live ? (order ? order[i0n + wid] : i0n + wid) : 0);
    pf_cnt = pf_live ? nb_cnt[pf_i] : 0;
    if (pf_live && lane < cap) {
      pf_j = nb_idx[(size_t)pf_i * cap + lane];
      pf_ud = pg[(size_t)pf_i * cap + lane];
    }
  };
  if (!SPLIT) prefetch_atom(al.first);
  for (int i0 = al.first; i0 < al.last; i0 += al.step) {
    const int aslot = SPLIT ? 0 : wid;  // which atom of this block iteration the wave works on
    const bool live_atom = SPLIT ? i0 + aslot < al.last : pf_live;
    const int i = SPLIT ? (live_atom ? (order ? order[i0 + aslot] : i0 + aslot) : 0) : pf_i;  // `order`: spatially sorted processing order
    const int cnt_all = SPLIT ? (live_atom ? nb_cnt[i] : 0) : __builtin_amdgcn_readfirstlane(pf_cnt);
    const int cur_j = pf_j;
    const float4 cur_ud = pf_ud;
    int m_lo = 0, cnt = cnt_all;
    int cmax = cnt;  // SPLIT: block-uniform trip count so that __syncthreads() is legal; otherwise the wave's own count
    if (SPLIT) {
      const int q4 = (cnt_all + 3) >> 2;
      m_lo = wid * q4;
      cnt = max(0, min(q4, cnt_all - m_lo));
      cmax = q4;
    }
    f2 acc[4][2];  // [component c][shift pair]: 2-wide vectors -> v_pk_fma_f32
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c][0] = acc[c][1] = mk2(0.f, 0.f);
    float accq[NQC];
#pragma unroll
    for (int ch = 0; ch < NQC; ++ch) accq[ch] = 0.0f;

    int p0_runs = 0;  // P0M: runs of equal elements in the staged chunk (wave-uniform)
    for (int c0 = 0; c0 < cmax; c0 += CH) {
      const int nch = max(0, min(CH, cnt - c0));
      lds_sync<SPLIT>();  // previous chunk fully consumed
      if (P0M) {
        // stage the chunk ordered by element: one round per distinct element present (3-5 in organic crystals)
        const bool valid = lane < nch;
        int zj = 0;
        float4 ud = make_float4(0.f, 0.f, 0.f, 1.f);
        float fcv = 0.f;
        if (valid) {
          const size_t p = (size_t)i * cap + m_lo + c0 + lane;
          const bool pre = !SPLIT && c0 == 0;  // the first chunk was requested one atom ahead
          zj = min(63, max(0, row_of[pre ? cur_j : nb_idx[p]]));
          ud = pre ? cur_ud : pg[p];
          float dfc;
          fcv = basis_fc(bp, ud.w, dfc);
        }
        unsigned long long rem = __ballot(valid);
        const unsigned long long lt = (1ull << lane) - 1ull;
        int base = 0, pos = lane;
        p0_runs = 0;
        while (rem) {  // wave-uniform
          const int zs = __shfl(zj, __ffsll((long long)rem) - 1, 64);
          const bool mine = valid && zj == zs;
          const unsigned long long m = __ballot(mine);
          if (mine) pos = base + __popcll(m & lt);
          base += __popcll(m);
          if (lane == 0) {
            L.rz[p0_runs] = zs;
            L.re[p0_runs] = base;
          }
          ++p0_runs;
          rem &= ~m;
        }
        if (!valid) pos = lane;  // slots nch..63 keep a defined (zero-weight) entry
        L.j[pos] = zj;
        L.ud[pos] = ud;
        L.fc[pos] = fcv;
      } else if (lane < nch) {
        const size_t p = (size_t)i * cap + m_lo + c0 + lane;
        const bool pre = !SPLIT && c0 == 0;  // the first chunk was requested one atom ahead
        const int j = pre ? cur_j : nb_idx[p];
        L.j[lane] = row_of ? min(63, max(0, row_of[j])) : j;
        const float4 ud = pre ? cur_ud : pg[p];
        L.ud[lane] = ud;
        float dfc;
        L.fc[lane] = basis_fc(bp, ud.w, dfc);
#pragma unroll
        for (int ch = 0; ch < NQ; ++ch) L.qj[ch][lane] = q[(size_t)ch * n_atoms + j];
      } else {  // slots past the row end: a valid row with zero weight (the pipelined loop rounds up to 4)
        L.j[lane] = 0;
        L.ud[lane] = make_float4(0.f, 0.f, 0.f, 1.f);
        L.fc[lane] = 0.f;
#pragma unroll
        for (int ch = 0; ch < NQ; ++ch) L.qj[ch][lane] = 0.f;
      }
      lds_sync<SPLIT>();
      // four pairs per step, four steps per trip (wave-uniform exit: a 68-neighbour row leaves a chunk of 4)
      for (int t0 = 0; 4 * t0 < nch; t0 += 4)
#pragma unroll
      for (int t = t0; t < t0 + 4; ++t) {
        const int e = lane + 64 * t;
        const int mm = e >> 4, g = e & 15;
        float v = 0.0f;
        if (mm < nch) {
          const float dd = L.ud[mm].w - shift_l;
#ifdef AIMNET_PROBE_FWD_NO_EXP  // measurement builds only: what the 16 exponentials per pair cost
          v = (1.0f - bp.eta * dd * dd) * L.fc[mm];
#else
          v = exp_neg(-bp.eta * dd * dd) * L.fc[mm];
#endif
        }
        L.gs[mm][g] = v;
      }
      lds_sync<SPLIT>();
      // 4-deep software pipeline over the neighbour rows: the kernel is latency-bound (25 VALU
      // instructions per 1 KiB row), so keep four row loads in flight per wave
      const int qc_f = lane & 3;
      const int qsel = max(qc_f - 1, 0);
      (void)qsel;
      const float fm0 = qc_f == 0 ? 1.f : 0.f, fm1 = qc_f == 1 ? 1.f : 0.f, fm2 = qc_f == 2 ? 1.f : 0.f,
                  fm3 = qc_f == 3 ? 1.f : 0.f;
      auto rowj = [&](int j) {  // j wave-uniform -> scalar address
        return reinterpret_cast<const float4*>(a + (size_t)__builtin_amdgcn_readfirstlane(j) * NF)[lane];
      };
      auto use = [&](int mm, const float4& av) {
        const float4 gv = *reinterpret_cast<const float4*>(&L.gs[mm][g4]);
        const float4 u = L.ud[mm];
        const f2 t0 = mk2(av.x, av.y) * mk2(gv.x, gv.y), t1 = mk2(av.z, av.w) * mk2(gv.z, gv.w);
        acc[0][0] += t0;       acc[0][1] += t1;
#ifdef AIMNET_PROBE_FWD_SPLAT_MOV  // measurement build: the compiler's form (a v_mov pair per component builds the {u, u} splat)
        acc[1][0] += t0 * u.x; acc[1][1] += t1 * u.x;
        acc[2][0] += t0 * u.y; acc[2][1] += t1 * u.y;
        acc[3][0] += t0 * u.z; acc[3][1] += t1 * u.z;
#else
        const f2 uxy = mk2(u.x, u.y), uzw = mk2(u.z, u.w);  // (halves of the 16-byte LDS read: no instruction)
        pk_fma_bcast<false>(acc[1][0], t0, uxy); pk_fma_bcast<false>(acc[1][1], t1, uxy);
        pk_fma_bcast<true>(acc[2][0], t0, uxy);  pk_fma_bcast<true>(acc[2][1], t1, uxy);
        pk_fma_bcast<false>(acc[3][0], t0, uzw); pk_fma_bcast<false>(acc[3][1], t1, uzw);
#endif
        if (HAS_Q) {
#ifdef AIMNET_PROBE_FWD_QSEL_FMA  // measurement build: the component selected by three FMAs with 0 / 1 weights (rounds 1 - 5)
          const float w = L.gs[mm][lane >> 2] * (fm0 + fm1 * u.x + fm2 * u.y + fm3 * u.z);
#else
          // the lane's component (1, ux, uy, uz)[lane & 3] read straight from the staged pair record: one 4-byte LDS read and a
          // select instead of three FMAs with 0 / 1 weights
          float uc = reinterpret_cast<const float*>(&L.ud[mm])[qsel];
          uc = qc_f == 0 ? 1.0f : uc;
          const float w = L.gs[mm][lane >> 2] * uc;
#endif
#pragma unroll
          for (int ch = 0; ch < NQ; ++ch) accq[ch] += L.qj[ch][mm] * w;
        }
      };
      if (P0M) {
        const int gq = lane >> 2;
        auto flush = [&](int z, float m) {  // S_i += afv[z] (x) M[z]
          L.mt[lane] = m;
          lds_sync<false>();
          const float4 av = reinterpret_cast<const float4*>(a + (size_t)z * NF)[lane];
          const float avv[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
          for (int gi = 0; gi < 4; ++gi) {
            const float4 m4 = *reinterpret_cast<const float4*>(&L.mt[(g4 + gi) * 4]);
            acc[0][gi >> 1][gi & 1] += avv[gi] * m4.x;
            acc[1][gi >> 1][gi & 1] += avv[gi] * m4.y;
            acc[2][gi >> 1][gi & 1] += avv[gi] * m4.z;
            acc[3][gi >> 1][gi & 1] += avv[gi] * m4.w;
          }
          lds_sync<false>();
        };
        // one run of equal elements after the other (the staging recorded them): the pair loop has no branch and no scalar
        // round trip per pair - two 4-byte LDS reads and one FMA, which the compiler can keep several of in flight
        const int cidx = max(qc_f - 1, 0);
        int start = 0;
        for (int r = 0; r < p0_runs; ++r) {
          const int z = __builtin_amdgcn_readfirstlane(L.rz[r]), end = __builtin_amdgcn_readfirstlane(L.re[r]);
          float accm = 0.f;
          for (int mm = start; mm < end; ++mm) {
            float uc = reinterpret_cast<const float*>(&L.ud[mm])[cidx];
            uc = qc_f == 0 ? 1.f : uc;
            accm += L.gs[mm][gq] * uc;
          }
          flush(z, accm);
          start = end;
        }
      } else {
#ifdef AIMNET_PROBE_FWD_DEEP  // measurement builds only: eight row loads in flight per wave instead of four
        const int nch8 = (nch + 7) & ~7;
        int4 jj = *reinterpret_cast<const int4*>(&L.j[0]);
        int4 jk = *reinterpret_cast<const int4*>(&L.j[4]);
        float4 r0 = rowj(jj.x), r1 = rowj(jj.y), r2 = rowj(jj.z), r3 = rowj(jj.w);
        float4 r4 = rowj(jk.x), r5 = rowj(jk.y), r6 = rowj(jk.z), r7 = rowj(jk.w);
        for (int mm = 0; mm < nch8; mm += 8) {
          const float4 c0 = r0, c1 = r1, c2 = r2, c3 = r3, c4 = r4, c5 = r5, c6 = r6, c7 = r7;
          if (mm + 8 < nch8) {
            jj = *reinterpret_cast<const int4*>(&L.j[mm + 8]);
            jk = *reinterpret_cast<const int4*>(&L.j[mm + 12]);
            r0 = rowj(jj.x); r1 = rowj(jj.y); r2 = rowj(jj.z); r3 = rowj(jj.w);
            r4 = rowj(jk.x); r5 = rowj(jk.y); r6 = rowj(jk.z); r7 = rowj(jk.w);
          }
          use(mm, c0); use(mm + 1, c1); use(mm + 2, c2); use(mm + 3, c3);
          use(mm + 4, c4); use(mm + 5, c5); use(mm + 6, c6); use(mm + 7, c7);
        }
#else
        const int nch4 = (nch + 3) & ~3;  // rows >= nch have gs = 0 and a valid (clamped) index: harmless
        // (the four ids of a group come with one uniform-address ds_read_b128; mm + 4 <= 60 inside the guard)
        int4 jj = *reinterpret_cast<const int4*>(&L.j[0]);
        float4 r0 = rowj(jj.x), r1 = rowj(jj.y), r2 = rowj(jj.z), r3 = rowj(jj.w);
#ifndef AIMNET_PROBE_FWD_PINGPONG  // one register set, rotated by copies (two 64-bit moves per pair)
        for (int mm = 0; mm < nch4; mm += 4) {
          const float4 c0 = r0, c1 = r1, c2 = r2, c3 = r3;
          if (mm + 4 < nch4) {
            jj = *reinterpret_cast<const int4*>(&L.j[mm + 4]);
            r0 = rowj(jj.x); r1 = rowj(jj.y); r2 = rowj(jj.z); r3 = rowj(jj.w);
          }
          use(mm, c0); use(mm + 1, c1); use(mm + 2, c2); use(mm + 3, c3);
        }
#else
        // measurement build: two register sets in turn (r and s), no copies - 55 instead of 67 vector instructions per four pairs, but
        // the kernel then needs 64 B of scratch at 128 VGPRs: 65.7 against 63.4 us per dispatch (profiles/r6_conv_fwd.md)
        float4 s0 = r0, s1 = r1, s2 = r2, s3 = r3;
        for (int mm = 0; mm < nch4; mm += 8) {
          if (mm + 4 < nch4) {
            jj = *reinterpret_cast<const int4*>(&L.j[mm + 4]);
            s0 = rowj(jj.x); s1 = rowj(jj.y); s2 = rowj(jj.z); s3 = rowj(jj.w);
          }
          use(mm, r0); use(mm + 1, r1); use(mm + 2, r2); use(mm + 3, r3);
          if (mm + 4 >= nch4) break;
          if (mm + 8 < nch4) {
            jj = *reinterpret_cast<const int4*>(&L.j[mm + 8]);
            r0 = rowj(jj.x); r1 = rowj(jj.y); r2 = rowj(jj.z); r3 = rowj(jj.w);
          }
          use(mm + 4, s0); use(mm + 5, s1); use(mm + 6, s2); use(mm + 7, s3);
        }
#endif
#endif
      }
    }
    if (!SPLIT) prefetch_atom(i0 + al.step);  // lands under the epilogue below
    lds_sync<SPLIT>();
    if (SPLIT) {  // waves 1..3 hand their 16 + NQ partial sums per lane to wave 0
      if (wid > 0) {
        float* r = s_red + (wid - 1) * (16 + NQC) * 64 + lane;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          r[(4 * c + 0) * 64] = acc[c][0].x; r[(4 * c + 1) * 64] = acc[c][0].y;
          r[(4 * c + 2) * 64] = acc[c][1].x; r[(4 * c + 3) * 64] = acc[c][1].y;
        }
#pragma unroll
        for (int ch = 0; ch < NQC; ++ch) r[(16 + ch) * 64] = accq[ch];
      }
      lds_sync<SPLIT>();
      if (wid == 0) {
#pragma unroll
        for (int w = 0; w < 3; ++w) {
          const float* r = s_red + w * (16 + NQC) * 64 + lane;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            acc[c][0] += mk2(r[(4 * c + 0) * 64], r[(4 * c + 1) * 64]);
            acc[c][1] += mk2(r[(4 * c + 2) * 64], r[(4 * c + 3) * 64]);
          }
#pragma unroll
          for (int ch = 0; ch < NQC; ++ch) accq[ch] += r[(16 + ch) * 64];
        }
      }
    }
    const bool live = live_atom && (!SPLIT || wid == 0);
    // ---- epilogue: agh contraction + square-sum, assemble the MLP input row -----------------
    float* sv = &L.gs[0][0];  // reuse: sv[(a*16+g)*3 + k], 768 floats; svq at 768.. (48 floats per charge channel)
    if (live) {
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) {
        const int f = lane * 4 + gi;
        sv[f * 3 + 0] = acc[1][gi >> 1][gi & 1];
        sv[f * 3 + 1] = acc[2][gi >> 1][gi & 1];
        sv[f * 3 + 2] = acc[3][gi >> 1][gi & 1];
      }
      if (HAS_Q && (lane & 3) != 0) {
#pragma unroll
        for (int ch = 0; ch < NQ; ++ch) sv[768 + ch * 48 + (lane >> 2) * 3 + (lane & 3) - 1] = accq[ch];
      }
      const int ri = row_of ? min(63, max(0, row_of[i])) : i;
      const float4 av = reinterpret_cast<const float4*>(a + (size_t)ri * NF)[lane];
      if (X3) {
        unsigned short* x3 = reinterpret_cast<unsigned short*>(x) + (size_t)i * (X3 == 2 ? 2 : 3) * ldx;
        store_split_x4<X3>(x3, 4 * lane, f32x4{av.x, av.y, av.z, av.w});
        store_split_x4<X3>(x3, NF + 4 * lane, f32x4{acc[0][0].x, acc[0][0].y, acc[0][1].x, acc[0][1].y});
      } else {
        float* xr = x + (size_t)i * ldx;
        reinterpret_cast<float4*>(xr)[lane] = av;
        reinterpret_cast<float4*>(xr + NF)[lane] = make_float4(acc[0][0].x, acc[0][0].y, acc[0][1].x, acc[0][1].y);
      }
    }
    lds_sync<SPLIT>();
    if (live) {
      float* xr = x + (size_t)i * ldx;
      unsigned short* x3 = reinterpret_cast<unsigned short*>(x) + (size_t)i * (X3 == 2 ? 2 : 3) * ldx;
      // X3: the columns from 2 NF on are staged in the wave's LDS scratch (behind the 864 floats of sv; the chunk arrays that follow
      // gs are idle in the epilogue) and leave as 8-byte plane pieces, four consecutive columns per lane
      float* xt = sv + 864;
      auto put = [&](int col, float v) __attribute__((always_inline)) {  // one element of the row
        if (X3) xt[col - 2 * NF] = v;
        else xr[col] = v;
      };
      // one-wave-per-atom form: rolled, the contraction keeps the kernel at 128 VGPRs with 6 spilled dwords instead of 22
#pragma unroll UNR_T
      for (int t = 0; t < 3; ++t) {
        const int o = lane + 64 * t;  // (a, h) = (o / 12, o % 12)
        const int aa = o / H_;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
#ifdef AIMNET_PROBE_FWD_NO_EPI  // measurement builds only (tests/tools/variant.sh): what the agh contraction costs
#pragma unroll 1
        for (int g = 0; g < 1; ++g) {
#else
#pragma unroll UNR_G
        for (int g = 0; g < G_; ++g) {
#endif
          const float w = s_agh[g * (A_ * H_) + o];  // o = aa * H + hh
          const float* s3 = &sv[(aa * G_ + g) * 3];
          v0 += w * s3[0];
          v1 += w * s3[1];
          v2 += w * s3[2];
        }
        float* vs = Vsave + (size_t)i * (NV * 3) + o;  // three planes [k][NV]: contiguous wave stores (and loads in unconcat)
        vs[0] = v0; vs[NV] = v1; vs[2 * NV] = v2;
        put(2 * NF + o, v0 * v0 + v1 * v1 + v2 * v2);
      }
      if (HAS_Q) {
        const int c0 = 2 * NF + NV;  // 704
        if (lane < NQ) put(c0 + lane, q[(size_t)lane * n_atoms + i]);
#pragma unroll
        for (int ch = 0; ch < NQ; ++ch) {
          if ((lane & 3) == 0) put(c0 + NQ + ch * G_ + (lane >> 2), accq[ch]);
          if (lane < H_) {
            float v0 = 0.f, v1 = 0.f, v2 = 0.f;
#pragma unroll
            for (int g = 0; g < G_; ++g) {
              const float w = s_aghq[(ch * G_ + g) * H_ + lane];
              v0 += w * sv[768 + ch * 48 + g * 3 + 0];
              v1 += w * sv[768 + ch * 48 + g * 3 + 1];
              v2 += w * sv[768 + ch * 48 + g * 3 + 2];
            }
            float* vs = Vqsave + ((size_t)i * NQ + ch) * (H_ * 3) + lane * 3;
            vs[0] = v0; vs[1] = v1; vs[2] = v2;
            put(c0 + NQ + NQ * G_ + ch * H_ + lane, v0 * v0 + v1 * v1 + v2 * v2);
          }
        }
        const int used = c0 + NQ * (1 + G_ + H_);  // 733 (762 with two channels)
        if (lane < ldx - used) put(used + lane, 0.0f);  // zero the K padding of the GEMM operand
      } else {
        const int used = 2 * NF + NV;
        if (lane < ldx - used) put(used + lane, 0.0f);
      }
      if (X3) {
        lds_sync<false>();
        const int c = 4 * lane;
        if (c < ldx - 2 * NF) store_split_x4<X3>(x3, 2 * NF + c, f32x4{xt[c], xt[c + 1], xt[c + 2], xt[c + 3]});
      }
    }
  }
}


// Split threshold (atoms up to which the four waves of a block share one centre atom): a per-engine setting
// (aimnet_engine::split_max, AIMNET_SPLIT_MAX / set_option("split_max"); tests: 0 runs every fixture, however small, through the
// one-wave-per-atom kernels that large systems take); the launchers receive it as an argument.
int conv_split_max_default() { return SPLIT_MAX_ATOMS; }

int launch_conv_fwd(hipStream_t s, int nq, const float* a, const int* row_of, const float* q, const int* nb_idx,
                    const int* nb_cnt, const float4* pg, int cap, const float* agh_a, const float* agh_q, BasisParams bp,
                    float* x, int ldx, float* Vsave, float* Vqsave, int n_atoms, const int* order, bool species_moments,
                    int split_max, int x_split) {
  const bool split = n_atoms <= split_max;

  // (one-wave-per-atom form: exactly the 4 blocks per CU that are resident - with twice as many the second half only queues
  // behind the first and pays the block prologue again: 0.217 -> 0.207 ms/step over the three launches)
  const int grid = split ? n_atoms : min(ceil_div(n_atoms, APB), device_cus() * AIMNET_PROBE_FWD_OCC);
  if (species_moments && row_of && nq == 0 && !split) {  // pass 0 of a large system: per-element moments instead of row gathers
    if (x_split == 2)
      hipLaunchKernelGGL((conv_fwd_kernel<0, false, true, 2>), dim3(grid), dim3(256), 0, s, a, row_of, q, nb_idx, nb_cnt, pg, cap,
                         agh_a, agh_q, bp, x, ldx, Vsave, Vqsave, n_atoms, order);
    else if (x_split)
      hipLaunchKernelGGL((conv_fwd_kernel<0, false, true, 1>), dim3(grid), dim3(256), 0, s, a, row_of, q, nb_idx, nb_cnt, pg, cap,
                         agh_a, agh_q, bp, x, ldx, Vsave, Vqsave, n_atoms, order);
    else
      hipLaunchKernelGGL((conv_fwd_kernel<0, false, true>), dim3(grid), dim3(256), 0, s, a, row_of, q, nb_idx, nb_cnt, pg, cap, agh_a,
                         agh_q, bp, x, ldx, Vsave, Vqsave, n_atoms, order);
    AIMNET_LAUNCH_CHECK();
    return 0;
  }
  if (x_split) {
#define AIMNET_FWD3(HQ, SP, F)                                                                                                 \
  hipLaunchKernelGGL((conv_fwd_kernel<HQ, SP, false, F>), dim3(grid), dim3(256), 0, s, a, row_of, q, nb_idx, nb_cnt, pg, cap, \
                     agh_a, agh_q, bp, x, ldx, Vsave, Vqsave, n_atoms, order)
#define AIMNET_FWD3F(HQ, SP) \
  do { if (x_split == 2) AIMNET_FWD3(HQ, SP, 2); else AIMNET_FWD3(HQ, SP, 1); } while (0)
    if (nq == 2) { if (split) AIMNET_FWD3F(2, true); else AIMNET_FWD3F(2, false); }
    else if (nq == 1) { if (split) AIMNET_FWD3F(1, true); else AIMNET_FWD3F(1, false); }
    else { if (split) AIMNET_FWD3F(0, true); else AIMNET_FWD3F(0, false); }
#undef AIMNET_FWD3F
#undef AIMNET_FWD3
    AIMNET_LAUNCH_CHECK();
    return 0;
  }
#define AIMNET_FWD(HQ, SP)                                                                                                  \
  hipLaunchKernelGGL((conv_fwd_kernel<HQ, SP>), dim3(grid), dim3(256), 0, s, a, row_of, q, nb_idx, nb_cnt, pg, cap, agh_a, agh_q, \
                     bp, x, ldx, Vsave, Vqsave, n_atoms, order)
  if (nq == 2) {
    if (split) AIMNET_FWD(2, true); else AIMNET_FWD(2, false);
  } else if (nq == 1) {
    if (split) AIMNET_FWD(1, true); else AIMNET_FWD(1, false);
  } else {
    if (split) AIMNET_FWD(0, true); else AIMNET_FWD(0, false);
  }
#undef AIMNET_FWD
  AIMNET_LAUNCH_CHECK();
  return 0;
}
template <int NQ>  // charge channels, as in conv_fwd_kernel; Vqsave [N][NQ][H*3], Sqbar [N][NQ][G*4]
__global__ __launch_bounds__(256) void unconcat_kernel(const float* __restrict__ xbar, int ldx,
                                                      const float* __restrict__ Vsave, const float* __restrict__ Vqsave,
                                                      const float* __restrict__ agh_a, const float* __restrict__ agh_q,
                                                      float* __restrict__ Sbar, float* __restrict__ Sqbar, int n_atoms) {
  constexpr bool HAS_Q = NQ > 0;
  constexpr int NQC = NQ > 0 ? NQ : 1;
  __shared__ float s_aghq[NQC * G_ * H_];
  __shared__ float s_vb[APB][NV * 3 + NQC * H_ * 3];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (HAS_Q)
    for (int k = threadIdx.x; k < NQ * G_ * H_; k += 256) s_aghq[k] = agh_q[k];
  float* vb = s_vb[wid];
  const AtomLoop al = atom_loop(n_atoms, APB);
  // the lane's 4 x 12 agh weights (a = lane >> 2, g = 4 (lane & 3) + gi) are the same for every atom: registers.  (Read from the
  // LDS table, address stride 48 floats across the lanes, they were a 32-way bank conflict: 48 such reads per atom.)
  float wreg[4][H_];
#pragma unroll
  for (int gi = 0; gi < 4; ++gi)
#pragma unroll
    for (int h = 0; h < H_; ++h) wreg[gi][h] = agh_a[((lane >> 2) * G_ + (lane & 3) * 4 + gi) * H_ + h];
  __syncthreads();  // the agh_q table (the only LDS the waves share; everything below is per wave)
  // A wave's atoms are independent and its LDS scratch is its own: wave-level hand-offs, and the NEXT atom's operands (2.9 KiB per
  // atom: the vector part of xbar, the saved V) are requested before the contraction of the current one - the kernel was
  // latency-bound at 2.5 TB/s of HBM traffic with every wave of a block waiting at two block barriers per atom.
  float pf_f[3], pf_v[3][3], pf_fq[NQC], pf_vq[NQC][3], pf_s[NQC];
  auto load_atom = [&](int i) __attribute__((always_inline)) {
    const float* xr = xbar + (size_t)i * ldx;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int o = lane + 64 * t;
      pf_f[t] = xr[2 * NF + o];
      const float* vs = Vsave + (size_t)i * (NV * 3) + o;  // planes [k][NV], see conv_fwd_kernel
      pf_v[t][0] = vs[0]; pf_v[t][1] = vs[NV]; pf_v[t][2] = vs[2 * NV];
    }
    if (HAS_Q) {
#pragma unroll
      for (int ch = 0; ch < NQ; ++ch) {
        pf_s[ch] = xr[2 * NF + NV + NQ + ch * G_ + (lane >> 2)];
        if (lane < H_) {
          pf_fq[ch] = xr[2 * NF + NV + NQ + NQ * G_ + ch * H_ + lane];
          const float* vs = Vqsave + ((size_t)i * NQ + ch) * (H_ * 3) + lane * 3;
          pf_vq[ch][0] = vs[0]; pf_vq[ch][1] = vs[1]; pf_vq[ch][2] = vs[2];
        }
      }
    }
  };
  int i = al.first + wid;
  bool live = i < al.last;
  if (live) load_atom(i);
  for (int i0 = al.first; i0 < al.last; i0 += al.step) {
    float qs[NQC];
    if (live) {
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int o = lane + 64 * t;
        const float f = 2.0f * pf_f[t];
        vb[o * 3 + 0] = f * pf_v[t][0];
        vb[o * 3 + 1] = f * pf_v[t][1];
        vb[o * 3 + 2] = f * pf_v[t][2];
      }
      if (HAS_Q) {
#pragma unroll
        for (int ch = 0; ch < NQ; ++ch) {
          qs[ch] = pf_s[ch];
          if (lane < H_) {
            const float f = 2.0f * pf_fq[ch];
            vb[NV * 3 + ch * (H_ * 3) + lane * 3 + 0] = f * pf_vq[ch][0];
            vb[NV * 3 + ch * (H_ * 3) + lane * 3 + 1] = f * pf_vq[ch][1];
            vb[NV * 3 + ch * (H_ * 3) + lane * 3 + 2] = f * pf_vq[ch][2];
          }
        }
      }
    }
    lds_sync<false>();
    const int i_next = i + al.step;
    const bool live_next = i_next < al.last;
    if (live_next) load_atom(i_next);
    if (live) {
      const int aa = lane >> 2;
      // Sbar row layout: the lane's 16 floats are stored COMPONENT-major, [c][gi] (c = 0 scalar, 1..3
      // vector; gi = the lane's 4 shifts), so that conv_bwd's float4 loads are (gi 0..3) of one
      // component and all its arithmetic packs into v_pk_* pairs over gi without register shuffles
      // ... and the four components are four 1 KiB planes of the row (float4 index c * 64 + lane), so that every one of
      // these stores - and of conv_bwd's loads of the row - is one contiguous wave access
      float4* out = reinterpret_cast<float4*>(Sbar + (size_t)i * (NF * 4)) + lane;
      float vv[3][4];
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) {
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
#pragma unroll
        for (int h = 0; h < H_; ++h) {
          const float w = wreg[gi][h];
          const float* v3 = &vb[(aa * H_ + h) * 3];
          v0 += w * v3[0];
          v1 += w * v3[1];
          v2 += w * v3[2];
        }
#ifdef AIMNET_PROBE_SBAR16  // measurement builds only: what storing the vector planes of Sbar in fp16 would do to the forces
        v0 = (float)(_Float16)v0; v1 = (float)(_Float16)v1; v2 = (float)(_Float16)v2;
#endif
        vv[0][gi] = v0; vv[1][gi] = v1; vv[2][gi] = v2;
      }
      // (plane 0, the scalar part of xbar, is not stored: conv_bwd_kernel reads it from the xbar row itself)
      out[64] = make_float4(vv[0][0], vv[0][1], vv[0][2], vv[0][3]);
      out[128] = make_float4(vv[1][0], vv[1][1], vv[1][2], vv[1][3]);
      out[192] = make_float4(vv[2][0], vv[2][1], vv[2][2], vv[2][3]);
      if (HAS_Q) {
        const int g = lane >> 2, c = lane & 3;
#pragma unroll
        for (int ch = 0; ch < NQ; ++ch) {
          float v;
          if (c == 0) {
            v = qs[ch];
          } else {
            v = 0.f;
#pragma unroll
            for (int h = 0; h < H_; ++h) v += s_aghq[(ch * G_ + g) * H_ + h] * vb[NV * 3 + ch * (H_ * 3) + h * 3 + c - 1];
          }
          Sqbar[((size_t)i * NQ + ch) * (G_ * 4) + lane] = v;
        }
      }
    }
    lds_sync<false>();  // the contraction has read vb before the next atom's values overwrite it
    i = i_next;
    live = live_next;
  }
}

int launch_unconcat(hipStream_t s, int nq, const float* xbar, int ldx, const float* Vsave, const float* Vqsave,
                    const float* agh_a, const float* agh_q, float* Sbar, float* Sqbar, int n_atoms) {
  const int grid = min(ceil_div(n_atoms, APB), 256 * 8);
  if (nq == 2)
    hipLaunchKernelGGL(unconcat_kernel<2>, dim3(grid), dim3(256), 0, s, xbar, ldx, Vsave, Vqsave, agh_a, agh_q, Sbar,
                       Sqbar, n_atoms);
  else if (nq == 1)
    hipLaunchKernelGGL(unconcat_kernel<1>, dim3(grid), dim3(256), 0, s, xbar, ldx, Vsave, Vqsave, agh_a, agh_q, Sbar,
                       Sqbar, n_atoms);
  else
    hipLaunchKernelGGL(unconcat_kernel<0>, dim3(grid), dim3(256), 0, s, xbar, ldx, Vsave, Vqsave, agh_a, agh_q, Sbar,
                       Sqbar, n_atoms);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
constexpr int CHB = 48;  // chunk of the backward kernels: 8.1 KiB of LDS per wave -> four blocks per CU
static_assert(CHB % 16 == 0, "the basis table of a chunk is filled 16 pairs per trip: other sizes write past the LDS arrays");
struct BwdWaveLds {
  float gs[CHB][G_];
  float dgs[CHB][G_];
  float4 ud[CHB];
  int j[CHB];      // neighbour atom
  int jr[CHB];     // its feature row (atomic number in pass 0)
  float qj[2][CHB];  // neighbour charges per charge channel
  float fc[CHB], dfc[CHB];
  float4 red[CHB];  // XE form: the pair's wave-reduced (D, U0, U1, U2)
};

// per-lane linear map (dbar, ubar) -> rbar = dbar*u + (ubar - (ubar.u) u)/d
__device__ __forceinline__ void rbar_of(float db, float ub0, float ub1, float ub2, const float4& u, float inv_d,
                                        float& r0, float& r1, float& r2) {
  const float dot = ub0 * u.x + ub1 * u.y + ub2 * u.z;
  r0 = db * u.x + (ub0 - dot * u.x) * inv_d;
  r1 = db * u.y + (ub1 - dot * u.y) * inv_d;
  r2 = db * u.z + (ub2 - dot * u.z) * inv_d;
}

// NQ: charge channels (0, 1, 2) as in conv_fwd_kernel: q / qbar planes [NQ][n_atoms], Sqbar [N][NQ][G*4]
// XE ("X eliminated", one wave per atom only): the half of the pair adjoints that contracts the NEIGHBOUR's features with the
// centre's Sbar (sum_a a_j Sbar_i) is the other half (sum_a a_i Sbar_j) of the REVERSE ordered pair with u -> -u.  So every
// ordered pair p = (i -> j) only evaluates D_p = sum dgs (1,-u).Y_p, U_p = sum gs Y_p[1:4] with Y_p = sum_a a_i Sbar_j, writes
//   F1(p) = (U_p - (U_p.u) u) / d - D_p u
// into `pairbuf[i * cap + m]` (summed over the passes when pb_accum), and dE/dx_i = sum_m F1(i -> j_m) - F1(j_m -> i) is
// formed afterwards by pair_force_kernel through the reverse-pair map; the virial is sum_p -r_p (x) F1(p).  No a_j gather
// (4 KiB per pair instead of 5.25 KiB), 40 % fewer packed FMAs, and the force / virial tail runs once per chunk with
// lane = pair instead of in every lane for every pair.
template <int NQ, bool NEED_ABAR, bool STRESS, bool SPLIT, bool XE>  // SPLIT: as in conv_fwd_kernel (4 waves per centre atom)
__global__ __launch_bounds__(256, 4) void conv_bwd_kernel(const float* __restrict__ a, const int* __restrict__ row_of,
                                                      const float* __restrict__ q,
                                                      const float* __restrict__ Sbar, const float* __restrict__ Sqbar,
                                                      const int* __restrict__ nb_idx, const int* __restrict__ nb_cnt,
                                                      const float4* __restrict__ pg, int cap, BasisParams bp,
                                                      const float* __restrict__ xbar, int ldx,
                                                      const float* __restrict__ abar_in, float* __restrict__ abar_out,
                                                      const float* __restrict__ qbar_in, float* __restrict__ qbar_out,
                                                      float* __restrict__ fgrad, float* __restrict__ virial_atom,
                                                      int n_atoms, const int* __restrict__ order,
                                                      float4* __restrict__ pairbuf, int pb_accum) {
  static_assert(!(XE && SPLIT), "the reverse-pair form is a one-wave-per-atom kernel");
  constexpr bool HAS_Q = NQ > 0;
  constexpr int NQC = NQ > 0 ? NQ : 1;
  __shared__ __attribute__((aligned(16))) BwdWaveLds wl[APB];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  BwdWaveLds& L = wl[wid];
  const int g4 = (lane & 3) * 4;
  const float shift_l = bp.shifts[lane & 15];  // the table loop below gives lane l the shift l & 15 in every iteration
  const int qg = lane >> 2, qc = lane & 3;
  const float qm0 = qc == 0 ? 1.f : 0.f, qm1 = qc == 1 ? 1.f : 0.f, qm2 = qc == 2 ? 1.f : 0.f, qm3 = qc == 3 ? 1.f : 0.f;
  const float qsgn = qc == 0 ? 1.f : -1.f;
  const int qc_sel = max(qc - 1, 0);  // word of the staged (ux, uy, uz, d) record that holds the lane's component (qc == 0: 1)
  (void)qc_sel;

  __shared__ float s_red[SPLIT ? 3 * (4 * 64 + 14) : 1];  // SPLIT: per-lane abar partials + 14 reduced scalars of waves 1..3

  const AtomLoop al = atom_loop(n_atoms, SPLIT ? 1 : APB);
  for (int i0 = al.first; i0 < al.last; i0 += al.step) {
    const int aslot = SPLIT ? 0 : wid;
    const bool live_atom = i0 + aslot < al.last;
    const bool live = live_atom;  // every wave loads the centre's rows; only the writer differs (see the epilogue)
    const int i = live_atom ? (order ? order[i0 + aslot] : i0 + aslot) : 0;
    const int cnt_all = live_atom ? nb_cnt[i] : 0;
    int m_lo = 0, cnt = cnt_all;
    int cmax = cnt;
    if (SPLIT) {
      const int q4 = (cnt_all + 3) >> 2;
      m_lo = wid * q4;
      cnt = max(0, min(q4, cnt_all - m_lo));
      cmax = q4;
    }
    // centre atom's own rows
    // [half]: gi pair (0,1) / (2,3);  Si[c][half]: component c of the centre's Sbar row
    f2 ai[2] = {mk2(0.f, 0.f), mk2(0.f, 0.f)};
    f2 Si[4][2];
#pragma unroll
    for (int c = 0; c < 4; ++c) Si[c][0] = Si[c][1] = mk2(0.f, 0.f);
    float qi[NQC], Sqi[NQC];
#pragma unroll
    for (int ch = 0; ch < NQC; ++ch) qi[ch] = Sqi[ch] = 0.0f;
    if (live) {
      const int ri = row_of ? min(63, max(0, row_of[i])) : i;
      const float4 t = reinterpret_cast<const float4*>(a + (size_t)ri * NF)[lane];
      ai[0] = mk2(t.x, t.y);
      ai[1] = mk2(t.z, t.w);
      if (!XE) {
        const float4* sp = reinterpret_cast<const float4*>(Sbar + (size_t)i * (NF * 4)) + lane;  // planes [c][lane], see unconcat_kernel
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float4 v = c == 0 ? reinterpret_cast<const float4*>(xbar + (size_t)i * ldx + NF)[lane] : sp[c * 64];
          Si[c][0] = mk2(v.x, v.y);
          Si[c][1] = mk2(v.z, v.w);
        }
      }
#pragma unroll
      for (int ch = 0; ch < NQ; ++ch) {
        qi[ch] = q[(size_t)ch * n_atoms + i];
        if (!XE) Sqi[ch] = Sqbar[((size_t)i * NQ + ch) * (G_ * 4) + lane];
      }
    }
    f2 ab[2] = {mk2(0.f, 0.f), mk2(0.f, 0.f)};
    float xa0 = 0.f, xa1 = 0.f, xa2 = 0.f;  // per-lane partial of dE/dx_i
    float qacc[NQC];
#pragma unroll
    for (int ch = 0; ch < NQC; ++ch) qacc[ch] = 0.0f;
    float W[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) W[k] = 0.0f;

    for (int c0 = 0; c0 < cmax; c0 += CHB) {
      const int nch = max(0, min(CHB, cnt - c0));
      lds_sync<SPLIT>();
      if (lane < nch) {
        const size_t p = (size_t)i * cap + m_lo + c0 + lane;
        const int j = nb_idx[p];
        L.j[lane] = j;
        L.jr[lane] = row_of ? min(63, max(0, row_of[j])) : j;
        const float4 ud = pg[p];
        L.ud[lane] = ud;
        float dfc;
        L.fc[lane] = basis_fc(bp, ud.w, dfc);  // one sincos per pair, not per (pair, shift)
        L.dfc[lane] = dfc;
#pragma unroll
        for (int ch = 0; ch < NQ; ++ch) L.qj[ch][lane] = q[(size_t)ch * n_atoms + j];
      }
      lds_sync<SPLIT>();
      // four pairs per step, four steps per trip (wave-uniform exit; slots >= nch are never read)
      for (int t0 = 0; 4 * t0 < nch; t0 += 4)
#pragma unroll
      for (int t = t0; t < t0 + 4; ++t) {
        const int e = lane + 64 * t;
        const int mm = e >> 4, g = e & 15;
        float v = 0.0f, dv = 0.0f;
        if (mm < nch) {
          const float fc = L.fc[mm], dfc = L.dfc[mm];
          const float dd = L.ud[mm].w - shift_l;
          const float Gg = exp_neg(-bp.eta * dd * dd);
          v = Gg * fc;
          dv = Gg * (dfc - 2.0f * bp.eta * dd * fc);
        }
        L.gs[mm][g] = v;
        L.dgs[mm][g] = dv;
      }
      lds_sync<SPLIT>();
      // Software pipeline over the neighbour rows in two named buffers (no register copies): the 80 B per lane of the next
      // neighbour are requested before the current one is consumed, so the L2 / Infinity-Cache latency of the gather hides
      // under ~100 VALU instructions.  Neighbour ids are wave-uniform: readfirstlane moves them to SGPRs, so the row addresses
      // are scalar (saddr-form global loads, no per-lane 64-bit address arithmetic on the VALU).
      struct Row {
        float4 aj, s0, s1, s2, s3;
        float sq[NQC];
      };
      auto load_row = [&](int mn, Row& R) {
        const int jn = __builtin_amdgcn_readfirstlane(nch > 0 ? L.j[mn] : 0);
        if (!XE) {
          const int jrn = __builtin_amdgcn_readfirstlane(nch > 0 ? L.jr[mn] : 0);
          R.aj = reinterpret_cast<const float4*>(a + (size_t)jrn * NF)[lane];
        }
        const float4* spn = reinterpret_cast<const float4*>(Sbar + (size_t)jn * (NF * 4)) + lane;
        // the scalar plane (c = 0) of Sbar_j is the s-block of j's xbar row: read it there, unconcat does not copy it
        R.s0 = reinterpret_cast<const float4*>(xbar + (size_t)jn * ldx + NF)[lane];
        R.s1 = spn[64]; R.s2 = spn[128]; R.s3 = spn[192];
#pragma unroll
        for (int ch = 0; ch < NQ; ++ch) R.sq[ch] = Sqbar[((size_t)jn * NQ + ch) * (G_ * 4) + lane];
      };
      // one neighbour: accumulates abar / qacc (and, in the combined form, dE/dx_i and the virial); returns the lane's partial
      // (D, U0, U1, U2) of the pair
      auto pair_math = [&](int mm, const Row& R, float& D, float& U0, float& U1, float& U2) {
        const float4 gv = *reinterpret_cast<const float4*>(&L.gs[mm][g4]);
        const float4 dgv = *reinterpret_cast<const float4*>(&L.dgs[mm][g4]);
        const float4 u = L.ud[mm];
        // Combined form: both directions of the pair enter dE/dx_i only through the COMBINED adjoints
        //   D = dbar_ij + dbar_ji,  U = ubar_ji - ubar_ij      (u_ji = -u_ij):
        //   dE/dx_i += rbar_ji - rbar_ij = -D u + (U - (U.u) u) / d
        // and the virial of the two ordered pairs is -1/2 r_ij (x) (rbar_ji - rbar_ij).
        // All of it in 2-wide vectors over the lane's shift pairs (v_pk_* instructions).
        const f2 aj[2] = {mk2(R.aj.x, R.aj.y), mk2(R.aj.z, R.aj.w)};
        const f2 gsv[2] = {mk2(gv.x, gv.y), mk2(gv.z, gv.w)};
        const f2 dg[2] = {mk2(dgv.x, dgv.y), mk2(dgv.z, dgv.w)};
        const f2 Sj[4][2] = {{mk2(R.s0.x, R.s0.y), mk2(R.s0.z, R.s0.w)}, {mk2(R.s1.x, R.s1.y), mk2(R.s1.z, R.s1.w)},
                             {mk2(R.s2.x, R.s2.y), mk2(R.s2.z, R.s2.w)}, {mk2(R.s3.x, R.s3.y), mk2(R.s3.z, R.s3.w)}};
        f2 Dv = mk2(0.f, 0.f), U0v = mk2(0.f, 0.f), U1v = mk2(0.f, 0.f), U2v = mk2(0.f, 0.f);
        const f2 uxy = mk2(u.x, u.y), uzw = mk2(u.z, u.w);  // (halves of the 16-byte LDS read: no instruction)
        (void)uxy; (void)uzw;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
#ifdef AIMNET_PROBE_BWD_SPLAT_MOV  // measurement build: the compiler's form (v_mov pairs build the {u, u} splats)
          const f2 Pp = Sj[0][hf] - (u.x * Sj[1][hf] + u.y * Sj[2][hf] + u.z * Sj[3][hf]);
#else
          // the component of u taken from its register pair by the packed instructions' op_sel modifiers (conv_common.h,
          // pk_fma_bcast): no splat moves.  (Not bit-identical to the compiler's form - it contracts the sums differently - but the
          // same arithmetic to fp32 rounding: checksums differ in the last bits, every parity test unchanged.)
          f2 Tj = pk_mul_bcast<false>(Sj[1][hf], uxy);
          pk_fma_bcast<true>(Tj, Sj[2][hf], uxy);
          pk_fma_bcast<false>(Tj, Sj[3][hf], uzw);
          const f2 Pp = Sj[0][hf] - Tj;
#endif
          if (NEED_ABAR) ab[hf] += gsv[hf] * Pp;
          const f2 tp = gsv[hf] * ai[hf];
          if (XE) {
            Dv += (dg[hf] * ai[hf]) * Pp;
            U0v += tp * Sj[1][hf];
            U1v += tp * Sj[2][hf];
            U2v += tp * Sj[3][hf];
          } else {
#ifdef AIMNET_PROBE_BWD_SPLAT_MOV
            const f2 P = Si[0][hf] + (u.x * Si[1][hf] + u.y * Si[2][hf] + u.z * Si[3][hf]);
#else
            f2 Ti = pk_mul_bcast<false>(Si[1][hf], uxy);
            pk_fma_bcast<true>(Ti, Si[2][hf], uxy);
            pk_fma_bcast<false>(Ti, Si[3][hf], uzw);
            const f2 P = Si[0][hf] + Ti;
#endif
            Dv += dg[hf] * (aj[hf] * P + ai[hf] * Pp);
            const f2 t = gsv[hf] * aj[hf];
            U0v += tp * Sj[1][hf] - t * Si[1][hf];
            U1v += tp * Sj[2][hf] - t * Si[2][hf];
            U2v += tp * Sj[3][hf] - t * Si[3][hf];
          }
        }
        D = Dv.x + Dv.y; U0 = U0v.x + U0v.y; U1 = U1v.x + U1v.y; U2 = U2v.x + U2v.y;
        if (HAS_Q) {
          // lane (g, c) of the charge convolution Sq[g,c] = sum_m q_j gs_g (1,u)_c, branch-free:
          // qm0..qm3 are the lane's one-hot component selectors, qsgn = (1,-1,-1,-1)[c] for the reverse pair
          const float gq = L.gs[mm][qg], dgq = L.dgs[mm][qg];
#ifdef AIMNET_PROBE_BWD_SPLAT_MOV
          const float uc = qm0 + qm1 * u.x + qm2 * u.y + qm3 * u.z;
#else
          // (1, ux, uy, uz)[c] read from the staged pair record and a select, instead of three FMAs with 0 / 1 weights
          float uc = reinterpret_cast<const float*>(&L.ud[mm])[qc_sel];
          uc = qc == 0 ? 1.0f : uc;
#endif
#pragma unroll
          for (int ch = 0; ch < NQ; ++ch) {
            const float sqj = R.sq[ch];
            const float sj_s = sqj * qsgn;
            qacc[ch] += gq * sj_s * uc;
            float v;
            if (XE) {
              D += dgq * uc * (qi[ch] * sj_s);
              v = gq * (qi[ch] * sqj);
            } else {
              const float qj = L.qj[ch][mm];
              D += dgq * uc * (qj * Sqi[ch] + qi[ch] * sj_s);
              v = gq * (qi[ch] * sqj - qj * Sqi[ch]);
            }
            U0 += qm1 * v;
            U1 += qm2 * v;
            U2 += qm3 * v;
          }
        }
        if (!XE) {
          const float inv_d = __builtin_amdgcn_rcpf(u.w);
          const float dot = U0 * u.x + U1 * u.y + U2 * u.z;
          const float f0 = (U0 - dot * u.x) * inv_d - D * u.x;
          const float f1 = (U1 - dot * u.y) * inv_d - D * u.y;
          const float f2 = (U2 - dot * u.z) * inv_d - D * u.z;
          xa0 += f0; xa1 += f1; xa2 += f2;
          if (STRESS) {
            const float hx = -0.5f * u.x * u.w, hy = -0.5f * u.y * u.w, hz = -0.5f * u.z * u.w;
            W[0] += hx * f0; W[1] += hx * f1; W[2] += hx * f2;
            W[3] += hy * f0; W[4] += hy * f1; W[5] += hy * f2;
            W[6] += hz * f0; W[7] += hz * f1; W[8] += hz * f2;
          }
        }
      };
      // XE: (D, U0, U1, U2) of 64 lanes -> per 16-lane row: two transposing quad steps leave value (lane & 3) in every lane,
      // then row_shr:4 / row_shr:8 put the row's four sums into its lanes 12..15
      auto quad_row_reduce = [&](float D, float U0, float U1, float U2) {
        const bool odd = (lane & 1) != 0, hi2 = (lane & 2) != 0;
        float k0 = odd ? U0 : D;
        const float s0 = odd ? D : U0;
        float k1 = odd ? U2 : U1;
        const float s1 = odd ? U1 : U2;
        k0 += dpp0<0xB1>(s0);
        k1 += dpp0<0xB1>(s1);
        float z = hi2 ? k1 : k0;
        const float sd = hi2 ? k0 : k1;
        z += dpp0<0x4E>(sd);
        z += dpp0<0x114>(z);
        z += dpp0<0x118>(z);
        return z;
      };
      Row RA, RB;
      load_row(0, RA);
      if (!XE) {  // combined form: one buffer ahead (the second named buffer does not fit next to Sbar_i and a_j in 128 VGPRs)
        for (int mm = 0; mm < nch; ++mm) {
          RB = RA;
          load_row(min(mm + 1, nch - 1), RA);
          float D, U0, U1, U2;
          pair_math(mm, RB, D, U0, U1, U2);
        }
      }
      for (int mm = 0; XE && mm < nch; mm += 2) {
        const bool two = mm + 1 < nch;  // wave-uniform
        float D, U0, U1, U2;
        load_row(min(mm + 1, nch - 1), RB);
        pair_math(mm, RA, D, U0, U1, U2);
        float zA = 0.f, zB = 0.f;
        if (XE) zA = quad_row_reduce(D, U0, U1, U2);
        load_row(min(mm + 2, nch - 1), RA);
        if (two) {
          pair_math(mm + 1, RB, D, U0, U1, U2);
          if (XE) zB = quad_row_reduce(D, U0, U1, U2);
        }
        if (XE) {
          // the rows of both pairs at once: v_permlane16_swap exchanges the odd rows of its first operand with the even rows
          // of the second, v_permlane32_swap the upper half of the first with the lower half of the second (gfx950; written as
          // asm - the builtin with two copies of one value is miscompiled, tests/tools/permlane_probe.hip).  After the two
          // steps rows 0 / 2 hold the first pair's totals and rows 1 / 3 the second's, in their lanes 12..15.
          asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(zA), "+v"(zB));
          float sA = zA + zB, sB = sA;
          asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(sA), "+v"(sB));
          const float tot = sA + sB;
          const int row = lane >> 4;
          if ((lane & 12) == 12 && row < 2 && (row == 0 || two)) reinterpret_cast<float*>(&L.red[mm + row])[lane & 3] = tot;
        }
      }
      if (XE) {  // chunk tail, lane = pair
        lds_sync<SPLIT>();
        if (lane < nch) {
          const float4 r = L.red[lane];
          const float4 u = L.ud[lane];
          const float inv_d = __builtin_amdgcn_rcpf(u.w);
          const float dot = r.y * u.x + r.z * u.y + r.w * u.z;
          float f0 = (r.y - dot * u.x) * inv_d - r.x * u.x;
          float f1 = (r.z - dot * u.y) * inv_d - r.x * u.y;
          float f2 = (r.w - dot * u.z) * inv_d - r.x * u.z;
          if (STRESS) {
            const float hx = -u.x * u.w, hy = -u.y * u.w, hz = -u.z * u.w;
            W[0] += hx * f0; W[1] += hx * f1; W[2] += hx * f2;
            W[3] += hy * f0; W[4] += hy * f1; W[5] += hy * f2;
            W[6] += hz * f0; W[7] += hz * f1; W[8] += hz * f2;
          }
          float4* pb = pairbuf + (size_t)i * cap + m_lo + c0 + lane;
          if (pb_accum) {
            const float4 o = *pb;
            f0 += o.x; f1 += o.y; f2 += o.z;
          }
          *pb = make_float4(f0, f1, f2, 0.f);
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
    if (SPLIT) {  // waves 1..3 hand their partial sums to wave 0, which writes
      lds_sync<SPLIT>();
      if (wid > 0) {
        float* r = s_red + (wid - 1) * (4 * 64 + 14);
        r[lane] = ab[0].x; r[64 + lane] = ab[0].y; r[128 + lane] = ab[1].x; r[192 + lane] = ab[1].y;
        if (lane == 0) {
          r[256] = xa0; r[257] = xa1; r[258] = xa2;
#pragma unroll
          for (int ch = 0; ch < NQC; ++ch) r[259 + ch] = qacc[ch];
#pragma unroll
          for (int k = 0; k < 9; ++k) r[261 + k] = W[k];
        }
      }
      lds_sync<SPLIT>();
      if (wid == 0) {
#pragma unroll
        for (int w = 0; w < 3; ++w) {
          const float* r = s_red + w * (4 * 64 + 14);
          ab[0] += mk2(r[lane], r[64 + lane]);
          ab[1] += mk2(r[128 + lane], r[192 + lane]);
          xa0 += r[256]; xa1 += r[257]; xa2 += r[258];
#pragma unroll
          for (int ch = 0; ch < NQC; ++ch) qacc[ch] += r[259 + ch];
#pragma unroll
          for (int k = 0; k < 9; ++k) W[k] += r[261 + k];
        }
      }
    }
    if (live && (!SPLIT || wid == 0)) {
      if (NEED_ABAR) {
        const float4 xb = reinterpret_cast<const float4*>(xbar + (size_t)i * ldx)[lane];
        float4 o = make_float4(ab[0].x + xb.x, ab[0].y + xb.y, ab[1].x + xb.z, ab[1].y + xb.w);
        if (abar_in) {
          const float4 p = reinterpret_cast<const float4*>(abar_in + (size_t)i * NF)[lane];
          o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
        }
        reinterpret_cast<float4*>(abar_out + (size_t)i * NF)[lane] = o;
      }
      if (lane == 0) {
        if (!XE) {
          fgrad[3 * i + 0] += xa0;
          fgrad[3 * i + 1] += xa1;
          fgrad[3 * i + 2] += xa2;
        }
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

int launch_conv_bwd(hipStream_t s, int nq, bool need_abar, bool stress, const float* a, const int* row_of,
                    const float* q, const float* Sbar, const float* Sqbar, const int* nb_idx, const int* nb_cnt, const float4* pg,
                    int cap, BasisParams bp, const float* xbar, int ldx, const float* abar_in, float* abar_out,
                    const float* qbar_in, float* qbar_out, float* fgrad, float* virial_atom, int n_atoms, const int* order,
                    float4* pairbuf, bool pb_accum, int split_max) {
  const bool split = n_atoms <= split_max;
  const bool xe = pairbuf != nullptr && !split;  // reverse-pair form: F1 of every ordered pair into pairbuf (see the kernel)
  const int grid = split ? n_atoms : min(ceil_div(n_atoms, APB), device_cus() * 4);
#define AIMNET_BWD(HQ, NA, ST, SP, XE)                                                                                      \
  hipLaunchKernelGGL((conv_bwd_kernel<HQ, NA, ST, SP, XE>), dim3(grid), dim3(256), 0, s, a, row_of, q, Sbar, Sqbar, nb_idx,  \
                     nb_cnt, pg, cap, bp, xbar, ldx, abar_in, abar_out, qbar_in, qbar_out, fgrad, virial_atom, n_atoms, order, \
                     pairbuf, pb_accum ? 1 : 0)
#define AIMNET_BWD3(HQ, NA, ST)                                \
  do {                                                         \
    if (split) AIMNET_BWD(HQ, NA, ST, true, false);            \
    else if (xe) AIMNET_BWD(HQ, NA, ST, false, true);          \
    else AIMNET_BWD(HQ, NA, ST, false, false);                 \
  } while (0)
#define AIMNET_BWD2(HQ)                                                              \
  do {                                                                               \
    if (need_abar) { if (stress) AIMNET_BWD3(HQ, true, true); else AIMNET_BWD3(HQ, true, false); }   \
    else { if (stress) AIMNET_BWD3(HQ, false, true); else AIMNET_BWD3(HQ, false, false); }           \
  } while (0)
  if (nq == 2) AIMNET_BWD2(2);  // NSE models convolve charges in every pass that reaches here with need_abar
  else if (nq == 1) AIMNET_BWD2(1);
  else AIMNET_BWD2(0);
#undef AIMNET_BWD2
#undef AIMNET_BWD3
#undef AIMNET_BWD
  AIMNET_LAUNCH_CHECK();
  return 0;
}

// Reverse-pair map of a full, symmetric neighbour matrix: rev[i * cap + m] = position of (i, -shift) in the row of
// j = idx[i][m] (-1 if the row of j does not hold it: only after a row overflow).
// (A first version sorted every row by neighbour - bitonic network in LDS, 17 us - and bisected the neighbour's row, 22 us; the
// hash tables below take 9 + 12 us on config 3 and leave the rows in walk order.)
constexpr int REV_ROW_MAX = 128;  // row capacity: positions fit 8 bits and a 256-slot table stays at most half full
bool pair_rev_supported(int n_atoms, int cap) { return cap <= REV_ROW_MAX && n_atoms < (1 << 25); }

// (the hash build and the lookup live in pairmap.h as device functions: they also ride on other launches, kernels.h PairMapRider)
__global__ __launch_bounds__(256) void pair_hash_kernel(const int* __restrict__ nb_idx, const int* __restrict__ nb_shift,
                                                        const int* __restrict__ nb_cnt, int cap, int n_atoms,
                                                        unsigned long long* __restrict__ tab, int* __restrict__ rev) {
  __shared__ unsigned long long s_tab[4][RH_SLOTS];
  pair_hash_block(nb_idx, nb_shift, nb_cnt, cap, n_atoms, tab, rev, blockIdx.x, s_tab);
}

__global__ __launch_bounds__(256) void pair_rev_hash_kernel(const int* __restrict__ nb_idx, const int* __restrict__ nb_shift,
                                                            const int* __restrict__ nb_cnt, int cap, int n_atoms,
                                                            const unsigned long long* __restrict__ tab, int* __restrict__ rev) {
  pair_rev_hash_block(nb_idx, nb_shift, nb_cnt, cap, n_atoms, tab, rev, blockIdx.x);
}

size_t pair_hash_bytes(int n_atoms) { return (size_t)n_atoms * RH_SLOTS * sizeof(unsigned long long); }

int launch_pair_rev_hash(hipStream_t s, const int* nb_idx, const int* nb_shift, const int* nb_cnt, int cap, int n_atoms,
                         unsigned long long* tab, int* rev) {
  hipLaunchKernelGGL(pair_hash_kernel, dim3(ceil_div(n_atoms, 4)), dim3(256), 0, s, nb_idx, nb_shift, nb_cnt, cap, n_atoms, tab, rev);
  AIMNET_LAUNCH_CHECK();
  hipLaunchKernelGGL(pair_rev_hash_kernel, dim3(ceil_div(n_atoms, 4)), dim3(256), 0, s, nb_idx, nb_shift, nb_cnt, cap, n_atoms, tab,
                     rev);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

// forces_i = -(fgrad_i + sum_m F1(i -> j_m) - F1(j_m -> i))  (conv_bwd_kernel XE form; one wave per atom, lane = pair): the last
// kernel of the backward, it writes the force output itself
__global__ __launch_bounds__(256) void pair_force_kernel(const int* __restrict__ nb_idx, const int* __restrict__ nb_cnt,
                                                         const int* __restrict__ rev, const float4* __restrict__ pairbuf, int cap,
                                                         int n_atoms, const float* __restrict__ fgrad, float* __restrict__ forces,
                                                         int* __restrict__ nf) {
  pair_force_block(nb_idx, nb_cnt, rev, pairbuf, cap, n_atoms, fgrad, forces, blockIdx.x, nf);  // (pairmap.h)
}

int launch_pair_force(hipStream_t s, const int* nb_idx, const int* nb_cnt, const int* rev, const float4* pairbuf, int cap,
                      int n_atoms, const float* fgrad, float* forces, int* nf) {
  hipLaunchKernelGGL(pair_force_kernel, dim3(ceil_div(n_atoms, 4)), dim3(256), 0, s, nb_idx, nb_cnt, rev, pairbuf, cap, n_atoms,
                     fgrad, forces, nf);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Pass-0 backward through SPECIES MOMENTS.
// In the first pass the neighbour features are rows of the embedding table, a_j = afv[Z_j], so the only
// per-pair unknown of the contraction sum_{a} a_j[a,g] Sbar_i[a,g,c] is the species of j.  Contract once per
// atom and species,
//     T_i[s][g][c] = sum_a Sbar_i[a,g,c] * afv[z(s)][a,g]            (256 B per (atom, species present)),
// and the combined adjoints of conv_bwd_kernel (same algebra, same sign conventions) collapse to
//     D   = sum_g dgs_g ( T_i[s_j][g][0] + T_j[s_i][g][0] + u . (T_i[s_j][g][1:4] - T_j[s_i][g][1:4]) )
//     U_c = sum_g  gs_g ( T_j[s_i][g][c] - T_i[s_j][g][c] ),                     c = 1..3
// i.e. 64 + 64 floats per pair instead of a 1 KiB feature row and a 4 KiB Sbar row, and 16x fewer FMAs.
// Exact re-association of the reference sums (Warp backward_g, conv_sv_2d_sp_wp.py:139-164, with the AEV
// backward of aev.py:94-110); no gradient with respect to a_j is needed because afv is a constant.
__global__ void species_kernel(const int* __restrict__ numbers, const int* __restrict__ slot_of_z, int n_atoms,
                               int* __restrict__ aslot, unsigned long long* __restrict__ present_part) {
  __shared__ unsigned long long s_mask;
  if (threadIdx.x == 0) s_mask = 0ull;
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  unsigned long long m = 0ull;
  if (i < n_atoms) {
    const int sl = slot_of_z[min(63, max(0, numbers[i]))];
    aslot[i] = sl;
    m = 1ull << sl;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m |= __shfl_xor(m, off, 64);
  if ((threadIdx.x & 63) == 0) atomicOr(&s_mask, m);
  __syncthreads();
  if (threadIdx.x == 0) present_part[blockIdx.x] = s_mask;
}

int launch_species(hipStream_t s, const int* numbers, const int* slot_of_z, int n_atoms, int* aslot,
                   unsigned long long* present_part) {
  hipLaunchKernelGGL(species_kernel, dim3(ceil_div(n_atoms, 256)), dim3(256), 0, s, numbers, slot_of_z, n_atoms, aslot,
                     present_part);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

constexpr int P0_AFV_LDS = 8;  // embedding rows of the species present that unconcat_p0_kernel keeps in LDS

// xbar of pass 0 -> Sbar_i (registers/LDS only, never written to HBM) -> T_i[s] for the species present.
// Stage 1 is unconcat_kernel<false>'s arithmetic (lane = (a, 4 shifts)); stage 2 re-reads the row from LDS with
// lane = (g, c) and contracts over a.  Row stride 65 keeps both the (a,gq)-major writes and the (g,c)-major reads
// conflict free.
__global__ __launch_bounds__(256) void unconcat_p0_kernel(const float* __restrict__ xbar, int ldx,
                                                         const float* __restrict__ Vsave, const float* __restrict__ agh_a,
                                                         const float* __restrict__ afv, const int* __restrict__ z_of_slot,
                                                         int nslots, const unsigned long long* __restrict__ present_part,
                                                         int n_part, float* __restrict__ T, int n_atoms) {
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  float* s_agh = dsm;                                  // A*G*H
  float* s_vb = s_agh + A_ * G_ * H_;                  // [APB][NV*3]
  float* s_sb = s_vb + APB * NV * 3;                   // [APB][A*65]
  float* s_afv = s_sb + APB * A_ * 65;                 // [npres][256]
  __shared__ int s_plist[64];
  __shared__ int s_npres;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  // (the agh table is read into registers below; its LDS region stays unused)
  if (wid == 0) {
    unsigned long long m = 0ull;
    for (int k = lane; k < n_part; k += 64) m |= present_part[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m |= __shfl_xor(m, off, 64);
    if (lane < nslots && ((m >> lane) & 1ull)) s_plist[__popcll(m & ((1ull << lane) - 1ull))] = lane;
    if (lane == 0) s_npres = __popcll(m);
  }
  __syncthreads();
  const int npres = s_npres;
  // (LDS holds the embedding rows of the first P0_AFV_LDS species present; a batch with more reads the others from global
  // memory - sizing the buffer for every slot of the model cost two of the three resident blocks per CU)
  for (int k = threadIdx.x; k < min(npres, P0_AFV_LDS) * NF; k += 256)
    s_afv[k] = afv[(size_t)z_of_slot[s_plist[k >> 8]] * NF + (k & 255)];
  float* vb = s_vb + wid * (NV * 3);
  float* sb = s_sb + wid * (A_ * 65);
  const AtomLoop al = atom_loop(n_atoms, APB);
  float wreg[4][H_];  // the lane's agh weights in registers (see unconcat_kernel: the LDS reads were a 32-way bank conflict)
#pragma unroll
  for (int gi = 0; gi < 4; ++gi)
#pragma unroll
    for (int h = 0; h < H_; ++h) wreg[gi][h] = agh_a[((lane >> 2) * G_ + (lane & 3) * 4 + gi) * H_ + h];
  __syncthreads();  // species list and embedding rows (shared, read-only from here on); vb / sb are per wave
  // wave-level hand-offs and the next atom's operands requested under the current atom's contractions (see unconcat_kernel)
  float pf_f[3], pf_v[3][3];
  float4 pf_s0 = make_float4(0.f, 0.f, 0.f, 0.f);
  auto load_atom = [&](int i) __attribute__((always_inline)) {
    const float* xr = xbar + (size_t)i * ldx;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int o = lane + 64 * t;
      pf_f[t] = xr[2 * NF + o];
      const float* vs = Vsave + (size_t)i * (NV * 3) + o;  // planes [k][NV], see conv_fwd_kernel
      pf_v[t][0] = vs[0]; pf_v[t][1] = vs[NV]; pf_v[t][2] = vs[2 * NV];
    }
    pf_s0 = reinterpret_cast<const float4*>(xr + NF)[lane];
  };
  int i = al.first + wid;
  bool live = i < al.last;
  if (live) load_atom(i);
  for (int i0 = al.first; i0 < al.last; i0 += al.step) {
    float s0v[4] = {0.f, 0.f, 0.f, 0.f};
    if (live) {
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int o = lane + 64 * t;
        const float f = 2.0f * pf_f[t];
        vb[o * 3 + 0] = f * pf_v[t][0];
        vb[o * 3 + 1] = f * pf_v[t][1];
        vb[o * 3 + 2] = f * pf_v[t][2];
      }
      s0v[0] = pf_s0.x; s0v[1] = pf_s0.y; s0v[2] = pf_s0.z; s0v[3] = pf_s0.w;
    }
    lds_sync<false>();
    const int i_next = i + al.step;
    const bool live_next = i_next < al.last;
    if (live_next) load_atom(i_next);
    if (live) {
      const int aa = lane >> 2, gq = lane & 3;
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) {
        const int g = gq * 4 + gi;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
#pragma unroll
        for (int h = 0; h < H_; ++h) {
          const float w = wreg[gi][h];
          const float* v3 = &vb[(aa * H_ + h) * 3];
          v0 += w * v3[0];
          v1 += w * v3[1];
          v2 += w * v3[2];
        }
        float* o = sb + aa * 65 + g * 4;
        o[0] = s0v[gi]; o[1] = v0; o[2] = v1; o[3] = v2;
      }
    }
    lds_sync<false>();
    if (live) {
      float sr[A_];
#pragma unroll
      for (int a = 0; a < A_; ++a) sr[a] = sb[a * 65 + lane];
      const int g = lane >> 2;
      for (int k = 0; k < npres; ++k) {
        const float* av = (k < P0_AFV_LDS ? s_afv + k * NF : afv + (size_t)z_of_slot[s_plist[k]] * NF) + g;
        float t = 0.f;
#pragma unroll
        for (int a = 0; a < A_; ++a) t += sr[a] * av[a * G_];
        T[((size_t)i * nslots + s_plist[k]) * 64 + lane] = t;
      }
    }
    lds_sync<false>();  // sb / vb are free for the next atom
    i = i_next;
    live = live_next;
  }
}

int launch_unconcat_p0(hipStream_t s, const float* xbar, int ldx, const float* Vsave, const float* agh_a, const float* afv,
                       const int* z_of_slot, int nslots, const unsigned long long* present_part, int n_part, float* T,
                       int n_atoms) {
  const int grid = min(ceil_div(n_atoms, APB), 256 * 8);
  const size_t lds = sizeof(float) * ((size_t)A_ * G_ * H_ + APB * NV * 3 + APB * A_ * 65 + (size_t)min(nslots, P0_AFV_LDS) * NF);
  static PerDeviceOnce once;
  if (once.first())
    AIMNET_HIP_CHECK(hipFuncSetAttribute((const void*)unconcat_p0_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  hipLaunchKernelGGL(unconcat_p0_kernel, dim3(grid), dim3(256), lds, s, xbar, ldx, Vsave, agh_a, afv, z_of_slot, nslots,
                     present_part, n_part, T, n_atoms);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

struct P0WaveLds {
  float4 ud[64];
  float4 du[64];  // per pair: (D, U0, U1, U2), handed from the (pair-of-four, g) lane map of the radial sums to the lane = pair map
  float fc[64], dfc[64];
  int j[64], sj[64];
};

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over each aligned group of 16 lanes, result in all 16 (quad xor 1, quad xor 2, row_half_mirror, row_mirror)
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  return v;
}

// One wave per centre atom; 4 neighbours per step, 16 lanes (one per shift g) each: the two 256 B moment blocks of a
// pair are single coalesced float4 loads, the radial basis is evaluated exactly once per (pair, g).
// XE (reverse-pair form, with conv_bwd_kernel<.., XE> in the later passes): the pair (i -> j) only evaluates the half built
// on ITS OWN moments T_i[s_j] - the other half, T_j[s_i], is that same expression of the reverse pair with u -> -u - and adds
//   G1(p) = (U - (U.u) u) / d - D u,   D = sum_g dgs_g (T_i[s_j][g][0] + u . T_i[s_j][g][1:4]),  U = -sum_g gs_g T_i[s_j][g][1:4]
// to the pair buffer; pair_force_kernel forms dE/dx_i = sum_m G1(i -> j_m) - G1(j_m -> i) together with the other passes'
// entries.  No neighbour moments are gathered at all.
template <bool STRESS, bool XE>
__global__ __launch_bounds__(256) void conv_bwd_p0_kernel(const float4* __restrict__ T4, int nslots,
                                                         const int* __restrict__ aslot, const int* __restrict__ nb_idx,
                                                         const int* __restrict__ nb_cnt, const float4* __restrict__ pg,
                                                         int cap, BasisParams bp, float* __restrict__ fgrad,
                                                         float* __restrict__ virial_atom, int n_atoms,
                                                         const int* __restrict__ order, float4* __restrict__ pairbuf) {
  __shared__ __attribute__((aligned(16))) P0WaveLds wl[APB];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  P0WaveLds& L = wl[wid];
  const int g = lane & 15, pq = lane >> 4;
  const float shift_g = bp.shifts[g];
  const AtomLoop al = atom_loop(n_atoms, APB);
  for (int i0 = al.first; i0 < al.last; i0 += al.step) {
    if (i0 + wid >= al.last) continue;  // no block barriers below: the LDS staging is private to the wave
    const int i = order ? order[i0 + wid] : i0 + wid;
    const int cnt = nb_cnt[i];
    const int si = aslot[i];
    const float4* Ti_base = T4 + (size_t)i * nslots * 16 + g;
    float xa0 = 0.f, xa1 = 0.f, xa2 = 0.f;
    float W[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) W[k] = 0.0f;
    for (int c0 = 0; c0 < cnt; c0 += 64) {
      const int nch = min(64, cnt - c0);
      __builtin_amdgcn_wave_barrier();
      float4 ud_own = make_float4(0.f, 0.f, 0.f, 1.f);
      if (lane < nch) {  // lane = pair: one sincos per pair
        const size_t p = (size_t)i * cap + c0 + lane;
        const int j = nb_idx[p];
        ud_own = pg[p];
        L.j[lane] = j;
        L.sj[lane] = aslot[j];
        L.ud[lane] = ud_own;
        float dfc;
        L.fc[lane] = basis_fc(bp, ud_own.w, dfc);
        L.dfc[lane] = dfc;
      }
      __builtin_amdgcn_wave_barrier();
      // phase 1, lanes = (pair of the step, g): the radial sums D and U of four pairs per step
      for (int m0 = 0; m0 < nch; m0 += 4) {
        const int m = min(m0 + pq, nch - 1);
        const bool valid = m0 + pq < nch;
        const int j = L.j[m], sj = L.sj[m];
        const float4 u = L.ud[m];
        const float fc = L.fc[m], dfc = L.dfc[m];
        const float4 Ti = Ti_base[(size_t)sj * 16];
        const float4 Tj = XE ? make_float4(0.f, 0.f, 0.f, 0.f) : T4[((size_t)j * nslots + si) * 16 + g];
        const float dd = u.w - shift_g;
        const float Gg = exp_neg(-bp.eta * dd * dd);
        const float gs = Gg * fc;
        const float dg = Gg * (dfc - 2.0f * bp.eta * dd * fc);
        float D = dg * ((Ti.x + Tj.x) + u.x * (Ti.y - Tj.y) + u.y * (Ti.z - Tj.z) + u.z * (Ti.w - Tj.w));
        float U0 = gs * (Tj.y - Ti.y), U1 = gs * (Tj.z - Ti.z), U2 = gs * (Tj.w - Ti.w);
        D = row16_sum(D); U0 = row16_sum(U0); U1 = row16_sum(U1); U2 = row16_sum(U2);
        if (valid && g == 0) L.du[m] = make_float4(D, U0, U1, U2);
      }
      __builtin_amdgcn_wave_barrier();
      // phase 2, lane = pair: the projection onto the pair's geometry, the pair-buffer entry (one coalesced 16-byte access per
      // lane instead of four scattered ones per step) and the virial terms - once per pair, not once per (pair, g)
      if (lane < nch) {
        const float4 u = ud_own;
        const float4 du = L.du[lane];
        const float D = du.x, U0 = du.y, U1 = du.z, U2 = du.w;
        const float inv_d = __builtin_amdgcn_rcpf(u.w);
        const float dot = U0 * u.x + U1 * u.y + U2 * u.z;
        const float f0 = (U0 - dot * u.x) * inv_d - D * u.x;
        const float f1 = (U1 - dot * u.y) * inv_d - D * u.y;
        const float f2 = (U2 - dot * u.z) * inv_d - D * u.z;
        xa0 += f0; xa1 += f1; xa2 += f2;
        if (XE) {  // the passes before this one left their F1 here
          float4* pb = pairbuf + (size_t)i * cap + c0 + lane;
          const float4 o = *pb;
          *pb = make_float4(o.x + f0, o.y + f1, o.z + f2, 0.f);
        }
        if (STRESS) {
          const float hs = XE ? -1.0f : -0.5f;  // XE: one ordered pair carries the whole -r (x) G1 term
          const float hx = hs * u.x * u.w, hy = hs * u.y * u.w, hz = hs * u.z * u.w;
          W[0] += hx * f0; W[1] += hx * f1; W[2] += hx * f2;
          W[3] += hy * f0; W[4] += hy * f1; W[5] += hy * f2;
          W[6] += hz * f0; W[7] += hz * f1; W[8] += hz * f2;
        }
      }
    }
    xa0 = wave_sum(xa0); xa1 = wave_sum(xa1); xa2 = wave_sum(xa2);
    if (STRESS) {
#pragma unroll
      for (int k = 0; k < 9; ++k) W[k] = wave_sum(W[k]);
    }
    if (!XE && lane == 0) {
      fgrad[3 * i + 0] += xa0;
      fgrad[3 * i + 1] += xa1;
      fgrad[3 * i + 2] += xa2;
    }
    if (STRESS && lane < 9) {
      float v = W[0];
#pragma unroll
      for (int k = 1; k < 9; ++k) v = (lane == k) ? W[k] : v;
      virial_atom[(size_t)i * 9 + lane] += v;
    }
  }
}

int launch_conv_bwd_p0(hipStream_t s, bool stress, const float* T, int nslots, const int* aslot, const int* nb_idx,
                       const int* nb_cnt, const float4* pg, int cap, BasisParams bp, float* fgrad, float* virial_atom,
                       int n_atoms, const int* order, float4* pairbuf) {
  const int grid = min(ceil_div(n_atoms, APB), 256 * 8);
#define AIMNET_P0(ST, XE_)                                                                                                  \
  hipLaunchKernelGGL((conv_bwd_p0_kernel<ST, XE_>), dim3(grid), dim3(256), 0, s, reinterpret_cast<const float4*>(T), nslots, aslot, \
                     nb_idx, nb_cnt, pg, cap, bp, fgrad, virial_atom, n_atoms, order, pairbuf)
  if (pairbuf) { if (stress) AIMNET_P0(true, true); else AIMNET_P0(false, true); }
  else { if (stress) AIMNET_P0(true, false); else AIMNET_P0(false, false); }
#undef AIMNET_P0
  AIMNET_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// stand-alone op with the exact contract of torch.ops.aimnet.conv_sv_2d_sp_fwd / _bwd
// (conv_sv_2d_sp_wp.py:252-340): generic (A, G) with A*G a multiple of 4, materialised g (B,M,G,4),
// sentinel index B-1, rows packed real-first (early exit at the first sentinel), padding row zero.
// out = conv(a, g) [+ conv(a2, g2) when a2 != NULL: the two terms of the double backward's grad_grad_output]
__global__ void conv_sv_fwd_kernel(const float* __restrict__ a, const int* __restrict__ idx, const float4* __restrict__ g,
                                   const float* __restrict__ a2, const float4* __restrict__ g2, float4* __restrict__ out,
                                   int B, int A, int G, int M) {
  const int b = blockIdx.x;
  const int AG = A * G;
  for (int f = threadIdx.x; f < AG; f += blockDim.x) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (b < B - 1) {
      const int gg = f % G;
      for (int m = 0; m < M; ++m) {
        const int j = idx[(size_t)b * M + m];
        if (j >= B - 1) break;
        const float av = a[(size_t)j * AG + f];
        const float4 gv = g[((size_t)b * M + m) * G + gg];
        acc.x += av * gv.x; acc.y += av * gv.y; acc.z += av * gv.z; acc.w += av * gv.w;
        if (a2) {
          const float bv = a2[(size_t)j * AG + f];
          const float4 hv = g2[((size_t)b * M + m) * G + gg];
          acc.x += bv * hv.x; acc.y += bv * hv.y; acc.z += bv * hv.z; acc.w += bv * hv.w;
        }
      }
    }
    out[(size_t)b * AG + f] = acc;
  }
}

// ---- A = G = 16 fast forms of the stand-alone op (the shape every shipped model uses) ------------------------------------
// One wave per row b, the engine's lane map: lane = (a = l >> 2, gq = l & 3) owns out[b, a, 4gq..4gq+3, 0..3].  A neighbour's
// feature row is one coalesced 1 KiB wave load, its g block (256 B) is read as four float4 per lane (16 lanes share each);
// the row's indices sit in registers (lane-distributed), so no row load waits for an index load.  M <= 256.
constexpr int SV_MAXW = 4;  // 64-entry index words per row

struct SvRow {
  int v[SV_MAXW];  // idx[b, 64 k + lane]
  int n;           // real entries (rows are packed real-first: everything from the first sentinel on is padding)
};
__device__ __forceinline__ SvRow sv_load_row(const int* __restrict__ idx, int b, int B, int M, int lane) {
  SvRow r;
  r.n = M;
#pragma unroll
  for (int k = 0; k < SV_MAXW; ++k) {
    const int m = 64 * k + lane;
    r.v[k] = (m < M) ? idx[(size_t)b * M + m] : B - 1;
  }
#pragma unroll
  for (int k = SV_MAXW - 1; k >= 0; --k) {
    const unsigned long long pad = __ballot(r.v[k] >= B - 1);
    if (pad) r.n = 64 * k + __ffsll((long long)pad) - 1;
  }
  r.n = min(r.n, M);
  return r;
}
__device__ __forceinline__ int sv_entry(const SvRow& r, int m) {  // wave-uniform m
  int v = 0;
#pragma unroll
  for (int k = 0; k < SV_MAXW; ++k)
    if ((m >> 6) == k) v = __builtin_amdgcn_readlane(r.v[k], m & 63);
  return v;
}
// first slot of row `r` that holds value x, or -1
__device__ __forceinline__ int sv_find(const SvRow& r, int x) {
  int pos = -1;
#pragma unroll
  for (int k = SV_MAXW - 1; k >= 0; --k) {
    const unsigned long long hit = __ballot(r.v[k] == x);
    if (hit) pos = 64 * k + __ffsll((long long)hit) - 1;
  }
  return (pos >= 0 && pos < r.n) ? pos : -1;
}

__global__ __launch_bounds__(256) void conv_sv_fwd16_kernel(const float* __restrict__ a, const int* __restrict__ idx,
                                                           const float4* __restrict__ g, const float* __restrict__ a2,
                                                           const float4* __restrict__ g2, float4* __restrict__ out, int B, int M) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int gq = lane & 3;
  float4 acc[4];
#pragma unroll
  for (int gi = 0; gi < 4; ++gi) acc[gi] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (b < B - 1) {
    const SvRow r = sv_load_row(idx, b, B, M, lane);
    auto fma4 = [&](const float4& av, const float4* gp) {
      const float avv[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) {
        const float4 gv = gp[gi];
        acc[gi].x += avv[gi] * gv.x; acc[gi].y += avv[gi] * gv.y; acc[gi].z += avv[gi] * gv.z; acc[gi].w += avv[gi] * gv.w;
      }
    };
    for (int m0 = 0; m0 < r.n; m0 += 4) {  // four rows in flight
      float4 av[4], bv[4];
      float4 gv[4][4], hv[4][4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int m = min(m0 + t, r.n - 1);
        const int j = sv_entry(r, m);
        av[t] = reinterpret_cast<const float4*>(a + (size_t)j * 256)[lane];
        const float4* gp = g + ((size_t)b * M + m) * 16 + 4 * gq;
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) gv[t][gi] = gp[gi];
        if (a2) {
          bv[t] = reinterpret_cast<const float4*>(a2 + (size_t)j * 256)[lane];
          const float4* hp = g2 + ((size_t)b * M + m) * 16 + 4 * gq;
#pragma unroll
          for (int gi = 0; gi < 4; ++gi) hv[t][gi] = hp[gi];
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (m0 + t < r.n) {
          fma4(av[t], gv[t]);
          if (a2) fma4(bv[t], hv[t]);
        }
      }
    }
  }
  float4* o = out + (size_t)b * 256 + lane * 4;
#pragma unroll
  for (int gi = 0; gi < 4; ++gi) o[gi] = acc[gi];
}

// grad_g[b,m,g,:] = sum_a a[idx[b,m],a,g] * grad_out[b,a,g,:]: lane = (g, c) keeps grad_out[b, 0..15, g, c] in registers, the
// neighbour's feature row goes through LDS (coalesced 1 KiB in, 16 conflict-free dword reads out)
__global__ __launch_bounds__(256) void conv_sv_bwd_g16_kernel(const float* __restrict__ grad_out, const float* __restrict__ a,
                                                             const int* __restrict__ idx, float* __restrict__ grad_g, int B,
                                                             int M) {
  __shared__ __attribute__((aligned(16))) float s_row[4][2][256];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, b = blockIdx.x * 4 + wid;
  if (b >= B) return;
  const int gg = lane >> 2;
  float* gout = grad_g + (size_t)b * M * 64;
  if (b >= B - 1) {
    for (int m = 0; m < M; ++m) gout[(size_t)m * 64 + lane] = 0.f;
    return;
  }
  float go[16];
#pragma unroll
  for (int aa = 0; aa < 16; ++aa) go[aa] = grad_out[((size_t)b * 256 + aa * 16) * 4 + lane];
  const SvRow r = sv_load_row(idx, b, B, M, lane);
  float4 nxt = make_float4(0.f, 0.f, 0.f, 0.f);
  if (r.n > 0) nxt = reinterpret_cast<const float4*>(a + (size_t)sv_entry(r, 0) * 256)[lane];
  for (int m = 0; m < r.n; ++m) {
    float* sr = s_row[wid][m & 1];
    reinterpret_cast<float4*>(sr)[lane] = nxt;
    if (m + 1 < r.n) nxt = reinterpret_cast<const float4*>(a + (size_t)sv_entry(r, m + 1) * 256)[lane];
    __builtin_amdgcn_wave_barrier();
    float acc = 0.f;
#pragma unroll
    for (int aa = 0; aa < 16; ++aa) acc += sr[aa * 16 + gg] * go[aa];
    gout[(size_t)m * 64 + lane] = acc;
  }
  for (int m = r.n; m < M; ++m) gout[(size_t)m * 64 + lane] = 0.f;
}

// grad_a, deterministic part: the op's contract allows any idx, but the lists it is called with are symmetric (j in row b
// <=> b in row j).  Destination row i GATHERS: for the first occurrence of every j in its own row it looks i up in row j
// (first slot m') and adds <grad_out[j], g[j, m']> - plain stores, fixed order.  Sources this does not reach (a repeated j,
// or a b that row j does not hold) are added by conv_sv_bwd_a16_rest_kernel with atomics, exactly as the Warp kernel adds
// everything (conv_sv_2d_sp_wp.py:115-136); on a symmetric, duplicate-free list that kernel finds nothing to do.
__global__ __launch_bounds__(256) void conv_sv_bwd_a16_gather_kernel(const float4* __restrict__ grad_out,
                                                                    const int* __restrict__ idx, const float4* __restrict__ g,
                                                                    float4* __restrict__ grad_a, int B, int M) {
  const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= B) return;
  const int gq = lane & 3;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < B - 1) {
    const SvRow r = sv_load_row(idx, i, B, M, lane);
    for (int m = 0; m < r.n; ++m) {
      const int j = sv_entry(r, m);
      if (sv_find(r, j) != m) continue;  // a repeated neighbour: left to the atomic pass
      const SvRow rj = sv_load_row(idx, j, B, M, lane);
      const int mp = sv_find(rj, i);
      if (mp < 0) continue;
      const float4* gop = grad_out + (size_t)j * 256 + lane * 4;
      const float4* gp = g + ((size_t)j * M + mp) * 16 + 4 * gq;
      float d[4];
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) {
        const float4 o = gop[gi], w = gp[gi];
        d[gi] = o.x * w.x + o.y * w.y + o.z * w.z + o.w * w.w;
      }
      acc.x += d[0]; acc.y += d[1]; acc.z += d[2]; acc.w += d[3];
    }
  }
  grad_a[(size_t)i * 64 + lane] = acc;
}

__global__ __launch_bounds__(256) void conv_sv_bwd_a16_rest_kernel(const float4* __restrict__ grad_out,
                                                                  const int* __restrict__ idx, const float4* __restrict__ g,
                                                                  float* __restrict__ grad_a, int B, int M) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B - 1) return;
  const int gq = lane & 3;
  const SvRow r = sv_load_row(idx, b, B, M, lane);
  for (int m = 0; m < r.n; ++m) {
    const int j = sv_entry(r, m);
    bool consumed = sv_find(r, j) == m;  // the gather at j only takes the FIRST slot of b's row that names j ...
    if (consumed) {
      const SvRow rj = sv_load_row(idx, j, B, M, lane);
      consumed = sv_find(rj, b) >= 0;  // ... and only if j's row names b at all
    }
    if (consumed) continue;
    const float4* gop = grad_out + (size_t)b * 256 + lane * 4;
    const float4* gp = g + ((size_t)b * M + m) * 16 + 4 * gq;
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
      const float4 o = gop[gi], w = gp[gi];
      atomicAdd(&grad_a[(size_t)j * 256 + lane * 4 + gi], o.x * w.x + o.y * w.y + o.z * w.z + o.w * w.w);
    }
  }
}

int launch_conv_sv_fwd(hipStream_t s, const float* a, const int* idx, const float* g, float* out, int B, int A, int G,
                       int M) {
  if (B <= 0) return 0;
  if (A == 16 && G == 16 && M <= 64 * SV_MAXW)
    hipLaunchKernelGGL(conv_sv_fwd16_kernel, dim3(ceil_div(B, 4)), dim3(256), 0, s, a, idx, (const float4*)g, nullptr, nullptr,
                       (float4*)out, B, M);
  else
    hipLaunchKernelGGL(conv_sv_fwd_kernel, dim3(B), dim3(256), 0, s, a, idx, (const float4*)g, nullptr, nullptr, (float4*)out, B,
                       A, G, M);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

// grad_g[b,m,g,:] = sum_a a[idx[b,m],a,g] * grad_out[b,a,g,:]
__global__ void conv_sv_bwd_g_kernel(const float4* __restrict__ grad_out, const float* __restrict__ a,
                                     const int* __restrict__ idx, float4* __restrict__ grad_g, int B, int A, int G, int M) {
  const int b = blockIdx.x;
  for (int e = threadIdx.x; e < M * G; e += blockDim.x) {
    const int m = e / G, gg = e % G;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int j = idx[(size_t)b * M + m];
    if (b < B - 1 && j < B - 1) {
      for (int aa = 0; aa < A; ++aa) {
        const float av = a[((size_t)j * A + aa) * G + gg];
        const float4 go = grad_out[((size_t)b * A + aa) * G + gg];
        acc.x += av * go.x; acc.y += av * go.y; acc.z += av * go.z; acc.w += av * go.w;
      }
    }
    grad_g[((size_t)b * M + m) * G + gg] = acc;
  }
}

// grad_a[j,a,g] += <grad_out[b,a,g,:], g[b,m,g,:]> for j = idx[b,m]: the op contract allows an
// arbitrary (non-symmetric) idx, so this stand-alone form scatters with atomics like the Warp
// kernel (conv_sv_2d_sp_wp.py:115-136); the engine itself never uses it (centre-major gather).
__global__ void conv_sv_bwd_a_kernel(const float4* __restrict__ grad_out, const int* __restrict__ idx,
                                     const float4* __restrict__ g, float* __restrict__ grad_a, int B, int A, int G, int M) {
  const int b = blockIdx.x;
  if (b >= B - 1) return;
  const int AG = A * G;
  for (int f = threadIdx.x; f < AG; f += blockDim.x) {
    const int gg = f % G;
    const float4 go = grad_out[(size_t)b * AG + f];
    for (int m = 0; m < M; ++m) {
      const int j = idx[(size_t)b * M + m];
      if (j >= B - 1) break;
      const float4 gv = g[((size_t)b * M + m) * G + gg];
      atomicAdd(&grad_a[(size_t)j * AG + f], go.x * gv.x + go.y * gv.y + go.z * gv.z + go.w * gv.w);
    }
  }
}

#define RC_SV(call)       \
  do {                    \
    int _rc = (call);     \
    if (_rc) return _rc;  \
  } while (0)

// grad_a[j] = sum over (b, m) with idx[b,m] = j of <grad_out[b], g[b,m]>: deterministic gather + atomic remainder (see the kernels)
static int sv_bwd_a16(hipStream_t s, const float* grad_out, const int* idx, const float* g, float* grad_a, int B, int M) {
  hipLaunchKernelGGL(conv_sv_bwd_a16_gather_kernel, dim3(ceil_div(B, 4)), dim3(256), 0, s, (const float4*)grad_out, idx,
                     (const float4*)g, (float4*)grad_a, B, M);
  AIMNET_LAUNCH_CHECK();
  hipLaunchKernelGGL(conv_sv_bwd_a16_rest_kernel, dim3(ceil_div(B, 4)), dim3(256), 0, s, (const float4*)grad_out, idx,
                     (const float4*)g, grad_a, B, M);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

int launch_conv_sv_bwd(hipStream_t s, const float* grad_out, const float* a, const int* idx, const float* g,
                       float* grad_a, float* grad_g, int B, int A, int G, int M) {
  if (B <= 0) return 0;
  if (A == 16 && G == 16 && M <= 64 * SV_MAXW) {
    hipLaunchKernelGGL(conv_sv_bwd_g16_kernel, dim3(ceil_div(B, 4)), dim3(256), 0, s, grad_out, a, idx, grad_g, B, M);
    AIMNET_LAUNCH_CHECK();
    RC_SV(sv_bwd_a16(s, grad_out, idx, g, grad_a, B, M));
    return 0;
  }
  AIMNET_HIP_CHECK(hipMemsetAsync(grad_a, 0, (size_t)B * A * G * sizeof(float), s));
  hipLaunchKernelGGL(conv_sv_bwd_g_kernel, dim3(B), dim3(256), 0, s, (const float4*)grad_out, a, idx, (float4*)grad_g, B, A,
                     G, M);
  AIMNET_LAUNCH_CHECK();
  hipLaunchKernelGGL(conv_sv_bwd_a_kernel, dim3(B), dim3(256), 0, s, (const float4*)grad_out, idx, (const float4*)g, grad_a,
                     B, A, G, M);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

// Double backward of the op (conv_sv_2d_sp_wp.py:342-446): with cotangents grad2_a of grad_a and grad2_g of grad_g,
//   grad_grad_output = conv(grad2_a, g) + conv(a, grad2_g)      (one fused launch)
//   grad_a_double    = backward_a(grad_out; g := grad2_g)        (d grad_g / d a)
//   grad_g_double    = backward_g(grad_out; a := grad2_a)        (d grad_a / d g)
// i.e. the forward and the two halves of the backward with swapped operands - no new arithmetic.
int launch_conv_sv_bwd_bwd(hipStream_t s, const float* grad_out, const float* grad2_a, const float* grad2_g, const float* a,
                           const int* idx, const float* g, float* ggo, float* ga2, float* gg2, int B, int A, int G, int M) {
  if (B <= 0) return 0;
  if (A == 16 && G == 16 && M <= 64 * SV_MAXW) {
    hipLaunchKernelGGL(conv_sv_fwd16_kernel, dim3(ceil_div(B, 4)), dim3(256), 0, s, grad2_a, idx, (const float4*)g, a,
                       (const float4*)grad2_g, (float4*)ggo, B, M);
    AIMNET_LAUNCH_CHECK();
    RC_SV(sv_bwd_a16(s, grad_out, idx, grad2_g, ga2, B, M));
    hipLaunchKernelGGL(conv_sv_bwd_g16_kernel, dim3(ceil_div(B, 4)), dim3(256), 0, s, grad_out, grad2_a, idx, gg2, B, M);
    AIMNET_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(conv_sv_fwd_kernel, dim3(B), dim3(256), 0, s, grad2_a, idx, (const float4*)g, a, (const float4*)grad2_g,
                     (float4*)ggo, B, A, G, M);
  AIMNET_LAUNCH_CHECK();
  AIMNET_HIP_CHECK(hipMemsetAsync(ga2, 0, (size_t)B * A * G * sizeof(float), s));
  hipLaunchKernelGGL(conv_sv_bwd_a_kernel, dim3(B), dim3(256), 0, s, (const float4*)grad_out, idx, (const float4*)grad2_g, ga2, B,
                     A, G, M);
  AIMNET_LAUNCH_CHECK();
  hipLaunchKernelGGL(conv_sv_bwd_g_kernel, dim3(B), dim3(256), 0, s, (const float4*)grad_out, grad2_a, idx, (float4*)gg2, B, A, G,
                     M);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

}  // namespace aimnet
