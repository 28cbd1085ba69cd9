// model.hip - the small per-atom / per-molecule kernels around the GEMMs and convolutions.
//
// Reference semantics (paths relative to /root/reference/aimnet):
//   nse_fwd         ops.nse ops.py:99-145 + AIMNet2._update_q models/aimnet2.py:122-139
//   energy_reduce   AtomicShift + AtomicSum modules/core.py:71-111 (fp64 SAE, fp64 molecule sums)
//   coulomb_sr      SRCoulomb / _calc_coulomb_sr modules/lr.py:21-62,986-1032 (exp_cutoff ops.py:88-90)
//   coulomb_simple  LRCoulomb.coul_simple lr.py:311-331
//   coulomb_dsf     LRCoulomb._coul_dsf_torch lr.py:559-615
//   nse_bwd / build_zbar   adjoint of ops.nse
//   finalize        forces = -dE/dx, stress = (dE/deps)/|det C| calculators/derivatives.py:118-137
// Per-molecule reductions run one block per molecule in a fixed tree order (deterministic); pair
// energies are accumulated in fp64 exactly where the reference does (lr.py:61,326,602,611).
#include "cellwalk.h"
#include "common.h"
#include "gemm_h2_common.h"
#include "kernels.h"
#include "pairmap.h"

namespace aimnet {

template <typename T>
__device__ __forceinline__ T block_sum(T v, T* sh) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) sh[wid] = v;
  __syncthreads();
  T r = 0;
  const int nw = (blockDim.x + 63) >> 6;
  for (int w = 0; w < nw; ++w) r += sh[w];
  return r;
}

struct UpdateA {  // a_new = a + delta_a (+ the transposed copy): arguments of update_a, also carried by the NSE launches below
  const float* a;
  const int* row_of;
  const float* y;
  int ldy, col0, n_atoms;
  float* a_new;
  float* a_t;
};

__device__ __forceinline__ void update_a_block(const UpdateA& u, size_t block) {
  const size_t e = block * 256 + threadIdx.x;
  if (e >= (size_t)u.n_atoms * 256) return;
  const size_t i = e >> 8;
  const int k = (int)(e & 255);
  const size_t ri = u.row_of ? (size_t)min(63, max(0, u.row_of[i])) : i;
  const float v = u.a[ri * 256 + k] + u.y[i * u.ldy + u.col0 + k];
  u.a_new[e] = v;
  if (u.a_t) {
    const int aa = k >> 4, g = k & 15;
    u.a_t[i * 256 + g * 16 + aa] = v;
  }
}

// ---- NSE forward -------------------------------------------------------------------------------
// One charge channel per launch: the channel's q~ and f~ sit in columns qcol / fcol of the MLP output row
// (aimnet2.py:123-130: split [nq, nq, rest]); q planes, charge, Fm, Dm are that channel's [N] / [n_mol] arrays.
// (blocks beyond the n_mol molecule blocks run update_a: a_new = a + delta_a reads the same MLP output rows and depends on nothing
// the charge update produces, so the two share one launch - every kernel boundary costs 4-5 us on the device)
__global__ __launch_bounds__(256) void nse_fwd_kernel(const float* __restrict__ y, int ldy, int qcol, int fcol,
                                                     const float* __restrict__ q_prev,
                                                     const int* __restrict__ mol_start, const float* __restrict__ charge,
                                                     float* __restrict__ q_new, float* __restrict__ Fm,
                                                     float* __restrict__ Dm, int n_mol, UpdateA upd) {
  if ((int)blockIdx.x >= n_mol) {
    update_a_block(upd, blockIdx.x - n_mol);
    return;
  }
  __shared__ float sh[4];
  const int m = blockIdx.x;
  const int i0 = mol_start[m], i1 = mol_start[m + 1];
  float sf = 0.f, sq = 0.f;
  for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
    const float ft = y[(size_t)i * ldy + fcol];
    sf += ft * ft;
    sq += (q_prev ? q_prev[i] : 0.0f) + y[(size_t)i * ldy + qcol];
  }
  const float F = block_sum(sf, sh) + 1.0e-6f;
  const float D = charge[m] - block_sum(sq, sh);
  if (threadIdx.x == 0) {
    Fm[m] = F;
    Dm[m] = D;
  }
  for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
    const float ft = y[(size_t)i * ldy + fcol];
    const float qr = (q_prev ? q_prev[i] : 0.0f) + y[(size_t)i * ldy + qcol];
    q_new[i] = qr + (ft * ft) / F * D;
  }
}

// Sliced variants for large molecules: S blocks per molecule write partial sums in a fixed layout,
// every consumer adds the S partials in slice order -> still deterministic, but a 10k-atom
// molecule is reduced by S blocks instead of one.
__device__ __forceinline__ void slice_bounds(const int* __restrict__ mol_start, int m, int s, int S, int& lo, int& hi) {
  const int i0 = mol_start[m], i1 = mol_start[m + 1];
  const int L = (i1 - i0 + S - 1) / S;
  lo = min(i1, i0 + s * L);
  hi = min(i1, lo + L);
}

__global__ __launch_bounds__(256) void nse_fwd_partial_kernel(const float* __restrict__ y, int ldy, int qcol, int fcol,
                                                             const float* __restrict__ q_prev,
                                                             const int* __restrict__ mol_start, int S,
                                                             float* __restrict__ part,
                                                             const float* __restrict__ owned = nullptr) {
  // owned != NULL (domain decomposition, DdLink): halo copies carry weight 0 - their owners' ranks count them
  __shared__ float sh[4];
  const int m = blockIdx.y, sl = blockIdx.x;
  int lo, hi;
  slice_bounds(mol_start, m, sl, S, lo, hi);
  float sf = 0.f, sq = 0.f;
  for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    if (owned && owned[i] == 0.0f) continue;
    const float ft = y[(size_t)i * ldy + fcol];
    sf += ft * ft;
    sq += (q_prev ? q_prev[i] : 0.0f) + y[(size_t)i * ldy + qcol];
  }
  sf = block_sum(sf, sh);
  sq = block_sum(sq, sh);
  if (threadIdx.x == 0) {
    part[((size_t)m * S + sl) * 2 + 0] = sf;
    part[((size_t)m * S + sl) * 2 + 1] = sq;
  }
}

__global__ __launch_bounds__(256) void nse_fwd_apply_kernel(const float* __restrict__ y, int ldy, int qcol, int fcol,
                                                           const float* __restrict__ q_prev,
                                                           const int* __restrict__ mol_start,
                                                           const float* __restrict__ charge, int S,
                                                           const float* __restrict__ part, float* __restrict__ q_new,
                                                           float* __restrict__ Fm, float* __restrict__ Dm, int n_mol,
                                                           UpdateA upd) {
  if ((int)blockIdx.x >= S * n_mol) {  // update_a blocks ride on the same launch (see nse_fwd_kernel)
    update_a_block(upd, blockIdx.x - S * n_mol);
    return;
  }
  __shared__ float shFD[2];
  const int m = blockIdx.x / S, sl = blockIdx.x % S;
  if (threadIdx.x == 0) {
    float sf = 0.f, sq = 0.f;
    for (int k = 0; k < S; ++k) {
      sf += part[((size_t)m * S + k) * 2 + 0];
      sq += part[((size_t)m * S + k) * 2 + 1];
    }
    shFD[0] = sf + 1.0e-6f;
    shFD[1] = charge[m] - sq;
    if (sl == 0) {
      Fm[m] = shFD[0];
      Dm[m] = shFD[1];
    }
  }
  __syncthreads();
  const float F = shFD[0], D = shFD[1];
  int lo, hi;
  slice_bounds(mol_start, m, sl, S, lo, hi);
  for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const float ft = y[(size_t)i * ldy + fcol];
    const float qr = (q_prev ? q_prev[i] : 0.0f) + y[(size_t)i * ldy + qcol];
    q_new[i] = qr + (ft * ft) / F * D;
  }
}

int launch_nse_fwd(hipStream_t s, const float* y, int ldy, int nq, const float* q_prev, const int* mol_start,
                   const float* charge, int n_mol, int n_atoms, int S, float* part, float* q_new, float* Fm, float* Dm,
                   const float* upd_a, const int* upd_row_of, float* upd_a_new, float* upd_a_t, const DdLink* dd) {
  // upd_a_new != NULL: the feature update a_new = a + delta_a (launch_update_a) rides on the launch of channel 0
  const UpdateA none{nullptr, nullptr, nullptr, 0, 0, 0, nullptr, nullptr};
  const UpdateA upd{upd_a, upd_row_of, y, ldy, 2 * nq, n_atoms, upd_a_new, upd_a_t};
  const int n_upd = upd_a_new ? (int)(((size_t)n_atoms * 256 + 255) / 256) : 0;
  for (int ch = 0; ch < nq; ++ch) {  // channels are independent (ops.nse works on the trailing channel axis)
    const float* qp = q_prev ? q_prev + (size_t)ch * n_atoms : nullptr;
    const float* Q = charge + (size_t)ch * n_mol;
    float* qn = q_new + (size_t)ch * n_atoms;
    float *F = Fm + (size_t)ch * n_mol, *D = Dm + (size_t)ch * n_mol;
    const int extra = ch == 0 ? n_upd : 0;
    if (dd) {
      // domain decomposition: one slice per molecule summed over the OWNED atoms, all-reduced over the ranks by the caller's
      // exchange function, then applied to every local atom (halo copies included: the convolutions of the next pass read them)
      hipLaunchKernelGGL(nse_fwd_partial_kernel, dim3(1, n_mol), dim3(256), 0, s, y, ldy, ch, nq + ch, qp, mol_start, 1, part, dd->owned);
      AIMNET_LAUNCH_CHECK();
      if (dd->fn(dd->ctx, 0 /* AIMNET_DD_SUM */, part, 2 * (int64_t)n_mol, (void*)s) != 0) {
        set_last_error("eval: the domain-decomposition exchange function failed (NSE sums)");
        return -1;
      }
      hipLaunchKernelGGL(nse_fwd_apply_kernel, dim3(n_mol + extra), dim3(256), 0, s, y, ldy, ch, nq + ch, qp, mol_start, Q, 1, part,
                         qn, F, D, n_mol, ch == 0 ? upd : none);
      AIMNET_LAUNCH_CHECK();
      continue;
    }
    if (S <= 1) {
      hipLaunchKernelGGL(nse_fwd_kernel, dim3(n_mol + extra), dim3(256), 0, s, y, ldy, ch, nq + ch, qp, mol_start, Q, qn, F, D, n_mol,
                         ch == 0 ? upd : none);
      AIMNET_LAUNCH_CHECK();
      continue;
    }
    hipLaunchKernelGGL(nse_fwd_partial_kernel, dim3(S, n_mol), dim3(256), 0, s, y, ldy, ch, nq + ch, qp, mol_start, S, part);
    AIMNET_LAUNCH_CHECK();
    hipLaunchKernelGGL(nse_fwd_apply_kernel, dim3(S * n_mol + extra), dim3(256), 0, s, y, ldy, ch, nq + ch, qp, mol_start, Q, S, part,
                       qn, F, D, n_mol, ch == 0 ? upd : none);
    AIMNET_LAUNCH_CHECK();
  }
  return 0;
}

// NSE models: total and spin charges from the two channel planes (aimnet2.py:102-106)
__global__ void charge_sum_kernel(const float* __restrict__ q2, int n_atoms, float* __restrict__ q_tot,
                                  float* __restrict__ q_spin) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_atoms) return;
  const float a = q2[i], b = q2[(size_t)n_atoms + i];
  q_tot[i] = a + b;
  if (q_spin) q_spin[i] = a - b;
}

int launch_charge_sum(hipStream_t s, const float* q2, int n_atoms, float* q_tot, float* q_spin) {
  hipLaunchKernelGGL(charge_sum_kernel, dim3(ceil_div(n_atoms, 256)), dim3(256), 0, s, q2, n_atoms, q_tot, q_spin);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

// row_of (may be NULL): feature row of atom i inside `a` - pass 0 adds delta_a to the embedding row afv[Z_i] directly, so
// the initial features a^0 = afv[Z] (aimnet2.py:145-148) are never materialised
// a_t (may be NULL): second copy of the new feature row in the operand layout of the MFMA conv kernels (conv_mfma.hip):
// the transpose [g][a], feature (a, g) at float g * 16 + a
__global__ void update_a_kernel(UpdateA u) { update_a_block(u, blockIdx.x); }

int launch_update_a(hipStream_t s, const float* a, const int* row_of, const float* y, int ldy, int nq, int n_atoms, float* a_new,
                    float* a_t) {
  const size_t n = (size_t)n_atoms * 256;
  const UpdateA u{a, row_of, y, ldy, 2 * nq, n_atoms, a_new, a_t};
  hipLaunchKernelGGL(update_a_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, u);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

// ---- energy head last layer (k -> 1) and its adjoint seed ---------------------------------------
// d / zbar (may be NULL): the adjoint seed of the backward sweep, zbar = w * GELU'(z) of the layer below (ldh wide, zero
// padding beyond k), is written by the same wave - dE/de_atom = 1, so it does not wait for anything
__device__ __forceinline__ void head_last_block(const float* __restrict__ h, int ldh, const float* __restrict__ w,
                                                const float* __restrict__ b, int k, int n_atoms, float* __restrict__ e_atom,
                                                const float* __restrict__ d, float* __restrict__ zbar, int block) {
  const int i = block * 4 + (threadIdx.x >> 6);
  if (i >= n_atoms) return;
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  for (int c = lane; c < k; c += 64) acc += h[(size_t)i * ldh + c] * w[c];
  acc = wave_sum(acc);
  if (lane == 0) e_atom[i] = acc + b[0];
  if (zbar)
    for (int c = lane; c < ldh; c += 64) zbar[(size_t)i * ldh + c] = (c < k) ? w[c] * d[(size_t)i * ldh + c] : 0.0f;
}

__global__ void head_last_kernel(const float* __restrict__ h, int ldh, const float* __restrict__ w,
                                 const float* __restrict__ b, int k, int n_atoms, float* __restrict__ e_atom,
                                 const float* __restrict__ d, float* __restrict__ zbar) {
  head_last_block(h, ldh, w, b, k, n_atoms, e_atom, d, zbar, blockIdx.x);
}

// bin-ordered (x, y, z, q) stream of the list-free DSF walk (see charge_stream_kernel below), one 256-thread block of it
__device__ __forceinline__ void charge_stream_block(const float4* __restrict__ xs, const float* __restrict__ q, int n_atoms,
                                                    float4* __restrict__ xq, float* __restrict__ charges_out, int block) {
  const int k = block * 256 + threadIdx.x;
  if (k >= n_atoms) return;
  const float4 c = xs[k];
  const int id = __float_as_int(c.w);
  const float qv = q[id];
  xq[k] = make_float4(c.x, c.y, c.z, qv);
  if (charges_out) charges_out[id] = qv;
}

int launch_head_last(hipStream_t s, const float* h, int ldh, const float* w, const float* b, int k, int n_atoms,
                     float* e_atom, const float* d, float* zbar) {
  hipLaunchKernelGGL(head_last_kernel, dim3(ceil_div(n_atoms, 4)), dim3(256), 0, s, h, ldh, w, b, k, n_atoms, e_atom, d, zbar);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

// partial sum number b = (molecule, slice) of E_m = sum_i e_i + SAE[Z_i] + pair energies (fp64)
__device__ __forceinline__ void energy_partial_block(const float* __restrict__ e_atom, const double* __restrict__ ecoul,
                                                     const int* __restrict__ numbers, const double* __restrict__ sae,
                                                     const int* __restrict__ mol_start, int S, double* __restrict__ part, int b,
                                                     int* __restrict__ nf = nullptr) {
  __shared__ double sh[4];
  const int m = b / S, sl = b % S;
  int lo, hi;
  slice_bounds(mol_start, m, sl, S, lo, hi);
  double acc = 0.0;
  for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const int z = min(63, max(0, numbers[i]));
    acc += (double)e_atom[i] + sae[z] + ecoul[i];
  }
  const double r = block_sum(acc, sh);
  if (threadIdx.x == 0) {
    part[(size_t)m * S + sl] = r;
    if (nf && !isfinite(r)) atomicOr(nf, STATUS_NONFINITE);
  }
}

// the whole molecule in ONE block (S == 1 on a few 10^4 atoms, as a rider beside a longer launch): eight atoms per thread and
// round, their loads issued together - the slice form's two dependent loads per atom and round would take 40 round trips
__device__ __forceinline__ void energy_whole_block(const float* __restrict__ e_atom, const double* __restrict__ ecoul,
                                                   const int* __restrict__ numbers, const double* __restrict__ sae,
                                                   const int* __restrict__ mol_start, double* __restrict__ energy, int m,
                                                   int* __restrict__ nf = nullptr) {
  __shared__ double sh[4];
  const int lo = mol_start[m], hi = mol_start[m + 1];
  double acc = 0.0;
  for (int i0 = lo + (int)threadIdx.x; i0 < hi; i0 += 8 * (int)blockDim.x) {
    int z[8];
    float ea[8];
    double ec[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = i0 + k * (int)blockDim.x, ii = i < hi ? i : lo;
      z[k] = min(63, max(0, numbers[ii]));
      ea[k] = e_atom[ii];
      ec[k] = ecoul[ii];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (i0 + k * (int)blockDim.x < hi) acc += (double)ea[k] + sae[z[k]] + ec[k];
  }
  const double r = block_sum(acc, sh);
  if (threadIdx.x == 0) {
    energy[m] = r;
    if (nf && !isfinite(r)) atomicOr(nf, STATUS_NONFINITE);
  }
}

__global__ __launch_bounds__(256) void energy_partial_kernel(const float* __restrict__ e_atom,
                                                            const double* __restrict__ ecoul,
                                                            const int* __restrict__ numbers,
                                                            const double* __restrict__ sae,
                                                            const int* __restrict__ mol_start, int S,
                                                            double* __restrict__ part, int n_red, const float* __restrict__ cp_src,
                                                            float* __restrict__ cp_dst, int cp_n, int n_cp, PairMapRider pm,
                                                            int* __restrict__ nf) {
  if ((int)blockIdx.x >= n_red) {  // riders (independent work; one kernel boundary less each)
    const int b = blockIdx.x - n_red;
    if (b < n_cp) {  // the copy of the charges into the output
      const int e = b * 256 + threadIdx.x;
      if (e < cp_n) cp_dst[e] = cp_src[e];
    } else {  // the lookup pass of the reverse-pair map
      pair_rev_hash_block(pm.nb_idx, pm.nb_shift, pm.nb_cnt, pm.cap, pm.n_atoms, pm.tab, pm.rev, b - n_cp);
    }
    return;
  }
  energy_partial_block(e_atom, ecoul, numbers, sae, mol_start, S, part, blockIdx.x, nf);
}

__global__ void energy_finish_kernel(const double* __restrict__ part, int S, int n_mol, double* __restrict__ energy) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n_mol) return;
  double r = 0.0;
  for (int k = 0; k < S; ++k) r += part[(size_t)m * S + k];
  energy[m] = r;
}

int launch_energy_reduce(hipStream_t s, const float* e_atom, const double* ecoul, const int* numbers,
                         const double* sae, const int* mol_start, int n_mol, int S, double* part, double* energy,
                         const float* copy_src, float* copy_dst, int copy_n, const PairMapRider* rev_rider, int* nf) {
  S = S < 1 ? 1 : S;
  const int n_red = S * n_mol, n_cp = copy_dst ? ceil_div(copy_n, 256) : 0;
  PairMapRider pm{};
  if (rev_rider) pm = *rev_rider;
  hipLaunchKernelGGL(energy_partial_kernel, dim3(n_red + n_cp + pm.n_blocks), dim3(256), 0, s, e_atom, ecoul, numbers, sae, mol_start, S,
                     S == 1 ? energy : part, n_red, copy_src, copy_dst, copy_n, n_cp, pm, nf);
  AIMNET_LAUNCH_CHECK();
  if (S > 1) {
    hipLaunchKernelGGL(energy_finish_kernel, dim3(ceil_div(n_mol, 64)), dim3(64), 0, s, part, S, n_mol, energy);
    AIMNET_LAUNCH_CHECK();
  }
  return 0;
}

// ---- Coulomb pair kernels: one wave per centre atom, lanes over neighbours ------------------------
// Common tail: E_i = sign k sum_m w q_i q_j (fp64 sum of fp32 pair terms), and for the full symmetric
// list  dE/dq_i = sign 2k sum_m w q_j ,  dE/dx_i = -sign 2k sum_m w' q_i q_j u_im ,
// dE/deps_ab += sign k sum_m w' q_i q_j r_a u_b  (ordered pairs, each once).
struct PairAcc {
  double e = 0.0;
  float qb = 0.f, f0 = 0.f, f1 = 0.f, f2 = 0.f;
  float W[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
};

template <bool GRAD, bool STRESS>
__device__ __forceinline__ void pair_add(PairAcc& A, float w, float dw, float qi, float qj, float ux, float uy, float uz,
                                         float d) {
  const float qq = qi * qj;
  A.e += (double)(w * qq);
  if (GRAD) {
    A.qb += w * qj;
    const float t = dw * qq;
    A.f0 += t * ux;
    A.f1 += t * uy;
    A.f2 += t * uz;
    if (STRESS) {
      const float rx = ux * d, ry = uy * d, rz = uz * d;
      A.W[0] += rx * t * ux; A.W[1] += rx * t * uy; A.W[2] += rx * t * uz;
      A.W[3] += ry * t * ux; A.W[4] += ry * t * uy; A.W[5] += ry * t * uz;
      A.W[6] += rz * t * ux; A.W[7] += rz * t * uy; A.W[8] += rz * t * uz;
    }
  }
}

template <bool GRAD, bool STRESS, bool ACCUM>
__device__ __forceinline__ void pair_store(PairAcc& A, int i, int lane, float sign_k, double e_extra, float qb_extra,
                                           double* ecoul, float* qbar, float* fgrad, float* virial_atom) {
  const double e = wave_sum(A.e);
  float qb = 0.f, f0 = 0.f, f1 = 0.f, f2 = 0.f;
  if (GRAD) {
    qb = wave_sum(A.qb);
    f0 = wave_sum(A.f0);
    f1 = wave_sum(A.f1);
    f2 = wave_sum(A.f2);
    if (STRESS) {
#pragma unroll
      for (int k = 0; k < 9; ++k) A.W[k] = wave_sum(A.W[k]);
    }
  }
  if (lane == 0) {
    const double ev = (double)sign_k * e + e_extra;
    ecoul[i] = ACCUM ? ecoul[i] + ev : ev;
    if (GRAD) {
      const float qv = 2.0f * sign_k * qb + qb_extra;
      qbar[i] = ACCUM ? qbar[i] + qv : qv;
      const float g0 = -2.0f * sign_k * f0, g1 = -2.0f * sign_k * f1, g2 = -2.0f * sign_k * f2;
      fgrad[3 * i + 0] = ACCUM ? fgrad[3 * i + 0] + g0 : g0;
      fgrad[3 * i + 1] = ACCUM ? fgrad[3 * i + 1] + g1 : g1;
      fgrad[3 * i + 2] = ACCUM ? fgrad[3 * i + 2] + g2 : g2;
    }
  }
  if (GRAD && STRESS && lane < 9) {
    float v = A.W[0];
#pragma unroll
    for (int k = 1; k < 9; ++k) v = (lane == k) ? A.W[k] : v;
    v *= sign_k;
    virial_atom[(size_t)i * 9 + lane] = ACCUM ? virial_atom[(size_t)i * 9 + lane] + v : v;
  }
}

// embedded short-range Coulomb, SUBTRACTED (sign -1); also initialises the adjoint buffers
template <bool GRAD, bool STRESS>
__global__ __launch_bounds__(256) void coulomb_sr_kernel(bool enabled, const float* __restrict__ q,
                                                        const int* __restrict__ nb_idx, const int* __restrict__ nb_cnt,
                                                        const float4* __restrict__ pg, int cap, CoulombParams cp,
                                                        int n_atoms, double* __restrict__ ecoul, float* __restrict__ qbar,
                                                        float* __restrict__ fgrad, float* __restrict__ virial_atom, SrRiders rd) {
  // independent work that rides on this launch (a kernel boundary costs 4-5 us on the device): the last energy-head layer with
  // its backward seed, and the charge stream of the list-free DSF walk
  const int n_sr = (n_atoms + 3) >> 2;
  // the status riders come FIRST in the grid: the one-block form reads 10^4 row counts and would otherwise start last and end last
  if ((int)blockIdx.x < rd.n_status_blocks) {
    if (rd.status_all)  // status words of the evaluation, stored by this one block (nothing was zeroed)
      nlist_status_owned_block(rd.cnt_true, n_atoms, rd.status_cap, rd.bad_part, rd.status_all, rd.keep7);
    else  // status words of the short-range list (max row length, overflow flag) into the zeroed array
      nlist_status_block(rd.cnt_true, n_atoms, rd.status_cap, rd.status_max, rd.status_ovf, blockIdx.x);
    return;
  }
  const int bx = (int)blockIdx.x - rd.n_status_blocks;
  if (bx >= n_sr) {
    const int b = bx - n_sr;
    if (b < rd.n_head_blocks) {
      head_last_block(rd.h, rd.ldh, rd.w, rd.b, rd.k, n_atoms, rd.e_atom, rd.d, rd.zbar, b);
    } else if (b < rd.n_head_blocks + rd.n_stream_blocks) {
      charge_stream_block(rd.xs, q, n_atoms, rd.xq, rd.charges_out, b - rd.n_head_blocks);
    } else {  // hash build of the reverse-pair map (pairmap.h)
      __shared__ unsigned long long s_tab[4][RH_SLOTS];
      pair_hash_block(rd.hash.nb_idx, rd.hash.nb_shift, rd.hash.nb_cnt, rd.hash.cap, n_atoms, rd.hash.tab, rd.hash.rev,
                      b - rd.n_head_blocks - rd.n_stream_blocks, s_tab);
    }
    return;
  }
  const int i = bx * 4 + (threadIdx.x >> 6);
  if (i >= n_atoms) return;
  const int lane = threadIdx.x & 63;
  PairAcc A;
  if (enabled) {
    const int cnt = nb_cnt[i];
    const float qi = q[i];
    for (int m = lane; m < cnt; m += 64) {
      const size_t p = (size_t)i * cap + m;
      const float4 u = pg[p];
      const float d = u.w;
      float fc, dfc;
      if (cp.sr_envelope == 0) {
        const float tr = d / cp.sr_rc;
        const float t = fminf(fmaxf(tr, 0.0f), 1.0f - 1e-6f);
        const float om = 1.0f - t * t;
        fc = expf(-1.0f / om) / 0.36787944117144233f;
        dfc = (tr < 1.0f - 1e-6f) ? fc * (-2.0f * t / (om * om)) / cp.sr_rc : 0.0f;
      } else {
        const float dc = fminf(fmaxf(d, 1e-6f), cp.sr_rc);
        const float w = 3.14159265358979323846f / cp.sr_rc;
        fc = 0.5f * (cosf(dc * w) + 1.0f);
        dfc = (d > 1e-6f && d < cp.sr_rc) ? -0.5f * w * sinf(dc * w) : 0.0f;
      }
      const float inv = 1.0f / d;
      pair_add<GRAD, STRESS>(A, fc * inv, dfc * inv - fc * inv * inv, qi, q[nb_idx[p]], u.x, u.y, u.z, d);
    }
  }
  pair_store<GRAD, STRESS, false>(A, i, lane, -cp.factor, 0.0, 0.0f, ecoul, qbar, fgrad, virial_atom);
  if (rd.simple_xw) {  // LRCoulomb "simple" (all pairs of the molecule, coulomb_simple_kernel below) in the same wave: it accumulates
    // onto what this wave has just stored - the same sums in the same order as the separate launch, one kernel boundary less
    const float* __restrict__ xw = rd.simple_xw;
    const int mi = rd.simple_mol_idx[i];
    const int j0 = rd.simple_mol_start[mi], j1 = rd.simple_mol_start[mi + 1];
    const float xi = xw[3 * i], yi = xw[3 * i + 1], zi = xw[3 * i + 2], qi = q[i];
    PairAcc B;
    for (int j = j0 + lane; j < j1; j += 64) {
      if (j == i) continue;
      const float rx = xw[3 * j] - xi, ry = xw[3 * j + 1] - yi, rz = xw[3 * j + 2] - zi;
      const float d = sqrtf(rx * rx + ry * ry + rz * rz);
      const float inv = 1.0f / d;
      pair_add<GRAD, false>(B, inv, -inv * inv, qi, q[j], rx * inv, ry * inv, rz * inv, d);
    }
    pair_store<GRAD, false, true>(B, i, lane, cp.factor, 0.0, 0.0f, ecoul, qbar, fgrad, nullptr);
  }
}

int launch_coulomb_sr(hipStream_t s, bool grad, bool stress, bool enabled, const float* q, const int* nb_idx,
                      const int* nb_cnt, const float4* pg, int cap, CoulombParams cp, int n_atoms, double* ecoul,
                      float* qbar, float* fgrad, float* virial_atom, const SrRiders* riders) {
  SrRiders rd{};
  if (riders) rd = *riders;
  dim3 grid(ceil_div(n_atoms, 4) + rd.n_head_blocks + rd.n_stream_blocks + rd.hash.n_blocks + rd.n_status_blocks), block(256);
  if (grad && stress)
    hipLaunchKernelGGL((coulomb_sr_kernel<true, true>), grid, block, 0, s, enabled, q, nb_idx, nb_cnt, pg, cap, cp, n_atoms,
                       ecoul, qbar, fgrad, virial_atom, rd);
  else if (grad)
    hipLaunchKernelGGL((coulomb_sr_kernel<true, false>), grid, block, 0, s, enabled, q, nb_idx, nb_cnt, pg, cap, cp, n_atoms,
                       ecoul, qbar, fgrad, virial_atom, rd);
  else
    hipLaunchKernelGGL((coulomb_sr_kernel<false, false>), grid, block, 0, s, enabled, q, nb_idx, nb_cnt, pg, cap, cp,
                       n_atoms, ecoul, qbar, fgrad, virial_atom, rd);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

// "simple": every other atom of the same molecule, w = 1/d
template <bool GRAD>
__global__ __launch_bounds__(256) void coulomb_simple_kernel(const float* __restrict__ q, const float* __restrict__ xw,
                                                            const int* __restrict__ mol_idx,
                                                            const int* __restrict__ mol_start, CoulombParams cp,
                                                            int n_atoms, double* __restrict__ ecoul,
                                                            float* __restrict__ qbar, float* __restrict__ fgrad) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_atoms) return;
  const int lane = threadIdx.x & 63;
  const int m = mol_idx[i];
  const int j0 = mol_start[m], j1 = mol_start[m + 1];
  const float xi = xw[3 * i], yi = xw[3 * i + 1], zi = xw[3 * i + 2], qi = q[i];
  PairAcc A;
  for (int j = j0 + lane; j < j1; j += 64) {
    if (j == i) continue;
    const float rx = xw[3 * j] - xi, ry = xw[3 * j + 1] - yi, rz = xw[3 * j + 2] - zi;
    const float d = sqrtf(rx * rx + ry * ry + rz * rz);
    const float inv = 1.0f / d;
    pair_add<GRAD, false>(A, inv, -inv * inv, qi, q[j], rx * inv, ry * inv, rz * inv, d);
  }
  pair_store<GRAD, false, true>(A, i, lane, cp.factor, 0.0, 0.0f, ecoul, qbar, fgrad, nullptr);
}

int launch_coulomb_simple(hipStream_t s, bool grad, const float* q, const float* xw, const int* mol_idx,
                          const int* mol_start, CoulombParams cp, int n_atoms, double* ecoul, float* qbar,
                          float* fgrad) {
  dim3 grid(ceil_div(n_atoms, 4)), block(256);
  if (grad)
    hipLaunchKernelGGL(coulomb_simple_kernel<true>, grid, block, 0, s, q, xw, mol_idx, mol_start, cp, n_atoms, ecoul, qbar,
                       fgrad);
  else
    hipLaunchKernelGGL(coulomb_simple_kernel<false>, grid, block, 0, s, q, xw, mol_idx, mol_start, cp, n_atoms, ecoul, qbar,
                       fgrad);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

// DSF over the long-range neighbour matrix, plus the self term -2k (erfc(a Rc)/(2 Rc) + a/sqrt(pi)) q_i^2
// SIMPLE: w = 1 / d over every entry of the matrix, no cutoff, no self term - LRCoulomb.coul_simple over a CALLER-SUPPLIED
// `nbmat_lr` (lr.py:311-331 sums over whatever list it is given; the engine's own lists use coulomb_simple_kernel above)
template <bool GRAD, bool STRESS, bool SIMPLE = false>
__global__ __launch_bounds__(256) void coulomb_dsf_kernel(const float* __restrict__ q, const float* __restrict__ xw,
                                                         const int* __restrict__ mol_idx, const float* __restrict__ cell,
                                                         int n_cell, const int* __restrict__ nb_idx,
                                                         const int* __restrict__ nb_shift, const int* __restrict__ nb_cnt,
                                                         int cap, CoulombParams cp, int n_atoms,
                                                         double* __restrict__ ecoul, float* __restrict__ qbar,
                                                         float* __restrict__ fgrad, float* __restrict__ virial_atom) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_atoms) return;
  const int lane = threadIdx.x & 63;
  const float al = cp.dsf_alpha, Rc = cp.dsf_rc;
  const float two_a_sqrtpi = 2.0f * al * 0.56418958354775629f;
  const float erfc_rc = erfcf(al * Rc);
  const float sv = erfc_rc / Rc;
  const float slope = erfc_rc / (Rc * Rc) + two_a_sqrtpi * expf(-al * al * Rc * Rc) / Rc;
  const float* c = cell ? cell + (n_cell == 1 ? 0 : (size_t)mol_idx[i] * 9) : nullptr;
  const float xi = xw[3 * i], yi = xw[3 * i + 1], zi = xw[3 * i + 2], qi = q[i];
  const int cnt = nb_cnt[i];
  PairAcc A;
  for (int m = lane; m < cnt; m += 64) {
    const size_t p = (size_t)i * cap + m;
    const int j = nb_idx[p];
    float rx = xw[3 * j] - xi, ry = xw[3 * j + 1] - yi, rz = xw[3 * j + 2] - zi;
    if (c) {
      int sx, sy, sz;
      unpack_shift(nb_shift[p], sx, sy, sz);
      rx += sx * c[0] + sy * c[3] + sz * c[6];
      ry += sx * c[1] + sy * c[4] + sz * c[7];
      rz += sx * c[2] + sy * c[5] + sz * c[8];
    }
    const float d = sqrtf(rx * rx + ry * ry + rz * rz);
    if (!SIMPLE && !(d < Rc)) continue;
    const float inv = 1.0f / d;
    float w, dw;
    if (SIMPLE) {
      w = inv;
      dw = -inv * inv;
    } else {
      // erfc(x) = exp(-x^2) t P(t), t = 1 / (1 + x / 2): the degree-9 fit of the list-free walk below (coulomb_dsf_walk_kernel) - one
      // exponential serves the value and the derivative, and matrix and walk evaluate the same pair term
      const float ex = __builtin_amdgcn_exp2f(-1.4426950408889634f * al * al * d * d);
      const float tt = __builtin_amdgcn_rcpf(fmaf(0.5f * al, d, 1.0f));
      float pe = 2.672036890e-02f;
      pe = fmaf(pe, tt, -2.020067459e-01f);
      pe = fmaf(pe, tt, 6.150174393e-01f);
      pe = fmaf(pe, tt, -9.195323909e-01f);
      pe = fmaf(pe, tt, 6.300562657e-01f);
      pe = fmaf(pe, tt, -2.111610618e-01f);
      pe = fmaf(pe, tt, 2.648643249e-01f);
      pe = fmaf(pe, tt, 2.301390953e-01f);
      pe = fmaf(pe, tt, 2.838921720e-01f);
      pe = fmaf(pe, tt, 2.820105286e-01f);
      const float ec = pe * tt * ex;
      w = ec * inv - sv + (d - Rc) * slope;
      dw = -ec * inv * inv - two_a_sqrtpi * ex * inv + slope;
    }
    pair_add<GRAD, STRESS>(A, w, dw, qi, q[j], rx * inv, ry * inv, rz * inv, d);
  }
  const float cs = SIMPLE ? 0.0f : -(sv * 0.5f + al * 0.56418958354775629f);
  const double e_self = 2.0 * (double)cp.factor * (double)(cs * qi * qi);
  const float qb_self = 4.0f * cp.factor * cs * qi;
  pair_store<GRAD, STRESS, true>(A, i, lane, cp.factor, e_self, qb_self, ecoul, qbar, fgrad, virial_atom);
}

int launch_coulomb_dsf(hipStream_t s, bool grad, bool stress, const float* q, const float* xw, const int* mol_idx,
                       const float* cell, int n_cell, const int* nb_idx, const int* nb_shift, const int* nb_cnt,
                       int cap, CoulombParams cp, int n_atoms, double* ecoul, float* qbar, float* fgrad,
                       float* virial_atom, bool simple) {
  dim3 grid(ceil_div(n_atoms, 4)), block(256);
  if (simple) {  // 1 / d over a caller-supplied matrix (never periodic: no virial)
    if (grad)
      hipLaunchKernelGGL((coulomb_dsf_kernel<true, false, true>), grid, block, 0, s, q, xw, mol_idx, cell, n_cell, nb_idx, nb_shift,
                         nb_cnt, cap, cp, n_atoms, ecoul, qbar, fgrad, virial_atom);
    else
      hipLaunchKernelGGL((coulomb_dsf_kernel<false, false, true>), grid, block, 0, s, q, xw, mol_idx, cell, n_cell, nb_idx, nb_shift,
                         nb_cnt, cap, cp, n_atoms, ecoul, qbar, fgrad, virial_atom);
    AIMNET_LAUNCH_CHECK();
    return 0;
  }
  if (grad && stress)
    hipLaunchKernelGGL((coulomb_dsf_kernel<true, true>), grid, block, 0, s, q, xw, mol_idx, cell, n_cell, nb_idx, nb_shift,
                       nb_cnt, cap, cp, n_atoms, ecoul, qbar, fgrad, virial_atom);
  else if (grad)
    hipLaunchKernelGGL((coulomb_dsf_kernel<true, false>), grid, block, 0, s, q, xw, mol_idx, cell, n_cell, nb_idx, nb_shift,
                       nb_cnt, cap, cp, n_atoms, ecoul, qbar, fgrad, virial_atom);
  else
    hipLaunchKernelGGL((coulomb_dsf_kernel<false, false>), grid, block, 0, s, q, xw, mol_idx, cell, n_cell, nb_idx, nb_shift,
                       nb_cnt, cap, cp, n_atoms, ecoul, qbar, fgrad, virial_atom);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

// periodic DSF without a neighbour matrix: the wave walks the bins of the short-range cell grid out to
// Rc and accumulates the pair terms directly (no 8 B x ~1800 per atom list write + re-read, no row
// capacity / overflow handling for the long-range cutoff)
template <bool GRAD, bool STRESS>
__global__ __launch_bounds__(256) void coulomb_dsf_walk_kernel(const float* __restrict__ q, const float* __restrict__ xw,
                                                              const int* __restrict__ mol_idx,
                                                              const NlistSystem* __restrict__ sys,
                                                              const int* __restrict__ bin_start,
                                                              const float4* __restrict__ xq, CoulombParams cp, int n_atoms,
                                                              double* __restrict__ ecoul, float* __restrict__ qbar,
                                                              float* __restrict__ fgrad, float* __restrict__ virial_atom,
                                                              PairMapRider pm) {
  if ((int)blockIdx.x >= ((n_atoms + 3) >> 2)) {  // rider: the lookup pass of the reverse-pair map (independent, latency-bound work
    // beside the VALU-bound walk; a kernel boundary less)
    pair_rev_hash_block(pm.nb_idx, pm.nb_shift, pm.nb_cnt, pm.cap, n_atoms, pm.tab, pm.rev, blockIdx.x - ((n_atoms + 3) >> 2));
    return;
  }
  // Only ~1/3 of the candidates the bin walk visits lie inside Rc, while a pair term costs ~150 VALU
  // instructions (erfc, exp, fp64 energy sum, virial).  So the walk only COMPACTS the hits (r, q_j) into a
  // per-wave LDS queue, and the expensive math runs on full 64-lane batches popped from that queue.
  __shared__ float4 queue[4][128];
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_atoms) return;
  const int lane = threadIdx.x & 63;
  float4* Q = queue[threadIdx.x >> 6];
  // Ewald (cp.ewald): the real-space term erfc(alpha d) / d inside the system's own cutoff - the same pair term without the DSF
  // shift and force-shift, the same self term -alpha / sqrt(pi) q_i^2 (ewald.hip adds reciprocal space and the background)
  const bool ewald = cp.ewald != nullptr;
  const float al = ewald ? cp.ewald[mol_idx[i]].alpha : cp.dsf_alpha, Rc = ewald ? cp.ewald[mol_idx[i]].rc : cp.dsf_rc;
  const float two_a_sqrtpi = 2.0f * al * 0.56418958354775629f;
  const float erfc_rc = erfcf(al * Rc);
  const float sv = ewald ? 0.0f : erfc_rc / Rc;
  const float slope = ewald ? 0.0f : erfc_rc / (Rc * Rc) + two_a_sqrtpi * expf(-al * al * Rc * Rc) / Rc;
  const float xi = xw[3 * i], yi = xw[3 * i + 1], zi = xw[3 * i + 2], qi = q[i];
  PairAcc A;
  auto pair_term = [&](const float4& e) {
    // single-instruction rsq / exp2 (1 ulp): the kernel is VALU-issue bound and the pair sums are fp64 anyway
    const float d2 = e.x * e.x + e.y * e.y + e.z * e.z;
    const float inv = __builtin_amdgcn_rsqf(d2);
    const float d = d2 * inv;
    if (d < Rc) {  // the walk tests the squared distance; the reference tests d < Rc on the root (a pair at Rc weighs 0)
      // erfc(x) = exp(-x^2) t P(t), t = 1 / (1 + x/2): degree-9 fit of erfcx(x) / t on x in [0, 6.5] (6.5e-9 relative; the fp32
      // Horner evaluation adds 2.5e-7, the class of erfcf itself) - the exponential is the one the derivative needs anyway
      const float ex = __builtin_amdgcn_exp2f(-1.4426950408889634f * al * al * d2);
      const float tt = __builtin_amdgcn_rcpf(fmaf(0.5f * al, d, 1.0f));
      float pe = 2.672036890e-02f;
      pe = fmaf(pe, tt, -2.020067459e-01f);
      pe = fmaf(pe, tt, 6.150174393e-01f);
      pe = fmaf(pe, tt, -9.195323909e-01f);
      pe = fmaf(pe, tt, 6.300562657e-01f);
      pe = fmaf(pe, tt, -2.111610618e-01f);
      pe = fmaf(pe, tt, 2.648643249e-01f);
      pe = fmaf(pe, tt, 2.301390953e-01f);
      pe = fmaf(pe, tt, 2.838921720e-01f);
      pe = fmaf(pe, tt, 2.820105286e-01f);
      const float ec = pe * tt * ex;
      const float w = ec * inv - sv + (d - Rc) * slope;
      const float dw = -ec * inv * inv - two_a_sqrtpi * ex * inv + slope;
      // pair_add in terms of r instead of u = r / d: t u = (t / d) r and the virial term r (x) t u = (t / d) r (x) r is symmetric
      const float qq = qi * e.w;
      A.e += (double)(w * qq);
      if (GRAD) {
        A.qb += w * e.w;
        const float sc = dw * qq * inv;
        const float px = sc * e.x, py = sc * e.y, pz = sc * e.z;
        A.f0 += px; A.f1 += py; A.f2 += pz;
        if (STRESS) {
          A.W[0] += px * e.x; A.W[1] += px * e.y; A.W[2] += px * e.z;
          A.W[4] += py * e.y; A.W[5] += py * e.z; A.W[8] += pz * e.z;
        }
      }
    }
  };
  int qn = 0;  // wave-uniform fill level of the queue
  __shared__ int s_runs[4][CELLWALK_RUN_INTS];
  cell_walk<false>(sys[mol_idx[i]], i, xi, yi, zi, Rc, bin_start, xq, lane, s_runs[threadIdx.x >> 6],
                   [&](float qj, float rx, float ry, float rz, bool ok, int) {
                     const unsigned long long mask = __ballot(ok);
                     if (ok) Q[qn + __popcll(mask & ((1ull << lane) - 1ull))] = make_float4(rx, ry, rz, qj);
                     qn += __popcll(mask);
                     __atomic_signal_fence(__ATOMIC_SEQ_CST);  // LDS ops of one wave execute in order
                     if (qn >= 64) {
                       const float4 e = Q[lane];
                       const float4 tail = Q[64 + lane];
                       __atomic_signal_fence(__ATOMIC_SEQ_CST);
                       qn -= 64;
                       if (lane < qn) Q[lane] = tail;
                       __atomic_signal_fence(__ATOMIC_SEQ_CST);
                       pair_term(e);
                     }
                   });
  if (lane < qn) pair_term(Q[lane]);
  if (GRAD && STRESS) {  // the symmetric half was accumulated
    A.W[3] = A.W[1];
    A.W[6] = A.W[2];
    A.W[7] = A.W[5];
  }
  const float cs = -(sv * 0.5f + al * 0.56418958354775629f);
  const double e_self = 2.0 * (double)cp.factor * (double)(cs * qi * qi);
  const float qb_self = 4.0f * cp.factor * cs * qi;
  pair_store<GRAD, STRESS, true>(A, i, lane, cp.factor, e_self, qb_self, ecoul, qbar, fgrad, virial_atom);
}

// bin-ordered (x, y, z, q) stream: the walk then needs ONE coalesced 16-byte load per candidate and no
// dependent q[j] gather (the list-free DSF kernel was latency-bound on that second load)
// (charges_out, may be NULL: the `charges` output of the evaluation is written on the way - every atom appears once in the stream)
__global__ void charge_stream_kernel(const float4* __restrict__ xs, const float* __restrict__ q, int n_atoms,
                                     float4* __restrict__ xq, float* __restrict__ charges_out) {
  charge_stream_block(xs, q, n_atoms, xq, charges_out, blockIdx.x);
}

int launch_coulomb_dsf_walk(hipStream_t s, bool grad, bool stress, const float* q, const int* mol_idx, NlistBuffers& b,
                            CoulombParams cp, int n_atoms, double* ecoul, float* qbar, float* fgrad, float* virial_atom,
                            float* charges_out, bool stream_done, const PairMapRider* rev_rider) {
  PairMapRider pm{};
  if (rev_rider) pm = *rev_rider;
  dim3 grid(ceil_div(n_atoms, 4) + pm.n_blocks), block(256);
  const NlistSystem* sys = (const NlistSystem*)b.sys;
  float4* xq = (float4*)b.sorted_tmp_xq;
  if (!stream_done) {  // (normally the stream rides on the SR-Coulomb launch, SrRiders)
    hipLaunchKernelGGL(charge_stream_kernel, dim3(ceil_div(n_atoms, 256)), dim3(256), 0, s, b.xs, q, n_atoms, xq, charges_out);
    AIMNET_LAUNCH_CHECK();
  }
  if (grad && stress)
    hipLaunchKernelGGL((coulomb_dsf_walk_kernel<true, true>), grid, block, 0, s, q, b.xw, mol_idx, sys, b.bin_start, xq, cp,
                       n_atoms, ecoul, qbar, fgrad, virial_atom, pm);
  else if (grad)
    hipLaunchKernelGGL((coulomb_dsf_walk_kernel<true, false>), grid, block, 0, s, q, b.xw, mol_idx, sys, b.bin_start, xq, cp,
                       n_atoms, ecoul, qbar, fgrad, virial_atom, pm);
  else
    hipLaunchKernelGGL((coulomb_dsf_walk_kernel<false, false>), grid, block, 0, s, q, b.xw, mol_idx, sys, b.bin_start, xq, cp,
                       n_atoms, ecoul, qbar, fgrad, virial_atom, pm);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

// ---- NSE backward --------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nse_bwd_partial_kernel(const float* __restrict__ qbar, const float* __restrict__ y,
                                                             int ldy, int fcol, const int* __restrict__ mol_start, int S,
                                                             float* __restrict__ part) {
  __shared__ float sh[4];
  const int m = blockIdx.y, sl = blockIdx.x;
  int lo, hi;
  slice_bounds(mol_start, m, sl, S, lo, hi);
  float acc = 0.f;
  for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const float ft = y[(size_t)i * ldy + fcol];
    acc += qbar[i] * (ft * ft);
  }
  const float r = block_sum(acc, sh);
  if (threadIdx.x == 0) part[(size_t)m * S + sl] = r;
}

// partial sums of sum_i qbar_i f_i per (channel, molecule, slice) into part[(ch n_mol + m) S + sl]; build_zbar adds the S
// slices in slice order (what a separate finish kernel used to do) and divides by F_m
int launch_nse_bwd_reduce(hipStream_t s, const float* qbar, const float* y, int ldy, int nq, const int* mol_start, int n_mol,
                          int n_atoms, int S, float* part) {
  S = S < 1 ? 1 : S;
  for (int ch = 0; ch < nq; ++ch) {
    hipLaunchKernelGGL(nse_bwd_partial_kernel, dim3(S, n_mol), dim3(256), 0, s, qbar + (size_t)ch * n_atoms, y, ldy, nq + ch,
                       mol_start, S, part + (size_t)ch * n_mol * S);
    AIMNET_LAUNCH_CHECK();
  }
  return 0;
}

// zbar[i] = adjoint of the LAST linear layer's pre-activation of this pass' MLP:
//   ybar = [q~bar (nq), f~bar (nq), delta_a bar(n_feat)], times GELU'(z_last) when the MLP ends with GELU.
// qbar / qbar_next: nq planes of [n_atoms]; Fm, Dm: nq planes of [n_mol]; wpart: launch_nse_bwd_reduce's partial sums.
__global__ __launch_bounds__(256) void build_zbar_kernel(const float* __restrict__ qbar, const float* __restrict__ abar,
                                  const float* __restrict__ y, int ldy, const float* __restrict__ dlast,
                                  const float* __restrict__ Fm, const float* __restrict__ Dm,
                                  const float* __restrict__ wpart, int S, const int* __restrict__ mol_idx, int n_atoms,
                                  int n_mol, int n_feat, int nq, int carry_q, float* __restrict__ zbar,
                                  float* __restrict__ qbar_next, unsigned short* __restrict__ zbar3, int fmt,
                                  const int* __restrict__ mol_start, const float* __restrict__ owned) {
  // wpart == NULL (small systems, launch_build_zbar): the block forms sum_i qbar_i f_i of the molecules of its four atoms itself
  // (same code and order in every block that needs a molecule: the same bits) - what nse_bwd_partial_kernel and a kernel boundary
  // did.  Pays while a molecule is a few hundred atoms (a 113-atom molecule: -6 us per pass); for 10^4 atoms the redundant sums cost
  // more than the launch (profiles/r5_nse_merged_ab.txt).
  __shared__ float s_w[2][4];
  __shared__ float sh[4];
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (wpart == nullptr) {
    // one sum per ATOM SLOT of the block (not per molecule id: sorted mol_idx may skip ids - empty molecules - so the ids of four
    // consecutive atoms can span any range); a slot whose molecule is the previous slot's copies that slot's sums
    const int a_lo = blockIdx.x * 4, a_hi = min(n_atoms, a_lo + 4);
    for (int k = 0; k < a_hi - a_lo; ++k) {
      const int m = mol_idx[a_lo + k];
      if (k > 0 && m == mol_idx[a_lo + k - 1]) {
        if (threadIdx.x == 0)
          for (int ch = 0; ch < nq; ++ch) s_w[ch][k] = s_w[ch][k - 1];
        continue;
      }
      const int i0 = mol_start[m], i1 = mol_start[m + 1];
      for (int ch = 0; ch < nq; ++ch) {
        float acc = 0.f;
        for (int q = i0 + threadIdx.x; q < i1; q += 256) {
          const float ft = y[(size_t)q * ldy + nq + ch];
          acc += qbar[(size_t)ch * n_atoms + q] * (ft * ft);
        }
        const float r = block_sum(acc, sh);
        if (threadIdx.x == 0) s_w[ch][k] = r;
      }
    }
    __syncthreads();
  }
  if (i >= n_atoms) return;
  const int m = mol_idx[i];
  // every lane reads both channels' qbar before anything is written: qbar_next may alias qbar
  float qr[2] = {0.f, 0.f}, fsc[2] = {0.f, 0.f};
  for (int ch = 0; ch < nq; ++ch) {
    const float F = Fm[(size_t)ch * n_mol + m];
    float wsum;
    if (wpart) {
      // Wbar_m = (sum_i qbar_i f_i) / F_m: the S slice sums, one per lane, added by the wave's fixed reduction tree (S dependent
      // wave-uniform loads in a row were half of this kernel's time on a 10 k-atom system, S = 20)
      float wl = 0.f;
      for (int k = lane; k < S; k += 64) wl += wpart[((size_t)ch * n_mol + m) * S + k];
      wsum = wave_sum(wl);
    } else {
      wsum = s_w[ch][threadIdx.x >> 6];
    }
    // (domain decomposition: the molecule sums ran over owned atoms only, so only they receive the sums' adjoint; wsum is the
    // all-reduced sum over every rank's local atoms)
    qr[ch] = (owned && owned[i] == 0.0f) ? qbar[(size_t)ch * n_atoms + i] : qbar[(size_t)ch * n_atoms + i] - wsum / F;
    fsc[ch] = Dm[(size_t)ch * n_mol + m] / F;
  }
  float* zr = zbar + (size_t)i * ldy;
  const float* dr = dlast ? dlast + (size_t)i * ldy : nullptr;
  // split form for gemm_bf3a.hip (zbar3): the row is assembled in LDS and leaves as 8-byte plane pieces, four columns per lane
  __shared__ float stage[4][512];
  float* st = stage[threadIdx.x >> 6];
  const bool to_lds = zbar3 != nullptr && ldy <= 512;
  for (int c = lane; c < ldy; c += 64) {
    float v;
    if (c < nq) v = c == 0 ? qr[0] : qr[1];
    else if (c < 2 * nq) v = 2.0f * y[(size_t)i * ldy + c] * (c == nq ? fsc[0] : fsc[1]) * (c == nq ? qr[0] : qr[1]);
    else if (c < 2 * nq + n_feat) v = abar[(size_t)i * n_feat + c - 2 * nq];
    else v = 0.0f;
    if (dr && c < 2 * nq + n_feat) v *= dr[c];
    if (to_lds) st[c] = v;
    else if (zbar3) {
      if (fmt == 2) store_h2_1(zbar3 + (size_t)i * 2 * ldy, c, v);
      else store_bf3_1(zbar3 + (size_t)i * 3 * ldy, c, v);
    }
    else zr[c] = v;
  }
  if (to_lds) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    unsigned short* zr3 = zbar3 + (size_t)i * (fmt == 2 ? 2 : 3) * ldy;
    if (fmt == 2)
      for (int c = 4 * lane; c < ldy; c += 256) store_h2_x4(zr3, c, f32x4{st[c], st[c + 1], st[c + 2], st[c + 3]});
    else
      for (int c = 4 * lane; c < ldy; c += 256) store_bf3_x4(zr3, c, f32x4{st[c], st[c + 1], st[c + 2], st[c + 3]});
  }
  if (lane < nq) qbar_next[(size_t)lane * n_atoms + i] = carry_q ? (lane == 0 ? qr[0] : qr[1]) : 0.0f;
}

int launch_build_zbar(hipStream_t s, const float* qbar, const float* abar, const float* y, int ldy, const float* dlast,
                      const float* Fm, const float* Dm, const float* wpart, int S, const int* mol_idx, int n_atoms, int n_mol,
                      int n_feat, int nq, bool carry_q, float* zbar, float* qbar_next, int zbar_split, const int* mol_start,
                      const float* owned) {
  if (!wpart && (!mol_start || qbar_next == qbar)) {  // (the blocks re-read qbar of whole molecules: it must not change under them)
    set_last_error("build_zbar: the merged form needs mol_start and a qbar_next that is not qbar");
    return -1;
  }
  hipLaunchKernelGGL(build_zbar_kernel, dim3(ceil_div(n_atoms, 4)), dim3(256), 0, s, qbar, abar, y, ldy, dlast, Fm, Dm, wpart,
                     S < 1 ? 1 : S, mol_idx, n_atoms, n_mol, n_feat, nq, carry_q ? 1 : 0, zbar, qbar_next,
                     zbar_split ? reinterpret_cast<unsigned short*>(zbar) : nullptr, zbar_split, mol_start, owned);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

// ---- domain decomposition: halo rows leave the sums --------------------------------------------------
// After the energy head and the Coulomb block: a halo copy contributes no energy (its pair-energy slot takes -SAE[Z], in double,
// so that e + SAE + pair sums to exactly 0), no Coulomb adjoint, no direct Coulomb force / virial, and seeds no backward sweep
// (its seed row is zeroed: zero is zero in every operand format).  One wave per atom.
__global__ __launch_bounds__(256) void dd_mask_kernel(const float* __restrict__ owned, const int* __restrict__ numbers,
                                                      const double* __restrict__ sae, float* __restrict__ e_atom,
                                                      double* __restrict__ ecoul, float* __restrict__ qbar, int nq,
                                                      float* __restrict__ fgrad, float* __restrict__ virial_atom,
                                                      unsigned int* __restrict__ seed, int seed_row_words, int n_atoms) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n_atoms || owned[i] != 0.0f) return;
  if (lane == 0) {
    e_atom[i] = 0.0f;
    ecoul[i] = -sae[min(63, max(0, numbers[i]))];
  }
  if (qbar && lane < nq) qbar[(size_t)lane * n_atoms + i] = 0.0f;
  if (fgrad && lane < 3) fgrad[(size_t)i * 3 + lane] = 0.0f;
  if (virial_atom && lane < 9) virial_atom[(size_t)i * 9 + lane] = 0.0f;
  if (seed)
    for (int c = lane; c < seed_row_words; c += 64) seed[(size_t)i * seed_row_words + c] = 0u;
}

int launch_dd_mask(hipStream_t s, const float* owned, const int* numbers, const double* sae, float* e_atom, double* ecoul,
                   float* qbar, int nq, float* fgrad, float* virial_atom, void* seed, int seed_row_bytes, int n_atoms) {
  hipLaunchKernelGGL(dd_mask_kernel, dim3(ceil_div(n_atoms, 4)), dim3(256), 0, s, owned, numbers, sae, e_atom, ecoul, qbar, nq,
                     fgrad, virial_atom, (unsigned int*)seed, seed_row_bytes / 4, n_atoms);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

// ---- outputs -------------------------------------------------------------------------------------
// riders (er.energy != NULL, S == 1): the molecule energy sums and the copy of the charges into the output - outputs only, so they
// need no launch of their own in front of the backward pass
__global__ __launch_bounds__(256) void forces_kernel(const float* __restrict__ fgrad, int n3, float* __restrict__ forces,
                                                     const int* __restrict__ mol_start, EnergyRider er, int n_force_blocks,
                                                     int* __restrict__ nf) {
  if ((int)blockIdx.x >= n_force_blocks) {
    const int b = blockIdx.x - n_force_blocks;
    if (b < er.n_mol) {
      energy_partial_block(er.e_atom, er.ecoul, er.numbers, er.sae, mol_start, 1, er.energy, b, nf);
    } else {
      const int e = (b - er.n_mol) * 256 + threadIdx.x;
      if (e < er.copy_n) er.copy_dst[e] = er.copy_src[e];
    }
    return;
  }
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n3) {
    const float f = fgrad[e];
    forces[e] = -f;
    if (nf && !isfinite(f)) atomicOr(nf, STATUS_NONFINITE);
  }
}

// stage 1: 9 partial sums per (system, slice); a single shared cell (n_cell == 1) spans all molecules
__global__ __launch_bounds__(256) void stress_partial_kernel(const float* __restrict__ virial_atom,
                                                            const int* __restrict__ mol_start, int n_cell, int n_mol,
                                                            int S, double* __restrict__ part, const float* __restrict__ fgrad,
                                                            int n_atoms, PairForceRider pf, EnergyRider er,
                                                            const float* __restrict__ cell, float* __restrict__ stress_whole,
                                                            int* __restrict__ nf) {
  // whole form (stress_whole != NULL, S == 1): ONE block per cell sums all of its atoms and writes the stress itself, one block
  // per molecule the energy - no finish launch; they come first in the grid and run beside the force-gather riders
  const int n_first = S * n_cell + (stress_whole ? er.n_mol : 0);
  if (stress_whole && (int)blockIdx.x >= n_cell && (int)blockIdx.x < n_first) {
    energy_whole_block(er.e_atom, er.ecoul, er.numbers, er.sae, mol_start, er.energy, blockIdx.x - n_cell, nf);
    return;
  }
  if ((int)blockIdx.x >= n_first) {  // riders (independent of the virial sums; a kernel boundary less each)
    const int b = blockIdx.x - n_first;
    if (b < pf.n_blocks)  // the force gather of the reverse-pair form
      pair_force_block(pf.nb_idx, pf.nb_cnt, pf.rev, pf.pairbuf, pf.cap, n_atoms, fgrad, pf.forces, b, nf);
    else  // the molecule energies' partial sums (S == 1: the energies themselves)
      energy_partial_block(er.e_atom, er.ecoul, er.numbers, er.sae, mol_start, S, S == 1 ? er.energy : er.part, b - pf.n_blocks, nf);
    return;
  }
  __shared__ double sh[4];
  const int sidx = blockIdx.x / S, sl = blockIdx.x % S;
  const int i0 = (n_cell == 1) ? mol_start[0] : mol_start[sidx];
  const int i1 = (n_cell == 1) ? mol_start[n_mol] : mol_start[sidx + 1];
  if (stress_whole) {  // four atoms (36 floats) per thread and round in flight
    double a9[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) a9[k] = 0.0;
    for (int j0 = i0 + (int)threadIdx.x; j0 < i1; j0 += 4 * (int)blockDim.x) {
      float v[4][9];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = j0 + u * (int)blockDim.x, jj = j < i1 ? j : i0;
#pragma unroll
        for (int k = 0; k < 9; ++k) v[u][k] = virial_atom[(size_t)jj * 9 + k];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (j0 + u * (int)blockDim.x < i1) {
#pragma unroll
          for (int k = 0; k < 9; ++k) a9[k] += (double)v[u][k];
        }
    }
    const float* c = cell + (size_t)sidx * 9;
    const double det = (double)c[0] * ((double)c[4] * c[8] - (double)c[5] * c[7]) -
                       (double)c[1] * ((double)c[3] * c[8] - (double)c[5] * c[6]) +
                       (double)c[2] * ((double)c[3] * c[7] - (double)c[4] * c[6]);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const double r = block_sum(a9[k], sh);
      if (threadIdx.x == 0) stress_whole[sidx * 9 + k] = (float)(r / fabs(det));
    }
    return;
  }
  const int L = (i1 - i0 + S - 1) / S;
  const int lo = min(i1, i0 + sl * L), hi = min(i1, lo + L);
  for (int k = 0; k < 9; ++k) {
    double acc = 0.0;
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) acc += (double)virial_atom[(size_t)i * 9 + k];
    const double r = block_sum(acc, sh);
    if (threadIdx.x == 0) part[((size_t)sidx * S + sl) * 9 + k] = r;
  }
}

__global__ void stress_finish_kernel(const double* __restrict__ part, int S, int n_cell, const float* __restrict__ cell,
                                     float* __restrict__ stress, EnergyRider er) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_cell * 9) {  // rider threads: the slice sums of the molecule energies, in slice order (energy_finish_kernel)
    const int m = e - n_cell * 9;
    if (S > 1 && m < er.n_mol) {
      double r = 0.0;
      for (int k = 0; k < S; ++k) r += er.part[(size_t)m * S + k];
      er.energy[m] = r;
    }
    return;
  }
  const int sidx = e / 9, k = e % 9;
  const float* c = cell + (size_t)sidx * 9;
  const double det = (double)c[0] * ((double)c[4] * c[8] - (double)c[5] * c[7]) -
                     (double)c[1] * ((double)c[3] * c[8] - (double)c[5] * c[6]) +
                     (double)c[2] * ((double)c[3] * c[7] - (double)c[4] * c[6]);
  double r = 0.0;
  for (int s = 0; s < S; ++s) r += part[((size_t)sidx * S + s) * 9 + k];
  stress[e] = (float)(r / fabs(det));
}

int launch_finalize(hipStream_t s, const float* fgrad, const float* virial_atom, const int* mol_start,
                    const float* cell, int n_cell, int n_mol, int n_atoms, int S, double* part, float* forces,
                    float* stress, const PairForceRider* pair_force, const EnergyRider* energy, bool whole_ok, int* nf) {
  PairForceRider pf{};
  if (pair_force) pf = *pair_force;
  EnergyRider er{};
  if (energy) er = *energy;
  const bool er_on_forces = energy && forces && !(stress && cell);  // (the engine only asks for this with S == 1)
  if (forces) {
    const int nfb = ceil_div(3 * n_atoms, 256);
    const int n_rider = er_on_forces ? er.n_mol + (er.copy_dst ? ceil_div(er.copy_n, 256) : 0) : 0;
    hipLaunchKernelGGL(forces_kernel, dim3(nfb + n_rider), dim3(256), 0, s, fgrad, 3 * n_atoms, forces, mol_start, er, nfb, nf);
    AIMNET_LAUNCH_CHECK();
  }
  if (stress && cell) {
    S = S < 1 ? 1 : S;
    // whole form: up to 16 384 atoms per launch with the energy riders and the force gather beside them (that gather is what the
    // single blocks hide behind) - the sums need no slices and no finish launch
    const bool whole = energy && pf.n_blocks > 0 && n_atoms <= 16384 && whole_ok;
    if (whole) {
      hipLaunchKernelGGL(stress_partial_kernel, dim3(n_cell + er.n_mol + pf.n_blocks), dim3(256), 0, s, virial_atom, mol_start, n_cell, n_mol,
                         1, part, fgrad, n_atoms, pf, er, cell, stress, nf);
      AIMNET_LAUNCH_CHECK();
      return 0;
    }
    const int n_er = energy ? S * er.n_mol : 0;
    hipLaunchKernelGGL(stress_partial_kernel, dim3(S * n_cell + pf.n_blocks + n_er), dim3(256), 0, s, virial_atom, mol_start, n_cell, n_mol,
                       S, part, fgrad, n_atoms, pf, er, cell, nullptr, nf);
    AIMNET_LAUNCH_CHECK();
    hipLaunchKernelGGL(stress_finish_kernel, dim3(ceil_div(n_cell * 9 + (energy && S > 1 ? er.n_mol : 0), 64)), dim3(64), 0, s, part, S,
                       n_cell, cell, stress, er);
    AIMNET_LAUNCH_CHECK();
  }
  return 0;
}

__global__ void copy_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) dst[e] = src[e];
}

int launch_copy_f32(hipStream_t s, const float* src, float* dst, size_t n) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(copy_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, n);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

}  // namespace aimnet
