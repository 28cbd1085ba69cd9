// hvp.hip - analytic Hessian-vector products: H v = d/d eps [dE/dx (x + eps v)] by a forward-mode TANGENT SWEEP through the
// forward and the backward sweep of the model, K directions at once.
//
// Replaces AIMNet2Calculator.hessian_vector_product (calculators/calculator.py:1753-1989: vjp of the force graph, one
// double backward per vector) and calculate_hessian (calculators/derivatives.py:149-192: vmap of that vjp over the 3N unit
// vectors); the op-level second derivatives the reference gets from conv_sv_2d_sp_bwd_bwd (kernels/conv_sv_2d_sp_wp.py:
// 167-244) appear here as the product-rule terms of hvp_conv_bwd_kernel.  Executable specification, block by block:
// oracle/aimnet2_analytic.py::evaluate_hvp (pinned to the autograd Hessian in fp64, tests/test_oracle_analytic.py).
//
// Shape of the computation.  Every quantity X of the force evaluation gets a tangent tX with a leading direction axis:
// primal arrays are [N][w], tangent arrays [K][N][w].  Linear steps (the MLP GEMMs) are the SAME GEMM on the stacked rows
// [N primal | K N tangent] (the engine's GEMM family through mlp_gemm: gemm_bf3.hip above 256 rows, the exact-fp32 kernels of
// gemm.hip below; no bias); the nonlinear steps are the kernels below, one block (or wave) per
// (atom, direction): it recomputes the primal values it needs from the primal inputs (pair geometry, radial basis, the
// centre's own moments) and applies the product rule.  A block with k == 0 also WRITES the primal outputs, so one sweep
// produces the forces as well; no kernel reads a primal array that the same launch writes.  All pair kernels keep the
// centre-major gather form of the force kernels (conv.hip): every ordered pair is visited from its centre, both halves of
// its adjoint are evaluated there, nothing is scattered, no atomics - results are bitwise reproducible.
//
// The second-order pieces: GELU''(z) = phi(z) (2 - z^2) in the backward of every hidden layer (hvp_act_bwd), the second
// derivative of the radial basis gs''(d) (the cosine envelope's jumps at rc: the Hessian is discontinuous where a pair crosses
// the cutoff, as the reference's is), the tangent of the unit vector t_u = (t_r - u (u . t_r)) / d, second derivatives of the
// Coulomb pair weights, and the tangent of the NSE charge normalisation and of its adjoint.
//
// These kernels are written for clarity and exactness, not for the roofline: a Hessian of a 40-atom molecule is 120
// directions x 40 atoms = 4 800 tangent rows (the GEMMs see a 4 800-row batch), everything else is far below a millisecond.
#include <algorithm>

#include "conv_common.h"
#include "engine.h"

namespace aimnet {
namespace {

constexpr int HCH = 64;  // neighbours staged in LDS per chunk

// Gaussian x envelope of one radial shift with two derivatives; fc3 = (fc, fc', fc'') of the pair
__device__ __forceinline__ void basis_g2(float eta, float shift, float d, float3 fc3, float& gs, float& dgs, float& d2gs) {
  const float x = d - shift;
  const float G = exp_neg(-eta * x * x);
  const float dG = -2.0f * eta * x * G;
  const float d2G = (4.0f * eta * eta * x * x - 2.0f * eta) * G;
  gs = G * fc3.x;
  dgs = dG * fc3.x + G * fc3.y;
  d2gs = d2G * fc3.x + 2.0f * dG * fc3.y + G * fc3.z;
}

__device__ __forceinline__ float3 envelope3(const BasisParams& bp, float d) {
  const float w = PI_F / bp.rc;
  const float dc = fminf(fmaxf(d, 1e-6f), bp.rc);
  float sn, cs;
  sincosf(dc * w, &sn, &cs);
  const bool in = d > 1e-6f && d < bp.rc;
  return make_float3(0.5f * (cs + 1.0f), in ? -0.5f * w * sn : 0.0f, in ? -0.5f * w * w * cs : 0.0f);
}

// tangent of the pair geometry: t_r = v_j - v_i (the cell is fixed), t_d = u . t_r, t_u = (t_r - u t_d) / d
__device__ __forceinline__ float4 geom_tangent(float4 ud, const float* __restrict__ tvk, int i, int j) {
  const float rx = tvk[3 * j] - tvk[3 * i], ry = tvk[3 * j + 1] - tvk[3 * i + 1], rz = tvk[3 * j + 2] - tvk[3 * i + 2];
  const float td = ud.x * rx + ud.y * ry + ud.z * rz;
  const float inv = 1.0f / ud.w;
  return make_float4((rx - ud.x * td) * inv, (ry - ud.y * td) * inv, (rz - ud.z * td) * inv, td);
}

template <typename T>
__device__ __forceinline__ T block_sum256(T v, T* sh) {  // 256 threads
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

struct Stage {  // one chunk of a centre's neighbour row in LDS
  float4 u[HCH];   // (u, d)
  float4 tu[HCH];  // (t_u, t_d)
  float4 fc[HCH];  // (fc, fc', fc'', -)
  int j[HCH];
};

__device__ __forceinline__ void stage_chunk(Stage& st, int i, int m0, int cnt, const int* __restrict__ nb_idx,
                                            const float4* __restrict__ pg, int cap, const float* __restrict__ tvk,
                                            const BasisParams& bp) {
  __syncthreads();
  const int t = threadIdx.x;
  if (t < HCH && m0 + t < cnt) {
    const size_t p = (size_t)i * cap + m0 + t;
    const float4 ud = pg[p];
    const int j = nb_idx[p];
    st.u[t] = ud;
    st.tu[t] = geom_tangent(ud, tvk, i, j);
    const float3 f = envelope3(bp, ud.w);
    st.fc[t] = make_float4(f.x, f.y, f.z, 0.0f);
    st.j[t] = j;
  }
  __syncthreads();
}

// ---- conv forward + tangent ------------------------------------------------------------------------------------------
// block = (centre i, direction k), thread = feature (a, g).  S[f][c] = sum_m a_j[f] gs_g (1, u)[c] and
// t_S = sum_m t_a_j gs (1,u) + a_j t_gs (1,u) + a_j gs (0, t_u); then V = agh . S_vec, the MLP input row
// [a | S0 | |V|^2 | q | Sq0 | |Vq|^2] and its tangent [t_a | t_S0 | 2 V . t_V | ...].
template <int NQ>
__global__ __launch_bounds__(256) void hvp_conv_fwd_kernel(const float* __restrict__ a, const int* __restrict__ row_of,
                                                          const float* __restrict__ ta, const float* __restrict__ q,
                                                          const float* __restrict__ tq, const int* __restrict__ nb_idx,
                                                          const int* __restrict__ nb_cnt, const float4* __restrict__ pg, int cap,
                                                          const float* __restrict__ tv, const float* __restrict__ agh_a,
                                                          const float* __restrict__ agh_q, BasisParams bp,
                                                          float* __restrict__ x, float* __restrict__ tx, int ldx,
                                                          float* __restrict__ V, float* __restrict__ tV,
                                                          float* __restrict__ Vq, float* __restrict__ tVq, int N) {
  __shared__ Stage st;
  __shared__ float sS[NF * 4], stS[NF * 4];
  __shared__ float sSq[(NQ ? NQ : 1) * 64], stSq[(NQ ? NQ : 1) * 64];
  __shared__ float s_shift[16];
  const int i = blockIdx.x, k = blockIdx.y, f = threadIdx.x, g = f & 15;
  if (f < 16) s_shift[f] = bp.shifts[f];
  const float* tvk = tv + (size_t)k * N * 3;
  const int cnt = nb_cnt[i];
  float S[4] = {0, 0, 0, 0}, tS[4] = {0, 0, 0, 0}, Sq[4] = {0, 0, 0, 0}, tSq[4] = {0, 0, 0, 0};
  const bool qthr = NQ > 0 && f < NQ * 16;
  const int qc = f >> 4;
  for (int m0 = 0; m0 < cnt; m0 += HCH) {
    stage_chunk(st, i, m0, cnt, nb_idx, pg, cap, tvk, bp);
    const float shift = s_shift[g];
    const int mc = min(HCH, cnt - m0);
    for (int m = 0; m < mc; ++m) {
      const float4 u = st.u[m], tu = st.tu[m], fc = st.fc[m];
      const int j = st.j[m];
      float gs, dgs, d2gs;
      basis_g2(bp.eta, shift, u.w, make_float3(fc.x, fc.y, fc.z), gs, dgs, d2gs);
      const float tgs = dgs * tu.w;
      const size_t rj = row_of ? (size_t)min(63, max(0, row_of[j])) : (size_t)j;
      const float aj = a[rj * NF + f];
      const float taj = ta ? ta[((size_t)k * N + j) * NF + f] : 0.0f;
      const float w0 = aj * gs, tw0 = taj * gs + aj * tgs;
      S[0] += w0; S[1] += w0 * u.x; S[2] += w0 * u.y; S[3] += w0 * u.z;
      tS[0] += tw0;
      tS[1] += tw0 * u.x + w0 * tu.x;
      tS[2] += tw0 * u.y + w0 * tu.y;
      tS[3] += tw0 * u.z + w0 * tu.z;
      if (qthr) {
        const float qj = q[(size_t)qc * N + j], tqj = tq[((size_t)k * NQ + qc) * N + j];
        const float v0 = qj * gs, tv0 = tqj * gs + qj * tgs;
        Sq[0] += v0; Sq[1] += v0 * u.x; Sq[2] += v0 * u.y; Sq[3] += v0 * u.z;
        tSq[0] += tv0;
        tSq[1] += tv0 * u.x + v0 * tu.x;
        tSq[2] += tv0 * u.y + v0 * tu.y;
        tSq[3] += tv0 * u.z + v0 * tu.z;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    sS[f * 4 + c] = S[c];
    stS[f * 4 + c] = tS[c];
    if (qthr) {
      sSq[f * 4 + c] = Sq[c];
      stSq[f * 4 + c] = tSq[c];
    }
  }
  __syncthreads();
  const size_t ri = row_of ? (size_t)min(63, max(0, row_of[i])) : (size_t)i;
  const size_t tr = (size_t)k * N + i;
  float* xr = x + (size_t)i * ldx;
  float* txr = tx + tr * ldx;
  const bool prim = k == 0;
  if (prim) {
    xr[f] = a[ri * NF + f];
    xr[NF + f] = S[0];
  }
  txr[f] = ta ? ta[tr * NF + f] : 0.0f;
  txr[NF + f] = tS[0];
  if (f < NV) {  // thread = (a, h)
    const int aa = f / H_, h = f % H_;
    float v[3] = {0, 0, 0}, tvv[3] = {0, 0, 0};
    for (int gg = 0; gg < G_; ++gg) {
      const float w = agh_a[(aa * G_ + gg) * H_ + h];
      const int s = (aa * G_ + gg) * 4;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        v[c] += w * sS[s + 1 + c];
        tvv[c] += w * stS[s + 1 + c];
      }
    }
    if (prim) {
      xr[2 * NF + f] = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
#pragma unroll
      for (int c = 0; c < 3; ++c) V[(size_t)i * (NV * 3) + f * 3 + c] = v[c];
    }
    txr[2 * NF + f] = 2.0f * (v[0] * tvv[0] + v[1] * tvv[1] + v[2] * tvv[2]);
#pragma unroll
    for (int c = 0; c < 3; ++c) tV[tr * (NV * 3) + f * 3 + c] = tvv[c];
  }
  if (NQ > 0) {
    const int c0 = 2 * NF + NV;
    if (f < NQ) {
      if (prim) xr[c0 + f] = q[(size_t)f * N + i];
      txr[c0 + f] = tq[((size_t)k * NQ + f) * N + i];
    }
    if (qthr) {
      if (prim) xr[c0 + NQ + f] = Sq[0];
      txr[c0 + NQ + f] = tSq[0];
    }
    if (f < NQ * H_) {  // thread = (channel, h)
      const int cq = f / H_, h = f % H_;
      float v[3] = {0, 0, 0}, tvv[3] = {0, 0, 0};
      for (int gg = 0; gg < G_; ++gg) {
        const float w = agh_q[(cq * G_ + gg) * H_ + h];
        const int s = (cq * G_ + gg) * 4;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          v[c] += w * sSq[s + 1 + c];
          tvv[c] += w * stSq[s + 1 + c];
        }
      }
      if (prim) {
        xr[c0 + NQ + NQ * G_ + f] = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
#pragma unroll
        for (int c = 0; c < 3; ++c) Vq[(size_t)i * (NQ * H_ * 3) + f * 3 + c] = v[c];
      }
      txr[c0 + NQ + NQ * G_ + f] = 2.0f * (v[0] * tvv[0] + v[1] * tvv[1] + v[2] * tvv[2]);
#pragma unroll
      for (int c = 0; c < 3; ++c) tVq[tr * (NQ * H_ * 3) + f * 3 + c] = tvv[c];
    }
  }
}

// ---- activations -----------------------------------------------------------------------------------------------------
// forward: h = GELU(z + b), t_h = GELU'(z + b) t_z.  Primal and tangent rows go through ONE bias-free GEMM launch (the primal
// rows stacked on top of the K N tangent rows), so z is stored without the bias and every consumer adds it
__global__ void hvp_act_fwd_kernel(const float* __restrict__ z, const float* __restrict__ tz, const float* __restrict__ bias, int ld,
                                   int N, size_t n_t, float* __restrict__ h, float* __restrict__ th) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_t) return;
  const size_t per = (size_t)N * ld;
  const size_t pe = e % per;
  float hv, d1;
  gelu_and_grad(z[pe] + bias[pe % ld], hv, d1);
  th[e] = d1 * tz[e];
  if (e < per) h[e] = hv;
}

__device__ __forceinline__ float gelu_grad2(float z) {  // GELU''(z) = phi(z) (2 - z^2)
  return 0.39894228040143268f * __builtin_amdgcn_exp2f(-0.72134752044448170f * z * z) * (2.0f - z * z);
}

// backward: t = g GELU'(z), t_t = t_g GELU'(z) + g GELU''(z) t_z.  g_row0: g is one row broadcast over the atoms (the energy
// head's last layer, whose tangent is zero: tg == NULL)
__global__ void hvp_act_bwd_kernel(const float* __restrict__ g, int g_row0, const float* __restrict__ tg,
                                   const float* __restrict__ z, const float* __restrict__ tz, const float* __restrict__ bias, int ld,
                                   int N, size_t n_t, float* __restrict__ t, float* __restrict__ tt) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_t) return;
  const size_t per = (size_t)N * ld;
  const size_t pe = e % per;
  const float zz = z[pe] + bias[pe % ld];
  float hv, d1;
  gelu_and_grad(zz, hv, d1);
  const float gv = g_row0 ? g[pe % ld] : g[pe];
  tt[e] = (tg ? tg[e] * d1 : 0.0f) + gv * gelu_grad2(zz) * tz[e];
  if (e < per) t[e] = gv * d1;
}

// one row (the last head layer's weights, zero beyond k) into a padded buffer
__global__ void hvp_pad_row_kernel(const float* __restrict__ w, int k, int ld, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < ld) out[c] = c < k ? w[c] : 0.0f;
}

// ---- NSE charge update + tangent (ops.py:99-145, aimnet2.py:122-139) ---------------------------------------------------
// block = (molecule, direction).  y row = [q~ (nq) | f~ (nq) | delta_a]
__global__ __launch_bounds__(256) void hvp_nse_fwd_kernel(const float* __restrict__ y, const float* __restrict__ ty, int ldy, int nq,
                                                         const float* __restrict__ q_prev, const float* __restrict__ tq_prev,
                                                         const int* __restrict__ mol_start, const float* __restrict__ charge,
                                                         int n_mol, int N, float* __restrict__ q_new, float* __restrict__ tq_new,
                                                         float* __restrict__ Fm, float* __restrict__ Dm,
                                                         float* __restrict__ tFm, float* __restrict__ tDm) {
  __shared__ float sh[4];
  const int m = blockIdx.x, k = blockIdx.y;
  const int i0 = mol_start[m], i1 = mol_start[m + 1];
  for (int ch = 0; ch < nq; ++ch) {
    float sf = 0.f, sq = 0.f, tsf = 0.f, tsq = 0.f;
    for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
      const float* yr = y + (size_t)i * ldy;
      const float* tyr = ty + ((size_t)k * N + i) * ldy;
      const float ft = yr[nq + ch];
      sf += ft * ft;
      tsf += 2.0f * ft * tyr[nq + ch];
      sq += (q_prev ? q_prev[(size_t)ch * N + i] : 0.0f) + yr[ch];
      tsq += (tq_prev ? tq_prev[((size_t)k * nq + ch) * N + i] : 0.0f) + tyr[ch];
    }
    const float F = block_sum256(sf, sh) + 1.0e-6f;
    const float D = charge[(size_t)ch * n_mol + m] - block_sum256(sq, sh);
    const float tF = block_sum256(tsf, sh);
    const float tD = -block_sum256(tsq, sh);
    if (threadIdx.x == 0) {
      if (k == 0) {
        Fm[(size_t)ch * n_mol + m] = F;
        Dm[(size_t)ch * n_mol + m] = D;
      }
      tFm[((size_t)k * nq + ch) * n_mol + m] = tF;
      tDm[((size_t)k * nq + ch) * n_mol + m] = tD;
    }
    for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
      const float* yr = y + (size_t)i * ldy;
      const float* tyr = ty + ((size_t)k * N + i) * ldy;
      const float ft = yr[nq + ch], tft = tyr[nq + ch];
      const float f = ft * ft, tf = 2.0f * ft * tft;
      const float qr = (q_prev ? q_prev[(size_t)ch * N + i] : 0.0f) + yr[ch];
      const float tqr = (tq_prev ? tq_prev[((size_t)k * nq + ch) * N + i] : 0.0f) + tyr[ch];
      if (k == 0) q_new[(size_t)ch * N + i] = qr + f / F * D;
      tq_new[((size_t)k * nq + ch) * N + i] = tqr + tf / F * D - f * tF / (F * F) * D + f / F * tD;
    }
  }
}

// a_new = a + delta_a, t_a_new = t_a + t_delta_a   (row_of: pass 0 reads the embedding row, whose tangent is zero)
__global__ void hvp_update_a_kernel(const float* __restrict__ a, const int* __restrict__ row_of, const float* __restrict__ ta,
                                    const float* __restrict__ y, const float* __restrict__ ty, int ldy, int col0, int N,
                                    size_t n_t, float* __restrict__ a_new, float* __restrict__ ta_new) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_t) return;
  const size_t row = e >> 8;  // k N + i
  const int f = (int)(e & 255);
  const size_t i = row % (size_t)N;
  ta_new[e] = (ta ? ta[e] : 0.0f) + ty[row * ldy + col0 + f];
  if (row < (size_t)N) {
    const size_t ri = row_of ? (size_t)min(63, max(0, row_of[i])) : i;
    a_new[e] = a[ri * NF + f] + y[i * ldy + col0 + f];
  }
}

// ---- Coulomb pair terms: adjoint seeds (dE/dq, dE/dx) and their tangents ----------------------------------------------
// wave = (centre i, direction k), lanes over the neighbours of a full symmetric list; E = sign k sum w(d) q_i q_j.
//   qbar_i  += 2 sign k sum_m w q_j            t_qbar_i += 2 sign k sum_m (w' t_d q_j + w t_q_j)
//   xbar_i  -= 2 sign k sum_m w' q_i q_j u     t_xbar_i -= 2 sign k sum_m [(w'' t_d q_i q_j + w' t(q_i q_j)) u + w' q_i q_j t_u]
struct CoulAcc {
  float qb = 0.f, tqb = 0.f, f[3] = {0, 0, 0}, tf[3] = {0, 0, 0};
};
__device__ __forceinline__ void coul_add(CoulAcc& A, float w, float dw, float d2w, float qi, float tqi, float qj, float tqj,
                                         float4 u, float4 tu) {
  A.qb += w * qj;
  A.tqb += dw * tu.w * qj + w * tqj;
  const float qq = qi * qj, tqq = tqi * qj + qi * tqj;
  const float t = dw * qq, tt = d2w * tu.w * qq + dw * tqq;
  A.f[0] += t * u.x; A.f[1] += t * u.y; A.f[2] += t * u.z;
  A.tf[0] += tt * u.x + t * tu.x;
  A.tf[1] += tt * u.y + t * tu.y;
  A.tf[2] += tt * u.z + t * tu.z;
}
__device__ __forceinline__ void coul_store(CoulAcc& A, int i, int k, int N, int nq, int lane, float sign_k, float self, float qi,
                                           float tqi, bool accum, float* qbar, float* tqbar, float* xbar, float* txbar) {
  const float qb = wave_sum(A.qb), tqb = wave_sum(A.tqb);
  float f[3], tf[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    f[c] = wave_sum(A.f[c]);
    tf[c] = wave_sum(A.tf[c]);
  }
  if (lane != 0) return;
  const float qv = 2.0f * sign_k * qb + self * qi, tqv = 2.0f * sign_k * tqb + self * tqi;
  for (int ch = 0; ch < nq; ++ch) {  // the Coulomb terms see alpha + beta: the seed is the same for every channel
    const size_t pe = (size_t)ch * N + i, te = ((size_t)k * nq + ch) * N + i;
    if (k == 0) qbar[pe] = accum ? qbar[pe] + qv : qv;
    tqbar[te] = accum ? tqbar[te] + tqv : tqv;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const size_t pe = (size_t)i * 3 + c, te = ((size_t)k * N + i) * 3 + c;
    const float v = -2.0f * sign_k * f[c], tvv = -2.0f * sign_k * tf[c];
    if (k == 0) xbar[pe] = accum ? xbar[pe] + v : v;
    txbar[te] = accum ? txbar[te] + tvv : tvv;
  }
}

// total charge of an atom (NSE models: alpha + beta) and its tangent
__device__ __forceinline__ void q_total(const float* __restrict__ q, const float* __restrict__ tq, int nq, int N, int k, int i,
                                        float& qv, float& tqv) {
  qv = q[i];
  tqv = tq[((size_t)k * nq) * N + i];
  if (nq == 2) {
    qv += q[(size_t)N + i];
    tqv += tq[((size_t)k * nq + 1) * N + i];
  }
}

// embedded short-range Coulomb (subtracted; lr.py:21-62): initialises the seeds (enabled == false: zeros)
__global__ __launch_bounds__(256) void hvp_coulomb_sr_kernel(bool enabled, const float* __restrict__ q, const float* __restrict__ tq,
                                                            int nq, const int* __restrict__ nb_idx, const int* __restrict__ nb_cnt,
                                                            const float4* __restrict__ pg, int cap, const float* __restrict__ tv,
                                                            CoulombParams cp, int N, float* __restrict__ qbar,
                                                            float* __restrict__ tqbar, float* __restrict__ xbar,
                                                            float* __restrict__ txbar) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), k = blockIdx.y;
  if (i >= N) return;
  const int lane = threadIdx.x & 63;
  const float* tvk = tv + (size_t)k * N * 3;
  float qi, tqi;
  q_total(q, tq, nq, N, k, i, qi, tqi);
  CoulAcc A;
  if (enabled) {
    const int cnt = nb_cnt[i];
    const float rc = cp.sr_rc;
    for (int m = lane; m < cnt; m += 64) {
      const size_t p = (size_t)i * cap + m;
      const float4 u = pg[p];
      const int j = nb_idx[p];
      const float4 tu = geom_tangent(u, tvk, i, j);
      const float d = u.w;
      float fc, dfc, d2fc;
      if (cp.sr_envelope == 0) {  // exp envelope: fc = exp(-1 / (1 - t^2)) e, t = d / rc
        const float tr = d / rc;
        const float t = fminf(fmaxf(tr, 0.0f), 1.0f - 1e-6f);
        const float om = 1.0f - t * t;
        fc = expf(-1.0f / om) / 0.36787944117144233f;
        const bool live = tr < 1.0f - 1e-6f;
        const float s1 = -2.0f * t / (om * om), s2 = -2.0f * (1.0f + 3.0f * t * t) / (om * om * om);
        dfc = live ? fc * s1 / rc : 0.0f;
        d2fc = live ? fc * (s1 * s1 + s2) / (rc * rc) : 0.0f;
      } else {
        const float dc = fminf(fmaxf(d, 1e-6f), rc);
        const float w = PI_F / rc;
        const bool live = d > 1e-6f && d < rc;
        fc = 0.5f * (cosf(dc * w) + 1.0f);
        dfc = live ? -0.5f * w * sinf(dc * w) : 0.0f;
        d2fc = live ? -0.5f * w * w * cosf(dc * w) : 0.0f;
      }
      const float inv = 1.0f / d;
      float qj, tqj;
      q_total(q, tq, nq, N, k, j, qj, tqj);
      coul_add(A, fc * inv, dfc * inv - fc * inv * inv, d2fc * inv - 2.0f * dfc * inv * inv + 2.0f * fc * inv * inv * inv, qi, tqi,
               qj, tqj, u, tu);
    }
  }
  coul_store(A, i, k, N, nq, lane, -cp.factor, 0.0f, qi, tqi, false, qbar, tqbar, xbar, txbar);
}

// "simple" (every other atom of the molecule, w = 1/d; lr.py:311-331) or DSF over the long-range list (lr.py:559-615)
template <bool DSF>
__global__ __launch_bounds__(256) void hvp_coulomb_lr_kernel(const float* __restrict__ q, const float* __restrict__ tq, int nq,
                                                            const float* __restrict__ xw, const int* __restrict__ mol_idx,
                                                            const int* __restrict__ mol_start, const float* __restrict__ cell,
                                                            int n_cell, const int* __restrict__ nb_idx,
                                                            const int* __restrict__ nb_shift, const int* __restrict__ nb_cnt,
                                                            int cap, const float* __restrict__ tv, CoulombParams cp, int N,
                                                            float* __restrict__ qbar, float* __restrict__ tqbar,
                                                            float* __restrict__ xbar, float* __restrict__ txbar) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), k = blockIdx.y;
  if (i >= N) return;
  const int lane = threadIdx.x & 63;
  const float* tvk = tv + (size_t)k * N * 3;
  float qi, tqi;
  q_total(q, tq, nq, N, k, i, qi, tqi);
  const float xi = xw[3 * i], yi = xw[3 * i + 1], zi = xw[3 * i + 2];
  const float al = cp.dsf_alpha, Rc = cp.dsf_rc;
  const float cpi = 2.0f * al * 0.56418958354775629f;
  const float erfc_rc = erfcf(al * Rc);
  const float sv = erfc_rc / Rc;
  const float slope = erfc_rc / (Rc * Rc) + cpi * expf(-al * al * Rc * Rc) / Rc;
  const int mi = mol_idx[i];
  const float* c = (DSF && cell) ? cell + (n_cell == 1 ? 0 : (size_t)mi * 9) : nullptr;
  const int lo = DSF ? 0 : mol_start[mi], hi = DSF ? nb_cnt[i] : mol_start[mi + 1];
  CoulAcc A;
  for (int m = lo + lane; m < hi; m += 64) {
    int j;
    float rx, ry, rz;
    if (DSF) {
      const size_t p = (size_t)i * cap + m;
      j = nb_idx[p];
      rx = xw[3 * j] - xi; ry = xw[3 * j + 1] - yi; rz = xw[3 * j + 2] - zi;
      if (c) {
        int sx, sy, sz;
        unpack_shift(nb_shift[p], sx, sy, sz);
        rx += sx * c[0] + sy * c[3] + sz * c[6];
        ry += sx * c[1] + sy * c[4] + sz * c[7];
        rz += sx * c[2] + sy * c[5] + sz * c[8];
      }
    } else {
      j = m;
      if (j == i) continue;
      rx = xw[3 * j] - xi; ry = xw[3 * j + 1] - yi; rz = xw[3 * j + 2] - zi;
    }
    const float d = sqrtf(rx * rx + ry * ry + rz * rz);
    if (DSF && !(d < Rc)) continue;
    const float inv = 1.0f / d;
    const float4 u = make_float4(rx * inv, ry * inv, rz * inv, d);
    const float4 tu = geom_tangent(u, tvk, i, j);
    float w, dw, d2w;
    if (DSF) {
      const float ec = erfcf(al * d), ex = expf(-al * al * d * d);
      w = ec * inv - sv + (d - Rc) * slope;
      dw = -ec * inv * inv - cpi * ex * inv + slope;
      d2w = 2.0f * ec * inv * inv * inv + 2.0f * cpi * ex * inv * inv + 2.0f * al * al * cpi * ex;
    } else {
      w = inv;
      dw = -inv * inv;
      d2w = 2.0f * inv * inv * inv;
    }
    float qj, tqj;
    q_total(q, tq, nq, N, k, j, qj, tqj);
    coul_add(A, w, dw, d2w, qi, tqi, qj, tqj, u, tu);
  }
  const float self = DSF ? -4.0f * cp.factor * (sv * 0.5f + al * 0.56418958354775629f) : 0.0f;  // d/dq of 2 k cs q^2
  coul_store(A, i, k, N, nq, lane, cp.factor, self, qi, tqi, true, qbar, tqbar, xbar, txbar);
}

// ---- unconcat + tangent: the adjoint of the MLP input row back onto (abar, Sbar, qbar, Sqbar) --------------------------
// Sbar[f][0] = xbar[256 + f];  Sbar[f][1 + c] = sum_h agh[a,g,h] 2 V[a,h,c] vbar[a,h]   and the product rule on V vbar
template <int NQ>
__global__ __launch_bounds__(256) void hvp_unconcat_kernel(const float* __restrict__ xb, const float* __restrict__ txb, int ldx,
                                                          const float* __restrict__ V, const float* __restrict__ tV,
                                                          const float* __restrict__ Vq, const float* __restrict__ tVq,
                                                          const float* __restrict__ agh_a, const float* __restrict__ agh_q,
                                                          float* __restrict__ abar, float* __restrict__ tabar,
                                                          float* __restrict__ qbar, float* __restrict__ tqbar,
                                                          float* __restrict__ Sbar, float* __restrict__ tSbar,
                                                          float* __restrict__ Sqbar, float* __restrict__ tSqbar, int N) {
  __shared__ float sVb[NV * 3], stVb[NV * 3];
  __shared__ float sVqb[(NQ ? NQ : 1) * H_ * 3], stVqb[(NQ ? NQ : 1) * H_ * 3];
  const int i = blockIdx.x, k = blockIdx.y, f = threadIdx.x;
  const size_t tr = (size_t)k * N + i;
  const float* xr = xb + (size_t)i * ldx;
  const float* txr = txb + tr * ldx;
  const bool prim = k == 0;
  if (abar) {
    if (prim) abar[(size_t)i * NF + f] += xr[f];
    tabar[tr * NF + f] += txr[f];
  }
  if (f < NV) {
    const float vb = xr[2 * NF + f], tvb = txr[2 * NF + f];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = V[(size_t)i * (NV * 3) + f * 3 + c], tvv = tV[tr * (NV * 3) + f * 3 + c];
      sVb[f * 3 + c] = 2.0f * v * vb;
      stVb[f * 3 + c] = 2.0f * (tvv * vb + v * tvb);
    }
  }
  const int c0 = 2 * NF + NV;
  if (NQ > 0 && f < NQ * H_) {
    const float vb = xr[c0 + NQ + NQ * G_ + f], tvb = txr[c0 + NQ + NQ * G_ + f];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = Vq[(size_t)i * (NQ * H_ * 3) + f * 3 + c], tvv = tVq[tr * (NQ * H_ * 3) + f * 3 + c];
      sVqb[f * 3 + c] = 2.0f * v * vb;
      stVqb[f * 3 + c] = 2.0f * (tvv * vb + v * tvb);
    }
  }
  __syncthreads();
  {
    const int aa = f >> 4;
    float s[3] = {0, 0, 0}, ts[3] = {0, 0, 0};
    for (int h = 0; h < H_; ++h) {
      const float w = agh_a[f * H_ + h];  // f = a * 16 + g
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        s[c] += w * sVb[(aa * H_ + h) * 3 + c];
        ts[c] += w * stVb[(aa * H_ + h) * 3 + c];
      }
    }
    if (prim) reinterpret_cast<float4*>(Sbar)[(size_t)i * NF + f] = make_float4(xr[NF + f], s[0], s[1], s[2]);
    reinterpret_cast<float4*>(tSbar)[tr * NF + f] = make_float4(txr[NF + f], ts[0], ts[1], ts[2]);
  }
  if (NQ > 0) {
    if (f < NQ) {
      if (prim) qbar[(size_t)f * N + i] += xr[c0 + f];
      tqbar[((size_t)k * NQ + f) * N + i] += txr[c0 + f];
    }
    if (f < NQ * G_) {
      const int cq = f >> 4;
      float s[3] = {0, 0, 0}, ts[3] = {0, 0, 0};
      for (int h = 0; h < H_; ++h) {
        const float w = agh_q[f * H_ + h];  // f = channel * 16 + g
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          s[c] += w * sVqb[(cq * H_ + h) * 3 + c];
          ts[c] += w * stVqb[(cq * H_ + h) * 3 + c];
        }
      }
      if (prim) reinterpret_cast<float4*>(Sqbar)[(size_t)i * (NQ * G_) + f] = make_float4(xr[c0 + NQ + f], s[0], s[1], s[2]);
      reinterpret_cast<float4*>(tSqbar)[tr * (NQ * G_) + f] = make_float4(txr[c0 + NQ + f], ts[0], ts[1], ts[2]);
    }
  }
}

// ---- conv backward + tangent ----------------------------------------------------------------------------------------
// dE/dr of an ordered pair from a thread's share of (dE/dd, dE/du): rbar = dbar u + (ubar - (ubar . u) u) / d, LINEAR in the
// share, so every thread accumulates its own 3-vector over the row and the block reduces once.  The two halves of a pair meet
// at the centre with opposite unit vectors, u_ji = -u_ij, and the projector (1 - u u^T) is even in u:
//   rbar_ji - rbar_ij = -(dbar_ij + dbar_ji) u + (1 - u u^T) (ubar_ji - ubar_ij) / d
// - one evaluation with D = dbar_ij + dbar_ji and U = ubar_ji - ubar_ij, and its tangent by the product rule (t_d = tu.w).
__device__ __forceinline__ void rbar_pair_add(float D, const float U[3], float tD, const float tU[3], float4 u, float4 tu,
                                              float acc[3], float tacc[3]) {
  const float ux[3] = {u.x, u.y, u.z}, tux[3] = {tu.x, tu.y, tu.z};
  const float inv = 1.0f / u.w;
  const float pu = U[0] * ux[0] + U[1] * ux[1] + U[2] * ux[2];
  const float tpu = tU[0] * ux[0] + tU[1] * ux[1] + tU[2] * ux[2] + U[0] * tux[0] + U[1] * tux[1] + U[2] * tux[2];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float perp = U[c] - pu * ux[c];
    const float tperp = tU[c] - tpu * ux[c] - pu * tux[c];
    acc[c] += perp * inv - D * ux[c];
    tacc[c] += tperp * inv - perp * inv * inv * tu.w - tD * ux[c] - D * tux[c];
  }
}

// block = (centre i, direction k), thread = feature (a, g); threads f < 16 NQ also carry the charge feature (channel, g).
//   P  = Sbar_i0 + u . Sbar_iv   (adjoint of the coefficient a_j gs of the pair seen from i)
//   P' = Sbar_j0 - u . Sbar_jv   (the same pair seen from j: u_ji = -u)
//   abar_i += gs P';  dbar_ij = a_j P gs', ubar_ij = gs a_j Sbar_iv;  dbar_ji = a_i P' gs', ubar_ji = gs a_i Sbar_jv
//   xbar_i += sum_m rbar_ji - rbar_ij
template <int NQ>
__global__ __launch_bounds__(256) void hvp_conv_bwd_kernel(bool need_abar, const float* __restrict__ a, const int* __restrict__ row_of,
                                                          const float* __restrict__ ta, const float* __restrict__ q,
                                                          const float* __restrict__ tq, const float* __restrict__ Sbar,
                                                          const float* __restrict__ tSbar, const float* __restrict__ Sqbar,
                                                          const float* __restrict__ tSqbar, const int* __restrict__ nb_idx,
                                                          const int* __restrict__ nb_cnt, const float4* __restrict__ pg, int cap,
                                                          const float* __restrict__ tv, BasisParams bp, float* __restrict__ abar,
                                                          float* __restrict__ tabar, float* __restrict__ qbar,
                                                          float* __restrict__ tqbar, float* __restrict__ xbar,
                                                          float* __restrict__ txbar, int N) {
  __shared__ Stage st;
  __shared__ float s_shift[16];
  __shared__ float sh[4];
  const int i = blockIdx.x, k = blockIdx.y, f = threadIdx.x, g = f & 15;
  if (f < 16) s_shift[f] = bp.shifts[f];
  const float* tvk = tv + (size_t)k * N * 3;
  const int cnt = nb_cnt[i];
  const size_t tr = (size_t)k * N + i;
  const size_t ri = row_of ? (size_t)min(63, max(0, row_of[i])) : (size_t)i;
  const float ai = a[ri * NF + f], tai = ta ? ta[tr * NF + f] : 0.0f;
  const float4 Sbi = reinterpret_cast<const float4*>(Sbar)[(size_t)i * NF + f];
  const float4 tSbi = reinterpret_cast<const float4*>(tSbar)[tr * NF + f];
  const bool qthr = NQ > 0 && f < NQ * 16;
  const int qc = f >> 4;
  float qi = 0.f, tqi = 0.f;
  float4 Sqi = make_float4(0, 0, 0, 0), tSqi = Sqi;
  if (qthr) {
    qi = q[(size_t)qc * N + i];
    tqi = tq[((size_t)k * NQ + qc) * N + i];
    Sqi = reinterpret_cast<const float4*>(Sqbar)[(size_t)i * (NQ * G_) + f];
    tSqi = reinterpret_cast<const float4*>(tSqbar)[tr * (NQ * G_) + f];
  }
  float ab = 0.f, tab = 0.f, qb = 0.f, tqb = 0.f;
  float xa[3] = {0, 0, 0}, txa[3] = {0, 0, 0};
  // one (coefficient, adjoint-moment) pair -> its share of the four pair adjoints and of abar-like sums
  auto pair_terms = [&](float cj, float tcj, float ci, float tci, float4 Si, float4 tSi, float4 Sj, float4 tSj, float4 u, float4 tu,
                        float gs, float dgs, float tgs, float tdgs, float& acc_b, float& tacc_b) __attribute__((always_inline)) {
    const float P = Si.x + u.x * Si.y + u.y * Si.z + u.z * Si.w;
    const float tP = tSi.x + tu.x * Si.y + tu.y * Si.z + tu.z * Si.w + u.x * tSi.y + u.y * tSi.z + u.z * tSi.w;
    const float Pp = Sj.x - (u.x * Sj.y + u.y * Sj.z + u.z * Sj.w);
    const float tPp = tSj.x - (tu.x * Sj.y + tu.y * Sj.z + tu.z * Sj.w) - (u.x * tSj.y + u.y * tSj.z + u.z * tSj.w);
    acc_b += gs * Pp;
    tacc_b += tgs * Pp + gs * tPp;
    // D = dbar_ij + dbar_ji = gs' (c_j P + c_i P'),  U = ubar_ji - ubar_ij = gs (c_i Sbar_jv - c_j Sbar_iv)
    const float w = cj * P + ci * Pp;
    const float tw = tcj * P + cj * tP + tci * Pp + ci * tPp;
    const float D = w * dgs, tD = tw * dgs + w * tdgs;
    const float v[3] = {ci * Sj.y - cj * Si.y, ci * Sj.z - cj * Si.z, ci * Sj.w - cj * Si.w};
    const float tvv[3] = {tci * Sj.y + ci * tSj.y - tcj * Si.y - cj * tSi.y, tci * Sj.z + ci * tSj.z - tcj * Si.z - cj * tSi.z,
                          tci * Sj.w + ci * tSj.w - tcj * Si.w - cj * tSi.w};
    const float U[3] = {gs * v[0], gs * v[1], gs * v[2]};
    const float tU[3] = {tgs * v[0] + gs * tvv[0], tgs * v[1] + gs * tvv[1], tgs * v[2] + gs * tvv[2]};
    rbar_pair_add(D, U, tD, tU, u, tu, xa, txa);
  };
  for (int m0 = 0; m0 < cnt; m0 += HCH) {
    stage_chunk(st, i, m0, cnt, nb_idx, pg, cap, tvk, bp);
    const float shift = s_shift[g];
    const int mc = min(HCH, cnt - m0);
#ifdef AIMNET_PROBE_HVP_PREFETCH  // measurement builds: the next neighbour's four rows requested ahead of this one's arithmetic
    auto rows = [&](int m, float& aj, float& taj, float4& Sbj, float4& tSbj) __attribute__((always_inline)) {
      const int j = st.j[m];
      const size_t rj = row_of ? (size_t)min(63, max(0, row_of[j])) : (size_t)j;
      const size_t trj = (size_t)k * N + j;
      aj = a[rj * NF + f];
      taj = ta ? ta[trj * NF + f] : 0.0f;
      Sbj = reinterpret_cast<const float4*>(Sbar)[(size_t)j * NF + f];
      tSbj = reinterpret_cast<const float4*>(tSbar)[trj * NF + f];
    };
    float aj_n = 0.f, taj_n = 0.f;
    float4 Sbj_n = make_float4(0, 0, 0, 0), tSbj_n = Sbj_n;
    if (mc > 0) rows(0, aj_n, taj_n, Sbj_n, tSbj_n);
#endif
    // (requesting the NEXT neighbour's four rows before this one's arithmetic was measured slower: 3.0 -> 4.5 ms per sweep on
    // 10 080 atoms - ten more live registers across ~250 instructions of product-rule terms)
    for (int m = 0; m < mc; ++m) {
      const float4 u = st.u[m], tu = st.tu[m], fc = st.fc[m];
      const int j = st.j[m];
      float gs, dgs, d2gs;
      basis_g2(bp.eta, shift, u.w, make_float3(fc.x, fc.y, fc.z), gs, dgs, d2gs);
      const float tgs = dgs * tu.w, tdgs = d2gs * tu.w;
      const size_t trj = (size_t)k * N + j;
#ifdef AIMNET_PROBE_HVP_PREFETCH
      const float aj = aj_n, taj = taj_n;
      const float4 Sbj = Sbj_n, tSbj = tSbj_n;
      if (m + 1 < mc) rows(m + 1, aj_n, taj_n, Sbj_n, tSbj_n);
#else
      const size_t rj = row_of ? (size_t)min(63, max(0, row_of[j])) : (size_t)j;
      const float aj = a[rj * NF + f], taj = ta ? ta[trj * NF + f] : 0.0f;
      const float4 Sbj = reinterpret_cast<const float4*>(Sbar)[(size_t)j * NF + f];
      const float4 tSbj = reinterpret_cast<const float4*>(tSbar)[trj * NF + f];
#endif
      pair_terms(aj, taj, ai, tai, Sbi, tSbi, Sbj, tSbj, u, tu, gs, dgs, tgs, tdgs, ab, tab);
      if (qthr) {
        const float qj = q[(size_t)qc * N + j], tqj = tq[((size_t)k * NQ + qc) * N + j];
        const float4 Sqj = reinterpret_cast<const float4*>(Sqbar)[(size_t)j * (NQ * G_) + f];
        const float4 tSqj = reinterpret_cast<const float4*>(tSqbar)[trj * (NQ * G_) + f];
        pair_terms(qj, tqj, qi, tqi, Sqi, tSqi, Sqj, tSqj, u, tu, gs, dgs, tgs, tdgs, qb, tqb);
      }
    }
  }
  if (need_abar) {
    if (k == 0) abar[(size_t)i * NF + f] += ab;
    tabar[tr * NF + f] += tab;
  }
  if (NQ > 0) {  // qbar_i[channel] += sum_g: the 16 lanes of a channel are one aligned group of a wave
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
      qb += __shfl_xor(qb, off, 64);
      tqb += __shfl_xor(tqb, off, 64);
    }
    if (qthr && g == 0) {
      if (k == 0) qbar[(size_t)qc * N + i] += qb;
      tqbar[((size_t)k * NQ + qc) * N + i] += tqb;
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = block_sum256(xa[c], sh), tvv = block_sum256(txa[c], sh);
    if (f == 0) {
      if (k == 0) xbar[(size_t)i * 3 + c] += v;
      txbar[tr * 3 + c] += tvv;
    }
  }
}

// ---- NSE adjoint + tangent: (qbar, abar) after the conv backward of pass p -> ybar of pass p-1's MLP ---------------------
// block = (molecule, direction).  ybar row = [qrbar (nq) | 2 f~ fbar (nq) | abar (256) | 0]
__global__ __launch_bounds__(256) void hvp_nse_bwd_kernel(const float* __restrict__ qbar, const float* __restrict__ tqbar,
                                                         const float* __restrict__ abar, const float* __restrict__ tabar,
                                                         const float* __restrict__ y, const float* __restrict__ ty, int ldy, int nq,
                                                         const float* __restrict__ Fm, const float* __restrict__ Dm,
                                                         const float* __restrict__ tFm, const float* __restrict__ tDm,
                                                         const int* __restrict__ mol_start, int n_mol, int N, int carry_q,
                                                         float* __restrict__ yb, float* __restrict__ tyb,
                                                         float* __restrict__ qbar_next, float* __restrict__ tqbar_next) {
  __shared__ float sh[4];
  const int m = blockIdx.x, k = blockIdx.y;
  const int i0 = mol_start[m], i1 = mol_start[m + 1];
  for (int ch = 0; ch < nq; ++ch) {
    const float F = Fm[(size_t)ch * n_mol + m], D = Dm[(size_t)ch * n_mol + m];
    const float tF = tFm[((size_t)k * nq + ch) * n_mol + m], tD = tDm[((size_t)k * nq + ch) * n_mol + m];
    float sw = 0.f, tsw = 0.f;
    for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
      const float ft = y[(size_t)i * ldy + nq + ch], tft = ty[((size_t)k * N + i) * ldy + nq + ch];
      const float f = ft * ft, tf = 2.0f * ft * tft;
      const float wl = f / F, twl = tf / F - f * tF / (F * F);
      const float qb = qbar[(size_t)ch * N + i], tqb = tqbar[((size_t)k * nq + ch) * N + i];
      sw += qb * wl;
      tsw += tqb * wl + qb * twl;
    }
    const float Wb = block_sum256(sw, sh), tWb = block_sum256(tsw, sh);
    const float r = D / F, trr = tD / F - D * tF / (F * F);
    for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
      const size_t trow = (size_t)k * N + i;
      const float ft = y[(size_t)i * ldy + nq + ch], tft = ty[trow * ldy + nq + ch];
      const float qr = qbar[(size_t)ch * N + i] - Wb, tqr = tqbar[((size_t)k * nq + ch) * N + i] - tWb;
      const float fb = r * qr, tfb = trr * qr + r * tqr;
      if (k == 0) {
        yb[(size_t)i * ldy + ch] = qr;
        yb[(size_t)i * ldy + nq + ch] = 2.0f * ft * fb;
        qbar_next[(size_t)ch * N + i] = carry_q ? qr : 0.0f;
      }
      tyb[trow * ldy + ch] = tqr;
      tyb[trow * ldy + nq + ch] = 2.0f * (tft * fb + ft * tfb);
      tqbar_next[((size_t)k * nq + ch) * N + i] = carry_q ? tqr : 0.0f;
    }
  }
}

// the feature columns of the same rows: ybar[:, 2 nq + f] = abar[:, f], zero padding beyond (all atoms in parallel - the NSE kernel
// above runs one block per molecule)
__global__ void hvp_ybar_a_kernel(const float* __restrict__ abar, const float* __restrict__ tabar, int ldy, int nq, int N, size_t n_t,
                                  float* __restrict__ yb, float* __restrict__ tyb) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // (row = k N + i, column)
  if (e >= n_t) return;
  const size_t row = e / ldy;
  const int c = (int)(e % ldy);
  if (c < 2 * nq) return;
  const int f = c - 2 * nq;
  tyb[e] = f < NF ? tabar[row * NF + f] : 0.0f;
  if (row < (size_t)N) yb[e] = f < NF ? abar[row * NF + f] : 0.0f;
}

// ---- external DFT-D3 block: central difference of ITS OWN analytic gradient --------------------------------------------
// The dispersion term is a smooth two-body sum with coordination-number dependent coefficients (d3.hip); its curvature is ~1 % of
// the model's (up to 0.6 eV/A^2 on config 4's molecule), its fp32 gradient noise ~1e-7 eV/A.  The coordination-number counting
// function 1 / (1 + exp(-16 (rcov / r - 1))) is steep - every derivative brings a factor ~16 / A - so a 2-point stencil at 0.01 A
// is 0.5 % off (measured 3e-3 eV/A^2); the 4-point stencil at h = 4e-3 A leaves (16 h)^4 / 30 ~ 6e-7 relative truncation, and the
// fp32 rounding of x +- h u (half an ulp of a 5 A coordinate over 4e-3 A) ~6e-5 relative: ~5e-5 eV/A^2 in all.  The reference
// does the same for the one block it cannot differentiate (the PME term, calculator.py:1777-1781).  All 4 K displaced copies
// are evaluated as ONE batch of independent systems: coordinates x + {h, -h, 2h, -2h} u_k, the neighbour rows of x re-based per
// copy (a pair that drifts across the cutoff sits where the S5 switch is zero anyway).
__global__ void hvp_d3_scale_kernel(const float* __restrict__ tv, int N, int K, float* __restrict__ scale) {
  __shared__ float sh[4];
  const int k = blockIdx.x;
  float m = 0.f;
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    const float* v = tv + ((size_t)k * N + i) * 3;
    m = fmaxf(m, v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) scale[k] = sqrtf(fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3])));  // largest per-atom |v_i|
}

__global__ void hvp_d3_replicate_kernel(const float* __restrict__ xw, const float* __restrict__ tv, const float* __restrict__ scale,
                                        const int* __restrict__ mol_idx, const int* __restrict__ aslot,
                                        const int* __restrict__ nb_idx, const int* __restrict__ nb_shift,
                                        const int* __restrict__ nb_cnt, int cap, int N, int K, float h, float* __restrict__ xd,
                                        int* __restrict__ mol_d, int* __restrict__ aslot_d, int* __restrict__ idx_d,
                                        int* __restrict__ shift_d, int* __restrict__ cnt_d) {
  const int i = blockIdx.x, c = blockIdx.y;  // copy c = 4 k + (0: +h, 1: -h, 2: +2h, 3: -2h)
  const int k = c >> 2;
  const float sg = ((c & 1) ? -h : h) * ((c & 2) ? 2.0f : 1.0f);
  const size_t row = (size_t)c * N + i;
  const float inv = 1.0f / fmaxf(scale[k], 1e-30f);
  if (threadIdx.x < 3) xd[row * 3 + threadIdx.x] = xw[(size_t)i * 3 + threadIdx.x] + sg * inv * tv[((size_t)k * N + i) * 3 + threadIdx.x];
  if (threadIdx.x == 0) {
    mol_d[row] = mol_idx[i];
    aslot_d[row] = aslot[i];
    cnt_d[row] = nb_cnt[i];
  }
  const int cnt = nb_cnt[i];
  for (int m = threadIdx.x; m < cnt; m += blockDim.x) {
    idx_d[row * cap + m] = nb_idx[(size_t)i * cap + m] + c * N;
    shift_d[row * cap + m] = nb_shift[(size_t)i * cap + m];
  }
}

__global__ void hvp_d3_combine_kernel(const float* __restrict__ gd, const float* __restrict__ scale, int N, size_t n_t, float h,
                                      float* __restrict__ txbar) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // (k, i, c)
  if (e >= n_t) return;
  const size_t per = (size_t)N * 3;
  const size_t k = e / per, r = e % per;
  const float* g4 = gd + 4 * k * per + r;
  txbar[e] += (8.0f * (g4[0] - g4[per]) - (g4[2 * per] - g4[3 * per])) * (scale[k] / (12.0f * h));
}

__global__ void hvp_out_kernel(const float* __restrict__ xbar, size_t n, float* __restrict__ forces) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) forces[e] = -xbar[e];
}

// ---- workspace --------------------------------------------------------------------------------------------------------
struct Carve {
  char* base;
  size_t off = 0;
  template <typename T>
  T* take(size_t n) {
    off = align_up(off, 256);
    T* p = base ? (T*)(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
};

struct HvpWs {
  NlistBuffers nl;
  int *nb_idx, *nb_shift, *nb_cnt, *lr_idx, *lr_shift, *lr_cnt;
  float4* pg;
  // primal [N][..] and tangent [K][N][..] twins
  float *a[AIMNET_MAX_PASS], *ta[AIMNET_MAX_PASS];
  float *q[AIMNET_MAX_PASS], *tq[AIMNET_MAX_PASS];
  float *x[AIMNET_MAX_PASS], *tx[AIMNET_MAX_PASS];
  float *V[AIMNET_MAX_PASS], *tV[AIMNET_MAX_PASS], *Vq[AIMNET_MAX_PASS], *tVq[AIMNET_MAX_PASS];
  float *z[AIMNET_MAX_PASS][AIMNET_MAX_LAYERS], *tz[AIMNET_MAX_PASS][AIMNET_MAX_LAYERS];
  float *y[AIMNET_MAX_PASS], *ty[AIMNET_MAX_PASS];  // output row of pass p's MLP (aliases z when it ends linear)
  float *Fm[AIMNET_MAX_PASS], *Dm[AIMNET_MAX_PASS], *tFm[AIMNET_MAX_PASS], *tDm[AIMNET_MAX_PASS];
  float *hz[AIMNET_MAX_LAYERS], *thz[AIMNET_MAX_LAYERS];
  float *h[2], *g[2];    // activations between GEMMs / adjoint ping-pong (stacked: tangent rows follow the N primal rows)
  float *Sbar, *tSbar, *Sqbar, *tSqbar, *abar, *tabar, *qbar[2], *tqbar[2], *xbar, *txbar;
  float* wlast;
  // external DFT-D3 block (4 K displaced copies as one batch)
  int *d3_idx, *d3_shift, *d3_cnt, *aslot, *d3c_idx, *d3c_shift, *d3c_cnt, *d3c_mol, *d3c_aslot;
  unsigned long long* present_part;
  float *d3c_x, *d3c_w, *d3c_dEdcn, *d3c_g, *d3_scale;
  float4* d3c_xs;
  double* d3c_e;
  size_t total;
};

int hvp_max_width(const aimnet_engine* e) {
  int w = 32;
  for (int p = 0; p < e->arch.n_pass; ++p)
    for (const Layer& L : e->mlp[p]) w = std::max(w, std::max(L.k_in, L.k_out));
  for (const Layer& L : e->head) w = std::max(w, std::max(L.k_in, L.k_out));
  return w;
}

void hvp_layout(const aimnet_engine* e, int N, int n_mol, int K, const aimnet_eval_options* opt, char* base, HvpWs& W) {
  Carve c{base};
  const size_t n = (size_t)N, kn = (size_t)K * N;
  const int np = e->arch.n_pass, nq = e->nq;
  const int cap = std::max(1, opt->max_nb), cap_lr = std::max(0, opt->max_nb_lr);
  char* nl_base = c.take<char>(nlist_scratch_bytes(N, n_mol));
  if (base) nlist_carve(W.nl, nl_base, N, n_mol);
  W.nb_idx = c.take<int>(n * cap);
  W.nb_shift = c.take<int>(n * cap);
  W.nb_cnt = c.take<int>(n);
  W.lr_idx = c.take<int>(n * cap_lr);
  W.lr_shift = c.take<int>(n * cap_lr);
  W.lr_cnt = c.take<int>(n);
  W.pg = c.take<float4>(n * cap);
  const int mw = hvp_max_width(e);
  for (int p = 0; p < np; ++p) {
    W.a[p] = p == 0 ? nullptr : c.take<float>(n * NF);
    W.ta[p] = p == 0 ? nullptr : c.take<float>(kn * NF);
    W.q[p] = c.take<float>(n * nq);
    W.tq[p] = c.take<float>(kn * nq);
    const int ldx = e->mlp[p][0].k_in;
    W.x[p] = c.take<float>((n + kn) * ldx);  // rows that go through a GEMM: primal rows on top of the tangent rows, one launch
    W.tx[p] = W.x[p] ? W.x[p] + n * ldx : nullptr;
    W.V[p] = c.take<float>(n * NV * 3);
    W.tV[p] = c.take<float>(kn * NV * 3);
    W.Vq[p] = c.take<float>(n * nq * H_ * 3);
    W.tVq[p] = c.take<float>(kn * nq * H_ * 3);
    const int nl = (int)e->mlp[p].size();
    for (int l = 0; l < nl; ++l) {
      W.z[p][l] = c.take<float>((n + kn) * e->mlp[p][l].k_out);
      W.tz[p][l] = W.z[p][l] ? W.z[p][l] + n * e->mlp[p][l].k_out : nullptr;
    }
    if (e->arch.last_linear[p]) {
      W.y[p] = W.z[p][nl - 1];
      W.ty[p] = W.tz[p][nl - 1];
    } else {
      W.y[p] = c.take<float>((n + kn) * e->mlp[p][nl - 1].k_out);
      W.ty[p] = W.y[p] ? W.y[p] + n * e->mlp[p][nl - 1].k_out : nullptr;
    }
    W.Fm[p] = c.take<float>((size_t)n_mol * nq);
    W.Dm[p] = c.take<float>((size_t)n_mol * nq);
    W.tFm[p] = c.take<float>((size_t)K * n_mol * nq);
    W.tDm[p] = c.take<float>((size_t)K * n_mol * nq);
  }
  for (size_t l = 0; l + 1 < e->head.size(); ++l) {
    W.hz[l] = c.take<float>((n + kn) * e->head[l].k_out);
    W.thz[l] = W.hz[l] ? W.hz[l] + n * e->head[l].k_out : nullptr;
  }
  for (int b = 0; b < 2; ++b) {
    W.h[b] = c.take<float>((n + kn) * mw);  // stacked like x / z; the tangent rows start N * (row width in use) floats in
    W.g[b] = c.take<float>((n + kn) * mw);
    W.qbar[b] = c.take<float>(n * nq);
    W.tqbar[b] = c.take<float>(kn * nq);
  }
  W.Sbar = c.take<float>(n * NF * 4);
  W.tSbar = c.take<float>(kn * NF * 4);
  W.Sqbar = c.take<float>(n * nq * G_ * 4);
  W.tSqbar = c.take<float>(kn * nq * G_ * 4);
  W.abar = c.take<float>(n * NF);
  W.tabar = c.take<float>(kn * NF);
  W.xbar = c.take<float>(n * 3);
  W.txbar = c.take<float>(kn * 3);
  W.wlast = c.take<float>((size_t)mw);
  if (opt->dftd3 != 0) {
    const int cap_d3 = std::max(1, opt->max_nb_d3);
    const size_t cn = 4 * kn;
    W.d3_idx = c.take<int>(n * cap_d3);
    W.d3_shift = c.take<int>(n * cap_d3);
    W.d3_cnt = c.take<int>(n);
    W.aslot = c.take<int>(n);
    W.present_part = c.take<unsigned long long>((n + 255) / 256);
    W.d3c_idx = c.take<int>(cn * cap_d3);
    W.d3c_shift = c.take<int>(cn * cap_d3);
    W.d3c_cnt = c.take<int>(cn);
    W.d3c_mol = c.take<int>(cn);
    W.d3c_aslot = c.take<int>(cn);
    W.d3c_x = c.take<float>(cn * 3);
    W.d3c_w = c.take<float>(cn * 12);
    W.d3c_dEdcn = c.take<float>(cn);
    W.d3c_g = c.take<float>(cn * 3);
    W.d3c_xs = c.take<float4>(cn);
    W.d3c_e = c.take<double>(cn);
    W.d3_scale = c.take<float>((size_t)K);
  }
  W.total = align_up(c.off, 256);
}

#define RC(call)         \
  do {                   \
    int _rc = (call);    \
    if (_rc) return _rc; \
  } while (0)

inline dim3 grid1(size_t n) { return dim3((unsigned)((n + 255) / 256)); }

// forward of one MLP (pass MLP or energy head) on the stacked rows [N primal | K N tangent]: one bias-free GEMM per layer
// (a second, biased one for the primal rows of a layer that ends the MLP linearly), then the activation kernel.
// `hin` = stacked input rows, `y` = stacked buffer for the output rows of an MLP that ends with GELU (may be NULL: discarded)
int mlp_forward(const aimnet_engine* e, hipStream_t s, const std::vector<Layer>& Ls, int n_layers, bool last_linear, const float* hin,
                int N, int K, float* const* z, float* y, HvpWs& W) {
  const size_t kn = (size_t)K * N;
  const int M = (int)(kn + N);
  int ld_in = Ls[0].k_in;
  for (int l = 0; l < n_layers; ++l) {
    const Layer& L = Ls[l];
    const size_t toff = (size_t)N * L.k_out;
    if (l == n_layers - 1 && last_linear) {  // y = z + b is read as it is: biased primal rows, bias-free tangent rows
      RC(mlp_gemm(e, s, EPI_BIAS, hin, ld_in, L, true, 0, 0, N, L.k_out, L.k_in, L.b, z[l], nullptr, L.k_out));
      RC(mlp_gemm(e, s, EPI_NONE, hin + (size_t)N * ld_in, ld_in, L, true, 0, 0, (int)kn, L.k_out, L.k_in, nullptr, z[l] + toff, nullptr,
                  L.k_out));
      break;
    }
    RC(mlp_gemm(e, s, EPI_NONE, hin, ld_in, L, true, 0, 0, M, L.k_out, L.k_in, nullptr, z[l], nullptr, L.k_out));
    float* h = (l == n_layers - 1 && y) ? y : W.h[l & 1];
    const size_t n_t = kn * L.k_out;
    hipLaunchKernelGGL(hvp_act_fwd_kernel, grid1(n_t), dim3(256), 0, s, z[l], z[l] + toff, L.b, L.k_out, N, n_t, h, h + toff);
    AIMNET_LAUNCH_CHECK();
    hin = h;
    ld_in = L.k_out;
  }
  return 0;
}

// backward of one MLP: (g, tg) = adjoint of its OUTPUT rows (after the last activation) -> adjoint of its input rows.
// The rows live in the stacked ping-pong buffers W.g[b] (tangent rows behind the N primal rows); *src = index of the pair holding the input, or -1 when the input
// is the external broadcast row `g_ext` with a zero tangent (the energy head's last layer).  Every step reads pair src and writes
// the other one; on return *src names the pair holding the result.
int mlp_backward(const aimnet_engine* e, hipStream_t s, const std::vector<Layer>& Ls, int n_layers, bool last_linear,
                 const float* g_ext, int N, int K, float* const* z, HvpWs& W, int* src) {
  const size_t kn = (size_t)K * N;
  const int M = (int)(kn + N);
  int cur = *src;
  for (int l = n_layers - 1; l >= 0; --l) {
    const Layer& L = Ls[l];
    const size_t toff = (size_t)N * L.k_out;
    if (!(l == n_layers - 1 && last_linear)) {
      const int dst = cur == 0 ? 1 : 0;
      const size_t n_t = kn * L.k_out;
      hipLaunchKernelGGL(hvp_act_bwd_kernel, grid1(n_t), dim3(256), 0, s, cur < 0 ? g_ext : W.g[cur], cur < 0 ? 1 : 0,
                         cur < 0 ? nullptr : W.g[cur] + toff, z[l], z[l] + toff, L.b, L.k_out, N, n_t, W.g[dst], W.g[dst] + toff);
      AIMNET_LAUNCH_CHECK();
      cur = dst;
    }
    if (cur < 0) {
      set_last_error("hvp: an MLP that ends linear cannot start from the broadcast row");
      return AIMNET_E_INVALID;
    }
    const int dst = cur ^ 1;  // one launch for the primal and the tangent rows (W.g is stacked)
    RC(mlp_gemm(e, s, EPI_NONE, W.g[cur], L.k_out, L, false, 0, 0, M, L.k_in, L.k_out, nullptr, W.g[dst], nullptr, L.k_in));
    cur = dst;
  }
  *src = cur;
  return 0;
}

}  // namespace
}  // namespace aimnet

using namespace aimnet;

extern "C" {

size_t aimnet_engine_hvp_workspace_bytes(const aimnet_engine* e, int32_t n_atoms, int32_t n_mol, int32_t n_vec,
                                         const aimnet_eval_options* opt) {
  if (!e || !opt || n_atoms <= 0 || n_mol <= 0 || n_vec <= 0) return 0;
  HvpWs W;
  hvp_layout(e, n_atoms, n_mol, n_vec, opt, nullptr, W);
  return W.total;
}

int aimnet_engine_hvp(aimnet_engine* e, const aimnet_inputs* in, const aimnet_eval_options* opt, const float* vectors,
                      int32_t n_vec, float* hv, float* forces, int32_t* status, void* workspace, size_t workspace_bytes,
                      void* hip_stream) {
  if (!e || !in || !opt || !vectors || !hv || !status || !workspace || n_vec <= 0) return AIMNET_E_INVALID;
  const int N = in->n_atoms, n_mol = in->n_mol, K = n_vec;
  if (N <= 0 || n_mol <= 0 || !in->coord || !in->numbers || !in->mol_idx || !in->charge) {
    set_last_error("hvp: null or empty input");
    return AIMNET_E_INVALID;
  }
  if (in->nbmat || in->nbmat_lr || in->nbmat_d3) {
    set_last_error("hvp: caller-supplied neighbour matrices are not read by the tangent sweep (it builds its own lists)");
    return AIMNET_E_INVALID;
  }
  const bool d3 = opt->dftd3 != 0;
  if (d3 && e->d3.ns == 0) {
    set_last_error("hvp: DFT-D3 requested but aimnet_engine_set_dftd3 was never called");
    return AIMNET_E_INVALID;
  }
  // the tangent kernels index the directions with grid.y (HIP limit 65535); the D3 block replicates every direction four times
  if (n_vec > (d3 ? 16383 : 65535)) {
    set_last_error("hvp: at most %d directions per sweep (%d given): split them over several calls", d3 ? 16383 : 65535, n_vec);
    return AIMNET_E_INVALID;
  }
  if (d3 && 4 * (size_t)n_vec * (size_t)in->n_atoms * (size_t)std::max(1, opt->max_nb_d3) >= (size_t)INT32_MAX) {
    set_last_error("hvp: n_vec * n_atoms * max_nb_d3 too large for one sweep (split the directions)");
    return AIMNET_E_INVALID;
  }
  const bool pbc = in->cell != nullptr;
  const int coulomb = opt->coulomb;
  if (coulomb == AIMNET_COULOMB_DSF && opt->max_nb_lr <= 0) {
    set_last_error("hvp: DSF Coulomb needs max_nb_lr > 0 (the tangent sweep runs on the neighbour list, also for periodic input)");
    return AIMNET_E_INVALID;
  }
  if (coulomb == AIMNET_COULOMB_EWALD || coulomb == AIMNET_COULOMB_PME) {
    set_last_error("hvp: the analytic tangent sweep does not cover Ewald summation (use DSF, or differences of forces as the reference "
                   "does for its PME block, lr.py:903-926)");
    return AIMNET_E_INVALID;
  }
  if (coulomb == AIMNET_COULOMB_SIMPLE && pbc) {
    set_last_error("hvp: 'simple' Coulomb is undefined for periodic input");
    return AIMNET_E_INVALID;
  }
  if (pbc && !(in->n_cell == 1 || in->n_cell == n_mol)) {
    set_last_error("hvp: n_cell must be 1 or n_mol");
    return AIMNET_E_INVALID;
  }
  if ((size_t)K * (size_t)N * (size_t)hvp_max_width(e) >= (size_t)INT32_MAX) {
    set_last_error("hvp: n_vec * n_atoms too large for one sweep (split the directions)");
    return AIMNET_E_INVALID;
  }
  hipStream_t s = (hipStream_t)hip_stream;
  AIMNET_HIP_CHECK(hipSetDevice(e->device));
  HvpWs W;
  hvp_layout(e, N, n_mol, K, opt, (char*)workspace, W);
  if (W.total > workspace_bytes) {
    set_last_error("hvp: workspace too small (%zu < %zu)", workspace_bytes, W.total);
    return AIMNET_E_WORKSPACE;
  }
  const aimnet_arch& ar = e->arch;
  const int np = ar.n_pass, nq = e->nq;
  const int cap = std::max(1, opt->max_nb), cap_lr = std::max(0, opt->max_nb_lr);
  const int n_cell = pbc ? in->n_cell : 0;
  const size_t kn = (size_t)K * N;

  // ---- lists + pair geometry: the same builders as aimnet_engine_eval ----
  AIMNET_HIP_CHECK(hipMemsetAsync(status, 0, 8 * sizeof(int), s));
  RC(launch_mol_start(s, in->mol_idx, N, n_mol, W.nl.mol_start, W.nl.mol_c, in->numbers, status + 6, d3 ? e->slot_of_z : nullptr,
                      d3 ? W.aslot : nullptr, d3 ? W.present_part : nullptr));
  const int* mol_c = W.nl.mol_c;
  RC(launch_wrap(s, in->coord, mol_c, N, n_mol, in->cell, n_cell, in->pbc, W.nl, in->pbc_sys, pbc ? ar.rc : 0.0f));
  if (!pbc && (long)N >= 1500L * n_mol) RC(launch_bbox(s, n_mol, W.nl));
  RC(launch_nlist(s, N, n_mol, mol_c, in->cell, n_cell, in->pbc, ar.rc, ar.rc, cap, N, 0, W.nl, W.nb_idx, W.nb_shift, W.nb_cnt,
                  status + 0, status + 2, W.pg));
  if (coulomb == AIMNET_COULOMB_DSF)
    RC(launch_nlist(s, N, n_mol, mol_c, in->cell, n_cell, in->pbc, opt->dsf_rc, -1.0f, cap_lr, N, 0, W.nl, W.lr_idx, W.lr_shift,
                    W.lr_cnt, status + 1, status + 3));

  // ---- forward + tangent ----
  const dim3 gik(N, K), gmk(n_mol, K), gwk(ceil_div(N, 4), K), b256(256);
  for (int p = 0; p < np; ++p) {
    const std::vector<Layer>& Ls = e->mlp[p];
    const int nl = (int)Ls.size(), ldx = Ls[0].k_in;
    AIMNET_HIP_CHECK(hipMemsetAsync(W.x[p], 0, (size_t)N * ldx * sizeof(float), s));    // padding columns
    AIMNET_HIP_CHECK(hipMemsetAsync(W.tx[p], 0, kn * ldx * sizeof(float), s));
    const float* a_p = p == 0 ? e->afv : W.a[p];
    const int* row_of = p == 0 ? in->numbers : nullptr;
    const float *q_p = p > 0 ? W.q[p - 1] : nullptr, *tq_p = p > 0 ? W.tq[p - 1] : nullptr;
#define HVP_CONV_FWD(NQV)                                                                                                       \
  hipLaunchKernelGGL((hvp_conv_fwd_kernel<NQV>), gik, b256, 0, s, a_p, row_of, W.ta[p], q_p, tq_p, W.nb_idx, W.nb_cnt, W.pg, cap, \
                     vectors, e->agh_a, e->agh_q, e->bp, W.x[p], W.tx[p], ldx, W.V[p], W.tV[p], W.Vq[p], W.tVq[p], N)
    if (p == 0) HVP_CONV_FWD(0);
    else if (nq == 1) HVP_CONV_FWD(1);
    else HVP_CONV_FWD(2);
#undef HVP_CONV_FWD
    AIMNET_LAUNCH_CHECK();
    RC(mlp_forward(e, s, Ls, nl, ar.last_linear[p] != 0, W.x[p], N, K, W.z[p], W.y[p], W));
    if (p < np - 1) {
      hipLaunchKernelGGL(hvp_nse_fwd_kernel, gmk, b256, 0, s, W.y[p], W.ty[p], Ls[nl - 1].k_out, nq, q_p, tq_p, W.nl.mol_start,
                         in->charge, n_mol, N, W.q[p], W.tq[p], W.Fm[p], W.Dm[p], W.tFm[p], W.tDm[p]);
      AIMNET_LAUNCH_CHECK();
      const size_t n_t = kn * NF;
      hipLaunchKernelGGL(hvp_update_a_kernel, grid1(n_t), b256, 0, s, a_p, row_of, W.ta[p], W.y[p], W.ty[p], Ls[nl - 1].k_out,
                         2 * nq, N, n_t, W.a[p + 1], W.ta[p + 1]);
      AIMNET_LAUNCH_CHECK();
    }
  }
  const int nh = (int)e->head.size();
  {
    RC(mlp_forward(e, s, e->head, nh - 1, false, W.y[np - 1], N, K, W.hz, nullptr, W));
  }

  // ---- Coulomb seeds (+ tangents) of qbar / xbar ----
  CoulombParams cp;
  cp.factor = (float)(0.5 * 27.211386024367243 * 0.5291772105638411);
  cp.sr_rc = ar.sr_rc;
  cp.sr_envelope = ar.sr_envelope;
  cp.dsf_rc = opt->dsf_rc;
  cp.dsf_alpha = opt->dsf_alpha;
  int qb = 0;  // index of the live qbar buffers
  const float *q_fin = W.q[np - 2], *tq_fin = W.tq[np - 2];
  hipLaunchKernelGGL(hvp_coulomb_sr_kernel, gwk, b256, 0, s, ar.sr_coulomb != 0, q_fin, tq_fin, nq, W.nb_idx, W.nb_cnt, W.pg, cap,
                     vectors, cp, N, W.qbar[qb], W.tqbar[qb], W.xbar, W.txbar);
  AIMNET_LAUNCH_CHECK();
  if (coulomb == AIMNET_COULOMB_SIMPLE) {
    hipLaunchKernelGGL(hvp_coulomb_lr_kernel<false>, gwk, b256, 0, s, q_fin, tq_fin, nq, W.nl.xw, mol_c, W.nl.mol_start, in->cell,
                       n_cell, W.lr_idx, W.lr_shift, W.lr_cnt, cap_lr, vectors, cp, N, W.qbar[qb], W.tqbar[qb], W.xbar, W.txbar);
    AIMNET_LAUNCH_CHECK();
  } else if (coulomb == AIMNET_COULOMB_DSF) {
    hipLaunchKernelGGL(hvp_coulomb_lr_kernel<true>, gwk, b256, 0, s, q_fin, tq_fin, nq, W.nl.xw, mol_c, W.nl.mol_start, in->cell,
                       n_cell, W.lr_idx, W.lr_shift, W.lr_cnt, cap_lr, vectors, cp, N, W.qbar[qb], W.tqbar[qb], W.xbar, W.txbar);
    AIMNET_LAUNCH_CHECK();
  }

  // ---- backward + tangent ----
  AIMNET_HIP_CHECK(hipMemsetAsync(W.abar, 0, (size_t)N * NF * sizeof(float), s));
  AIMNET_HIP_CHECK(hipMemsetAsync(W.tabar, 0, kn * NF * sizeof(float), s));
  int cur = -1;
  {
    const Layer& Ll = e->head[nh - 1];
    const int ld = e->head[nh - 2].k_out;
    hipLaunchKernelGGL(hvp_pad_row_kernel, dim3(ceil_div(ld, 256)), b256, 0, s, e->head_w_last, Ll.n_in, ld, W.wlast);
    AIMNET_LAUNCH_CHECK();
    RC(mlp_backward(e, s, e->head, nh - 1, false, W.wlast, N, K, W.hz, W, &cur));
  }
  for (int p = np - 1; p >= 0; --p) {
    const std::vector<Layer>& Ls = e->mlp[p];
    const int nl = (int)Ls.size(), ldx = Ls[0].k_in;
    RC(mlp_backward(e, s, Ls, nl, ar.last_linear[p] != 0, nullptr, N, K, W.z[p], W, &cur));
    const float *xb = W.g[cur], *txb = W.g[cur] + (size_t)N * ldx;
    const float* a_p = p == 0 ? e->afv : W.a[p];
    const int* row_of = p == 0 ? in->numbers : nullptr;
    const float *q_p = p > 0 ? W.q[p - 1] : nullptr, *tq_p = p > 0 ? W.tq[p - 1] : nullptr;
    float* abar_u = p > 0 ? W.abar : nullptr;  // a^0 is the constant embedding: its adjoint is not needed
#define HVP_UNCONCAT(NQV)                                                                                                         \
  hipLaunchKernelGGL((hvp_unconcat_kernel<NQV>), gik, b256, 0, s, xb, txb, ldx, W.V[p], W.tV[p], W.Vq[p], W.tVq[p], e->agh_a,     \
                     e->agh_q, abar_u, W.tabar, W.qbar[qb], W.tqbar[qb], W.Sbar, W.tSbar, W.Sqbar, W.tSqbar, N)
#define HVP_CONV_BWD(NQV)                                                                                                         \
  hipLaunchKernelGGL((hvp_conv_bwd_kernel<NQV>), gik, b256, 0, s, p > 0, a_p, row_of, W.ta[p], q_p, tq_p, W.Sbar, W.tSbar, W.Sqbar, \
                     W.tSqbar, W.nb_idx, W.nb_cnt, W.pg, cap, vectors, e->bp, W.abar, W.tabar, W.qbar[qb], W.tqbar[qb], W.xbar,  \
                     W.txbar, N)
    if (p == 0) HVP_UNCONCAT(0);
    else if (nq == 1) HVP_UNCONCAT(1);
    else HVP_UNCONCAT(2);
    AIMNET_LAUNCH_CHECK();
    if (p == 0) HVP_CONV_BWD(0);
    else if (nq == 1) HVP_CONV_BWD(1);
    else HVP_CONV_BWD(2);
    AIMNET_LAUNCH_CHECK();
#undef HVP_UNCONCAT
#undef HVP_CONV_BWD
    if (p == 0) break;
    const std::vector<Layer>& Lq = e->mlp[p - 1];
    const int ldy = Lq[Lq.size() - 1].k_out;
    cur ^= 1;  // xb has been consumed: the adjoint rows of pass p-1's MLP output go into the other ping-pong pair
    hipLaunchKernelGGL(hvp_nse_bwd_kernel, gmk, b256, 0, s, W.qbar[qb], W.tqbar[qb], W.abar, W.tabar, W.y[p - 1], W.ty[p - 1], ldy, nq,
                       W.Fm[p - 1], W.Dm[p - 1], W.tFm[p - 1], W.tDm[p - 1], W.nl.mol_start, n_mol, N, p - 1 > 0 ? 1 : 0, W.g[cur],
                       W.g[cur] + (size_t)N * ldy, W.qbar[qb ^ 1], W.tqbar[qb ^ 1]);
    AIMNET_LAUNCH_CHECK();
    hipLaunchKernelGGL(hvp_ybar_a_kernel, grid1(kn * ldy), b256, 0, s, W.abar, W.tabar, ldy, nq, N, kn * ldy, W.g[cur], W.g[cur] + (size_t)N * ldy);
    AIMNET_LAUNCH_CHECK();
    qb ^= 1;
  }
  if (d3) {  // the dispersion block: central difference of the D3 gradient over all directions in one batch (see the kernels)
    const int cap_d3 = std::max(1, opt->max_nb_d3);
    const float h = 4e-3f;
    const int NC = 4 * K * N;
    RC(launch_nlist(s, N, n_mol, mol_c, in->cell, n_cell, in->pbc, opt->d3_cutoff, -1.0f, cap_d3, N, 0, W.nl, W.d3_idx, W.d3_shift,
                    W.d3_cnt, status + 4, status + 5));
    hipLaunchKernelGGL(hvp_d3_scale_kernel, dim3(K), b256, 0, s, vectors, N, K, W.d3_scale);
    AIMNET_LAUNCH_CHECK();
    hipLaunchKernelGGL(hvp_d3_replicate_kernel, dim3(N, 4 * K), dim3(64), 0, s, W.nl.xw, vectors, W.d3_scale, mol_c, W.aslot,
                       W.d3_idx, W.d3_shift, W.d3_cnt, cap_d3, N, K, h, W.d3c_x, W.d3c_mol, W.d3c_aslot, W.d3c_idx, W.d3c_shift,
                       W.d3c_cnt);
    AIMNET_LAUNCH_CHECK();
    AIMNET_HIP_CHECK(hipMemsetAsync(W.d3c_g, 0, (size_t)NC * 3 * sizeof(float), s));
    AIMNET_HIP_CHECK(hipMemsetAsync(W.d3c_e, 0, (size_t)NC * sizeof(double), s));
    D3Params dp;
    dp.s6 = opt->d3_s6; dp.s8 = opt->d3_s8; dp.a1 = opt->d3_a1; dp.a2 = opt->d3_a2;
    dp.r_on = opt->d3_smoothing_on * 1.8897261258369282f;
    dp.r_off = opt->d3_cutoff * 1.8897261258369282f;
    RC(launch_dftd3(s, true, false, W.d3c_x, W.d3c_mol, in->cell, n_cell, W.d3c_aslot, W.d3c_idx, W.d3c_shift, W.d3c_cnt, cap_d3,
                    e->d3, dp, opt->d3_cutoff, NC, W.d3c_xs, W.d3c_w, W.d3c_dEdcn, W.d3c_e, W.d3c_g, nullptr, false, cp, nullptr,
                    nullptr));
    hipLaunchKernelGGL(hvp_d3_combine_kernel, grid1(kn * 3), b256, 0, s, W.d3c_g, W.d3_scale, N, kn * 3, h, W.txbar);
    AIMNET_LAUNCH_CHECK();
    if (forces) {  // the forces of the sweep include the dispersion term: its gradient at x itself (one more evaluation)
      AIMNET_HIP_CHECK(hipMemsetAsync(W.d3c_e, 0, (size_t)N * sizeof(double), s));
      RC(launch_dftd3(s, true, false, W.nl.xw, mol_c, in->cell, n_cell, W.aslot, W.d3_idx, W.d3_shift, W.d3_cnt, cap_d3, e->d3, dp,
                      opt->d3_cutoff, N, W.d3c_xs, W.d3c_w, W.d3c_dEdcn, W.d3c_e, W.xbar, nullptr, false, cp, nullptr, nullptr));
    }
  }
  AIMNET_HIP_CHECK(hipMemcpyAsync(hv, W.txbar, kn * 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (forces) {
    hipLaunchKernelGGL(hvp_out_kernel, grid1((size_t)N * 3), b256, 0, s, W.xbar, (size_t)N * 3, forces);
    AIMNET_LAUNCH_CHECK();
  }
  return AIMNET_OK;
}

}  // extern "C"
