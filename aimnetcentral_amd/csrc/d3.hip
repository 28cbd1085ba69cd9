// d3.hip - DFT-D3(BJ) two-body dispersion: energy, forces, virial.
//
// Reference semantics (paths relative to /root/reference/aimnet): modules/lr.py DFTD3 (:1335-1820), whose GPU
// path is the third-party nvalchemiops `dftd3` kernel and whose in-tree torch twin - the arithmetic restated
// here and in oracle/aimnet2_oracle.py::dftd3_energy - is
//   _calc_torch_coord_num :1595   cn_i = sum_j sigmoid(16 ((rcov_i + rcov_j) / d_ij - 1))        (d in Bohr)
//   _calc_torch_c6ij      :1605   C6_ij = sum_ab c6ref_ab W_ab / sum_ab W_ab over the 5x5 reference systems,
//                                 W_ab = exp(-4 ((cn_i - cnref_i[a])^2 + (cn_j - cnref_j[b])^2) - max), dropped below e^-12
//   _compute_energy_torch :1626   e_ij = -C6_ij (s6 / (d^6 + R0^6) + s8 3 r4r2_i r4r2_j / (d^8 + R0^8)) S5(d),
//                                 R0 = a1 sqrt(3 r4r2_i r4r2_j) + a2,  E = Hartree/2 * sum over ORDERED pairs
//   _s5_switch_torch      :1580   quintic switch between cutoff (1 - smoothing_fraction) and cutoff.
//
// What is different from the reference arithmetic, and why it is the same number:
//   * the reference tables store cnref_i[a] redundantly per partner species and reference b; they factorise
//     (checked at upload, engine.hip), so exp(-4 dcn_i^2 - 4 dcn_j^2) = w_i[a] w_j[b] with FIVE exponentials per
//     atom (d3_cn_kernel) instead of 25 per pair; the e^-12 drop rule is applied per (a, b) on the sum of the two
//     shifted exponents exactly as the reference does;
//   * forces are analytic (the reference kernel returns them too): the C6(cn) dependence goes through
//     dE/dcn_i accumulated in the pair pass and is pushed to positions by a third, cheap pass over the same list.
//
// Three passes over the D3 neighbour list (wave per atom, lanes over list slots), HBM-bound on the 8 B list entries
// with L2-resident gathers of 48 B per neighbour:
//   d3_cn_kernel       cn_i, then the 5 shifted exponents / weights of atom i
//   d3_pair_kernel     C6_ij, pair energy (fp64 sums), direct pair force + virial, dE/dcn_i
//   d3_cnforce_kernel  (dE/dcn_i + dE/dcn_j) dsigma/dd along u_ij  -> force + virial
#include "common.h"
#include "kernels.h"

namespace aimnet {

constexpr float BOHR_INV_F = 1.8897261258369282f;   // 1 / 0.5291772105638411 (constants.py:8-9)
constexpr float HALF_HARTREE_F = 13.605693012183622f;
constexpr int D3W = 12;  // floats per atom in the weight table: s[5], w[5], cn, pad
constexpr int D3_SLOTS = 4;  // list slots per loop trip in the two light passes (cn, cnforce)

// The D3 passes are VALU-bound (112 / 397 / 147 instructions per list slot measured with IEEE division, sqrtf and expf):
// single-instruction reciprocal, square root and exp2 (1 ulp each) are ample for a term whose parity gate is 6e-6 eV on 7 eV.
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

// PairAcc / pair_add / pair_store live in model.hip; the same accumulation contract is restated here
struct D3Acc {
  double e = 0.0;   // dispersion pair energies (Hartree)
  double ec = 0.0;  // DSF Coulomb pair terms w q_i q_j when the two share one pass
  float f0 = 0.f, f1 = 0.f, f2 = 0.f;
  float W[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
};

template <bool STRESS>
__device__ __forceinline__ void d3_add(D3Acc& A, float t, float ux, float uy, float uz, float d) {
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

// E = k sum_ordered e  ->  dE/dx_i = -2k sum_j e' u_ij ,  dE/deps_ab = k sum_ordered e' r_a u_b   (as coulomb pair_store)
template <bool GRAD, bool STRESS>
__device__ __forceinline__ void d3_store(D3Acc& A, int i, int lane, float k, double* ecoul, float* fgrad, float* virial_atom,
                                         double kc = 0.0) {
  const double e = wave_sum(A.e) + (kc != 0.0 ? kc / (double)k * wave_sum(A.ec) : 0.0);
  float f0 = 0.f, f1 = 0.f, f2 = 0.f;
  if (GRAD) {
    f0 = wave_sum(A.f0);
    f1 = wave_sum(A.f1);
    f2 = wave_sum(A.f2);
    if (STRESS) {
#pragma unroll
      for (int q = 0; q < 9; ++q) A.W[q] = wave_sum(A.W[q]);
    }
  }
  if (lane == 0) {
    ecoul[i] += (double)k * e;
    if (GRAD) {
      fgrad[3 * i + 0] += -2.0f * k * f0;
      fgrad[3 * i + 1] += -2.0f * k * f1;
      fgrad[3 * i + 2] += -2.0f * k * f2;
    }
  }
  if (GRAD && STRESS && lane < 9) {
    float v = A.W[0];
#pragma unroll
    for (int q = 1; q < 9; ++q) v = (lane == q) ? A.W[q] : v;
    virial_atom[(size_t)i * 9 + lane] += k * v;
  }
}

struct D3Pair {
  float ux, uy, uz, d;  // unit vector i -> j and distance in Angstrom
  int j, sj;            // neighbour atom and its species slot
  bool ok;
};

// xs4[j] = (x, y, z, slot as int bits), packed by d3_pack_kernel: one 16-byte gather per neighbour.
// Cell row vectors of the centre atom's system in registers (zeros when non-periodic: the shift codes are 0 then).
struct D3Cell {
  float m[9];
  __device__ __forceinline__ D3Cell(const float* c) {
#pragma unroll
    for (int k = 0; k < 9; ++k) m[k] = c ? c[k] : 0.0f;
  }
};

// Pair geometry from the gathered neighbour record and the packed lattice shift.  The list loops below issue the loads of
// TWO list slots (entry -> xs4[j] -> per-neighbour data) before any arithmetic and are branch-free inside, so two
// dependent-load chains overlap per wave; out-of-range slots are clamped to a valid one and masked out.
__device__ __forceinline__ D3Pair d3_geom(const float4& xj, int j, int sh, const D3Cell& C, float xi, float yi, float zi,
                                          float cutoff, bool valid) {
  D3Pair r;
  r.j = j;
  r.sj = __float_as_int(xj.w);
  int sx, sy, sz;
  unpack_shift(sh, sx, sy, sz);
  const float rx = (xj.x - xi) + (sx * C.m[0] + sy * C.m[3] + sz * C.m[6]);
  const float ry = (xj.y - yi) + (sx * C.m[1] + sy * C.m[4] + sz * C.m[7]);
  const float rz = (xj.z - zi) + (sx * C.m[2] + sy * C.m[5] + sz * C.m[8]);
  const float d2 = fmaxf(rx * rx + ry * ry + rz * rz, 1e-24f);
  const float inv = __builtin_amdgcn_rsqf(d2);
  r.d = d2 * inv;
  r.ux = rx * inv;
  r.uy = ry * inv;
  r.uz = rz * inv;
  r.ok = valid && r.d < cutoff;
  return r;
}

__global__ void d3_pack_kernel(const float* __restrict__ xw, const int* __restrict__ aslot, int n_atoms, float4* __restrict__ xs4) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n_atoms) xs4[i] = make_float4(xw[3 * i], xw[3 * i + 1], xw[3 * i + 2], __int_as_float(aslot[i]));
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void d3_cn_kernel(const float4* __restrict__ xs4, const int* __restrict__ mol_idx,
                                                   const float* __restrict__ cell, int n_cell,
                                                   const int* __restrict__ nb_idx,
                                                   const int* __restrict__ nb_shift, const int* __restrict__ nb_cnt, int cap,
                                                   D3Tables T, float cutoff, int n_atoms, float* __restrict__ d3w) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_atoms) return;
  const int lane = threadIdx.x & 63;
  const float* c = cell ? cell + (n_cell == 1 ? 0 : (size_t)mol_idx[i] * 9) : nullptr;
  const float4 x4 = xs4[i];
  const float xi = x4.x, yi = x4.y, zi = x4.z;
  const int si = __float_as_int(x4.w);
  const float rci = T.rcov[si];
  const int cnt = nb_cnt[i];
  float cn = 0.0f;
  const D3Cell C(c);
  const size_t row = (size_t)i * cap;
  for (int m = lane; m < cnt; m += 64 * D3_SLOTS) {  // D3_SLOTS list slots per trip, their load chains issued together
    int j[D3_SLOTS], h[D3_SLOTS];
    bool v[D3_SLOTS];
    float4 x[D3_SLOTS];
    float rc[D3_SLOTS];
#pragma unroll
    for (int k = 0; k < D3_SLOTS; ++k) {
      v[k] = m + 64 * k < cnt;
      const size_t p = row + (v[k] ? m + 64 * k : m);
      j[k] = nb_idx[p];
      h[k] = c ? nb_shift[p] : 0;
    }
#pragma unroll
    for (int k = 0; k < D3_SLOTS; ++k) x[k] = xs4[j[k]];
#pragma unroll
    for (int k = 0; k < D3_SLOTS; ++k) rc[k] = T.rcov[__float_as_int(x[k].w)];
#pragma unroll
    for (int k = 0; k < D3_SLOTS; ++k) {
      const D3Pair P = d3_geom(x[k], j[k], h[k], C, xi, yi, zi, cutoff, v[k]);
      const float t = frcp(1.0f + fexp(-16.0f * ((rci + rc[k]) * frcp(fmaxf(P.d * BOHR_INV_F, 1e-12f)) - 1.0f)));
      cn += P.ok ? t : 0.0f;
    }
  }
  cn = wave_sum(cn);
  // the five reference-system exponents of atom i, shifted by their maximum (every lane computes all five)
  const int nref = T.nref[si];
  float e[5], mx = -1e30f;
#pragma unroll
  for (int a = 0; a < 5; ++a) {
    const float dc = cn - T.cnref[si * 5 + a];
    e[a] = a < nref ? -4.0f * dc * dc : -1e30f;
    mx = fmaxf(mx, e[a]);
  }
#pragma unroll
  for (int a = 0; a < 5; ++a) {
    if (lane == a) {
      const float sa = a < nref ? e[a] - mx : -1e30f;
      d3w[(size_t)i * D3W + a] = sa;
      d3w[(size_t)i * D3W + 5 + a] = a < nref ? expf(sa) : 0.0f;
    }
  }
  if (lane == 5) d3w[(size_t)i * D3W + 10] = cn;
  if (lane == 6) d3w[(size_t)i * D3W + 11] = 0.0f;
}

// ------------------------------------------------------------------------------------------------
template <bool GRAD, bool STRESS, bool DSF>
__global__ __launch_bounds__(256) void d3_pair_kernel(const float4* __restrict__ xs4, const int* __restrict__ mol_idx,
                                                     const float* __restrict__ cell, int n_cell,
                                                     const int* __restrict__ nb_idx,
                                                     const int* __restrict__ nb_shift, const int* __restrict__ nb_cnt, int cap,
                                                     D3Tables T, D3Params P3, float cutoff, int n_atoms,
                                                     const float* __restrict__ d3w, double* __restrict__ ecoul,
                                                     float* __restrict__ fgrad, float* __restrict__ virial_atom,
                                                     float* __restrict__ dEdcn, CoulombParams cp,
                                                     const float* __restrict__ q, float* __restrict__ qbar) {
  // DSF = true: the damped-shifted-force Coulomb pair terms (model.hip coulomb_dsf_kernel, same arithmetic) ride on this
  // pass - one list traversal and one geometry evaluation for both long-range terms when their cutoffs agree.
  extern __shared__ float c6lds[];  // [4 waves][ns][25]: the C6 reference block of the centre's species against every slot
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + wid;
  if (i >= n_atoms) return;
  const int ns = T.ns;
  float* c6s = c6lds + (size_t)wid * ns * 25;
  const float4 x4 = xs4[i];
  const int si = __float_as_int(x4.w);
  for (int k = lane; k < ns * 25; k += 64) c6s[k] = T.c6slot[(size_t)si * ns * 25 + k];
  __builtin_amdgcn_wave_barrier();
  const float* c = cell ? cell + (n_cell == 1 ? 0 : (size_t)mol_idx[i] * 9) : nullptr;
  const float xi = x4.x, yi = x4.y, zi = x4.z;
  const int nref_i = T.nref[si];
  float s_i[5], w_i[5], g_i[5];
  const float cn_i = d3w[(size_t)i * D3W + 10];
#pragma unroll
  for (int a = 0; a < 5; ++a) {
    s_i[a] = d3w[(size_t)i * D3W + a];
    w_i[a] = d3w[(size_t)i * D3W + 5 + a];
    g_i[a] = -8.0f * (cn_i - T.cnref[si * 5 + a]);  // d log w_i[a] / d cn_i (the max shift cancels in the ratio)
  }
  const float q_i = T.r4r2[si];
  const int cnt = nb_cnt[i];
  D3Acc A;
  float dcn = 0.0f, qb = 0.0f;
  const float al = cp.dsf_alpha, Rc = cp.dsf_rc;
  const float two_a_sqrtpi = 2.0f * al * 0.56418958354775629f;
  float sv = 0.f, slope = 0.f, qc_i = 0.f;
  const float kratio = cp.factor / HALF_HARTREE_F;  // Coulomb terms are accumulated in units of the D3 prefactor
  if (DSF) {
    const float erfc_rc = erfcf(al * Rc);
    sv = erfc_rc / Rc;
    slope = erfc_rc / (Rc * Rc) + two_a_sqrtpi * expf(-al * al * Rc * Rc) / Rc;
    qc_i = q[i];
  }
  const D3Cell C(c);
  const size_t row = (size_t)i * cap;
  auto body = [&](const D3Pair& P, const float4& sj0, const float4& sj1, const float2& sj2, float qj) {
    const int sj = P.sj;
    const float s_j[5] = {sj0.x, sj0.y, sj0.z, sj0.w, sj1.x};
    const float w_j[5] = {sj1.y, sj1.z, sj1.w, sj2.x, sj2.y};
    const float* cr = c6s + sj * 25;
    // sum_ab c6ref_ab w_i[a] w_j[b] [kept]: inner sums over b first, so the centre's weights enter once per a
    float N = 0.f, D = 0.f, G = 0.f, H = 0.f;
#pragma unroll
    for (int a = 0; a < 5; ++a) {
      // (a reference of the CENTRE whose log-weight is below the -12 cut on its own fails the pair cut s_i[a] + s_j[b] >= -12 for
      // every b, s_j <= 0: skipped for the whole wave - typically three or four of the five)
      if (a < nref_i && s_i[a] >= -12.0f) {
        float sb = 0.f, tb = 0.f;
#pragma unroll
        for (int b = 0; b < 5; ++b) {
          const float c6r = cr[a * 5 + b];
          const float w = (s_i[a] + s_j[b] >= -12.0f && c6r != 0.0f) ? w_j[b] : 0.0f;
          sb += c6r * w;
          tb += w;
        }
        N += w_i[a] * sb;
        D += w_i[a] * tb;
        if (GRAD) {
          const float wg = w_i[a] * g_i[a];
          G += wg * sb;
          H += wg * tb;
        }
      }
    }
    const bool has = D > 1.0e-12f;
    const float invD = frcp(fmaxf(D, 1.0e-12f));
    const float c6 = has ? N * invD : 0.0f;
    const float db = fmaxf(P.d * BOHR_INV_F, 1e-12f);
    const float qq = 3.0f * q_i * T.r4r2[sj];
    const float r0 = P3.a1 * __builtin_amdgcn_sqrtf(qq) + P3.a2;
    const float d2 = db * db, d4 = d2 * d2, d6 = d4 * d2, d8 = d4 * d4;
    const float r2 = r0 * r0, r4 = r2 * r2, r6 = r4 * r2, r8 = r4 * r4;
    const float i6 = frcp(d6 + r6), i8 = frcp(d8 + r8);
    const float damp = P3.s6 * i6 + P3.s8 * qq * i8;
    float sw = 1.0f, dsw = 0.0f;
    if (P3.r_off > P3.r_on && db > P3.r_on) {
      const float iw = frcp(P3.r_off - P3.r_on);
      const float t = fminf(fmaxf((db - P3.r_on) * iw, 0.0f), 1.0f);
      const float t2 = t * t;
      sw = 1.0f - t2 * t * (10.0f - 15.0f * t + 6.0f * t2);
      dsw = -30.0f * t2 * (1.0f - 2.0f * t + t2) * iw;
    }
    A.e += (double)(-c6 * damp * sw);
    float tc = 0.0f;
    if (DSF) {
      const float inv = frcp(P.d);
      // erfc(x) = exp(-x^2) t P(t), t = 1 / (1 + x / 2): the degree-9 fit of the list-free walk (model.hip, coulomb_dsf_walk_kernel:
      // 6.5e-9 relative + 2.5e-7 of the fp32 Horner evaluation, the class of erfcf itself) - one exponential serves the value and
      // the derivative, where erfcf() and a second exponential cost ~40 instructions more per list slot
      const float ex = fexp(-al * al * P.d * P.d);
      const float tt = frcp(fmaf(0.5f * al, P.d, 1.0f));
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
#ifdef AIMNET_PROBE_D3_ERFCF  // measurement build: the library erfc of rounds 1 - 5 (tests/tools/d3prof.sh)
      const float ec = erfcf(al * P.d);
#else
      const float ec = pe * tt * ex;
#endif
      const float w = ec * inv - sv + (P.d - Rc) * slope;
      A.ec += (double)(w * qc_i * qj);
      if (GRAD) {
        qb += w * qj;
        tc = kratio * (-ec * inv * inv - two_a_sqrtpi * ex * inv + slope) * qc_i * qj;
      }
    }
    if (GRAD) {
      const float ddamp = -6.0f * P3.s6 * d4 * db * i6 * i6 - 8.0f * P3.s8 * qq * d6 * db * i8 * i8;
      const float de = -c6 * (ddamp * sw + damp * dsw) * BOHR_INV_F;  // d e_ij / d d_ij per Angstrom, C6 held fixed
      d3_add<STRESS>(A, de + tc, P.ux, P.uy, P.uz, P.d);
      if (has) dcn += -damp * sw * (G - c6 * H) * invD;
    }
  };
  for (int m = lane; m < cnt; m += 128) {  // two list slots per trip, all of their loads issued before the arithmetic
    const bool v1 = m + 64 < cnt;
    const int m1 = v1 ? m + 64 : m;
    const int j0 = nb_idx[row + m], j1 = nb_idx[row + m1];
    const int h0 = c ? nb_shift[row + m] : 0, h1 = c ? nb_shift[row + m1] : 0;
    const float4 x0 = xs4[j0], x1 = xs4[j1];
    const float* w0 = d3w + (size_t)j0 * D3W;
    const float* w1 = d3w + (size_t)j1 * D3W;
    const float4 a0 = *reinterpret_cast<const float4*>(w0), b0 = *reinterpret_cast<const float4*>(w0 + 4);
    const float2 c0 = *reinterpret_cast<const float2*>(w0 + 8);
    const float4 a1 = *reinterpret_cast<const float4*>(w1), b1 = *reinterpret_cast<const float4*>(w1 + 4);
    const float2 c1 = *reinterpret_cast<const float2*>(w1 + 8);
    float q0 = 0.f, q1 = 0.f;
    if (DSF) {
      q0 = q[j0];
      q1 = q[j1];
    }
    const D3Pair P0 = d3_geom(x0, j0, h0, C, xi, yi, zi, cutoff, true);
    const D3Pair P1 = d3_geom(x1, j1, h1, C, xi, yi, zi, cutoff, v1);
    if (P0.ok) body(P0, a0, b0, c0, q0);
    if (P1.ok) body(P1, a1, b1, c1, q1);
  }
  d3_store<GRAD, STRESS>(A, i, lane, HALF_HARTREE_F, ecoul, fgrad, virial_atom, DSF ? (double)cp.factor : 0.0);
  if (DSF) {  // self term and dE/dq_i of the DSF sum (lr.py:606-613)
    const float cs = -(sv * 0.5f + al * 0.56418958354775629f);
    if (GRAD) qb = wave_sum(qb);
    if (lane == 0) {
      ecoul[i] += 2.0 * (double)cp.factor * (double)(cs * qc_i * qc_i);
      if (GRAD) qbar[i] += 2.0f * cp.factor * qb + 4.0f * cp.factor * cs * qc_i;
    }
  }
  if (GRAD) {
    dcn = wave_sum(dcn);
    if (lane == 0) dEdcn[i] = 2.0f * HALF_HARTREE_F * dcn;  // e_ij and e_ji both depend on cn_i, symmetrically
  }
}

// ------------------------------------------------------------------------------------------------
template <bool STRESS>
__global__ __launch_bounds__(256) void d3_cnforce_kernel(const float4* __restrict__ xs4, const int* __restrict__ mol_idx,
                                                        const float* __restrict__ cell, int n_cell,
                                                        const int* __restrict__ nb_idx,
                                                        const int* __restrict__ nb_shift, const int* __restrict__ nb_cnt,
                                                        int cap, D3Tables T, float cutoff, int n_atoms,
                                                        const float* __restrict__ dEdcn, double* __restrict__ ecoul,
                                                        float* __restrict__ fgrad, float* __restrict__ virial_atom) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_atoms) return;
  const int lane = threadIdx.x & 63;
  const float* c = cell ? cell + (n_cell == 1 ? 0 : (size_t)mol_idx[i] * 9) : nullptr;
  const float4 x4 = xs4[i];
  const float xi = x4.x, yi = x4.y, zi = x4.z;
  const float rci = T.rcov[__float_as_int(x4.w)];
  const float gi = dEdcn[i];
  const int cnt = nb_cnt[i];
  D3Acc A;
  const D3Cell C(c);
  const size_t row = (size_t)i * cap;
  auto term = [&](const D3Pair& P, float rcj, float gj) {
    const float R = rci + rcj;
    const float idb = frcp(fmaxf(P.d * BOHR_INV_F, 1e-12f));
    const float sg = frcp(1.0f + fexp(-16.0f * (R * idb - 1.0f)));
    const float dsg = sg * (1.0f - sg) * (-16.0f * R * idb * idb) * BOHR_INV_F;  // d sigma / d d_ij per Angstrom
    // sum_l dE/dcn_l cn_l as a pair "energy" with k = 1: per ordered pair 1/2 (g_i + g_j) sigma(d_ij)
    d3_add<STRESS>(A, P.ok ? 0.5f * (gi + gj) * dsg : 0.0f, P.ux, P.uy, P.uz, P.d);
  };
  for (int m = lane; m < cnt; m += 64 * D3_SLOTS) {
    int j[D3_SLOTS], h[D3_SLOTS];
    bool v[D3_SLOTS];
    float4 x[D3_SLOTS];
    float rc[D3_SLOTS], g[D3_SLOTS];
#pragma unroll
    for (int k = 0; k < D3_SLOTS; ++k) {
      v[k] = m + 64 * k < cnt;
      const size_t p = row + (v[k] ? m + 64 * k : m);
      j[k] = nb_idx[p];
      h[k] = c ? nb_shift[p] : 0;
    }
#pragma unroll
    for (int k = 0; k < D3_SLOTS; ++k) {
      x[k] = xs4[j[k]];
      g[k] = dEdcn[j[k]];
    }
#pragma unroll
    for (int k = 0; k < D3_SLOTS; ++k) rc[k] = T.rcov[__float_as_int(x[k].w)];
#pragma unroll
    for (int k = 0; k < D3_SLOTS; ++k) term(d3_geom(x[k], j[k], h[k], C, xi, yi, zi, cutoff, v[k]), rc[k], g[k]);
  }
  d3_store<true, STRESS>(A, i, lane, 1.0f, ecoul, fgrad, virial_atom);
}

// ------------------------------------------------------------------------------------------------
int launch_dftd3(hipStream_t s, bool grad, bool stress, const float* xw, const int* mol_idx, const float* cell, int n_cell,
                 const int* aslot, const int* nb_idx, const int* nb_shift, const int* nb_cnt, int cap, D3Tables T, D3Params P,
                 float cutoff, int n_atoms, float4* xs4, float* d3w, float* dEdcn, double* ecoul, float* fgrad,
                 float* virial_atom, bool with_dsf, CoulombParams cp, const float* q, float* qbar, bool cn_done, const DdLink* dd) {
  hipLaunchKernelGGL(d3_pack_kernel, dim3(ceil_div(n_atoms, 256)), dim3(256), 0, s, xw, aslot, n_atoms, xs4);
  AIMNET_LAUNCH_CHECK();
  dim3 grid(ceil_div(n_atoms, 4)), block(256);
  if (!cn_done) {  // (otherwise the list build left cn_i and the weights in d3w: kernels.h, D3CnRider)
    hipLaunchKernelGGL(d3_cn_kernel, grid, block, 0, s, xs4, mol_idx, cell, n_cell, nb_idx, nb_shift, nb_cnt, cap, T, cutoff, n_atoms,
                       d3w);
    AIMNET_LAUNCH_CHECK();
  }
  // domain decomposition: the coordination number (and with it the reference weights) of a halo copy is exact only if its own
  // 15 A neighbourhood is inside the cluster - the owners' rows come in through the exchange function, and so does dE/dcn below
  // (for an owned centre both are complete: the pair pass forms both directions of a pair at the centre)
  if (dd && dd->fn(dd->ctx, 2 /* AIMNET_DD_ROWS */, d3w, (int64_t)D3W * n_atoms, (void*)s) != 0) {
    set_last_error("eval: the domain-decomposition exchange function failed (DFT-D3 weights)");
    return -1;
  }
  const size_t lds = (size_t)4 * T.ns * 25 * sizeof(float);
#define AIMNET_D3_PAIR(G_, S_, C_)                                                                                             \
  hipLaunchKernelGGL((d3_pair_kernel<G_, S_, C_>), grid, block, lds, s, xs4, mol_idx, cell, n_cell, nb_idx, nb_shift, nb_cnt, cap, \
                     T, P, cutoff, n_atoms, d3w, ecoul, fgrad, virial_atom, dEdcn, cp, q, qbar)
  if (with_dsf) {
    if (grad && stress) AIMNET_D3_PAIR(true, true, true);
    else if (grad) AIMNET_D3_PAIR(true, false, true);
    else AIMNET_D3_PAIR(false, false, true);
  } else {
    if (grad && stress) AIMNET_D3_PAIR(true, true, false);
    else if (grad) AIMNET_D3_PAIR(true, false, false);
    else AIMNET_D3_PAIR(false, false, false);
  }
#undef AIMNET_D3_PAIR
  AIMNET_LAUNCH_CHECK();
  if (grad && dd && dd->fn(dd->ctx, 2 /* AIMNET_DD_ROWS */, dEdcn, (int64_t)n_atoms, (void*)s) != 0) {
    set_last_error("eval: the domain-decomposition exchange function failed (DFT-D3 dE/dcn)");
    return -1;
  }
  if (grad) {
    if (stress)
      hipLaunchKernelGGL(d3_cnforce_kernel<true>, grid, block, 0, s, xs4, mol_idx, cell, n_cell, nb_idx, nb_shift, nb_cnt, cap, T,
                         cutoff, n_atoms, dEdcn, ecoul, fgrad, virial_atom);
    else
      hipLaunchKernelGGL(d3_cnforce_kernel<false>, grid, block, 0, s, xs4, mol_idx, cell, n_cell, nb_idx, nb_shift, nb_cnt, cap, T,
                         cutoff, n_atoms, dEdcn, ecoul, fgrad, virial_atom);
    AIMNET_LAUNCH_CHECK();
  }
  return 0;
}

}  // namespace aimnet
