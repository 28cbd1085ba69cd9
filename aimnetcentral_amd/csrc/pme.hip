// Smooth particle-mesh Ewald: the reciprocal-space sum of LRCoulomb "pme" (reference aimnet/modules/lr.py:752-775, parameters
// and the per-call real-space list calculator.py:1560-1603) on a mesh.  The reference delegates the arithmetic to
// nvalchemiops.particle_mesh_ewald (un-vendored, 0.4.0: UNPINNED against that kernel).  Restated from the published method
// (Essmann et al., J. Chem. Phys. 103, 8577 (1995)); the CPU twin is oracle/pme.py, which is pinned to the exact structure-factor
// sum (ewald.hip / oracle ewald_reciprocal): both converge to the same energy, forces and strain derivative.
//
//   real space, self term       the cell-grid walk of model.hip with this file's (alpha, rc)             (as for "ewald")
//   reciprocal space            Q(mesh) = sum_i q_i M8 M8 M8  ->  Q^ = DFT(Q)  ->  theta Q^  ->  pot = DFT^-1  ->  phi_i, grad phi_i
//   background                  -pi Q^2 / (2 V alpha^2)                                                   (as for "ewald")
//
// Splitting: the Ewald estimate eta = (V^2 / N)^(1/6) / sqrt(2 pi), f = sqrt(-2 ln accuracy) until its real-space cutoff f eta
// reaches 10 A; from there rc = 10 A and alpha = f / (sqrt(2) rc), so the walk is O(N) and the mesh carries the rest.  Mesh: order-8
// cardinal B-splines, K_a = even(ceil(over (2 kc |a_a| / 2 pi + 1))) >= 8 points along cell vector a_a with kc = sqrt(2) f alpha and
// over = 1 + (-log10(accuracy) - 4) / 4 (1.0 at 1e-4, 1.5 at 1e-6, 2.0 at 1e-8): the rms force error against the exact sum stays
// below accuracy x the rms reciprocal force (tests/tools/pme_calibrate.py).  Everything is decided on the device from the cell (no
// host round trip, NPT-safe); the host only provides the mesh capacity and grows it when status[7] says so.
//
// MI355X notes.  (1) Charge assignment uses 64-bit integer atomics on a 2^-44 fixed-point mesh: integer sums do not depend on the
// order of arrival, so the evaluation stays bitwise repeatable.  (2) The transforms are DIRECT axis transforms in double: a block
// stages whole mesh lines in LDS and every thread forms output coefficients as K-term sums with twiddles from an LDS table
// ((m n) mod K by integer stepping, exact).  For the meshes of 10^4 - 10^6 atoms (K <= 512 per axis) that is 10^8 - 10^10 fp64
// FMAs, microseconds to a millisecond on 256 CUs at full-rate fp64, any K, no radix restriction, no library.  (3) One wave per atom
// for assignment and for interpolation: 8 x 64 = 512 mesh points, lane = (j2, j3), loop over j1.
#include <hip/hip_runtime.h>

#include <climits>

#include "common.h"
#include "ewald_common.h"
#include "kernels.h"

namespace aimnet {

namespace {

constexpr int PME_P = 8;                          // spline order
constexpr double PME_FIX = 17592186044416.0;      // 2^44
constexpr double PME_RC_MAX = 10.0;               // Angstrom
constexpr int PME_MIN_MESH = 8;
constexpr int PME_LDS_PTS = 3072;                 // complex doubles of mesh lines a transform block stages (48 KiB + 8 KiB of twiddles)
constexpr double PME_PI = 3.141592653589793;

__global__ void pme_setup_kernel(const float* __restrict__ cell, int n_cell, const int* __restrict__ mol_start,
                                 const float* __restrict__ charge, int nq, int n_mol, float accuracy, int max_mesh,
                                 EwaldSystem* __restrict__ es, int* __restrict__ status) {
  __shared__ int s_need;
  if (threadIdx.x == 0) s_need = 0;
  __syncthreads();
  for (int s = threadIdx.x; s < n_mol; s += blockDim.x) {
    const float* c = cell + (n_cell == 1 ? 0 : (size_t)s * 9);
    double m[9];
    EwaldSystem E;
    const double det = ewald_cell_geometry(c, m, E);
    const double vol = fabs(det);
    const int ns = max(1, mol_start[s + 1] - mol_start[s]);
    const double eta = cbrt(sqrt(vol * vol / (double)ns)) / sqrt(2.0 * PME_PI);
    const double f = sqrt(-2.0 * log((double)accuracy));
    const double rc = fmin(f * eta, PME_RC_MAX);
    const double alpha = f / (sqrt(2.0) * rc), kc = sqrt(2.0) * f * alpha;
    const double over = 1.0 + 0.25 * fmax(0.0, -log10((double)accuracy) - 4.0);
    E.alpha = (float)alpha;
    E.rc = (float)rc;
    E.kc2 = (float)(kc * kc);
    E.inv4a2 = (float)(1.0 / (4.0 * alpha * alpha));
    double Q = 0.0;
    for (int ch = 0; ch < nq; ++ch) Q += (double)charge[(size_t)ch * n_mol + s];
    E.phi_bg = (float)(-PME_PI * Q / (vol * alpha * alpha));
    E.pref = 8.0 * PME_PI / vol;
    long pts = 1;
    bool axis_ok = true;
    for (int a = 0; a < 3; ++a) {
      for (int cc = 0; cc < 3; ++cc) E.b[a * 3 + cc] = 2.0 * PME_PI * E.inv[cc * 3 + a];
      const double len = sqrt(m[3 * a] * m[3 * a] + m[3 * a + 1] * m[3 * a + 1] + m[3 * a + 2] * m[3 * a + 2]);
      const double nmax = kc * len / (2.0 * PME_PI);
      const double want = ceil(over * (2.0 * nmax + 1.0));
      int k = want > 1e6 ? 1000000 : (int)want;
      k += k & 1;
      k = max(PME_MIN_MESH, k);
      if (k > PME_MAX_AXIS) axis_ok = false;
      E.mesh[a] = k;
      E.nmax[a] = 0;
      pts *= k;
    }
    E.n2w = E.n3w = 1;
    E.k_offset = E.n_box = 0;
    const int need = !axis_ok ? INT_MAX : (int)min(pts, (long)INT_MAX - 1);
    atomicMax(&s_need, need);
    if (!axis_ok || pts > (long)max_mesh) E.mesh[0] = E.mesh[1] = E.mesh[2] = 0, pts = 0;
    E.mesh_pts = (int)pts;
    es[s] = E;
  }
  __syncthreads();
  if (threadIdx.x == 0) *status = s_need;
}

// M_8(d + 7 - j), j = 0 .. 7: the weights of the mesh points floor(u) - 7 + j of an atom at scaled coordinate u = floor(u) + d, and
// their derivatives with respect to u (Essmann eq. 4.1 recursion; the twin of oracle/pme.py bspline)
__device__ __forceinline__ void bspline8(double d, double w[PME_P], double dw[PME_P]) {
#pragma unroll
  for (int j = 0; j < PME_P; ++j) w[j] = 0.0;
  w[0] = 1.0 - d;
  w[1] = d;
#pragma unroll
  for (int k = 3; k <= PME_P; ++k) {
    if (k == PME_P) {
      dw[0] = -w[0];
#pragma unroll
      for (int j = 1; j < PME_P; ++j) dw[j] = w[j - 1] - w[j];
    }
    const double div = 1.0 / (double)(k - 1);
    w[k - 1] = div * d * w[k - 2];
#pragma unroll
    for (int j = 1; j < k - 1; ++j) w[k - 1 - j] = div * ((d + (double)j) * w[k - 2 - j] + ((double)(k - j) - d) * w[k - 1 - j]);
    w[0] = div * (1.0 - d) * w[0];
  }
}

__device__ __forceinline__ double pick8(const double v[PME_P], int j) {
  double r = v[0];
#pragma unroll
  for (int t = 1; t < PME_P; ++t) r = j == t ? v[t] : r;
  return r;
}

// scaled coordinate of one axis: base mesh index floor(u) (in [0, K)) and the fractional part
__device__ __forceinline__ void pme_scaled(double f, int K, int& fl, double& d) {
  f -= floor(f);
  double u = f * (double)K;
  fl = (int)u;
  if (fl >= K) fl = K - 1;  // (f rounded to 1.0)
  d = u - (double)fl;
}

__device__ __forceinline__ int pme_wrap(int t, int K) { return t < 0 ? t + K : t; }  // t in [-7, K): K >= 8

// grid (x, n_mol): zero the system's charge mesh; block 0 forms the spline moduli 1 / |sum_j M8(j + 1) e^{2 pi i m j / K}|^2
__global__ __launch_bounds__(256) void pme_prepare_kernel(const EwaldSystem* __restrict__ es, long long* __restrict__ meshq,
                                                         double* __restrict__ bmod, size_t max_mesh) {
  const int s = blockIdx.y;
  const EwaldSystem& E = es[s];
  long long* mq = meshq + (size_t)s * max_mesh;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < (size_t)E.mesh_pts; e += (size_t)gridDim.x * 256) mq[e] = 0;
  if (blockIdx.x != 0 || E.mesh_pts == 0) return;
  double w[PME_P], dw[PME_P];
  bspline8(0.0, w, dw);  // w[j] = M8(7 - j)
  for (int a = 0; a < 3; ++a) {
    const int K = E.mesh[a];
    double* out = bmod + ((size_t)s * 3 + a) * PME_MAX_AXIS;
    for (int m = threadIdx.x; m < K; m += 256) {
      double re = 0.0, im = 0.0;
#pragma unroll
      for (int j = 0; j < PME_P - 1; ++j) {
        double sn, cs;
        sincospi(2.0 * (double)((m * j) % K) / (double)K, &sn, &cs);
        re += w[PME_P - 2 - j] * cs;  // M8(j + 1)
        im += w[PME_P - 2 - j] * sn;
      }
      out[m] = 1.0 / (re * re + im * im);  // (order 8: the sum has no zero on the mesh frequencies)
    }
  }
}

// one wave per atom: q_i M8 M8 M8 onto the 512 mesh points around the atom, integer atomics
__global__ __launch_bounds__(256) void pme_spread_kernel(const double* __restrict__ frac, const float* __restrict__ q,
                                                        const int* __restrict__ mol_idx, int n_atoms,
                                                        const EwaldSystem* __restrict__ es, long long* __restrict__ meshq,
                                                        size_t max_mesh) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_atoms) return;
  const int lane = threadIdx.x & 63;
  const int s = mol_idx[i];
  const EwaldSystem& E = es[s];
  if (E.mesh_pts == 0) return;
  const int K0 = E.mesh[0], K1 = E.mesh[1], K2 = E.mesh[2];
  int f0, f1, f2;
  double d0, d1, d2, w0[PME_P], w1[PME_P], w2[PME_P], dw[PME_P];
  pme_scaled(frac[(size_t)i * 3], K0, f0, d0);
  pme_scaled(frac[(size_t)i * 3 + 1], K1, f1, d1);
  pme_scaled(frac[(size_t)i * 3 + 2], K2, f2, d2);
  bspline8(d0, w0, dw);
  bspline8(d1, w1, dw);
  bspline8(d2, w2, dw);
  const int j1 = lane >> 3, j2 = lane & 7;
  const double w12 = (double)q[i] * pick8(w1, j1) * pick8(w2, j2);
  const size_t off12 = (size_t)pme_wrap(f1 - (PME_P - 1) + j1, K1) * K2 + pme_wrap(f2 - (PME_P - 1) + j2, K2);
  unsigned long long* mq = reinterpret_cast<unsigned long long*>(meshq + (size_t)s * max_mesh);
#pragma unroll
  for (int j0 = 0; j0 < PME_P; ++j0) {
    const size_t at = (size_t)pme_wrap(f0 - (PME_P - 1) + j0, K0) * K1 * K2 + off12;
    const long long v = __double2ll_rn(w0[j0] * w12 * PME_FIX);
    atomicAdd(mq + at, (unsigned long long)v);
  }
}

// One axis of the 3-D transform: out[.., m, ..] = sum_n in[.., n, ..] e^{-+ 2 pi i m n / K}.  The mesh is [K0][K1][K2]; seen from
// axis a it is [outer][K][inner].  A block takes T consecutive lines (T K <= PME_LDS_PTS).
template <bool FIXED_IN, bool REAL_OUT, bool INV>
__global__ __launch_bounds__(256) void pme_dft_kernel(const EwaldSystem* __restrict__ es, int axis, const void* __restrict__ in_,
                                                     void* __restrict__ out_, size_t max_mesh) {
  const int s = blockIdx.y;
  const EwaldSystem& E = es[s];
  if (E.mesh_pts == 0) return;
  const int K = E.mesh[axis];
  const int inner = axis == 0 ? E.mesh[1] * E.mesh[2] : axis == 1 ? E.mesh[2] : 1;
  const int n_lines = E.mesh_pts / K;
  const int T = max(1, min(16, PME_LDS_PTS / K));
  const int line0 = blockIdx.x * T;
  if (line0 >= n_lines) return;
  const int nt = min(T, n_lines - line0);
  __shared__ double xr[PME_LDS_PTS], xi[PME_LDS_PTS];
  __shared__ double twr[PME_MAX_AXIS], twi[PME_MAX_AXIS];
  for (int j = threadIdx.x; j < K; j += 256) {
    double sn, cs;
    sincospi(2.0 * (double)j / (double)K, &sn, &cs);
    twr[j] = cs;
    twi[j] = INV ? sn : -sn;
  }
  const size_t base = (size_t)s * max_mesh;
  const bool contiguous = inner == 1;  // lines along the fastest index: element-major mapping keeps the accesses coalesced
  for (int e = threadIdx.x; e < nt * K; e += 256) {
    const int t = contiguous ? e / K : e % nt, n = contiguous ? e % K : e / nt;
    const int line = line0 + t;
    const size_t at = (size_t)(line / inner) * K * inner + (size_t)(line % inner) + (size_t)n * inner;
    double re, im;
    if (FIXED_IN) {
      re = (double)reinterpret_cast<const long long*>(in_)[base + at] * (1.0 / PME_FIX);
      im = 0.0;
    } else {
      const double2 v = reinterpret_cast<const double2*>(in_)[base + at];
      re = v.x;
      im = v.y;
    }
    xr[n * T + t] = re;
    xi[n * T + t] = im;
  }
  __syncthreads();
  for (int o = threadIdx.x; o < nt * K; o += 256) {
    const int t = contiguous ? o / K : o % nt, m = contiguous ? o % K : o / nt;
    double ar = 0.0, ai = 0.0;
    int idx = 0;
    for (int n = 0; n < K; ++n) {
      const double a = xr[n * T + t], b = xi[n * T + t], c = twr[idx], d = twi[idx];
      ar += a * c - b * d;
      ai += a * d + b * c;
      idx += m;
      if (idx >= K) idx -= K;
    }
    const int line = line0 + t;
    const size_t at = (size_t)(line / inner) * K * inner + (size_t)(line % inner) + (size_t)m * inner;
    if (REAL_OUT)
      reinterpret_cast<double*>(out_)[base + at] = ar;
    else
      reinterpret_cast<double2*>(out_)[base + at] = double2{ar, ai};
  }
}

// Q^ <- theta Q^ with theta(m) = (4 pi / V) exp(-k^2 / 4 alpha^2) / k^2 |b0 b1 b2|^2 (m != 0), and per block the sums of
// theta |Q^|^2 (vfac k_a k_b [6], 1) for the strain derivative
__global__ __launch_bounds__(256) void pme_influence_kernel(const EwaldSystem* __restrict__ es, double* __restrict__ mesh,
                                                           const double* __restrict__ bmod, double* __restrict__ vpart,
                                                           size_t max_mesh, int max_parts) {
  const int s = blockIdx.y;
  const EwaldSystem& E = es[s];
  if ((size_t)blockIdx.x * PME_PART >= (size_t)E.mesh_pts) return;
  const int K0 = E.mesh[0], K1 = E.mesh[1], K2 = E.mesh[2];
  const double* b0 = bmod + ((size_t)s * 3 + 0) * PME_MAX_AXIS;
  const double* b1 = bmod + ((size_t)s * 3 + 1) * PME_MAX_AXIS;
  const double* b2 = bmod + ((size_t)s * 3 + 2) * PME_MAX_AXIS;
  double2* A = reinterpret_cast<double2*>(mesh) + (size_t)s * max_mesh;
  double acc[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int r = 0; r < PME_PART / 256; ++r) {
    const int e = blockIdx.x * PME_PART + r * 256 + threadIdx.x;
    if (e >= E.mesh_pts) break;
    const int i0 = e / (K1 * K2), rem = e - i0 * (K1 * K2), i1 = rem / K2, i2 = rem - i1 * K2;
    const int m0 = i0 <= K0 / 2 ? i0 : i0 - K0, m1 = i1 <= K1 / 2 ? i1 : i1 - K1, m2 = i2 <= K2 / 2 ? i2 : i2 - K2;
    double kx[3], k2 = 0.0;
    for (int c = 0; c < 3; ++c) {
      kx[c] = (double)m0 * E.b[c] + (double)m1 * E.b[3 + c] + (double)m2 * E.b[6 + c];
      k2 += kx[c] * kx[c];
    }
    double theta = 0.0;
    if (e != 0) theta = 0.5 * E.pref * exp(-k2 * (double)E.inv4a2) / k2 * b0[i0] * b1[i1] * b2[i2];
    const double2 v = A[e];
    A[e] = double2{theta * v.x, theta * v.y};
    if (e != 0) {
      const double ts = theta * (v.x * v.x + v.y * v.y), tv = ts * 2.0 * (1.0 / k2 + (double)E.inv4a2);
      acc[0] += tv * kx[0] * kx[0]; acc[1] += tv * kx[0] * kx[1]; acc[2] += tv * kx[0] * kx[2];
      acc[3] += tv * kx[1] * kx[1]; acc[4] += tv * kx[1] * kx[2]; acc[5] += tv * kx[2] * kx[2];
      acc[6] += ts;
    }
  }
  __shared__ double sh[4][7];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int c = 0; c < 7; ++c) {
    const double v = wave_sum(acc[c]);
    if (lane == 0) sh[w][c] = v;
  }
  __syncthreads();
  if (threadIdx.x < 7)
    vpart[((size_t)s * max_parts + blockIdx.x) * 8 + threadIdx.x] =
        sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}

// one wave per atom: phi_i and grad phi_i from the potential mesh; the system's first atom also carries the mesh part of the strain
// derivative (the per-atom rows are only ever summed per system)
template <bool GRAD, bool STRESS>
__global__ __launch_bounds__(256) void pme_gather_kernel(const double* __restrict__ frac, const float* __restrict__ q,
                                                        const int* __restrict__ mol_idx, const int* __restrict__ mol_start,
                                                        int n_atoms, const EwaldSystem* __restrict__ es,
                                                        const double* __restrict__ pot_, const double* __restrict__ vpart,
                                                        size_t max_mesh, int max_parts, float factor, double* __restrict__ ecoul,
                                                        float* __restrict__ qbar, float* __restrict__ fgrad,
                                                        float* __restrict__ virial_atom) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_atoms) return;
  const int lane = threadIdx.x & 63;
  const int s = mol_idx[i];
  const EwaldSystem& E = es[s];
  if (E.mesh_pts == 0) return;
  const int K0 = E.mesh[0], K1 = E.mesh[1], K2 = E.mesh[2];
  int f0, f1, f2;
  double d0, d1, d2, w0[PME_P], w1[PME_P], w2[PME_P], dw0[PME_P], dw1[PME_P], dw2[PME_P];
  pme_scaled(frac[(size_t)i * 3], K0, f0, d0);
  pme_scaled(frac[(size_t)i * 3 + 1], K1, f1, d1);
  pme_scaled(frac[(size_t)i * 3 + 2], K2, f2, d2);
  bspline8(d0, w0, dw0);
  bspline8(d1, w1, dw1);
  bspline8(d2, w2, dw2);
  const int j1 = lane >> 3, j2 = lane & 7;
  const double a1 = pick8(w1, j1), a2 = pick8(w2, j2), da1 = pick8(dw1, j1), da2 = pick8(dw2, j2);
  const size_t off12 = (size_t)pme_wrap(f1 - (PME_P - 1) + j1, K1) * K2 + pme_wrap(f2 - (PME_P - 1) + j2, K2);
  const double* pot = pot_ + (size_t)s * max_mesh;
  double phi = 0.0, g0 = 0.0, g1 = 0.0, g2 = 0.0;
#pragma unroll
  for (int j0 = 0; j0 < PME_P; ++j0) {
    const double p = pot[(size_t)pme_wrap(f0 - (PME_P - 1) + j0, K0) * K1 * K2 + off12];
    phi += w0[j0] * a1 * a2 * p;
    if (GRAD) {
      g0 += dw0[j0] * a1 * a2 * p;
      g1 += w0[j0] * da1 * a2 * p;
      g2 += w0[j0] * a1 * da2 * p;
    }
  }
  phi = wave_sum(phi);
  if (GRAD) {
    g0 = wave_sum(g0) * (double)K0;
    g1 = wave_sum(g1) * (double)K1;
    g2 = wave_sum(g2) * (double)K2;
  }
  double W[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  const bool first = STRESS && i == mol_start[s];
  if (first) {  // (wave-uniform)
    const int np = (E.mesh_pts + PME_PART - 1) / PME_PART;
    for (int b = lane; b < np; b += 64)
      for (int c = 0; c < 7; ++c) W[c] += vpart[((size_t)s * max_parts + b) * 8 + c];
    for (int c = 0; c < 7; ++c) W[c] = wave_sum(W[c]);
  }
  if (lane != 0) return;
  const double qi = q[i];
  const double phi_all = phi + (double)E.phi_bg;
  ecoul[i] += (double)factor * qi * phi_all;
  if (GRAD) {
    qbar[i] += (float)(2.0 * (double)factor * phi_all);
    for (int c = 0; c < 3; ++c)
      fgrad[3 * i + c] += (float)(2.0 * (double)factor * qi * (E.inv[c * 3] * g0 + E.inv[c * 3 + 1] * g1 + E.inv[c * 3 + 2] * g2));
    if (STRESS) {
      float* v = virial_atom + (size_t)i * 9;
      const double bg = -(double)factor * qi * (double)E.phi_bg;  // the background's share of -delta_ab E
      const double f = (double)factor;
      v[0] += (float)(bg + f * (W[0] - W[6])); v[1] += (float)(f * W[1]); v[2] += (float)(f * W[2]);
      v[3] += (float)(f * W[1]); v[4] += (float)(bg + f * (W[3] - W[6])); v[5] += (float)(f * W[4]);
      v[6] += (float)(f * W[2]); v[7] += (float)(f * W[4]); v[8] += (float)(bg + f * (W[5] - W[6]));
    }
  }
}

}  // namespace

int launch_pme_setup(hipStream_t s, const float* cell, int n_cell, const int* mol_start, const int* mol_idx, const float* xw,
                     const float* charge, int nq, int n_atoms, int n_mol, float accuracy, EwaldBuffers& b, int* status) {
  hipLaunchKernelGGL(pme_setup_kernel, dim3(1), dim3(256), 0, s, cell, n_cell, mol_start, charge, nq, n_mol, accuracy, b.max_mesh, b.sys,
                     status);
  AIMNET_LAUNCH_CHECK();
  return launch_ewald_frac(s, xw, mol_idx, n_atoms, b);
}

int launch_pme_recip(hipStream_t s, bool grad, bool stress, const float* q, const int* mol_idx, const int* mol_start, int n_atoms,
                     int n_mol, const EwaldBuffers& b, float factor, double* ecoul, float* qbar, float* fgrad, float* virial_atom) {
  const size_t mm = (size_t)b.max_mesh;
  const dim3 blk(256);
  hipLaunchKernelGGL(pme_prepare_kernel, dim3(std::max(1, std::min(1024, ceil_div(b.max_mesh, 1024))), n_mol), blk, 0, s, b.sys,
                     b.meshq, b.bmod, mm);
  AIMNET_LAUNCH_CHECK();
  hipLaunchKernelGGL(pme_spread_kernel, dim3(ceil_div(n_atoms, 4)), blk, 0, s, b.frac, q, mol_idx, n_atoms, b.sys, b.meshq, mm);
  AIMNET_LAUNCH_CHECK();
  // blocks of a transform pass: lines / T with T K >= min(PME_LDS_PTS, 16 K) / 2 ... bounded by the smallest line tile, 16 lines of 8
  const dim3 gdft(ceil_div(b.max_mesh, 16 * PME_MIN_MESH), n_mol);
  hipLaunchKernelGGL((pme_dft_kernel<true, false, false>), gdft, blk, 0, s, b.sys, 2, b.meshq, b.ma, mm);
  AIMNET_LAUNCH_CHECK();
  hipLaunchKernelGGL((pme_dft_kernel<false, false, false>), gdft, blk, 0, s, b.sys, 1, b.ma, b.mb, mm);
  AIMNET_LAUNCH_CHECK();
  hipLaunchKernelGGL((pme_dft_kernel<false, false, false>), gdft, blk, 0, s, b.sys, 0, b.mb, b.ma, mm);
  AIMNET_LAUNCH_CHECK();
  hipLaunchKernelGGL(pme_influence_kernel, dim3(b.max_parts, n_mol), blk, 0, s, b.sys, b.ma, b.bmod, b.vpart, mm, b.max_parts);
  AIMNET_LAUNCH_CHECK();
  hipLaunchKernelGGL((pme_dft_kernel<false, false, true>), gdft, blk, 0, s, b.sys, 0, b.ma, b.mb, mm);
  AIMNET_LAUNCH_CHECK();
  hipLaunchKernelGGL((pme_dft_kernel<false, false, true>), gdft, blk, 0, s, b.sys, 1, b.mb, b.ma, mm);
  AIMNET_LAUNCH_CHECK();
  hipLaunchKernelGGL((pme_dft_kernel<false, true, true>), gdft, blk, 0, s, b.sys, 2, b.ma, b.meshq, mm);
  AIMNET_LAUNCH_CHECK();
  const dim3 grid(ceil_div(n_atoms, 4));
  const double* pot = reinterpret_cast<const double*>(b.meshq);
  if (grad && stress)
    hipLaunchKernelGGL((pme_gather_kernel<true, true>), grid, blk, 0, s, b.frac, q, mol_idx, mol_start, n_atoms, b.sys, pot, b.vpart, mm,
                       b.max_parts, factor, ecoul, qbar, fgrad, virial_atom);
  else if (grad)
    hipLaunchKernelGGL((pme_gather_kernel<true, false>), grid, blk, 0, s, b.frac, q, mol_idx, mol_start, n_atoms, b.sys, pot, b.vpart, mm,
                       b.max_parts, factor, ecoul, qbar, fgrad, virial_atom);
  else
    hipLaunchKernelGGL((pme_gather_kernel<false, false>), grid, blk, 0, s, b.frac, q, mol_idx, mol_start, n_atoms, b.sys, pot, b.vpart, mm,
                       b.max_parts, factor, ecoul, qbar, fgrad, virial_atom);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

}  // namespace aimnet
