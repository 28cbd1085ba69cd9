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
// Launches per evaluation: setup (+ mesh clear + spline moduli) in front of the walk; assignment, 3 + 3 axis transforms around the
// influence function, interpolation behind it: 10.
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
constexpr int PME_LDS_PTS = 1024;                 // complex doubles of mesh lines a transform block stages (16 KiB)
constexpr double PME_PI = 3.141592653589793;

// the splitting and the mesh of one system; returns the mesh points it needs (INT_MAX: an axis beyond PME_MAX_AXIS)
__device__ int pme_system(const float* __restrict__ c, int n_atoms_sys, const float* __restrict__ charge, int nq, int n_mol, int s,
                          float accuracy, int max_mesh, EwaldSystem& E) {
  double m[9];
  const double det = ewald_cell_geometry(c, m, E);
  const double vol = fabs(det);
  const int ns = max(1, n_atoms_sys);
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
  if (!axis_ok || pts > (long)max_mesh) E.mesh[0] = E.mesh[1] = E.mesh[2] = 0, pts = 0;
  E.mesh_pts = (int)pts;
  return need;
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

// scaled coordinate of one axis: base mesh index floor(u) (in [0, K)) and the fractional part; the fractional coordinate in
// double from the fp32 position (any image)
__device__ __forceinline__ void pme_scaled(const float* __restrict__ xw, int i, int ax, const EwaldSystem& E, int K, int& fl, double& d) {
  double f = (double)xw[3 * (size_t)i] * E.inv[ax] + (double)xw[3 * (size_t)i + 1] * E.inv[3 + ax] + (double)xw[3 * (size_t)i + 2] * E.inv[6 + ax];
  f -= floor(f);
  double u = f * (double)K;
  fl = (int)u;
  if (fl >= K) fl = K - 1;  // (f rounded to 1.0)
  d = u - (double)fl;
}

__device__ __forceinline__ int pme_wrap(int t, int K) { return t < 0 ? t + K : t; }  // t in [-7, K): K >= 8

// The splines of one atom, once per wave: lane a < 3 runs the recursion for axis a and leaves the 8 weights, their 8 derivatives
// and the base index in LDS (every lane running all three recursions costs three times the arithmetic and, with per-lane picks
// from the arrays, private memory).
struct PmeSplines {
  double w[3][16];  // [axis][0..7 weights, 8..15 derivatives]
  int fl[3];
  int pad;
};
__device__ __forceinline__ void pme_wave_splines(const float* __restrict__ xw, int i, bool live, const EwaldSystem& E, int lane,
                                                 PmeSplines& S) {
  if (live && lane < 3) {
    int fl;
    double d, w[PME_P], dw[PME_P];
    pme_scaled(xw, i, lane, E, E.mesh[lane], fl, d);
    bspline8(d, w, dw);
#pragma unroll
    for (int j = 0; j < PME_P; ++j) {
      S.w[lane][j] = w[j];
      S.w[lane][8 + j] = dw[j];
    }
    S.fl[lane] = fl;
  }
}

// grid (x, n_mol), ONE launch in front of the real-space walk: every block derives its system's splitting and mesh from the cell
// (a few hundred instructions, cheaper than a launch of its own), blocks x = 0 publish them; then the blocks zero the system's
// charge mesh and block x = 0 forms the spline moduli 1 / |sum_j M8(j + 1) e^{2 pi i m j / K}|^2.  Block (0, 0) also reports the
// mesh points the largest system needs.
__global__ __launch_bounds__(256) void pme_setup_kernel(const float* __restrict__ cell, int n_cell, const int* __restrict__ mol_start,
                                                       const float* __restrict__ charge, int nq, int n_mol, float accuracy,
                                                       int max_mesh, EwaldSystem* __restrict__ es, int* __restrict__ status,
                                                       long long* __restrict__ meshq, double* __restrict__ bmod) {
  const int s = blockIdx.y;
  __shared__ EwaldSystem sE;
  __shared__ int s_need;
  if (threadIdx.x == 0) {
    s_need = 0;
    pme_system(cell + (n_cell == 1 ? 0 : (size_t)s * 9), mol_start[s + 1] - mol_start[s], charge, nq, n_mol, s, accuracy, max_mesh, sE);
    if (blockIdx.x == 0) es[s] = sE;
  }
  __syncthreads();
  if (blockIdx.x == 0 && s == 0) {  // (thread t's private copy: the reported need covers every system)
    for (int t = threadIdx.x; t < n_mol; t += 256) {
      EwaldSystem tmp;
      atomicMax(&s_need, pme_system(cell + (n_cell == 1 ? 0 : (size_t)t * 9), mol_start[t + 1] - mol_start[t], charge, nq, n_mol, t,
                                    accuracy, max_mesh, tmp));
    }
    __syncthreads();
    if (threadIdx.x == 0) *status = s_need;
  }
  const EwaldSystem& E = sE;
  long long* mq = meshq + (size_t)s * (size_t)max_mesh;
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

// Charge assignment: q_i M8 M8 M8 onto the 512 mesh points around every atom, as integer atomics.  Memory-side atomics are the
// cost (5 x 10^6 of them for 10^4 atoms: ~60 us), so a block takes PME_SG atoms that follow each other in the engine's bin order
// (`order`: neighbours in space) and sums them into an LDS tile over their common footprint first; only the tile's non-zero points go to
// the mesh (3 - 4 x fewer atomics).  Atoms that do not fit one tile (input order, a system boundary inside the block) take the
// direct form.  Both forms add the same integers: the result does not depend on which one ran.
constexpr int PME_SG = 32;       // atoms per block
constexpr int PME_TILE = 6144;   // tile points (48 KiB)
__global__ __launch_bounds__(256) void pme_spread_kernel(const float* __restrict__ xw, const float* __restrict__ q,
                                                        const int* __restrict__ mol_idx, const int* __restrict__ order,
                                                        int n_atoms, const EwaldSystem* __restrict__ es,
                                                        long long* __restrict__ meshq, size_t max_mesh) {
  __shared__ unsigned long long tile[PME_TILE];
  __shared__ double sw[PME_SG][3][PME_P];
  __shared__ int sfl[PME_SG][3], srel[PME_SG][3];
  __shared__ int smin[3], smax[3], satom[PME_SG], s_mixed;
  const int a0 = blockIdx.x * PME_SG, na = min(PME_SG, n_atoms - a0);
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  if (tid < 3) smin[tid] = INT_MAX, smax[tid] = INT_MIN;
  if (tid == 0) s_mixed = 0;
  if (tid < na) satom[tid] = order ? order[a0 + tid] : a0 + tid;  // the atoms of this block, in processing order
  __syncthreads();
  const int s = mol_idx[satom[0]];
  if (tid < 3 * na) {  // the splines of (atom, axis): 96 recursions side by side
    const int a = tid / 3, ax = tid - 3 * a, ai = satom[a];
    const EwaldSystem& Ea = es[mol_idx[ai]];
    if (mol_idx[ai] != s) s_mixed = 1;
    int fl = 0;
    double d = 0.0, w[PME_P], dw[PME_P];
    pme_scaled(xw, ai, ax, Ea, max(1, Ea.mesh[ax]), fl, d);
    bspline8(d, w, dw);
#pragma unroll
    for (int j = 0; j < PME_P; ++j) sw[a][ax][j] = w[j];
    sfl[a][ax] = fl;
  }
  __syncthreads();
  const EwaldSystem& E = es[s];
  const int K0 = E.mesh[0], K1 = E.mesh[1], K2 = E.mesh[2];
  const bool one_system = s_mixed == 0;
  if (one_system && E.mesh_pts != 0 && tid < 3 * na) {
    const int a = tid / 3, ax = tid - 3 * a, K = E.mesh[ax];
    int rel = sfl[a][ax] - sfl[0][ax];  // nearest image of the first atom's base point
    if (2 * rel > K) rel -= K;
    if (2 * rel < -K) rel += K;
    srel[a][ax] = rel;
    atomicMin(&smin[ax], rel);
    atomicMax(&smax[ax], rel);
  }
  __syncthreads();
  unsigned long long* mq = reinterpret_cast<unsigned long long*>(meshq);
  const int j1 = lane >> 3, j2 = lane & 7;
  // (extents only where the min / max were formed: with mixed systems they still hold INT_MAX / INT_MIN - signed overflow)
  const bool spans = one_system && E.mesh_pts != 0;
  const int e0 = spans ? smax[0] - smin[0] + PME_P : 0, e1 = spans ? smax[1] - smin[1] + PME_P : 0, e2 = spans ? smax[2] - smin[2] + PME_P : 0;
  const bool tiled = spans && (long)e0 * e1 * e2 <= (long)PME_TILE;
  if (!tiled) {  // direct form, one wave per atom in turn
    for (int a = wv; a < na; a += 4) {
      const int sa = mol_idx[satom[a]];
      const EwaldSystem& Ea = es[sa];
      if (Ea.mesh_pts == 0) continue;
      const int A1 = Ea.mesh[1], A2 = Ea.mesh[2];
      const double w12 = (double)q[satom[a]] * sw[a][1][j1] * sw[a][2][j2];
      const size_t off12 = (size_t)pme_wrap(sfl[a][1] - (PME_P - 1) + j1, A1) * A2 + pme_wrap(sfl[a][2] - (PME_P - 1) + j2, A2);
      unsigned long long* m = mq + (size_t)sa * max_mesh;
#pragma unroll
      for (int j0 = 0; j0 < PME_P; ++j0) {
        const size_t at = (size_t)pme_wrap(sfl[a][0] - (PME_P - 1) + j0, Ea.mesh[0]) * A1 * A2 + off12;
        atomicAdd(m + at, (unsigned long long)__double2ll_rn(sw[a][0][j0] * w12 * PME_FIX));
      }
    }
    return;
  }
  const int np = e0 * e1 * e2;
  for (int t = tid; t < np; t += 256) tile[t] = 0ull;
  __syncthreads();
  for (int a = wv; a < na; a += 4) {
    const double w12 = (double)q[satom[a]] * sw[a][1][j1] * sw[a][2][j2];
    const int base = ((srel[a][0] - smin[0]) * e1 + (srel[a][1] - smin[1] + j1)) * e2 + (srel[a][2] - smin[2] + j2);
#pragma unroll
    for (int j0 = 0; j0 < PME_P; ++j0)
      atomicAdd(&tile[base + j0 * e1 * e2], (unsigned long long)__double2ll_rn(sw[a][0][j0] * w12 * PME_FIX));
  }
  __syncthreads();
  // tile point (t0, t1, t2) is mesh point first atom's base + min - 7 + t, wrapped
  int o0 = (sfl[0][0] + smin[0] - (PME_P - 1)) % K0, o1 = (sfl[0][1] + smin[1] - (PME_P - 1)) % K1,
      o2 = (sfl[0][2] + smin[2] - (PME_P - 1)) % K2;
  o0 += o0 < 0 ? K0 : 0;
  o1 += o1 < 0 ? K1 : 0;
  o2 += o2 < 0 ? K2 : 0;
  unsigned long long* m = mq + (size_t)s * max_mesh;
  for (int t = tid; t < np; t += 256) {
    const unsigned long long v = tile[t];
    if (v == 0ull) continue;
    const int t0 = t / (e1 * e2), r = t - t0 * (e1 * e2), t1 = r / e2, t2 = r - t1 * e2;
    atomicAdd(m + ((size_t)((o0 + t0) % K0) * K1 + (o1 + t1) % K1) * K2 + (o2 + t2) % K2, v);
  }
}

// The 3-D transform, one axis per launch, as DIRECT sums.  The charge mesh is real, so only the half spectrum i2 <= K2 / 2 is ever
// formed: the work meshes are [K0][K1][H], H = K2 / 2 + 1.
//   MODE 0  axis 2, real (fixed point) -> half spectrum   X[m] = sum_n x[n] e^{-2 pi i m n / K2},  m < H
//   MODE 1  axis 1 or 0, complex, forward                 MODE 2  the same, inverse
//   MODE 3  axis 2, half spectrum -> real                 x[n] = sum_{m < H} c_m Re(X[m] e^{+2 pi i m n / K2}),  c = 1, 2, .., 2, 1
// A block stages T whole lines in LDS (16 KiB: eight blocks per CU; small meshes are latency, not throughput); a thread forms one
// output as a sum over the line, the twiddle advanced by complex multiplications in two independent chains (even / odd terms:
// half the dependent latency; <= 256 steps each, ~1e-14 of drift in double) - one 16-byte LDS read and 6 - 8 fp64 instructions
// per term.
template <int MODE>
__global__ __launch_bounds__(256) void pme_dft_kernel(const EwaldSystem* __restrict__ es, int axis, const void* __restrict__ in_,
                                                     void* __restrict__ out_, size_t max_mesh) {
  const int s = blockIdx.y;
  const EwaldSystem& E = es[s];
  if (E.mesh_pts == 0) return;
  const int K0 = E.mesh[0], K1 = E.mesh[1], K2 = E.mesh[2], H = K2 / 2 + 1;
  const int K = MODE == 0 || MODE == 3 ? K2 : E.mesh[axis];   // transform length (twiddle modulus)
  const int n_in = MODE == 3 ? H : K, n_out = MODE == 0 ? H : K;
  const int inner = MODE == 0 || MODE == 3 ? 1 : (axis == 0 ? K1 * H : H);
  const int n_lines = MODE == 0 || MODE == 3 ? K0 * K1 : (axis == 0 ? K1 * H : K0 * H);
  const int T = max(1, min((256 + n_out - 1) / n_out, PME_LDS_PTS / n_in));  // >= 256 outputs per block where the tile allows
  const int line0 = blockIdx.x * T;
  if (line0 >= n_lines) return;
  const int nt = min(T, n_lines - line0);
  __shared__ double2 xs[PME_LDS_PTS];
  const size_t base_c = (size_t)s * max_mesh;  // (complex meshes are sized like the real one: the half spectrum fits)
  // line -> offset of its first element, in elements of the respective mesh
  auto line_at = [&](int line, int len_axis2) -> size_t {
    if (MODE == 0 || MODE == 3) return (size_t)line * len_axis2;
    return (size_t)(line / inner) * K * inner + (size_t)(line % inner);
  };
  for (int e = threadIdx.x; e < nt * n_in; e += 256) {
    const int t = inner == 1 ? e / n_in : e % nt, n = inner == 1 ? e % n_in : e / nt;
    double2 v;
    if (MODE == 0) {
      v.x = (double)reinterpret_cast<const long long*>(in_)[base_c + line_at(line0 + t, K2) + n] * (1.0 / PME_FIX);
      v.y = 0.0;
    } else {
      v = reinterpret_cast<const double2*>(in_)[base_c + line_at(line0 + t, H) + (size_t)n * inner];
      if (MODE == 3 && n != 0 && 2 * n != K2) v.x *= 2.0, v.y *= 2.0;
    }
    xs[n * T + t] = v;
  }
  __syncthreads();
  for (int o = threadIdx.x; o < nt * n_out; o += 256) {
    const int t = inner == 1 ? o / n_out : o % nt, m = inner == 1 ? o % n_out : o / nt;
    double sn, cs;
    sincospi(2.0 * (double)m / (double)K, &sn, &cs);
    if (MODE == 0 || MODE == 1) sn = -sn;
    const double c2 = cs * cs - sn * sn, s2 = 2.0 * cs * sn;  // two terms ahead
    double ar = 0.0, ai = 0.0, br = 0.0, bi = 0.0, ur = 1.0, ui = 0.0, vr = cs, vi = sn;
    auto term = [&](const double2 x, double wr, double wi, double& accr, double& acci) {
      if (MODE == 0) {
        accr += x.x * wr;
        acci += x.x * wi;
      } else if (MODE == 3) {
        accr += x.x * wr - x.y * wi;
      } else {
        accr += x.x * wr - x.y * wi;
        acci += x.x * wi + x.y * wr;
      }
    };
    int n = 0;
    for (; n + 1 < n_in; n += 2) {
      term(xs[n * T + t], ur, ui, ar, ai);
      term(xs[(n + 1) * T + t], vr, vi, br, bi);
      const double nu = ur * c2 - ui * s2, nv = vr * c2 - vi * s2;
      ui = ur * s2 + ui * c2;
      vi = vr * s2 + vi * c2;
      ur = nu;
      vr = nv;
    }
    if (n < n_in) term(xs[n * T + t], ur, ui, ar, ai);
    ar += br;
    ai += bi;
    if (MODE == 3)
      reinterpret_cast<double*>(out_)[base_c + line_at(line0 + t, K2) + m] = ar;
    else
      reinterpret_cast<double2*>(out_)[base_c + line_at(line0 + t, H) + (size_t)m * inner] = double2{ar, ai};
  }
}

// Q^ <- theta Q^ on the half spectrum, theta(m) = (4 pi / V) exp(-k^2 / 4 alpha^2) / k^2 |b0 b1 b2|^2 (m != 0), and per block the
// sums of c theta |Q^|^2 (vfac k_a k_b [6], 1) for the strain derivative (c = 2 where the mirrored half is not stored)
__global__ __launch_bounds__(256) void pme_influence_kernel(const EwaldSystem* __restrict__ es, double* __restrict__ mesh,
                                                           const double* __restrict__ bmod, double* __restrict__ vpart,
                                                           size_t max_mesh, int max_parts) {
  const int s = blockIdx.y;
  const EwaldSystem& E = es[s];
  const int K0 = E.mesh[0], K1 = E.mesh[1], K2 = E.mesh[2], H = K2 / 2 + 1;
  const int half_pts = K0 * K1 * H;
  if (E.mesh_pts == 0 || (size_t)blockIdx.x * PME_PART >= (size_t)half_pts) return;
  const double* b0 = bmod + ((size_t)s * 3 + 0) * PME_MAX_AXIS;
  const double* b1 = bmod + ((size_t)s * 3 + 1) * PME_MAX_AXIS;
  const double* b2 = bmod + ((size_t)s * 3 + 2) * PME_MAX_AXIS;
  double2* A = reinterpret_cast<double2*>(mesh) + (size_t)s * max_mesh;
  double acc[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int r = 0; r < PME_PART / 256; ++r) {
    const int e = blockIdx.x * PME_PART + r * 256 + threadIdx.x;
    if (e >= half_pts) break;
    const int i0 = e / (K1 * H), rem = e - i0 * (K1 * H), i1 = rem / H, i2 = rem - i1 * H;
    const int m0 = i0 <= K0 / 2 ? i0 : i0 - K0, m1 = i1 <= K1 / 2 ? i1 : i1 - K1, m2 = i2;
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
      const double cw = (i2 == 0 || 2 * i2 == K2) ? 1.0 : 2.0;
      const double ts = cw * theta * (v.x * v.x + v.y * v.y), tv = ts * 2.0 * (1.0 / k2 + (double)E.inv4a2);
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
// derivative (the per-atom rows are only ever summed per system).  The results are added by the lanes that own an output element.
template <bool GRAD, bool STRESS>
__global__ __launch_bounds__(256) void pme_gather_kernel(const float* __restrict__ xw, const float* __restrict__ q,
                                                        const int* __restrict__ mol_idx, const int* __restrict__ mol_start,
                                                        int n_atoms, const EwaldSystem* __restrict__ es,
                                                        const double* __restrict__ pot_, const double* __restrict__ vpart,
                                                        size_t max_mesh, int max_parts, float factor, double* __restrict__ ecoul,
                                                        float* __restrict__ qbar, float* __restrict__ fgrad,
                                                        float* __restrict__ virial_atom) {
  __shared__ PmeSplines sp[4];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + wv;
  const int s = i < n_atoms ? mol_idx[i] : 0;
  const EwaldSystem& E = es[s];
  const bool live = i < n_atoms && E.mesh_pts != 0;
  pme_wave_splines(xw, i, live, E, lane, sp[wv]);
  __syncthreads();
  if (!live) return;
  const PmeSplines& S = sp[wv];
  const int K0 = E.mesh[0], K1 = E.mesh[1], K2 = E.mesh[2];
  const int j1 = lane >> 3, j2 = lane & 7;
  const double a1 = S.w[1][j1], a2 = S.w[2][j2], da1 = S.w[1][8 + j1], da2 = S.w[2][8 + j2];
  const size_t off12 = (size_t)pme_wrap(S.fl[1] - (PME_P - 1) + j1, K1) * K2 + pme_wrap(S.fl[2] - (PME_P - 1) + j2, K2);
  const double* pot = pot_ + (size_t)s * max_mesh;
  double p[PME_P];
#pragma unroll
  for (int j0 = 0; j0 < PME_P; ++j0) p[j0] = pot[(size_t)pme_wrap(S.fl[0] - (PME_P - 1) + j0, K0) * K1 * K2 + off12];
  double s0 = 0.0, sd = 0.0;
#pragma unroll
  for (int j0 = 0; j0 < PME_P; ++j0) {
    s0 += S.w[0][j0] * p[j0];
    if (GRAD) sd += S.w[0][8 + j0] * p[j0];
  }
  double phi = wave_sum(s0 * a1 * a2), g0 = 0.0, g1 = 0.0, g2 = 0.0;
  if (GRAD) {
    g0 = wave_sum(sd * a1 * a2) * (double)K0;
    g1 = wave_sum(s0 * da1 * a2) * (double)K1;
    g2 = wave_sum(s0 * a1 * da2) * (double)K2;
  }
  double W[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  const bool first = STRESS && i == mol_start[s];
  if (first) {  // (wave-uniform)
    const int np = (K0 * K1 * (K2 / 2 + 1) + PME_PART - 1) / PME_PART;
    for (int b = lane; b < np; b += 64)
      for (int c = 0; c < 7; ++c) W[c] += vpart[((size_t)s * max_parts + b) * 8 + c];
    for (int c = 0; c < 7; ++c) W[c] = wave_sum(W[c]);
  }
  const double qi = q[i], f = (double)factor;
  const double phi_all = phi + (double)E.phi_bg;
  if (lane == 0) ecoul[i] += f * qi * phi_all;
  if (!GRAD) return;
  if (lane == 1) qbar[i] += (float)(2.0 * f * phi_all);
  if (lane >= 2 && lane < 5) {
    const int c = lane - 2;
    fgrad[3 * i + c] += (float)(2.0 * f * qi * (E.inv[c * 3] * g0 + E.inv[c * 3 + 1] * g1 + E.inv[c * 3 + 2] * g2));
  }
  if (STRESS && lane >= 8 && lane < 17) {
    const int c = lane - 8, r = c / 3, cc = c % 3;
    const int sym = r == cc ? (r == 0 ? 0 : r == 1 ? 3 : 5) : (r + cc == 1 ? 1 : r + cc == 2 ? 2 : 4);  // index into the 6 sums
    double Wc = 0.0;
#pragma unroll
    for (int t = 0; t < 6; ++t) Wc = sym == t ? W[t] : Wc;
    double v = f * Wc;
    if (r == cc) v += -f * qi * (double)E.phi_bg - f * W[6];  // -delta_ab E: the background's share and the mesh energy
    virial_atom[(size_t)i * 9 + c] += (float)v;
  }
}

}  // namespace

int launch_pme_setup(hipStream_t s, const float* cell, int n_cell, const int* mol_start, const float* charge, int nq, int n_mol,
                     float accuracy, EwaldBuffers& b, int* status) {
  hipLaunchKernelGGL(pme_setup_kernel, dim3(std::max(1, std::min(1024, ceil_div(b.max_mesh, 1024))), n_mol), dim3(256), 0, s, cell, n_cell,
                     mol_start, charge, nq, n_mol, accuracy, b.max_mesh, b.sys, status, b.meshq, b.bmod);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

int launch_pme_recip(hipStream_t s, bool grad, bool stress, const float* xw, const float* q, const int* mol_idx, const int* mol_start,
                     const int* order, int n_atoms, int n_mol, const EwaldBuffers& b, float factor, double* ecoul, float* qbar, float* fgrad, float* virial_atom) {
  const size_t mm = (size_t)b.max_mesh;
  const dim3 blk(256);
  hipLaunchKernelGGL(pme_spread_kernel, dim3(ceil_div(n_atoms, PME_SG)), blk, 0, s, xw, q, mol_idx, order, n_atoms, b.sys, b.meshq,
                     mm);
  AIMNET_LAUNCH_CHECK();
  // blocks of a transform pass = lines / T: a block forms >= 256 outputs (pme_dft_kernel's T)
  const dim3 gdft(b.max_mesh / 256 + 2, n_mol);
  hipLaunchKernelGGL((pme_dft_kernel<0>), gdft, blk, 0, s, b.sys, 2, b.meshq, b.ma, mm);
  AIMNET_LAUNCH_CHECK();
  hipLaunchKernelGGL((pme_dft_kernel<1>), gdft, blk, 0, s, b.sys, 1, b.ma, b.mb, mm);
  AIMNET_LAUNCH_CHECK();
  hipLaunchKernelGGL((pme_dft_kernel<1>), gdft, blk, 0, s, b.sys, 0, b.mb, b.ma, mm);
  AIMNET_LAUNCH_CHECK();
  hipLaunchKernelGGL(pme_influence_kernel, dim3(b.max_parts, n_mol), blk, 0, s, b.sys, b.ma, b.bmod, b.vpart, mm, b.max_parts);
  AIMNET_LAUNCH_CHECK();
  hipLaunchKernelGGL((pme_dft_kernel<2>), gdft, blk, 0, s, b.sys, 0, b.ma, b.mb, mm);
  AIMNET_LAUNCH_CHECK();
  hipLaunchKernelGGL((pme_dft_kernel<2>), gdft, blk, 0, s, b.sys, 1, b.mb, b.ma, mm);
  AIMNET_LAUNCH_CHECK();
  hipLaunchKernelGGL((pme_dft_kernel<3>), gdft, blk, 0, s, b.sys, 2, b.ma, b.meshq, mm);
  AIMNET_LAUNCH_CHECK();
  const dim3 grid(ceil_div(n_atoms, 4));
  const double* pot = reinterpret_cast<const double*>(b.meshq);
  if (grad && stress)
    hipLaunchKernelGGL((pme_gather_kernel<true, true>), grid, blk, 0, s, xw, q, mol_idx, mol_start, n_atoms, b.sys, pot, b.vpart, mm,
                       b.max_parts, factor, ecoul, qbar, fgrad, virial_atom);
  else if (grad)
    hipLaunchKernelGGL((pme_gather_kernel<true, false>), grid, blk, 0, s, xw, q, mol_idx, mol_start, n_atoms, b.sys, pot, b.vpart, mm,
                       b.max_parts, factor, ecoul, qbar, fgrad, virial_atom);
  else
    hipLaunchKernelGGL((pme_gather_kernel<false, false>), grid, blk, 0, s, xw, q, mol_idx, mol_start, n_atoms, b.sys, pot, b.vpart, mm,
                       b.max_parts, factor, ecoul, qbar, fgrad, virial_atom);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

}  // namespace aimnet
