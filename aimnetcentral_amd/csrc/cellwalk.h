// cellwalk.h - per-system cell grid and the wave-per-atom walk over neighbouring bins, shared by the
// neighbour-list builder (nlist.hip) and the list-free DSF Coulomb kernel (model.hip).
#pragma once

#include "common.h"

namespace aimnet {

struct NlistSystem {
  float c[9];    // cell row vectors
  float inv[9];  // inverse: frac = x . inv
  float h[3];    // perpendicular heights
  float o[3];    // origin of the grid: 0 for periodic cells, the corner of the bounding box for non-periodic systems
  float lam;     // Gershgorin bound on lambda_max of the Gram matrix of the slab normals (1 = orthogonal cell)
  int per[3];
  int nb[3];     // bins per lattice axis (slabs of the fractional coordinate)
  int bin_offset;
  int n_bins;
};

// Fractional coordinate k of a position in ONE fixed arithmetic form (explicit fmas, nothing left to the compiler's contraction
// choices): every kernel that wraps or bins an atom gets the same bits, so the separate preparation kernels and the single-launch
// one (nlist.hip, prep_small_kernel) build identical bins.
__device__ __forceinline__ float cell_frac(const NlistSystem& S, float x, float y, float z, int k) {
  return __builtin_fmaf(z, S.inv[6 + k], __builtin_fmaf(y, S.inv[3 + k], x * S.inv[k]));
}

__device__ __forceinline__ int bin_of(const NlistSystem& S, float x, float y, float z, int b[3]) {
  for (int k = 0; k < 3; ++k) {
    const float f = cell_frac(S, x - S.o[0], y - S.o[1], z - S.o[2], k);
    b[k] = max(0, min(S.nb[k] - 1, (int)floorf(f * (float)S.nb[k])));
  }
  return S.bin_offset + (b[0] * S.nb[1] + b[1]) * S.nb[2] + b[2];
}

// Wrap a position into the periodic cell the way the reference does (neighbors.py:265-306: frac = coord @ inv(cell),
// frac -= floor(frac), coord = frac @ cell), in ONE fixed arithmetic form: k-ordered fma chains, what an fp32 matrix product
// with an inner dimension of 3 computes.  The round trip is not the identity in fp32 (an atom inside the cell moves by ~1e-6 A, worth
// 1e-5..1e-4 eV on a hot geometry), so parity with the reference's energies means rounding like the reference here; a form that
// subtracts whole lattice vectors instead (bit-exact for atoms inside the cell) put the `pbc96_dsf8_wrapped` golden 8.9e-5 eV away
// (gate 4.8e-5).
__device__ __forceinline__ void wrap_into_cell(const NlistSystem& S, float x, float y, float z, float w[3]) {
  float f[3];
  for (int k = 0; k < 3; ++k) {
    f[k] = cell_frac(S, x, y, z, k);
    if (S.per[k]) {
      f[k] -= floorf(f[k]);
      if (f[k] >= 1.0f) f[k] = 0.0f;
    }
  }
  for (int c = 0; c < 3; ++c) w[c] = __builtin_fmaf(f[2], S.c[6 + c], __builtin_fmaf(f[1], S.c[3 + c], f[0] * S.c[c]));
}

// status words of a finished list build: max_i count_i into *status_max and "some row overflowed its capacity" into *status_ovf,
// for the 1024 atoms of block b (the words must have been zeroed; one atomic pair per 1024 atoms - 10^4 same-address atomics, one
// per row, would cost ~12 ns each).  Called by nlist_status_kernel or, when no second list follows, as rider blocks of a later launch.
__device__ __forceinline__ void nlist_status_block(const int* __restrict__ cnt_true, int n_atoms, int cap, int* __restrict__ status_max,
                                                   int* __restrict__ status_ovf, int b) {
  __shared__ int s_max[16];
  int v = 0;
  for (int i = b * 1024 + (int)threadIdx.x; i < min(n_atoms, (b + 1) * 1024); i += blockDim.x) v = max(v, cnt_true[i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
  if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) v = max(v, s_max[w]);
    if (v > __hip_atomic_load(status_max, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(status_max, v);
    if (v > cap) atomicMax(status_ovf, 1);
  }
}

// The same for a status array that nobody has zeroed: ONE block reduces all row counts and the per-wave sanity flags of
// launch_mol_start and stores all eight words (0 longest row, 2 overflow, 6 sanity flags, the others 0; word 7 untouched if keep7).
__device__ __forceinline__ void nlist_status_owned_block(const int* __restrict__ cnt_true, int n_atoms, int cap,
                                                         const int* __restrict__ bad_part, int* __restrict__ status, int keep7) {
  __shared__ int s_max[16], s_bad[16];
  int v = 0, f = 0;
  for (int i0 = (int)threadIdx.x; i0 < n_atoms; i0 += 8 * blockDim.x) {  // eight independent loads in flight (one block, 10^4 rows)
    int c[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = i0 + k * (int)blockDim.x;
      c[k] = i < n_atoms ? cnt_true[i] : 0;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) v = max(v, c[k]);
  }
  for (int w = (int)threadIdx.x; w < (n_atoms + 63) / 64; w += blockDim.x) f |= bad_part[w];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    v = max(v, __shfl_xor(v, off, 64));
    f |= __shfl_xor(f, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    s_max[threadIdx.x >> 6] = v;
    s_bad[threadIdx.x >> 6] = f;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {
      v = max(v, s_max[w]);
      f |= s_bad[w];
    }
    status[0] = v; status[1] = 0; status[2] = v > cap ? 1 : 0; status[3] = 0; status[4] = 0; status[5] = 0; status[6] = f;
    if (!keep7) status[7] = 0;
  }
}

// Visit every (neighbour image) candidate of atom i (one wave per atom) whose bin lies within the
// search range of `cutoff`.  f(w, rx, ry, rz, ok, code) is called convergently by all 64 lanes once
// per 64-candidate chunk; w = 4th component of the candidate's stream entry (atom id bits when
// BY_ID, else a payload such as the charge); ok = candidate exists, |r| < cutoff and it is not the
// atom itself (BY_ID: id test in the home image; otherwise r != 0); r is formed as
// (x_j - x_i) + s.C so that (i,j,s) and (j,i,-s) see exactly opposite vectors.
//
// Two phases per batch of 64 (dx, dy) bin columns.  (1) lane = column: prune the column against the cutoff sphere, wrap it
// into the cell, find its z range and the runs of z-contiguous bins that share one lattice shift, and compact the non-empty
// runs {first, last stream entry, shift vector, shift code} into `runbuf` (per-wave LDS, CELLWALK_RUN_INTS ints).  (2) the
// wave consumes the runs one by one, 64 stream entries per chunk.  Steering the walk with scalar loops instead cost ~160
// instructions per column - more than the candidate tests themselves.
constexpr int CELLWALK_RUN_INTS = 64 * 8;

__device__ __forceinline__ int cw_floor_div(int v, int n, float inv_n) {  // exact for |v| < 2^20, n <= 1024 (margin 0.5 / n)
  return (int)floorf(((float)v + 0.5f) * inv_n);
}

template <bool BY_ID, class F>
__device__ __forceinline__ void cell_walk(const NlistSystem& S, int i, float xi, float yi, float zi, float cutoff,
                                          const int* __restrict__ bin_start, const float4* __restrict__ xs, int lane,
                                          int* __restrict__ runbuf, F&& f) {
  // home bin and the atom's position inside it (bin units, [0,1]): the pruning below measures slab distances from the ATOM,
  // not from its bin, which visits ~45 % fewer candidates than the bin-to-bin bound
  int bv[3];
  float fb[3];
  for (int k = 0; k < 3; ++k) {
    const float fr = (xi - S.o[0]) * S.inv[k] + (yi - S.o[1]) * S.inv[3 + k] + (zi - S.o[2]) * S.inv[6 + k];
    const float sc = fr * (float)S.nb[k];
    bv[k] = max(0, min(S.nb[k] - 1, (int)floorf(sc)));
    fb[k] = fminf(fmaxf(sc - (float)bv[k], 0.0f), 1.0f);
  }
  auto U = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
  auto UF = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
  const int nb0 = U(S.nb[0]), nb1 = U(S.nb[1]), nb2 = U(S.nb[2]);
  const int p0 = U(S.per[0]), p1 = U(S.per[1]), p2 = U(S.per[2]);
  // search radius in bins: slab thickness = h_k / nb_k
  // (single-instruction reciprocals / rsq throughout: every bound below carries a 1e-3 bin margin, 1 ulp does not matter)
  const float ih0 = __builtin_amdgcn_rcpf(UF(S.h[0])), ih1 = __builtin_amdgcn_rcpf(UF(S.h[1])), ih2 = __builtin_amdgcn_rcpf(UF(S.h[2]));
  int R0 = (int)ceilf(cutoff * (float)nb0 * ih0 + 1e-3f), R1 = (int)ceilf(cutoff * (float)nb1 * ih1 + 1e-3f),
      R2 = (int)ceilf(cutoff * (float)nb2 * ih2 + 1e-3f);
  R0 = U(min(120, p0 ? R0 : min(R0, nb0 - 1)));
  R1 = U(min(120, p1 ? R1 : min(R1, nb1 - 1)));
  R2 = U(min(120, p2 ? R2 : min(R2, nb2 - 1)));
  const int b0 = U(bv[0]), b1 = U(bv[1]), b2 = U(bv[2]);
  const int boff = U(S.bin_offset);
  const float cutoff2 = cutoff * cutoff;
  const float f0 = UF(fb[0]), f1 = UF(fb[1]), f2 = UF(fb[2]);
  auto gap = [](int d, float fr) {  // slab distance from the atom in bin units
    const float g = d > 0 ? (float)d - fr : d < 0 ? (float)(-d) - 1.0f + fr : 0.0f;
    return fmaxf(g - 1e-3f, 0.0f);
  };
  // Bin pruning.  A point in the slab d_k bins away along axis k is at least
  //   g_k = (d_k - f_k) t_k  (d_k > 0),   (|d_k| - 1 + f_k) t_k  (d_k < 0),   0  (d_k = 0)
  // from the atom along the slab normal n_k (f_k = the atom's position inside its own bin, minus a 1e-3 bin safety margin
  // for the rounding of f_k); with r.n_k = p_k, |r|^2 = p^T (N N^T)^-1 p >= |p|^2 / lambda_max(N N^T), so the bin can hold
  // a neighbour only if g_x^2 + g_y^2 + g_z^2 <= lam * cutoff^2 (rigorous for any cell).
  const float t0 = UF(S.h[0]) * __builtin_amdgcn_rcpf((float)nb0), t1 = UF(S.h[1]) * __builtin_amdgcn_rcpf((float)nb1);
  const float it2 = (float)nb2 * ih2;  // 1 / slab thickness along z
  const float lim2 = UF(S.lam) * cutoff2;
  float c[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) c[k] = UF(S.c[k]);
  const float in0 = __builtin_amdgcn_rcpf((float)nb0), in1 = __builtin_amdgcn_rcpf((float)nb1), in2 = __builtin_amdgcn_rcpf((float)nb2);
  const int W1 = 2 * R1 + 1, ncol = (2 * R0 + 1) * W1;
  const float iW1 = __builtin_amdgcn_rcpf((float)W1);
  const unsigned long long lt_mask = (1ull << lane) - 1ull;

  for (int cb = 0; cb < ncol; cb += 64) {
    // ---- phase 1: this lane's column ------------------------------------------------------------
    const int idx = cb + lane;
    bool live = idx < ncol;
    const int qx = cw_floor_div(idx, W1, iW1);
    const int dx = qx - R0, dy = idx - qx * W1 - R1;
    const float gx = gap(dx, f0) * t0, gy = gap(dy, f1) * t1;
    const float rem = lim2 - gx * gx - gy * gy;
    live = live && rem >= 0.0f;
    int bx = b0 + dx, by = b1 + dy, sx = 0, sy = 0;
    if (p0) {
      sx = cw_floor_div(bx, nb0, in0);
      bx -= sx * nb0;
    } else {
      live = live && bx >= 0 && bx < nb0;
    }
    if (p1) {
      sy = cw_floor_div(by, nb1, in1);
      by -= sy * nb1;
    } else {
      live = live && by >= 0 && by < nb1;
    }
    const float remc = fmaxf(rem, 0.0f);
    const float sz_bins = remc * __builtin_amdgcn_rsqf(fmaxf(remc, 1e-12f)) * it2 + 1e-3f;  // slabs dz < 0 reach (|dz| - 1 + f2), dz > 0 reach (dz - f2)
    const int Rlo = min(R2, (int)floorf(sz_bins + 1.0f - f2)), Rhi = min(R2, (int)floorf(sz_bins + f2));
    // bins along z are contiguous in memory: whole runs [z0, z1] that share one lattice shift sz
    int zlo = b2 - Rlo, zhi = b2 + Rhi, sz_lo = 0, sz_hi = 0;
    if (p2) {
      sz_lo = cw_floor_div(zlo, nb2, in2);
      sz_hi = cw_floor_div(zhi, nb2, in2);
    } else {
      zlo = max(zlo, 0);
      zhi = min(zhi, nb2 - 1);
    }
    const int nruns = live ? sz_hi - sz_lo + 1 : 0;
    const int row_bin = boff + (bx * nb1 + by) * nb2;
    const int max_r = U(wave_max(nruns));
    for (int r = 0; r < max_r; ++r) {
      const int sz = sz_lo + r;
      const int z0 = max(zlo, sz * nb2) - sz * nb2, z1 = min(zhi, sz * nb2 + nb2 - 1) - sz * nb2;
      bool has = r < nruns && z1 >= z0;
      int s0 = 0, s1 = 0;
      if (has) {
        s0 = bin_start[row_bin + z0];
        s1 = bin_start[row_bin + z1 + 1];
        has = s0 < s1;
      }
      const unsigned long long mask = __ballot(has);
      if (has) {
        int* d = runbuf + 8 * __popcll(mask & lt_mask);
        d[0] = s0;
        d[1] = s1;
        d[2] = __float_as_int(sx * c[0] + sy * c[3] + sz * c[6]);
        d[3] = __float_as_int(sx * c[1] + sy * c[4] + sz * c[7]);
        d[4] = __float_as_int(sx * c[2] + sy * c[5] + sz * c[8]);
        d[5] = pack_shift(sx, sy, sz);
      }
      const int n_run = __popcll(mask);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // ---- phase 2: the wave walks the runs -------------------------------------------------------
      for (int k = 0; k < n_run; ++k) {
        const int* d = runbuf + 8 * k;
        const int s0u = U(d[0]), s1u = U(d[1]), code = U(d[5]);
        const float ox = UF(__int_as_float(d[2])), oy = UF(__int_as_float(d[3])), oz = UF(__int_as_float(d[4]));
        const bool self_image = code == 0;
        // one 16-byte stream entry per lane and chunk; the next chunk's entry is requested before this one is consumed
        // (past the run end the index clamps to the run's last entry, masked by e < s1)
        int e = s0u + lane;
        float4 cj = xs[min(e, s1u - 1)];
        for (int base = s0u; base < s1u; base += 64) {
          const float4 nx = xs[min(e + 64, s1u - 1)];
          const float rx = (cj.x - xi) + ox;
          const float ry = (cj.y - yi) + oy;
          const float rz = (cj.z - zi) + oz;
          const float d2 = rx * rx + ry * ry + rz * rz;
          const bool ok = e < s1u && d2 < cutoff2 && (BY_ID ? !(self_image && __float_as_int(cj.w) == i) : d2 > 0.0f);
          f(cj.w, rx, ry, rz, ok, code);
          cj = nx;
          e += 64;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

}  // namespace aimnet
