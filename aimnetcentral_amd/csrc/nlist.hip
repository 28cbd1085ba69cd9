// nlist.hip - full (both-direction) neighbour matrices on gfx950.
//
// Replaces nvalchemiops.torch.neighbors.neighbor_list as the reference calls it
// (aimnet/calculators/neighbors.py:106-125: half_fill=False, same-batch pairs only, rows packed
// real-first, integer PBC shifts, overflow reported) and move_coord_to_cell (neighbors.py:331-381).
//
//   non-periodic : one wave per atom scans its own molecule (atoms of a molecule are contiguous).
//   periodic     : per-system cell list.  bins along lattice axis k are slabs of the fractional
//                  coordinate, at least cutoff/bin_sub thick in perpendicular distance; the search
//                  range R_k = ceil(cutoff / slab thickness) also covers cells smaller than the
//                  cutoff (several images of the same atom, self images included).
// Pair vectors are always formed as (x_j - x_i) + s.C so that (i,j,s) and (j,i,-s) evaluate to
// exactly opposite vectors: the list is exactly symmetric, which the centre-major backward relies on.
// Row order is deterministic: bins are traversed in a fixed order and each bin is sorted by atom id.
#include "cellwalk.h"
#include "common.h"
#include "kernels.h"

namespace aimnet {

size_t nlist_scratch_bytes(int n_atoms, int n_mol) {
  const size_t max_bins = (size_t)n_atoms + 8 * (size_t)n_mol + 8;
  size_t b = 0;
  b += align_up((size_t)(n_mol + 1) * sizeof(int), 256);
  b += align_up((size_t)n_atoms * 3 * sizeof(float), 256);
  b += align_up((size_t)n_mol * sizeof(NlistSystem), 256);
  b += align_up((size_t)n_atoms * sizeof(int), 256);        // atom_bin
  b += 3 * align_up((max_bins + 1) * sizeof(int), 256);     // count, start, fill
  b += 2 * align_up((size_t)n_atoms * sizeof(int), 256);    // sorted_tmp, sorted
  b += 2 * align_up((size_t)n_atoms * sizeof(float4), 256); // xs: bin-ordered (x, y, z, atom id); xq: (x, y, z, charge)
  b += align_up((size_t)n_atoms * sizeof(int), 256);        // mol_c: molecule index of every atom clamped to [0, n_mol)
  return b;
}

size_t nlist_xw_offset(int n_mol) { return align_up((size_t)(n_mol + 1) * sizeof(int), 256); }

void nlist_carve(NlistBuffers& b, char* p, int n_atoms, int n_mol) {
  const size_t max_bins = (size_t)n_atoms + 8 * (size_t)n_mol + 8;
  auto take = [&](size_t bytes) {
    char* r = p;
    p += align_up(bytes, 256);
    return r;
  };
  b.mol_start = (int*)take((size_t)(n_mol + 1) * sizeof(int));
  b.xw = (float*)take((size_t)n_atoms * 3 * sizeof(float));
  b.sys = (void*)take((size_t)n_mol * sizeof(NlistSystem));
  b.atom_bin = (int*)take((size_t)n_atoms * sizeof(int));
  b.bin_count = (int*)take((max_bins + 1) * sizeof(int));
  b.bin_start = (int*)take((max_bins + 1) * sizeof(int));
  b.bin_fill = (int*)take((max_bins + 1) * sizeof(int));
  b.sorted_tmp = (int*)take((size_t)n_atoms * sizeof(int));
  b.sorted = (int*)take((size_t)n_atoms * sizeof(int));
  b.xs = (float4*)take((size_t)n_atoms * sizeof(float4));
  b.sorted_tmp_xq = (void*)take((size_t)n_atoms * sizeof(float4));
  b.mol_c = (int*)take((size_t)n_atoms * sizeof(int));
}

// ------------------------------------------------------------------------------------------------
// Also the input sanity pass (numbers / bad may be NULL): the engine clamps atomic numbers to the 64 embedding rows and molecule
// indices to [0, n_mol) for memory safety; *bad gets bit 0 when an atomic number lies outside [0, 63] (the reference's
// nn.Embedding(64) raises an index error there, core.py:49) and bit 1 when a molecule index lies outside [0, n_mol)
// (the reference fails in mol_sum / index_add, nbops.py:309-377).  HipEngine.eval turns either into a ValueError.
// Third job (slot_of_z may be NULL): the species pass of the engine - aslot[i] = model slot of atom i's element and, per block of
// 256 atoms, the mask of the slots present (present_part[blockIdx.x]); it reads the same `numbers` and saves a launch.
__device__ void cell_bins_setup_block(const CellSetupRider& r, const int* __restrict__ mol_start, int n_mol, const int* __restrict__ mol_idx,
                                      int n_atoms);

__global__ __launch_bounds__(256) void mol_start_kernel(const int* __restrict__ mol_idx, const int* __restrict__ numbers,
                                                        int n_atoms, int n_mol, int* __restrict__ mol_start,
                                                        int* __restrict__ bad, const int* __restrict__ slot_of_z,
                                                        int* __restrict__ aslot, unsigned long long* __restrict__ present_part,
                                                        int* __restrict__ mol_c, CellSetupRider cs, int* __restrict__ bad_part) {
  if (cs.sys && blockIdx.x == 0) {  // rider: cell + bin-grid setup of the periodic fast path (cell_bins_setup_kernel),
    // independent of this launch's output (the atom counts come from a binary search in mol_idx): a kernel boundary less.  FIRST
    // block of the grid: it is the longest one (fp64 cell inverse, zeroing of the bin counters)
    cell_bins_setup_block(cs, nullptr, n_mol, mol_idx, n_atoms);
    return;
  }
  const int blk = (int)blockIdx.x - (cs.sys ? 1 : 0);
  const int i = blk * blockDim.x + threadIdx.x;
  if (slot_of_z) {  // (block-uniform)
    __shared__ unsigned long long s_mask;
    if (threadIdx.x == 0) s_mask = 0ull;
    __syncthreads();
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
    if (threadIdx.x == 0) present_part[blk] = s_mask;
  }
  if (i >= n_atoms) return;
  const int prev = (i == 0) ? -1 : min(max(mol_idx[i - 1], -1), n_mol - 1);
  const int raw = mol_idx[i];
  const int cur = min(max(raw, 0), n_mol - 1);
  mol_c[i] = cur;  // every later kernel indexes per-molecule data through this clamped copy, never through the caller's array
  if (bad) {
    int f = (raw < 0 || raw >= n_mol) ? 2 : 0;
    if (i > 0 && raw < mol_idx[i - 1]) f |= 4;  // not sorted: molecules must be contiguous
    if (numbers) {
      const int z = numbers[i];
      if (z < 0 || z > 63) f |= 1;
    }
    if (bad_part) {  // one plain store per wave instead of atomics into a zeroed word: nothing has to be zeroed beforehand
      // (the reader, nlist_status_owned_block, ORs the ceil(n_atoms / 64) slots)
      const int wf = (__ballot(f & 1) ? 1 : 0) | (__ballot(f & 2) ? 2 : 0) | (__ballot(f & 4) ? 4 : 0);
      if ((threadIdx.x & 63) == 0) bad_part[i >> 6] = wf;
    } else if (f) {
      atomicOr(bad, f);
    }
  }
  for (int m = prev + 1; m <= cur && m <= n_mol; ++m) mol_start[m] = i;
  if (i == n_atoms - 1)
    for (int m = cur + 1; m <= n_mol; ++m) mol_start[m] = n_atoms;
}

int launch_mol_start(hipStream_t s, const int* mol_idx, int n_atoms, int n_mol, int* mol_start, int* mol_c, const int* numbers,
                     int* bad, const int* slot_of_z, int* aslot, unsigned long long* present_part, const CellSetupRider* cell_setup,
                     int* bad_part) {
  CellSetupRider cs{};
  if (cell_setup) cs = *cell_setup;
  hipLaunchKernelGGL(mol_start_kernel, dim3(ceil_div(n_atoms, 256) + (cs.sys ? 1 : 0)), dim3(256), 0, s, mol_idx, numbers, n_atoms, n_mol,
                     mol_start, bad, (numbers && aslot) ? slot_of_z : nullptr, aslot, present_part, mol_c, cs, bad ? bad_part : nullptr);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// pbc_sys (may be NULL): per-system periodicity flags [n_cell][3] (normalize_pbc, neighbors.py:309-321) overriding p0..p2
__device__ void cell_setup_one(const float* __restrict__ cell, int n_cell, int s, int p0, int p1, int p2,
                               const int* __restrict__ pbc_sys, NlistSystem* __restrict__ sys) {
  const float* c = cell + (n_cell == 1 ? 0 : (size_t)s * 9);
  NlistSystem S;
  double m[9];
  for (int k = 0; k < 9; ++k) {
    S.c[k] = c[k];
    m[k] = c[k];
  }
  const double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
                     m[2] * (m[3] * m[7] - m[4] * m[6]);
  const double id = 1.0 / det;
  double inv[9];
  inv[0] = (m[4] * m[8] - m[5] * m[7]) * id;
  inv[1] = (m[2] * m[7] - m[1] * m[8]) * id;
  inv[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  inv[3] = (m[5] * m[6] - m[3] * m[8]) * id;
  inv[4] = (m[0] * m[8] - m[2] * m[6]) * id;
  inv[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  inv[6] = (m[3] * m[7] - m[4] * m[6]) * id;
  inv[7] = (m[1] * m[6] - m[0] * m[7]) * id;
  inv[8] = (m[0] * m[4] - m[1] * m[3]) * id;
  for (int k = 0; k < 9; ++k) S.inv[k] = (float)inv[k];
  double nrm[3][3];
  for (int k = 0; k < 3; ++k) {
    const double* a1 = m + 3 * ((k + 1) % 3);
    const double* a2 = m + 3 * ((k + 2) % 3);
    const double cx = a1[1] * a2[2] - a1[2] * a2[1];
    const double cy = a1[2] * a2[0] - a1[0] * a2[2];
    const double cz = a1[0] * a2[1] - a1[1] * a2[0];
    const double len = sqrt(cx * cx + cy * cy + cz * cz);
    S.h[k] = (float)(fabs(det) / len);
    nrm[k][0] = cx / len; nrm[k][1] = cy / len; nrm[k][2] = cz / len;
    S.nb[k] = 1;
  }
  double lam = 1.0;  // Gershgorin: lambda_max(N N^T) <= 1 + max_i sum_{j != i} |n_i . n_j|
  for (int i2 = 0; i2 < 3; ++i2) {
    double off = 0.0;
    for (int j2 = 0; j2 < 3; ++j2)
      if (j2 != i2) off += fabs(nrm[i2][0] * nrm[j2][0] + nrm[i2][1] * nrm[j2][1] + nrm[i2][2] * nrm[j2][2]);
    lam = fmax(lam, 1.0 + off);
  }
  S.lam = (float)(lam * 1.0001);
  S.o[0] = S.o[1] = S.o[2] = 0.0f;
  const int* ps = pbc_sys ? pbc_sys + (n_cell == 1 ? 0 : (size_t)s * 3) : nullptr;
  S.per[0] = ps ? (ps[0] != 0) : p0;
  S.per[1] = ps ? (ps[1] != 0) : p1;
  S.per[2] = ps ? (ps[2] != 0) : p2;
  S.bin_offset = 0;
  S.n_bins = 1;
  sys[s] = S;
}

__global__ void cell_setup_kernel(const float* __restrict__ cell, int n_cell, int n_mol, int p0, int p1, int p2,
                                  const int* __restrict__ pbc_sys, NlistSystem* __restrict__ sys) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_mol) return;
  cell_setup_one(cell, n_cell, s, p0, p1, p2, pbc_sys, sys);
}

__global__ __launch_bounds__(256) void bbox_setup_kernel(const float* __restrict__ coord, const int* __restrict__ mol_start,
                                                        NlistSystem* __restrict__ sys) {
  const int s = blockIdx.x;
  const int j0 = mol_start[s], j1 = mol_start[s + 1];
  float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (int j = j0 + threadIdx.x; j < j1; j += 256)
    for (int k = 0; k < 3; ++k) {
      const float v = coord[3 * j + k];
      lo[k] = fminf(lo[k], v);
      hi[k] = fmaxf(hi[k], v);
    }
  __shared__ float s_lo[4][3], s_hi[4][3];
  for (int k = 0; k < 3; ++k) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo[k] = fminf(lo[k], __shfl_xor(lo[k], off, 64));
      hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
      s_lo[threadIdx.x >> 6][k] = lo[k];
      s_hi[threadIdx.x >> 6][k] = hi[k];
    }
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  NlistSystem S;
  for (int k = 0; k < 9; ++k) S.c[k] = S.inv[k] = 0.0f;
  for (int k = 0; k < 3; ++k) {
    const float a = fminf(fminf(s_lo[0][k], s_lo[1][k]), fminf(s_lo[2][k], s_lo[3][k]));
    const float b = fmaxf(fmaxf(s_hi[0][k], s_hi[1][k]), fmaxf(s_hi[2][k], s_hi[3][k]));
    const float L = (j1 > j0 ? b - a : 0.0f) + 0.02f;  // 0.01 A of margin on either side
    S.o[k] = (j1 > j0 ? a : 0.0f) - 0.01f;
    S.c[4 * k] = L;
    S.inv[4 * k] = 1.0f / L;
    S.h[k] = L;
    S.per[k] = 0;
    S.nb[k] = 1;
  }
  S.lam = 1.0001f;
  S.bin_offset = 0;
  S.n_bins = 1;
  sys[s] = S;
}

__global__ void wrap_kernel(const float* __restrict__ coord, const int* __restrict__ mol_idx, int n_atoms,
                            const NlistSystem* __restrict__ sys, float* __restrict__ xw) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_atoms) return;
  const float x = coord[3 * i], y = coord[3 * i + 1], z = coord[3 * i + 2];
  if (sys == nullptr) {
    xw[3 * i] = x;
    xw[3 * i + 1] = y;
    xw[3 * i + 2] = z;
    return;
  }
  float w[3];
  wrap_into_cell(sys[mol_idx[i]], x, y, z, w);
  for (int k = 0; k < 3; ++k) xw[3 * i + k] = w[k];
}

// (defined with the binning kernels below)
__global__ void cell_bins_setup_kernel(CellSetupRider r, const int* __restrict__ mol_start, int n_mol);
__global__ void wrap_bin_count_kernel(const float* __restrict__ coord, const int* __restrict__ mol_idx, int n_atoms,
                                      const NlistSystem* __restrict__ sys, float* __restrict__ xw, int* __restrict__ atom_bin,
                                      int* __restrict__ bin_count, int* __restrict__ slot);

// bin_width > 0 (periodic systems only): the bins of the following launch_bins(width = bin_width) are prepared on the way -
// cell + bin-grid setup in one launch, wrapping + bin counting in one launch; launch_bins then starts at the scan.
bool cell_setup_rides(int n_atoms, int n_mol) { return n_mol <= 4096 && n_atoms + 8 * n_mol + 9 <= 65536; }

CellSetupRider cell_setup_rider(const float* cell, int n_cell, const int pbc[3], const int* pbc_sys, float bin_width, int n_atoms,
                                int n_mol, NlistBuffers& b) {
  CellSetupRider r;
  r.cell = cell; r.n_cell = n_cell; r.p0 = pbc[0]; r.p1 = pbc[1]; r.p2 = pbc[2]; r.pbc_sys = pbc_sys; r.sys = b.sys; r.w = bin_width;
  r.bin_count = b.bin_count;
  r.n_zero = n_atoms + 8 * n_mol + 9;
  return r;
}

int launch_wrap(hipStream_t s, const float* coord, const int* mol_idx, int n_atoms, int n_mol, const float* cell,
                int n_cell, const int pbc[3], NlistBuffers& b, const int* pbc_sys, float bin_width, bool setup_done) {
  NlistSystem* sys = nullptr;
  b.prebinned_width = 0.0f;
  if (cell != nullptr && bin_width > 0.0f && n_mol <= 4096) {
    sys = (NlistSystem*)b.sys;
    const int max_bins = n_atoms + 8 * n_mol + 8;
    // the single setup block also zeroes the bin counters - fine for the 10^4 entries of a 10 k-atom system (it saves a launch),
    // a serial tail on the critical path at 10^6: above 64 k counters a memset does it at full width
    if (!setup_done) {  // (setup_done: the setup block rode on launch_mol_start, CellSetupRider)
      int n_zero = max_bins + 1;
      if (n_zero > 65536) {
        AIMNET_HIP_CHECK(hipMemsetAsync(b.bin_count, 0, (size_t)n_zero * sizeof(int), s));
        n_zero = 0;
      }
      CellSetupRider r = cell_setup_rider(cell, n_cell, pbc, pbc_sys, bin_width, n_atoms, n_mol, b);
      r.n_zero = n_zero;
      hipLaunchKernelGGL(cell_bins_setup_kernel, dim3(1), dim3(256), 0, s, r, b.mol_start, n_mol);
      AIMNET_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(wrap_bin_count_kernel, dim3(ceil_div(n_atoms, 256)), dim3(256), 0, s, coord, mol_idx, n_atoms, sys, b.xw,
                       b.atom_bin, b.bin_count, b.bin_fill);
    AIMNET_LAUNCH_CHECK();
    b.binned = true;
    b.prebinned_width = bin_width;
    return 0;
  }
  if (cell != nullptr) {
    sys = (NlistSystem*)b.sys;
    hipLaunchKernelGGL(cell_setup_kernel, dim3(ceil_div(n_mol, 64)), dim3(64), 0, s, cell, n_cell, n_mol, pbc[0], pbc[1],
                       pbc[2], pbc_sys, sys);
    AIMNET_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(wrap_kernel, dim3(ceil_div(n_atoms, 256)), dim3(256), 0, s, coord, mol_idx, n_atoms, sys, b.xw);
  AIMNET_LAUNCH_CHECK();
  b.binned = cell != nullptr;
  return 0;
}

int launch_bbox(hipStream_t s, int n_mol, NlistBuffers& b) {
  hipLaunchKernelGGL(bbox_setup_kernel, dim3(n_mol), dim3(256), 0, s, b.xw, b.mol_start, (NlistSystem*)b.sys);
  AIMNET_LAUNCH_CHECK();
  b.binned = true;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// non-periodic: one wave per atom, lanes over the atoms of the same molecule
// pair geometry of ops.calc_distances (ops.py:37-66) as the conv kernels consume it: (r / d, d); written by the row builders
// of the short-range list themselves (the pair vector is in registers there), so no separate pass re-reads the list
__device__ __forceinline__ float4 unit_and_norm(float rx, float ry, float rz) {
  const float d = sqrtf(rx * rx + ry * ry + rz * rz);
  const float inv = 1.0f / d;
  return make_float4(rx * inv, ry * inv, rz * inv, d);
}

__global__ __launch_bounds__(256) void nlist_brute_kernel(const float* __restrict__ xw, const int* __restrict__ mol_idx,
                                                         const int* __restrict__ mol_start, int n_atoms, float cutoff2,
                                                         int cap, int fill_value, int fill_rows, int* __restrict__ nb_idx,
                                                         int* __restrict__ nb_cnt, int* __restrict__ cnt_true,
                                                         float4* __restrict__ pg) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_atoms) return;
  const int m = mol_idx[i];
  const int j0 = mol_start[m], j1 = mol_start[m + 1];
  const float xi = xw[3 * i], yi = xw[3 * i + 1], zi = xw[3 * i + 2];
  int* row = nb_idx + (size_t)i * cap;
  int count = 0;
  for (int base = j0; base < j1; base += 64) {
    const int j = base + lane;
    bool ok = false;
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (j < j1 && j != i) {
      dx = xw[3 * j] - xi, dy = xw[3 * j + 1] - yi, dz = xw[3 * j + 2] - zi;
      ok = (dx * dx + dy * dy + dz * dz) < cutoff2;
    }
    const unsigned long long mask = __ballot(ok);
    const int pos = count + __popcll(mask & ((1ull << lane) - 1ull));
    if (ok && pos < cap) {
      row[pos] = j;
      if (pg) pg[(size_t)i * cap + pos] = unit_and_norm(dx, dy, dz);
    }
    count += __popcll(mask);
  }
  if (fill_rows)
    for (int p = min(count, cap) + lane; p < cap; p += 64) row[p] = fill_value;
  if (lane == 0) {
    nb_cnt[i] = min(count, cap);
    cnt_true[i] = count;  // max / overflow are reduced by nlist_status_kernel: 10^4 same-address atomics cost ~12 ns each
  }
}

// (the launcher of this kernel is launch_nlist below)

// periodic: choose the bin grid of every system for this cutoff, then a serial prefix of bin offsets
// first atom whose (clamped) molecule index is >= m, in a sorted mol_idx (what mol_start[m] holds once launch_mol_start has run)
__device__ __forceinline__ int first_atom_of(const int* __restrict__ mol_idx, int n_atoms, int n_mol, int m) {
  if (m <= 0) return 0;  // (a single system never searches: 14 dependent loads would be the longest thing in the launch)
  int lo = 0, hi = n_atoms;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (min(max(mol_idx[mid], 0), n_mol - 1) < m) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// mol_start == NULL: the atom counts come from a binary search in the caller's (sorted) mol_idx - the form that rides on the
// launch that is still writing mol_start
__device__ void bins_setup_block(NlistSystem* __restrict__ sys, const int* __restrict__ mol_start, int n_mol, float w,
                                 const int* __restrict__ mol_idx = nullptr, int n_atoms = 0) {
  for (int s = threadIdx.x; s < n_mol; s += blockDim.x) {
    NlistSystem S = sys[s];
    const int ns = mol_start ? mol_start[s + 1] - mol_start[s]
                             : (s + 1 < n_mol ? first_atom_of(mol_idx, n_atoms, n_mol, s + 1) : n_atoms) - first_atom_of(mol_idx, n_atoms, n_mol, s);
    // (an unsorted mol_idx can make the racing mol_start differences negative: the clamp keeps the loop below finite - every nb
    // reaches 1 - and the input-sanity status bit reports the batch as unusable)
    const long cap_bins = ns + 8 > 1 ? (long)ns + 8 : 1;
    int nb[3];
    for (int k = 0; k < 3; ++k) nb[k] = max(1, min(1024, (int)floorf(S.h[k] / w)));
    while ((long)nb[0] * nb[1] * nb[2] > cap_bins) {
      const float f = cbrtf((float)cap_bins / (float)((long)nb[0] * nb[1] * nb[2]));
      int big = 0;
      for (int k = 0; k < 3; ++k) {
        const int t = max(1, (int)floorf(nb[k] * f));
        if (nb[k] > nb[big]) big = k;
        nb[k] = t;
      }
      if ((long)nb[0] * nb[1] * nb[2] > cap_bins && nb[big] > 1) nb[big] -= 1;
    }
    for (int k = 0; k < 3; ++k) S.nb[k] = nb[k];
    S.n_bins = nb[0] * nb[1] * nb[2];
    sys[s] = S;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int off = 0;
    for (int s = 0; s < n_mol; ++s) {
      sys[s].bin_offset = off;
      off += sys[s].n_bins;
    }
  }
}

__global__ void bins_setup_kernel(NlistSystem* __restrict__ sys, const int* __restrict__ mol_start, int n_mol,
                                  float w) {
  bins_setup_block(sys, mol_start, n_mol, w);
}

// periodic fast path (launch_wrap with a bin width): cell setup, bin grids and the zeroing of the bin counters in ONE
// single-block launch (these are a few microseconds of work each; as separate kernels they cost a launch latency apiece)
__device__ void cell_bins_setup_block(const CellSetupRider& r, const int* __restrict__ mol_start, int n_mol, const int* __restrict__ mol_idx,
                                      int n_atoms) {
  NlistSystem* sys = (NlistSystem*)r.sys;
  for (int s = threadIdx.x; s < n_mol; s += blockDim.x) cell_setup_one(r.cell, r.n_cell, s, r.p0, r.p1, r.p2, r.pbc_sys, sys);
  for (int k = threadIdx.x; k < r.n_zero; k += blockDim.x) r.bin_count[k] = 0;
  __syncthreads();
  bins_setup_block(sys, mol_start, n_mol, r.w, mol_idx, n_atoms);
}

__global__ void cell_bins_setup_kernel(CellSetupRider r, const int* __restrict__ mol_start, int n_mol) {
  cell_bins_setup_block(r, mol_start, n_mol, nullptr, 0);
}

// slot[i] = arrival rank of atom i inside its bin (any order: bin_sort_kernel orders every bin by atom id afterwards)
__global__ void bin_count_kernel(const float* __restrict__ xw, const int* __restrict__ mol_idx, int n_atoms,
                                 const NlistSystem* __restrict__ sys, int* __restrict__ atom_bin,
                                 int* __restrict__ bin_count, int* __restrict__ slot) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_atoms) return;
  int b[3];
  const int bin = bin_of(sys[mol_idx[i]], xw[3 * i], xw[3 * i + 1], xw[3 * i + 2], b);
  atom_bin[i] = bin;
  slot[i] = atomicAdd(&bin_count[bin], 1);
}

// wrap + bin_count in one pass over the atoms (periodic fast path)
__global__ void wrap_bin_count_kernel(const float* __restrict__ coord, const int* __restrict__ mol_idx, int n_atoms,
                                      const NlistSystem* __restrict__ sys, float* __restrict__ xw,
                                      int* __restrict__ atom_bin, int* __restrict__ bin_count, int* __restrict__ slot) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_atoms) return;
  const NlistSystem& S = sys[mol_idx[i]];
  const float x = coord[3 * i], y = coord[3 * i + 1], z = coord[3 * i + 2];
  float w[3];
  wrap_into_cell(S, x, y, z, w);
  xw[3 * i] = w[0];
  xw[3 * i + 1] = w[1];
  xw[3 * i + 2] = w[2];
  int b[3];
  const int bin = bin_of(S, w[0], w[1], w[2], b);
  atom_bin[i] = bin;
  slot[i] = atomicAdd(&bin_count[bin], 1);
}

// bins in use = end of the last system's grid (the buffers are sized for the worst case, one bin per atom)
__device__ __forceinline__ int bins_in_use(const NlistSystem* __restrict__ sys, int n_mol) {
  return sys[n_mol - 1].bin_offset + sys[n_mol - 1].n_bins;
}

// single-block exclusive scan over the bins in use: each thread owns a contiguous chunk
__global__ __launch_bounds__(1024) void scan_kernel(const int* __restrict__ in, int* __restrict__ out,
                                                   const NlistSystem* __restrict__ sys, int n_mol) {
  __shared__ int part[1024];
  const int n = bins_in_use(sys, n_mol);
  const int t = threadIdx.x;
  const int chunk = (n + 1023) / 1024;
  const int lo = min(n, t * chunk), hi = min(n, lo + chunk);
  int s = 0;
  for (int k = lo; k < hi; ++k) s += in[k];
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = (t == 0) ? 0 : part[t - 1];
  for (int k = lo; k < hi; ++k) {
    out[k] = run;
    run += in[k];
  }
  if (t == 1023) out[n] = part[1023];
}

__global__ void bin_fill_kernel(const int* __restrict__ atom_bin, const int* __restrict__ bin_start, int n_atoms,
                                const int* __restrict__ slot, int* __restrict__ sorted_tmp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_atoms) return;
  sorted_tmp[bin_start[atom_bin[i]] + slot[i]] = i;
}

// order every bin by atom id (rank by counting; bins hold tens of atoms): one wave per bin
// also emits the bin-ordered coordinate stream xs[k] = (x, y, z, atom id): the row builder then reads
// its candidates as ONE coalesced 16-byte load per lane instead of an index load + 3 scattered dwords
__global__ __launch_bounds__(256) void bin_sort_kernel(const int* __restrict__ bin_start, const NlistSystem* __restrict__ sys,
                                                      int n_mol, const int* __restrict__ sorted_tmp,
                                                      const float* __restrict__ xw, int* __restrict__ sorted,
                                                      float4* __restrict__ xs) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= bins_in_use(sys, n_mol)) return;
  const int s0 = bin_start[b], n = bin_start[b + 1] - s0;
  for (int e = lane; e < n; e += 64) {
    const int v = sorted_tmp[s0 + e];
    int rank = 0;
    for (int f = 0; f < n; ++f) rank += (sorted_tmp[s0 + f] < v) ? 1 : 0;
    sorted[s0 + rank] = v;
    xs[s0 + rank] = make_float4(xw[3 * v], xw[3 * v + 1], xw[3 * v + 2], __int_as_float(v));
  }
}

// ---- small periodic batches: the whole preparation in ONE single-block launch ----------------------------------------------
// status zeroing, mol_start + input sanity + species pass, cell + bin-grid setup, wrapping, bin counting, scan, fill and the
// per-bin ordering are a few microseconds of work each on 10^4 atoms, but seven dependent launches cost ~5 us apiece on this part
// (profiles/r4_kernel_sequence.txt).  One block of 1024 threads does them back to back with block barriers in between; bin
// counters / starts and the arrival-ordered atom ids live in LDS.  Results are identical to the separate kernels: every bin is
// ordered by atom id at the end, whatever order the LDS atomics arrived in.
// Size limit: one CU's address coalescer handles about one cache line per clock, and the three scattered passes (stride-12
// coordinates, bin-ordered stores) of 10^4 atoms take 25 000 clocks on it - the phases measure 2.5 / 5 / 1.3 / 10 / 2 / 1.3 / 12 us at
// 10 080 atoms (tests/tools/prep_timing.sh), 43 us against 36.5 us for the seven launches.  Step time of periodic glucose
// supercells, this kernel against the separate ones (tests/tools/prep_ab.py): 96 atoms -4.2 %, 768 -2.8 %, 1 728 -2.4 %,
// 3 456 -0.8 %, 6 144 +0.4 %, 10 080 +1.5 %.  Hence 4 096 atoms.
constexpr int PREP_SMALL_MAX_ATOMS = 4096;
constexpr int PREP_SMALL_MAX_MOL = 64;

struct PrepSmallArgs {
  const float* coord; const int* mol_idx; const int* numbers; int n_atoms, n_mol;
  const float* cell; int n_cell, p0, p1, p2; const int* pbc_sys; float w;
  int* status;  // [8], zeroed here; status + 6 = the input sanity flags
  const int* slot_of_z; int* aslot; unsigned long long* present_part;
  int* mol_start; int* mol_c; NlistSystem* sys; float* xw; int* bin_start; int* sorted; float4* xs;
};

// Every thread owns the atoms t, t + 1024, ... (at most PS_K of them) through all phases, with their molecule, bin and arrival
// rank in registers; every phase issues its global loads for all owned atoms before the first use (a dependent load chain per
// atom and iteration made the first version of this kernel slower than the seven launches it replaces).
constexpr int PS_K = PREP_SMALL_MAX_ATOMS / 1024;

__device__ __forceinline__ unsigned wave_or_to_lane63(unsigned v) {  // OR over the wave, valid in lane 63
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);  // row_shr:1
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);  // row_shr:2
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);  // row_shr:4
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);  // row_shr:8: lane 15 of a row = the row
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);  // row_bcast:15 into rows 1, 3
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);  // row_bcast:31 into rows 2, 3
  return v;
}
#ifdef AIMNET_PREP_TIMING  // measurement build (tests/tools/prep_timing.sh): wall-clock stamps (100 MHz) of thread 0 per phase
__device__ unsigned long long g_prep_stamps[16];
#define PREP_STAMP(k) do { if (threadIdx.x == 0) g_prep_stamps[k] = wall_clock64(); } while (0)
int prep_read_stamps(unsigned long long* host16) {
  return hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_prep_stamps), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#else
#define PREP_STAMP(k)
#endif

__global__ __launch_bounds__(1024) void prep_small_kernel(PrepSmallArgs a) {
  extern __shared__ int s_prep[];
  const int t = threadIdx.x, lane = t & 63;
  const int n_atoms = a.n_atoms, n_mol = a.n_mol;
  const int max_bins = n_atoms + 8 * n_mol + 8;
  int* s_bin = s_prep;                 // [max_bins + 1] counters, then (in place) starts
  int* s_tmp = s_prep + max_bins + 1;  // [n_atoms] atom ids in arrival order per bin
  __shared__ int part[1024];
  __shared__ unsigned long long s_mask[PREP_SMALL_MAX_ATOMS / 256];
  __shared__ int s_slot_of_z[64];
  __shared__ NlistSystem s_sys[PREP_SMALL_MAX_MOL];
  PREP_STAMP(0);
  if (t < 8) a.status[t] = 0;
  if (t < PREP_SMALL_MAX_ATOMS / 256) s_mask[t] = 0ull;
  if (t < 64) s_slot_of_z[t] = a.slot_of_z ? a.slot_of_z[t] : 0;
  if (a.cell)  // (molecules: no bins, no dynamic LDS)
    for (int k = t; k <= max_bins; k += 1024) s_bin[k] = 0;
  if (a.cell)
    for (int s = t; s < n_mol; s += 1024) cell_setup_one(a.cell, a.n_cell, s, a.p0, a.p1, a.p2, a.pbc_sys, a.sys);
  __syncthreads();
  PREP_STAMP(1);
  // ---- mol_start_kernel's three jobs
  int cur[PS_K];
  {
    int zraw[PS_K], raw[PS_K], rprev[PS_K];
#pragma unroll
    for (int k = 0; k < PS_K; ++k) {
      const int i = t + 1024 * k, ii = i < n_atoms ? i : 0;
      zraw[k] = a.numbers ? a.numbers[ii] : 0;
      raw[k] = a.mol_idx[ii];
      rprev[k] = a.mol_idx[ii > 0 ? ii - 1 : 0];
    }
    int* bad = a.status + 6;
#pragma unroll
    for (int k = 0; k < PS_K; ++k) {
      const int i = t + 1024 * k;
      if (1024 * k >= n_atoms) break;  // (block-uniform)
      unsigned long long m = 0ull;
      cur[k] = 0;
      if (i < n_atoms) {
        if (a.slot_of_z) {
          const int sl = s_slot_of_z[min(63, max(0, zraw[k]))];
          a.aslot[i] = sl;
          m = 1ull << sl;
        }
        const int prev = (i == 0) ? -1 : min(max(rprev[k], -1), n_mol - 1);
        cur[k] = min(max(raw[k], 0), n_mol - 1);
        a.mol_c[i] = cur[k];
        int f = (raw[k] < 0 || raw[k] >= n_mol) ? 2 : 0;
        if (i > 0 && raw[k] < rprev[k]) f |= 4;
        if (a.numbers && (zraw[k] < 0 || zraw[k] > 63)) f |= 1;
        if (f) atomicOr(bad, f);
        for (int mm = prev + 1; mm <= cur[k] && mm <= n_mol; ++mm) a.mol_start[mm] = i;
        if (i == n_atoms - 1)
          for (int mm = cur[k] + 1; mm <= n_mol; ++mm) a.mol_start[mm] = n_atoms;
      }
      if (a.slot_of_z) {  // a wave's 64 atoms lie in one 256-atom group; DPP reduction (a ds_bpermute chain per k cost 0.7 us)
        const unsigned lo = wave_or_to_lane63((unsigned)m), hi = wave_or_to_lane63((unsigned)(m >> 32));
        if (lane == 63) atomicOr(&s_mask[(1024 * k + (t & ~63)) >> 8], ((unsigned long long)hi << 32) | lo);
      }
    }
  }
  __syncthreads();
  PREP_STAMP(2);
  if (a.slot_of_z)
    for (int g = t; g * 256 < n_atoms; g += 1024) a.present_part[g] = s_mask[g];
  if (a.cell == nullptr) {  // molecules: no cell, no bins - the coordinates as they are (launch_wrap's copy)
    for (int k = t; k < 3 * n_atoms; k += 1024) a.xw[k] = a.coord[k];
    return;
  }
  bins_setup_block(a.sys, a.mol_start, n_mol, a.w);
  __syncthreads();
  PREP_STAMP(3);
  {
    const int* src = (const int*)a.sys;
    int* dst = (int*)s_sys;
    for (int k = t; k < n_mol * (int)(sizeof(NlistSystem) / sizeof(int)); k += 1024) dst[k] = src[k];
  }
  __syncthreads();
  PREP_STAMP(4);
  // ---- wrap + count (wrap_bin_count_kernel)
  int bin[PS_K], slot[PS_K];
  auto wrap4 = [&](auto K0) {
    constexpr int k0 = decltype(K0)::value;
    float x[PS_K], y[PS_K], z[PS_K];
#pragma unroll
    for (int k = 0; k < PS_K; ++k) {
      const int i = t + 1024 * (k0 + k), ii = i < n_atoms ? i : 0;
      x[k] = a.coord[3 * ii], y[k] = a.coord[3 * ii + 1], z[k] = a.coord[3 * ii + 2];
    }
#pragma unroll
    for (int k = 0; k < PS_K; ++k) {
      const int i = t + 1024 * (k0 + k);
      bin[k0 + k] = 0, slot[k0 + k] = 0;
      if (i >= n_atoms) continue;
      const NlistSystem& S = s_sys[cur[k0 + k]];
      float w[3];
      wrap_into_cell(S, x[k], y[k], z[k], w);
      a.xw[3 * i] = w[0];
      a.xw[3 * i + 1] = w[1];
      a.xw[3 * i + 2] = w[2];
      int b[3];
      bin[k0 + k] = bin_of(S, w[0], w[1], w[2], b);
      slot[k0 + k] = atomicAdd(&s_bin[bin[k0 + k]], 1);
    }
  };
  wrap4(std::integral_constant<int, 0>{});
  static_assert(PS_K == 4, "one batch of four atoms per thread");
  __syncthreads();
  PREP_STAMP(5);
  // ---- exclusive scan over the bins in use, in place (scan_kernel)
  const int n = s_sys[n_mol - 1].bin_offset + s_sys[n_mol - 1].n_bins;
  {
    const int chunk = (n + 1023) / 1024;
    const int lo = min(n, t * chunk), hi = min(n, lo + chunk);
    int sum = 0;
    for (int k = lo; k < hi; ++k) sum += s_bin[k];
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const int v = (t >= off) ? part[t - off] : 0;
      __syncthreads();
      part[t] += v;
      __syncthreads();
    }
    int run = (t == 0) ? 0 : part[t - 1];
    for (int k = lo; k < hi; ++k) {
      const int c = s_bin[k];
      s_bin[k] = run;
      run += c;
    }
    if (t == 1023) s_bin[n] = part[1023];
  }
  __syncthreads();
  PREP_STAMP(6);
  for (int k = t; k <= n; k += 1024) a.bin_start[k] = s_bin[k];
#pragma unroll
  for (int k = 0; k < PS_K; ++k) {
    const int i = t + 1024 * k;
    if (i < n_atoms) s_tmp[s_bin[bin[k]] + slot[k]] = i;
  }
  __syncthreads();
  PREP_STAMP(7);
  // ---- order every bin by atom id, emit the bin-ordered coordinate stream (bin_sort_kernel)
  auto sort4 = [&](auto K0) {
    constexpr int k0 = decltype(K0)::value;
    float x[PS_K], y[PS_K], z[PS_K];
    int dest[PS_K];
#pragma unroll
    for (int k = 0; k < PS_K; ++k) {
      const int i = t + 1024 * (k0 + k), ii = i < n_atoms ? i : 0;
      x[k] = a.xw[3 * ii], y[k] = a.xw[3 * ii + 1], z[k] = a.xw[3 * ii + 2];  // (this thread's own stores)
    }
#pragma unroll
    for (int k = 0; k < PS_K; ++k) {
      const int i = t + 1024 * (k0 + k);
      const int s0 = s_bin[bin[k0 + k]], cnt = i < n_atoms ? s_bin[bin[k0 + k] + 1] - s0 : 0;
      int rank = 0;
      for (int f = 0; f < cnt; f += 8) {  // 8 independent LDS reads in flight
        int v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = s_tmp[min(s0 + f + j, n_atoms - 1)];
#pragma unroll
        for (int j = 0; j < 8; ++j) rank += (f + j < cnt && v[j] < i) ? 1 : 0;
      }
      dest[k] = s0 + rank;
    }
#pragma unroll
    for (int k = 0; k < PS_K; ++k) {
      const int i = t + 1024 * (k0 + k);
      if (i >= n_atoms) continue;
      a.sorted[dest[k]] = i;
      a.xs[dest[k]] = make_float4(x[k], y[k], z[k], __int_as_float(i));
    }
  };
  sort4(std::integral_constant<int, 0>{});
  PREP_STAMP(8);
}

// periodic: the systems' cell / bin descriptors live in LDS (PREP_SMALL_MAX_MOL of them); molecules need none
bool prep_small_applies(int n_atoms, int n_mol, bool periodic) {
  return n_atoms <= PREP_SMALL_MAX_ATOMS && (!periodic || n_mol <= PREP_SMALL_MAX_MOL);
}

// replaces memset(status) + launch_mol_start + launch_wrap(bin_width) + the bin kernels of the following launch_nlist(bin_width);
// cell == NULL (molecules): memset(status) + launch_mol_start + launch_wrap's copy
int launch_prep_small(hipStream_t s, const float* coord, const int* mol_idx, const int* numbers, int n_atoms, int n_mol,
                      const float* cell, int n_cell, const int pbc[3], const int* pbc_sys, float bin_width, int* status,
                      const int* slot_of_z, int* aslot, unsigned long long* present_part, NlistBuffers& b) {
  PrepSmallArgs a;
  a.coord = coord; a.mol_idx = mol_idx; a.numbers = numbers; a.n_atoms = n_atoms; a.n_mol = n_mol;
  a.cell = cell; a.n_cell = n_cell; a.p0 = pbc[0]; a.p1 = pbc[1]; a.p2 = pbc[2]; a.pbc_sys = pbc_sys; a.w = bin_width;
  a.status = status;
  a.slot_of_z = (numbers && aslot) ? slot_of_z : nullptr; a.aslot = aslot; a.present_part = present_part;
  a.mol_start = b.mol_start; a.mol_c = b.mol_c; a.sys = (NlistSystem*)b.sys; a.xw = b.xw;
  a.bin_start = b.bin_start; a.sorted = b.sorted; a.xs = b.xs;
  const size_t lds = cell ? ((size_t)(n_atoms + 8 * n_mol + 9) + (size_t)n_atoms) * sizeof(int) : 0;  // <= 35 KB
  hipLaunchKernelGGL(prep_small_kernel, dim3(1), dim3(1024), lds, s, a);
  AIMNET_LAUNCH_CHECK();
  b.binned = cell != nullptr;
  b.prebinned_width = cell ? bin_width : 0.0f;
  b.bins_done = cell != nullptr;
  return 0;
}

__global__ __launch_bounds__(256) void nlist_cell_kernel(const float* __restrict__ xw, const int* __restrict__ mol_idx,
                                                        const NlistSystem* __restrict__ sys,
                                                        const int* __restrict__ bin_start, const float4* __restrict__ xs,
                                                        int n_atoms, float cutoff, int cap, int fill_value,
                                                        int fill_rows, int* __restrict__ nb_idx,
                                                        int* __restrict__ nb_shift, int* __restrict__ nb_cnt,
                                                        int* __restrict__ cnt_true, float4* __restrict__ pg, D3CnRider cn) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_atoms) return;
  const float xi = xw[3 * i], yi = xw[3 * i + 1], zi = xw[3 * i + 2];
  int* row = nb_idx + (size_t)i * cap;
  int* rsh = nb_shift + (size_t)i * cap;
  int count = 0;
  // D3CnRider: cn_i = sum_j 1 / (1 + exp(-16 ((rcov_i + rcov_j) / d_ij [Bohr] - 1))) over the hits (d3.hip, d3_cn_kernel)
  const int si = cn.d3w ? cn.aslot[i] : 0;
  const float rci = cn.d3w ? cn.rcov[si] : 0.0f;
  float cnv = 0.0f;
  __shared__ int s_runs[4][CELLWALK_RUN_INTS];
  cell_walk<true>(sys[mol_idx[i]], i, xi, yi, zi, cutoff, bin_start, xs, lane, s_runs[threadIdx.x >> 6],
            [&](float w, float rx, float ry, float rz, bool ok, int code) {
              const unsigned long long mask = __ballot(ok);
              const int pos = count + __popcll(mask & ((1ull << lane) - 1ull));
              if (ok && pos < cap) {
                row[pos] = __float_as_int(w);
                rsh[pos] = code;
                if (pg) pg[(size_t)i * cap + pos] = unit_and_norm(rx, ry, rz);
              }
              if (cn.d3w && ok) {  // (wave-uniform pointer test)
                const float d2 = fmaxf(rx * rx + ry * ry + rz * rz, 1e-24f);
                const float d = d2 * __builtin_amdgcn_rsqf(d2);
                const float rcj = cn.rcov[cn.aslot[__float_as_int(w)]];
                const float arg = -16.0f * ((rci + rcj) * __builtin_amdgcn_rcpf(fmaxf(d * 1.8897261258369282f, 1e-12f)) - 1.0f);
                cnv += __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(arg * 1.4426950408889634f));
              }
              count += __popcll(mask);
            });
  if (cn.d3w) {  // the five shifted exponents / weights of atom i (as d3_cn_kernel leaves them)
    cnv = wave_sum(cnv);
    const int nref = cn.nref[si];
    float e[5], mx = -1e30f;
#pragma unroll
    for (int a = 0; a < 5; ++a) {
      const float dc = cnv - cn.cnref[si * 5 + a];
      e[a] = a < nref ? -4.0f * dc * dc : -1e30f;
      mx = fmaxf(mx, e[a]);
    }
#pragma unroll
    for (int a = 0; a < 5; ++a) {
      if (lane == a) {
        const float sa = a < nref ? e[a] - mx : -1e30f;
        cn.d3w[(size_t)i * 12 + a] = sa;
        cn.d3w[(size_t)i * 12 + 5 + a] = a < nref ? expf(sa) : 0.0f;
      }
    }
    if (lane == 5) cn.d3w[(size_t)i * 12 + 10] = cnv;
    if (lane == 6) cn.d3w[(size_t)i * 12 + 11] = 0.0f;
  }
  if (fill_rows)
    for (int p = min(count, cap) + lane; p < cap; p += 64) {
      row[p] = fill_value;
      rsh[p] = 0;
    }
  if (lane == 0) {
    nb_cnt[i] = min(count, cap);
    cnt_true[i] = count;  // max / overflow are reduced by nlist_status_kernel: 10^4 same-address atomics cost ~12 ns each
  }
}

// bin all atoms of every periodic system into slabs >= `width` thick (count, scan, fill, per-bin sort)
int launch_bins(hipStream_t s, int n_atoms, int n_mol, const int* mol_idx, float width, NlistBuffers& b) {
  NlistSystem* sys = (NlistSystem*)b.sys;
  const int max_bins = n_atoms + 8 * n_mol + 8;
  if (b.bins_done && b.prebinned_width == width && width > 0.0f) {  // launch_prep_small did everything for this width
    b.bins_done = false;
    b.prebinned_width = 0.0f;
    return 0;
  }
  b.bins_done = false;
  if (b.prebinned_width == width && width > 0.0f) {
    b.prebinned_width = 0.0f;  // launch_wrap already set the grids up and counted the atoms per bin for this width
  } else {
    hipLaunchKernelGGL(bins_setup_kernel, dim3(1), dim3(256), 0, s, sys, b.mol_start, n_mol, width);
    AIMNET_LAUNCH_CHECK();
    AIMNET_HIP_CHECK(hipMemsetAsync(b.bin_count, 0, (size_t)(max_bins + 1) * sizeof(int) * 1, s));
    // b.bin_fill holds each atom's arrival rank in its bin (max_bins >= n_atoms entries): one atomic pass instead of two
    hipLaunchKernelGGL(bin_count_kernel, dim3(ceil_div(n_atoms, 256)), dim3(256), 0, s, b.xw, mol_idx, n_atoms, sys,
                       b.atom_bin, b.bin_count, b.bin_fill);
    AIMNET_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, s, b.bin_count, b.bin_start, sys, n_mol);
  AIMNET_LAUNCH_CHECK();
  hipLaunchKernelGGL(bin_fill_kernel, dim3(ceil_div(n_atoms, 256)), dim3(256), 0, s, b.atom_bin, b.bin_start, n_atoms,
                     b.bin_fill, b.sorted_tmp);
  AIMNET_LAUNCH_CHECK();
  hipLaunchKernelGGL(bin_sort_kernel, dim3(ceil_div(max_bins, 4)), dim3(256), 0, s, b.bin_start, sys, n_mol, b.sorted_tmp, b.xw,
                     b.sorted, b.xs);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

// ---- caller-supplied neighbour matrix -> the engine's row format ---------------------------------------------------------
// The reference skips its list builder when the input already carries `nbmat` (calculator.py:1069-1071) and hands the matrix to
// the model as it is: rows [n_atoms][width] int32, entries outside [0, n_atoms) are padding (the reference pads with the index
// of its padding atom, n_atoms), integer lattice shifts [n_atoms][width][3] for periodic input.  One wave per atom compacts the
// valid entries of its row to the front IN THEIR ORDER, packs the shifts, and (pg != NULL) emits the pair geometry (u, d) of
// every entry from the coordinates AS GIVEN (r = x_j + s C - x_i: the shifts of a foreign list refer to the caller's
// coordinates, so nothing is wrapped in this mode).  bad: bit 3 is raised for a self pair without a shift or a shift outside
// the packed range (|s| <= 127).
__global__ __launch_bounds__(256) void import_list_kernel(const int* __restrict__ ext_idx, const int* __restrict__ ext_shift,
                                                         int width, int n_atoms, const float* __restrict__ xw,
                                                         const int* __restrict__ mol_idx, const float* __restrict__ cell,
                                                         int n_cell, int cap, int* __restrict__ nb_idx,
                                                         int* __restrict__ nb_shift, int* __restrict__ nb_cnt,
                                                         int* __restrict__ cnt_true, float4* __restrict__ pg,
                                                         int* __restrict__ bad) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_atoms) return;
  const int lane = threadIdx.x & 63;
  const float* c = cell ? cell + (n_cell == 1 ? 0 : (size_t)mol_idx[i] * 9) : nullptr;
  const float xi = xw[3 * i], yi = xw[3 * i + 1], zi = xw[3 * i + 2];
  int count = 0;  // wave-uniform
  bool flag = false;
  for (int m0 = 0; m0 < width; m0 += 64) {
    const int m = m0 + lane;
    int j = -1, sx = 0, sy = 0, sz = 0;
    if (m < width) {
      j = ext_idx[(size_t)i * width + m];
      if (ext_shift && j >= 0 && j < n_atoms) {
        const int* sp = ext_shift + ((size_t)i * width + m) * 3;
        sx = sp[0]; sy = sp[1]; sz = sp[2];
      }
    }
    const bool valid = j >= 0 && j < n_atoms;
    if (valid && (max(abs(sx), max(abs(sy), abs(sz))) > 127 || (j == i && sx == 0 && sy == 0 && sz == 0))) flag = true;
    const unsigned long long mask = __ballot(valid);
    const int pos = count + __popcll(mask & ((1ull << lane) - 1ull));
    if (valid && pos < cap) {
      const size_t p = (size_t)i * cap + pos;
      nb_idx[p] = j;
      if (nb_shift) nb_shift[p] = pack_shift(sx, sy, sz);
      if (pg) {
        float rx = xw[3 * j] - xi, ry = xw[3 * j + 1] - yi, rz = xw[3 * j + 2] - zi;
        if (c) {
          rx += sx * c[0] + sy * c[3] + sz * c[6];
          ry += sx * c[1] + sy * c[4] + sz * c[7];
          rz += sx * c[2] + sy * c[5] + sz * c[8];
        }
        const float d = sqrtf(rx * rx + ry * ry + rz * rz);
        const float inv = 1.0f / d;
        pg[p] = make_float4(rx * inv, ry * inv, rz * inv, d);
      }
    }
    count += __popcll(mask);
  }
  if (__ballot(flag) && lane == 0) atomicOr(bad, 8);
  if (lane == 0) {
    nb_cnt[i] = min(count, cap);
    cnt_true[i] = count;
  }
}

__global__ __launch_bounds__(1024) void nlist_status_kernel(const int* __restrict__ cnt_true, int n_atoms, int cap,
                                                           int* __restrict__ status_max, int* __restrict__ status_ovf) {
  nlist_status_block(cnt_true, n_atoms, cap, status_max, status_ovf, blockIdx.x);
}

int launch_nlist(hipStream_t s, int n_atoms, int n_mol, const int* mol_idx, const float* cell, int n_cell,
                 const int pbc[3], float cutoff, float bin_width, int cap, int fill_value, int fill_rows,
                 NlistBuffers& b, int* nb_idx, int* nb_shift, int* nb_cnt, int* status_max, int* status_ovf, float4* pg,
                 const int** status_later, const D3CnRider* cn, bool* cn_done) {
  if (cn_done) *cn_done = false;
  (void)n_cell;
  (void)pbc;
  (void)cell;
  int* cnt_true = b.sorted_tmp;  // free once the bins are sorted (and never used by the per-molecule scan)
  if (!b.binned) {
    hipLaunchKernelGGL(nlist_brute_kernel, dim3(ceil_div(n_atoms, 4)), dim3(256), 0, s, b.xw, mol_idx, b.mol_start, n_atoms,
                       cutoff * cutoff, cap, fill_value, fill_rows, nb_idx, nb_cnt, cnt_true, pg);
    AIMNET_LAUNCH_CHECK();
  } else {
    if (bin_width > 0.0f) {  // bin_width <= 0: reuse the bins of the previous call
      int rc = launch_bins(s, n_atoms, n_mol, mol_idx, bin_width, b);
      if (rc) return rc;
    }
    hipLaunchKernelGGL(nlist_cell_kernel, dim3(ceil_div(n_atoms, 4)), dim3(256), 0, s, b.xw, mol_idx, (NlistSystem*)b.sys,
                       b.bin_start, b.xs, n_atoms, cutoff, cap, fill_value, fill_rows, nb_idx, nb_shift, nb_cnt, cnt_true, pg,
                       cn ? *cn : D3CnRider{});
    AIMNET_LAUNCH_CHECK();
    if (cn && cn->d3w && cn_done) *cn_done = true;
  }
  if (status_later) {  // the caller runs nlist_status_block over these counts as riders of a later launch (no list build in between)
    *status_later = cnt_true;
    return 0;
  }
  hipLaunchKernelGGL(nlist_status_kernel, dim3(ceil_div(n_atoms, 1024)), dim3(1024), 0, s, cnt_true, n_atoms, cap, status_max,
                     status_ovf);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

int launch_import_list(hipStream_t s, const int* ext_idx, const int* ext_shift, int width, int n_atoms, const int* mol_idx,
                       const float* cell, int n_cell, int cap, NlistBuffers& b, int* nb_idx, int* nb_shift, int* nb_cnt,
                       int* status_max, int* status_ovf, float4* pg, int* bad) {
  int* cnt_true = b.sorted_tmp;
  hipLaunchKernelGGL(import_list_kernel, dim3(ceil_div(n_atoms, 4)), dim3(256), 0, s, ext_idx, ext_shift, width, n_atoms, b.xw,
                     mol_idx, cell, n_cell, cap, nb_idx, nb_shift, nb_cnt, cnt_true, pg, bad);
  AIMNET_LAUNCH_CHECK();
  hipLaunchKernelGGL(nlist_status_kernel, dim3(ceil_div(n_atoms, 1024)), dim3(1024), 0, s, cnt_true, n_atoms, cap, status_max,
                     status_ovf);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

// every entry (i -> j, s) of a full neighbour matrix needs its mirror (j -> i, -s): the atomics-free conv backward evaluates the
// adjoint of the pair seen from j at the centre i.  Brute force over the row of j (caller-supplied matrices only: the engine's own
// builder emits both directions by construction).  The same test runs on the caller's long-range / D3 matrices: a half list would
// silently halve the Coulomb / dispersion energy.  bad: bit 4.
__global__ __launch_bounds__(256) void list_symmetry_kernel(const int* __restrict__ nb_idx, const int* __restrict__ nb_shift,
                                                           const int* __restrict__ nb_cnt, int cap, int n_atoms,
                                                           int* __restrict__ bad, int max_check) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_atoms) return;
  const int lane = threadIdx.x & 63;
  // max_check: pairs per row that are verified.  The short-range matrix is verified in full (the reverse-pair map needs EXACTLY one
  // mirror per pair); of a long-range / D3 row (2 000 entries at 15 A) the first max_check pairs are - the test there is against a
  // HALF list, which fails on the first pairs of almost every row, and the full scan is quadratic in the row length
  const int cnt = min(nb_cnt[i], max_check);
  // The wave takes the pairs of row i one after the other and scans row j with all 64 lanes (coalesced; a lane-per-pair scan of
  // 64 different rows took 0.39 s on a 10 080 x 2 064 long-range matrix, this form a few ms)
  bool missing = false;
  for (int m0 = 0; m0 < cnt; m0 += 64) {
    int jl = 0, wl = 0;
    if (m0 + lane < cnt) {
      const size_t p = (size_t)i * cap + m0 + lane;
      jl = nb_idx[p];
      if (nb_shift) {
        int sx, sy, sz;
        unpack_shift(nb_shift[p], sx, sy, sz);
        wl = pack_shift(-sx, -sy, -sz);
      }
    }
    const int nm = min(64, cnt - m0);
    for (int k = 0; k < nm; ++k) {
      const int j = __shfl(jl, k), want = __shfl(wl, k);
      const int cj = nb_cnt[j];
      // EXACTLY one mirror: a duplicated (j, shift) entry would give two entries of the reverse-pair map one slot, and the force
      // gather would subtract one pair-buffer entry twice (silently wrong forces)
      int found = 0;
      for (int t0 = 0; t0 < cj; t0 += 64) {
        const int t = t0 + lane;
        bool hit = false;
        if (t < cj) {
          const size_t r = (size_t)j * cap + t;
          hit = nb_idx[r] == i && (!nb_shift || (nb_shift[r] & 0xffffff) == (want & 0xffffff));
        }
        found += __popcll(__ballot(hit));
      }
      missing = missing || found != 1;
    }
  }
  if (__ballot(missing) && lane == 0) atomicOr(bad, 16);
}

int launch_list_symmetry_check(hipStream_t s, const int* nb_idx, const int* nb_shift, const int* nb_cnt, int cap, int n_atoms,
                               int* bad, int max_check) {
  hipLaunchKernelGGL(list_symmetry_kernel, dim3(ceil_div(n_atoms, 4)), dim3(256), 0, s, nb_idx, nb_shift, nb_cnt, cap, n_atoms, bad,
                     max_check);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

}  // namespace aimnet
