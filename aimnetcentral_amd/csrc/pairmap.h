// pairmap.h - the reverse-pair map of a full, symmetric neighbour matrix, as device functions: the two small kernels ride on
// other launches of the evaluation (role-dispatched blocks, kernels.h SrRiders / PairMapRider) and also exist as kernels of their own
// (conv.hip).  rev[i * cap + m] = position of (i, -shift) in the row of j = idx[i][m] (-1 if the row of j does not hold it).
#pragma once

#include "common.h"
#include "conv_common.h"

namespace aimnet {

// Every atom j publishes a 256-slot open-addressing table of its row,
// entry = neighbour << 32 | shift code << 8 | position, slot = hash(neighbour, shift) with linear probing (rows hold <= 128
// entries: load factor <= 1/2), built with LDS compare-and-swap by the wave of j and written out as one 2 KiB block.  The pair
// (i -> j) then finds (i, -shift) in the table of j with ~1.3 dependent 8-byte loads instead of a 7-step bisection.
constexpr int RH_SLOTS = 256;
constexpr unsigned long long RH_EMPTY = ~0ull;
__device__ __forceinline__ unsigned rh_slot(unsigned id, unsigned code) {
  return ((id * 2654435761u) ^ (code * 0x9E3779B1u) ^ (code >> 11)) >> 24;
}

// s_tab: 4 x RH_SLOTS entries of LDS (one table per wave of the 256-thread block)
__device__ __forceinline__ void pair_hash_block(const int* __restrict__ nb_idx, const int* __restrict__ nb_shift,
                                                const int* __restrict__ nb_cnt, int cap, int n_atoms,
                                                unsigned long long* __restrict__ tab, int* __restrict__ rev, int block,
                                                unsigned long long (*s_tab)[RH_SLOTS]) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int i = block * 4 + wid;
  if (i >= n_atoms) return;  // (no block barrier below)
  unsigned long long* T = s_tab[wid];
  for (int t = lane; t < RH_SLOTS; t += 64) T[t] = RH_EMPTY;
  lds_sync<false>();
  const int cnt = min(nb_cnt[i], cap);
  for (int m = lane; m < cnt; m += 64) {
    const size_t p = (size_t)i * cap + m;
    rev[p] = -1;  // an entry whose reverse pair is missing (a truncated row) is written by nobody in pair_rev_hash_kernel
    const unsigned id = (unsigned)nb_idx[p], code = nb_shift ? (unsigned)(nb_shift[p] & 0xffffff) : 0u;
    const unsigned long long entry = ((unsigned long long)id << 32) | ((unsigned long long)code << 8) | (unsigned long long)m;
    unsigned h = rh_slot(id, code);
    while (atomicCAS(&T[h], RH_EMPTY, entry) != RH_EMPTY) h = (h + 1) & (RH_SLOTS - 1);
  }
  lds_sync<false>();
  for (int t = lane; t < RH_SLOTS; t += 64) tab[(size_t)i * RH_SLOTS + t] = T[t];
}

__device__ __forceinline__ void pair_rev_hash_block(const int* __restrict__ nb_idx, const int* __restrict__ nb_shift,
                                                    const int* __restrict__ nb_cnt, int cap, int n_atoms,
                                                    const unsigned long long* __restrict__ tab, int* __restrict__ rev, int block) {
  const int lane = threadIdx.x & 63;
  const int i = block * 4 + (threadIdx.x >> 6);
  if (i >= n_atoms) return;
  const int cnt = min(nb_cnt[i], cap);
  for (int m = lane; m < cnt; m += 64) {
    const size_t p = (size_t)i * cap + m;
    const int j = nb_idx[p];
    if (j < i) continue;  // rev(rev(p)) = p: the pair with the smaller centre looks up and writes both entries
    unsigned want = 0;
    if (nb_shift) {
      int sx, sy, sz;
      unpack_shift(nb_shift[p], sx, sy, sz);
      want = (unsigned)pack_shift(-sx, -sy, -sz) & 0xffffffu;
    }
    const unsigned long long* Tj = tab + (size_t)j * RH_SLOTS;
    const unsigned long long match = ((unsigned long long)(unsigned)i << 24) | want;
    unsigned h = rh_slot((unsigned)i, want);
    int found = -1;
    for (int probe = 0; probe < RH_SLOTS; ++probe) {
      const unsigned long long e = Tj[h];
      if (e == RH_EMPTY) break;
      if ((e >> 8) == match) {
        found = (int)(e & 0xff);
        break;
      }
      h = (h + 1) & (RH_SLOTS - 1);
    }
    rev[p] = found;
    if (found >= 0 && j != i) rev[(size_t)j * cap + found] = m;
  }
}

// forces_i = -(fgrad_i + sum_m F1(i -> j_m) - F1(j_m -> i))  (conv_bwd_kernel XE form; one wave per atom, lane = pair)
__device__ __forceinline__ void pair_force_block(const int* __restrict__ nb_idx, const int* __restrict__ nb_cnt,
                                                 const int* __restrict__ rev, const float4* __restrict__ pairbuf, int cap,
                                                 int n_atoms, const float* __restrict__ fgrad, float* __restrict__ forces, int block,
                                                 int* __restrict__ nf = nullptr) {
  const int lane = threadIdx.x & 63;
  const int i = block * 4 + (threadIdx.x >> 6);
  if (i >= n_atoms) return;
  const int cnt = min(nb_cnt[i], cap);
  float f0 = 0.f, f1 = 0.f, f2 = 0.f;
  for (int m = lane; m < cnt; m += 64) {
    const size_t p = (size_t)i * cap + m;
    const int r = rev[p];
    const float4 own = pairbuf[p];
    f0 += own.x; f1 += own.y; f2 += own.z;
    if (r >= 0) {
      const float4 oth = pairbuf[(size_t)nb_idx[p] * cap + r];
      f0 -= oth.x; f1 -= oth.y; f2 -= oth.z;
    }
  }
  f0 = wave_sum(f0); f1 = wave_sum(f1); f2 = wave_sum(f2);
  if (lane == 0) {
    forces[3 * i + 0] = -(fgrad[3 * i + 0] + f0);
    forces[3 * i + 1] = -(fgrad[3 * i + 1] + f1);
    forces[3 * i + 2] = -(fgrad[3 * i + 2] + f2);
    if (nf && !(isfinite(f0 + fgrad[3 * i]) && isfinite(f1 + fgrad[3 * i + 1]) && isfinite(f2 + fgrad[3 * i + 2]))) atomicOr(nf, 32);
  }
}

}  // namespace aimnet
