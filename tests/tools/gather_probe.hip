// What bounds the neighbour-row gathers of conv_bwd (5 KiB per ordered pair: a_j 1 KiB + Sbar_j 4 KiB)?
// Stand-alone probe: a jittered lattice at the density of the config-3 crystal (10 080 atoms, ~68 neighbours within 5 A,
// atoms numbered in cell-list bin order like the engine processes them), one wave per centre atom, every neighbour row
// loaded with the engine's access pattern (five coalesced 1 KiB wave loads) and folded into a checksum - no other work.
// Variants: resident waves per SIMD (launch bounds) x rows kept in flight per wave (register ring).
//   hipcc --offload-arch=gfx950 -O3 tests/tools/gather_probe.hip -o gpurun_in/gather_probe && gpurun_in/gather_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <random>
#include <array>
#include <vector>

// SPLIT: the four waves of a block share one centre atom (a quarter of its row each) instead of taking one atom each
template <int WPS, int RING, int ROWF4, int SPLIT, int TABLES = 0, int GRIDX = 1>  // TABLES 1: a / Sbar / Sqbar as three arrays like the engine (ROWF4 = 5 + a dword); GRIDX: grid = GRIDX x resident blocks; SPLIT 2: the four waves of a block all gather the SAME atom's full list (L1 sharing test); ROWF4: float4 loads per lane per row (5 = bwd: 5 KiB rows, 1 = fwd: 1 KiB rows)
__global__ __launch_bounds__(256, WPS) void probe(const float4* __restrict__ tab, const int* __restrict__ nb_idx,
                                                 const int* __restrict__ nb_cnt, int cap, int n_atoms, const int* __restrict__ order,
                                                 float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nb = gridDim.x, b = blockIdx.x;
  const int apb = SPLIT ? 1 : 4;
  const int xcd = b & 7, slot = b >> 3, per = (nb >> 3) + (xcd < (nb & 7) ? 1 : 0);
  const int nblk = (n_atoms + apb - 1) / apb, chunk = (nblk + 7) >> 3, lo = xcd * chunk, hi = min(nblk, lo + chunk);
  float acc = 0.f;
  for (int ib = lo + slot; ib < hi; ib += per) {
    const int ii = SPLIT ? ib : ib * 4 + wid;
    if (ii >= n_atoms) continue;
    const int i = order[ii];
    const int call = __builtin_amdgcn_readfirstlane(nb_cnt[i]);
    const int q4 = (call + 3) >> 2;
    const int mlo = SPLIT == 1 ? wid * q4 : 0;
    const int cnt = SPLIT == 1 ? max(0, min(q4, call - mlo)) : call;
    if (cnt == 0) continue;
    const int* row = nb_idx + (size_t)i * cap + mlo;
    // the wave's neighbour indices in registers (the engine stages them in LDS): a row load must not wait for an index
    // load, vmcnt is in-order and that would drain the ring
    const int jv0 = row[min(lane, cnt - 1)], jv1 = row[min(lane + 64, cnt - 1)];
    float4 r[RING][ROWF4];
    float rq[RING];
    auto load = [&](int m, int s) {
      const int mm = min(m, cnt - 1);
      const int j = mm < 64 ? __builtin_amdgcn_readlane(jv0, mm) : __builtin_amdgcn_readlane(jv1, mm - 64);
      if (TABLES == 0) {
        const float4* p = tab + (size_t)j * (64 * ROWF4) + lane;
#pragma unroll
        for (int k = 0; k < ROWF4; ++k) r[s][k] = p[64 * k];
        rq[s] = 0.f;
      } else {  // Sbar rows first in the buffer (n_atoms x 4 KiB), then a (n_atoms x 1 KiB), then Sqbar (n_atoms x 256 B)
        const float4* ps = tab + (size_t)j * 256 + lane;
        const float4* pa = tab + (size_t)n_atoms * 256 + (size_t)j * 64 + lane;
        const float* pq = reinterpret_cast<const float*>(tab + (size_t)n_atoms * 320) + (size_t)j * 64 + lane;
        r[s][0] = pa[0];
#pragma unroll
        for (int k = 1; k < ROWF4; ++k) r[s][k] = ps[64 * (k - 1)];
        rq[s] = pq[0];
      }
    };
#pragma unroll
    for (int s = 0; s < RING - 1; ++s) load(s, s);
    for (int m0 = 0; m0 < cnt; m0 += RING) {
#pragma unroll
      for (int s = 0; s < RING; ++s) {
        load(m0 + s + RING - 1, (s + RING - 1) % RING);
        if (m0 + s < cnt) {
#pragma unroll
          for (int k = 0; k < ROWF4; ++k) acc += (r[s][k].x + r[s][k].y) + (r[s][k].z + r[s][k].w);
          acc += rq[s];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int WPS, int RING, int ROWF4>
__global__ __launch_bounds__(256, WPS) void probe_cluster(const float4* __restrict__ tab, const int* __restrict__ u_idx,
                                                         const int* __restrict__ u_cnt, int ucap, int n_clusters, int n_atoms,
                                                         float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nb = gridDim.x, b = blockIdx.x;
  const int xcd = b & 7, slot = b >> 3, per = (nb >> 3) + (xcd < (nb & 7) ? 1 : 0);
  const int nblk = (n_clusters + 3) / 4, chunk = (nblk + 7) >> 3, lo = xcd * chunk, hi = min(nblk, lo + chunk);
  float acc = 0.f;
  for (int ib = lo + slot; ib < hi; ib += per) {
    const int cl = ib * 4 + wid;
    if (cl >= n_clusters) continue;
    const int cnt = __builtin_amdgcn_readfirstlane(u_cnt[cl]);
    const int* row = u_idx + (size_t)cl * ucap;
    __shared__ int s_idx[4][512];
    for (int k = lane; k < cnt; k += 64) s_idx[wid][k] = row[k];  // indices staged in LDS: a row load must never wait behind an index load (vmcnt is in-order)
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    float4 r[RING][ROWF4];
    auto load = [&](int m, int s) {
      const int j = __builtin_amdgcn_readfirstlane(s_idx[wid][min(m, cnt - 1)]);
      const float4* ps = tab + (size_t)j * 256 + lane;
#pragma unroll
      for (int k = 0; k < ROWF4; ++k) r[s][k] = ps[64 * k];
    };
#pragma unroll
    for (int s = 0; s < RING - 1; ++s) load(s, s);
    for (int m0 = 0; m0 < cnt; m0 += RING) {
#pragma unroll
      for (int s = 0; s < RING; ++s) {
        load(m0 + s + RING - 1, (s + RING - 1) % RING);
        if (m0 + s < cnt) {
#pragma unroll
          for (int k = 0; k < ROWF4; ++k) acc += (r[s][k].x + r[s][k].y) + (r[s][k].z + r[s][k].w);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int WPS, int RING, int ROWF4>
void run_cluster(const float4* tab, const int* u_idx, const int* u_cnt, int ucap, int ncl, int n, float* out, long entries, long pairs) {
  const int grid = std::min((ncl + 3) / 4, 256 * WPS);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) probe_cluster<WPS, RING, ROWF4><<<grid, 256>>>(tab, u_idx, u_cnt, ucap, ncl, n, out);
  hipEventRecord(e0);
  for (int w = 0; w < 10; ++w) probe_cluster<WPS, RING, ROWF4><<<grid, 256>>>(tab, u_idx, u_cnt, ucap, ncl, n, out);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("CLUSTER of 4: row %d KiB  waves/SIMD %d  ring %d: %ld union entries for %ld pairs (sharing %.2f): %7.1f us  %6.2f TB/s\n", ROWF4, WPS, RING,
         entries, pairs, (double)pairs / entries, ms * 100.0, (double)entries * ROWF4 * 1024 / (ms * 100.0) * 1e-6);
}

template <int WPS, int RING, int ROWF4, int SPLIT, int TABLES = 0, int GRIDX = 1>
void run(const float4* tab, const int* idx, const int* cnt, int cap, int n, const int* order, const char* oname, float* out, long pairs) {
  const int resident = 256 * WPS;  // blocks of 4 waves: WPS waves per SIMD = WPS blocks per CU
  const int grid = std::min(SPLIT ? n : (n + 3) / 4, resident * GRIDX);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) probe<WPS, RING, ROWF4, SPLIT, TABLES, GRIDX><<<grid, 256>>>(tab, idx, cnt, cap, n, order, out);
  hipEventRecord(e0);
  for (int w = 0; w < 10; ++w) probe<WPS, RING, ROWF4, SPLIT, TABLES, GRIDX><<<grid, 256>>>(tab, idx, cnt, cap, n, order, out);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 100.0, bytes = (double)pairs * ROWF4 * 1024 * (SPLIT == 2 ? 4 : 1);
  printf("row %d KiB %s grid x%d  waves/SIMD %d  ring %d  %s  order %-10s centres in flight per XCD %4d: %7.1f us  %6.2f TB/s\n", ROWF4, TABLES ? "3 tables" : "1 table ", GRIDX, WPS, RING,
         SPLIT == 2 ? "dup4   " : SPLIT ? "split4 " : "1w/atom", oname, 32 * WPS * (SPLIT ? 1 : 4), us, bytes / us * 1e-6);
}

int main() {
  // jittered lattice in the config-3 box, density 0.1298 / A^3
  const double L[3] = {34.87, 37.69, 59.08}, rc = 5.0;
  const int g[3] = {17, 18, 29};  // 8874 sites x ... ~ density; use 2 atoms per site pattern to reach ~10k
  std::mt19937 rng(1);
  std::uniform_real_distribution<double> U(-0.35, 0.35);
  std::vector<std::array<double, 3>> x;
  for (int a = 0; a < g[0]; ++a)
    for (int b = 0; b < g[1]; ++b)
      for (int c = 0; c < g[2]; ++c) {
        x.push_back({(a + 0.5 + U(rng)) * L[0] / g[0], (b + 0.5 + U(rng)) * L[1] / g[1], (c + 0.5 + U(rng)) * L[2] / g[2]});
        if ((a + b + c) % 7 == 0) x.push_back({(a + 0.9 + U(rng)) * L[0] / g[0], (b + 0.1 + U(rng)) * L[1] / g[1], (c + 0.5 + U(rng)) * L[2] / g[2]});
      }
  int n = (int)x.size();
  // bin order (x fastest), like the cell list
  int nbx[3];
  for (int k = 0; k < 3; ++k) nbx[k] = std::max(1, (int)std::floor(L[k] / rc));
  std::vector<int> key(n), ord(n);
  for (int i = 0; i < n; ++i) {
    int bi[3];
    for (int k = 0; k < 3; ++k) bi[k] = std::min(nbx[k] - 1, std::max(0, (int)(x[i][k] / L[k] * nbx[k])));
    key[i] = (bi[2] * nbx[1] + bi[1]) * nbx[0] + bi[0];
    ord[i] = i;
  }
  std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return key[a] < key[b]; });
  std::vector<std::array<double, 3>> xs(n);
  for (int i = 0; i < n; ++i) xs[i] = x[ord[i]];
  const int cap = 112;
  std::vector<int> idx((size_t)n * cap, 0), cnt(n, 0);
  long pairs = 0;
  for (int i = 0; i < n; ++i) {
    int c = 0;
    for (int j = 0; j < n && c < cap; ++j) {
      if (j == i) continue;
      double d2 = 0;
      for (int k = 0; k < 3; ++k) {
        double d = xs[j][k] - xs[i][k];
        d -= L[k] * std::round(d / L[k]);
        d2 += d * d;
      }
      if (d2 < rc * rc) idx[(size_t)i * cap + c++] = j;
    }
    cnt[i] = c;
    pairs += c;
  }
  printf("%d atoms, %.1f neighbours on average, %ld ordered pairs\n", n, (double)pairs / n, pairs);
  float4* tab; int *didx, *dcnt; float* out;
  hipMalloc(&tab, (size_t)n * 6 * 1024); hipMemset(tab, 0, (size_t)n * 6 * 1024);
  hipMalloc(&didx, idx.size() * 4); hipMalloc(&dcnt, n * 4); hipMalloc(&out, (size_t)16384 * 256 * 4);
  hipMemcpy(didx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dcnt, cnt.data(), n * 4, hipMemcpyHostToDevice);
  // processing orders.  "bin": identity (atoms are numbered in bin order, x fastest; XCD x takes the x-th eighth of it).
  // "boxcol": the box is cut into 2x2x2 XCD domains, each domain into columns of ~9 A x 9 A along its longest axis, atoms
  // sorted by (domain, column, position along the column): an XCD sweeps one thin column at a time.
  std::vector<int> o_bin(n), o_col(n);
  for (int i = 0; i < n; ++i) o_bin[i] = i;
  {
    std::vector<std::array<double, 5>> k(n);
    for (int i = 0; i < n; ++i) {
      const int dx = xs[i][0] < L[0] / 2 ? 0 : 1, dy = xs[i][1] < L[1] / 2 ? 0 : 1, dz = xs[i][2] < L[2] / 2 ? 0 : 1;
      const double lx = xs[i][0] - dx * L[0] / 2, ly = xs[i][1] - dy * L[1] / 2;
      const int cx = std::min(1, (int)(lx / (L[0] / 4))), cy = std::min(1, (int)(ly / (L[1] / 4)));
      k[i] = {(double)(dz * 4 + dy * 2 + dx), (double)(cy * 2 + cx), xs[i][2], 0, (double)i};
    }
    std::vector<int> o(n);
    for (int i = 0; i < n; ++i) o[i] = i;
    std::sort(o.begin(), o.end(), [&](int a, int b) { return k[a] < k[b]; });
    o_col = o;
  }
  int *d_bin, *d_col;
  hipMalloc(&d_bin, n * 4); hipMalloc(&d_col, n * 4);
  hipMemcpy(d_bin, o_bin.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_col, o_col.data(), n * 4, hipMemcpyHostToDevice);
#define R(W, G, F, SP, O) run<W, G, F, SP>(tab, didx, dcnt, cap, n, O == 0 ? d_bin : d_col, O == 0 ? "bin" : "boxcol", out, pairs)
  R(4, 1, 5, 0, 0); R(4, 2, 5, 0, 0); R(2, 2, 5, 0, 0); R(2, 4, 5, 0, 0); R(3, 2, 5, 0, 0); R(3, 3, 5, 0, 0);
#define R2(W, G, TB, GX) run<W, G, 5, 0, TB, GX>(tab, didx, dcnt, cap, n, d_bin, "bin", out, pairs)
  R2(4, 1, 1, 1); R2(4, 1, 0, 2); R2(4, 1, 1, 2); R2(4, 2, 1, 1); R2(2, 2, 1, 1); R2(2, 4, 1, 1); R2(3, 2, 1, 1);
  {  // the ENGINE's own lists and processing order (tests/tools/dump_lists.py), if present
    FILE* fm = fopen("gpurun_in/lists/meta.txt", "r");
    if (fm) {
      int rn = 0, rcap = 0;
      if (fscanf(fm, "%d %d", &rn, &rcap) == 2) {
        std::vector<int> ridx((size_t)rn * rcap), rcnt(rn), rord(rn);
        FILE* f1 = fopen("gpurun_in/lists/nb_idx.bin", "rb"); FILE* f2 = fopen("gpurun_in/lists/nb_cnt.bin", "rb"); FILE* f3 = fopen("gpurun_in/lists/order.bin", "rb");
        size_t got = fread(ridx.data(), 4, ridx.size(), f1) + fread(rcnt.data(), 4, rn, f2) + fread(rord.data(), 4, rn, f3);
        (void)got;
        long rp = 0; for (int v : rcnt) rp += v;
        int *e_idx, *e_cnt, *e_ord, *e_id;
        hipMalloc(&e_idx, ridx.size() * 4); hipMalloc(&e_cnt, rn * 4); hipMalloc(&e_ord, rn * 4); hipMalloc(&e_id, rn * 4);
        hipMemcpy(e_idx, ridx.data(), ridx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(e_cnt, rcnt.data(), rn * 4, hipMemcpyHostToDevice);
        hipMemcpy(e_ord, rord.data(), rn * 4, hipMemcpyHostToDevice);
        std::vector<int> ident(rn); for (int i = 0; i < rn; ++i) ident[i] = i;
        hipMemcpy(e_id, ident.data(), rn * 4, hipMemcpyHostToDevice);
        printf("engine lists: %d atoms, cap %d, %ld pairs\n", rn, rcap, rp);
#define RE(W, G, TB, GX, ORD, NAME) run<W, G, 5, 0, TB, GX>(tab, e_idx, e_cnt, rcap, rn, ORD, NAME, out, rp)
        RE(4, 1, 1, 2, e_ord, "eng-bin"); RE(2, 2, 1, 1, e_ord, "eng-bin"); RE(2, 3, 1, 1, e_ord, "eng-bin"); RE(2, 4, 1, 1, e_ord, "eng-bin");
        RE(1, 4, 1, 1, e_ord, "eng-bin"); RE(1, 8, 1, 1, e_ord, "eng-bin"); RE(3, 2, 1, 1, e_ord, "eng-bin");
        {  // clusters of 4 consecutive atoms in processing order, union of their rows (by neighbour index)
          const int ncl = (rn + 3) / 4, ucap = 4 * rcap;
          std::vector<int> uidx((size_t)ncl * ucap, 0), ucnt(ncl, 0);
          long entries = 0;
          for (int c = 0; c < ncl; ++c) {
            std::vector<int> u;
            for (int k = 0; k < 4 && 4 * c + k < rn; ++k) {
              const int i = rord[4 * c + k];
              for (int m = 0; m < rcnt[i]; ++m) u.push_back(ridx[(size_t)i * rcap + m]);
            }
            std::sort(u.begin(), u.end());
            u.erase(std::unique(u.begin(), u.end()), u.end());
            ucnt[c] = (int)u.size();
            entries += ucnt[c];
            std::copy(u.begin(), u.end(), uidx.begin() + (size_t)c * ucap);
          }
          int *d_ui, *d_uc;
          hipMalloc(&d_ui, uidx.size() * 4); hipMalloc(&d_uc, ncl * 4);
          hipMemcpy(d_ui, uidx.data(), uidx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d_uc, ucnt.data(), ncl * 4, hipMemcpyHostToDevice);
          run_cluster<2, 2, 4>(tab, d_ui, d_uc, ucap, ncl, rn, out, entries, rp);
          run_cluster<2, 3, 4>(tab, d_ui, d_uc, ucap, ncl, rn, out, entries, rp);
          run_cluster<2, 4, 4>(tab, d_ui, d_uc, ucap, ncl, rn, out, entries, rp);
          run_cluster<3, 3, 4>(tab, d_ui, d_uc, ucap, ncl, rn, out, entries, rp);
          run_cluster<1, 6, 4>(tab, d_ui, d_uc, ucap, ncl, rn, out, entries, rp);
          run_cluster<2, 4, 1>(tab, d_ui, d_uc, ucap, ncl, rn, out, entries, rp);
          run_cluster<2, 8, 1>(tab, d_ui, d_uc, ucap, ncl, rn, out, entries, rp);
          run_cluster<4, 8, 1>(tab, d_ui, d_uc, ucap, ncl, rn, out, entries, rp);
        }
#define RS(W, G, GX) run<W, G, 5, 1, 1, GX>(tab, e_idx, e_cnt, rcap, rn, e_ord, "eng-bin", out, rp)
        RS(4, 1, 1); RS(4, 2, 1); RS(2, 2, 1); RS(2, 4, 1); RS(3, 2, 1); RS(4, 1, 8);
      }
      fclose(fm);
    }
  }
  return 0;
}
