// conv_cluster.hip - the ConvSV backward for CLUSTERS of four centre atoms (large systems).
//
// Reference semantics: Warp backward kernels conv_sv_2d_sp_wp.py:115-164 fused with the AEV backward aev.py:94-110, the same
// centre-major algebra as conv_bwd_kernel (conv.hip, oracle/aimnet2_analytic.py).  What changes is the unit of work.
//
// conv_bwd_kernel is bound by the bytes it gathers: 5.25 KiB (a_j and Sbar_j) per ordered pair, one wave per centre atom.
// Four centres that are neighbours in space share about half of their neighbours, and
//
//   (1) the pair adjoints only need  Y_p[g,c] = sum_a a_i[a,g] Sbar_j[a,g,c]  of every ordered pair p = (i -> j):
//         D_p   = sum_g dgs_g(d) sum_c (1,-u)_c Y_p[g,c],      U_p,k = sum_g gs_g(d) Y_p[g,k+1],
//         F1(p) = (U_p - (U_p.u) u) / d - D_p u.
//       The "X" half of conv_bwd_kernel (sum_a a_j Sbar_i) of the pair (i -> j) IS the Y of the reverse pair (j -> i) with
//       u -> -u, so   dE/dx_i = sum_{p: centre i} F1(p) - sum_{p': neighbour i} F1(p')   and the virial is sum_p -r_p (x) F1(p):
//       every ordered pair is evaluated once, writes F1 into a pair buffer, and a gather kernel subtracts the reverse pair's
//       entry (cluster_rev_kernel builds that map once per neighbour list).  a_j is never gathered: 4 KiB per row, not 5.25.
//   (2) Y for the four centres of a cluster is ONE v_mfma_f32_4x4x1 chain: block = shift g, row = centre r, column = c,
//       K = a (16 MFMAs per neighbour row, A = a_{i_r}[a,g] held in registers, B = the gathered row in the T layout of
//       unconcat_t_kernel).  A row that is a neighbour of k of the four centres is loaded once instead of k times
//       (k ~ 2 for bin-ordered clusters of a molecular crystal), entries that are not a neighbour of a centre carry weight 0.
//
// Lists: cluster c = centres order[4c..4c+3]; its union list holds the distinct (j, shift) of the four rows, sorted, and for
// every (entry, centre) the pair geometry (u, d) of the neighbour list or the sentinel d = -1.
#include <stdint.h>

#include "conv_common.h"

namespace aimnet {

namespace {
constexpr int CLW = 4;            // centres per cluster
constexpr int CL_ROW_MAX = 128;   // largest neighbour-row capacity the builder sorts (4 rows = 512 keys per wave)

__device__ __forceinline__ int neg_shift(int code) {
  int sx, sy, sz;
  unpack_shift(code, sx, sy, sz);
  return pack_shift(-sx, -sy, -sz) & 0xffffff;
}
}  // namespace

bool cluster_lists_supported(int n_atoms, int cap) {
  return cap <= CL_ROW_MAX && (size_t)((n_atoms + CLW - 1) / CLW) * (size_t)(CLW * cap) * 4 < (size_t)INT32_MAX;
}

// ------------------------------------------------------------------------------------------------
// Union lists.  One wave per cluster: the (<= 4 x cap) entries of the four rows are tagged (j, shift, m, r), bitonic-sorted
// in LDS, and the first entry of every (j, shift) group writes the union entry with the geometry of the centres that hold it.
__global__ __launch_bounds__(256) void cluster_build_kernel(const int* __restrict__ nb_idx, const int* __restrict__ nb_shift,
                                                            const int* __restrict__ nb_cnt, const float4* __restrict__ pg, int cap,
                                                            int capU, const int* __restrict__ order, int n_atoms, int n_cl,
                                                            int* __restrict__ cl_cnt, int* __restrict__ cl_idx,
                                                            int* __restrict__ cl_shift, float4* __restrict__ cl_ud,
                                                            int* __restrict__ pos_of) {
  __shared__ unsigned long long s_key[4][4 * CL_ROW_MAX];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int c = blockIdx.x * 4 + wid;
  if (c >= n_cl) return;  // (no block barrier below)
  unsigned long long* K = s_key[wid];
  int ir[CLW], cr[CLW];
#pragma unroll
  for (int r = 0; r < CLW; ++r) {
    const int pos = 4 * c + r;
    const bool ok = pos < n_atoms;
    ir[r] = __builtin_amdgcn_readfirstlane(ok ? (order ? order[pos] : pos) : 0);
    cr[r] = __builtin_amdgcn_readfirstlane(ok ? min(nb_cnt[ir[r]], cap) : 0);
  }
  if (lane < CLW && 4 * c + lane < n_atoms) pos_of[order ? order[4 * c + lane] : 4 * c + lane] = 4 * c + lane;
  const int o1 = cr[0], o2 = o1 + cr[1], o3 = o2 + cr[2], n = o3 + cr[3];
  const int npow = n <= 64 ? 64 : n <= 128 ? 128 : n <= 256 ? 256 : 512;
  for (int t = lane; t < npow; t += 64) {
    unsigned long long key = ~0ull;
    if (t < n) {
      const int r = (t >= o1) + (t >= o2) + (t >= o3);
      const int m = t - (r == 0 ? 0 : r == 1 ? o1 : r == 2 ? o2 : o3);
      const int i = r == 0 ? ir[0] : r == 1 ? ir[1] : r == 2 ? ir[2] : ir[3];
      const size_t p = (size_t)i * cap + m;
      const unsigned long long j = (unsigned)nb_idx[p];
      const unsigned long long code = nb_shift ? (unsigned)(nb_shift[p] & 0xffffff) : 0u;
      key = (j << 33) | (code << 9) | ((unsigned long long)m << 2) | (unsigned long long)r;
    }
    K[t] = key;
  }
  lds_sync<false>();
  for (int k = 2; k <= npow; k <<= 1) {
    for (int s = k >> 1; s > 0; s >>= 1) {
      for (int t = lane; t < npow; t += 64) {
        const int x = t ^ s;
        if (x > t) {
          const unsigned long long a = K[t], b = K[x];
          const bool up = (t & k) == 0;
          if ((a > b) == up) {
            K[t] = b;
            K[x] = a;
          }
        }
      }
      lds_sync<false>();
    }
  }
  // group heads and their rank: lane owns `per` consecutive sorted entries
  const int per = npow >> 6;
  int heads = 0;
  for (int k = 0; k < per; ++k) {
    const int t = lane * per + k;
    const unsigned long long key = K[t];
    const bool head = t < n && (t == 0 || (key >> 9) != (K[t - 1] >> 9));
    heads += head ? 1 : 0;
  }
  int incl = heads;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off, 64);
    if (lane >= off) incl += v;
  }
  int u = incl - heads;  // heads in front of this lane's entries
  const int total = __shfl(incl, 63, 64);
  const size_t base = (size_t)c * capU;
  for (int k = 0; k < per; ++k) {
    const int t = lane * per + k;
    const unsigned long long key = K[t];
    const bool head = t < n && (t == 0 || (key >> 9) != (K[t - 1] >> 9));
    if (head) {
      const float4 none = make_float4(0.f, 0.f, 0.f, -1.f);
      float4 rec0 = none, rec1 = none, rec2 = none, rec3 = none;
#pragma unroll
      for (int qn = 0; qn < CLW; ++qn) {  // the group's entries: one per centre that holds this (j, shift)
        if (t + qn < n) {
          const unsigned long long kq = K[t + qn];
          if ((kq >> 9) == (key >> 9)) {
            const int r = (int)(kq & 3), m = (int)((kq >> 2) & 127);
            const int i = r == 0 ? ir[0] : r == 1 ? ir[1] : r == 2 ? ir[2] : ir[3];
            const float4 v = pg[(size_t)i * cap + m];
            rec0 = r == 0 ? v : rec0;
            rec1 = r == 1 ? v : rec1;
            rec2 = r == 2 ? v : rec2;
            rec3 = r == 3 ? v : rec3;
          }
        }
      }
      cl_idx[base + u] = (int)(key >> 33);
      cl_shift[base + u] = (int)((key >> 9) & 0xffffff);
      float4* o = cl_ud + (base + u) * CLW;
      o[0] = rec0; o[1] = rec1; o[2] = rec2; o[3] = rec3;
      ++u;
    }
  }
  if (lane == 0) cl_cnt[c] = total;
}

// rev[(c, u, r)] = flat (cluster, entry, centre) index of the REVERSE ordered pair, -1 where centre r does not hold entry u.
// The reverse of (i -> j, shift s) is the entry (i, -s) of the cluster of j, found by bisection in its sorted union list.
__global__ __launch_bounds__(256) void cluster_rev_kernel(const int* __restrict__ cl_cnt, const int* __restrict__ cl_idx,
                                                          const int* __restrict__ cl_shift, const float4* __restrict__ cl_ud,
                                                          int capU, const int* __restrict__ order, const int* __restrict__ pos_of,
                                                          int n_atoms, int n_cl, int* __restrict__ rev, int* __restrict__ n_missing) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int c = blockIdx.x * 4 + wid;
  if (c >= n_cl) return;
  const int cnt = cl_cnt[c];
  const size_t base = (size_t)c * capU;
  int missing = 0;
  for (int t = lane; t < cnt * CLW; t += 64) {
    const int u = t >> 2, r = t & 3;
    int out = -1;
    if (cl_ud[(base + u) * CLW + r].w > 0.0f) {
      const int j = cl_idx[base + u];
      const int pj = pos_of[j];
      const int cj = pj >> 2;
      const int i = order ? order[4 * c + r] : 4 * c + r;
      const unsigned long long want = ((unsigned long long)(unsigned)i << 24) | (unsigned)neg_shift(cl_shift[base + u]);
      const size_t bj = (size_t)cj * capU;
      int lo = 0, hi = cl_cnt[cj];
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const unsigned long long km = ((unsigned long long)(unsigned)cl_idx[bj + mid] << 24) | (unsigned)cl_shift[bj + mid];
        if (km < want) lo = mid + 1; else hi = mid;
      }
      const bool found = lo < cl_cnt[cj] && cl_idx[bj + lo] == i && cl_shift[bj + lo] == (int)(want & 0xffffff) &&
                         cl_ud[(bj + lo) * CLW + (pj & 3)].w > 0.0f;
      if (found) out = (int)((bj + lo) * CLW + (pj & 3));
      else ++missing;
    }
    rev[(base + u) * CLW + r] = out;
  }
  if (missing) atomicAdd(n_missing, missing);  // a truncated (overflowed) row: the evaluation is flagged by the list status anyway
}

int launch_cluster_build(hipStream_t s, const int* nb_idx, const int* nb_shift, const int* nb_cnt, const float4* pg, int cap,
                         const int* order, int n_atoms, ClusterLists cl) {
  const int n_cl = (n_atoms + CLW - 1) / CLW;
  const int grid = ceil_div(n_cl, 4);
  hipLaunchKernelGGL(cluster_build_kernel, dim3(grid), dim3(256), 0, s, nb_idx, nb_shift, nb_cnt, pg, cap, cl.capU, order, n_atoms,
                     n_cl, cl.cnt, cl.idx, cl.shift, cl.ud, cl.pos_of);
  AIMNET_LAUNCH_CHECK();
  hipLaunchKernelGGL(cluster_rev_kernel, dim3(grid), dim3(256), 0, s, cl.cnt, cl.idx, cl.shift, cl.ud, cl.capU, order, cl.pos_of,
                     n_atoms, n_cl, cl.rev, cl.n_missing);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
constexpr int RINGU = 3;  // Sbar rows in flight per wave (static ring slots; a fourth does not fit 256 VGPRs next to abar of four centres)
constexpr int CHU = 15;   // union entries staged per chunk (a multiple of the ring depth, <= 16: one (entry, centre) per lane)
struct ClLds {
  float4 gs[CHU][G_];    // [entry][shift] -> the four centres' radial basis value
  float4 dgs[CHU][G_];   //                   and its derivative
  float4 ud[CHU][CLW];   // (u, d) per (entry, centre); absent -> (0, 0, 0, 1) with fc = dfc = 0
  float fc[CHU][CLW], dfc[CHU][CLW];
  float4 red[CHU][CLW][4];  // per (entry, centre): the four 16-lane row partials of (D, U0, U1, U2)
  int j[4 * CL_ROW_MAX];    // the cluster's whole union list: row loads never wait for a chunk's staging
};

template <int NQ, bool NEED_ABAR, bool STRESS>
__global__ __launch_bounds__(256, 2) void conv_bwd_cl_kernel(const float* __restrict__ a_t, const float* __restrict__ q,
                                                             const float* __restrict__ SbarT, const float* __restrict__ Sqbar,
                                                             const int* __restrict__ cl_cnt, const int* __restrict__ cl_idx,
                                                             const float4* __restrict__ cl_ud, int capU, BasisParams bp,
                                                             const float* __restrict__ xbar, int ldx,
                                                             const float* __restrict__ abar_in, float* __restrict__ abar_out,
                                                             const float* __restrict__ qbar_in, float* __restrict__ qbar_out,
                                                             float4* __restrict__ pairbuf, int pb_accum,
                                                             float* __restrict__ virial_atom, int n_atoms, int n_cl,
                                                             const int* __restrict__ order) {
  constexpr bool HAS_Q = NQ > 0;
  constexpr int NQC = NQ > 0 ? NQ : 1;
  __shared__ __attribute__((aligned(16))) ClLds wl[APB];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  ClLds& L = wl[wid];
  const int g = lane >> 2, c = lane & 3;
  const float m0 = c == 0 ? 1.f : 0.f, m1 = c == 1 ? 1.f : 0.f, m2 = c == 2 ? 1.f : 0.f, m3 = c == 3 ? 1.f : 0.f;
  const bool wr_lane = (lane & 12) == 12;  // lanes 12..15 of every 16-lane row hold that row's (D, U0, U1, U2) partials

  const AtomLoop al = atom_loop(n_cl, APB);  // clusters, XCD-contiguous like the atom loops
  for (int k0 = al.first; k0 < al.last; k0 += al.step) {
    const bool live = k0 + wid < al.last;
    const int cl = __builtin_amdgcn_readfirstlane(live ? k0 + wid : 0);
    const int cnt = __builtin_amdgcn_readfirstlane(live ? cl_cnt[cl] : 0);
    const size_t base = (size_t)cl * capU;
    int ir[CLW];
    bool okr[CLW];
#pragma unroll
    for (int r = 0; r < CLW; ++r) {
      const int pos = 4 * cl + r;
      okr[r] = live && pos < n_atoms;
      ir[r] = __builtin_amdgcn_readfirstlane(okr[r] ? (order ? order[pos] : pos) : 0);
    }
    // A operand of the Y chain: lane (g, r) holds a_{i_r}[a][g]
    float ai[A_];
    {
      const int irl = c == 0 ? ir[0] : c == 1 ? ir[1] : c == 2 ? ir[2] : ir[3];
      const bool okl = c == 0 ? okr[0] : c == 1 ? okr[1] : c == 2 ? okr[2] : okr[3];
      const float4* ap = reinterpret_cast<const float4*>(a_t + (size_t)irl * NF) + 4 * g;  // float4 4g + k = a_i[4k..4k+3][g]
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float4 v = okl ? ap[k] : make_float4(0.f, 0.f, 0.f, 0.f);
        ai[4 * k] = v.x; ai[4 * k + 1] = v.y; ai[4 * k + 2] = v.z; ai[4 * k + 3] = v.w;
      }
    }
    float qs[CLW][NQC];
#pragma unroll
    for (int r = 0; r < CLW; ++r)
#pragma unroll
      for (int ch = 0; ch < NQC; ++ch) qs[r][ch] = (HAS_Q && okr[r]) ? q[(size_t)ch * n_atoms + ir[r]] : 0.0f;
    for (int t = lane; t < cnt; t += 64) L.j[t] = cl_idx[base + t];
    f2 ab[CLW][8];  // abar_{i_r}[a = 2h, 2h+1][g], the lane's c-term only (summed over the quad in the epilogue)
#pragma unroll
    for (int r = 0; r < CLW; ++r)
#pragma unroll
      for (int h = 0; h < 8; ++h) ab[r][h] = mk2(0.f, 0.f);
    float qacc[CLW][NQC];
#pragma unroll
    for (int r = 0; r < CLW; ++r)
#pragma unroll
      for (int ch = 0; ch < NQC; ++ch) qacc[r][ch] = 0.0f;
    float W[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) W[k] = 0.0f;
    lds_sync<false>();

    float4 S[RINGU][4];
    float sq[RINGU][NQC];
    auto load_S = [&](int e, int slot) {  // past the list end: the last row again (valid memory, never consumed with weight)
      const int jn = __builtin_amdgcn_readfirstlane(cnt > 0 ? L.j[min(e, cnt - 1)] : 0);
      const float4* sp = reinterpret_cast<const float4*>(SbarT + (size_t)jn * (NF * 4)) + lane;
      S[slot][0] = sp[0]; S[slot][1] = sp[64]; S[slot][2] = sp[128]; S[slot][3] = sp[192];
#pragma unroll
      for (int ch = 0; ch < NQC; ++ch) sq[slot][ch] = HAS_Q ? Sqbar[((size_t)jn * NQ + ch) * (G_ * 4) + lane] : 0.0f;
    };
    auto process = [&](int el, int slot) {
      const float Sj[A_] = {S[slot][0].x, S[slot][0].y, S[slot][0].z, S[slot][0].w, S[slot][1].x, S[slot][1].y,
                            S[slot][1].z, S[slot][1].w, S[slot][2].x, S[slot][2].y, S[slot][2].z, S[slot][2].w,
                            S[slot][3].x, S[slot][3].y, S[slot][3].z, S[slot][3].w};
      f32x4 y0 = f32x4{0.f, 0.f, 0.f, 0.f}, y1 = y0;
#pragma unroll
      for (int aa = 0; aa < A_; aa += 2) {
        y0 = mfma4(ai[aa], Sj[aa], y0);
        y1 = mfma4(ai[aa + 1], Sj[aa + 1], y1);
      }
      const f32x4 Yv = y0 + y1;  // [r] at lane (g,c) = Y_{(i_r -> j)}[g,c]
      const float4 gv4 = L.gs[el][g], dgv4 = L.dgs[el][g];
      const float gvr[CLW] = {gv4.x, gv4.y, gv4.z, gv4.w}, dgvr[CLW] = {dgv4.x, dgv4.y, dgv4.z, dgv4.w};
#pragma unroll
      for (int r = 0; r < CLW; ++r) {
        const float4 u = L.ud[el][r];
        const float ucm = m0 - (m1 * u.x + m2 * u.y + m3 * u.z);  // (1, -u)_c
        const float w = gvr[r] * ucm;
        if (NEED_ABAR) {
#pragma unroll
          for (int h = 0; h < 8; ++h) ab[r][h] += w * mk2(Sj[2 * h], Sj[2 * h + 1]);
        }
        float Y = Yv[r];
        if (HAS_Q) {
#pragma unroll
          for (int ch = 0; ch < NQ; ++ch) {
            Y += qs[r][ch] * sq[slot][ch];
            qacc[r][ch] += w * sq[slot][ch];
          }
        }
        float Dl = dgvr[r] * ucm * Y;
        const float Vl = gvr[r] * Y;
        Dl += dpp0<0xB1>(Dl);  // quad sum: D gets a term from every component c
        Dl += dpp0<0x4E>(Dl);
        float Z = c == 0 ? Dl : Vl;  // lane (g,0): D of shift g;  lane (g,c>0): U_{c-1} of shift g
        Z += dpp0<0x114>(Z);         // row_shr:4, row_shr:8: lanes 12..15 of each row = sums over the row's four shifts
        Z += dpp0<0x118>(Z);
        if (wr_lane) reinterpret_cast<float*>(&L.red[el][r][lane >> 4])[c] = Z;
      }
    };

    load_S(0, 0); load_S(1, 1);
    for (int c0 = 0; c0 < cnt; c0 += CHU) {
      const int nch = min(CHU, cnt - c0);
      lds_sync<false>();  // previous chunk's tail done with ud / red
      {
        const int el = lane >> 2, r = lane & 3;  // 64 (entry, centre) slots = one per lane
        float4 ud = make_float4(0.f, 0.f, 0.f, -1.f);
        if (el < nch) ud = cl_ud[(base + c0 + el) * CLW + r];
        const bool has = ud.w > 0.0f;
        float dfc = 0.0f;
        float fc = basis_fc(bp, has ? ud.w : 1.0f, dfc);
        if (!has) {
          fc = 0.0f;
          dfc = 0.0f;
          ud = make_float4(0.f, 0.f, 0.f, 1.f);
        }
        if (el < CHU) {
          L.ud[el][r] = ud;
          L.fc[el][r] = fc;
          L.dfc[el][r] = dfc;
        }
      }
      lds_sync<false>();
#pragma unroll
      for (int t = 0; t < CHU * CLW * G_ / 64; ++t) {
        const int e = lane + 64 * t;
        const int el = e >> 6, gg = (e >> 2) & 15, r = e & 3;
        const float fc = L.fc[el][r], dfc = L.dfc[el][r];
        const float dd = L.ud[el][r].w - bp.shifts[gg];
        const float Gg = exp_neg(-bp.eta * dd * dd);
        reinterpret_cast<float*>(&L.gs[el][gg])[r] = Gg * fc;
        reinterpret_cast<float*>(&L.dgs[el][gg])[r] = Gg * (dfc - 2.0f * bp.eta * dd * fc);
      }
      lds_sync<false>();

      for (int t = 0; RINGU * t < nch; ++t) {
        const int e = c0 + RINGU * t;
        // (scheduling fences keep the entries' LDS reads and row loads from being hoisted to the top of the body)
        load_S(e + 2, 2);
        __builtin_amdgcn_sched_barrier(0);
        process(RINGU * t, 0);
        __builtin_amdgcn_sched_barrier(0);
        load_S(e + 3, 0);
        __builtin_amdgcn_sched_barrier(0);
        process(RINGU * t + 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        load_S(e + 4, 1);
        __builtin_amdgcn_sched_barrier(0);
        process(RINGU * t + 2, 2);
        __builtin_amdgcn_sched_barrier(0);
      }
      lds_sync<false>();
      // chunk tail, lane = (entry, centre): F1 of the ordered pair into the pair buffer, its virial term into W
      {
        const int el = lane >> 2, r = lane & 3;
        if (el < nch) {
          const float4 r0 = L.red[el][r][0], r1 = L.red[el][r][1], r2 = L.red[el][r][2], r3 = L.red[el][r][3];
          const bool on = el < nch;
          const float D = on ? (r0.x + r1.x) + (r2.x + r3.x) : 0.f, U0 = on ? (r0.y + r1.y) + (r2.y + r3.y) : 0.f;
          const float U1 = on ? (r0.z + r1.z) + (r2.z + r3.z) : 0.f, U2 = on ? (r0.w + r1.w) + (r2.w + r3.w) : 0.f;
          const float4 u = L.ud[el][r];
          const float inv_d = __builtin_amdgcn_rcpf(u.w);
          const float dot = U0 * u.x + U1 * u.y + U2 * u.z;
          float f0 = (U0 - dot * u.x) * inv_d - D * u.x;
          float f1 = (U1 - dot * u.y) * inv_d - D * u.y;
          float f2 = (U2 - dot * u.z) * inv_d - D * u.z;
          if (STRESS) {
            const float hx = -u.x * u.w, hy = -u.y * u.w, hz = -u.z * u.w;
            W[0] += hx * f0; W[1] += hx * f1; W[2] += hx * f2;
            W[3] += hy * f0; W[4] += hy * f1; W[5] += hy * f2;
            W[6] += hz * f0; W[7] += hz * f1; W[8] += hz * f2;
          }
          if (on) {
            float4* pb = pairbuf + (base + c0 + el) * CLW + r;
            if (pb_accum) {
              const float4 o = *pb;
              f0 += o.x; f1 += o.y; f2 += o.z;
            }
            *pb = make_float4(f0, f1, f2, 0.f);
          }
        }
      }
    }
    // ---- epilogue ---------------------------------------------------------------------------
    if (NEED_ABAR) {
#pragma unroll
      for (int r = 0; r < CLW; ++r) {
        float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < 8; ++h) {
          float vx = ab[r][h].x, vy = ab[r][h].y;
          vx += dpp0<0xB1>(vx); vx += dpp0<0x4E>(vx);
          vy += dpp0<0xB1>(vy); vy += dpp0<0x4E>(vy);
          // lane (g, c) writes a = 4c..4c+3: a = 2h -> c = h >> 1, slot (2h) & 3
          if ((h >> 1) == c) {
            o[(2 * h) & 3] = vx;
            o[(2 * h + 1) & 3] = vy;
          }
        }
        if (okr[r]) {
          const size_t row = (size_t)ir[r];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int col = (4 * c + e) * G_ + g;
            float v = o[e] + xbar[row * ldx + col];
            if (abar_in) v += abar_in[row * NF + col];
            abar_out[row * NF + col] = v;
          }
        }
      }
    }
    if (HAS_Q) {
#pragma unroll
      for (int r = 0; r < CLW; ++r)
#pragma unroll
        for (int ch = 0; ch < NQ; ++ch) {
          const float v = wave_sum(qacc[r][ch]);
          if (lane == 0 && okr[r])
            qbar_out[(size_t)ch * n_atoms + ir[r]] =
                qbar_in[(size_t)ch * n_atoms + ir[r]] + xbar[(size_t)ir[r] * ldx + 2 * NF + NV + ch] + v;
        }
    }
    if (STRESS) {
      const int irl = c == 0 ? ir[0] : c == 1 ? ir[1] : c == 2 ? ir[2] : ir[3];
      const bool okl = c == 0 ? okr[0] : c == 1 ? okr[1] : c == 2 ? okr[2] : okr[3];
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        float v = W[k];
#pragma unroll
        for (int off = 4; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);  // lanes of the same centre (lane & 3)
        if (lane < CLW && okl) virial_atom[(size_t)irl * 9 + k] += v;
      }
    }
  }
}

int launch_conv_bwd_cluster(hipStream_t s, int nq, bool need_abar, bool stress, const float* a_t, const float* q,
                            const float* SbarT, const float* Sqbar, ClusterLists cl, BasisParams bp, const float* xbar, int ldx,
                            const float* abar_in, float* abar_out, const float* qbar_in, float* qbar_out, bool pb_accum,
                            float* virial_atom, int n_atoms, const int* order) {
  const int n_cl = (n_atoms + CLW - 1) / CLW;
  const int grid = min(ceil_div(n_cl, APB), 256 * 2);
#define AIMNET_BWDC(HQ, NA, ST)                                                                                              \
  hipLaunchKernelGGL((conv_bwd_cl_kernel<HQ, NA, ST>), dim3(grid), dim3(256), 0, s, a_t, q, SbarT, Sqbar, cl.cnt, cl.idx, cl.ud, \
                     cl.capU, bp, xbar, ldx, abar_in, abar_out, qbar_in, qbar_out, cl.pairbuf, pb_accum ? 1 : 0, virial_atom,     \
                     n_atoms, n_cl, order)
#define AIMNET_BWDC2(HQ)                                                        \
  do {                                                                          \
    if (need_abar) { if (stress) AIMNET_BWDC(HQ, true, true); else AIMNET_BWDC(HQ, true, false); } \
    else { if (stress) AIMNET_BWDC(HQ, false, true); else AIMNET_BWDC(HQ, false, false); }         \
  } while (0)
  if (nq == 2) AIMNET_BWDC2(2);
  else if (nq == 1) AIMNET_BWDC2(1);
  else AIMNET_BWDC2(0);
#undef AIMNET_BWDC2
#undef AIMNET_BWDC
  AIMNET_LAUNCH_CHECK();
  return 0;
}

// dE/dx_i += sum over the ordered pairs (i -> j) of F1(i -> j) - F1(j -> i): one wave per cluster, lane = (entry, centre)
__global__ __launch_bounds__(256) void cluster_force_kernel(const int* __restrict__ cl_cnt, const int* __restrict__ rev,
                                                            const float4* __restrict__ pairbuf, int capU,
                                                            const int* __restrict__ order, int n_atoms, int n_cl,
                                                            float* __restrict__ fgrad) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int c = blockIdx.x * 4 + wid;
  if (c >= n_cl) return;
  const int cnt = cl_cnt[c];
  const size_t base = (size_t)c * capU * CLW;
  float f0 = 0.f, f1 = 0.f, f2 = 0.f;
  for (int t = lane; t < cnt * CLW; t += 64) {  // (64 % 4 == 0: a lane always serves the same centre)
    const int rv = rev[base + t];
    if (rv >= 0) {
      const float4 own = pairbuf[base + t], oth = pairbuf[rv];
      f0 += own.x - oth.x;
      f1 += own.y - oth.y;
      f2 += own.z - oth.z;
    }
  }
#pragma unroll
  for (int off = 4; off < 64; off <<= 1) {
    f0 += __shfl_xor(f0, off, 64);
    f1 += __shfl_xor(f1, off, 64);
    f2 += __shfl_xor(f2, off, 64);
  }
  const int pos = 4 * c + lane;
  if (lane < CLW && pos < n_atoms) {
    const int i = order ? order[pos] : pos;
    fgrad[3 * i + 0] += f0;
    fgrad[3 * i + 1] += f1;
    fgrad[3 * i + 2] += f2;
  }
}

int launch_cluster_force(hipStream_t s, ClusterLists cl, const int* order, int n_atoms, float* fgrad) {
  const int n_cl = (n_atoms + CLW - 1) / CLW;
  hipLaunchKernelGGL(cluster_force_kernel, dim3(ceil_div(n_cl, 4)), dim3(256), 0, s, cl.cnt, cl.rev, cl.pairbuf, cl.capU, order,
                     n_atoms, n_cl, fgrad);
  AIMNET_LAUNCH_CHECK();
  return 0;
}

}  // namespace aimnet
