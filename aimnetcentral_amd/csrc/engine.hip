// engine.hip - C ABI, device weight store, workspace layout and the launch sequence of one
// AIMNet2 energy(+force, +virial) evaluation.  See include/aimnet_hip.h for the boundary and
// DESIGN.md for the data layout; oracle/aimnet2_analytic.py is the executable specification of
// the order of operations below.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "engine.h"

namespace aimnet {

static thread_local char g_err[512] = "";
void set_last_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

}  // namespace aimnet

using namespace aimnet;

namespace {

enum { FAM_NLIST = 0, FAM_GEOM, FAM_CONV_FWD, FAM_GEMM, FAM_POINTWISE, FAM_COULOMB, FAM_UNCONCAT, FAM_CONV_BWD, FAM_OTHER, FAM_COUNT };

// mark "kernels launched from here on belong to family `fam`" (fam < 0 closes the last interval)
int prof_mark(aimnet_engine* e, hipStream_t s, int fam) {
  if (e->prof_level == 0 || !e->prof_on) return 0;
  if (e->prof_level == 1 && fam >= 0) fam = (fam == FAM_GEMM) ? FAM_GEMM : FAM_OTHER;
  if (fam == e->prof_last) return 0;
  if (e->prof_used == e->prof_ev.size()) {
    hipEvent_t ev;
    AIMNET_HIP_CHECK(hipEventCreate(&ev));
    e->prof_ev.push_back(ev);
    e->prof_fam.push_back(0);
  }
  AIMNET_HIP_CHECK(hipEventRecord(e->prof_ev[e->prof_used], s));
  e->prof_fam[e->prof_used] = fam;
  e->prof_used++;
  e->prof_last = fam;
  return 0;
}

template <typename T>
int dev_upload(aimnet_engine* e, const T* host, size_t n, T** out) {
  void* p = nullptr;
  AIMNET_HIP_CHECK(hipMalloc(&p, n * sizeof(T)));
  e->allocs.push_back(p);
  AIMNET_HIP_CHECK(hipMemcpy(p, host, n * sizeof(T), hipMemcpyHostToDevice));
  *out = (T*)p;
  return 0;
}

// k0_fwd: first k-column the forward GEMM of this layer reads by default (pass 0's first layer skips the embedding block)
int upload_layer(aimnet_engine* e, const float* w, const float* b, int n_in, int n_out, Layer* L, int k0_fwd = 0) {
  L->n_in = n_in;
  L->n_out = n_out;
  L->k_in = pad32(n_in);
  L->k_out = pad32(n_out);
  std::vector<float> wp((size_t)L->k_out * L->k_in, 0.0f), wtp((size_t)L->k_in * L->k_out, 0.0f), bp(L->k_out, 0.0f);
  for (int o = 0; o < n_out; ++o) {
    bp[o] = b[o];
    for (int i = 0; i < n_in; ++i) {
      const float v = w[(size_t)o * n_in + i];
      wp[(size_t)o * L->k_in + i] = v;
      wtp[(size_t)i * L->k_out + o] = v;
    }
  }
  int rc;
  if ((rc = dev_upload(e, wp.data(), wp.size(), &L->w))) return rc;
  if ((rc = dev_upload(e, wtp.data(), wtp.size(), &L->wt))) return rc;
  if ((rc = dev_upload(e, bp.data(), bp.size(), &L->b))) return rc;
  {  // split once, on the host (round to nearest even like v_cvt_pk_bf16_f32)
    // the last 44 % of the k-steps a launch runs over accumulate with the opposite sign (gemm_bf3.hip, "Accumulation bias").
    // 0.56 from the end-to-end energy error against the fp64 oracle (tests/tools/cfg5_ratio.py, relaxed256_probe.py): on the hot
    // config-5 frames the engine's rms distance from fp64 is 1.03 / 1.05 / 1.30 / 1.37 / 1.92 x the fp32 oracle's with the flip
    // at 0.50 / 0.56 / 0.60 / 0.64 / none (exact-fp32 kernels: 0.97); on the relaxed 256-molecule set 0.56 gives the smallest
    // worst-case error (0.66 of the un-widened gate; exact-fp32 kernels 1.05).
    std::vector<unsigned short> s3(wp.size() * 3);
    const int kb0 = k0_fwd / 32, nkf = L->k_in / 32 - kb0, nkb = L->k_out / 32;
    const char* fenv = getenv("AIMNET_BF3_FLIP");  // experiment knob: per mille of the k-steps in the first phase
    const int pm = fenv ? atoi(fenv) : 560;
    L->neg_w3 = kb0 + (pm * nkf + 500) / 1000;
    L->neg_wt3 = (pm * nkb + 500) / 1000;
    split_bf3_host(wp.data(), L->k_out, L->k_in, s3.data(), L->neg_w3);
    if ((rc = dev_upload(e, s3.data(), s3.size(), &L->w3))) return rc;
    split_bf3_host(wtp.data(), L->k_in, L->k_out, s3.data(), L->neg_wt3);
    if ((rc = dev_upload(e, s3.data(), s3.size(), &L->wt3))) return rc;
    // the operand form of gemm_bf3a.hip / gemm_head.hip: odd k-blocks negated, two interleaved accumulator sets (no fitted constant)
    split_bf3_host(wp.data(), L->k_out, L->k_in, s3.data(), BF3_ALT);
    if ((rc = dev_upload(e, s3.data(), s3.size(), &L->w3a))) return rc;
    split_bf3_host(wtp.data(), L->k_in, L->k_out, s3.data(), BF3_ALT);
    if ((rc = dev_upload(e, s3.data(), s3.size(), &L->wt3a))) return rc;
    // the operand form of gemm_h2.hip: fp16 hi + scaled fp16 lo, hi planes of the odd k-blocks negated
    std::vector<unsigned short> s2(wp.size() * 2);
    bool fits = split_h2_host(wp.data(), L->k_out, L->k_in, s2.data(), H2_WEIGHT);
    if ((rc = dev_upload(e, s2.data(), s2.size(), &L->w2a))) return rc;
    L->h_w2a = s2;
    fits = split_h2_host(wtp.data(), L->k_in, L->k_out, s2.data(), H2_WEIGHT) && fits;
    if ((rc = dev_upload(e, s2.data(), s2.size(), &L->wt2a))) return rc;
    L->h_wt2a = s2;
    if (!fits) e->h2_fits = false;
  }
  return 0;
}


// ---- chain plans (gemm_chain.hip): match an MLP sweep against the instantiated shapes and pack its weight streams ----------------
constexpr int CHAIN_NW = 8;  // waves per block of the instantiated shapes
struct PassDesc { int layer, n0, ncols, nk, nt, kb0; bool fwd; };
int chain_build(aimnet_engine* e, const std::vector<Layer>& Ls, const std::vector<PassDesc>& ps, bool bwd, int n_hidden, ChainPlan* plan) {
  plan->shape = -1;
  if (ps.empty() || (int)ps.size() > CHAIN_MAX_PASS) return 0;
  int nk[CHAIN_MAX_PASS] = {}, nt[CHAIN_MAX_PASS] = {};
  for (size_t i = 0; i < ps.size(); ++i) { nk[i] = ps[i].nk; nt[i] = ps[i].nt; }
  if (ps[0].nk > CHAIN_MAX_KB) return 0;
  const int shape = chain_find_shape(CHAIN_NW, bwd, (int)ps.size(), nk, nt, n_hidden);
  if (shape < 0) return 0;
  std::vector<unsigned short> packed;
  for (size_t i = 0; i < ps.size(); ++i) {
    const PassDesc& d = ps[i];
    const Layer& L = Ls[d.layer];
    // forward: rows of W [k_out][k_in]; backward: rows of W^T [k_in][k_out]
    const std::vector<unsigned short>& w2 = d.fwd ? L.h_w2a : L.h_wt2a;
    const int n_rows = d.fwd ? L.k_out : L.k_in, ldk = d.fwd ? L.k_in : L.k_out;
    if (w2.empty()) return 0;
    const int nta = chain_group_a(d.nt);
    for (int g = 0; g < 2; ++g) {  // column group A: the first nta tile slots of every wave, B: the rest
      const int ntg = g == 0 ? nta : d.nt - nta, s0 = g == 0 ? 0 : nta;
      plan->pass[i].w[g] = nullptr;
      if (ntg == 0) continue;
      chain_pack_weights(w2.data(), std::min(n_rows, d.n0 + d.ncols), ldk, d.n0 + 16 * CHAIN_NW * s0, CHAIN_NW, ntg, d.kb0, d.nk, packed);
      unsigned short* dev = nullptr;
      int rc = dev_upload(e, packed.data(), packed.size(), &dev);
      if (rc) return rc;
      plan->pass[i].w[g] = dev;
    }
    plan->pass[i].layer = d.layer;
    plan->pass[i].n0 = d.n0;
    plan->pass[i].ncols = d.ncols;
  }
  plan->n_pass = (int)ps.size();
  plan->shape = shape;
  return 0;
}
// forward sweep of one MLP; k0: leading input columns of the first layer that are skipped (the embedding block behind the bias table)
int chain_plan_fwd(aimnet_engine* e, const std::vector<Layer>& Ls, int k0, ChainPlan* plan) {
  std::vector<PassDesc> ps;
  const int nl = (int)Ls.size();
  for (int l = 0; l < nl; ++l) {
    const int K = Ls[l].k_in - (l == 0 ? k0 : 0), N = Ls[l].k_out;
    if (l > 0 && Ls[l].k_in != Ls[l - 1].k_out) return 0;
    ps.push_back(PassDesc{l, 0, N, K / 32, ceil_div(N / 16, CHAIN_NW), l == 0 ? k0 / 32 : 0, true});
  }
  return chain_build(e, Ls, ps, false, nl - 1, plan);
}
// backward sweep: zbar (k_out of the last layer) -> ... -> xbar (k_in of the first layer from column n0 on; wide outputs in column passes)
int chain_plan_bwd(aimnet_engine* e, const std::vector<Layer>& Ls, int n0, ChainPlan* plan) {
  std::vector<PassDesc> ps;
  const int nl = (int)Ls.size();
  for (int l = nl - 1; l >= 1; --l) ps.push_back(PassDesc{l, 0, Ls[l].k_in, Ls[l].k_out / 32, ceil_div(Ls[l].k_in / 16, CHAIN_NW), 0, false});
  const int N = Ls[0].k_in - n0, tiles = N / 16;
  const int n_col = ceil_div(tiles, 4 * CHAIN_NW);  // column passes of the last product (<= 4 tile slots per wave)
  if (tiles % n_col) return 0;
  for (int c = 0; c < n_col; ++c)
    ps.push_back(PassDesc{0, n0 + c * (N / n_col), N / n_col, Ls[0].k_out / 32, ceil_div(tiles / n_col, CHAIN_NW), 0, false});
  return chain_build(e, Ls, ps, true, nl - 1, plan);
}

}  // namespace

namespace aimnet {
// One MLP GEMM C = epilogue(A . W^T) (fwd: W = L.w [k_out][k_in]) or C = epilogue(A . W) (bwd: L.wt [k_in][k_out]) over the
// operand's k-columns [k0, k0 + K) (fwd) resp. output rows [n0, n0 + N) (bwd) - the sub-blocks the embedding-bias table and the
// pass-0 backward use.  Picks the bf16x3-split kernel or the exact-fp32 one (aimnet_engine::gemm_bf3).
int mlp_gemm(const aimnet_engine* e, hipStream_t s, int epi, const float* A, int lda, const Layer& L, bool fwd, int k0, int n0, int M,
             int N, int K, const float* bias, float* C, float* D, int ldc, const int* brow, int ldbias) {
  const int ldw = fwd ? L.k_in : L.k_out;  // row stride of the weight operand
  const bool bf3 = e->gemm_bf3 == 2 || (e->gemm_bf3 == 1 && M > 256);
  if (bf3) {
    const unsigned short* w3 = (fwd ? L.w3 : L.wt3) + (size_t)n0 * 3 * ldw + (size_t)(k0 / 32) * 96;
    const int kneg = std::max(0, (fwd ? L.neg_w3 : L.neg_wt3) - k0 / 32);  // leading k-steps of this launch with the stored sign
    return launch_gemm_bf3_cfg(s, 0, epi, A, lda, w3, 3 * ldw, M, N, K, bias, C, D, ldc, brow, ldbias, kneg);
  }
  const float* w = (fwd ? L.w : L.wt) + (size_t)n0 * ldw + k0;
  return launch_gemm_nt(s, epi, A, lda, w, ldw, M, N, K, bias, C, D, ldc, brow, ldbias);
}
int mlp_gemm3(const aimnet_engine* e, hipStream_t s, int fmt, int epi, bool out3, const unsigned short* A3, int lda3, const Layer& L,
              bool fwd, int k0, int n0, int M, int N, int K, const float* bias, float* C, unsigned short* C3, int ldc3, float* D, int ldc,
              const int* brow, int ldbias) {
  const int ldw = fwd ? L.k_in : L.k_out;
  const int alt = ((k0 / 32) & 1) ? 2 : 1;
  if (fmt == 2) {
    const unsigned short* w2 = (fwd ? L.w2a : L.wt2a) + (size_t)n0 * 2 * ldw + (size_t)(k0 / 32) * 64;
    return launch_gemm_h2_cfg(s, 0, epi, out3, A3 + (size_t)(k0 / 32) * 64, lda3, w2, 2 * ldw, M, N, K, bias, C, C3, ldc3, D, ldc, brow,
                              ldbias, alt);
  }
  const unsigned short* w3 = (fwd ? L.w3a : L.wt3a) + (size_t)n0 * 3 * ldw + (size_t)(k0 / 32) * 96;
  return launch_gemm_bf3a_cfg(s, 0, epi, out3, A3 + (size_t)(k0 / 32) * 96, lda3, w3, 3 * ldw, M, N, K, bias, C, C3, ldc3, D, ldc, brow,
                              ldbias, alt);
}
// the one-launch energy head of gemm_head.hip covers the shipped architecture (256 -> 128 -> 128 -> 1)
bool head_fusable(const aimnet_engine* e) {
  return e->head.size() == 3 && e->head[0].n_in == 256 && e->head[0].n_out == 128 && e->head[1].n_in == 128 &&
         e->head[1].n_out == 128 && e->head[2].n_in == 128 && e->head[2].n_out == 1 && e->mlp[e->arch.n_pass - 1].back().k_out == 256 &&
         !e->arch.last_linear[e->arch.n_pass - 1] && e->head_fused != 0;
}
// activations in split form for this batch? (layout() and eval() must agree)
bool presplit_active(const aimnet_engine* e, int N) {
  const bool bf3 = e->gemm_bf3 == 2 || (e->gemm_bf3 == 1 && N > 256);  // the batches that take the split GEMMs at all (mlp_gemm)
  return e->gemm_presplit != 0 && bf3 && !e->keep_intermediates && !(e->conv_mfma & 1);
}
int split_format(const aimnet_engine* e, int n_rows) {
  if (!presplit_active(e, n_rows)) return 0;
  return (e->gemm_h2 && e->h2_fits) ? 2 : 1;
}

// ---- one MLP sweep on split activations (fmt 1 = bf16x3, 2 = fp16x2): ONE launch of gemm_chain.hip where the pass has a plan
// (fp16x2 form only), else one launch per layer.  x: the input rows in split form; H[l]: layer outputs (hidden ones in split form -
// the chain does not write them -, the last in fp32, or in split form for the fused energy head: `split_last`); D[l]: GELU' (fp32).
int mlp_sweep_fwd(const aimnet_engine* e, hipStream_t s, int sfmt, int p, int N, const int* numbers, const float* x, float* const* H,
                  float* const* D, bool split_last, bool chain) {
  const aimnet_arch& ar = e->arch;
  const std::vector<Layer>& Ls = e->mlp[p];
  const int nl = (int)Ls.size(), pm = sfmt == 2 ? 2 : 3;
  const bool emb0 = p == 0 && e->emb_bias && e->emb_bias0;
  const ChainPlan& cf = e->chain_fwd[p][emb0 ? 1 : 0];
  if (sfmt == 2 && chain && cf.shape >= 0) {
    ChainArgs ca{};
    ca.x = reinterpret_cast<const unsigned short*>(x) + (emb0 ? (256 / 32) * 64 : 0);
    ca.ldx = 2 * Ls[0].k_in;
    ca.M = N;
    for (int i = 0; i < cf.n_pass; ++i) {
      const int l = cf.pass[i].layer, ko = Ls[l].k_out;
      const bool last = l == nl - 1, linear = last && ar.last_linear[p];
      ChainPass& cp = ca.p[i];
      cp.w[0] = cf.pass[i].w[0];
      cp.w[1] = cf.pass[i].w[1];
      cp.kb0 = 0;
      cp.ncols = cf.pass[i].ncols;
      cp.epi = linear ? CH_BIAS_F32 : (last && split_last) ? CH_GELU_H2G : CH_GELU_F32;
      cp.bias = (l == 0 && emb0) ? e->emb_bias0 : Ls[l].b;
      cp.brow = (l == 0 && emb0) ? numbers : nullptr;
      cp.ldbias = ko;
      cp.D = linear ? nullptr : D[l];
      cp.ldd = ko;
      cp.C = H[l];
      cp.ldc = ko;
      cp.C2 = last ? reinterpret_cast<unsigned short*>(H[l]) : nullptr;
      cp.ldc2 = 2 * ko;
    }
    return launch_gemm_chain(s, cf.shape, ca);
  }
  const unsigned short* a3 = reinterpret_cast<const unsigned short*>(x);
  int lda3 = pm * Ls[0].k_in;
  for (int l = 0; l < nl; ++l) {
    const bool last = l == nl - 1, linear = last && ar.last_linear[p];
    const bool f32out = last && !split_last;  // the last layer's output is read by pointwise kernels - or by the fused head
    const int epi = linear ? EPI_BIAS : EPI_BIAS_GELU, ko = Ls[l].k_out;
    float* Cf = f32out ? H[l] : nullptr;
    unsigned short* C3 = f32out ? nullptr : reinterpret_cast<unsigned short*>(H[l]);
    float* Dl = linear ? nullptr : D[l];
    int rc;
    if (l == 0 && emb0)
      rc = mlp_gemm3(e, s, sfmt, epi, !f32out, a3, lda3, Ls[l], true, 256, 0, N, ko, Ls[l].k_in - 256, e->emb_bias0, Cf, C3, pm * ko, Dl, ko,
                     numbers, ko);
    else
      rc = mlp_gemm3(e, s, sfmt, epi, !f32out, a3, lda3, Ls[l], true, 0, 0, N, ko, Ls[l].k_in, Ls[l].b, Cf, C3, pm * ko, Dl, ko);
    if (rc) return rc;
    a3 = C3;
    lda3 = pm * ko;
  }
  return 0;
}
// adjoint sweep: zcur = adjoint of the last layer's pre-activation (split form, GELU' applied) -> zcur = xbar (fp32, row stride
// k_in of the first layer; conv_only: only its columns 256.. are formed).  zcur / znext are the ping-pong buffers.
int mlp_sweep_bwd(const aimnet_engine* e, hipStream_t s, int sfmt, int p, int N, bool conv_only, float*& zcur, float*& znext,
                  float* const* D, bool chain) {
  const std::vector<Layer>& Ls = e->mlp[p];
  const int nl = (int)Ls.size(), pm = sfmt == 2 ? 2 : 3;
  int ld = Ls[nl - 1].k_out;
  const ChainPlan& cb = e->chain_bwd[p][(p == 0 && conv_only) ? 1 : 0];
  if (sfmt == 2 && chain && cb.shape >= 0 && !(conv_only && p != 0)) {
    ChainArgs ca{};
    ca.x = reinterpret_cast<const unsigned short*>(zcur);
    ca.ldx = 2 * ld;
    ca.M = N;
    for (int i = 0; i < cb.n_pass; ++i) {
      const int l = cb.pass[i].layer;
      ChainPass& cp = ca.p[i];
      cp.w[0] = cb.pass[i].w[0];
      cp.w[1] = cb.pass[i].w[1];
      cp.kb0 = 0;
      cp.ncols = cb.pass[i].ncols;
      if (l > 0) {  // adjoint of a hidden activation: x GELU'(z_{l-1}), stays in LDS
        cp.D = D[l - 1];
        cp.ldd = Ls[l].k_in;
      } else {  // xbar (fp32), in column passes
        cp.C = znext + cb.pass[i].n0;
        cp.ldc = Ls[0].k_in;
      }
    }
    int rc = launch_gemm_chain(s, cb.shape, ca);
    std::swap(zcur, znext);
    return rc;
  }
  for (int l = nl - 1; l >= 0; --l) {
    const Layer& L = Ls[l];
    const unsigned short* z3 = reinterpret_cast<const unsigned short*>(zcur);
    int rc;
    if (l > 0)
      rc = mlp_gemm3(e, s, sfmt, EPI_MUL, true, z3, pm * ld, L, false, 0, 0, N, L.k_in, L.k_out, nullptr, nullptr,
                     reinterpret_cast<unsigned short*>(znext), pm * L.k_in, D[l - 1], L.k_in);
    else if (conv_only)
      rc = mlp_gemm3(e, s, sfmt, EPI_NONE, false, z3, pm * ld, L, false, 0, 256, N, L.k_in - 256, L.k_out, nullptr, znext + 256, nullptr, 0,
                     nullptr, L.k_in);
    else
      rc = mlp_gemm3(e, s, sfmt, EPI_NONE, false, z3, pm * ld, L, false, 0, 0, N, L.k_in, L.k_out, nullptr, znext, nullptr, 0, nullptr,
                     L.k_in);
    if (rc) return rc;
    std::swap(zcur, znext);
    ld = L.k_in;
  }
  return 0;
}
}  // namespace aimnet

namespace {

// ---- workspace layout ---------------------------------------------------------------------------
struct Carver {
  char* base;
  size_t off = 0;
  std::map<std::string, View>* views = nullptr;
  template <typename T>
  T* take(size_t n, const char* name = nullptr, int row_stride = 0) {
    off = align_up(off, 256);
    T* p = base ? (T*)(base + off) : nullptr;
    if (name && views) (*views)[name] = View{off, n, (int)sizeof(T), row_stride};
    off += n * sizeof(T);
    return p;
  }
};

struct Workspace {
  NlistBuffers nl;
  int *nb_idx, *nb_shift, *nb_cnt;
  int *lr_idx, *lr_shift, *lr_cnt;
  float4* pg;
  float* a[AIMNET_MAX_PASS];       // features entering pass p
  float* at[AIMNET_MAX_PASS];      // the same in the MFMA operand layout (conv_mfma.hip), NULL when those kernels are off
  float* q[AIMNET_MAX_PASS];       // charges after pass p (p < n_pass-1)
  float* x[AIMNET_MAX_PASS];       // MLP input rows
  float* V[AIMNET_MAX_PASS];
  float* Vq[AIMNET_MAX_PASS];
  float* H[AIMNET_MAX_PASS][AIMNET_MAX_LAYERS];
  float* D[AIMNET_MAX_PASS][AIMNET_MAX_LAYERS];
  float* Fm[AIMNET_MAX_PASS];
  float* Dm[AIMNET_MAX_PASS];
  float* hH[AIMNET_MAX_LAYERS];
  float* hD[AIMNET_MAX_LAYERS];
  float* e_atom;
  double* ecoul;
  float *qbar, *qbar2, *fgrad, *virial_atom, *abar;
  float* qtot;   // NSE models: alpha + beta charges (the Coulomb kernels and the `charges` output see these)
  double* part;  // per-(system, slice) partial sums of the molecule reductions
  double* part_e;  // [n_mol][S] energy partial sums when the energy reduction rides on the stress launches
  int S;         // slices per molecule
  bool xe = false;   // reverse-pair conv backward: pair buffer + reverse map
  float4* pairbuf;
  int* rev;
  unsigned long long* rev_tab;  // per-atom hash tables of the rows (hash form of the reverse-pair map)
  float *zb0, *zb1;  // ping-pong adjoint buffers (N x max padded width)
  float *Sbar, *Sqbar;
  int *d3_idx, *d3_shift, *d3_cnt;   // DFT-D3 neighbour matrix (aliases the LR list when both use one cutoff)
  float *d3w, *dEdcn;                // per-atom D3 reference weights (12 floats) and dE/dCN
  float4* d3xs;                      // (x, y, z, species slot) per atom: one 16 B gather per D3 neighbour
  EwaldBuffers ew;                   // Ewald: per-system parameters, fractional coordinates, k entries (ewald.hip)
  int* bad_part;                     // per-wave input sanity flags of launch_mol_start (status array not zeroed, SrRiders::status_all)
  int* aslot;                        // species slot of every atom (pass-0 moments, DFT-D3)
  unsigned long long* present_part;  // per-block masks of the slots present
  int n_part;
  size_t total;
};

int max_width(const aimnet_engine* e) {
  int w = 32;
  for (int p = 0; p < e->arch.n_pass; ++p)
    for (const Layer& L : e->mlp[p]) w = std::max(w, std::max(L.k_in, L.k_out));
  for (const Layer& L : e->head) w = std::max(w, std::max(L.k_in, L.k_out));
  return w;
}

// The D3 list and the list-based (non-periodic) DSF list are the same neighbour matrix when their cutoffs agree
// (both default to 15 A): build and store it once.  The workspace layout does not know about periodicity, so the
// test is on the capacities the caller passed: max_nb_lr > 0 means "a DSF list will be built".
bool d3_shares_lr_list(const aimnet_eval_options* opt, int cap_lr) {
  return opt->coulomb == AIMNET_COULOMB_DSF && cap_lr > 0 && opt->d3_cutoff == opt->dsf_rc;
}

void layout(const aimnet_engine* e, int N, int n_mol, const aimnet_eval_options* opt, char* base, Workspace& W,
            std::map<std::string, View>* views) {
  Carver c{base, 0, views};
  const int np = e->arch.n_pass;
  const bool grad = (opt->flags & (AIMNET_FORCES | AIMNET_STRESS)) != 0;
  const size_t n = (size_t)N;
  const int cap = std::max(1, opt->max_nb), cap_lr = std::max(0, opt->max_nb_lr);
  const bool mfma_rows = e->conv_mfma != 0 && N > e->split_max;
  W.xe = e->conv_xe != 0 && !(e->conv_mfma & 2) && grad && np > 1 && N > e->split_max &&
         pair_rev_supported(N, cap) && n * (size_t)cap < (size_t)INT32_MAX;
  char* nl_base = c.take<char>(nlist_scratch_bytes(N, n_mol));
  if (base) nlist_carve(W.nl, nl_base, N, n_mol);
  if (views)  // the wrapped coordinates sit at a fixed position inside the nlist scratch
    (*views)["xw"] = View{(size_t)(nl_base - base) + nlist_xw_offset(n_mol), n * 3, 4, 3};
  W.nb_idx = c.take<int>(n * cap, "nb_idx", cap);
  W.nb_shift = c.take<int>(n * cap, "nb_shift", cap);
  W.nb_cnt = c.take<int>(n, "nb_cnt", 1);
  W.lr_idx = c.take<int>(n * cap_lr, "lr_idx", cap_lr);
  W.lr_shift = c.take<int>(n * cap_lr, "lr_shift", cap_lr);
  W.lr_cnt = c.take<int>(n, "lr_cnt", 1);
  {
    const bool d3 = opt->dftd3 != 0;
    const bool share = d3 && d3_shares_lr_list(opt, cap_lr);
    const int cap_d3 = d3 && !share ? std::max(1, opt->max_nb_d3) : 0;
    W.d3_idx = share ? W.lr_idx : c.take<int>(n * cap_d3, "d3_idx", cap_d3);
    W.d3_shift = share ? W.lr_shift : c.take<int>(n * cap_d3, "d3_shift", cap_d3);
    W.d3_cnt = share ? W.lr_cnt : c.take<int>(d3 ? n : 0, "d3_cnt", 1);
    W.d3w = c.take<float>(d3 ? n * 12 : 0);
    W.dEdcn = c.take<float>(d3 ? n : 0);
    W.d3xs = c.take<float4>(d3 ? n : 0);
  }
  {
    const bool pme = opt->coulomb == AIMNET_COULOMB_PME;
    const bool ew = opt->coulomb == AIMNET_COULOMB_EWALD;
    W.ew.max_k = ew ? std::max(EWALD_KB, opt->ewald_max_k / EWALD_KB * EWALD_KB) : 0;
    W.ew.sys = c.take<EwaldSystem>(ew || pme ? (size_t)n_mol : 0);
    W.ew.frac = c.take<double>(ew ? n * 3 : 0);
    W.ew.k = c.take<EwaldK>(ew ? (size_t)W.ew.max_k : 0);
    W.ew.max_mesh = pme ? std::max(512, opt->pme_max_mesh) : 0;
    W.ew.max_parts = pme ? ceil_div(W.ew.max_mesh, PME_PART) : 0;
    const size_t mesh_all = (size_t)W.ew.max_mesh * (pme ? (size_t)n_mol : 0);
    W.ew.meshq = c.take<long long>(mesh_all);
    W.ew.ma = c.take<double>(2 * mesh_all);
    W.ew.mb = c.take<double>(2 * mesh_all);
    W.ew.bmod = c.take<double>(pme ? (size_t)n_mol * 3 * PME_MAX_AXIS : 0);
    W.ew.vpart = c.take<double>(pme ? (size_t)n_mol * W.ew.max_parts * 8 : 0);
  }
  W.pg = c.take<float4>(n * cap, "pair_geom", cap);
  char name[32];
  float* x_shared = nullptr;
  float* h_shared[2] = {nullptr, nullptr};
  const bool share = !e->keep_intermediates;  // (not the pointers: in the size-query pass every pointer is NULL)
  // pre-split activations (gemm_bf3a.hip): the shared operand buffers hold 6 instead of 4 bytes per element
  const size_t ps_num = presplit_active(e, N) ? 3 : 2;
  if (share) {
    int ldx_max = 32, h_max = 32;
    for (int p = 0; p < np; ++p) {
      ldx_max = std::max(ldx_max, e->mlp[p][0].k_in);
      for (size_t l = 0; l + 1 < e->mlp[p].size(); ++l) h_max = std::max(h_max, e->mlp[p][l].k_out);
    }
    for (size_t l = 0; l + 1 < e->head.size(); ++l) h_max = std::max(h_max, e->head[l].k_out);
    x_shared = c.take<float>(n * ldx_max * ps_num / 2, "x_shared", ldx_max);
    h_shared[0] = c.take<float>(n * h_max * ps_num / 2, "h_shared0", h_max);
    h_shared[1] = c.take<float>(n * h_max * ps_num / 2, "h_shared1", h_max);
  }
  for (int p = 0; p < np; ++p) {
    snprintf(name, sizeof name, "a%d", p);
    W.a[p] = p == 0 ? nullptr : c.take<float>(n * 256, name, 256);  // pass 0 reads the embedding table itself
    W.at[p] = (p == 0 || !mfma_rows) ? nullptr : c.take<float>(n * 256);
    snprintf(name, sizeof name, "q%d", p);
    W.q[p] = c.take<float>(n * e->nq, name, 1);
    const int ldx = e->mlp[p][0].k_in;
    snprintf(name, sizeof name, "x%d", p);
    W.x[p] = share ? x_shared : c.take<float>(n * ldx, name, ldx);
    W.V[p] = c.take<float>(n * 576);
    W.Vq[p] = c.take<float>(n * 36 * e->nq);
    for (size_t l = 0; l < e->mlp[p].size(); ++l) {
      const int ld = e->mlp[p][l].k_out;
      snprintf(name, sizeof name, "h%d_%d", p, (int)l);
      const bool hidden = l + 1 < e->mlp[p].size();  // the last layer's output (q~, f~, delta_a / aim) is read again later
      // (pre-split activations: the last pass' output feeds the fused head in split form - 6 bytes per element)
      const size_t own = (!hidden && p == np - 1) ? n * ld * ps_num / 2 : n * ld;
      W.H[p][l] = (hidden && share) ? h_shared[l & 1] : c.take<float>(own, name, ld);
      snprintf(name, sizeof name, "d%d_%d", p, (int)l);
      W.D[p][l] = grad ? c.take<float>(n * ld, name, ld) : nullptr;
    }
    W.Fm[p] = c.take<float>((size_t)n_mol * e->nq);
    W.Dm[p] = c.take<float>((size_t)n_mol * e->nq);
  }
  for (size_t l = 0; l + 1 < e->head.size(); ++l) {
    const int ld = e->head[l].k_out;
    W.hH[l] = share ? h_shared[l & 1] : c.take<float>(n * ld);
    W.hD[l] = grad ? c.take<float>(n * ld) : nullptr;
  }
  W.e_atom = c.take<float>(n, "e_atom", 1);
  W.ecoul = c.take<double>(n, "ecoul", 1);
  W.qbar = c.take<float>(n * e->nq, "qbar", 1);
  W.qbar2 = c.take<float>(n * e->nq, "qbar2", 1);  // (the merged NSE adjoint writes the next pass' qbar beside the one it sums over)
  W.qtot = c.take<float>(e->nq > 1 ? n : 0, "qtot", 1);
  W.fgrad = c.take<float>(n * 3, "fgrad", 3);
  W.virial_atom = c.take<float>(n * 9);
  W.S = std::min(128, std::max(1, (N / std::max(1, n_mol) + 511) / 512));
  W.part = c.take<double>((size_t)n_mol * W.S * 9);
  W.part_e = c.take<double>((size_t)n_mol * W.S);
  if (grad) {
    const int mw = max_width(e);
    W.abar = c.take<float>(n * 256, "abar", 256);
    W.zb0 = c.take<float>(n * mw * ps_num / 2, "zb0", mw);
    W.zb1 = c.take<float>(n * mw * ps_num / 2, "zb1", mw);
    // Sbar doubles as the species-moment table T of pass 0 (N x nslots x 64), which outlives no Sbar
    W.Sbar = c.take<float>(n * std::max(1024, e->nslots * 64), "Sbar", 1024);
    W.Sqbar = c.take<float>(n * 64 * e->nq, "Sqbar", 64);
  } else {
    W.abar = W.zb0 = W.zb1 = W.Sbar = W.Sqbar = nullptr;
  }
  W.pairbuf = c.take<float4>(W.xe ? n * cap : 0);
  W.rev = c.take<int>(W.xe ? n * cap : 0);
  W.rev_tab = c.take<unsigned long long>(W.xe ? pair_hash_bytes(N) / sizeof(unsigned long long) : 0);
  W.n_part = (N + 255) / 256;
  W.aslot = c.take<int>(n);
  W.bad_part = c.take<int>((n + 63) / 64);
  W.present_part = c.take<unsigned long long>((size_t)W.n_part);
  W.total = align_up(c.off, 256);
}

}  // namespace

// ================================================================================================
extern "C" {

int aimnet_abi_version(void) { return AIMNET_ABI_VERSION; }

const char* aimnet_last_error(void) { return g_err; }

int aimnet_engine_create(const aimnet_arch* arch, const aimnet_weights* w, int device, aimnet_engine** out) {
  if (!arch || !w || !out) return AIMNET_E_INVALID;
  if (arch->nfeature != 16 || arch->nshifts != 16 || arch->ncomb_v != 12) {
    set_last_error("unsupported architecture: kernels are specialised for nfeature=16, nshifts=16, ncomb_v=12 (got %d,%d,%d)",
                   arch->nfeature, arch->nshifts, arch->ncomb_v);
    return AIMNET_E_INVALID;
  }
  if (arch->n_pass < 2 || arch->n_pass > AIMNET_MAX_PASS || arch->head_n_layers < 2 || arch->head_n_layers > AIMNET_MAX_LAYERS) {
    set_last_error("unsupported architecture: n_pass=%d head layers=%d", arch->n_pass, arch->head_n_layers);
    return AIMNET_E_INVALID;
  }
  if (arch->n_charge_channels < 0 || arch->n_charge_channels > 2) {
    set_last_error("unsupported architecture: n_charge_channels=%d (1 closed shell, 2 NSE)", arch->n_charge_channels);
    return AIMNET_E_INVALID;
  }
  AIMNET_HIP_CHECK(hipSetDevice(device));
  aimnet_engine* e = new aimnet_engine();
  e->arch = *arch;
  e->nq = std::max(1, arch->n_charge_channels);
  e->device = device;
  int rc = 0;
  const int AG = 256;
  if ((rc = dev_upload(e, w->afv, (size_t)64 * AG, &e->afv))) goto fail;
  {
    std::vector<float> t((size_t)64 * AG);
    for (int z = 0; z < 64; ++z)
      for (int aa = 0; aa < 16; ++aa)
        for (int g = 0; g < 16; ++g) t[(size_t)z * AG + g * 16 + aa] = w->afv[(size_t)z * AG + aa * 16 + g];
    if ((rc = dev_upload(e, t.data(), t.size(), &e->afv_t))) goto fail;
  }
  {
    int soz[64], zos[64];
    int ns = 0, bad_row = -1;
    for (int z = 0; z < 64; ++z) {
      bool ok = true;
      for (int k = 0; k < AG; ++k) ok = ok && std::isfinite(w->afv[(size_t)z * AG + k]);
      if (ok) { soz[z] = ns; zos[ns++] = z; } else { soz[z] = -1; if (bad_row < 0) bad_row = z; }
    }
    if (bad_row >= 0) {
      for (int z = 0; z < 64; ++z) if (soz[z] < 0) soz[z] = ns;
      zos[ns++] = bad_row;
    }
    e->nslots = ns;
    e->z_of_slot_h.assign(zos, zos + ns);
    if ((rc = dev_upload(e, soz, (size_t)64, &e->slot_of_z))) goto fail;
    if ((rc = dev_upload(e, zos, (size_t)ns, &e->z_of_slot))) goto fail;
    const char* env = getenv("AIMNET_P0_MOMENTS");
    if (env) e->p0_moments = atoi(env) != 0;
    env = getenv("AIMNET_SPATIAL_ORDER");
    if (env) e->spatial_order = atoi(env) != 0;
    env = getenv("AIMNET_KEEP_INTERMEDIATES");
    if (env) e->keep_intermediates = atoi(env) != 0;
    env = getenv("AIMNET_CONV_MFMA");
    if (env) e->conv_mfma = atoi(env);
    env = getenv("AIMNET_EMB_BIAS");
    if (env) e->emb_bias = atoi(env) != 0;
    env = getenv("AIMNET_GEMM_BF3");
    if (env) e->gemm_bf3 = std::min(2, std::max(0, atoi(env)));
    env = getenv("AIMNET_GEMM_PRESPLIT");
    if (env) e->gemm_presplit = atoi(env) != 0;
    env = getenv("AIMNET_GEMM_H2");
    if (env) e->gemm_h2 = atoi(env) != 0;
    env = getenv("AIMNET_HEAD_FUSED");
    if (env) e->head_fused = atoi(env) != 0;
    env = getenv("AIMNET_PREP_FUSED");
    if (env) e->prep_fused = atoi(env) != 0;
    env = getenv("AIMNET_ENERGY_RIDES");
    if (env) e->energy_rides = atoi(env) != 0;
    env = getenv("AIMNET_STATUS_RIDES");
    if (env) e->status_rides = atoi(env) != 0;
    env = getenv("AIMNET_SETUP_RIDES");
    if (env) e->setup_rides = atoi(env) != 0;
    env = getenv("AIMNET_STATUS_OWNED");
    if (env) e->status_owned = atoi(env) != 0;
    env = getenv("AIMNET_SUMS_WHOLE");
    if (env) e->sums_whole = atoi(env) != 0;
    env = getenv("AIMNET_NSE_MERGED");
    if (env) e->nse_merged = atoi(env) != 0;
    env = getenv("AIMNET_GEMM_CHAIN");
    if (env) e->gemm_chain = atoi(env) != 0;
    (void)gemm_h2_set_attributes();  // AIMNET_H2_TILE / AIMNET_H2_DEEP (gemm_h2.hip)
    env = getenv("AIMNET_D3_CN_RIDES");
    if (env) e->d3_cn_rides = atoi(env) != 0;
    env = getenv("AIMNET_DSF_NP_WALK");
    if (env) e->dsf_np_walk = atoi(env) != 0;
    env = getenv("AIMNET_CONV_XE");
    if (env) e->conv_xe = atoi(env);
    env = getenv("AIMNET_SPLIT_MAX");
    if (env) e->split_max = std::max(0, atoi(env));
    env = getenv("AIMNET_OVERLAP_COULOMB");
    if (env) e->overlap_coulomb = atoi(env) != 0;
  }
  if ((rc = dev_upload(e, w->agh_a, (size_t)16 * 16 * 12, &e->agh_a))) goto fail;
  if ((rc = dev_upload(e, w->agh_q, (size_t)e->nq * 16 * 12, &e->agh_q))) goto fail;
  if ((rc = dev_upload(e, w->sae, (size_t)64, &e->sae))) goto fail;
  for (int p = 0; p < arch->n_pass; ++p) {
    const int nl = arch->n_layers[p];
    if (nl < 1 || nl > AIMNET_MAX_LAYERS) { rc = AIMNET_E_INVALID; set_last_error("bad n_layers[%d]=%d", p, nl); goto fail; }
    const int n_in_expect = (p == 0) ? 704 : 704 + 29 * e->nq;  // [a | conv_a] (+ [q | conv_q] per charge channel)
    const int n_out_expect = (p < arch->n_pass - 1) ? 256 + 2 * e->nq : arch->layer_dims[p][nl];
    if (arch->layer_dims[p][0] != n_in_expect || arch->layer_dims[p][nl] != n_out_expect) {
      rc = AIMNET_E_INVALID;
      set_last_error("pass %d MLP dims %d->%d do not match the feature layout (%d->%d)", p, arch->layer_dims[p][0],
                     arch->layer_dims[p][nl], n_in_expect, n_out_expect);
      goto fail;
    }
    for (int l = 0; l < nl; ++l) {
      Layer L;
      if ((rc = upload_layer(e, w->mlp_w[p][l], w->mlp_b[p][l], arch->layer_dims[p][l], arch->layer_dims[p][l + 1], &L,
                             (p == 0 && l == 0 && arch->layer_dims[p][l] >= AG) ? AG : 0)))
        goto fail;
      e->mlp[p].push_back(L);
      if (p == 0 && l == 0 && L.n_in >= AG) {  // the embedding block of the first layer as a per-element bias table
        std::vector<float> tab((size_t)64 * L.k_out, 0.0f);
        for (int z = 0; z < 64; ++z)
          for (int o = 0; o < L.n_out; ++o) {
            double acc = (double)w->mlp_b[p][l][o];
            for (int k = 0; k < AG; ++k) acc += (double)w->mlp_w[p][l][(size_t)o * L.n_in + k] * (double)w->afv[(size_t)z * AG + k];
            tab[(size_t)z * L.k_out + o] = (float)acc;
          }
        if ((rc = dev_upload(e, tab.data(), tab.size(), &e->emb_bias0))) goto fail;
      }
    }
  }
  if (arch->head_dims[0] != arch->layer_dims[arch->n_pass - 1][arch->n_layers[arch->n_pass - 1]] ||
      arch->head_dims[arch->head_n_layers] != 1) {
    rc = AIMNET_E_INVALID;
    set_last_error("energy head dims do not chain from the last MLP / do not end in 1");
    goto fail;
  }
  for (int l = 0; l + 1 < arch->head_n_layers; ++l) {
    Layer L;
    if ((rc = upload_layer(e, w->head_w[l], w->head_b[l], arch->head_dims[l], arch->head_dims[l + 1], &L))) goto fail;
    e->head.push_back(L);
  }
  {
    const int l = arch->head_n_layers - 1;
    Layer L{};  // vector form of the final (k -> 1) layer
    L.n_in = arch->head_dims[l];
    L.n_out = 1;
    L.k_in = pad32(L.n_in);
    L.k_out = 1;
    e->head.push_back(L);
    if ((rc = dev_upload(e, w->head_w[l], (size_t)L.n_in, &e->head_w_last))) goto fail;
    if ((rc = dev_upload(e, w->head_b[l], (size_t)1, &e->head_b_last))) goto fail;
    {
      static const float unit[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
      if ((rc = dev_upload(e, unit, (size_t)9, &e->unit_cell))) goto fail;
    }
  }
  if (e->h2_fits) {  // one-launch MLP sweeps (gemm_chain.hip) where the layer sizes match an instantiated shape
    for (int p = 0; p < arch->n_pass; ++p) {
      if ((rc = chain_plan_fwd(e, e->mlp[p], 0, &e->chain_fwd[p][0]))) goto fail;
      if ((rc = chain_plan_bwd(e, e->mlp[p], 0, &e->chain_bwd[p][0]))) goto fail;
      if (p == 0 && e->emb_bias0 && (rc = chain_plan_fwd(e, e->mlp[p], AG, &e->chain_fwd[p][1]))) goto fail;
      if (p == 0 && (rc = chain_plan_bwd(e, e->mlp[p], AG, &e->chain_bwd[p][1]))) goto fail;
    }
  }
  for (int p = 0; p < arch->n_pass; ++p)
    for (Layer& L : e->mlp[p]) {
      std::vector<unsigned short>().swap(L.h_w2a);
      std::vector<unsigned short>().swap(L.h_wt2a);
    }
  for (Layer& L : e->head) {
    std::vector<unsigned short>().swap(L.h_w2a);
    std::vector<unsigned short>().swap(L.h_wt2a);
  }
  e->bp.rc = arch->rc;
  e->bp.eta = arch->eta;
  for (int g = 0; g < 16; ++g) e->bp.shifts[g] = arch->shifts[g];
  if ((rc = gemm_set_attributes())) goto fail;
  if ((rc = gemm_bf3_set_attributes())) goto fail;
  if ((rc = gemm_bf3a_set_attributes())) goto fail;
  if (hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming) != hipSuccess) {
    set_last_error("engine_create: cannot create the side stream / events");
    rc = AIMNET_E_HIP;
    goto fail;
  }
  *out = e;
  return AIMNET_OK;
fail:
  aimnet_engine_destroy(e);
  return rc;
}

void aimnet_engine_destroy(aimnet_engine* e) {
  if (!e) return;
  for (void* p : e->allocs) (void)hipFree(p);
  for (hipEvent_t ev : e->prof_ev) (void)hipEventDestroy(ev);
  if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
  if (e->ev_join) (void)hipEventDestroy(e->ev_join);
  if (e->side) (void)hipStreamDestroy(e->side);
  delete e;
}

int aimnet_engine_set_profiling(aimnet_engine* e, int level) {
  if (!e || level < 0 || level > 2) return AIMNET_E_INVALID;
  e->prof_level = level;
  e->prof_used = 0;
  e->prof_last = -2;
  e->prof_evals = e->prof_sampled = 0;
  return AIMNET_OK;
}

int aimnet_engine_set_profile_sampling(aimnet_engine* e, int every) {
  if (!e || every < 1) return AIMNET_E_INVALID;
  e->prof_every = every;
  return AIMNET_OK;
}

int aimnet_engine_profile_read(aimnet_engine* e, double* ms, int n_families, int reset) {
  if (!e || !ms || n_families < FAM_COUNT) return AIMNET_E_INVALID;
  for (int k = 0; k < n_families; ++k) ms[k] = 0.0;
  if (e->prof_used > 0) AIMNET_HIP_CHECK(hipEventSynchronize(e->prof_ev[e->prof_used - 1]));
  for (size_t k = 0; k + 1 < e->prof_used; ++k) {
    const int fam = e->prof_fam[k];
    if (fam < 0) continue;  // gap between two evals
    float t = 0.f;
    AIMNET_HIP_CHECK(hipEventElapsedTime(&t, e->prof_ev[k], e->prof_ev[k + 1]));
    ms[fam] += (double)t;
  }
  if (n_families > FAM_COUNT) ms[FAM_COUNT] = (double)e->prof_sampled;  // evaluations the sums above cover
  if (reset) {
    e->prof_used = 0;
    e->prof_last = -2;
    e->prof_evals = e->prof_sampled = 0;
  }
  return AIMNET_OK;
}

int aimnet_engine_set_option(aimnet_engine* e, const char* name, int value) {
  if (!e || !name) return AIMNET_E_INVALID;
  const std::string n(name);
  if (n == "conv_mfma") e->conv_mfma = value & 3;
  else if (n == "conv_xe") e->conv_xe = value != 0;
  else if (n == "emb_bias") e->emb_bias = value != 0;
  else if (n == "gemm_bf3") e->gemm_bf3 = std::min(2, std::max(0, value));
  else if (n == "gemm_presplit") e->gemm_presplit = value != 0;
  else if (n == "gemm_h2") e->gemm_h2 = value != 0;
  else if (n == "head_fused") e->head_fused = value != 0;
  else if (n == "prep_fused") e->prep_fused = value != 0;
  else if (n == "energy_rides") e->energy_rides = value != 0;
  else if (n == "status_rides") e->status_rides = value != 0;
  else if (n == "setup_rides") e->setup_rides = value != 0;
  else if (n == "status_owned") e->status_owned = value != 0;
  else if (n == "sums_whole") e->sums_whole = value != 0;
  else if (n == "nse_merged") e->nse_merged = value != 0;
  else if (n == "gemm_chain") e->gemm_chain = value != 0;
  else if (n == "d3_cn_rides") e->d3_cn_rides = value != 0;
  else if (n == "dsf_np_walk") e->dsf_np_walk = value != 0;
  else if (n == "split_max") e->split_max = value < 0 ? conv_split_max_default() : value;
  else if (n == "p0_moments") e->p0_moments = value != 0;
  else if (n == "overlap_coulomb") e->overlap_coulomb = value != 0;
  else if (n == "spatial_order") e->spatial_order = value != 0;
  else {
    set_last_error("set_option: unknown option '%s'", name);
    return AIMNET_E_INVALID;
  }
  return AIMNET_OK;
}

int aimnet_engine_get_option(const aimnet_engine* e, const char* name, int* value) {
  if (!e || !name || !value) return AIMNET_E_INVALID;
  const std::string n(name);
  if (n == "conv_mfma") *value = e->conv_mfma;
  else if (n == "conv_xe") *value = e->conv_xe;
  else if (n == "emb_bias") *value = e->emb_bias ? 1 : 0;
  else if (n == "gemm_bf3") *value = e->gemm_bf3;
  else if (n == "gemm_presplit") *value = e->gemm_presplit;
  else if (n == "gemm_h2") *value = e->gemm_h2 && e->h2_fits;
  else if (n == "head_fused") *value = e->head_fused;
  else if (n == "prep_fused") *value = e->prep_fused;
  else if (n == "energy_rides") *value = e->energy_rides;
  else if (n == "status_rides") *value = e->status_rides;
  else if (n == "setup_rides") *value = e->setup_rides;
  else if (n == "status_owned") *value = e->status_owned;
  else if (n == "sums_whole") *value = e->sums_whole;
  else if (n == "nse_merged") *value = e->nse_merged;
  else if (n == "gemm_chain") *value = e->gemm_chain;
  else if (n == "d3_cn_rides") *value = e->d3_cn_rides;
  else if (n == "dsf_np_walk") *value = e->dsf_np_walk;
  else if (n == "split_max") *value = e->split_max;
  else if (n == "p0_moments") *value = e->p0_moments ? 1 : 0;
  else if (n == "overlap_coulomb") *value = e->overlap_coulomb ? 1 : 0;
  else if (n == "spatial_order") *value = e->spatial_order ? 1 : 0;
  else {
    set_last_error("get_option: unknown option '%s'", name);
    return AIMNET_E_INVALID;
  }
  return AIMNET_OK;
}

int aimnet_engine_set_dd(aimnet_engine* e, const float* owned, aimnet_dd_exchange_fn fn, void* ctx) {
  if (!e || (owned && !fn)) {
    set_last_error("set_dd: an owned-atom mask needs an exchange function");
    return AIMNET_E_INVALID;
  }
  e->dd = aimnet::DdLink{owned, owned ? fn : nullptr, owned ? ctx : nullptr};
  return AIMNET_OK;
}

int aimnet_engine_set_dftd3(aimnet_engine* e, const aimnet_dftd3_tables* t) {
  if (!e || !t || t->n_z <= 0 || !t->c6ab || !t->cn_ref || !t->rcov || !t->r4r2) {
    set_last_error("set_dftd3: null argument");
    return AIMNET_E_INVALID;
  }
  AIMNET_HIP_CHECK(hipSetDevice(e->device));
  const int ns = e->nslots, nz = t->n_z;
  auto at = [&](const float* tab, int zi, int zj, int a, int b) { return tab[(((size_t)zi * nz + zj) * 5 + a) * 5 + b]; };
  std::vector<float> c6((size_t)ns * ns * 25, 0.0f), cnref((size_t)ns * 5, 0.0f), rcov(ns, 0.0f), r4r2(ns, 0.0f);
  std::vector<int> nref(ns, 0);
  for (int si = 0; si < ns; ++si) {
    const int zi = e->z_of_slot_h[si];
    if (zi >= nz || zi <= 0) continue;  // outside the table: all-zero rows (no dispersion for that slot)
    rcov[si] = t->rcov[zi];
    r4r2[si] = t->r4r2[zi];
    int n = 0;
    while (n < 5 && at(t->c6ab, zi, zi, n, n) != 0.0f) ++n;
    nref[si] = n;
    for (int a = 0; a < n; ++a) cnref[(size_t)si * 5 + a] = at(t->cn_ref, zi, zi, a, 0);
  }
  for (int si = 0; si < ns; ++si)
    for (int sj = 0; sj < ns; ++sj) {
      const int zi = e->z_of_slot_h[si], zj = e->z_of_slot_h[sj];
      if (zi >= nz || zj >= nz || zi <= 0 || zj <= 0) continue;
      for (int a = 0; a < 5; ++a)
        for (int b = 0; b < 5; ++b) {
          const float v = at(t->c6ab, zi, zj, a, b);
          const bool expect = a < nref[si] && b < nref[sj];
          if ((v != 0.0f) != expect || (expect && at(t->cn_ref, zi, zj, a, b) != cnref[(size_t)si * 5 + a])) {
            set_last_error("set_dftd3: the C6/CN table of Z = %d, %d does not factorise (ref %d, %d)", zi, zj, a, b);
            return AIMNET_E_INVALID;
          }
          c6[(((size_t)si * ns + sj) * 5 + a) * 5 + b] = v;
        }
    }
  int rc;
  float *d_c6, *d_cn, *d_rc, *d_r4;
  int* d_nr;
  if ((rc = dev_upload(e, c6.data(), c6.size(), &d_c6))) return rc;
  if ((rc = dev_upload(e, cnref.data(), cnref.size(), &d_cn))) return rc;
  if ((rc = dev_upload(e, nref.data(), nref.size(), &d_nr))) return rc;
  if ((rc = dev_upload(e, rcov.data(), rcov.size(), &d_rc))) return rc;
  if ((rc = dev_upload(e, r4r2.data(), r4r2.size(), &d_r4))) return rc;
  e->d3 = D3Tables{ns, d_c6, d_cn, d_nr, d_rc, d_r4};
  return AIMNET_OK;
}

size_t aimnet_engine_workspace_bytes(const aimnet_engine* e, int32_t n_atoms, int32_t n_mol, int32_t n_cell,
                                     const aimnet_eval_options* opt) {
  (void)n_cell;
  if (!e || !opt || n_atoms <= 0 || n_mol <= 0) return 0;
  Workspace W;
  layout(e, n_atoms, n_mol, opt, nullptr, W, nullptr);
  return W.total;
}

int aimnet_engine_debug_view(const aimnet_engine* e, const char* name, size_t* byte_offset, size_t* n_elem,
                             int32_t* elem_size, int32_t* row_stride) {
  if (!e || !name) return AIMNET_E_INVALID;
  auto it = e->views.find(name);
  if (it == e->views.end()) return AIMNET_E_INVALID;
  if (byte_offset) *byte_offset = it->second.off;
  if (n_elem) *n_elem = it->second.n_elem;
  if (elem_size) *elem_size = it->second.elem_size;
  if (row_stride) *row_stride = it->second.row_stride;
  return AIMNET_OK;
}

#define RC(call)            \
  do {                      \
    int _rc = (call);       \
    if (_rc) return _rc;    \
  } while (0)

int aimnet_engine_eval(aimnet_engine* e, const aimnet_inputs* in, const aimnet_eval_options* opt,
                       const aimnet_outputs* out, void* workspace, size_t workspace_bytes, void* hip_stream) {
  if (!e || !in || !opt || !out || !workspace) return AIMNET_E_INVALID;
  const int N = in->n_atoms, n_mol = in->n_mol;
  if (N <= 0 || n_mol <= 0 || !in->coord || !in->numbers || !in->mol_idx || !in->charge || !out->energy ||
      !out->charges || !out->status) {
    set_last_error("eval: null or empty input/output");
    return AIMNET_E_INVALID;
  }
  const bool want_f = (opt->flags & AIMNET_FORCES) != 0, want_s = (opt->flags & AIMNET_STRESS) != 0;
  const bool grad = want_f || want_s;
  const bool pbc = in->cell != nullptr;
  if (want_s && ((!pbc && !e->dd.owned) || !out->stress)) {  // (domain decomposition: the rank's virial, see aimnet_engine_set_dd)
    set_last_error("eval: stress requires a cell and a stress buffer");
    return AIMNET_E_INVALID;
  }
  if (want_f && !out->forces) {
    set_last_error("eval: forces requested without a forces buffer");
    return AIMNET_E_INVALID;
  }
  const int nq = e->nq;
  if (nq == 1 && out->spin_charges) {
    set_last_error("eval: spin_charges requested from a 1-channel (closed-shell) model");
    return AIMNET_E_INVALID;
  }
  if (pbc && !(in->n_cell == 1 || in->n_cell == n_mol)) {
    set_last_error("eval: n_cell must be 1 or n_mol");
    return AIMNET_E_INVALID;
  }
  int coulomb = opt->coulomb;
  // large non-periodic systems get a bounding-box cell grid (launch_bbox below): DSF walks it like a periodic cell's grid, no matrix
  const bool np_walk = e->dsf_np_walk && !pbc && in->nbmat == nullptr && coulomb == AIMNET_COULOMB_DSF && (long)N >= 1500L * n_mol &&
                       !(opt->dftd3 != 0 && opt->d3_cutoff == opt->dsf_rc);
  if (coulomb == AIMNET_COULOMB_DSF && !pbc && opt->max_nb_lr <= 0 && !np_walk) {
    set_last_error("eval: non-periodic DSF Coulomb needs max_nb_lr > 0 (periodic DSF walks the cell grid, no list)");
    return AIMNET_E_INVALID;
  }
  if (coulomb == AIMNET_COULOMB_SIMPLE && pbc) {
    set_last_error("eval: 'simple' Coulomb is undefined for periodic input (host must switch to DSF, calculator.py:1044)");
    return AIMNET_E_INVALID;
  }
  if (coulomb == AIMNET_COULOMB_EWALD || coulomb == AIMNET_COULOMB_PME) {
    if (!pbc || in->pbc_sys || !(in->pbc[0] && in->pbc[1] && in->pbc[2])) {
      set_last_error("eval: Ewald summation needs a cell that is periodic along all three axes (lr.py:655-657)");
      return AIMNET_E_INVALID;
    }
    if (in->nbmat) {
      set_last_error("eval: Ewald summation walks the engine's own cell grid: caller-supplied neighbour matrices are not taken with it "
                     "(the reference builds its own per-call list for this method too, calculator.py:1560-1603)");
      return AIMNET_E_INVALID;
    }
    if (!(opt->ewald_accuracy > 0.0f && opt->ewald_accuracy < 1.0f) ||
        (coulomb == AIMNET_COULOMB_EWALD ? opt->ewald_max_k < EWALD_KB : opt->pme_max_mesh < 512)) {
      set_last_error("eval: Ewald summation needs 0 < ewald_accuracy < 1 and ewald_max_k >= %d (PME: pme_max_mesh >= 512)", EWALD_KB);
      return AIMNET_E_INVALID;
    }
  }
  // caller-supplied neighbour matrices (aimnet_inputs.nbmat ...): no list is built, the coordinates are taken as given
  const bool ext = in->nbmat != nullptr;
  if (ext) {
    if (in->nbmat_width <= 0 || opt->max_nb < in->nbmat_width) {
      set_last_error("eval: caller-supplied nbmat needs 0 < nbmat_width <= options.max_nb (got %d, %d)", in->nbmat_width, opt->max_nb);
      return AIMNET_E_INVALID;
    }
    if (pbc && !in->shifts) {
      set_last_error("eval: a caller-supplied nbmat of a periodic system needs its shifts");
      return AIMNET_E_INVALID;
    }
    if (in->nbmat_lr && (in->nbmat_lr_width <= 0 || opt->max_nb_lr < in->nbmat_lr_width || (pbc && !in->shifts_lr))) {
      set_last_error("eval: caller-supplied nbmat_lr needs 0 < nbmat_lr_width <= options.max_nb_lr, and shifts_lr when periodic");
      return AIMNET_E_INVALID;
    }
    if (coulomb == AIMNET_COULOMB_DSF && !in->nbmat_lr) {
      set_last_error("eval: DSF Coulomb with a caller-supplied nbmat needs nbmat_lr as well (no list is built in this mode)");
      return AIMNET_E_INVALID;
    }
    if (opt->dftd3 != 0) {
      const bool own = in->nbmat_d3 != nullptr;
      if (!own && !in->nbmat_lr) {
        set_last_error("eval: DFT-D3 with a caller-supplied nbmat needs nbmat_d3 or nbmat_lr");
        return AIMNET_E_INVALID;
      }
      if (own && (in->nbmat_d3_width <= 0 || (pbc && !in->shifts_d3))) {
        set_last_error("eval: caller-supplied nbmat_d3 needs a width, and shifts_d3 when periodic");
        return AIMNET_E_INVALID;
      }
    }
  } else if (in->nbmat_lr || in->nbmat_d3) {
    set_last_error("eval: nbmat_lr / nbmat_d3 are only read together with nbmat");
    return AIMNET_E_INVALID;
  }
  // spatial domain decomposition (aimnet_engine_set_dd): the local cluster of owned + halo atoms is a non-periodic system
  const aimnet::DdLink* dd = e->dd.owned ? &e->dd : nullptr;
  if (dd && (pbc || ext || !(coulomb == AIMNET_COULOMB_NONE || coulomb == AIMNET_COULOMB_DSF))) {
    set_last_error("eval: a domain-decomposed evaluation takes a non-periodic cluster (no cell, no caller-supplied lists), Coulomb "
                   "'none' or 'dsf'");
    return AIMNET_E_INVALID;
  }
  hipStream_t s = (hipStream_t)hip_stream;
  AIMNET_HIP_CHECK(hipSetDevice(e->device));
  Workspace W;
  e->views.clear();
  layout(e, N, n_mol, opt, (char*)workspace, W, &e->views);
  if (W.total > workspace_bytes) {
    set_last_error("eval: workspace too small (%zu < %zu)", workspace_bytes, W.total);
    return AIMNET_E_WORKSPACE;
  }
  const aimnet_arch& ar = e->arch;
  const int np = ar.n_pass;
  const int cap = std::max(1, opt->max_nb), cap_lr = std::max(0, opt->max_nb_lr);
  const int n_cell = pbc ? in->n_cell : 0;

  // ---- neighbour lists + pair geometry ------------------------------------------------------
  e->prof_on = e->prof_level > 0 && (e->prof_evals++ % e->prof_every) == 0;
  if (e->prof_on) e->prof_sampled++;
  RC(prof_mark(e, s, FAM_NLIST));
  const bool want_species = (e->p0_moments && (opt->flags & (AIMNET_FORCES | AIMNET_STRESS))) || opt->dftd3 != 0;
  // small periodic batches: status zeroing, molecule offsets / sanity / species, cell + bin setup, wrapping and binning in one launch
  const bool prep1 = e->prep_fused && !ext && prep_small_applies(N, n_mol, pbc);
  bool setup_rides = false, status_owned = false;
  if (prep1) {
    RC(launch_prep_small(s, in->coord, in->mol_idx, in->numbers, N, n_mol, pbc ? in->cell : nullptr, n_cell, in->pbc, in->pbc_sys, ar.rc,
                         out->status,
                         want_species ? e->slot_of_z : nullptr, W.aslot, W.present_part, W.nl));
  } else {
    // One list, its status words reduced by a rider of the SR-Coulomb launch: that rider can just as well STORE all eight status
    // words (with the sanity flags of the launch below collected per wave), and nothing has to be zeroed in front of the evaluation.
    const bool one_list = !ext && !(coulomb == AIMNET_COULOMB_DSF && !pbc) && opt->dftd3 == 0;
    status_owned = e->status_rides && e->status_owned && one_list && N <= 32768;
    if (!status_owned) AIMNET_HIP_CHECK(hipMemsetAsync(out->status, 0, 8 * sizeof(int), s));
    // periodic fast path: the cell + bin-grid setup block rides on this launch (it needs none of its output)
    setup_rides = e->setup_rides && !ext && pbc && cell_setup_rides(N, n_mol);
    CellSetupRider csr{};
    if (setup_rides) csr = cell_setup_rider(in->cell, n_cell, in->pbc, in->pbc_sys, ar.rc, N, n_mol, W.nl);
    RC(launch_mol_start(s, in->mol_idx, N, n_mol, W.nl.mol_start, W.nl.mol_c, in->numbers, out->status + 6,
                        want_species ? e->slot_of_z : nullptr, W.aslot, W.present_part,  // + aslot / present species
                        setup_rides ? &csr : nullptr, status_owned ? W.bad_part : nullptr));
  }
  const int* mol_c = W.nl.mol_c;  // clamped to [0, n_mol): memory-safe whatever the caller passed (status[6] reports it)
  const bool d3 = opt->dftd3 != 0;
  if (d3 && e->d3.ns == 0) {
    set_last_error("eval: DFT-D3 requested but aimnet_engine_set_dftd3 was never called");
    return AIMNET_E_INVALID;
  }
  int cap_d3 = cap_lr;
  bool d3_shared = false, d3_cn_done = false;
  const int* sr_cnt_true = nullptr;  // != NULL: the short-range list's status words are still to be reduced (SrRiders)
  if (ext) {
    // the reference hands a caller's matrices to the model as they are (calculator.py:1069-1071): import them into the row format
    // of the kernels; coordinates as given (the shifts refer to them), no bins, centres processed in input order
    RC(launch_wrap(s, in->coord, mol_c, N, n_mol, nullptr, 0, in->pbc, W.nl));
    RC(launch_import_list(s, in->nbmat, pbc ? in->shifts : nullptr, in->nbmat_width, N, mol_c, in->cell, n_cell, cap, W.nl, W.nb_idx,
                          W.nb_shift, W.nb_cnt, out->status + 0, out->status + 2, W.pg, out->status + 6));
    RC(launch_list_symmetry_check(s, W.nb_idx, pbc ? W.nb_shift : nullptr, W.nb_cnt, cap, N, out->status + 6));
    if (in->nbmat_lr && coulomb != AIMNET_COULOMB_NONE) {
      RC(launch_import_list(s, in->nbmat_lr, pbc ? in->shifts_lr : nullptr, in->nbmat_lr_width, N, mol_c, in->cell, n_cell, cap_lr, W.nl,
                            W.lr_idx, W.lr_shift, W.lr_cnt, out->status + 1, out->status + 3, nullptr, out->status + 6));
      RC(launch_list_symmetry_check(s, W.lr_idx, pbc ? W.lr_shift : nullptr, W.lr_cnt, cap_lr, N, out->status + 6, 64));
    }
    if (d3) {
      d3_shared = d3_shares_lr_list(opt, cap_lr);  // one cutoff for both: the layout stores ONE matrix
      const int* src = in->nbmat_d3 ? in->nbmat_d3 : in->nbmat_lr;
      const int* src_sh = in->nbmat_d3 ? in->shifts_d3 : in->shifts_lr;
      const int src_w = in->nbmat_d3 ? in->nbmat_d3_width : in->nbmat_lr_width;
      if (d3_shared) {
        if (src != in->nbmat_lr) {
          set_last_error("eval: with d3_cutoff == dsf_rc one caller-supplied matrix serves both terms: pass it as nbmat_lr only");
          return AIMNET_E_INVALID;
        }
      } else {
        cap_d3 = std::max(1, opt->max_nb_d3);
        if (cap_d3 < src_w) {
          set_last_error("eval: options.max_nb_d3 (%d) is smaller than the caller-supplied D3 matrix (%d)", cap_d3, src_w);
          return AIMNET_E_INVALID;
        }
        RC(launch_import_list(s, src, pbc ? src_sh : nullptr, src_w, N, mol_c, in->cell, n_cell, cap_d3, W.nl, W.d3_idx, W.d3_shift,
                              W.d3_cnt, out->status + 4, out->status + 5, nullptr, out->status + 6));
        RC(launch_list_symmetry_check(s, W.d3_idx, pbc ? W.d3_shift : nullptr, W.d3_cnt, cap_d3, N, out->status + 6, 64));
      }
    }
  } else {
  if (!prep1) RC(launch_wrap(s, in->coord, mol_c, N, n_mol, in->cell, n_cell, in->pbc, W.nl, in->pbc_sys, pbc ? ar.rc : 0.0f, setup_rides));
  // large non-periodic molecules (>= 1500 atoms on average) get a bounding-box cell list instead of the O(n^2) scan
  if (!pbc && (long)N >= 1500L * n_mol) RC(launch_bbox(s, n_mol, W.nl));
  // no second list build follows (periodic DSF walks the grid, "simple" sums all pairs, no D3 list): the status words of this list
  // are reduced by rider blocks of the SR-Coulomb launch instead of a launch of their own
  const bool status_rides = e->status_rides && !(coulomb == AIMNET_COULOMB_DSF && !pbc) && !d3;
  RC(launch_nlist(s, N, n_mol, mol_c, in->cell, n_cell, in->pbc, ar.rc, ar.rc, cap, N, 0, W.nl, W.nb_idx, W.nb_shift,
                  W.nb_cnt, out->status + 0, out->status + 2, W.pg, status_rides ? &sr_cnt_true : nullptr));
  if (coulomb == AIMNET_COULOMB_DSF && !pbc && !np_walk)  // periodic DSF (and large non-periodic systems) need no list: they walk the short-range cell grid
    RC(launch_nlist(s, N, n_mol, mol_c, in->cell, n_cell, in->pbc, opt->dsf_rc, -1.0f, cap_lr, N, 0, W.nl, W.lr_idx,
                    W.lr_shift, W.lr_cnt, out->status + 1, out->status + 3));
  d3_shared = d3 && d3_shares_lr_list(opt, cap_lr) && !pbc;
  if (d3 && !d3_shared) {
    D3CnRider cnr;  // the coordination numbers ride on the (cell-grid) build of the D3 matrix
    if (e->d3_cn_rides) {
      cnr.aslot = W.aslot; cnr.rcov = e->d3.rcov; cnr.nref = e->d3.nref; cnr.cnref = e->d3.cnref; cnr.d3w = W.d3w;
    }
    if (d3_shares_lr_list(opt, cap_lr)) {  // periodic DSF walks the grid: the shared buffers are free for the D3 list
      RC(launch_nlist(s, N, n_mol, mol_c, in->cell, n_cell, in->pbc, opt->d3_cutoff, -1.0f, cap_lr, N, 0, W.nl, W.d3_idx,
                      W.d3_shift, W.d3_cnt, out->status + 4, out->status + 5, nullptr, nullptr, &cnr, &d3_cn_done));
    } else {
      cap_d3 = std::max(1, opt->max_nb_d3);
      RC(launch_nlist(s, N, n_mol, mol_c, in->cell, n_cell, in->pbc, opt->d3_cutoff, -1.0f, cap_d3, N, 0, W.nl, W.d3_idx,
                      W.d3_shift, W.d3_cnt, out->status + 4, out->status + 5, nullptr, nullptr, &cnr, &d3_cn_done));
    }
  }
  }
  RC(prof_mark(e, s, FAM_GEOM));
  // (the pair geometry (u, d) of the short-range list was written by the list builder itself)

  // ---- forward --------------------------------------------------------------------------------
  // binned systems: process centre atoms in the bin-sorted order of the cell list (kernels.h, `order`)
  const int* order = (!ext && W.nl.binned && e->spatial_order) ? W.nl.sorted : nullptr;
  // a^0 = afv[Z] is never materialised: pass 0 gathers the embedding rows directly (conv_fwd / conv_bwd row_of, update_a)
  const bool p0m = e->p0_moments && (opt->flags & (AIMNET_FORCES | AIMNET_STRESS));
  const int sfmt = split_format(e, N);  // GEMM activations in split form: 1 bf16x3 (gemm_bf3a.hip), 2 fp16x2 (gemm_h2.hip)
  const bool ps = sfmt != 0;
  const int pm = sfmt == 2 ? 2 : 3;  // 16-bit elements per fp32 value of a split row
  const bool hfused = ps && head_fusable(e);  // energy head forward + backward in one launch (gemm_head.hip)
  const bool mfma_fwd = (e->conv_mfma & 1) && N > e->split_max;
  const bool mfma_bwd = (e->conv_mfma & 2) && N > e->split_max;
  // reverse-pair map through per-atom hash tables of the rows (once per neighbour list)
  // (its only reader is launch_pair_force, the last kernel of the backward.  On one stream the two small kernels ride on later
  // launches instead of standing in front of the forward pass: the hash build on the SR-Coulomb launch, the lookup on the DSF walk
  // (VALU-bound, the lookup is latency-bound) or else on the energy reduction - kernels.h PairMapRider)
  PairMapRider pmap{};
  const bool overlap_early = e->overlap_coulomb && e->prof_level < 2 && !dd;
  if (W.xe && want_f) {
    if (overlap_early) {
      RC(launch_pair_rev_hash(s, W.nb_idx, n_cell > 0 ? W.nb_shift : nullptr, W.nb_cnt, cap, N, W.rev_tab, W.rev));
    } else {
      pmap = PairMapRider{W.nb_idx, n_cell > 0 ? W.nb_shift : nullptr, W.nb_cnt, cap, N, W.rev_tab, W.rev, ceil_div(N, 4)};
    }
  }
  bool rev_done = pmap.n_blocks == 0;
  // ---- Coulomb: energies, and the seeds of qbar / dE/dx / virial (a closure: it runs on the eval stream or on the side one) ----
  const float* q_fin = nq == 2 ? W.qtot : W.q[np - 2];
  const bool overlap = e->overlap_coulomb && e->prof_level < 2 && !dd;  // per-family profiling wants one stream
  bool charges_written = false;  // the DSF walk's charge stream kernel copies q to the `charges` output on its way
  SrRiders head_rider{};  // the last energy-head layer rides on the SR-Coulomb launch (filled in below when both run on one stream)
  auto coulomb_block = [&](hipStream_t cs) -> int {
    if (nq == 2)  // NSE: alpha + beta is the charge everything downstream sees (aimnet2.py:102-106)
      RC(launch_charge_sum(cs, W.q[np - 2], N, W.qtot, out->spin_charges));
    CoulombParams cp;
    cp.factor = (float)(0.5 * 27.211386024367243 * 0.5291772105638411);
    cp.sr_rc = ar.sr_rc;
    cp.sr_envelope = ar.sr_envelope;
    cp.dsf_rc = opt->dsf_rc;
    cp.dsf_alpha = opt->dsf_alpha;
    const bool pme = coulomb == AIMNET_COULOMB_PME;
    const bool ewald = coulomb == AIMNET_COULOMB_EWALD || pme;  // (the real-space walk and the self term are the same)
    if (pme) {  // per-system (alpha, rc, mesh) from the cell, fractional coordinates in double (pme.hip)
      RC(launch_pme_setup(cs, in->cell, n_cell, W.nl.mol_start, in->charge, nq, n_mol, opt->ewald_accuracy, W.ew, out->status + 7));
      cp.ewald = W.ew.sys;
    } else if (ewald) {  // per-system (alpha, rc, kc) and k boxes from the cell, fractional coordinates in double (ewald.hip)
      RC(launch_ewald_setup(cs, in->cell, n_cell, W.nl.mol_start, mol_c, W.nl.xw, in->charge, nq, N, n_mol, opt->ewald_accuracy, W.ew,
                            out->status + 7));
      cp.ewald = W.ew.sys;
    }
    const bool walk = ewald || np_walk || (coulomb == AIMNET_COULOMB_DSF && pbc && !ext &&
                                           !(d3 && opt->d3_cutoff == opt->dsf_rc));  // the list-free walk runs below: its charge stream rides here
    SrRiders rd = head_rider;
    if (walk) {
      rd.xs = W.nl.xs;
      rd.xq = (float4*)W.nl.sorted_tmp_xq;
      rd.charges_out = out->charges;
      rd.n_stream_blocks = ceil_div(N, 256);
    }
    const bool simple_all = coulomb == AIMNET_COULOMB_SIMPLE && !(ext && in->nbmat_lr);  // all pairs of the molecule: same waves
    if (simple_all) {
      rd.simple_xw = W.nl.xw;
      rd.simple_mol_idx = mol_c;
      rd.simple_mol_start = W.nl.mol_start;
    }
    rd.hash = pmap;  // hash build of the reverse-pair map (n_blocks = 0: none)
    if (sr_cnt_true) {
      rd.cnt_true = sr_cnt_true;
      rd.status_cap = cap;
      rd.status_max = out->status + 0;
      rd.status_ovf = out->status + 2;
      rd.n_status_blocks = ceil_div(N, 1024);
      if (status_owned) {
        rd.n_status_blocks = 1;
        rd.status_all = out->status;
        rd.bad_part = W.bad_part;
        rd.keep7 = ewald ? 1 : 0;
      }
    }
    RC(launch_coulomb_sr(cs, grad, want_s, ar.sr_coulomb != 0, q_fin, W.nb_idx, W.nb_cnt, W.pg, cap, cp, N, W.ecoul, W.qbar,
                         W.fgrad, W.virial_atom, &rd));
    // DSF and DFT-D3 with one cutoff: the Coulomb pair terms ride on the D3 pair pass (one list, one geometry evaluation)
    const bool dsf_in_d3 = !ext && d3 && coulomb == AIMNET_COULOMB_DSF && opt->d3_cutoff == opt->dsf_rc;
    if (coulomb == AIMNET_COULOMB_SIMPLE && ext && in->nbmat_lr)  // coul_simple over the caller's matrix (lr.py:311-331)
      RC(launch_coulomb_dsf(cs, grad, false, q_fin, W.nl.xw, mol_c, in->cell, n_cell, W.lr_idx, W.lr_shift, W.lr_cnt, cap_lr, cp, N,
                            W.ecoul, W.qbar, W.fgrad, W.virial_atom, true));
    else if (coulomb == AIMNET_COULOMB_SIMPLE)
      ;  // ran inside the SR-Coulomb launch above (SrRiders::simple_xw)
    else if (dsf_in_d3)
      ;  // see launch_dftd3 below
    else if (ewald || np_walk || (coulomb == AIMNET_COULOMB_DSF && pbc && !ext)) {
      RC(launch_coulomb_dsf_walk(cs, grad, want_s, q_fin, mol_c, W.nl, cp, N, W.ecoul, W.qbar, W.fgrad, W.virial_atom,
                                 out->charges, true, rev_done ? nullptr : &pmap));
      rev_done = true;
      charges_written = true;
      if (pme)  // reciprocal space on the mesh + neutralising background (pme.hip)
        RC(launch_pme_recip(cs, grad, want_s, W.nl.xw, q_fin, mol_c, W.nl.mol_start, order, N, n_mol, W.ew, cp.factor, W.ecoul, W.qbar, W.fgrad,
                            W.virial_atom));
      else if (ewald)  // reciprocal space + neutralising background, accumulated onto what the pair kernels have stored
        RC(launch_ewald_recip(cs, grad, want_s, q_fin, mol_c, W.nl.mol_start, N, n_mol, W.ew, cp.factor, W.ecoul, W.qbar, W.fgrad,
                              W.virial_atom));
    } else if (coulomb == AIMNET_COULOMB_DSF)
      RC(launch_coulomb_dsf(cs, grad, want_s, q_fin, W.nl.xw, mol_c, in->cell, n_cell, W.lr_idx, W.lr_shift, W.lr_cnt,
                            cap_lr, cp, N, W.ecoul, W.qbar, W.fgrad, W.virial_atom));
    if (d3) {  // external DFT-D3: adds to the per-atom pair energies, dE/dx and the virial seeded by the Coulomb kernels
      D3Params dp;
      dp.s6 = opt->d3_s6; dp.s8 = opt->d3_s8; dp.a1 = opt->d3_a1; dp.a2 = opt->d3_a2;
      dp.r_on = opt->d3_smoothing_on * 1.8897261258369282f;
      dp.r_off = opt->d3_cutoff * 1.8897261258369282f;
      RC(launch_dftd3(cs, grad, want_s, W.nl.xw, mol_c, in->cell, n_cell, W.aslot, W.d3_idx, W.d3_shift, W.d3_cnt, cap_d3,
                      e->d3, dp, opt->d3_cutoff, N, W.d3xs, W.d3w, W.dEdcn, W.ecoul, W.fgrad, W.virial_atom, dsf_in_d3, cp, q_fin,
                      W.qbar, d3_cn_done, dd));
    }
    if (grad && nq == 2) RC(launch_copy_f32(cs, W.qbar, W.qbar + N, (size_t)N));  // dE/dq_alpha = dE/dq_beta = dE/dq at this point
    return 0;
  };

  for (int p = 0; p < np; ++p) {
    const std::vector<Layer>& Ls = e->mlp[p];
    const int nl = (int)Ls.size();
    RC(prof_mark(e, s, FAM_CONV_FWD));
    if (mfma_fwd)
      RC(launch_conv_fwd_mfma(s, p > 0 ? nq : 0, p == 0 ? e->afv : W.a[p], p == 0 ? e->afv_t : W.at[p], p == 0 ? in->numbers : nullptr,
                              p > 0 ? W.q[p - 1] : nullptr, W.nb_idx, W.nb_cnt, W.pg, cap, e->agh_a, e->agh_q, e->bp, W.x[p],
                              Ls[0].k_in, W.V[p], W.Vq[p], N, order));
    else
      RC(launch_conv_fwd(s, p > 0 ? nq : 0, p == 0 ? e->afv : W.a[p], p == 0 ? in->numbers : nullptr, p > 0 ? W.q[p - 1] : nullptr,
                         W.nb_idx, W.nb_cnt, W.pg, cap, e->agh_a, e->agh_q, e->bp, W.x[p], Ls[0].k_in, W.V[p], W.Vq[p], N, order,
                         p == 0 && e->p0_moments, e->split_max, sfmt));
    const float* hin = W.x[p];
    int ld_in = Ls[0].k_in;
    RC(prof_mark(e, s, FAM_GEMM));
    if (ps) {  // activations in split form: one launch per MLP (gemm_chain.hip) or one per layer
      RC(mlp_sweep_fwd(e, s, sfmt, p, N, in->numbers, W.x[p], W.H[p], W.D[p], hfused && p == np - 1, e->gemm_chain != 0));
    } else
    for (int l = 0; l < nl; ++l) {
      const bool linear = (l == nl - 1) && ar.last_linear[p];
      if (p == 0 && l == 0 && e->emb_bias && e->emb_bias0)  // embedding columns folded into the per-element bias table
        RC(mlp_gemm(e, s, linear ? EPI_BIAS : EPI_BIAS_GELU, hin + 256, ld_in, Ls[l], true, 256, 0, N, Ls[l].k_out, Ls[l].k_in - 256,
                    e->emb_bias0, W.H[p][l], linear ? nullptr : W.D[p][l], Ls[l].k_out, in->numbers, Ls[l].k_out));
      else
        RC(mlp_gemm(e, s, linear ? EPI_BIAS : EPI_BIAS_GELU, hin, ld_in, Ls[l], true, 0, 0, N, Ls[l].k_out, Ls[l].k_in, Ls[l].b,
                    W.H[p][l], linear ? nullptr : W.D[p][l], Ls[l].k_out));
      hin = W.H[p][l];
      ld_in = Ls[l].k_out;
    }
    if (p < np - 1) {
      RC(prof_mark(e, s, FAM_POINTWISE));
      // (the feature update a^{p+1} = a^p + delta_a rides on the NSE launch: independent work, one kernel boundary less)
      RC(launch_nse_fwd(s, W.H[p][nl - 1], Ls[nl - 1].k_out, nq, p > 0 ? W.q[p - 1] : nullptr, W.nl.mol_start, in->charge,
                        n_mol, N, W.S, (float*)W.part, W.q[p], W.Fm[p], W.Dm[p], p == 0 ? e->afv : W.a[p],
                        p == 0 ? in->numbers : nullptr, W.a[p + 1], W.at[p + 1], dd));
      // domain decomposition: the final charges of halo copies are exact only within one cutoff of the owned region, the Coulomb
      // sums reach further - the owners' values come in through the exchange function
      if (dd && p == np - 2 && dd->fn(dd->ctx, AIMNET_DD_CHARGES, W.q[p], (int64_t)nq * N, (void*)s) != 0) {
        set_last_error("eval: the domain-decomposition exchange function failed (charges)");
        return AIMNET_E_INVALID;
      }
      if (p == np - 2 && overlap) {  // the final charges exist: the Coulomb block starts on the side stream
        AIMNET_HIP_CHECK(hipEventRecord(e->ev_fork, s));
        AIMNET_HIP_CHECK(hipStreamWaitEvent(e->side, e->ev_fork, 0));
        RC(coulomb_block(e->side));
        AIMNET_HIP_CHECK(hipEventRecord(e->ev_join, e->side));
      }
    }
  }
  {
    const int nlp = (int)e->mlp[np - 1].size();
    const float* hin = W.H[np - 1][nlp - 1];
    int ld_in = e->mlp[np - 1][nlp - 1].k_out;
    const int nh = (int)e->head.size();
    RC(prof_mark(e, s, FAM_GEMM));
    if (hfused) {
      HeadFusedArgs ha{};
      ha.aim3 = reinterpret_cast<const unsigned short*>(hin);
      ha.lda3 = pm * ld_in;
      ha.fmt = sfmt;
      if (sfmt == 2) { ha.w1 = e->head[0].w2a; ha.w2 = e->head[1].w2a; ha.w2t = e->head[1].wt2a; ha.w1t = e->head[0].wt2a; }
      else { ha.w1 = e->head[0].w3a; ha.w2 = e->head[1].w3a; ha.w2t = e->head[1].wt3a; ha.w1t = e->head[0].wt3a; }
      ha.b1 = e->head[0].b; ha.b2 = e->head[1].b; ha.w3 = e->head_w_last; ha.b3 = e->head_b_last;
      ha.dlast = grad ? W.D[np - 1][nlp - 1] : nullptr;
      ha.ldd = ld_in;
      ha.e_atom = W.e_atom;
      ha.zbar3 = grad ? reinterpret_cast<unsigned short*>(W.zb0) : nullptr;
      ha.ldz3 = pm * ld_in;
      ha.M = N;
      ha.grad = grad ? 1 : 0;
      RC(launch_head_fused(s, ha));
    } else {
    for (int l = 0; l + 1 < nh; ++l) {
      const Layer& L = e->head[l];
      RC(mlp_gemm(e, s, EPI_BIAS_GELU, hin, ld_in, L, true, 0, 0, N, L.k_out, L.k_in, L.b, W.hH[l], W.hD[l], L.k_out));
      hin = W.hH[l];
      ld_in = L.k_out;
    }
    RC(prof_mark(e, s, FAM_POINTWISE));
    // with gradients: the same kernel writes the backward seed d e / d z_{nh-2} = w_last * GELU'(z) into zb0
    if (overlap) {
      RC(launch_head_last(s, hin, ld_in, e->head_w_last, e->head_b_last, e->head[nh - 1].n_in, N, W.e_atom,
                          grad ? W.hD[nh - 2] : nullptr, grad ? W.zb0 : nullptr));
    } else {  // same stream: independent of the Coulomb block, so it shares that block's first launch (kernels.h, SrRiders)
      head_rider.h = hin; head_rider.ldh = ld_in; head_rider.w = e->head_w_last; head_rider.b = e->head_b_last;
      head_rider.k = e->head[nh - 1].n_in; head_rider.e_atom = W.e_atom;
      head_rider.d = grad ? W.hD[nh - 2] : nullptr; head_rider.zbar = grad ? W.zb0 : nullptr;
      head_rider.n_head_blocks = ceil_div(N, 4);
    }
    }
  }

  if (!overlap) {
    RC(prof_mark(e, s, FAM_COULOMB));
    RC(coulomb_block(s));
  }
  if (dd) {  // halo copies: no energy, no Coulomb adjoint / direct force, no backward seed (model.hip, dd_mask_kernel)
    const int nlp = (int)e->mlp[np - 1].size();
    const int seed_bytes = hfused ? pm * e->mlp[np - 1][nlp - 1].k_out * 2 : e->head[e->head.size() - 2].k_out * 4;
    RC(launch_dd_mask(s, dd->owned, in->numbers, e->sae, W.e_atom, W.ecoul, grad ? W.qbar : nullptr, nq, grad ? W.fgrad : nullptr,
                      want_s ? W.virial_atom : nullptr, grad ? W.zb0 : nullptr, seed_bytes, N));
  }
  // results of the Coulomb block (ecoul, qbar / fgrad / virial seeds, qtot) are first needed here (energy only) or in front of
  // the first conv backward (see `join` below)
  // The molecule energies are outputs only: with a stress request (and nothing else riding on the energy launch) their sums ride
  // on the two stress launches at the end of the evaluation instead of standing as two launches of their own here.
  bool energy_deferred = false, copy_deferred = false;
  auto join = [&]() -> int {
    if (overlap) AIMNET_HIP_CHECK(hipStreamWaitEvent(s, e->ev_join, 0));
    RC(prof_mark(e, s, FAM_POINTWISE));
    if (e->energy_rides && grad && want_s && pbc && charges_written && rev_done) {
      energy_deferred = true;
      return 0;
    }
    // forces only, one slice per molecule, the force-negation launch at the end (not the reverse-pair gather): the sums and the
    // copy of the charges ride there
    if (e->energy_rides && grad && want_f && !(want_s && pbc) && !W.xe && W.S == 1 && rev_done) {
      energy_deferred = true;
      copy_deferred = !charges_written;
      return 0;
    }
    RC(launch_energy_reduce(s, W.e_atom, W.ecoul, in->numbers, e->sae, W.nl.mol_start, n_mol, W.S, W.part, out->energy,
                            q_fin, charges_written ? nullptr : out->charges, N,  // + the charges output, unless the DSF walk wrote it
                            rev_done ? nullptr : &pmap,                          // + the lookup of the reverse-pair map, unless the walk ran it
                            out->status + 6));
    rev_done = true;
    return 0;
  };
  if (!grad) {
    RC(join());
    RC(prof_mark(e, s, -1));
    return AIMNET_OK;
  }
  bool joined = false;

  // ---- backward -------------------------------------------------------------------------------
  float* zcur = W.zb0;
  float* znext = W.zb1;
  if (!hfused) {  // (the fused head left the split adjoint of the last MLP's output in zb0)
    const int nh = (int)e->head.size();
    const Layer& Lp = e->head[nh - 2];  // zcur = zb0 holds the seed written by launch_head_last
    int ld = Lp.k_out;
    RC(prof_mark(e, s, FAM_GEMM));
    for (int l = nh - 2; l >= 0; --l) {
      const Layer& L = e->head[l];
      float* dprev;
      if (l > 0) dprev = W.hD[l - 1];
      else dprev = W.D[np - 1][e->mlp[np - 1].size() - 1];  // aim = GELU(z_last) of the last MLP
      RC(mlp_gemm(e, s, dprev ? EPI_MUL : EPI_NONE, zcur, ld, L, false, 0, 0, N, L.k_in, L.k_out, nullptr, znext, dprev, L.k_in));
      std::swap(zcur, znext);
      ld = L.k_in;
    }
    if (ps) {  // (interim: the head still runs on fp32 operands; its adjoint is split for the MLP backward)
      if (sfmt == 2) RC(launch_split_h2(s, zcur, ld, N, ld, reinterpret_cast<unsigned short*>(znext), 2 * ld, H2_ACT));
      else RC(launch_split_bf3(s, zcur, ld, N, ld, reinterpret_cast<unsigned short*>(znext), 3 * ld));
      std::swap(zcur, znext);
    }
  }
  for (int p = np - 1; p >= 0; --p) {
    const std::vector<Layer>& Ls = e->mlp[p];
    const int nl = (int)Ls.size();
    int ld = Ls[nl - 1].k_out;  // zcur = adjoint of the last layer's pre-activation (GELU' already applied)
    RC(prof_mark(e, s, FAM_GEMM));
    if (ps) {  // zcur holds the adjoint in split form; the sweep leaves xbar (fp32) in zcur
      RC(mlp_sweep_bwd(e, s, sfmt, p, N, p == 0 && p0m, zcur, znext, W.D[p], e->gemm_chain != 0));
      ld = Ls[0].k_in;
    } else
    for (int l = nl - 1; l >= 0; --l) {
      const Layer& L = Ls[l];
      if (l > 0)
        RC(mlp_gemm(e, s, EPI_MUL, zcur, ld, L, false, 0, 0, N, L.k_in, L.k_out, nullptr, znext, W.D[p][l - 1], L.k_in));
      else if (p == 0 && p0m)  // only the conv columns 256.. of xbar_0 are consumed (the embedding is a constant)
        RC(mlp_gemm(e, s, EPI_NONE, zcur, ld, L, false, 0, 256, N, L.k_in - 256, L.k_out, nullptr, znext + 256, nullptr, L.k_in));
      else
        RC(mlp_gemm(e, s, EPI_NONE, zcur, ld, L, false, 0, 0, N, L.k_in, L.k_out, nullptr, znext, nullptr, L.k_in));
      std::swap(zcur, znext);
      ld = L.k_in;
    }
    // zcur = xbar_p  (N x k_in of the first layer)
    if (!joined) {  // the conv backward below is the first consumer of the Coulomb block's qbar / dE/dx / virial seeds
      RC(join());
      joined = true;
    }
    RC(prof_mark(e, s, FAM_UNCONCAT));
    if (p == 0 && p0m) {
      RC(launch_unconcat_p0(s, zcur, ld, W.V[0], e->agh_a, e->afv, e->z_of_slot, e->nslots, W.present_part, W.n_part, W.Sbar, N));
      RC(prof_mark(e, s, FAM_CONV_BWD));
      RC(launch_conv_bwd_p0(s, want_s, W.Sbar, e->nslots, W.aslot, W.nb_idx, W.nb_cnt, W.pg, cap, e->bp, W.fgrad, W.virial_atom, N,
                            order, (W.xe && want_f) ? W.pairbuf : nullptr));
      break;
    }
    RC((mfma_bwd ? launch_unconcat_t : launch_unconcat)(s, p > 0 ? nq : 0, zcur, ld, W.V[p], W.Vq[p], e->agh_a, e->agh_q, W.Sbar,
                                                         W.Sqbar, N));
    RC(prof_mark(e, s, FAM_CONV_BWD));
    if (mfma_bwd) {
      RC(launch_conv_bwd_mfma(s, p > 0 ? nq : 0, p > 0, want_s, p == 0 ? e->afv_t : W.at[p], p == 0 ? in->numbers : nullptr,
                              p > 0 ? W.q[p - 1] : nullptr, W.Sbar, W.Sqbar, W.nb_idx, W.nb_cnt, W.pg, cap, e->bp, zcur, ld,
                              (p < np - 1) ? W.abar : nullptr, W.abar, W.qbar, W.qbar, W.fgrad, W.virial_atom, N, order));
    } else {
      RC(launch_conv_bwd(s, p > 0 ? nq : 0, p > 0, want_s, p == 0 ? e->afv : W.a[p], p == 0 ? in->numbers : nullptr,
                         p > 0 ? W.q[p - 1] : nullptr, W.Sbar, W.Sqbar, W.nb_idx, W.nb_cnt, W.pg, cap, e->bp, zcur, ld,
                         (p < np - 1) ? W.abar : nullptr, W.abar, W.qbar, W.qbar, W.fgrad, W.virial_atom, N, order,
                         (W.xe && p > 0) ? W.pairbuf : nullptr, p < np - 1, e->split_max));
    }
    if (p == 0) break;
    // NSE adjoint of pass p-1, then the adjoint of its MLP output
    const std::vector<Layer>& Lq = e->mlp[p - 1];
    const int nlq = (int)Lq.size();
    const float* y = W.H[p - 1][nlq - 1];
    const int ldy = Lq[nlq - 1].k_out;
    RC(prof_mark(e, s, FAM_POINTWISE));
    if (dd) {  // domain decomposition: the adjoint sums run over every local atom and are all-reduced over the ranks
      RC(launch_nse_bwd_reduce(s, W.qbar, y, ldy, nq, W.nl.mol_start, n_mol, N, 1, (float*)W.part));
      if (dd->fn(dd->ctx, AIMNET_DD_SUM, W.part, (int64_t)nq * n_mol, (void*)s) != 0) {
        set_last_error("eval: the domain-decomposition exchange function failed (NSE adjoint sums)");
        return AIMNET_E_INVALID;
      }
      RC(launch_build_zbar(s, W.qbar, W.abar, y, ldy, ar.last_linear[p - 1] ? nullptr : W.D[p - 1][nlq - 1], W.Fm[p - 1],
                           W.Dm[p - 1], (const float*)W.part, 1, mol_c, N, n_mol, 256, nq, p - 1 > 0, znext, W.qbar, sfmt, nullptr,
                           dd->owned));
    } else if (e->nse_merged && N <= 1024) {  // small systems: the molecule sums inside build_zbar, one launch instead of two
      RC(launch_build_zbar(s, W.qbar, W.abar, y, ldy, ar.last_linear[p - 1] ? nullptr : W.D[p - 1][nlq - 1], W.Fm[p - 1],
                           W.Dm[p - 1], nullptr, 1, mol_c, N, n_mol, 256, nq, p - 1 > 0, znext, W.qbar2, sfmt, W.nl.mol_start));
      std::swap(W.qbar, W.qbar2);
    } else {
      RC(launch_nse_bwd_reduce(s, W.qbar, y, ldy, nq, W.nl.mol_start, n_mol, N, W.S, (float*)W.part));
      RC(launch_build_zbar(s, W.qbar, W.abar, y, ldy, ar.last_linear[p - 1] ? nullptr : W.D[p - 1][nlq - 1], W.Fm[p - 1],
                           W.Dm[p - 1], (const float*)W.part, W.S, mol_c, N, n_mol, 256, nq, p - 1 > 0, znext, W.qbar, sfmt));
    }
    std::swap(zcur, znext);
  }
  RC(prof_mark(e, s, FAM_POINTWISE));
  // reverse-pair form: the pair buffer holds F1 of both passes; its gather is the last contribution to dE/dx and writes the forces
  // (with a stress request the force gather rides on the launch of the virial sums: independent work, one kernel boundary less)
  const bool pf_rides = W.xe && want_f && want_s && pbc;
  const PairForceRider pfr{W.nb_idx, W.nb_cnt, W.rev, W.pairbuf, cap, out->forces, ceil_div(N, 4)};
  const EnergyRider erd{W.e_atom, W.ecoul, in->numbers, e->sae, W.part_e, out->energy, n_mol,
                        q_fin, copy_deferred ? out->charges : nullptr, N};
  if (W.xe && want_f && !pf_rides) RC(launch_pair_force(s, W.nb_idx, W.nb_cnt, W.rev, W.pairbuf, cap, N, W.fgrad, out->forces, out->status + 6));
  // (domain decomposition: no cell - the virial sums are divided by the volume of a unit cube, i.e. the `stress` output takes the
  // rank's share of dE/d(strain) itself; the caller adds the ranks' shares and divides by the cell volume)
  RC(launch_finalize(s, W.fgrad, W.virial_atom, W.nl.mol_start, dd ? e->unit_cell : in->cell, dd ? 1 : n_cell, n_mol, N, W.S, W.part,
                     (want_f && !W.xe) ? out->forces : nullptr, want_s ? out->stress : nullptr, pf_rides ? &pfr : nullptr,
                     energy_deferred ? &erd : nullptr, e->sums_whole != 0, out->status + 6));
  RC(prof_mark(e, s, -1));
  return AIMNET_OK;
}

// ---- stand-alone entry points --------------------------------------------------------------------
size_t aimnet_neighbor_list_workspace_bytes(int32_t n_atoms, int32_t n_mol, int32_t max_nb) {
  if (n_atoms <= 0 || n_mol <= 0 || max_nb <= 0) return 0;
  return align_up(nlist_scratch_bytes(n_atoms, n_mol), 256) + align_up((size_t)n_atoms * (size_t)max_nb * sizeof(int), 256);
}

__global__ void expand_shifts_kernel(const int* __restrict__ code, size_t n, int* __restrict__ out) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  int sx, sy, sz;
  unpack_shift(code[e], sx, sy, sz);
  out[3 * e] = sx;
  out[3 * e + 1] = sy;
  out[3 * e + 2] = sz;
}

int aimnet_neighbor_list(const float* coord, const int32_t* mol_idx, int32_t n_atoms, int32_t n_mol,
                         const float* cell, int32_t n_cell, const int32_t pbc[3], float cutoff, int32_t max_nb,
                         int32_t fill_value, int32_t* nbmat, int32_t* shifts, int32_t* num_nb, int32_t* status,
                         float* coord_wrapped, void* workspace, size_t workspace_bytes, void* hip_stream) {
  if (!coord || !mol_idx || !nbmat || !num_nb || !status || !workspace || n_atoms <= 0 || n_mol <= 0 || max_nb <= 0)
    return AIMNET_E_INVALID;
  if (cell && !shifts) {
    set_last_error("neighbor_list: periodic input needs a shifts buffer");
    return AIMNET_E_INVALID;
  }
  if (cell && !(n_cell == 1 || n_cell == n_mol)) {
    set_last_error("neighbor_list: n_cell must be 1 or n_mol");
    return AIMNET_E_INVALID;
  }
  const size_t need = aimnet_neighbor_list_workspace_bytes(n_atoms, n_mol, max_nb);
  if (workspace_bytes < need) {
    set_last_error("neighbor_list: workspace too small (%zu < %zu)", workspace_bytes, need);
    return AIMNET_E_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)hip_stream;
  NlistBuffers nl;
  nlist_carve(nl, (char*)workspace, n_atoms, n_mol);
  int* codes = (int*)((char*)workspace + align_up(nlist_scratch_bytes(n_atoms, n_mol), 256));
  const int pz[3] = {1, 1, 1};
  const int* pb = pbc ? pbc : pz;
  AIMNET_HIP_CHECK(hipMemsetAsync(status, 0, 2 * sizeof(int), s));
  RC(launch_mol_start(s, mol_idx, n_atoms, n_mol, nl.mol_start, nl.mol_c));
  mol_idx = nl.mol_c;  // clamped to [0, n_mol)
  RC(launch_wrap(s, coord, mol_idx, n_atoms, n_mol, cell, cell ? n_cell : 0, pb, nl));
  if (!cell && (long)n_atoms >= 1500L * n_mol) RC(launch_bbox(s, n_mol, nl));
  RC(launch_nlist(s, n_atoms, n_mol, mol_idx, cell, cell ? n_cell : 0, pb, cutoff, cutoff, max_nb, fill_value, 1, nl, nbmat,
                  codes, num_nb, status + 0, status + 1));
  if (cell) {
    const size_t n_pairs = (size_t)n_atoms * max_nb;
    hipLaunchKernelGGL(expand_shifts_kernel, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, s, codes, n_pairs, shifts);
    AIMNET_LAUNCH_CHECK();
  }
  if (coord_wrapped)
    AIMNET_HIP_CHECK(hipMemcpyAsync(coord_wrapped, nl.xw, (size_t)n_atoms * 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
  return AIMNET_OK;
}

int aimnet_debug_gemm(int cfg, int epi, const float* A, int lda, const float* Bt, int ldb, int M, int N, int K,
                      const float* bias, float* C, float* D, int ldc, void* hip_stream) {
  static bool attr = false;
  if (!attr) {
    int rc = gemm_set_attributes();
    if (rc) return rc;
    attr = true;
  }
  return launch_gemm_nt_cfg((hipStream_t)hip_stream, cfg, epi, A, lda, Bt, ldb, M, N, K, bias, C, D, ldc);
}

int aimnet_debug_split_bf3(const float* src, int ld, int M, int K, void* dst, int ldd, int neg_from_block, void* hip_stream) {
  if (!src || !dst || M <= 0 || K <= 0 || ldd < 3 * pad32(K) || ldd % 96 || (neg_from_block < 0 && neg_from_block != aimnet::BF3_ALT))
    return AIMNET_E_INVALID;
  return launch_split_bf3((hipStream_t)hip_stream, src, ld, M, K, (unsigned short*)dst, ldd, neg_from_block);
}

int aimnet_debug_gemm_bf3(int cfg, int epi, const float* A, int lda, const void* Bt3, int ldb, int M, int N, int K,
                          const float* bias, float* C, float* D, int ldc, int kneg, void* hip_stream) {
  static bool attr = false;
  if (!attr) {
    int rc = gemm_bf3_set_attributes();
    if (rc) return rc;
    attr = true;
  }
  return launch_gemm_bf3_cfg((hipStream_t)hip_stream, cfg, epi, A, lda, (const unsigned short*)Bt3, ldb, M, N, K, bias, C, D, ldc,
                             nullptr, 0, kneg < 0 ? BF3_NO_NEG : kneg);
}

int aimnet_debug_gemm_bf3a(int cfg, int epi, int out3, const void* A3, int lda3, const void* Bt3, int ldb, int M, int N, int K,
                           const float* bias, float* C, void* C3, int ldc3, float* D, int ldc, int alt, void* hip_stream) {
  using namespace aimnet;
  if (alt < 0 || alt > 2) return AIMNET_E_INVALID;
  static bool once = false;
  if (!once) {
    int rc = gemm_bf3a_set_attributes();
    if (rc) return rc;
    once = true;
  }
  return launch_gemm_bf3a_cfg((hipStream_t)hip_stream, cfg, epi, out3 != 0, (const unsigned short*)A3, lda3, (const unsigned short*)Bt3,
                              ldb, M, N, K, bias, C, (unsigned short*)C3, ldc3, D, ldc, nullptr, 0, alt);
}
int aimnet_debug_split_h2(const float* src, int ld, int M, int K, void* dst, int ldd, int mode, void* hip_stream) {
  if (!src || !dst || M <= 0 || K <= 0 || ldd < 2 * pad32(K) || ldd % 64 || mode < 0 || mode > 2) return AIMNET_E_INVALID;
  return aimnet::launch_split_h2((hipStream_t)hip_stream, src, ld, M, K, (unsigned short*)dst, ldd, mode);
}

int aimnet_debug_gemm_h2(int cfg, int epi, int out2, const void* A2, int lda2, const void* Bt2, int ldb, int M, int N, int K,
                         const float* bias, float* C, void* C2, int ldc2, float* D, int ldc, int alt, void* hip_stream) {
  using namespace aimnet;
  if (alt < 0 || alt > 2) return AIMNET_E_INVALID;
  static bool once = false;
  if (!once) {
    int rc = gemm_h2_set_attributes();
    if (rc) return rc;
    once = true;
  }
  return launch_gemm_h2_cfg((hipStream_t)hip_stream, cfg, epi, out2 != 0, (const unsigned short*)A2, lda2, (const unsigned short*)Bt2,
                            ldb, M, N, K, bias, C, (unsigned short*)C2, ldc2, D, ldc, nullptr, 0, alt);
}
int aimnet_engine_debug_mlp_sweep(aimnet_engine* e, int pass, int backward, int chain, int flag, const void* x2, int M,
                                  const int32_t* numbers, float* const* H, float* const* D, float* const* zb, int* which,
                                  void* hip_stream) {
  using namespace aimnet;
  if (!e || pass < 0 || pass >= e->arch.n_pass || M <= 0 || !D) return AIMNET_E_INVALID;
  if (!(e->gemm_h2 && e->h2_fits)) {
    set_last_error("debug_mlp_sweep: the fp16x2 operand form is off");
    return AIMNET_E_INVALID;
  }
  hipStream_t s = (hipStream_t)hip_stream;
  if (!backward) {
    if (!x2 || !H) return AIMNET_E_INVALID;
    return mlp_sweep_fwd(e, s, 2, pass, M, numbers, reinterpret_cast<const float*>(x2), H, D, flag != 0, chain != 0);
  }
  if (!zb || !zb[0] || !zb[1] || !which) return AIMNET_E_INVALID;
  float *zcur = zb[0], *znext = zb[1];
  const int rc = mlp_sweep_bwd(e, s, 2, pass, M, flag != 0, zcur, znext, D, chain != 0);
  *which = zcur == zb[0] ? 0 : 1;
  return rc;
}

#ifdef AIMNET_PREP_TIMING
int aimnet_debug_prep_stamps(unsigned long long* host16) { return aimnet::prep_read_stamps(host16); }
#endif
#ifdef AIMNET_BF3_TIMING
int aimnet_debug_bf3a_stamps(unsigned long long* host1024) { return aimnet::gemm_bf3a_read_stamps(host1024); }
int aimnet_debug_h2_stamps(unsigned long long* host1024) { return aimnet::gemm_h2_read_stamps(host1024); }
int aimnet_debug_bf3_stamps(unsigned long long* host1024) { return aimnet::gemm_bf3_read_stamps(host1024); }
#endif

int aimnet_debug_pme_recip(const float* xw, const float* q, const int* order, const float* cell, float total_charge, int n_atoms,
                           float accuracy,
                           int max_mesh, double* e_atom, float* qbar, float* fgrad, float* virial_atom, double* host_info,
                           void* hip_stream) {
  using namespace aimnet;
  if (!xw || !q || !cell || !e_atom || !qbar || !fgrad || !virial_atom || !host_info || n_atoms <= 0 || max_mesh < 512)
    return AIMNET_E_INVALID;
  hipStream_t st = (hipStream_t)hip_stream;
  EwaldBuffers b{};
  b.max_mesh = max_mesh;
  b.max_parts = ceil_div(max_mesh, PME_PART);
  int *mol_idx = nullptr, *mol_start = nullptr, *status = nullptr;
  float* charge = nullptr;
  const int ms[2] = {0, n_atoms};
  int rc = 0;
#define PME_DBG(x)            \
  if ((x) != hipSuccess) {    \
    rc = AIMNET_E_HIP;        \
    goto done;                \
  }
  PME_DBG(hipMalloc(&b.sys, sizeof(EwaldSystem)));
  PME_DBG(hipMalloc(&b.meshq, sizeof(long long) * (size_t)max_mesh));
  PME_DBG(hipMalloc(&b.ma, sizeof(double) * 2 * (size_t)max_mesh));
  PME_DBG(hipMalloc(&b.mb, sizeof(double) * 2 * (size_t)max_mesh));
  PME_DBG(hipMalloc(&b.bmod, sizeof(double) * 3 * PME_MAX_AXIS));
  PME_DBG(hipMalloc(&b.vpart, sizeof(double) * 8 * (size_t)b.max_parts));
  PME_DBG(hipMalloc(&mol_idx, sizeof(int) * n_atoms));
  PME_DBG(hipMalloc(&mol_start, sizeof(int) * 2));
  PME_DBG(hipMalloc(&status, sizeof(int)));
  PME_DBG(hipMalloc(&charge, sizeof(float)));
  PME_DBG(hipMemsetAsync(mol_idx, 0, sizeof(int) * n_atoms, st));
  PME_DBG(hipMemcpyAsync(mol_start, ms, sizeof(ms), hipMemcpyHostToDevice, st));
  PME_DBG(hipMemcpyAsync(charge, &total_charge, sizeof(float), hipMemcpyHostToDevice, st));
  rc = launch_pme_setup(st, cell, 1, mol_start, charge, 1, 1, accuracy, b, status);
  if (!rc) rc = launch_pme_recip(st, true, true, xw, q, mol_idx, mol_start, order, n_atoms, 1, b, 1.0f, e_atom, qbar, fgrad, virial_atom);
  if (!rc) {
    EwaldSystem E;
    int need = 0;
    PME_DBG(hipMemcpyAsync(&E, b.sys, sizeof(E), hipMemcpyDeviceToHost, st));
    PME_DBG(hipMemcpyAsync(&need, status, sizeof(int), hipMemcpyDeviceToHost, st));
    PME_DBG(hipStreamSynchronize(st));
    host_info[0] = E.alpha; host_info[1] = E.rc; host_info[2] = E.mesh[0]; host_info[3] = E.mesh[1]; host_info[4] = E.mesh[2];
    host_info[5] = need; host_info[6] = E.phi_bg; host_info[7] = 0.0;
  }
done:
#undef PME_DBG
  (void)hipStreamSynchronize(st);
  (void)hipFree(b.sys); (void)hipFree(b.frac); (void)hipFree(b.meshq); (void)hipFree(b.ma); (void)hipFree(b.mb); (void)hipFree(b.bmod);
  (void)hipFree(b.vpart); (void)hipFree(mol_idx); (void)hipFree(mol_start); (void)hipFree(status); (void)hipFree(charge);
  return rc;
}

int aimnet_debug_mfma4_probe(float* out, void* hip_stream) {
  if (!out) return AIMNET_E_INVALID;
  return launch_mfma4_probe((hipStream_t)hip_stream, out);
}

int aimnet_conv_sv_2d_sp_fwd(const float* a, const int32_t* idx, const float* g, float* out, int32_t B, int32_t A,
                             int32_t G, int32_t M, void* hip_stream) {
  if (!a || !idx || !g || !out || B < 0 || A <= 0 || G <= 0 || M <= 0) return AIMNET_E_INVALID;
  return launch_conv_sv_fwd((hipStream_t)hip_stream, a, idx, g, out, B, A, G, M);
}

int aimnet_conv_sv_2d_sp_bwd(const float* grad_out, const float* a, const int32_t* idx, const float* g,
                             float* grad_a, float* grad_g, int32_t B, int32_t A, int32_t G, int32_t M,
                             void* hip_stream) {
  if (!grad_out || !a || !idx || !g || !grad_a || !grad_g || B < 0 || A <= 0 || G <= 0 || M <= 0) return AIMNET_E_INVALID;
  return launch_conv_sv_bwd((hipStream_t)hip_stream, grad_out, a, idx, g, grad_a, grad_g, B, A, G, M);
}

int aimnet_conv_sv_2d_sp_bwd_bwd(const float* grad_out, const float* grad2_a, const float* grad2_g, const float* a,
                                 const int32_t* idx, const float* g, float* grad_grad_out, float* grad_a_double,
                                 float* grad_g_double, int32_t B, int32_t A, int32_t G, int32_t M, void* hip_stream) {
  if (!grad_out || !grad2_a || !grad2_g || !a || !idx || !g || !grad_grad_out || !grad_a_double || !grad_g_double || B < 0 ||
      A <= 0 || G <= 0 || M <= 0)
    return AIMNET_E_INVALID;
  return launch_conv_sv_bwd_bwd((hipStream_t)hip_stream, grad_out, grad2_a, grad2_g, a, idx, g, grad_grad_out, grad_a_double,
                                grad_g_double, B, A, G, M);
}

}  // extern "C"
