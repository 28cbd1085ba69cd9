// engine.h - the engine object behind the opaque `aimnet_engine` handle of include/aimnet_hip.h: device weight store, switches,
// profiling state.  Internal to the library (engine.hip builds it, hvp.hip reads the weights and the architecture).
#pragma once

#include <map>
#include <string>
#include <vector>

#include "../../include/aimnet_hip.h"
#include "common.h"
#include "kernels.h"

namespace aimnet {

static inline int pad32(int n) { return (n + 31) / 32 * 32; }

struct Layer {
  int n_in, n_out, k_in, k_out;  // real and padded (multiple of 32) sizes
  float* w;                      // [k_out][k_in]  forward operand  (Bt of  H = X . W^T)
  float* wt;                     // [k_in][k_out]  backward operand (Bt of dX = dZ . W)
  float* b;                      // [k_out]
  unsigned short* w3 = nullptr;  // the same two operands split into three bf16 planes ("bf3" layout of gemm_bf3.hip:
  unsigned short* wt3 = nullptr; // 3 * k_in resp. 3 * k_out bf16 elements per row) for the bf16x3-split MFMA GEMM
  int neg_w3 = 0, neg_wt3 = 0;   // k-block (of 32) from which w3 / wt3 are stored negated (sign-flipped accumulation phase)
  unsigned short* w3a = nullptr;   // the same two operands with every ODD k-block negated (BF3_ALT): gemm_bf3a.hip / gemm_head.hip,
  unsigned short* wt3a = nullptr;  // which accumulate even / odd k-steps separately and subtract (no data-dependent split point)
  unsigned short* w2a = nullptr;   // the same two operands in the fp16x2-split form of gemm_h2.hip ("h2", H2_WEIGHT: hi planes of the
  unsigned short* wt2a = nullptr;  // odd k-blocks negated; 2 * k_in resp. 2 * k_out 16-bit elements per row); NULL: out of fp16's range
  std::vector<unsigned short> h_w2a, h_wt2a;  // host copies of the two, kept until the chain plans are packed (engine create)
};

// One MLP sweep as a single launch of gemm_chain.hip: the instantiated shape and, per pass, the packed weight stream
struct ChainPlan {
  int shape = -1;  // -1: this sweep takes the per-layer launches
  int n_pass = 0;
  struct Pass {
    void* w[2];  // packed streams of the column groups A / B (device; w[0] NULL: no group A)
    int layer;   // MLP layer this pass belongs to
    int n0;      // first output column of the pass inside the layer's output
    int ncols;   // output columns of the pass
  } pass[CHAIN_MAX_PASS];
};

struct View {
  size_t off, n_elem;
  int elem_size, row_stride;
};

}  // namespace aimnet

using aimnet::BasisParams;
using aimnet::ChainPlan;
using aimnet::D3Tables;
using aimnet::Layer;
using aimnet::View;

struct aimnet_engine {
  aimnet_arch arch;
  int nq = 1;  // charge channels (arch.n_charge_channels, 0 -> 1)
  int device;
  std::vector<void*> allocs;
  float *afv, *afv_t, *agh_a, *agh_q;  // afv_t: the embedding rows in the operand layout of the MFMA conv kernels
  // Pass 0, first MLP layer: the first 256 input columns are the constant embedding row afv[Z_i], so their product with the
  // weights is one of 64 constant vectors: emb_bias0[z] = b + W[:, :256] . afv[z] (fp64 sums at create time).  The GEMM then
  // runs over the 448 conv columns only (K 704 -> 448) with this table as a row-indexed bias.  set_option("emb_bias", 0) keeps
  // the full-width GEMM (A/B and parity runs).
  float* emb_bias0 = nullptr;
  bool emb_bias = true;
  // MLP GEMMs: 1 (default) = bf16x3-split operands on the bf16 matrix pipe (gemm_bf3.hip: fp32 == three bf16 planes exactly, six
  // products per tile, fp32 accumulation - the fp32 result to within the fp32 rounding of the accumulation itself) for batches above
  // 256 rows, the exact-fp32 skinny kernel below; 2 = bf3 for every batch size (parity runs on small fixtures); 0 = the exact-fp32
  // MFMA kernels of gemm.hip everywhere.  set_option("gemm_bf3", v) / AIMNET_GEMM_BF3.
  int gemm_bf3 = 1;
  // AIMNET_GEMM_PRESPLIT / set_option("gemm_presplit"): with the bf16x3-split GEMMs (batches above 256 rows; every batch in mode 2) every
  // GEMM activation operand (MLP input rows, hidden activations, backward adjoints) stays in the split "bf3" form in memory: the producer
  // splits once (conv_fwd's row assembly, the GELU / chain-rule epilogues) and gemm_bf3a.hip streams both operands by DMA with no
  // vector work in its main loop.  0 = fp32 activations, split inside gemm_bf3.hip's loop (round 3; also what the tangent sweep of
  // hvp.hip, small batches and AIMNET_KEEP_INTERMEDIATES use).
  int gemm_presplit = 1;
  // AIMNET_GEMM_H2 / set_option("gemm_h2"): wherever the activations are pre-split, they and the weights take the fp16x2-split form
  // (gemm_h2_common.h: fp32 == hi + lo / 4096 to 2^-24, three matrix instructions per tile and k-step, 4 bytes per element) instead
  // of the bf16x3 form (six instructions, 6 bytes).  0 = bf16x3 (gemm_bf3a.hip).  Needs every weight inside fp16's range
  // (h2_fits; activations beyond it surface as non-finite outputs and the Python layer repeats the call with gemm_h2 = 0).
  int gemm_h2 = 1;
  bool h2_fits = true;
  // AIMNET_HEAD_FUSED / set_option("head_fused"): with pre-split activations, the energy head 256 -> 128 -> 128 -> 1 runs forward
  // and backward in ONE launch (gemm_head.hip) instead of four N = 128 GEMM launches and the last-layer rider
  int head_fused = 1;
  // AIMNET_PREP_FUSED / set_option("prep_fused"): periodic batches of up to 4 096 atoms / 64 systems prepare their cell grid in one
  // single-block launch (nlist.hip, prep_small_kernel) instead of seven small dependent ones.  0 = the separate kernels (any size).
  int prep_fused = 1;
  // AIMNET_ENERGY_RIDES / set_option("energy_rides"): periodic evaluations with a stress request sum the molecule energies on the two
  // stress launches at the end (model.hip, EnergyRider) instead of two launches of their own in front of the backward pass
  int energy_rides = 1;
  // AIMNET_STATUS_RIDES / set_option("status_rides"): when the short-range list is the only list built, its status words (longest
  // row, overflow flag) are reduced by rider blocks of the SR-Coulomb launch instead of a launch of their own
  int status_rides = 1;
  // AIMNET_SETUP_RIDES / set_option("setup_rides"): periodic batches - the cell + bin-grid setup block rides on the molecule-offset
  // launch (atom counts by binary search in mol_idx) instead of following it as a launch of its own
  int setup_rides = 1;
  // AIMNET_STATUS_OWNED / set_option("status_owned"): evaluations with one list of up to 32 768 atoms do not zero the status array in
  // front: the status rider stores all eight words (cellwalk.h, nlist_status_owned_block) - the memset launch goes
  int status_owned = 1;
  // AIMNET_SUMS_WHOLE / set_option("sums_whole"): up to 16 384 atoms, with the energy sums riding on the stress launch and the force
  // gather beside them: one block per cell / per molecule sums everything (no slices, no finish launch)
  int sums_whole = 1;
  // AIMNET_NSE_MERGED / set_option("nse_merged"): systems of up to 1 024 atoms form the molecule sums of the NSE adjoint inside
  // build_zbar_kernel instead of by a partial-sum launch in front of it (two launches fewer per evaluation)
  int nse_merged = 1;
  // AIMNET_GEMM_CHAIN / set_option("gemm_chain"): with fp16x2-split activations every MLP sweep (forward or backward, one per pass)
  // is ONE launch of gemm_chain.hip - a block owns 48 rows and the full width of every layer, the hidden activations stay in LDS -
  // instead of one GEMM launch per layer (bitwise-equal results).  0 = the per-layer launches.  chain_fwd[p][e]: e = 1 with the
  // embedding block of pass 0 folded into the bias table; chain_bwd[p][m]: m = 1 when only the conv columns of xbar_0 are formed.
  int gemm_chain = 1;
  ChainPlan chain_fwd[AIMNET_MAX_PASS][2], chain_bwd[AIMNET_MAX_PASS][2];
  // AIMNET_D3_CN_RIDES / set_option("d3_cn_rides"): the DFT-D3 coordination numbers are formed by the cell-grid list build that serves
  // D3 (kernels.h, D3CnRider) instead of by a pass over the finished matrix
  int d3_cn_rides = 1;
  // AIMNET_DSF_NP_WALK / set_option("dsf_np_walk"): non-periodic systems large enough for the bounding-box cell grid (>= 1 500 atoms per
  // molecule on average) evaluate DSF Coulomb by the list-free walk over that grid, like periodic cells do - no 15 A neighbour matrix
  // (N x ~2 000 entries built and read every step).  0 = the matrix form.  (With DFT-D3 at the same cutoff the pair terms ride on
  // the D3 matrix pass either way.)  The clusters of a domain-decomposed evaluation are such systems.
  int dsf_np_walk = 1;
  // aimnet_engine_set_dd: spatial domain decomposition of one system over ranks - owned-atom mask of the local cluster and the
  // caller's exchange function (dd.owned == NULL: off)
  aimnet::DdLink dd{nullptr, nullptr, nullptr};
  float* unit_cell = nullptr;  // 3 x 3 identity (device): "cell" of the virial sums of a decomposed evaluation
  double* sae;
  // species slots of the pass-0 moment backward: slot = rank of the atomic number among the embedding rows that
  // are finite (supported elements); every other Z shares one extra slot that points at its NaN row
  int *slot_of_z, *z_of_slot;
  int nslots = 0;
  std::vector<int> z_of_slot_h;  // host copy (slot -> atomic number)
  // DFT-D3 tables re-indexed by species slot (aimnet_engine_set_dftd3); d3.ns == 0 until set
  D3Tables d3{0, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool spatial_order = true;  // AIMNET_SPATIAL_ORDER=0: conv kernels walk the atoms in input order
  // AIMNET_KEEP_INTERMEDIATES=1: every MLP input row x[p] and hidden activation h[p][l] gets its own buffer (debug views of
  // all of them stay valid after an evaluation).  Default: they share one / two buffers - each is dead as soon as the next
  // GEMM has consumed it (the backward reads GELU', not the activations), and a buffer that is rewritten while its lines
  // are still in the Infinity Cache never costs HBM write bandwidth
  bool keep_intermediates = false;
  bool p0_moments = true;  // AIMNET_P0_MOMENTS=0 keeps the generic conv_fwd / conv_bwd for pass 0 (A/B and parity runs)
  // AIMNET_CONV_MFMA / set_option("conv_mfma"): bit 0 = conv_fwd, bit 1 = conv_bwd (+ unconcat T layout) on the 4x4x1 MFMA
  // kernels of conv_mfma.hip for systems above the split threshold; 0 (default) = the packed-FMA VALU kernels of conv.hip.
  // Measured on config 3 (profiles/r2_conv_mfma.md): forward 81 us either way, backward 293 vs 226 us - neither form is
  // arithmetic-bound, so the matrix pipe buys nothing here and the VALU kernels stay the default.
  int conv_mfma = 0;
  // AIMNET_CONV_XE / set_option("conv_xe"): the reverse-pair form of the conv backward (conv.hip, conv_bwd_kernel<.., XE>) for
  // passes >= 1 of systems above the split threshold: every ordered pair evaluates only its own half of the pair adjoints (no
  // a_j gather, 4 KiB per pair instead of 5.25 KiB), F1 goes through a pair buffer and a reverse-pair map (per-atom hash tables of the rows).
  // Config 3: kernel 215 -> 152 us per pass, +35 us per step for the map (hash build 9 us, lookup 12 us) and the force gather (14 us).
  // 0 restores the combined-adjoint kernel (A/B and parity runs).
  int conv_xe = 1;
  // atoms up to which the 4-waves-per-atom "split" conv kernels are used (AIMNET_SPLIT_MAX / set_option("split_max")); per engine
  int split_max = aimnet::conv_split_max_default();
  // AIMNET_OVERLAP_COULOMB / set_option("overlap_coulomb"): the Coulomb / DFT-D3 pair kernels (VALU-bound, they need only the
  // final charges) run on a second HIP stream next to the last pass' MLP, the energy head and the first backward GEMMs
  // (MFMA-bound): forked after the last charge update, joined in front of the first conv backward
  bool overlap_coulomb = false;  // measured (profiles/r2_summary.md): 2.135 vs 2.118 ms/step - concurrent kernels of one process slow each other down here too
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  std::vector<Layer> mlp[AIMNET_MAX_PASS];
  std::vector<Layer> head;
  float* head_w_last;  // [k] last head layer as a vector
  float* head_b_last;  // [1]
  BasisParams bp;
  std::map<std::string, View> views;
  // optional HIP-event profiling: one event per change of kernel family on the eval stream
  int prof_level = 0;  // 0 off, 1 GEMM vs everything else, 2 every family
  int prof_every = 1;  // events are recorded on every prof_every-th evaluation only (they cost ~3 % of a 2 ms step)
  long prof_evals = 0, prof_sampled = 0;
  bool prof_on = false;  // this evaluation records events
  std::vector<hipEvent_t> prof_ev;
  std::vector<int> prof_fam;
  size_t prof_used = 0;
  int prof_last = -2;
};

namespace aimnet {
// One MLP GEMM C = epilogue(A . W^T) (fwd) or C = epilogue(A . W) (bwd) over a sub-block of the layer, on the kernel family the
// engine's `gemm_bf3` switch selects (engine.hip)
int mlp_gemm(const aimnet_engine* e, hipStream_t s, int epi, const float* A, int lda, const Layer& L, bool fwd, int k0, int n0, int M,
             int N, int K, const float* bias, float* C, float* D, int ldc, const int* brow = nullptr, int ldbias = 0);
// the same with the activation operand pre-split (A3, lda3 = 3 x its padded width in bf16 elements; gemm_bf3a.hip); out3: C is
// written in bf3 form into C3 (ldc3 bf16 elements per row) instead of fp32 into C; D is fp32 [M][ldc]
// fmt: 1 = bf16x3 planes (gemm_bf3a.hip, lda3 / ldc3 = 3 x the padded width), 2 = fp16x2 planes (gemm_h2.hip, 2 x)
int mlp_gemm3(const aimnet_engine* e, hipStream_t s, int fmt, int epi, bool out3, const unsigned short* A3, int lda3, const Layer& L,
              bool fwd, int k0, int n0, int M, int N, int K, const float* bias, float* C, unsigned short* C3, int ldc3, float* D, int ldc,
              const int* brow = nullptr, int ldbias = 0);
int split_format(const aimnet_engine* e, int n_rows);  // 0: fp32 activations, 1: bf3, 2: h2 (engine.hip)
}  // namespace aimnet
