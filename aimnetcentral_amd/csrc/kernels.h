// kernels.h - host-side launchers of the gfx950 kernels (one .hip file per family).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <vector>

namespace aimnet {

struct DdLink;  // domain decomposition of one system over ranks (below)

// ---- gemm.hip ---------------------------------------------------------------------------------
enum { EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_GELU = 2, EPI_MUL = 3 };
// C[M,N] = A[M,K] . Bt[N,K]^T with fused epilogue.  K % 32 == 0, lda/ldb % 4 == 0.
//   EPI_BIAS_GELU: C = gelu(acc + bias), D = gelu'(acc + bias) (D may be NULL)
//   EPI_MUL:       C = acc * D
int device_cus();  // compute units of the current device (256 on MI355X), queried once per device (gemm.hip)
// brow / ldbias (optional): the bias of output row m is the row brow[m] (clamped to 0..63) of a [64][ldbias] table
int launch_gemm_nt(hipStream_t stream, int epi, const float* A, int lda, const float* Bt, int ldb, int M, int N,
                   int K, const float* bias, float* C, float* D, int ldc, const int* brow = nullptr, int ldbias = 0);
int launch_gemm_nt_cfg(hipStream_t stream, int cfg, int epi, const float* A, int lda, const float* Bt, int ldb, int M, int N,
                       int K, const float* bias, float* C, float* D, int ldc, const int* brow = nullptr, int ldbias = 0);
int gemm_set_attributes();

// ---- gemm_bf3.hip: the same GEMMs on the bf16 matrix pipe: both operands split into three bf16 planes (fp32 == p0 + p1 + p2
// exactly), six products per tile, fp32 accumulation.  A stays fp32 in memory (split inside the kernel), Bt is the pre-split
// weight matrix in the "bf3" layout: per row, K/32 blocks of [plane0: 32 bf16][plane1][plane2] = 192 B; ldb counts bf16 elements
// per row = 3 x the padded K of the full matrix.  Everything else as launch_gemm_nt.
// neg_from_block: k-blocks (of 32) from this index on are stored negated - the second, sign-flipped phase of the accumulation
// (gemm_bf3.hip, "Accumulation bias"); kneg of the launcher = the number of leading k-steps of THIS launch that are not negated
constexpr int BF3_NO_NEG = 1 << 30;
// neg_from_block = BF3_ALT: every ODD k-block is stored negated - the operand form of gemm_bf3a.hip / gemm_head.hip, which keep two
// accumulator sets (even / odd k-steps) and subtract them in the epilogue: the one-signed truncation of the bf16 MFMA accumulation
// hits two interleaved half-sums of identical statistics and cancels without a tunable split point (gemm_bf3a.hip, "Accumulation")
constexpr int BF3_ALT = -2;
void split_bf3_host(const float* w, int rows, int K, unsigned short* out, int neg_from_block);                 // host form (weights)
int launch_split_bf3(hipStream_t s, const float* src, int ld, int M, int K, unsigned short* dst, int ldd,
                     int neg_from_block = BF3_NO_NEG);  // device form (tests)
int launch_gemm_bf3_cfg(hipStream_t stream, int cfg, int epi, const float* A, int lda, const unsigned short* Bt, int ldb, int M,
                        int N, int K, const float* bias, float* C, float* D, int ldc, const int* brow = nullptr, int ldbias = 0,
                        int kneg = BF3_NO_NEG);
int gemm_bf3_set_attributes();

// ---- gemm_bf3a.hip: the same product with the ACTIVATIONS pre-split as well (A3: bf3 layout, lda3 = bf16 elements per row, a
// multiple of 96): both operands reach LDS by DMA, no vector work in the main loop.  out3: the GELU / chain-rule epilogues write
// C in bf3 form (C3, ldc3 >= 3 * pad32(N)) for the next layer instead of fp32 (C, ldc); D is fp32 [M][ldc] either way.
int launch_gemm_bf3a_cfg(hipStream_t stream, int cfg, int epi, bool out3, const unsigned short* A3, int lda3, const unsigned short* Bt,
                         int ldb, int M, int N, int K, const float* bias, float* C, unsigned short* C3, int ldc3, float* D, int ldc,
                         const int* brow = nullptr, int ldbias = 0, int alt = 0);
// alt: 0 = weights stored with their own sign (result = sum of the two accumulator sets), 1 = BF3_ALT weights and the launch starts
// on an even k-block (result = even - odd), 2 = BF3_ALT weights, odd first k-block (result = odd - even)
int gemm_bf3a_set_attributes();

// ---- gemm_h2.hip: the same product on fp16x2-split operands ("h2", gemm_h2_common.h: fp32 == hi + lo / 4096 to 2^-24; per row,
// K/32 blocks of [hi: 32 fp16][lo: 32 fp16] = 128 B; ld counts 16-bit elements = 2 x the padded K): three matrix instructions per
// tile and k-step instead of six.  out2: the GELU / chain-rule epilogues write C in h2 form (C2, ldc2 >= 2 * pad32(N)).
// alt as launch_gemm_bf3a_cfg (1 / 2: weights in the H2_WEIGHT form and activations in the H2_ACT form; 0: both plain).
int launch_gemm_h2_cfg(hipStream_t stream, int cfg, int epi, bool out2, const unsigned short* A2, int lda2, const unsigned short* Bt,
                       int ldb, int M, int N, int K, const float* bias, float* C, unsigned short* C2, int ldc2, float* D, int ldc,
                       const int* brow = nullptr, int ldbias = 0, int alt = 0);
int gemm_h2_set_attributes();
// mode: 0 plain, 1 activation form (lo of the odd k-blocks negated), 2 weight form (hi of the odd k-blocks negated)
enum { H2_PLAIN = 0, H2_ACT = 1, H2_WEIGHT = 2 };
bool split_h2_host(const float* w, int rows, int K, unsigned short* out, int mode);  // false: an entry does not fit fp16's range
int launch_split_h2(hipStream_t s, const float* src, int ld, int M, int K, unsigned short* dst, int ldd, int mode, int* ovf = nullptr);
#ifdef AIMNET_BF3_TIMING
int gemm_h2_read_stamps(unsigned long long* host1024);
#endif

// ---- gemm_chain.hip: one MLP (forward or backward sweep) as ONE launch.  A block owns a panel of 16 / 32 / 48 rows and the full width
// of every layer; hidden activations stay in LDS (h2 form), weights stream L2 -> registers in a host-packed fragment order
// (chain_pack_weights).  Same products and accumulation order as gemm_h2.hip: bitwise-equal results.
constexpr int CHAIN_MAX_KB = 23, CHAIN_MAX_PASS = 5;
// tile slots per wave of a pass' column group A (group B takes the rest; gemm_chain.hip)
constexpr int chain_group_a(int nt) { return nt >= 3 ? 2 : 0; }
enum { CH_BIAS_F32 = 0, CH_GELU_F32 = 1, CH_GELU_H2G = 2 };  // epilogue of a forward chain's LAST pass (hidden passes: GELU -> LDS)
struct ChainPass {          // a layer, or a column range of a wide layer
  const void* w[2];         // packed weight streams of the pass' column groups A and B (chain_pack_weights; w[0] NULL: no group A)
  int kb0;                  // first k-block of the LDS operand this pass reads (0)
  int ncols;                // real output columns (a multiple of 16; of 32 when the output is the next pass' operand)
  int epi;                  // CH_* (last forward pass only)
  const float* bias;        // forward: [ncols], or a [64][ldbias] table indexed by brow[row]
  const int* brow;
  int ldbias;
  float* D;                 // forward: GELU' out (may be NULL); backward hidden passes: GELU' in; [M][ldd] at this pass' first column
  int ldd;
  float* C;                 // fp32 output of a non-LDS pass, at this pass' first column
  int ldc;
  unsigned short* C2;       // CH_GELU_H2G: the output in h2 form [M][ldc2]
  int ldc2;
};
struct ChainArgs {
  const unsigned short* x;  // input rows, h2 form, at the first k-block pass 0 reads; ldx 16-bit elements per row
  int ldx, M;
  ChainPass p[CHAIN_MAX_PASS];
};
// id of the instantiated shape that has these k-steps / tile slots per pass (-1: none - use the per-layer launches)
int chain_find_shape(int n_waves, bool bwd, int n_pass, const int* nk, const int* nt, int n_hidden);
int launch_gemm_chain(hipStream_t stream, int shape, const ChainArgs& a);
// columns n0 .. n0 + 16 nw nt - 1 (= rows of the h2 weight matrix w2 [n_rows][2 ldk], H2_WEIGHT form; rows >= n_rows: zeros), k-blocks
// kb0 .. kb0 + nk - 1, in the order the kernel consumes them
void chain_pack_weights(const unsigned short* w2, int n_rows, int ldk, int n0, int nw, int nt, int kb0, int nk, std::vector<unsigned short>& out);

// ---- gemm_head.hip: the energy head 256 -> 128 -> 128 -> 1 forward and backward in one launch (operands of gemm_bf3a.hip)
struct HeadFusedArgs {
  const unsigned short* aim3;  // [M][lda3] head input in bf3 form (written by the last MLP layer's epilogue)
  int lda3;
  const unsigned short *w1, *w2, *w2t, *w1t;  // bf3 weights (BF3_ALT): W1 [128][3*256], W2 [128][3*128], W2^T [128][3*128], W1^T [256][3*128]
  const float *b1, *b2, *w3, *b3;             // biases [128], last layer [128] and its bias [1]
  // (the four weight operands are in the BF3_ALT form: odd k-blocks negated)
  const float* dlast;                         // GELU' of the layer that produced the head input, fp32 [M][ldd] (grad only)
  int ldd;
  float* e_atom;                              // [M]
  unsigned short* zbar3;                      // [M][ldz3] adjoint of that layer's pre-activation, bf3 form (grad only)
  int ldz3;
  int M, grad;
  int fmt = 1;  // 1: operands in the bf16x3 form (BF3_ALT weights), 2: in the fp16x2 form (H2_WEIGHT weights, H2_ACT activations)
};
int launch_head_fused(hipStream_t s, const HeadFusedArgs& a);
#ifdef AIMNET_BF3_TIMING
int gemm_bf3_read_stamps(unsigned long long* host1024);  // measurement build only (tests/tools/bf3_timing.sh)
int gemm_bf3a_read_stamps(unsigned long long* host1024);
#endif

// ---- nlist.hip --------------------------------------------------------------------------------
struct NlistBuffers {      // all device pointers, carved from the caller's workspace
  int* mol_start;          // [n_mol + 1]
  float* xw;               // [n_atoms, 3] coordinates wrapped into the cell (== coord when non-periodic)
  void* sys;               // [n_mol] NlistSystem
  int* atom_bin;           // [n_atoms]
  int* bin_count;          // [max_bins + 1]
  int* bin_start;          // [max_bins + 1]
  int* bin_fill;           // [max_bins]
  int* sorted_tmp;         // [n_atoms]
  int* sorted;             // [n_atoms]
  float4* xs;              // [n_atoms] bin-ordered (x, y, z, atom id)
  void* sorted_tmp_xq;     // [n_atoms] float4 bin-ordered (x, y, z, charge) for the list-free DSF walk
  int* mol_c;              // [n_atoms] the caller's mol_idx clamped to [0, n_mol) (launch_mol_start): what every kernel indexes with
  float prebinned_width = 0.0f;  // host flag: launch_wrap prepared the bins for this width (launch_bins starts at the scan)
  bool bins_done = false;  // host flag: launch_prep_small also scanned / filled / ordered the bins for prebinned_width
  bool binned = false;     // host flag: `sys` + bins describe this batch (periodic cells, or bounding boxes via launch_bbox)
};
size_t nlist_scratch_bytes(int n_atoms, int n_mol);
size_t nlist_xw_offset(int n_mol);  // byte offset of NlistBuffers::xw inside the scratch
void nlist_carve(NlistBuffers& b, char* base, int n_atoms, int n_mol);
// bad: input sanity flags (bit 0 atomic number outside [0,63], bit 1 mol_idx outside [0,n_mol), bit 2 mol_idx not sorted);
// mol_c [n_atoms]: the clamped copy of mol_idx; + the species pass (launch_species) when slot_of_z is given
// cell + bin-grid setup of the periodic fast path (launch_wrap with a bin width) as a rider block of launch_mol_start: it does not
// need that launch's output (atom counts by binary search in the sorted mol_idx).  sys == NULL: none.
struct CellSetupRider {
  const float* cell; int n_cell, p0, p1, p2; const int* pbc_sys; void* sys; float w; int* bin_count; int n_zero;
};
bool cell_setup_rides(int n_atoms, int n_mol);  // the setup fits one block's zeroing pass and system loop
CellSetupRider cell_setup_rider(const float* cell, int n_cell, const int pbc[3], const int* pbc_sys, float bin_width, int n_atoms,
                                int n_mol, NlistBuffers& b);
int launch_mol_start(hipStream_t s, const int* mol_idx, int n_atoms, int n_mol, int* mol_start, int* mol_c,
                     const int* numbers = nullptr, int* bad = nullptr, const int* slot_of_z = nullptr, int* aslot = nullptr,
                     unsigned long long* present_part = nullptr, const CellSetupRider* cell_setup = nullptr,
                     // != NULL: the sanity flags go to bad_part[ceil(n_atoms / 64)] (one plain store per wave) instead of atomics
                     // into *bad - for a status array that nobody has zeroed (nlist_status_owned_block combines them)
                     int* bad_part = nullptr);
// wrap coordinates (periodic) or copy them (non-periodic) into b.xw
int launch_wrap(hipStream_t s, const float* coord, const int* mol_idx, int n_atoms, int n_mol, const float* cell,
                int n_cell, const int pbc[3], NlistBuffers& b, const int* pbc_sys = nullptr, float bin_width = 0.0f,  // pbc_sys: device [n_cell][3] or NULL
                bool setup_done = false);  // the cell + bin-grid setup already ran as a rider of launch_mol_start
// small batches: status zeroing + launch_mol_start + launch_wrap(bin_width) + the bin kernels of launch_nlist in ONE launch
// (cell == NULL, molecules: status zeroing + launch_mol_start + the coordinate copy)
bool prep_small_applies(int n_atoms, int n_mol, bool periodic);
int launch_prep_small(hipStream_t s, const float* coord, const int* mol_idx, const int* numbers, int n_atoms, int n_mol,
                      const float* cell, int n_cell, const int pbc[3], const int* pbc_sys, float bin_width, int* status,
                      const int* slot_of_z, int* aslot, unsigned long long* present_part, NlistBuffers& b);
#ifdef AIMNET_PREP_TIMING
int prep_read_stamps(unsigned long long* host16);  // measurement build only (tests/tools/prep_timing.sh)
#endif
// non-periodic systems: give every molecule the cell grid of its bounding box (after launch_wrap), so that launch_nlist
// takes the cell-list path instead of the O(n^2) per-molecule scan; worth it from ~10^3 atoms per molecule
int launch_bbox(hipStream_t s, int n_mol, NlistBuffers& b);
// build one full neighbour matrix for `cutoff` from b.xw; rows real-first; entries beyond the row
// count are set to `fill_value` when fill_rows != 0; status[0] = max count (atomicMax), status[1] = overflow
// bin_width > 0: (re)bin the periodic systems into slabs >= bin_width thick first; <= 0: reuse the last bins
// DFT-D3 coordination numbers as a rider of the (cell-grid) list build that serves D3: the builder has every pair's distance in
// hand, so cn_i and the atom's five reference weights (d3.hip, d3_cn_kernel's outputs) cost a few instructions per hit there
// instead of a pass over the finished matrix.  d3w == NULL: none.
struct D3CnRider {
  const int* aslot = nullptr;    // [n_atoms] species slot
  const float* rcov = nullptr;   // [ns]
  const int* nref = nullptr;     // [ns]
  const float* cnref = nullptr;  // [ns][5]
  float* d3w = nullptr;          // [n_atoms][12]
};
int launch_nlist(hipStream_t s, int n_atoms, int n_mol, const int* mol_idx, const float* cell, int n_cell,
                 const int pbc[3], float cutoff, float bin_width, int cap, int fill_value, int fill_rows,
                 NlistBuffers& b, int* nb_idx, int* nb_shift, int* nb_cnt, int* status_max, int* status_ovf,
                 float4* pg = nullptr,
                 // != NULL: the status reduction is NOT launched; *status_later = the per-row counts for nlist_status_block
                 // (cellwalk.h) as riders of a later launch - valid until the next list build
                 const int** status_later = nullptr,  // pg [n_atoms, cap] (may be NULL): also emit the pair geometry (u, d) of every entry
                 // cn != NULL and the batch is binned: the D3 coordination numbers ride (then *cn_done = true)
                 const D3CnRider* cn = nullptr, bool* cn_done = nullptr);
int launch_bins(hipStream_t s, int n_atoms, int n_mol, const int* mol_idx, float width, NlistBuffers& b);
// caller-supplied neighbour matrix [n_atoms][width] (+ integer shifts [n_atoms][width][3] or NULL) -> the engine's row format
// (valid entries compacted in order, shifts packed, optional pair geometry from the coordinates as given in b.xw); status as
// launch_nlist; bad: bit 3 = a shift outside +-127 or an unshifted self pair
int launch_import_list(hipStream_t s, const int* ext_idx, const int* ext_shift, int width, int n_atoms, const int* mol_idx,
                       const float* cell, int n_cell, int cap, NlistBuffers& b, int* nb_idx, int* nb_shift, int* nb_cnt,
                       int* status_max, int* status_ovf, float4* pg, int* bad);
// bad |= 16 unless every entry (i -> j, s) has its mirror (j -> i, -s) in the row of j (caller-supplied matrices)
int launch_list_symmetry_check(hipStream_t s, const int* nb_idx, const int* nb_shift, const int* nb_cnt, int cap, int n_atoms,
                               int* bad,
                               int max_check = 1 << 30);  // pairs per row that are verified (long-range matrices: a sample);

// ---- conv.hip ---------------------------------------------------------------------------------
struct BasisParams {  // radial basis of AEVSV (aev.py:66-81), passed by value
  float rc, eta;
  float shifts[16];
};
// nq (conv_fwd / unconcat / conv_bwd): charge channels convolved with the features - 0 in pass 0, else 1 or 2 (NSE models);
// q / qbar are planes [nq][n_atoms], agh_q [nq][G][H], Vqsave [N][nq][H*3], Sqbar [N][nq][G*4]
int launch_conv_fwd(hipStream_t s, int nq, const float* a, const int* row_of, const float* q, const int* nb_idx,
                    const int* nb_cnt, const float4* pg, int cap, const float* agh_a, const float* agh_q, BasisParams bp,
                    float* x, int ldx, float* Vsave, float* Vqsave, int n_atoms, const int* order,
                    bool species_moments = false,  // pass 0 (row_of given, nq = 0): per-element moments, no row gathers
                    int split_max = 1024,          // atoms up to which the 4-waves-per-atom form is used
                    int x_split = 0);              // rows pre-split: 1 for gemm_bf3a.hip (x: bf16 elements, 3 * ldx per row), 2 for gemm_h2.hip (fp16 hi / lo, 2 * ldx)
int launch_unconcat(hipStream_t s, int nq, const float* xbar, int ldx, const float* Vsave, const float* Vqsave,
                    const float* agh_a, const float* agh_q, float* Sbar, float* Sqbar, int n_atoms);
int launch_conv_bwd(hipStream_t s, int nq, bool need_abar, bool stress, const float* a, const int* row_of,
                    const float* q, const float* Sbar, const float* Sqbar, const int* nb_idx, const int* nb_cnt, const float4* pg,
                    int cap, BasisParams bp, const float* xbar, int ldx, const float* abar_in, float* abar_out,
                    const float* qbar_in, float* qbar_out, float* fgrad, float* virial_atom, int n_atoms, const int* order,
                    float4* pairbuf = nullptr, bool pb_accum = false,  // pairbuf: the reverse-pair (XE) form, see conv.hip
                    int split_max = 1024);
bool pair_rev_supported(int n_atoms, int cap);  // row capacity / atom count the reverse-pair map handles
// reverse-pair map rev[i * cap + m] = position of (i, -shift) in the row of idx[i][m]: per-atom 256-slot hash tables `tab`
// (pair_hash_bytes) built and probed on the device
size_t pair_hash_bytes(int n_atoms);
int launch_pair_rev_hash(hipStream_t s, const int* nb_idx, const int* nb_shift, const int* nb_cnt, int cap, int n_atoms,
                         unsigned long long* tab, int* rev);
int launch_pair_force(hipStream_t s, const int* nb_idx, const int* nb_cnt, const int* rev, const float4* pairbuf, int cap,
                      int n_atoms, const float* fgrad, float* forces, int* nf = nullptr);  // forces = -(fgrad + pair terms)
// `order` (conv_fwd / conv_bwd / conv_bwd_p0): optional permutation of the atoms giving the PROCESSING order - the
// bin-sorted order of the cell list for periodic systems - so that an XCD's centres and the rows they gather stay
// spatially coherent (and L2-resident) whatever the order of the input file; NULL = input order.
// pass-0 backward through species moments (conv.hip)
int launch_species(hipStream_t s, const int* numbers, const int* slot_of_z, int n_atoms, int* aslot,
                   unsigned long long* present_part);
int launch_unconcat_p0(hipStream_t s, const float* xbar, int ldx, const float* Vsave, const float* agh_a, const float* afv,
                       const int* z_of_slot, int nslots, const unsigned long long* present_part, int n_part, float* T,
                       int n_atoms);
int launch_conv_bwd_p0(hipStream_t s, bool stress, const float* T, int nslots, const int* aslot, const int* nb_idx,
                       const int* nb_cnt, const float4* pg, int cap, BasisParams bp, float* fgrad, float* virial_atom,
                       int n_atoms, const int* order,
                       float4* pairbuf = nullptr);  // pairbuf: reverse-pair form (own moments only, G1 added to the pair buffer)
// ---- conv_mfma.hip: the same three steps with the pair contractions on v_mfma_f32_4x4x1_16B_f32 (one wave per centre atom,
// systems above SPLIT_MAX_ATOMS).  SbarT is the Sbar buffer in the lane-(g,c) plane layout that conv_bwd_mfma reads.
// a_t: the feature table transposed to [g][a] (the MFMA operand layout), a: the natural [a][g] one
int launch_conv_fwd_mfma(hipStream_t s, int nq, const float* a, const float* a_t, const int* row_of, const float* q, const int* nb_idx,
                         const int* nb_cnt, const float4* pg, int cap, const float* agh_a, const float* agh_q, BasisParams bp,
                         float* x, int ldx, float* Vsave, float* Vqsave, int n_atoms, const int* order);
int launch_unconcat_t(hipStream_t s, int nq, const float* xbar, int ldx, const float* Vsave, const float* Vqsave,
                      const float* agh_a, const float* agh_q, float* SbarT, float* Sqbar, int n_atoms);
int launch_conv_bwd_mfma(hipStream_t s, int nq, bool need_abar, bool stress, const float* a_t, const int* row_of, const float* q,
                         const float* SbarT, const float* Sqbar, const int* nb_idx, const int* nb_cnt, const float4* pg, int cap,
                         BasisParams bp, const float* xbar, int ldx, const float* abar_in, float* abar_out, const float* qbar_in,
                         float* qbar_out, float* fgrad, float* virial_atom, int n_atoms, const int* order);
int launch_mfma4_probe(hipStream_t s, float* out);  // lane-layout probe of the 4x4x1 16-block MFMA (tests)
int conv_split_max_default();  // 1024
// stand-alone reference-op forms (conv_sv_2d_sp_wp.py:90-164)
int launch_conv_sv_fwd(hipStream_t s, const float* a, const int* idx, const float* g, float* out, int B, int A, int G,
                       int M);
int launch_conv_sv_bwd(hipStream_t s, const float* grad_out, const float* a, const int* idx, const float* g,
                       float* grad_a, float* grad_g, int B, int A, int G, int M);
int launch_conv_sv_bwd_bwd(hipStream_t s, const float* grad_out, const float* grad2_a, const float* grad2_g, const float* a,
                           const int* idx, const float* g, float* ggo, float* ga2, float* gg2, int B, int A, int G, int M);

// the reverse-pair map as a rider (pairmap.h): `n_blocks` = ceil(n_atoms / 4) blocks of 256 threads run the hash build
// (SrRiders::hash) or the lookup (launch_coulomb_dsf_walk / launch_energy_reduce), 0 = none
struct PairMapRider {
  const int* nb_idx; const int* nb_shift; const int* nb_cnt; int cap, n_atoms; unsigned long long* tab; int* rev; int n_blocks;
};
// ---- model.hip --------------------------------------------------------------------------------
// S = slices (blocks) per molecule for the per-molecule reductions; `part` = scratch [n_sys * S * 9] doubles
// nq = charge channels (1, or 2 for NSE models): q planes [nq][n_atoms], charge / Fm / Dm / Wbar planes [nq][n_mol];
// the MLP output row is [q~ (nq) | f~ (nq) | delta_a (256)] (aimnet2.py:123-130)
// Domain decomposition of one system over ranks (include/aimnet_hip.h, aimnet_engine_set_dd): the rank's owned-atom mask and the
// caller's exchange function (what = 0: all-reduce n floats in place, 1: fill the halo entries of the charge planes)
struct DdLink {
  const float* owned;
  int (*fn)(void* ctx, int32_t what, void* dev_ptr, int64_t n_float, void* hip_stream);
  void* ctx;
};
// halo rows leave the energy sums, the Coulomb adjoints / direct forces and the backward seed (model.hip, dd_mask_kernel)
int launch_dd_mask(hipStream_t s, const float* owned, const int* numbers, const double* sae, float* e_atom, double* ecoul,
                   float* qbar, int nq, float* fgrad, float* virial_atom, void* seed, int seed_row_bytes, int n_atoms);
int launch_nse_fwd(hipStream_t s, const float* y, int ldy, int nq, const float* q_prev, const int* mol_start,
                   const float* charge, int n_mol, int n_atoms, int S, float* part, float* q_new, float* Fm, float* Dm,
                   // upd_a_new != NULL: a_new = a + delta_a (launch_update_a with these arguments) rides on the same launch
                   const float* upd_a = nullptr, const int* upd_row_of = nullptr, float* upd_a_new = nullptr,
                   float* upd_a_t = nullptr,
                   const DdLink* dd = nullptr);  // != NULL: sums over owned atoms, all-reduced by dd->fn between the two launches
int launch_charge_sum(hipStream_t s, const float* q2, int n_atoms, float* q_tot, float* q_spin);
int launch_update_a(hipStream_t s, const float* a, const int* row_of, const float* y, int ldy, int nq, int n_atoms, float* a_new,
                    float* a_t = nullptr);  // a_t: optional copy in the operand layout of the MFMA conv kernels
// d / zbar (may be NULL): also writes the backward seed zbar = w * d (d = GELU' of the layer below, ldh wide)
int launch_head_last(hipStream_t s, const float* h, int ldh, const float* w, const float* b, int k, int n_atoms,
                     float* e_atom, const float* d, float* zbar);
int launch_energy_reduce(hipStream_t s, const float* e_atom, const double* ecoul, const int* numbers,
                         const double* sae, const int* mol_start, int n_mol, int S, double* part, double* energy,
                         // copy_dst != NULL: copy_n floats copy_src -> copy_dst ride on the same launch (the charges output)
                         const float* copy_src = nullptr, float* copy_dst = nullptr, int copy_n = 0,
                         const PairMapRider* rev_rider = nullptr,  // the lookup of the reverse-pair map rides on this launch
                         int* nf = nullptr);                       // status word that takes STATUS_NONFINITE
// ---- ewald.hip: Ewald summation of a periodic system (LRCoulomb "ewald", lr.py:617-720) ------------------------------------
struct EwaldSystem {   // per periodic system, written by ewald_setup_kernel from the cell, the atom count and the accuracy
  float alpha, rc;     // splitting parameter, real-space cutoff
  float kc2, inv4a2;   // reciprocal-space cutoff squared, 1 / (4 alpha^2)
  float phi_bg;        // potential of the neutralising background, -pi Q / (V alpha^2)
  int nmax[3];         // |n_a| <= nmax[a]: the box of integer triplets that holds the k sphere
  int n2w, n3w;        // 2 nmax[1] + 1, 2 nmax[2] + 1
  int k_offset, n_box; // this system's slice [k_offset, k_offset + n_box) of the k arrays (a multiple of EWALD_KB entries)
  double pref;         // 8 pi / V (the potential's prefactor, half space doubled)
  double b[9];         // k_c = sum_a n_a b[a * 3 + c]
  double inv[9];       // fractional coordinate a = sum_c x_c inv[c * 3 + a]
  int mesh[3];         // pme.hip: mesh points along the three cell vectors (0: this system's mesh did not fit the capacity)
  int mesh_pts;        // mesh[0] mesh[1] mesh[2]
};
struct EwaldK {        // one entry of the k box (P = Q = 0: outside the half-space sphere)
  double P, Q;         // A(k) Re S(k), A(k) Im S(k);  A = pref exp(-k^2 / 4 alpha^2) / k^2
  float kx, ky, kz, vfac;  // the k vector; 2 (1 / k^2 + 1 / (4 alpha^2)) for the strain derivative
  int n1, n2, n3, pad;
};
constexpr int EWALD_KB = 8;  // k entries per block of the structure-factor kernel
struct EwaldBuffers {  // device pointers carved from the workspace
  EwaldSystem* sys;    // [n_mol]
  double* frac;        // [n_atoms][3] fractional coordinates in double
  EwaldK* k;           // [max_k]
  int max_k;
  // pme.hip (AIMNET_COULOMB_PME): per system a slice of max_mesh points
  long long* meshq;    // [n_mol][max_mesh] charge mesh in 2^-44 fixed point (integer atomics: order-independent sums), later the
                       // potential mesh as doubles
  double* ma;          // [n_mol][max_mesh][2] complex work meshes (ping-pong of the axis transforms)
  double* mb;
  double* bmod;        // [n_mol][3][PME_MAX_AXIS] inverse squared moduli of the spline's Fourier coefficients
  double* vpart;       // [n_mol][max_parts][8] per-block sums of theta |Q^|^2 (k_a k_b vfac, 1)
  int max_mesh, max_parts;
};
constexpr int PME_MAX_AXIS = 512;   // mesh points per axis at most (direct axis transforms in LDS)
constexpr int PME_PART = 1024;      // mesh points per block of the influence-function kernel
// parameters + k boxes of every system; *status_k = k entries the batch needs (> max_k: the boxes were truncated, results meaningless)
int launch_ewald_setup(hipStream_t s, const float* cell, int n_cell, const int* mol_start, const int* mol_idx, const float* xw,
                       const float* charge, int nq, int n_atoms, int n_mol, float accuracy, EwaldBuffers& b, int* status_k);
int launch_ewald_frac(hipStream_t s, const float* xw, const int* mol_idx, int n_atoms, const EwaldBuffers& b);
// ---- pme.hip: smooth particle-mesh Ewald (LRCoulomb "pme", lr.py:752-775): the reciprocal-space sum on a mesh --------------
// per-system (alpha, rc, mesh), the charge mesh zeroed, the spline moduli; *status = mesh points the largest system needs (> max_mesh: its mesh
// was skipped, results meaningless; INT32_MAX: an axis beyond PME_MAX_AXIS)
int launch_pme_setup(hipStream_t s, const float* cell, int n_cell, const int* mol_start, const float* charge, int nq, int n_mol,
                     float accuracy, EwaldBuffers& b, int* status);
// order: the engine's bin order of the atoms (a permutation of 0 .. n_atoms-1) or NULL - the charge assignment groups atoms by it
int launch_pme_recip(hipStream_t s, bool grad, bool stress, const float* xw, const float* q, const int* mol_idx, const int* mol_start,
                     const int* order, int n_atoms, int n_mol, const EwaldBuffers& b, float factor, double* ecoul, float* qbar, float* fgrad, float* virial_atom);
// reciprocal-space sum + neutralising background, ACCUMULATED onto the per-atom energies / adjoints the pair kernels have stored
int launch_ewald_recip(hipStream_t s, bool grad, bool stress, const float* q, const int* mol_idx, const int* mol_start, int n_atoms,
                       int n_mol, const EwaldBuffers& b, float factor, double* ecoul, float* qbar, float* fgrad, float* virial_atom);

struct CoulombParams {
  float factor;      // 1/2 Hartree Bohr
  float sr_rc;       // exp / cosine envelope radius (SRCoulomb)
  int sr_envelope;   // 0 exp, 1 cosine
  float dsf_rc, dsf_alpha;
  const EwaldSystem* ewald = nullptr;  // != NULL: the list-free walk sums the REAL-space Ewald term erfc(alpha d) / d with the
                                       // system's own (alpha, rc) instead of the shifted DSF pair term
};

// ---- d3.hip: DFT-D3(BJ) two-body dispersion on a full neighbour list ------------------------------
struct D3Params {
  float s6, s8, a1, a2;
  float r_on, r_off;  // S5 switch window in Bohr
};
struct D3Tables {  // device pointers, indexed by species SLOT (engine.hip: slot_of_z), built from the Z-indexed reference tables
  int ns;
  const float* c6slot;  // [ns][ns][5][5]
  const float* cnref;   // [ns][5]   reference coordination numbers of each element's reference systems
  const int* nref;      // [ns]      number of reference systems
  const float* rcov;    // [ns]      covalent radii (Bohr, already scaled as in dftd3_data.pt)
  const float* r4r2;    // [ns]
};
int launch_dftd3(hipStream_t s, bool grad, bool stress, const float* xw, const int* mol_idx, const float* cell, int n_cell,
                 const int* aslot, const int* nb_idx, const int* nb_shift, const int* nb_cnt, int cap, D3Tables T, D3Params P,
                 float cutoff, int n_atoms, float4* xs4, float* d3w, float* dEdcn, double* ecoul, float* fgrad,
                 float* virial_atom, bool with_dsf, CoulombParams cp, const float* q, float* qbar,
                 bool cn_done = false,
                 const DdLink* dd = nullptr);  // domain decomposition: halo rows of d3w / dE/dcn come from their owners (dd->fn, what = 2)  // cn_done: d3w was filled by the list build (D3CnRider): no d3_cn_kernel launch
// with_dsf: the DSF Coulomb pair sum (cutoff == cp.dsf_rc) is evaluated in the same pair pass; adds to ecoul / qbar too
// Independent work that rides on the SR-Coulomb launch (role-dispatched blocks behind the pair blocks; a kernel boundary costs
// 4-5 us on the device): launch_head_last's arguments (n_head_blocks = ceil(n_atoms / 4), 0 = none) and the charge stream of the
// list-free DSF walk (n_stream_blocks = ceil(n_atoms / 256), 0 = none; xs / xq = NlistBuffers::xs / sorted_tmp_xq)
struct SrRiders {
  const float* h; int ldh; const float* w; const float* b; int k; float* e_atom; const float* d; float* zbar; int n_head_blocks;
  const float4* xs; float4* xq; float* charges_out; int n_stream_blocks;
  // simple_xw != NULL: the "simple" LRCoulomb term (launch_coulomb_simple with these arguments) in the same waves
  const float* simple_xw; const int* simple_mol_idx; const int* simple_mol_start;
  PairMapRider hash;  // the hash build of the reverse-pair map (its lookup rides on a later launch)
  // the status words of the short-range list (nlist_status_block, cellwalk.h): n_status_blocks = ceil(n_atoms / 1024), 0 = none
  const int* cnt_true; int status_cap; int* status_max; int* status_ovf; int n_status_blocks;
  // status_all != NULL ("owned" form, ONE rider block): nobody zeroed the status array - the block reduces the row counts and the
  // sanity flags of launch_mol_start (bad_part) and STORES all eight words (word 7 is left alone when keep7: Ewald wrote it)
  int* status_all; const int* bad_part; int keep7;
};
// embedded SRCoulomb subtraction over the rc list (sets ecoul/qbar/fgrad/virial_atom)
int launch_coulomb_sr(hipStream_t s, bool grad, bool stress, bool enabled, const float* q, const int* nb_idx,
                      const int* nb_cnt, const float4* pg, int cap, CoulombParams cp, int n_atoms, double* ecoul,
                      float* qbar, float* fgrad, float* virial_atom, const SrRiders* riders = nullptr);
int launch_coulomb_simple(hipStream_t s, bool grad, const float* q, const float* xw, const int* mol_idx,
                          const int* mol_start, CoulombParams cp, int n_atoms, double* ecoul, float* qbar,
                          float* fgrad);
int launch_coulomb_dsf(hipStream_t s, bool grad, bool stress, const float* q, const float* xw, const int* mol_idx,
                       const float* cell, int n_cell, const int* nb_idx, const int* nb_shift, const int* nb_cnt,
                       int cap, CoulombParams cp, int n_atoms, double* ecoul, float* qbar, float* fgrad,
                       float* virial_atom, bool simple = false);  // simple: w = 1 / d over every entry (caller-supplied nbmat_lr)
// periodic DSF straight from the cell grid of the last launch_bins (no neighbour matrix)
int launch_coulomb_dsf_walk(hipStream_t s, bool grad, bool stress, const float* q, const int* mol_idx, NlistBuffers& b,
                            CoulombParams cp, int n_atoms, double* ecoul, float* qbar, float* fgrad, float* virial_atom,
                            float* charges_out = nullptr,  // charges_out: also copy q to the charges output
                            bool stream_done = false,     // the (x, y, z, q) stream was written by an SrRiders launch
                            const PairMapRider* rev_rider = nullptr);  // the lookup of the reverse-pair map rides on this launch
int launch_nse_bwd_reduce(hipStream_t s, const float* qbar, const float* y, int ldy, int nq, const int* mol_start, int n_mol,
                          int n_atoms, int S, float* part);  // part: [nq][n_mol][S] partial sums, consumed by launch_build_zbar
int launch_build_zbar(hipStream_t s, const float* qbar, const float* abar, const float* y, int ldy, const float* dlast,
                      const float* Fm, const float* Dm, const float* wpart, int S, const int* mol_idx, int n_atoms, int n_mol,
                      int n_feat, int nq, bool carry_q, float* zbar, float* qbar_next,
                      int zbar_split = 0,
                      // wpart == NULL: the blocks form the molecule sums themselves (no launch_nse_bwd_reduce in front: small systems);
                      // needs mol_start and qbar_next != qbar
                      const int* mol_start = nullptr,
                      const float* owned = nullptr);  // (domain decomposition: halo rows take no share of the sums' adjoint)  1: rows in the split form of gemm_bf3a.hip (bf16 elements, 3 * ldy per row), 2: of gemm_h2.hip (2 * ldy)
// launch_pair_force's arguments as a rider of the stress reduction (launch_finalize): n_blocks = ceil(n_atoms / 4), 0 = none
struct PairForceRider {
  const int* nb_idx; const int* nb_cnt; const int* rev; const float4* pairbuf; int cap; float* forces; int n_blocks;
};
// bit of status[6] that the kernels writing the energies and forces raise when a value is not finite: an MLP activation beyond
// fp16's range with the fp16x2-split GEMM operands (gemm_h2_common.h) surfaces there (or a genuine blow-up of the input geometry)
constexpr int STATUS_NONFINITE = 32;
// the molecule energy sums as riders of the stress launches (partial sums beside the virial sums, the slice sums beside the stress
// finish): they are only needed at the end, and two launch boundaries go.  part: its own [n_mol][S] partial sums.
struct EnergyRider {
  const float* e_atom; const double* ecoul; const int* numbers; const double* sae; double* part; double* energy; int n_mol;
  // forces-only evaluations (no stress launches): with one slice per molecule (S == 1) the sums ride on the force-negation launch,
  // together with the copy of the charges into the output (copy_dst != NULL) that the energy launch would have carried
  const float* copy_src; float* copy_dst; int copy_n;
};
// nf: status word that takes STATUS_NONFINITE (NULL: not reported)
int launch_finalize(hipStream_t s, const float* fgrad, const float* virial_atom, const int* mol_start,
                    const float* cell, int n_cell, int n_mol, int n_atoms, int S, double* part, float* forces,
                    float* stress, const PairForceRider* pair_force = nullptr, const EnergyRider* energy = nullptr,
                    bool whole_ok = false,  // the sums may run as whole-cell / whole-molecule blocks without a finish launch
                    int* nf = nullptr);
int launch_copy_f32(hipStream_t s, const float* src, float* dst, size_t n);

}  // namespace aimnet
