/*
 * aimnet_hip.h - C ABI of libaimnet_hip.so: the MI355X (gfx950) native AIMNet2 energy / force /
 * virial engine.  Plain pointers and sizes only; all buffers are DEVICE pointers owned by the
 * caller (the Python host allocates them as torch tensors), the library never allocates in the
 * hot path and never frees caller memory.  Every call is asynchronous on the given HIP stream.
 *
 * Each entry point replaces one boundary of the reference (paths relative to /root/reference):
 *
 *   aimnet_engine_eval            <- the span `data = self.model(data)` .. `get_derivatives`
 *                                    of AIMNet2Calculator.eval, aimnet/calculators/calculator.py:917-936,
 *                                    plus make_nbmat :1521-1702 and move_coord_to_cell neighbors.py:331
 *   aimnet_neighbor_list          <- nvalchemiops.torch.neighbors.neighbor_list as called in
 *                                    aimnet/calculators/neighbors.py:106-125 (full list, packed
 *                                    real-first, fill_value padding, integer shifts, overflow report)
 *   aimnet_conv_sv_2d_sp_fwd/bwd  <- torch.ops.aimnet.conv_sv_2d_sp_fwd / _bwd,
 *                                    aimnet/kernels/conv_sv_2d_sp_wp.py:252-340 (Warp kernels :90-164)
 *   aimnet_engine_create          <- "state_dict -> module on device" of aimnet/models/base.py:65-89
 *   aimnet_engine_hvp             <- AIMNet2Calculator.hessian_vector_product, calculator.py:1753-1989 (one double backward
 *                                    per vector) and calculate_hessian, calculators/derivatives.py:149-192 (the same vjp
 *                                    vmapped over the 3N unit vectors): here one analytic tangent sweep for K directions
 *
 * Errors: functions return 0 on success or a negative AIMNET_E_* code; no exceptions cross the
 * ABI.  Neighbour overflow is reported asynchronously through the `status` words (the host grows
 * the row capacity x1.5 and retries, mirroring AdaptiveNeighborList, neighbors.py:127-130).
 */
#ifndef AIMNET_HIP_H
#define AIMNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AIMNET_ABI_VERSION 11

#define AIMNET_OK 0
#define AIMNET_E_INVALID (-1)   /* bad argument / unsupported architecture */
#define AIMNET_E_HIP (-2)       /* a HIP runtime call failed; see aimnet_last_error */
#define AIMNET_E_WORKSPACE (-3) /* workspace too small */

#define AIMNET_MAX_PASS 4
#define AIMNET_MAX_LAYERS 6
#define AIMNET_MAX_SHIFTS 32

/* flags for aimnet_eval_options.flags */
#define AIMNET_FORCES 1u
#define AIMNET_STRESS 2u

/* Coulomb method of the external LRCoulomb module (aimnet/modules/lr.py:928-983) */
#define AIMNET_COULOMB_NONE 0
#define AIMNET_COULOMB_SIMPLE 1 /* all pairs inside a molecule, lr.py:311-331 */
#define AIMNET_COULOMB_DSF 2    /* damped shifted force, lr.py:559-615 */
#define AIMNET_COULOMB_EWALD 3  /* Ewald summation of a fully periodic system, lr.py:617-720 (`ewald`): the exact structure-factor
                                   sum; options.ewald_accuracy, options.ewald_max_k */
#define AIMNET_COULOMB_PME 4    /* smooth particle-mesh Ewald, lr.py:752-775 (`pme`): the reciprocal-space sum on a mesh (order-8
                                   B-splines, csrc/pme.hip); options.ewald_accuracy, options.pme_max_mesh */

/* Architecture of one AIMNet2 core model (aimnet/models/aimnet2.py:12-106 hyper-parameters). */
typedef struct aimnet_arch {
  int32_t nfeature;  /* A, 16 */
  int32_t nshifts;   /* G, 16 */
  int32_t ncomb_v;   /* H, 12 */
  int32_t n_pass;    /* number of message-passing MLPs, 3 */
  int32_t n_layers[AIMNET_MAX_PASS];                       /* linear layers per pass MLP */
  int32_t layer_dims[AIMNET_MAX_PASS][AIMNET_MAX_LAYERS + 1]; /* n_in, hidden..., n_out */
  int32_t last_linear[AIMNET_MAX_PASS];                    /* 1: no GELU after the last layer */
  int32_t head_n_layers;                                   /* energy head, last layer linear */
  int32_t head_dims[AIMNET_MAX_LAYERS + 1];
  float rc;                         /* aev.rc_s */
  float eta;                        /* aev.eta_s */
  float shifts[AIMNET_MAX_SHIFTS];  /* aev.shifts_s */
  int32_t sr_coulomb;               /* 1: embedded SRCoulomb subtraction (lr.py:986-1032) */
  int32_t sr_envelope;              /* 0 exp (ops.py:88), 1 cosine (ops.py:82) */
  float sr_rc;                      /* 4.6 */
  int32_t n_charge_channels;        /* num_charge_channels of AIMNet2.__init__ (aimnet2.py:21-28): 1 closed shell (0 reads
                                       as 1), 2 = open-shell NSE (alpha / beta electron-count channels) */
} aimnet_arch;

/* Host pointers to the fp32 weights, copied to the device once at create time.
 * mlp_w[p][l] is torch Linear.weight [out, in] row-major, mlp_b[p][l] the bias [out]. */
typedef struct aimnet_weights {
  const float* afv;     /* [64, A*G] embedding; rows of unsupported Z may be NaN */
  const float* agh_a;   /* [A, G, H] */
  const float* agh_q;   /* [n_charge_channels, G, H] */
  const float* mlp_w[AIMNET_MAX_PASS][AIMNET_MAX_LAYERS];
  const float* mlp_b[AIMNET_MAX_PASS][AIMNET_MAX_LAYERS];
  const float* head_w[AIMNET_MAX_LAYERS];
  const float* head_b[AIMNET_MAX_LAYERS];
  const double* sae;    /* [64] fp64 atomic shifts (core.py:71-97, utils.py:369-376) */
} aimnet_weights;

typedef struct aimnet_engine aimnet_engine;

/* Device inputs of one evaluation.  Flat layout: atoms of a molecule are contiguous and
 * mol_idx is non-decreasing (the reference assumes the same, nbops.py:346). */
typedef struct aimnet_inputs {
  int32_t n_atoms;
  int32_t n_mol;
  const float* coord;      /* [n_atoms, 3] */
  const int32_t* numbers;  /* [n_atoms] */
  const int32_t* mol_idx;  /* [n_atoms] */
  const float* charge;     /* [n_charge_channels, n_mol] channel-major: the total molecular charge, or for NSE models
                              the two planes Q/2 + (mult-1)/2 and Q/2 - (mult-1)/2 (aimnet2.py:94-100) */
  const float* cell;       /* [n_cell, 3, 3] row vectors, or NULL (non-periodic) */
  int32_t n_cell;          /* 0, 1 (shared) or n_mol */
  int32_t pbc[3];          /* periodic axes (used when cell != NULL) */
  const int32_t* pbc_sys;  /* device, [n_cell, 3] or NULL: per-system periodic axes (normalize_pbc, neighbors.py:309-321);
                              overrides pbc[] */
  /* Optional caller-supplied neighbour matrices - the reference skips its list builder when the input already carries `nbmat`
   * and hands the matrices to the model as they are (calculator.py:1069-1071).  All NULL (default): the engine builds its own
   * lists.  nbmat != NULL: NO list is built; device int32 rows [n_atoms][width], entries outside [0, n_atoms) are padding (the
   * reference pads with n_atoms); shifts int32 [n_atoms][width][3] lattice translations (required iff cell != NULL, |s| <= 127)
   * applied to the coordinates AS GIVEN (nothing is wrapped in this mode).  Every matrix must be FULL (both directions of every
   * pair present, as the reference's builders emit them): the short-range matrix is verified (status[6] bit 4), the others are
   * trusted.  nbmat_lr serves the LRCoulomb term (simple: 1/d over every entry, lr.py:311-331; DSF: required) and, without
   * nbmat_d3, the DFT-D3 term; with d3_cutoff == dsf_rc one matrix serves both (pass it as nbmat_lr).  The row capacities
   * options.max_nb / max_nb_lr / max_nb_d3 must be >= the widths. */
  const int32_t* nbmat;
  const int32_t* shifts;
  int32_t nbmat_width;
  const int32_t* nbmat_lr;
  const int32_t* shifts_lr;
  int32_t nbmat_lr_width;
  const int32_t* nbmat_d3;
  const int32_t* shifts_d3;
  int32_t nbmat_d3_width;
} aimnet_inputs;

typedef struct aimnet_eval_options {
  uint32_t flags;          /* AIMNET_FORCES | AIMNET_STRESS */
  int32_t coulomb;         /* AIMNET_COULOMB_* */
  float dsf_rc;            /* 15.0 */
  float dsf_alpha;         /* 0.2 */
  int32_t max_nb;          /* row capacity of the short-range (rc) neighbour matrix */
  int32_t max_nb_lr;       /* row capacity of the DSF neighbour matrix (0 if unused) */
  /* external DFT-D3(BJ) two-body dispersion, DFTD3 of aimnet/modules/lr.py:1335-1820 as wired by
   * calculator.py:234-247,999-1032; needs aimnet_engine_set_dftd3 first.  0 = off. */
  int32_t dftd3;
  float d3_s6, d3_s8, d3_a1, d3_a2;
  float d3_cutoff;         /* Angstrom: list cutoff = end of the S5 switch (smoothing_off, 15.0) */
  float d3_smoothing_on;   /* Angstrom: start of the S5 switch (cutoff * (1 - smoothing_fraction), 12.0) */
  int32_t max_nb_d3;       /* row capacity of the D3 neighbour matrix */
  /* AIMNET_COULOMB_EWALD: target accuracy of the splitting (set_lrcoulomb_method(..., ewald_accuracy=1e-6), calculator.py
   * :1566-1586; alpha, the real-space cutoff and the reciprocal-space cutoff follow from it, the cell volume and the atom count
   * of every system, on the device) and the capacity of the k arrays (entries of the integer boxes that hold the k spheres, summed
   * over the systems, a multiple of 8).  status[7] reports the entries the batch needed: > ewald_max_k means the sums were
   * truncated - evaluate again with at least that capacity. */
  float ewald_accuracy;
  int32_t ewald_max_k;
  /* AIMNET_COULOMB_PME: capacity of the mesh of ONE system in points (every system of the batch gets a slice of that size; 40
   * bytes per point).  The mesh dimensions follow from ewald_accuracy and the cell on the device; status[7] reports the points the
   * largest system needs: > pme_max_mesh means its mesh was skipped - evaluate again with at least that capacity; INT32_MAX means
   * a cell vector needs more than 512 mesh points (not supported). */
  int32_t pme_max_mesh;
} aimnet_eval_options;

typedef struct aimnet_outputs {
  double* energy;   /* [n_mol] eV, fp64 like the reference */
  float* charges;   /* [n_atoms] (NSE: alpha + beta, aimnet2.py:102-106) */
  float* forces;    /* [n_atoms, 3] or NULL */
  float* stress;    /* [max(n_cell,1), 3, 3] or NULL */
  int32_t* status;  /* [8]: 0 max neighbours found (rc list), 1 same for the LR list,
                              2 overflow flag rc list, 3 overflow flag LR list,
                              4 max neighbours found (D3 list), 5 overflow flag D3 list,
                              6 input sanity flags: bit 0 an atomic number outside [0, 63], bit 1 a mol_idx outside
                                [0, n_mol), bit 2 mol_idx not sorted, bit 3 a caller-supplied matrix holds a shift outside +-127
                                or an unshifted self pair, bit 4 the caller-supplied short-range matrix is not symmetric.  The first kernel writes clamped copies (atomic number
                                slots, mol_idx) into the workspace and every later kernel indexes through those, so any device
                                array is memory-safe; with a flag raised the results are meaningless.
                                Bit 5 (32) is NOT an input flag: the kernels that write the energies / forces saw a non-finite value.
                                With the fp16x2-split GEMM operands ("gemm_h2", the default above 256 rows) that is how an MLP
                                activation or adjoint beyond fp16's range (|x| >= 65504) surfaces: set_option("gemm_h2", 0) and
                                repeat the evaluation (the bf16x3 operands have fp32's range); still raised, the input is the cause.
                              7 Ewald: k-array entries the batch needs (compare with options.ewald_max_k); PME: mesh points the
                                largest system needs (compare with options.pme_max_mesh); 0 for other methods */
  float* spin_charges; /* [n_atoms] alpha - beta of an NSE model (aimnet2.py:103), or NULL; must be NULL for 1-channel models */
} aimnet_outputs;

int aimnet_abi_version(void);

int aimnet_engine_create(const aimnet_arch* arch, const aimnet_weights* w, int device, aimnet_engine** out);
void aimnet_engine_destroy(aimnet_engine* e);
const char* aimnet_last_error(void);

/* Reference tables of DFT-D3 (the reference loads aimnet/dftd3_data.pt in DFTD3.__init__, lr.py:1405-1423): HOST
 * pointers, indexed by atomic number 0 .. n_z-1; c6ab and cn_ref are [n_z][n_z][5][5], rcov and r4r2 [n_z].  Copied
 * (re-indexed by the model's species) at the call; AIMNET_E_INVALID if the CN table does not have the product
 * structure every published D3 table has (cn_ref[zi][zj][a][b] independent of zj and b). */
typedef struct aimnet_dftd3_tables {
  int32_t n_z;
  const float* c6ab;
  const float* cn_ref;
  const float* rcov;
  const float* r4r2;
} aimnet_dftd3_tables;
int aimnet_engine_set_dftd3(aimnet_engine* e, const aimnet_dftd3_tables* t);

/* Spatial domain decomposition of ONE large system over ranks (the reference has no counterpart and points
 * its users at one GPU per system, docs/tutorials/performance.md:275-285).  A rank evaluates the atoms it OWNS together with halo
 * copies of every atom (periodic images resolved by the host-side partitioner) within 3 x the model cutoff (and within the Coulomb
 * cutoff) as ONE NON-PERIODIC cluster: `owned` is a device array [n_atoms] of 1.0f (owned) / 0.0f (halo) that must stay valid
 * while evaluations run.  With it set, aimnet_engine_eval
 *   - sums the NSE charge normalisation (ops.nse, aimnet/ops.py:99-145) over OWNED atoms only and hands the per-molecule partial
 *     sums to `fn` (AIMNET_DD_SUM: all-reduce n floats in place) before it applies them - once per charge channel and pass, and
 *     once more per pass for the adjoint sums of the backward sweep;
 *   - hands the final charges to `fn` (AIMNET_DD_CHARGES: nq planes of n_atoms floats; overwrite the halo entries with their
 *     owners' values) in front of the Coulomb block, whose 15 A sums reach beyond the shell where the local charges are exact;
 *   - with DFT-D3 hands the per-atom reference weights (12 floats per atom, coordination number included) and, with gradients,
 *     dE/dcn (1 float per atom) to `fn` (AIMNET_DD_ROWS: [n_atoms][n_float / n_atoms] records; overwrite the halo rows with their
 *     owners'): a halo copy's coordination number needs ITS 15 A neighbourhood, which the cluster does not hold;
 *   - counts energies, Coulomb adjoints, direct Coulomb / dispersion forces and the backward seed of OWNED atoms only.
 * energy[m] is then the rank's share (sum over ranks = the system's energy); forces[] holds, for owned AND halo atoms, the rank's
 * partial -dE_rank/dx: the caller adds the halo rows onto their owners (one reverse halo exchange; aimnetcentral_amd/dd.py).
 * `fn` is called on the host from inside aimnet_engine_eval, between launches; the work it enqueues (or performs after a
 * synchronisation) must be ordered on `hip_stream`.  A non-zero return aborts the evaluation with AIMNET_E_INVALID.
 * With AIMNET_STRESS the `stress` output [1][3][3] takes the rank's share of dE/d(strain) UNDIVIDED (there is no cell here): the sum
 * over the ranks divided by the cell volume is the stress.
 * Restrictions: no cell (the cluster is non-periodic), Coulomb NONE or DSF, no caller-supplied lists.
 * owned == NULL switches the mode off. */
#define AIMNET_DD_SUM 0
#define AIMNET_DD_CHARGES 1
#define AIMNET_DD_ROWS 2
typedef int (*aimnet_dd_exchange_fn)(void* ctx, int32_t what, void* dev_ptr, int64_t n_float, void* hip_stream);
int aimnet_engine_set_dd(aimnet_engine* e, const float* owned, aimnet_dd_exchange_fn fn, void* ctx);

/* Bytes of scratch `aimnet_engine_eval` needs for the given problem size. */
size_t aimnet_engine_workspace_bytes(const aimnet_engine* e, int32_t n_atoms, int32_t n_mol, int32_t n_cell,
                                     const aimnet_eval_options* opt);

/* One energy(+forces, +stress) evaluation.  `workspace` must stay alive until the stream has
 * drained; it also holds the intermediates that aimnet_engine_debug_view exposes. */
int aimnet_engine_eval(aimnet_engine* e, const aimnet_inputs* in, const aimnet_eval_options* opt,
                       const aimnet_outputs* out, void* workspace, size_t workspace_bytes, void* hip_stream);

/* Analytic Hessian-vector products H v = d/d eps [dE/dx (x + eps v)] for n_vec directions at once (csrc/hvp.hip: forward-mode
 * tangent sweep through the forward and the backward sweep of the model; specification oracle/aimnet2_analytic.py::evaluate_hvp).
 * `in` / `opt` as for aimnet_engine_eval (opt->flags is ignored; DSF - periodic or not - runs on the neighbour list and needs
 * max_nb_lr > 0; with opt->dftd3 the dispersion block is a 4-point central difference of the D3 gradient alone, h = 4e-3 A along
 * the normalised direction, all 4 n_vec displaced copies in one batch - as the reference treats its PME block, calculator.py:1777-1781).
 * vectors [n_vec, n_atoms, 3] and hv [n_vec, n_atoms, 3] are device fp32; forces [n_atoms, 3] (may be NULL) receives the forces
 * of the same sweep; status [8] as aimnet_outputs.status (a raised overflow flag invalidates hv: grow the rows and call again).
 * Any number of molecules / cells (the reference restricts itself to one structure; the host keeps that contract).  Memory:
 * about 50 KB per (direction, atom) - split the directions into several calls when n_vec * n_atoms is large. */
size_t aimnet_engine_hvp_workspace_bytes(const aimnet_engine* e, int32_t n_atoms, int32_t n_mol, int32_t n_vec,
                                         const aimnet_eval_options* opt);
int aimnet_engine_hvp(aimnet_engine* e, const aimnet_inputs* in, const aimnet_eval_options* opt, const float* vectors,
                      int32_t n_vec, float* hv, float* forces, int32_t* status, void* workspace, size_t workspace_bytes,
                      void* hip_stream);

/* Test hook: byte offset / element count / element size of a named intermediate inside the
 * workspace of the LAST eval (e.g. "nb_idx", "nb_cnt", "pair_geom", "x0", "y1", "q0", "aim",
 * "e_atom", "xbar0").  Returns AIMNET_E_INVALID for unknown names. */
int aimnet_engine_debug_view(const aimnet_engine* e, const char* name, size_t* byte_offset, size_t* n_elem,
                             int32_t* elem_size, int32_t* row_stride);

/* Optional HIP-event profiling on the eval stream (no reference counterpart; the reference documents
 * a manual torch.cuda.synchronize + perf_counter recipe, docs/tutorials/performance.md:183-238).
 * level 0 off, 1 = {GEMM, everything else}, 2 = every kernel family.  profile_read sums the elapsed
 * milliseconds per family since the last reset into ms[AIMNET_PROF_FAMILIES]; family order:
 * nlist, geom, conv_fwd, gemm, pointwise, coulomb, unconcat, conv_bwd, other; with n_families > AIMNET_PROF_FAMILIES,
 * ms[AIMNET_PROF_FAMILIES] receives the number of evaluations the sums cover.  The events themselves cost ~3 % of a
 * 2 ms evaluation: set_profile_sampling(every) records them on every `every`-th evaluation only (default 1). */
#define AIMNET_PROF_FAMILIES 9
int aimnet_engine_set_profiling(aimnet_engine* e, int level);
int aimnet_engine_set_profile_sampling(aimnet_engine* e, int every);
int aimnet_engine_profile_read(aimnet_engine* e, double* ms, int n_families, int reset);

/* Test / tuning hook for the fp32 MFMA GEMM of the MLP stack: C[M,N] = A[M,K] . Bt[N,K]^T with epilogue
 * epi (0 none, 1 +bias, 2 gelu(+bias) with D = gelu', 3 C = acc * D); K % 32 == 0; cfg 0 = automatic tile. */
int aimnet_debug_gemm(int cfg, int epi, const float* A, int lda, const float* Bt, int ldb, int M, int N, int K,
                      const float* bias, float* C, float* D, int ldc, void* hip_stream);

/* Test / tuning hooks for the bf16x3-split MFMA GEMM (csrc/gemm_bf3.hip): same contract as aimnet_debug_gemm, but the weight
 * operand is pre-split into three bf16 planes.  "bf3" layout: per row, K/32 blocks of [plane 0: 32 bf16][plane 1: 32 bf16]
 * [plane 2: 32 bf16] (192 bytes); fp32 value == plane 0 + plane 1 + plane 2 exactly.
 * aimnet_debug_split_bf3: fp32 src [M][ld] (K columns) -> bf3 dst [M][ldd bf16 elements], ldd >= 3 * pad32(K), ldd % 96 == 0;
 * k-blocks (of 32) from neg_from_block on are stored negated (pass a value >= K / 32 for none).
 * aimnet_debug_gemm_bf3: C = epilogue(A . Bt^T), A fp32 [M][lda], Bt3 the bf3 split of Bt [N][K] (ldb in bf16 elements per
 * row); kneg = the number of leading k-steps whose weight blocks are NOT negated (the accumulators change sign there, the
 * epilogue restores it; see "Accumulation bias" in csrc/gemm_bf3.hip), < 0 or >= K / 32: none; cfg 0 = automatic tile. */
int aimnet_debug_split_bf3(const float* src, int ld, int M, int K, void* dst, int ldd, int neg_from_block, void* hip_stream);
int aimnet_debug_gemm_bf3(int cfg, int epi, const float* A, int lda, const void* Bt3, int ldb, int M, int N, int K,
                          const float* bias, float* C, float* D, int ldc, int kneg, void* hip_stream);
/* aimnet_debug_gemm_bf3a (csrc/gemm_bf3a.hip): the same product with the activations pre-split too - A3 = bf3 form of A [M][K]
 * (lda3 bf16 elements per row); out3 != 0: C is written in bf3 form into C3 (ldc3 bf16 elements per row; epilogues 2 and 3
 * only), else as fp32 into C; D (epilogue 2 output / epilogue 3 input) is fp32 [M][ldc].  This kernel accumulates even and odd
 * k-steps in two accumulator sets: alt = 0: Bt3 carries its own signs (result = sum of the sets); alt = 1: Bt3 was split with
 * neg_from_block = -2 (every odd k-block negated; aimnet_debug_split_bf3 accepts -2) and the result is even - odd - the form the
 * engine uses, in which the truncation bias of the bf16 MFMA accumulation cancels without a tunable; alt = 2: the same for an
 * operand that starts on an odd k-block. */
int aimnet_debug_gemm_bf3a(int cfg, int epi, int out3, const void* A3, int lda3, const void* Bt3, int ldb, int M, int N, int K,
                           const float* bias, float* C, void* C3, int ldc3, float* D, int ldc, int alt, void* hip_stream);

/* aimnet_debug_split_h2 / aimnet_debug_gemm_h2 (csrc/gemm_h2.hip): the same product on fp16x2-split operands ("h2" layout: per row,
 * K/32 blocks of [hi: 32 fp16][lo: 32 fp16] = 128 bytes, fp32 value == hi + lo / 4096 to 2^-24; ld* count 16-bit elements,
 * >= 2 * pad32(K), multiples of 64): three matrix instructions per tile and k-step instead of six.  mode of the split: 0 plain,
 * 1 activation form (lo planes of the odd k-blocks negated), 2 weight form (hi planes of the odd k-blocks negated); alt as in
 * aimnet_debug_gemm_bf3a (1 / 2 want A2 in form 1 and Bt2 in form 2; 0 wants both plain); out2 != 0: C is written in h2 form 1. */
int aimnet_debug_split_h2(const float* src, int ld, int M, int K, void* dst, int ldd, int mode, void* hip_stream);
int aimnet_debug_gemm_h2(int cfg, int epi, int out2, const void* A2, int lda2, const void* Bt2, int ldb, int M, int N, int K,
                         const float* bias, float* C, void* C2, int ldc2, float* D, int ldc, int alt, void* hip_stream);

/* aimnet_engine_debug_mlp_sweep (csrc/gemm_chain.hip against csrc/gemm_h2.hip): ONE sweep of the MLP of `pass` on fp16x2-split
 * activations, as the evaluation runs it - `chain` != 0: the single launch of gemm_chain.hip (a block owns 48 rows and the full width
 * of every layer; hidden activations stay in LDS), 0: one launch per layer.  Forward (`backward` == 0): x2 = the input rows in h2 form 1
 * [M][2 * k_in]; H[l] / D[l] (host arrays of device pointers, one per layer): the layer outputs (hidden ones in h2 form - the chain does
 * not write them -, the last one fp32 [M][k_out], or h2 when `flag` != 0) and GELU' (fp32 [M][k_out]; NULL entries allowed for a linear last
 * layer).  Backward: x2 = zb[0] holds the adjoint of the last pre-activation in h2 form and is OVERWRITTEN; zb[0] / zb[1] are the two
 * ping-pong buffers ([M][2 * widest layer] 16-bit elements each); D[l] are inputs; `flag` != 0 forms only the conv columns 256.. of
 * xbar (pass 0); *which = index of the buffer that holds xbar (fp32 [M][k_in]) afterwards.  Test / measurement hook. */
int aimnet_engine_debug_mlp_sweep(aimnet_engine* engine, int pass, int backward, int chain, int flag, const void* x2, int M,
                                  const int32_t* numbers, float* const* H, float* const* D, float* const* zb, int* which,
                                  void* hip_stream);

/* Engine switches for A/B and parity runs (all have an AIMNET_* environment twin read at create time):
 *   "conv_xe"       1 (default): reverse-pair form of the conv backward for systems above the split threshold, 0: combined form
 *   "gemm_bf3"      1 (default): MLP GEMMs of batches above 256 rows with bf16x3-split operands on the bf16 matrix pipe
 *                   (csrc/gemm_bf3.hip: fp32 == three bf16 planes exactly, six products, fp32 accumulation), 2: for every batch
 *                   size, 0: the exact-fp32 MFMA kernels of csrc/gemm.hip everywhere
 *   "gemm_presplit" 1 (default): wherever the split GEMMs run, every GEMM activation operand is kept in the
 *                   split "bf3" form in memory (written by its producer; csrc/gemm_bf3a.hip streams both operands by DMA), 0: fp32
 *                   activations split inside the GEMM's main loop (csrc/gemm_bf3.hip)
 *   "gemm_h2"       1 (default): wherever the activations are pre-split, they and the weights take the fp16x2-split form "h2"
 *                   (csrc/gemm_h2.hip: fp32 == hi + lo / 4096 to 2^-24, three matrix instructions per tile and k-step instead of six,
 *                   4 instead of 6 bytes per element; rms error below the bf16x3 form's, profiles/r5_gemm_h2.md), 0: the bf16x3 form
 *                   (csrc/gemm_bf3a.hip).  fp16 holds |x| < 65504: weights beyond it switch the option off at create time,
 *                   activations beyond it surface as non-finite outputs (the Python layer then repeats the call with 0 and stays there)
 *   "gemm_chain"    1 (default): with fp16x2-split activations every MLP sweep (forward or backward, one per pass) whose layer sizes
 *                   match an instantiated shape is ONE launch (csrc/gemm_chain.hip: a block owns 16 / 32 / 48 rows and the full width
 *                   of every layer, hidden activations stay in LDS, weights stream L2 -> registers in a packed fragment order); 0: one
 *                   launch per layer (csrc/gemm_h2.hip).  Bitwise-identical results (tests/test_gpu_chain.py)
 *   "head_fused"    1 (default): with pre-split activations the energy head 256 -> 128 -> 128 -> 1 runs forward and backward in one
 *                   launch (csrc/gemm_head.hip), 0: four GEMM launches + the last-layer rider
 *   "prep_fused"    1 (default): periodic batches of <= 4 096 atoms / 64 systems prepare their cell grid (status zeroing, molecule
 *                   offsets + input sanity + species, cell + bin setup, wrapping, counting, scan, fill, per-bin ordering) in ONE
 *                   single-block launch (csrc/nlist.hip, prep_small_kernel), molecules of <= 4 096 atoms their status zeroing +
 *                   molecule offsets + coordinate copy; 0: the separate kernels; identical results
 *   "energy_rides"  1 (default): periodic evaluations with a stress request sum the molecule energies as riders of the two stress
 *                   launches at the end instead of two launches of their own; 0: separate launches; identical results
 *   "status_rides"  1 (default): when the short-range list is the only list built, status[0] / status[2] are reduced by rider blocks
 *                   of the SR-Coulomb launch instead of a launch of their own; 0: separate launch
 *   "setup_rides"   1 (default): periodic batches - the cell + bin-grid setup block rides on the molecule-offset launch; 0: its own launch
 *   "status_owned"  1 (default): with "status_rides", up to 32 768 atoms: the status array is not zeroed in front of the evaluation,
 *                   the rider block stores all eight words; 0: memset + atomics
 *   "sums_whole"    1 (default): up to 16 384 atoms, energy sums riding on the stress launch beside the force gather: one block per
 *                   cell / molecule sums everything, no finish launch; 0: sliced sums + finish launch
 *   "emb_bias"      1 (default): pass 0's first GEMM runs over the conv columns, the embedding block is a per-element bias table
 *   "conv_mfma"     bit 0: conv forward, bit 1: conv backward on the v_mfma_f32_4x4x1_16B_f32 kernels (csrc/conv_mfma.hip)
 *                   instead of the packed-FMA VALU kernels (default 0; systems above the split threshold only)
 *   "split_max"     atoms up to which the 4-waves-per-atom "split" conv kernels are used (default 1024; per engine; < 0 = default)
 *   "p0_moments"    0: generic row-gather conv kernels in pass 0 instead of the element-moment forward / species-moment backward
 *   "dsf_np_walk" 0: non-periodic systems of >= 1 500 atoms per molecule evaluate DSF over a neighbour matrix (options.max_nb_lr) instead of
 *                    walking their bounding-box cell grid (the default there, as for periodic cells: max_nb_lr may then be 0)
 *   "overlap_coulomb" 1: Coulomb / DFT-D3 pair kernels on a second HIP stream (default 0: measured slower)
 *   "spatial_order" 0: conv kernels walk the atoms in input order instead of cell-list bin order */
int aimnet_engine_set_option(aimnet_engine* e, const char* name, int value);
/* current value of a switch (what a measurement should record instead of guessing from the environment) */
int aimnet_engine_get_option(const aimnet_engine* e, const char* name, int* value);

/* Test hook: lane layout of v_mfma_f32_4x4x1_16B_f32 as the conv kernels assume it.  out: f32[64][4][64] (device),
 * out[lb][r][l] = VGPR r, lane l of D = A x B with A[l] = l + 1 and B = one-hot(lb), C = 0. */
/* aimnet_debug_pme_recip (csrc/pme.hip): the reciprocal-space part of AIMNET_COULOMB_PME alone, for ONE system - what
 * particle_mesh_ewald's mesh half computes (aimnet/modules/lr.py:752-775) - with unit prefactor: device xw [n][3] (any image of
 * the atoms), q [n], order [n] (device, a permutation: the sequence in which the charge assignment groups the atoms; NULL = as
 * given - the results do not depend on it, bit for bit), cell [9] row vectors; ACCUMULATES onto device e_atom [n] (double: q_i phi_i), qbar [n] (2 phi_i), fgrad [n][3]
 * (2 q_i grad phi_i), virial_atom [n][9] (per-atom shares of 2 dE/d eps; the mesh term on atom 0) - zero them first.  host_info[8]:
 * alpha, rc, mesh[3], mesh points needed (> max_mesh: nothing was computed), phi_bg.  Allocates its own scratch; synchronises. */
int aimnet_debug_pme_recip(const float* xw, const float* q, const int* order, const float* cell, float total_charge, int n_atoms,
                           float accuracy,
                           int max_mesh, double* e_atom, float* qbar, float* fgrad, float* virial_atom, double* host_info,
                           void* hip_stream);
int aimnet_debug_mfma4_probe(float* out, void* hip_stream);

/* Stand-alone neighbour list with the nvalchemiops contract.  nbmat [n_atoms, max_nb] int32 is
 * filled with `fill_value` beyond each row's count; shifts [n_atoms, max_nb, 3] int32 may be NULL
 * when cell == NULL; num_nb [n_atoms] int32; status [2] = {max count, overflow flag}. */
int aimnet_neighbor_list(const float* coord, const int32_t* mol_idx, int32_t n_atoms, int32_t n_mol,
                         const float* cell, int32_t n_cell, const int32_t pbc[3], float cutoff, int32_t max_nb,
                         int32_t fill_value, int32_t* nbmat, int32_t* shifts, int32_t* num_nb, int32_t* status,
                         float* coord_wrapped, void* workspace, size_t workspace_bytes, void* hip_stream);
size_t aimnet_neighbor_list_workspace_bytes(int32_t n_atoms, int32_t n_mol, int32_t max_nb);

/* conv_sv_2d_sp forward: out[b,a,g,0:4] = sum_{m: idx[b,m] < B-1} a[idx[b,m],a,g] * g[b,m,g,0:4],
 * rows packed real-first, last row (b = B-1) is padding and comes out zero.
 * a [B,A,G] f32, idx [B,M] i32, g [B,M,G,4] f32, out [B,A,G,4] f32. */
int aimnet_conv_sv_2d_sp_fwd(const float* a, const int32_t* idx, const float* g, float* out, int32_t B, int32_t A,
                             int32_t G, int32_t M, void* hip_stream);
/* backward: grad_a[j,a,g] = sum_{(b,m): idx[b,m]=j} <grad_out[b,a,g,:], g[b,m,g,:]>   (zero-initialised here)
 *           grad_g[b,m,g,:] = sum_a a[idx[b,m],a,g] * grad_out[b,a,g,:]  (zero for padded slots) */
int aimnet_conv_sv_2d_sp_bwd(const float* grad_out, const float* a, const int32_t* idx, const float* g,
                             float* grad_a, float* grad_g, int32_t B, int32_t A, int32_t G, int32_t M,
                             void* hip_stream);

/* torch.ops.aimnet.conv_sv_2d_sp_bwd_bwd (conv_sv_2d_sp_wp.py:342-446): cotangents grad2_a [B,A,G] of grad_a and grad2_g
 * [B,M,G,4] of grad_g -> grad_grad_out [B,A,G,4], grad_a_double [B,A,G], grad_g_double [B,M,G,4] (all caller-allocated). */
int aimnet_conv_sv_2d_sp_bwd_bwd(const float* grad_out, const float* grad2_a, const float* grad2_g, const float* a,
                                 const int32_t* idx, const float* g, float* grad_grad_out, float* grad_a_double,
                                 float* grad_g_double, int32_t B, int32_t A, int32_t G, int32_t M, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* AIMNET_HIP_H */
