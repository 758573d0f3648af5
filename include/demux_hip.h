/*
 * demux_hip.h -- C ABI of libdemux_hip.so, the MI355X (gfx950) implementation of
 * demuxalot's Demultiplexer EM hot path.
 *
 * The reference (arogozhnikov/demuxalot, pure Python/numpy) has no FFI boundary of
 * its own: its boundary is the Python call surface of demuxalot/demux.py.  The
 * Python host layer in demuxalot_amd/ keeps that surface verbatim and binds the
 * entry points below with ctypes (see INTEGRATION.md for the stub a maintainer of
 * the reference would add).  Each entry point cites the reference code it replaces
 * (paths relative to the reference checkout).
 *
 * Conventions
 *   - plain pointers and sizes only; all host arrays are C-contiguous, caller-owned;
 *     the library owns every device allocation.
 *   - every function returns 0 on success and a negative dmx_status otherwise;
 *     dmx_last_error() gives the message of the last failure on the calling thread.
 *   - a dmx_ctx is bound to one GPU and one HIP stream; calls on one ctx must not
 *     overlap in time (the reference's caller is single-threaded too); different
 *     ctxs are independent.
 *   - this header is the surface a front-end binds; the tuning switches, controller read-outs and device
 *     self-tests the tests and bench.py use are declared in demux_hip_debug.h.
 *   - "options" are the K posterior columns: K = G without doublets, G(G+1)/2 with
 *     (singlets first, then pairs g1<g2 row-major: demuxalot/demux.py:175-191).
 */
#ifndef DEMUX_HIP_H
#define DEMUX_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dmx_ctx dmx_ctx;

typedef enum {
    DMX_OK = 0,
    DMX_ERR_INVALID = -1,   /* bad argument / call order            */
    DMX_ERR_HIP = -2,       /* a HIP runtime call failed            */
    DMX_ERR_NO_DEVICE = -3, /* no usable GPU                        */
    DMX_ERR_RCCL = -4,      /* RCCL missing or a collective failed  */
    DMX_ERR_UNSUPPORTED = -5
} dmx_status;

/* dtype tags for optional float inputs */
#define DMX_F32 0
#define DMX_F64 1

/* timing slots of dmx_get_timings */
/* DMX_T_ALLREDUCE: the collectives of the multi-GPU exchange (reduce-scatter + all-gather, or the all-reduce) */
enum { DMX_T_PSTEP = 0, DMX_T_ESTEP = 1, DMX_T_MSTEP = 2, DMX_T_MCOMBINE = 3, DMX_T_ALLREDUCE = 4, DMX_T_COUNT = 5 };

const char *dmx_last_error(void);
const char *dmx_version(void);

/* Number of visible GPUs (does not initialise a device). */
int dmx_device_count(int *count);

int dmx_create(int device, dmx_ctx **out);
int dmx_destroy(dmx_ctx *ctx);
int dmx_synchronize(dmx_ctx *ctx);

/* ------------------------------------------------------------------------- *
 * Host-side repack (no GPU involved).
 * Replaces Demultiplexer.pack_calls' variant matching + molecule_calls2barcode_calls
 * (demuxalot/demux.py:276-300, 332-365): every molecule call is matched to the variant
 * with the same (chromosome, position, base); unmatched calls are dropped; the rest are
 * reduced to unique (variant, barcode) pairs sorted by variant then barcode, whose
 * p_base_wrong is the float32 product of the members in input order.
 *
 *   var_*        [n_variants]   key of variant row i (chromosome as a caller-side index)
 *   call_*       [n_calls]      molecule calls in the reference's order
 *   call_variant [n_calls]      nullable; matched variant row of every input call, -1 if none
 *   out_*        caller-allocated, capacity n_calls; *n_unique entries are written
 *   mol_per_variant [n_variants] number of matched molecule calls per variant
 *                (np.bincount(molecule_calls['variant_id']) of demux.py:381)
 * ------------------------------------------------------------------------- */
int dmx_pack_calls_host(int64_t n_variants, const int32_t *var_chrom, const int32_t *var_pos,
                        const uint8_t *var_base, int64_t n_calls, const int32_t *call_chrom,
                        const int32_t *call_pos, const uint8_t *call_base, const int32_t *call_cb,
                        const float *call_p, int32_t *call_variant, int64_t *n_matched,
                        int64_t *n_unique, int32_t *out_variant, int32_t *out_cb, float *out_p, int64_t *out_count,
                        int64_t *mol_per_variant);

/* 64-bit content hash of a host buffer, computed on `threads` host threads (0: up to 16) at memory bandwidth.  What the
 * Python front-end keys the resident packed problem with: the reference re-packs on every call (demux.py:303), so the
 * problem of an earlier call is reused only when EVERY record of every container hashes as it did. */
int dmx_hash_host(const void *data, int64_t bytes, int32_t threads, uint64_t *hash_out);

/* ------------------------------------------------------------------------- *
 * Problem upload.  The calls come as the reference's `barcode_calls` columns
 * (variant_id, compressed_cb, p_base_wrong; demux.py:290-300), in any order; the
 * library derives a barcode-major CSR (E-step) and a variant-major CSC (M-step),
 * both stable with respect to the given order so that float64 sums run in the
 * reference's bincount order.  v2snp is genotypes.get_snp_ids_for_variants()
 * (demuxalot/genotypes.py:56-66).
 * ------------------------------------------------------------------------- */
int dmx_set_problem(dmx_ctx *ctx, int64_t n_barcodes, int64_t n_variants, int32_t n_genotypes,
                    int64_t n_calls, const int32_t *variant_id, const int32_t *compressed_cb,
                    const float *p_base_wrong, const int32_t *v2snp);

/* Device pack: dmx_pack_calls_host's matching + de-duplication done on the GPU and installed
 * directly as the resident problem (as dmx_set_problem would with the unique calls), without a
 * host round trip.  mol_per_variant [n_variants] as in dmx_pack_calls_host.  The unique calls can be
 * read back with dmx_get_packed_calls (variant-major; arrays of *n_unique entries, each nullable). */
int dmx_pack_and_set_problem(dmx_ctx *ctx, int64_t n_barcodes, int64_t n_variants, int32_t n_genotypes,
                             const int32_t *var_chrom, const int32_t *var_pos, const uint8_t *var_base,
                             const int32_t *v2snp, int64_t n_calls, const int32_t *call_chrom,
                             const int32_t *call_pos, const uint8_t *call_base, const int32_t *call_cb,
                             const float *call_p, int64_t *n_matched, int64_t *n_unique,
                             int64_t *mol_per_variant);
int dmx_get_packed_calls(dmx_ctx *ctx, int32_t *variant_id, int32_t *compressed_cb, float *p_base_wrong,
                         int64_t *barcode_variant_count);

/* The same, fed with the reference's containers as they are (one per chromosome: CompressedSNPCalls.snp_calls
 * [:n_snp_calls] and .molecules[:n_molecules], demuxalot/snp_counter.py:77-139): the packed numpy records are
 * uploaded raw and taken apart on the GPU, replacing the per-field flattening of demux.py:332-358.
 *   snp_calls record (13 bytes, numpy packed): int32 molecule_index, int32 snp_position, uint8 base_index,
 *                                             float32 p_base_wrong
 *   molecules record (12 bytes):              int32 compressed_cb, int32 compressed_ub, float32 p_group_misaligned
 * `chrom` is the chromosome's index in the numbering used by var_chrom.  Call order = container order, then
 * record order (the order demux.py:339-358 writes molecule_calls in). */
typedef struct dmx_call_container {
    const void *snp_calls;
    int64_t n_snp_calls;
    const void *molecules;
    int64_t n_molecules;
    int32_t chrom;
} dmx_call_container;
int dmx_pack_containers_and_set_problem(dmx_ctx *ctx, int64_t n_barcodes, int64_t n_variants, int32_t n_genotypes,
                                        const int32_t *var_chrom, const int32_t *var_pos, const uint8_t *var_base,
                                        const int32_t *v2snp, const dmx_call_container *containers,
                                        int32_t n_containers, int64_t *n_matched, int64_t *n_unique,
                                        int64_t *mol_per_variant);

/* The same in two steps, so that the upload of the containers (which needs nothing of the genotypes) can run while the
 * caller is still turning var2varid into the variant key arrays: dmx_stage_containers uploads the records and takes
 * them apart (`chrom` = any provisional numbering, e.g. the container's position in the list; the resident problem is
 * not touched), dmx_pack_staged_and_set_problem does the rest; chrom_of_container[k] = var_chrom's number of provisional
 * chromosome k, or -1 when no variant lies on it (staged calls there are an error: demux.py:339-341, 359), NULL = the
 * numbers are final already. */
int dmx_stage_containers(dmx_ctx *ctx, const dmx_call_container *containers, int32_t n_containers);
int dmx_pack_staged_and_set_problem(dmx_ctx *ctx, int64_t n_barcodes, int64_t n_variants, int32_t n_genotypes,
                                    const int32_t *var_chrom, const int32_t *var_pos, const uint8_t *var_base,
                                    const int32_t *v2snp, const int32_t *chrom_of_container, int32_t n_containers,
                                    int64_t *n_matched, int64_t *n_unique, int64_t *mol_per_variant);

/* Regularised prior betas float32[V*G] (output of pack_calls, demux.py:372-388). */
int dmx_set_betas(dmx_ctx *ctx, const float *prior_betas);

/* The same regularisation computed on the GPU from the raw betas (genotypes.get_betas(), float32[V*G]):
 * compute_prior_betas of demux.py:372-388, including numpy's float32 pairwise row sums and float64
 * per-SNP sums.  With add_data_prior the molecule counts per variant are those left by
 * dmx_pack_and_set_problem, or mol_per_variant[V] when given.  prior_out (nullable) receives the result. */
int dmx_set_prior_betas(dmx_ctx *ctx, const float *raw_betas, double default_prior, int add_data_prior,
                        const int64_t *mol_per_variant, float *prior_out);

/* M-step summation mode.  A variant with more calls than the work-item length (1024 .. 16384, chosen from the
 * problem size) is summed by several wavefronts, and adding
 * their float64 partial sums is not the reference's single left-to-right float64 sum (np.bincount,
 * utils.py:35-36): the two can differ in the last bits, which changes the float32 result when the total sits on a
 * float32 rounding boundary (seen on ~4e-5 of the [V, G] entries of a stress case, always by one float32 ulp).
 * exact != 0 (default): such sums are detected (error bound of either order versus the distance to the
 * boundary) and redone in the reference's order -- additions bit-identical to the reference for any input
 * (+3.6 % per EM iteration on the 200k x 100k x 64 workload).  exact == 0: accept the combined sum.  Single-item
 * variants, and everything in the E-step, are bit-identical in both modes. */
int dmx_set_exact_additions(dmx_ctx *ctx, int exact);
/* The tile-major M-step (csrc/kernels.h: MTileArgs; switches and the long description in demux_hip_debug.h) keeps the M-step
 * records once more, sorted by (tile of <= 128 variants, barcode), and sums in 64-bit fixed point: order-independent, bit-
 * reproducible, within one float32 ulp + n 2^-51 of the reference's float64 sum.  The library builds the records when enough M-steps
 * are to come to pay for the sort - in the running dmx_em / dmx_run_iterations call, or as announced here - and, where the incremental
 * M-step applies (whose full passes the work items make with the same fixed-point arithmetic, no records needed), only once full
 * passes keep coming.
 * dmx_set_msteps_expected: a hint - the caller will run about n more M-steps on the resident problem (a front-end that drives the
 * iterations call by call, a benchmark that warms up first); counted down as M-steps run. */
int dmx_set_msteps_expected(dmx_ctx *ctx, int64_t n);

/* E-step arithmetic.
 * DMX_ESTEP_EXACT (default): every term log(p (1 - e) + max(e, 1e-4)) is evaluated with numpy's float32 log
 *   operation for operation and accumulated in float64 in the reference's order: logits and posteriors are
 *   bit-identical to the reference's (demux.py:246-265).
 * DMX_ESTEP_FAST: the contract of the path only (assignments identical, posteriors within 1e-5 on the reference's
 *   test inputs): the terms of 8 consecutive calls are multiplied in float32 and one hardware log2 is taken of the
 *   product's mantissa (exponents summed as integers, mantissa logs in float64).  Deviations from the reference
 *   are of the size of one float32 rounding of the logit (the reference's own logits carry that much rounding
 *   noise); roughly 6x less VALU work per term, which leaves the E-step bound by the genotype-row gather. */
/* DMX_ESTEP_GUARDED: the tolerance-mode arithmetic with the contract GUARANTEED per E-step instead of observed.  The
 *   epilogue of every barcode bounds |logit_fast - logit_reference| for each option k,
 *       D_k = RHO |S_k| (numpy's float32 log is within RHO = 2.73e-7 relative of the true log on [1e-4, 2]: exhaustive,
 *             tests/test_oracle_npsimd.py) + 7e-8 per call (7 float32 roundings of the 8-term product + a 2-ulp hardware
 *             log2 per 8 calls) + the float32 roundings of the logit itself,
 *   and keeps the fast result only when, with D = max_k D_k,
 *       min(p_k, 1 - p_k) (e^{2D} - 1) <= 8e-6 for every option (=> |posterior - reference posterior| <= 1e-5 with
 *       2e-6 left for the float32 evaluation of the softmax on either side; 6e-6 / 4e-6 for rows of more than 1024
 *       options), and
 *       no second logit lies within 2 D of the largest (=> the same argmax);
 *   every other barcode is queued on the device and redone by the exact kernel in the same E-step, so its logits and
 *   posteriors are the reference's bit for bit.  Where the dictionary form applies (exact and faster) it is used
 *   unchanged.  dmx_get_guard_stats counts the barcodes redone.  The workgroup-per-barcode forms (option tables beyond
 *   1024, doublet tables beyond 256) bound |S_k| by |logit_k| + |penalty_k|; with prior logits they run the exact mode. */
#define DMX_ESTEP_EXACT 0
#define DMX_ESTEP_FAST 1
#define DMX_ESTEP_GUARDED 2
int dmx_set_estep_mode(dmx_ctx *ctx, int mode);
/* Barcodes the guarded E-steps computed with the exact kernel (the queued ones, or all of them in an E-step that ran direct:
 * below): in the last E-step, and in all E-steps / out of how many barcode rows since the context was created or
 * dmx_reset_timings (instrumentation; any pointer may be NULL). */
int dmx_get_guard_stats(dmx_ctx *ctx, int64_t *redone_last, int64_t *redone_total, int64_t *rows_total);
/* The reference's learn_genotypes returns the learnt genotypes and the LAST iteration's posteriors - no logits (demux.py:55-66).  A
 * caller that will not read the logits of the last E-step of its dmx_em / dmx_run_iterations calls says so with needed = 0: that E-step
 * is then one "whose logits nobody reads" like the ones before it and may take the coarse pass - its posteriors proven within the
 * contract per barcode like every coarse E-step's.  dmx_em with a logits output pointer keeps them regardless.  After such a call
 * dmx_get_logits / dmx_get_block(DMX_LOGITS) fail (DMX_ERR_INVALID) until an E-step has kept its logits again (dmx_estep, or a call
 * with needed = 1); posteriors, assignments and the reductions are available as ever.  Default 1. */
int dmx_set_logits_needed(dmx_ctx *ctx, int needed);
/* Memory first (default 0).  A problem large enough for the tile-major E-step schedule holds its E-step records three times: barcode-major
 * (16 bytes per call), in the order the bins consume them for the fine pass (16) and as the coarse pass's records (8), built from the second at
 * the first E-step that may take the coarse pass.  With lean = 1 the second copy is released as soon as the third exists, and the dictionary
 * form's row array (4 bytes per call) with it (dmx_get_device_bytes: 67 -> 46.7 bytes per call at 200k x 100k x 64).  The coarse pass is
 * untouched; the E-steps that keep their logits - the last one of a dmx_em call by default, dmx_estep - run the fine pass on the COARSE pass's
 * records (float32 table, float64 sums; the guard prices r = floor / keep and the slot tag in it: 7.8e-7 per call instead of 7e-8), in the time
 * of the fine pass they replace (1.48 ms at that size), the guard and its exact redo as ever; an E-step on the prior table that keeps its
 * logits runs that pass too instead of the dictionary form.  Applies to the resident problem (at once if its coarse records exist, else behind
 * their build) and to those installed afterwards; lean = 0 keeps what is still there.  A communicator attached AFTER the release re-bases the
 * table rows and leaves such a problem without the tile-major schedule altogether (install it again). */
int dmx_set_lean_memory(dmx_ctx *ctx, int lean);
/* Inside the guarded mode the library chooses per E-step, on the device, between the coarse pass (genotype table as binary16, for
 * E-steps whose logits nobody reads), the fine pass and the exact kernel on every barcode, and per M-step between the full and the
 * incremental pass; the switches that pin those choices, the controller's read-outs, the emulated wire and the device self-tests
 * are test and measurement surface: include/demux_hip_debug.h (same library, same conventions). */
/* The prior betas resident on the device (as set by dmx_set_betas or computed by dmx_set_prior_betas), float32[V*G]. */
int dmx_get_prior_betas(dmx_ctx *ctx, float *out);
/* The learnt genotypes of demux.py:65: `genotypes.get_betas() + genotype_addition`, float32[V*G] - the raw betas dmx_set_prior_betas was
 * given (they stay on the device) plus the addition of the last M-step, one float32 addition per element as numpy's, formed on the
 * device: the host neither downloads the addition nor adds 4 V G bytes to it (10 of learn_genotypes' 36 ms at 200k x 100k x 64).
 * DMX_ERR_INVALID after dmx_set_betas (a prior table given as such has no raw betas behind it). */
int dmx_get_learnt_betas(dmx_ctx *ctx, float *out);

/* genotype_addition float32[V*G]; NULL resets it to zero (demux.py:86). */
int dmx_set_addition(dmx_ctx *ctx, const float *addition);

/* P-step: Demultiplexer._compute_probs_from_betas (demux.py:267-274) applied to
 * prior_betas + addition (float32 add, demux.py:90).  clip_lo / clip_hi are
 * float32(p_genotype_clip) and float32(1 - p_genotype_clip).  prob_out (float32[V*G],
 * nullable) receives a copy of the table. */
int dmx_probs_from_betas(dmx_ctx *ctx, float clip_lo, float clip_hi, float *prob_out);

/* The same P-step on caller-supplied float64 betas[V*G] (the public helper Demultiplexer._compute_probs_from_betas
 * accepts any dtype; numpy then divides float64 by float64 and rounds to float32 once): no addition is applied,
 * the resident prior betas are left alone. */
int dmx_probs_from_betas_f64(dmx_ctx *ctx, const double *betas, float clip_lo, float clip_hi, float *prob_out);

/* Direct upload of a genotype_prob table float32[V*G] (for callers of
 * compute_barcode_logits_using_barcode_calls that bring their own table).  Entries must be probabilities:
 * a table with values outside [0, 1] or NaN is refused (DMX_ERR_INVALID). */
int dmx_set_probs(dmx_ctx *ctx, const float *prob);

/* E-step + posterior: Demultiplexer.compute_barcode_logits_using_barcode_calls
 * (demux.py:246-265) followed by scipy softmax (demux.py:101,152).
 *   with_doublets  0: K = G;  1: K = G(G+1)/2
 *   penalties      float32[K] = Demultiplexer._doublet_penalties (demux.py:158-173)
 *   prior_logits   nullable [B*K] added to the logits before the softmax
 *                  (demux.py:97-99); prior_dtype says float32 or float64
 *   logits_out / probs_out  nullable float32[B*K]; NULL keeps the result on the GPU */
int dmx_estep(dmx_ctx *ctx, int with_doublets, const float *penalties, const void *prior_logits,
              int prior_dtype, float *logits_out, float *probs_out);

/* M-step: genotype_addition of demux.py:113-118 from the posteriors of the last
 * dmx_estep (singlet columns only); contribution_power as Demultiplexer.contribution_power.
 * With a communicator attached the per-rank partial sums are all-reduced over RCCL.
 * addition_out nullable float32[V*G]. */
int dmx_mstep(dmx_ctx *ctx, float contribution_power, float *addition_out);

/* Fused EM driver: the loop of staged_genotype_learning (demux.py:86-118) without the
 * dead M-step after the last iteration.  Outputs (all nullable) are those of the last
 * iteration: logits, posteriors and the addition that iteration's E-step used. */
int dmx_em(dmx_ctx *ctx, int n_iterations, float clip_lo, float clip_hi, int with_doublets,
           const float *penalties, const void *prior_logits, int prior_dtype,
           float contribution_power, float *logits_out, float *probs_out, float *addition_out);

/* Enqueues n_iterations x (P-step, E-step + softmax, M-step [+ all-reduce]) on the ctx stream
 * and returns WITHOUT synchronising (pair with dmx_synchronize).  Unlike dmx_em every iteration
 * includes its M-step, the addition is not reset, and no prior logits are applied: this is the
 * steady-state EM iteration the benchmark times.  Options/penalties are those of the last
 * dmx_estep / dmx_em call. */
int dmx_run_iterations(dmx_ctx *ctx, int n_iterations, float clip_lo, float clip_hi, float contribution_power);

/* Copy the device-resident results of the last E-step / M-step. */
int dmx_get_logits(dmx_ctx *ctx, float *logits_out);
int dmx_get_probs(dmx_ctx *ctx, float *probs_out);
int dmx_get_addition(dmx_ctx *ctx, float *addition_out);

/* A sub-block of the logits (what = DMX_LOGITS) or posteriors (DMX_PROBS) of the last E-step: rows
 * [b0, b1) x columns [k0, k1) into out float32[(b1-b0)*(k1-k0)], row-major.  Lets a caller read sampled
 * barcodes or the singlet columns only, without moving the whole [B, K] matrix over PCIe. */
#define DMX_LOGITS 0
#define DMX_PROBS 1
int dmx_get_block(dmx_ctx *ctx, int what, int64_t b0, int64_t b1, int64_t k0, int64_t k1, float *out);

/* Per-barcode reduction of the posterior on the GPU: argmax option and its
 * probability (what users take from the DataFrame: probs.idxmax(axis=1)). */
int dmx_get_assignments(dmx_ctx *ctx, int32_t *best_option, float *best_prob);

/* The reductions users of the reference apply to the posterior DataFrame, done on the GPU so that the [B, K]
 * matrix (4.3 GB per rank at 130k barcodes x 8256 options) never has to cross PCIe:
 *   dmx_get_assignments_above  probs[probs.max(axis=1).gt(threshold)].idxmax(axis=1)
 *                              (examples/2-with-detection-of-new-SNPs.ipynb cell 14, demuxalot/snp_detection.py:166):
 *                              best_option[b] = first arg-max column if its posterior is > threshold, else -1;
 *                              best_prob[b] = the row maximum either way; *n_assigned = rows above the threshold.
 *   dmx_get_top_options        the k (1..4) best options of every barcode, best first, ties to the lower column:
 *                              options int32[B*k] (-1 past the end of a short row), probs float32[B*k].
 *   dmx_get_option_sums        probs.sum(axis=0) (`probs[genotype_names].sum()`, same notebook cells 19/21) as
 *                              float64[K], added in a fixed order (reproducible run to run).
 * All outputs except the sums are nullable. */
int dmx_get_assignments_above(dmx_ctx *ctx, float threshold, int32_t *best_option, float *best_prob,
                              int64_t *n_assigned);
int dmx_get_top_options(dmx_ctx *ctx, int32_t k, int32_t *options, float *probs);
int dmx_get_option_sums(dmx_ctx *ctx, double *sums);

/* ------------------------------------------------------------------------- *
 * The alternative E-step of the reference, Demultiplexer.aggregate_on_snps = True (demux.py:204-244): likelihoods
 * are summed per (barcode, SNP) pair over the MOLECULE calls (not the de-duplicated barcode calls), divided by
 * count ** compensation, passed through log_softmax, mixed with a 1 % "bad SNP" floor (np.logaddexp, float64 from
 * here on), log_softmax again, and summed per barcode.  Logits and posteriors are float64, as the reference's are;
 * the doublet penalties are not applied (the reference computes and then ignores them in this mode).
 *   dmx_set_keep_molecule_calls  before dmx_pack_and_set_problem / dmx_pack_containers_and_set_problem: keep the
 *                                matched molecule calls on the device, grouped by (barcode, SNP)
 *   dmx_set_molecule_calls       the same from caller-supplied matched molecule calls (pack_calls' molecule_calls
 *                                columns variant_id / compressed_cb / p_base_wrong, in their order)
 *   dmx_get_max_pair_count       most molecule calls in one pair (length of the table below, minus one)
 *   dmx_estep_snp                count_pow[c] = c ** compensation_during_computing_barcode_logits as numpy evaluates
 *                                `counts ** 0.5` for an int64 array (the caller builds the table with numpy so that
 *                                the reference's pow() rounding is repeated exactly); prior_logits (nullable) are
 *                                added to the float64 logits before the softmax (demux.py:97-99)
 *   dmx_mstep_f64                demux.py:113-118 on those float64 posteriors (the products and powers are float64 in
 *                                the reference when the posteriors are); single GPU
 * Everything up to the first log_softmax is float32 and bit-identical to numpy; the float64 part agrees with the
 * reference to a few ulps (numpy's float64 exp / log1p are its own SIMD kernels or libm, depending on the host).
 * Up to 8448 options (doublets of 128 genotypes: 8256); beyond 1024 options one workgroup walks one barcode. */
int dmx_set_keep_molecule_calls(dmx_ctx *ctx, int keep);
int dmx_set_molecule_calls(dmx_ctx *ctx, int64_t n, const int32_t *variant_id, const int32_t *compressed_cb,
                           const float *p_base_wrong);
int dmx_get_max_pair_count(dmx_ctx *ctx, int64_t *max_count);
int dmx_estep_snp(dmx_ctx *ctx, int with_doublets, const double *count_pow, int64_t n_count_pow,
                  const void *prior_logits, int prior_dtype, double *logits_out, double *probs_out);
int dmx_mstep_f64(dmx_ctx *ctx, double contribution_power, float *addition_out);
/* The same M-step, returning the UNROUNDED float64 sums float64[V*G] (np.bincount's result before the float32 store,
 * utils.py:35-36): a barcode-sharded aggregate_on_snps run adds the ranks' sums on the host and rounds once
 * (demuxalot_amd/distributed.py: staged_genotype_learning). */
int dmx_mstep_f64_sums(dmx_ctx *ctx, double contribution_power, double *sums_out);

/* ------------------------------------------------------------------------- *
 * Multi-GPU: one ctx per rank, barcodes sharded by the caller (every rank installs the calls of ITS barcodes, all
 * variants, the whole beta table).  E-step rows need nothing from other ranks.  The M-step (demux.py:113-118) sums over
 * the calls of a variant, i.e. over the barcodes of all ranks; it is sharded on VARIANTS (slices cut at SNP boundaries,
 * one per rank):
 *   set-up (dmx_comm_init* with a problem resident, or installing a problem with a communicator attached): the ranks'
 *     variant-major call records are all-gathered once; rank r keeps the calls of its variant slice from the barcodes of
 *     all ranks;
 *   per EM iteration, inside dmx_mstep / dmx_probs_from_betas / dmx_em / dmx_run_iterations:
 *     all-gather of what the M-step reads of a barcode (8-byte posterior code, bitmap, singlet posteriors)
 *     -> rank r sums slice r over ALL barcodes in the reference's order, float64, one rounding: the additions are
 *        bit-identical to a single-GPU run for any number of ranks - nothing is added across ranks - PROVIDED the calls of
 *        every variant ascend by barcode in the caller's arrays (what the reference's np.unique, dmx_pack_* and pack_calls
 *        produce); for calls in another order the n-rank sum takes a variant's calls rank after rank instead of in the
 *        given order: the same float64 terms, a float32 rounding tie at most -
 *     -> P-step (demux.py:267-274) of slice r -> all-gather of the float32 genotype_prob slices.
 * (Round 6: both all-gathers travel COMPACTLY where they can - the posterior rows only of the barcodes with several live posteriors,
 * the others being described by their 8-byte codes; the rows of genotype_prob only where they changed since they were sent -, as
 * capacity-bounded lists with the whole-table all-gather as the fallback, decided alike on every rank from the gathered counts: one host
 * synchronisation per exchange.  Same bits.  DEMUXALOT_AMD_EXCHANGE_COMPACT=0 switches that off; demux_hip_debug.h:
 * dmx_get_exchange_compact.)
 * That exchange moves 4 G + 8 + 8 ceil(G / 64) bytes per barcode OF THE WHOLE JOB.  When that is more than 1.25 x the
 * [V, G] partial sums (many more barcodes than variants: n x 200k-barcode weak scaling), the exchange of the sums is
 * taken instead: M-step on every rank's own barcodes over all variants, reduce-scatter of the float64 / float32 partial
 * sums over the slices, then P-step and all-gather as above (per-rank sums are added: results within a float32 ulp of
 * the single-GPU ones, not bit-identical).  Every rank sees the same sizes and decides alike.
 * DEMUXALOT_AMD_EXCHANGE = variant | reduce_scatter forces either (variant also with a one-rank communicator: tests); = allreduce, or SNPs whose variants are not
 * contiguous in the variant numbering (no slices can be cut): all-reduce of the sums, P-step on every rank.
 * reduce_dtype matters for the exchanges of sums only: DMX_F64 exchanges float64 partial sums and rounds once, DMX_F32
 * halves the bytes.
 * dmx_comm_unique_id fills 128 bytes on rank 0 (ncclGetUniqueId); the caller broadcasts them and every rank calls
 * dmx_comm_init (before or after installing the problem; once per installed problem).
 * Collective calls -- every rank must make them, in the same order: dmx_set_problem / dmx_pack_*_and_set_problem with a
 * communicator attached (or dmx_comm_init* with a problem resident), dmx_probs_from_betas, dmx_mstep (all ranks pass
 * addition_out or none does), dmx_em, dmx_run_iterations, dmx_get_addition.
 * ------------------------------------------------------------------------- */
/* The exchange the resident problem runs (see above): no communicator, or with one attached the M-step sharded on
 * variants / the reduce-scatter of the sums / the all-reduce of the sums (with one rank nothing travels either way). */
#define DMX_EXCHANGE_NONE 0
#define DMX_EXCHANGE_VARIANT 1
#define DMX_EXCHANGE_REDUCE_SCATTER 2
#define DMX_EXCHANGE_ALLREDUCE 3
int dmx_get_exchange_mode(dmx_ctx *ctx, int32_t *mode);

/* Which HIP / RCCL runtime files this process has mapped, one "key=path" per line: hip=... (one line per distinct
 * libamdhip64 - exactly one in a healthy process), rccl_mapped=..., rccl_loaded=<the file dmx_comm_* bound, if any>.
 * RCCL is always taken from the directory of the HIP runtime libdemux_hip.so itself resolved, and dmx_comm_unique_id /
 * dmx_comm_init refuse a process that has two HIP runtimes mapped (e.g. one that imported torch): streams and
 * buffers of one runtime must not be handed to collectives of another.  Environment: DEMUXALOT_AMD_RCCL=<file>,
 * DEMUXALOT_AMD_ALLOW_FOREIGN_RCCL=1. */
int dmx_runtime_info(char *out, int64_t capacity);

/* Host only: the variant slices dmx_comm_init would cut for nranks ranks: cuts int64[nranks + 1] (first variant of
 * every slice, each at the first variant of a SNP), *slice_rows = rows of the longest slice (nullable),
 * *contiguous = 1 when every SNP's variants are contiguous in the numbering (nullable). */
int dmx_exchange_slices(int64_t n_variants, const int32_t *v2snp, int32_t nranks, int64_t *cuts, int64_t *slice_rows,
                        int32_t *contiguous);

#define DMX_UNIQUE_ID_BYTES 128
int dmx_comm_unique_id(void *id_out);
int dmx_comm_init(dmx_ctx *ctx, int rank, int nranks, const void *unique_id, int reduce_dtype);

/* The same exchange over collectives the CALLER provides (MPI, gloo, a test harness running several ranks on one
 * GPU ...) instead of RCCL: the library stages the buffer through pinned host memory and calls `collective` on the
 * ctx stream's host thread; every rank's callback must complete the same collective.  `buf` is a host array:
 *   DMX_COLL_ALL_REDUCE      buf[count]            in-place sum over ranks
 *   DMX_COLL_REDUCE_SCATTER  buf[nranks * count]   on return block `rank` (buf + rank * count) = sum over ranks of
 *                                                   their block `rank`; the other blocks are scratch
 *   DMX_COLL_ALL_GATHER      buf[nranks * count]   on entry block `rank` is filled; on return every block is
 * dtype = DMX_F32 or DMX_F64 (element type of buf).  Return 0, or non-zero to fail the calling dmx_* function. */
enum { DMX_COLL_ALL_REDUCE = 0, DMX_COLL_REDUCE_SCATTER = 1, DMX_COLL_ALL_GATHER = 2 };
typedef int (*dmx_host_collective)(void *user, int op, void *buf, int64_t count, int dtype);
int dmx_comm_init_host(dmx_ctx *ctx, int rank, int nranks, dmx_host_collective collective, void *user, int reduce_dtype);
/* Accumulated kernel time per slot (ms, from HIP events on the ctx stream) and launch
 * counts since the last dmx_reset_timings. Arrays of DMX_T_COUNT entries.
 * The events are recorded only while the phase timers are on (dmx_set_phase_timers(ctx, 1); default off, the launch counts are
 * kept either way): an event record is a barrier packet of its own on the queue, 6 us at each of an EM iteration's four phase
 * boundaries - 2 % of a 1.14 ms iteration that nobody reading no timings should pay (two per boundary, as shipped until round 5:
 * 40 us; consecutive phases share the event between them now).  ms[slot] = -1 when the slot's launches all ran with the timers
 * off (no time was measured: not 0 ms); with the timers on for some of them it is the time of those. */
int dmx_set_phase_timers(dmx_ctx *ctx, int on);
int dmx_get_timings(dmx_ctx *ctx, double *ms, int64_t *launches);
int dmx_reset_timings(dmx_ctx *ctx);

/* Device memory currently held by the ctx, bytes. */
int dmx_device_bytes(dmx_ctx *ctx, int64_t *bytes);

/* Device blocks a context has released (the previous problem's arrays, the temporaries of a repack) stay with it for its
 * next allocations: on this stack a round of hipFree + hipMalloc of a few gigabytes costs ~350 ms, the whole repack of
 * 78.65 M calls 65 ms (csrc/dmx_ctx.h: ctx_malloc).  DEMUXALOT_AMD_CACHE_GB caps the idle bytes per context (default 24;
 * 0: no caching).  dmx_destroy passes a context's blocks on to the contexts created later on the same device (same cap).
 * dmx_trim_cache gives this context's idle blocks and the device's retired ones back to the driver now and reports how
 * many bytes that were. */
int dmx_trim_cache(dmx_ctx *ctx, int64_t *released_bytes);
/* The resident problem of a context (call layouts, tables, results, staged containers) released into its block cache: what
 * a holder does with a context it parks for later (demuxalot_amd/device.py: release_private_context).  The context stays
 * usable; the next problem re-uses the blocks. */
int dmx_release_problem(dmx_ctx *ctx);
/* Everything parked on a device back to the driver: the idle blocks of EVERY live context of this process on it and the
 * retired list.  An allocation that fails for lack of memory does this by itself before it gives up. */
int dmx_trim_device_caches(int device, int64_t *released_bytes);

#ifdef __cplusplus
}
#endif
#endif /* DEMUX_HIP_H */
