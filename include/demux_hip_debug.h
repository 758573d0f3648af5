/*
 * demux_hip_debug.h -- test, measurement and tuning surface of libdemux_hip.so.
 *
 * Not what a front-end binds (that is demux_hip.h): the switches that pin the choices the library otherwise
 * makes by itself (E-step level, schedule and form, M-step form), the read-outs of the device-side controller, the emulated
 * multi-GPU wire and the device self-tests of the float32 building blocks.  Used by tests/, bench.py and scripts/; every
 * default is what a production call runs.  Same conventions as demux_hip.h (status codes, caller-owned host arrays).
 */
#ifndef DEMUX_HIP_DEBUG_H
#define DEMUX_HIP_DEBUG_H

#include "demux_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Number of (variant, genotype) sums the last exact-mode M-step redid in the reference's order (instrumentation). */
int dmx_get_redo_count(dmx_ctx *ctx, int64_t *count);

/* Tile-major form of the M-step (kernels.h: MTileArgs).  Possible when the exact additions are off, G <= 64 and
 * contribution_power > 0: the M-step records are kept once more, sorted by (tile of <= 128 variants, barcode), +8 bytes per
 * call, and the sums are formed in 64-bit FIXED POINT (every contribution, a float32 in [0, 1], added as the integer
 * rint(c 2^s), s = 50 for all but the hottest tiles): independent of the order of the additions, so bit-reproducible from run
 * to run like the reference's np.bincount (utils.py:35-36), exact for contributions of 2^-27 and more, and within one float32
 * ulp + n 2^-(s + 1) (n calls of the variant) of the reference's float64 sum in general.  Building the records is a sort of
 * the calls (2.6 ms on 200k x 100k x 64, where an M-step then takes 0.34 instead of 0.70 ms), so
 *   1 (default) builds them at the first M-step that has 8 or more M-steps still to come - in the running dmx_em /
 *     dmx_run_iterations call, or announced with dmx_set_msteps_expected -, or when the resident problem has seen 8; where the
 *     incremental M-step applies (dmx_set_mstep_incremental: its full passes are made by the work items with the same fixed-point
 *     arithmetic, dmx_get_mstep_form = 3) only once the device's count of full passes, read at the 4th / 16th / 64th M-step,
 *     says that they keep coming;
 *   2 at the first M-step;  0 never (the float64 work-item form).
 * dmx_set_msteps_expected: a hint - the caller will run about n more M-steps on the resident problem (a front-end that drives
 * the iterations call by call, a benchmark that warms up first); counted down as M-steps run.
 * dmx_get_mstep_tiles_info: whether the records exist and the host wall time their build took. */
int dmx_set_mstep_tiles(dmx_ctx *ctx, int enable);
int dmx_get_mstep_tiles_info(dmx_ctx *ctx, int32_t *built, double *build_ms);
/* form of the last M-step launch: 0 none yet, 1 work items (float64 partial sums), 2 tiles, 3 work items adding the tile-major form's
 * integers under the incremental M-step (the dense regime's kernel may still have taken any of them) */
int dmx_get_mstep_form(dmx_ctx *ctx, int32_t *form);

/* Worst case of the guarded mode.  With the fast pass taking F, the exact kernel over every barcode E, and a fraction f of the
 * barcodes queued, a guarded E-step costs F + f E: more than the exact mode's E once f > 1 - F / E, and (F + E) / E on a
 * workload where nothing can be proven.  F / E depends on the workload (0.56 at 200k x 100k x 64, 0.34 at 1M x 650k x 128 with
 * doublets, above 1 on small problems), so the two passes are TIMED on the device (wall-clock stamps between the launches)
 * and - adaptive = 1, the default - the E-step that follows one of the same resident problem and option
 * table for which F + f E > E runs DIRECT: the fast kernels stand back and the exact kernel walks every barcode (results then
 * bit-identical to the reference's on all of them), counting what the guard would have queued, so that the fast pass
 * returns when it pays again (3 % of hysteresis).  Decided on the device between two E-steps, no host synchronisation.  E is
 * first estimated from the redo's time over its share of the barcodes (only when that share is at least 5 %), then measured
 * by the first direct E-step.  An iterated run is then never slower than the exact mode by more than one mispredicted
 * E-step plus the fast kernels' launches standing back (~10 us per E-step); a single E-step (predict_posteriors) has no
 * history and runs the fast pass + redo.  adaptive = 0: always the fast pass + redo.
 * dmx_get_guard_direct: whether the last guarded E-step ran direct, how many did since dmx_reset_timings, the number of
 * barcodes the last one queued (direct: would have queued), and the device's own timings: the fast pass over all barcodes
 * and the exact kernel over all barcodes in ms (0: not known yet; negative: the estimate, not yet measured by a direct
 * E-step); any pointer may be NULL. */
int dmx_set_guard_adaptive(dmx_ctx *ctx, int adaptive);
/* The COARSE pass of the guarded mode (csrc/kernels.hip: k_estep_tiled_coarse).  For singlet runs of 33..64 genotypes under the
 * tile-major schedule, an E-step whose logits nobody can read - every E-step of a dmx_em / dmx_run_iterations call but the
 * last - may read the genotype table as binary16 (half the bytes of every row gather; relative error 2^-11 per term, priced
 * per call by the guard): posteriors of the barcodes it keeps are proven within the contract exactly as in the fine pass,
 * the others are redone by the exact kernel; the LOGITS of such an E-step are only within the guard's bound D (0.2 at 400
 * calls per barcode) of the reference's, which is why the last E-step of a call - the one whose logits dmx_get_logits /
 * dmx_get_block / dmx_em return - never takes it.  Coarse pass, fine pass or the direct form: chosen per E-step on the
 * device from the measured times of the passes and the fractions both guards flag (both are evaluated whichever pass runs).
 * coarse = 0: never (the guarded mode of round 4).  Default 1.  coarse = 2: admissible for EVERY E-step, the last one of a call and
 * dmx_estep included - their logits then carry the bound D (tests and measurements).  dmx_get_guard_levels: level of the last guarded E-step
 * (0 coarse, 1 fine, 2 direct; -1: none), E-steps that took the coarse pass since dmx_reset_timings, barcodes the fine / the
 * coarse guard flagged in the last one (-1: not evaluated; the guard of a pass that did not run is shown one barcode in 8: an
 * estimate), and the device's timings of the three passes over all barcodes
 * in ms (0: not run yet; exact: negative while it is an estimate).  Any pointer may be NULL. */
int dmx_set_coarse_pass(dmx_ctx *ctx, int coarse);

/* Incremental M-step (csrc/kernels.h: MIncrArgs).  The tile-major M-step adds integers, so its sums can be updated exactly: once a
 * full pass has left them on the device, an M-step visits only the barcodes whose posteriors changed where it matters (a posterior
 * below 2^-26 contributes exactly 0 on the sums' grid) and adds the differences of their new and old contributions - a fraction of a
 * percent of the calls on converged iterations.  The additions are the full pass's, bit for bit; the device falls back to the full
 * pass whenever the changed barcodes hold more than an eighth of the calls, the kept sums are not valid (a new problem, dmx_set_addition,
 * the first M-step of a dmx_em call, another M-step form in between) or the posteriors are dense.  Taken where the tile-major form is
 * (dmx_set_mstep_tiles) on one context that holds all calls of its barcodes.  incremental = 0: every M-step the full pass.  Default 1.
 * incremental = 2 (tests, measurements): the first sums too are built by the delta pass, every barcode against an all-zero row - the
 * full pass's bits from another kernel and another walk of the calls, at 20 x its time.
 * dmx_get_mstep_incremental: full and delta passes since dmx_reset_timings, barcodes the last M-step found changed (-1: it had no
 * valid sums to compare with). */
int dmx_set_mstep_incremental(dmx_ctx *ctx, int incremental);
int dmx_get_mstep_incremental(dmx_ctx *ctx, int64_t *full_passes, int64_t *delta_passes, int64_t *barcodes_changed_last);
int dmx_get_guard_levels(dmx_ctx *ctx, int32_t *level_last, int64_t *coarse_steps, int64_t *flagged_fine_last, int64_t *flagged_coarse_last,
                         double *coarse_pass_ms, double *fine_pass_ms, double *exact_pass_ms);
/* The device times a pass only when it runs, so the time of a pass that is not chosen goes stale - and a pass timed once under other
 * conditions (the first E-step on a device that had idled runs at a fraction of its clocks) would not be chosen again because of that
 * time.  After 64 E-steps in a row on one level the cheapest other admissible level runs once, if its standing price is below twice the
 * running one's (csrc/kernels.h: GUARD_PROBE_STREAK).  dmx_get_guard_probes: such E-steps since dmx_reset_timings, the current streak.
 * dmx_debug_set_pass_ms (testing aid): overwrite the device's times of the coarse / fine / exact pass (ms over all barcodes; negative:
 * leave; exact 0: back to "not measured"). */
int dmx_get_guard_probes(dmx_ctx *ctx, int64_t *probes, int64_t *streak);
int dmx_debug_set_pass_ms(dmx_ctx *ctx, double coarse_pass_ms, double fine_pass_ms, double exact_pass_ms);
int dmx_get_guard_direct(dmx_ctx *ctx, int32_t *last_ran_direct, int64_t *direct_steps, int64_t *would_queue_last, double *fast_pass_ms,
                         double *exact_pass_ms);

/* E-step work distribution.  For singlet runs of 17..128 genotypes on at least 8 192 barcodes with a genotype table
 * of 1 MB or more, the problem upload also builds a tile-major schedule (bins of 8 barcodes with equal numbers of
 * calls, walked variant tile by variant tile, so that the wavefronts of an XCD gather genotype rows from the same
 * ~2 MB of the table at any time: csrc/kernels.hip, k_estep_tiled).  tiled = 1 (default): used where it pays, i.e.
 * in the tolerance mode (DMX_ESTEP_FAST), whose time is the row gathers; the exact mode is bound by its arithmetic
 * and keeps one barcode per wavefront.  tiled = 2: used whenever built; tiled = 0: never.  Results of a given
 * E-step mode are bit-identical under every schedule. */
int dmx_set_estep_schedule(dmx_ctx *ctx, int tiled);

/* Dictionary form of the exact E-step (csrc/estep_dict.hip).  Before the first M-step - predict_posteriors
 * (demux.py:120-156) and iteration 0 of learn_genotypes (demux.py:86-101) - a row of genotype_prob holds a handful of
 * distinct float32 values (the importers write betas from {0, s/2, s, 0.1 x mean}: genotypes.py:147-164).  When every
 * row has at most 8 distinct values (singlet runs) or 4 (doublet runs: at most 10 values of (p1 + p2) * 0.5,
 * demux.py:190) numpy's float32 log is evaluated once per (call, distinct value) instead of once per (call, option);
 * every option still receives the same float32 addends in the same order, so logits and posteriors are bit-identical
 * to the direct form's.  mode = 1 (default): the form is tried whenever the table was computed without a beta
 * addition or supplied by the caller, and used when every row fits; 0: never; 2: tried for every E-step.
 * dmx_get_estep_form reports what the last E-step ran. */
#define DMX_FORM_NONE 0
#define DMX_FORM_DIRECT 1   /* one numpy log per (call, option): k_estep_direct / k_estep_tiled / k_estep_block */
#define DMX_FORM_DICT 2     /* dictionary form, lane-per-option kernel */
#define DMX_FORM_DICT_BLOCK 3   /* dictionary form, workgroup-per-barcode kernel (wide doublet tables) */
#define DMX_FORM_PACKED 4   /* one numpy log per (call, option), several option slots per lane (narrow doublet tables:
                               csrc/estep_packed.hip) */
int dmx_set_estep_dictionary(dmx_ctx *ctx, int mode);
int dmx_get_estep_form(dmx_ctx *ctx, int32_t *form, int32_t *distinct_values);

/* Narrow doublet tables in the exact mode (K = G (G + 1) / 2 options that fill a power-of-two lane group badly: K = 36
 * takes 36 of 64 lanes): lane groups of 8 / 16 / 32 lanes with 3 or 5 option slots per lane (csrc/estep_packed.hip).
 * A barcode's calls are added in order, so with A slots per lane its walk is A times longer; the barcodes with more
 * calls than a third of what a SIMD gets on average therefore take 64-lane wavefronts inside the same launch.
 * mode = 1 (default): used where the shape wastes fewer slots than the direct form and at most an eighth of the
 * barcodes are such long ones; 0: never; 2: every barcode on packed lane groups; 3: the split wherever the shape exists.
 * Bit-identical results. */
int dmx_set_estep_packing(dmx_ctx *ctx, int mode);

/* M-step loads (G <= 64).  wide = 0 (default): 32-bit buffer offsets wherever the tables allow (posterior table below
 * 4 GiB, fewer than 2^24 barcodes), 64-bit addresses otherwise.  wide = 1: always 64-bit addresses - the form the
 * largest problems run, selectable so that it can be exercised at any size.  Results are bit-identical. */
int dmx_set_mstep_wide_addresses(dmx_ctx *ctx, int wide);

/* Emulated wire, for MEASURING what of the exchange a schedule leaves exposed on a box with one GPU: this context behaves
 * as rank `rank` of `nranks` - variant slices, padded tables, sliced P-step, every kernel and copy of the real exchange -
 * but the three collectives move nothing between processes: this rank's block is copied to where the collective would leave
 * it, and the stream is then held by a one-wavefront kernel for the modelled wire time, latency_us + block bytes /
 * link_gbytes_per_s per collective (a direct exchange on a fully connected xGMI node: one block per peer link and
 * direction, all links at once).  The other ranks contribute nothing - their slices of genotype_prob keep the table
 * without addition - so the numbers such a run produces are not an EM of any experiment; its TIMINGS are those of one
 * rank of an nranks-GPU run whose wire behaves as modelled (scripts/emulated_scaling.py). */
int dmx_comm_init_emulated(dmx_ctx *ctx, int rank, int nranks, double link_gbytes_per_s, double latency_us, int reduce_dtype);

/* Variant-sharded M-step, G <= 64: the all-gather of the singlet posteriors is COMPACT - a barcode with one live posterior is described
 * by its 8-byte code, which travels anyway, and its row is rebuilt by the receivers; only the rows of the barcodes with several live
 * posteriors travel, in a list of at most capacity_rows per rank (rows_pad / 4; DEMUXALOT_AMD_EXCHANGE_COMPACT=<rows> sets it, =0
 * switches the compact form off).  Every rank reads every rank's count behind the all-gather - the exchange's one host
 * synchronisation - and all fall back to the all-gather of the whole table when a list overflowed.  The additions keep their bits
 * (a posterior that is not live contributes exactly +0).  E-steps exchanged compactly / that fell back, since the context was created. */
int dmx_get_exchange_compact(dmx_ctx *ctx, int64_t *taken, int64_t *overflows, int64_t *capacity_rows);
/* The same for the all-gather of genotype_prob behind the sliced P-step: a rank lists the rows of its slice that changed since it sent
 * them (27 % at the second EM iteration of 200k x 100k x 64, 1 % at the fifth), the receivers - whose copies are what was sent last -
 * write them; capacity slice_rows / 4, the whole slices when a list overflowed or the table was written by somebody else
 * (dmx_set_probs, the first P-step of a layout).  Bit-identical tables.  P-steps exchanged compactly / that fell back. */
int dmx_get_exchange_compact_table(dmx_ctx *ctx, int64_t *taken, int64_t *overflows, int64_t *capacity_rows);

/* ------------------------------------------------------------------------- *
 * Device self-tests of the float32 building blocks (used by tests/ on the GPU box):
 * the device restatements of numpy's float32 log / exp and of scipy's row softmax.
 * ------------------------------------------------------------------------- */
int dmx_test_logf(dmx_ctx *ctx, const float *in, float *out, int64_t n);
/* the form the E-step kernels inline: positive finite arguments only, range-restricted division */
int dmx_test_logf_hot(dmx_ctx *ctx, const float *in, float *out, int64_t n);
int dmx_test_expf(dmx_ctx *ctx, const float *in, float *out, int64_t n);
/* the hardware log2 (v_log_f32) the tolerance / guarded E-step modes take of a product's mantissa */
int dmx_test_log2_hw(dmx_ctx *ctx, const float *in, float *out, int64_t n);
int dmx_test_softmax(dmx_ctx *ctx, const float *in, float *out, int64_t rows, int64_t cols);

#ifdef __cplusplus
}
#endif
#endif /* DEMUX_HIP_DEBUG_H */
