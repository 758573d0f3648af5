// dmx_ctx.h -- the context object behind the opaque dmx_ctx of the C ABI, shared by the
// translation units of libdemux_hip.so (dmx_api.cpp, repack_device.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "dmx_internal.h"
#include "kernels.h"

using dmx::fail;

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) return fail(DMX_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

#define DMX_TRY(expr)         \
    do {                      \
        int _s = (expr);      \
        if (_s != 0) return _s; \
    } while (0)

// ------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------
// Phase timers (dmx_get_timings).  A stamp is one recorded event; a span is two of them.  Two phases that follow each other
// inside one API call share the stamp between them: an event record is a barrier packet of its own on the queue, and two of them
// back to back at each of an EM iteration's four phase boundaries were 10 us each - 40 us of a 1.14 ms iteration (round 5,
// scripts/iteration_timeline.sh).  Events release to the device only (hipEventReleaseToDevice): nothing on the host reads what the
// kernels wrote at a phase boundary.
struct TimerStamp {
    hipEvent_t ev = nullptr;
    int refs = 0;
};
typedef std::pair<TimerStamp *, TimerStamp *> TimerSpan;

struct TimerSlot {
    std::vector<TimerSpan> pending;
    double ms = 0.0;
    int64_t launches = 0;
    int64_t timed = 0;  // launches that were bracketed by events (phase timers on)
};

struct dmx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    long long B = 0, V = 0, N = 0, S = 0;
    int G = 0, K = 0;
    bool have_problem = false, have_betas = false, have_probs = false, have_post = false;

    long long *d_pair_ptr = nullptr;
    dmx::CallPair *d_call_pairs = nullptr;
    unsigned *d_call_rows = nullptr;  // table row of every call of d_call_pairs (EstepArgs::call_rows)
    long long n_pairs = 0;
    uint2 *d_csc = nullptr;
    long long n_csc = 0;  // M-step records held: N, or the calls of this rank's variant slice over the barcodes of all ranks
    long long *d_item_start = nullptr;
    int *d_item_len = nullptr;
    long long *d_item_ptr = nullptr;
    int *d_item_variant = nullptr;  // [n_items] variant of every work item (MstepArgs::item_variant)
    int *d_bc_order = nullptr, *d_item_order = nullptr;
    long long n_items = 0;
    // tile-major E-step schedule (repack_device.hip; n_bins == 0: not built)
    long long n_bins = 0;
    int n_tiles = 0, bin_rows_cap = 0;       // rows per bin (<= TILE_R_MAX)
    int *d_bin_rows = nullptr, *d_bin_order = nullptr;
    long long *d_bin_ptr = nullptr;          // [n_bins + 1] first stream group of every bin
    dmx::CallPair *d_tile_stream = nullptr;  // the E-step records once more, in the order the bins consume them
    int *d_v2snp = nullptr, *d_snp_ptr = nullptr, *d_snp_vars = nullptr;
    float *d_prior = nullptr, *d_add = nullptr, *d_prob = nullptr;
    float *d_raw = nullptr;  // the raw betas dmx_set_prior_betas was given (dmx_get_learnt_betas: raw + addition); have_raw
    bool have_raw = false;
    unsigned short *d_prob16 = nullptr;  // d_prob as binary16 at the same row offsets (the coarse pass of the guarded E-step), cap_prob16 values
    size_t cap_prob16 = 0;
    unsigned *d_coarse_stream = nullptr;      // the coarse pass's records (kernels.hip: coarse_walk), cap_coarse_stream dwords; built at the first
    long long *d_coarse_bin_ptr = nullptr;    //   admissible E-step of a problem from the tile-major stream; [n_bins + 1]
    double *d_log2_keep = nullptr;            // [B]
    size_t cap_coarse_stream = 0;
    bool coarse_ready = false;
    bool prob16_valid = false;           // ... and it holds the current d_prob
    int coarse_pass = 1;                 // dmx_set_coarse_pass
    int lean_memory = 0;                 // dmx_set_lean_memory: the tile-major E-step stream goes once the coarse pass's records are built from it
    bool logits_needed = true;           // dmx_set_logits_needed: the last E-step of a dmx_em / dmx_run_iterations call keeps its logits readable
    bool logits_readable = true;         // the last E-step's logits are the fine pass's / the exact kernel's (dmx_get_logits, dmx_get_block)
    float p_clip_lo = 0.0f;              // lower clip of the P-step that produced d_prob (0: a caller's table, dmx_set_probs)
    double *d_add64 = nullptr, *d_partial = nullptr;
    float *d_logits = nullptr, *d_post = nullptr;
    unsigned long long *d_nz = nullptr;
    unsigned long long *d_redo = nullptr;  // (variant, genotype) sums to be redone in order (k_mcombine)
    unsigned *d_n_redo = nullptr;
    size_t cap_redo = 0;
    bool mstep_wide = false;  // dmx_set_mstep_wide_addresses
    // tile-major M-step (kernels.hip: k_mstep_tiles; built on first use by build_mstep_tiles, n_mt == 0: not built / not eligible)
    uint2 *d_mt_stream = nullptr;   // [n_mt_stream] the M-step records once more, sorted by (variant tile, barcode row): x = row | variant in tile << 24
    long long *d_mt_ptr = nullptr;  // [n_mt + 1] first record of every tile
    int *d_mt_first = nullptr;      // [n_mt + 1] first variant of every tile
    int *d_mt_order = nullptr;      // [n_mt] tiles by decreasing number of calls
    int *d_mt_shift = nullptr;      // [n_mt] fixed-point exponent of every tile (MTileArgs::shift)
    unsigned char *d_mt_shift_v = nullptr;  // [V] the same per variant (incremental M-step, fixed-point work-item form; one context's own records only)
    // incremental M-step (kernels.h: MIncrArgs): the tiles' integer sums and the posteriors they were formed from, kept between M-steps
    unsigned long long *d_acc64 = nullptr;  // [V, G]
    float *d_prev_post = nullptr;           // [B, G]
    uint2 *d_prev_first = nullptr;          // [B]
    int *d_incr_list = nullptr;             // [B]
    unsigned char *d_incr_touched = nullptr;  // [V]
    unsigned *d_incr_state = nullptr;       // [2 x IS_WORDS] the state words of this and of the next M-step, alternating
    int incr_parity = 0;
    bool incr_valid = false;                // the device state may be trusted (else the state words are zeroed: a full pass)
    float incr_power = 0.0f;                // contribution power of the sums in d_acc64
    int mstep_incremental = 1;              // dmx_set_mstep_incremental
    long long mstep_incr_launches = 0;      // M-steps that went through the incremental launch sequence
    long long n_mt = 0;
    std::vector<long long> h_col_ptr;  // [V + 1] first M-step record of every variant (host copy made by the repack: the tiles are cut from it)
    long long n_mt_stream = 0;      // records d_mt_stream holds room for (the calls; with the padding calls' slots when built from the barcode-major records)
    int mt_tv = 0;                  // variants per tile at most
    bool mt_tried = false;          // a build was attempted for the resident M-step records
    bool mt_shift_tried = false;    // ... the tile cut for the exponents alone (plan_mstep_shifts)
    int mstep_tiles = 1;            // dmx_set_mstep_tiles: 0 never, 1 when building the records pays, 2 always
    long long msteps_done = 0;      // M-steps run on the resident problem
    long long incr_rows = 0;        // barcode rows the incremental state was allocated for (a variant-sharded rank: those of all ranks)
    unsigned char *d_incr_map = nullptr;  // [incr_rows] flags of the changed barcodes (variant-sharded rank: MIncrArgs::changed_map)
    // variant-sharded rank, incremental M-step: the slice's records once more, BARCODE-major over the rows of all ranks ({variant, bits(1-e)},
    // a row's calls by variant; build_slice_row_index at the first incremental M-step) - the delta pass reads the changed barcodes' calls only
    uint2 *d_slice_rec = nullptr;         // [n_slice_rec]
    long long *d_slice_ptr = nullptr;     // [incr_rows + 1]
    long long n_slice_rec = 0;
    bool slice_index_tried = false;
    bool incr_heavy = false;        // the incremental M-step keeps falling back to full passes on this problem: the tile-major records pay (run_mstep)
    int msteps_ahead = 0;           // M-steps the running dmx_em / dmx_run_iterations call still has to do (0 outside)
    long long msteps_expected = 0;  // dmx_set_msteps_expected: M-steps the caller says it will still run (counted down as they run)
    double mt_build_ms = 0.0;       // host wall time of the last build of the tile-major records
    int mstep_form = 0;             // form of the last M-step launch: 0 none, 1 work items, 2 tiles (dmx_get_mstep_form)
    int *d_sum_plan = nullptr;  // np.sum over a row of K values as a leaf / level plan (dmx_api.cpp: ensure_options)
    size_t cap_sum_plan = 0;
    long long sum_plan_k = -1;
    int sum_plan_values = 0;    // leaves + inner nodes
    int item_calls = 1024;  // work-item length of the resident problem (kernels.h: item_calls_for)
    bool exact_additions = true;  // dmx_set_exact_additions
    int estep_mode = DMX_ESTEP_EXACT;  // dmx_set_estep_mode
    // guarded mode: barcodes queued by the epilogues of the fast kernels for the exact redo (kernels.h: EstepArgs::guard)
    unsigned *d_guard_count = nullptr;  // [GUARD_STATE_WORDS] device state of the guarded mode (kernels.h: GS_*)
    int guard_adaptive = 1;             // dmx_set_guard_adaptive: the device picks coarse pass / fine pass / the exact kernel on every barcode per E-step from its own timings
    int *d_guard_list = nullptr;        // [B] EstepArgs::guard_list
    int *d_guard_sub = nullptr;         // [GUARD_QUEUES x guard_sub_cap] EstepArgs::guard_sub
    unsigned guard_sub_cap = 0;
    long long guard_rows_total = 0;     // barcode rows the guarded kernels have walked since the last reset
    bool guard_ran = false;             // the last E-step evaluated the guard
    int tiled_estep = 1;               // dmx_set_estep_schedule: 0 never, 1 when it pays, 2 whenever the repack built one
    // dictionary form of the E-step (estep_dict.hip): tried when the genotype table was computed without a beta
    // addition (or supplied by the caller), used when every row has few distinct values
    int dict_mode = 1;            // dmx_set_estep_dictionary: 0 never, 1 when the table is a candidate, 2 try always
    bool add_is_zero = true;      // d_add holds zeros (no M-step / dmx_set_addition since the last reset)
    bool dict_candidate = false;  // the current d_prob was computed without an addition, or set by the caller
    int estep_form = 0;           // form of the last E-step: DMX_FORM_*
    int estep_packing = 1;        // dmx_set_estep_packing: 0 never, 1 where it pays, 2 wherever the shape exists
    long long max_row_calls = 0;  // calls of the longest barcode row (device repack)
    int n_simd = 0;               // SIMDs of the device (4 per CU)
    long long long_row_calls[3] = {0, 0, 0};  // calls per SIMD at 8 / 4 / 2 barcodes per wavefront (device repack) ...
    long long n_long_rows[3] = {0, 0, 0};     // ... and how many barcode rows have more
    int dict_distinct = 0;        // most distinct values per row found by the last dictionary build (0: none built)
    float *d_dict = nullptr;             // [prob_rows, DICT_CAP]
    unsigned char *d_codes = nullptr;    // [prob_rows, G]
    unsigned char *d_dtab = nullptr;     // the packed table the kernel reads (estep_dict.hip: DictRow)
    unsigned *d_dict_stat = nullptr;     // [1]
    size_t cap_dict_rows = 0, cap_dtab = 0;
    // split rows of the tolerance / guarded E-step (kernels.h: EstepArgs::segs; dmx_api.cpp: build_row_segments)
    dmx::EstepSegment *d_segs = nullptr;
    int *d_split_first = nullptr;
    double *d_seg_sums = nullptr;
    size_t cap_seg_sums = 0;
    long long n_segs = 0, n_split = 0;
    float nz_floor = 0.0f;     // threshold the current d_nz / d_first were built with
    uint2 *d_first = nullptr;  // [B] {posterior of the lowest live singlet column, count | first live columns} (G <= 64): EstepArgs::first
    unsigned long long *d_dense_calls = nullptr;  // [1] E-step statistic read by the M-step kernels (kernels.h)
    bool dense_stat_valid = false;
    long long cap_bk = 0;
    float *d_pen = nullptr;
    unsigned *d_pairs = nullptr;
    unsigned *d_pair_blocks = nullptr;  // EstepArgs::pair_blocks (doublet runs whose options go to the workgroup-per-barcode forms)
    int n_pair_blocks = 0, cap_pair_blocks = 0;
    int cap_k = 0;
    void *d_prior_logits = nullptr;
    size_t cap_prior = 0;
    int *d_best = nullptr;
    float *d_bestp = nullptr;
    // unique (variant, barcode) calls left by the device pack (variant-major), kept for dmx_get_packed_calls
    int *d_u_variant = nullptr, *d_u_cb = nullptr;
    float *d_u_p = nullptr;
    long long *d_u_count = nullptr;
    long long n_u = 0;
    unsigned long long *d_mol = nullptr;  // matched molecule calls per variant (device pack), for the data prior
    // aggregate_on_snps (snp_aggregate.hip): matched molecule calls grouped by (barcode, SNP)
    bool keep_molecule_calls = false;     // dmx_set_keep_molecule_calls: the device pack leaves them behind
    int *d_mc_variant = nullptr;          // [n_mc] variant row | 0x80000000 on the first call of a pair
    float *d_mc_e = nullptr;              // [n_mc] p_base_wrong
    long long *d_mc_start = nullptr;      // [B + 1] first call of every barcode
    long long n_mc = 0;
    unsigned mc_max_count = 0;            // most molecule calls in one (barcode, SNP) pair
    double *d_logits64 = nullptr, *d_post64 = nullptr;  // float64 results of dmx_estep_snp
    size_t cap_bk64 = 0;
    bool have_post64 = false;
    void *d_scratch = nullptr;  // self tests
    size_t cap_scratch = 0;

    // ---- multi-GPU (dmx_api.cpp: "exchange") ----
    // The [V, G] tables that cross ranks live in a PADDED row layout: the variants are cut into nranks slices at
    // SNP boundaries, every slice padded to slice_rows rows, so that slice r of any such table is the contiguous,
    // equally sized block ncclReduceScatter / ncclAllGather want.  prow(v) = padded row of variant v.  With one
    // rank (or SNPs whose variants are not contiguous: `sliced` false) the layout is the dense one.
    ncclComm_t comm = nullptr;                // RCCL communicator (dmx_comm_init), or
    dmx_host_collective host_coll = nullptr;  // the caller's collectives over host buffers (dmx_comm_init_host)
    void *host_user = nullptr;
    void *h_stage = nullptr;                  // pinned staging of the host collectives
    size_t h_stage_bytes = 0;
    // Emulated wire (dmx_comm_init_emulated): the collectives move nothing between processes - this rank's block is copied
    // where the collective would leave it, then the stream waits for the modelled wire time.  For measuring what of the
    // exchange a schedule leaves exposed on a box with one GPU; the results of such a run are not a real EM.
    bool emulated = false;
    double emu_link_gbps = 50.0, emu_latency_us = 10.0;
    bool in_group = false, group_paid = false;  // coll_group_begin .. coll_group_end: several collectives, one launch
    double group_ns = 0.0;                       // emulated wire: the group's modelled time, held by ONE delay at its end
    bool emu_table_filled = false;            // the other ranks' slices of genotype_prob hold the table without addition
    double emu_ticks_per_ns = 0.1;            // wall-clock ticks of the delay kernel per nanosecond
    bool attached() const { return comm != nullptr || host_coll != nullptr || emulated; }
    int rank = 0, nranks = 1, reduce_dtype = DMX_F64;
    bool sliced = false;            // reduce-scatter + sliced P-step + all-gather (else: all-reduce + replicated P-step)
    bool add_partial = false;       // only this rank's slice of d_add is current (sliced mode, after an M-step)
    long long slice_rows = 0;       // rows per slice of the padded tables
    long long prob_rows = 0;        // rows of d_prob (= nranks * slice_rows when sliced, else V)
    std::vector<long long> cut;     // [nranks + 1] first variant of every slice
    std::vector<int> h_v2snp;       // host copy of v2snp (layout decisions)
    int *d_prow = nullptr;          // [V] padded row of every variant (sliced mode)
    int *d_row_variant = nullptr;   // [prob_rows] ... and back (padding rows: variant 0; the incremental M-step of a rank that exchanges sums)
    // M-step sharded on variants (dmx_api.cpp: shard_mstep_by_variant): d_csc and the work items hold the calls of this rank's
    // variant slice from the barcodes of ALL ranks; the three tables the M-step reads of a barcode are global
    // (row = owner rank * rows_pad + barcode), filled block by block by the ranks' E-steps and all-gathered
    bool mshard = false;
    long long rows_pad = 0, rows_total = 0;  // barcode rows per rank block / of all ranks
    uint2 *d_first_g = nullptr;              // [rows_total] EstepArgs::first of every barcode
    unsigned long long *d_nz_g = nullptr;    // [rows_total, ceil(G / 64)]
    float *d_post_g = nullptr;               // [rows_total, G] singlet posteriors
    bool post_gathered = false;              // the tables hold the last E-step of every rank
    bool emu_post_filled = false;            // emulated wire: the other ranks' blocks were filled once
    // compact exchange of the posterior rows (gather_posteriors; G <= 64): per rank a block of {rows listed, 3 pad, cap x (row, G floats)}
    unsigned *d_post_compact = nullptr;      // [nranks * post_compact_words]
    uint2 *d_post_seen = nullptr;            // [rows_total] the code every row of d_post_g was last rebuilt from (0xFF..: unknown)
    float *d_post_sent = nullptr;            // [B, G] this rank's rows with several live posteriors as the other ranks hold them ...
    unsigned char *d_post_sent_multi = nullptr;  // [B] ... where they do (0: the row was rebuilt from its code since, or never listed)
    unsigned *h_post_counts = nullptr;       // pinned, [nranks + 1]: the lists' lengths, read behind the all-gather, and the sequence number the host polls
    unsigned list_seq = 0;                   // (k_post_counts / wait_counts)
    size_t post_compact_words = 0;           // words per rank block (0: the whole table travels, as until round 6)
    unsigned post_compact_cap = 0;           // rows a block can list at most
    unsigned post_cap_now = 0;               // ... in the coming exchange: twice what the longest list of the last one held (every rank reads every count: the same choice everywhere)
    long long post_compact_taken = 0, post_compact_overflows = 0;  // E-steps exchanged compactly / that fell back to the whole table
    // compact exchange of the genotype table (run_pstep; sliced P-step): the rows of this rank's slice that changed since it sent them
    unsigned *d_prob_list = nullptr;         // [nranks * prob_list_words] {rows listed, 3 pad, cap x (row, G floats)} per rank
    float *d_prob_prev = nullptr;            // [slice_rows, G] this rank's slice as the other ranks hold it
    size_t prob_list_words = 0;              // (0: the whole slices travel)
    unsigned prob_list_cap = 0;
    unsigned prob_cap_now = 0;               // (as post_cap_now)
    bool prob_prev_valid = false;            // d_prob_prev is what every rank holds of this slice
    unsigned *h_prob_counts = nullptr;       // pinned, [nranks + 1]
    long long prob_compact_taken = 0, prob_compact_overflows = 0;
    void *d_exch = nullptr;         // padded send buffer of the reduce-scatter (float64 or float32 partial sums)
    void *d_recv = nullptr;         // this rank's reduced slice
    size_t exch_bytes = 0, recv_bytes = 0;

    // flat call arrays of staged containers (dmx_stage_containers -> dmx_pack_staged_and_set_problem); n_staged < 0: none
    int *st_chrom = nullptr, *st_pos = nullptr, *st_cb = nullptr;
    unsigned char *st_base = nullptr;
    float *st_p = nullptr;
    long long n_staged = -1;
    int64_t bytes = 0;
    TimerSlot timers[DMX_T_COUNT];
    bool phase_timers = false;  // dmx_set_phase_timers: events around the phases (off: the launch counts only)
    std::vector<TimerStamp *> idle_stamps;
    TimerStamp *boundary = nullptr;  // the stamp the last phase ended on, while nothing else has been enqueued behind it (bind() clears it)

    // Device blocks this context has released, kept for its next allocations (ctx_malloc / ctx_free below).
    std::multimap<size_t, void *> idle_blocks;          // capacity -> block
    std::unordered_map<void *, size_t> block_capacity;  // every block handed out through ctx_malloc, idle or not
    size_t idle_bytes = 0;
    std::mutex cache_lock;  // the three above: another context's out-of-memory retry may trim this one's idle blocks
};

// Why the context keeps its device blocks: a second predict / learn call on a context frees the previous problem
// (gigabytes in ~100 blocks) and allocates the next one, and on this stack ONE hipMalloc after such a round of hipFree
// took 350 ms (rocprofv3 --hip-trace of scripts/e2e_breakdown.py: 87 ms for the first pack of 78.65 M calls, 430 ms for
// every later one).  Blocks go back to the context that allocated them and are handed out again to requests of
// (nearly) their size; everything a context launches is ordered on its stream, so a block can be re-used while the
// work of its previous life is still queued.  DEMUXALOT_AMD_CACHE_GB caps the idle bytes per context (default 24; 0 =
// free at once).  dmx_destroy and dmx_trim release them.
size_t ctx_cache_limit();
int ctx_malloc(dmx_ctx *c, void **p, size_t bytes);
void ctx_free(dmx_ctx *c, void *p);
void ctx_trim(dmx_ctx *c, size_t keep_bytes);
void ctx_retire(dmx_ctx *c);  // dmx_destroy: idle blocks to the device's retired list (for contexts created later)
void ctx_register(dmx_ctx *c);    // dmx_create / dmx_destroy: the live contexts of a device, whose idle blocks an
void ctx_unregister(dmx_ctx *c);  // out-of-memory retry anywhere on that device may give back
size_t trim_device_caches(int device);  // idle blocks of every live context + the retired list; returns the bytes freed

template <typename T>
inline int dev_alloc(dmx_ctx *c, T **p, size_t count)
{
    *p = nullptr;
    if (count == 0) count = 1;
    const int rc = ctx_malloc(c, (void **)p, count * sizeof(T));
    if (rc) return rc;
    c->bytes += (int64_t)(count * sizeof(T));
    return 0;
}

template <typename T>
inline void dev_free(dmx_ctx *c, T **p, size_t count)
{
    if (*p) {
        ctx_free(c, (void *)*p);
        c->bytes -= (int64_t)((count ? count : 1) * sizeof(T));
        *p = nullptr;
    }
}


namespace dmx {
// Derives the E-step call records (barcode-major, padded pairs), the M-step records
// (variant-major), the work items and the length-sorted work lists on the GPU from the
// uploaded COO columns, and stores them in the ctx (repack_device.hip).
int repack_on_device(dmx_ctx *c, const int32_t *h_variant, const int32_t *h_cb, const float *h_p);
// Variant matching + de-duplication of molecule calls on the GPU (demux.py:276-300, 332-365), then the
// layouts; sets c->N to the number of unique calls.
int pack_on_device(dmx_ctx *c, long long V, const int *var_chrom, const int *var_pos, const unsigned char *var_base,
                   long long n_calls, const int *call_chrom, const int *call_pos, const unsigned char *call_base,
                   const int *call_cb, const float *call_p, long long *n_matched, long long *n_unique,
                   long long *mol_per_variant);
// matched molecule calls (molecule order) -> the (barcode, SNP)-grouped layout of the aggregate_on_snps E-step
int ensure_sum_plan(dmx_ctx *c, long long K);  // dmx_api.cpp
// multi-GPU, M-step records by variant slice (repack_device.hip)
int wire_records_of(dmx_ctx *c, long long row_base, uint4 *d_out, long long capacity);
int install_mstep_records(dmx_ctx *c, const uint4 *d_rec, long long n, long long v_lo, long long v_hi);
// tile-major M-step records of variants [v_lo, v_hi) (the ctx's M-step records must cover exactly those); leaves n_mt == 0
// when the problem does not fit the form
int build_mstep_tiles(dmx_ctx *c, long long v_lo, long long v_hi);
int build_slice_row_index(dmx_ctx *c);  // d_slice_rec / d_slice_ptr of a variant-sharded rank (left null where it does not apply)
int plan_mstep_shifts(dmx_ctx *c);  // the tiles' fixed-point exponents per variant without the tile-major records (fixed-point work-item M-step)
void release_mstep_tiles(dmx_ctx *c);
int build_snp_groups(dmx_ctx *c, const unsigned long long *vb_keys, const unsigned *src_idx, const float *src_p, long long m);
int stage_containers_on_device(dmx_ctx *c, const dmx_call_container *parts, int n_parts);
int pack_staged_on_device(dmx_ctx *c, long long V, const int *var_chrom, const int *var_pos, const unsigned char *var_base,
                          const int *chrom_table, int n_table, long long *n_matched, long long *n_unique, long long *mol_per_variant);
void release_staged_calls(dmx_ctx *c);
int pack_containers_on_device(dmx_ctx *c, long long V, const int *var_chrom, const int *var_pos,
                              const unsigned char *var_base, const dmx_call_container *parts, int n_parts,
                              long long *n_matched, long long *n_unique, long long *mol_per_variant);
}  // namespace dmx
