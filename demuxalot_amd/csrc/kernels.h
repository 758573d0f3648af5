// kernels.h -- argument blocks and launchers of kernels.hip
#pragma once
#include <hip/hip_runtime.h>

#include "demux_hip.h"
#include "demux_hip_debug.h"

namespace dmx {

// E-step call records.  A barcode's calls are stored in input order, padded to a multiple of 8
// with neutral calls (keep 0, floor 1 -> log(p*0 + 1) = +0), two calls per 32-byte record so that
// a wavefront can fetch them with scalar loads and feed the packed float32 instructions straight
// from SGPR pairs:
//   row_off  byte offset of the call's variant row in the [V, G] float32 prob table
//   keep     1 - p_base_wrong                    (float32, as numpy evaluates `1 - e`)
//   floor    max(p_base_wrong, 1e-4)             (`e.clip(1e-4)`)
struct alignas(32) CallPair {
    unsigned row_off[2];
    float keep[2];
    float floor[2];
    unsigned reserved[2];
};
static_assert(sizeof(CallPair) == 32, "CallPair layout");

// block of the option triangle one thread of k_estep_pairblocks takes: g1 in R1 consecutive genotypes x g2 in R2.  E-step of
// 130k x 650k x 128 / 20k x 20k x 64 / 20k x 20k x 32, all with doublets (K = 8256 / 2080 / 528), in ms: 2 x 3: 74.8 / 2.97 / 1.23,
// 3 x 3: 75.3 / 2.61 / 1.82, 2 x 4: 77.4 / 3.56 / 1.56, 3 x 4: 77.1 / 3.08 / 2.18, 4 x 4: 86.0 / 3.67 / 2.57 (k_estep_block<6,true>: 88.2 / 3.45 / 1.2)
#ifndef DMX_PAIRBLOCK_R1
#define DMX_PAIRBLOCK_R1 2
#define DMX_PAIRBLOCK_R2 3
#endif
constexpr int PAIRBLOCK_R1 = DMX_PAIRBLOCK_R1, PAIRBLOCK_R2 = DMX_PAIRBLOCK_R2;

struct EstepSegment {
    int barcode;     // row
    int first_pair;  // first CallPair of the segment inside the row (a multiple of 4: whole 8-call groups)
    int n_pairs;
    int pad;
};

struct EstepArgs {
    const long long *pair_ptr;  // [B+1] offsets into `pairs` (barcode-major, rows padded to 4 pairs)
    const int *order;           // [B] barcodes by decreasing row length (work distribution)
    const CallPair *pairs;      // call records, see above (CALL_PAD_PAIRS neutral records behind the last row)
    const unsigned *call_rows;  // [2 * (n_pairs + CALL_PAD_PAIRS)] table row of every call of `pairs` (dictionary form)
    unsigned pairs_bytes;       // extent of `pairs` incl. the padding records when below 4 GiB (dictionary form: buffer addressing), else 0
    const float *prob;          // [V, G] genotype_prob, row-major
    const unsigned short *prob16;  // nullable: the same table rounded to bfloat16 (round to nearest even), entry (row, g) at BYTE row * G * 4 + g * 2 -
                                   // the float32 table's row offsets address it, half of every row's bytes are used (k_estep_tiled_coarse)
    const int *sum_plan;        // np.sum over a row of K values: {n_leaves, n_levels, n_roots, level offsets [n_levels + 1],
                                // leaves (start, length), inner nodes (left value, right value) level by level, roots}
    int sum_plan_values;        // leaves + inner nodes
    const unsigned *opt_pairs;  // [K] g1 | g2 << 16 (doublet runs only)
    const unsigned *pair_blocks;  // [n_pair_blocks] i | j << 16: 2 x 3 blocks (g1 in {2i, 2i+1}, g2 in {3j .. 3j+2}) that hold an option g1 <= g2 (k_estep_pairblocks)
    int n_pair_blocks;
    const float *pen;           // [K] doublet penalties
    const void *prior;          // nullable [B, K] prior logits (device)
    int prior_dtype;            // DMX_F32 / DMX_F64
    float *logits;              // [B, K]
    float *post;                // [B, K]
    float *post_singlets;       // nullable [B, G]: the singlet columns once more, row stride G (multi-GPU: this rank's block of the
                                // table the variant-sharded M-step reads, dmx_api.cpp: shard_mstep_by_variant)
    unsigned long long *nz;     // [B, ceil(G/64)] bit g set <=> !(post[b, g] <= nz_floor) (singlet columns; read by the M-step)
    uint2 *first;               // nullable [B] (G <= 64): {bits of post[b, lowest live genotype], count | first four live
                                // genotypes} (kernels.hip: nz_code): the M-step's one gather per call
    float nz_floor;             // 0, or NZ_FLOOR_SQUARE when the M-step squares (see below)
    unsigned long long *dense_calls;  // nullable [1 + DENSE_SLOTS]: slot 1 + (b % DENSE_SLOTS) += (padded) calls of every
                                      // barcode b with more than 4 live posteriors (G <= 64); slot 0: their sum (launch_sum_dense)
    long long B;
    unsigned prob_bytes;        // V * G * 4 (< 4 GiB): extent of the prob table for buffer addressing
    int G;
    int K;
    int fast;                   // DMX_ESTEP_FAST: tolerance mode (products of 8 terms + hardware log2), see kernels.hip
    // guarded mode (DMX_ESTEP_GUARDED; estep_epilogue.h: guard_flags): the fast kernels bound their deviation from the
    // reference per barcode and queue the barcodes whose posteriors / argmax are not provably within the contract
    float guard_accum;          // the guard's allowance for sums of log2 accumulated in float32 (k_estep_tiled_coarse), else 0: per addition
                                // 2^-24 of the running sum (estep_epilogue.h)
    float guard_alt_per_call;   // != 0: the OTHER pass's allowances (the fine pass: the coarse one's and vice versa; the exact kernels of a
    float guard_alt_accum;      //   direct E-step: the coarse one's): that guard is evaluated too and its flags are counted (GS_SLOTS_*)
    int guard_main_coarse;      // guard_per_call / guard_accum are the coarse pass's (the alternative is then the fine pass's)
    float guard_per_call;       // the guard's allowance per call for the fast arithmetic of the launching form (estep_epilogue.h: GUARD_PER_CALL,
                                // GUARD_PER_CALL_PRESCALED for k_estep_pairblocks' pre-scaled rows)
    int guard;                  // 1: the epilogue evaluates the guard and appends to guard_list (the fast kernels of a guarded E-step);
                                // 2: the exact kernels of a guarded E-step: when the E-step runs DIRECT (below) the guard is evaluated on
                                //    their own results and the barcodes that would have been queued are only counted
    unsigned *guard_count;      // the guard's device state (GuardState below): [GS_COUNT] barcodes queued (direct: that would have been) by this E-step
    int *guard_list;            // [B] the queued barcodes, dense (what the exact launch walks): written by k_guard_compact
    int *guard_sub;             // [GUARD_QUEUES x guard_sub_cap] the sub-queues the guard appends to: barcode b -> queue b % GUARD_QUEUES (one
    unsigned guard_sub_cap;     //   queue counter took ~5 ns per append - 0.2 ms of a 0.5 ms fast pass that queued 19 % of 200k barcodes)
    const unsigned *order_count;  // nullable: `order` holds *order_count entries (<= B), known on the device only (the exact
                                  // redo of the queued barcodes: k_estep_direct over guard_list)
    // Adaptive guarded mode.  With the fast pass taking F, the exact kernel over every barcode E and a fraction f of the
    // barcodes queued, a guarded E-step costs F + f E against the exact mode's E: more once f > 1 - F / E.  F / E is 0.56 at
    // 200k x 100k x 64 and 0.34 at K = 8256, but above 1 on short rows (50 calls per barcode: the per-barcode epilogue is
    // most of either kernel), so no fixed threshold on f is right.  The E-step is therefore TIMED on the device: the wall
    // clock is stored before the fast launches (k_guard_begin), between them and the exact launch (k_guard_stamp) and behind
    // it; k_guard_begin (between two E-steps on the stream) turns the stamps of the finished E-step into F and E (E from the redo's time over
    // its share of the barcodes until an E-step has run direct and measured it) and lets the next E-step run DIRECT when
    // F + f E > E: the fast kernels stand back (they read *direct and return), k_guard_compact lists `order_direct` - every
    // barcode - in the queue, so that the exact launch behind them walks every barcode; it counts the barcodes the guard would have
    // queued, so that f stays known and the fast pass returns when it pays again (3 % of hysteresis).  No host
    // synchronisation anywhere.
    const unsigned *direct;       // nullable: &state[GS_DIRECT]
    const int *order_direct;      // [B] what the queue is filled with when the E-step runs direct (barcodes by decreasing row length)
    // Split rows (tolerance / guarded mode, 64-lane form; dmx_api.cpp: build_row_segments): a launch cannot end before its
    // longest barcode does, and a barcode's walk is a chain of memory latencies (~0.7 us per 8 calls) - 0.35 ms for a
    // 4 000-call row, which is the whole E-step of a 25k-barcode shard.  The n_split longest barcodes (the first entries
    // of `order`) are therefore cut into segments walked by separate wavefronts; k_estep_join adds a barcode's
    // segment sums in order and runs the epilogue.  Any fixed association of the float64 sum is inside the guard's bound;
    // the exact kernels never split (their order is the reference's).
    const EstepSegment *segs;     // nullable [n_segs] segments of the split barcodes, longest first
    long long n_segs;
    long long n_split;            // barcodes cut into segments: order[0 .. n_split)
    const int *split_first;       // [n_split + 1] first segment of every split barcode
    double *seg_sums;             // [n_segs, K] float64 sums of the segments, in log2 units
    // tile-major schedule (k_estep_tiled); n_bins == 0: not built for this problem
    int tiled;                  // dmx_set_estep_schedule: 0 never, 1 when it pays (tolerance mode), 2 whenever built
    long long n_bins;
    int bin_rows_cap;           // R: barcode slots per bin (<= TILE_R_MAX)
    const int *bin_order;       // [n_bins] bins by decreasing number of calls
    const int *bin_rows;        // [n_bins][R] barcodes of the bin (-1: empty slot)
    const long long *bin_ptr;   // [n_bins + 1] first group (4 CallPairs = 8 calls) of every bin in tile_stream
    const unsigned *coarse_stream;    // nullable: the coarse pass's records (kernels.hip: coarse_walk), 16 dwords per block and record
    const long long *coarse_bin_ptr;  // [n_bins + 1] first record of every bin in coarse_stream
    const double *log2_keep;          // [B] sum of log2(keep) over the barcode's calls: the coarse pass's terms are (p + floor / keep)
    const CallPair *tile_stream;  // the call records in bin-major, tile-major, slot-minor order; reserved[0] of a
                                  // group's first pair = the slot (accumulator) the group belongs to
    // dictionary form (estep_dict.hip); dict_n == 0: not used by this launch
    int dict_n;                   // most distinct values in a row of `prob` (<= DICT_CAP)
    const unsigned char *dtab;    // [rows, dtab_pitch] per row of `prob`: its distinct values, then the codes of the options
                                  // (estep_dict.hip: DictRow)
    unsigned dtab_bytes;          // extent of dtab (< 4 GiB: buffer addressing)
    int dtab_pitch;
    // wide doublet tables (K > DICT_LANE_K): the workgroup-per-barcode dictionary kernel reads the two arrays of
    // launch_build_dict themselves
    // packed form (estep_packed.hip): the n_long longest barcodes (the first entries of `order`) take 64-lane
    // wavefronts inside the same launch
    long long n_long;
    const float *dict;            // [rows, DICT_CAP]
    const unsigned char *codes;   // [rows, dict_code_pitch(G)] 8 x index of every genotype's value in its row's dictionary
};

// device state of the guarded mode (dmx_ctx::d_guard_count: GS_WORDS + GUARD_SLOTS x unsigned)
enum { GS_COUNT = 0,         // barcodes queued by the current E-step (a direct one: that the guard would have queued)
       GS_DIRECT = 1,        // the current E-step runs the exact kernel on every barcode
       GS_VALID = 2,         // the words below describe a finished E-step of the resident problem
       GS_ROWS = 3,          // its barcode rows
       GS_PENDING = 4,       // ... and it is not yet part of the totals
       GS_DIRECT_STEPS = 5,  // E-steps run direct since the last reset
       GS_TOTAL = 6,         // (64 bit, two words) barcodes computed by the exact kernel since the last reset
       GS_T_FAST = 8,        // wall clock (low 32 bits) before the fast launches / between them and the exact launch / behind it
       GS_T_REDO = 9,
       GS_T_END = 10,
       GS_F_TICKS = 11,      // duration of the fast pass over all barcodes (0: not measured yet)
       GS_E_TICKS = 12,      // duration of the exact kernel over all barcodes: measured by a direct E-step, estimated before
       GS_E_MEASURED = 13,
       GS_O_TICKS = 14,      // duration of the exact launch on an empty queue
       GS_K = 15,            // option count of the finished E-step (another K: F and E start over)
       // Three levels (estep_epilogue.h; kernels.hip: k_guard_begin): 0 the COARSE pass (binary16 table: k_estep_tiled_coarse) + redo,
       // 1 the FINE pass (the tolerance arithmetic on the float32 table) + redo, 2 DIRECT.  Whichever kernels run evaluate BOTH guards,
       // so the queued fraction of either pass is known after every E-step, and both passes are timed when they run.
       GS_LEVEL = 16,        // level of the current E-step
       GS_SKIP_COARSE = 17,  // != 0: the coarse launch of the current E-step stands back (what its EstepArgs::direct points at)
       GS_SKIP_FINE = 18,    // ... the fine launch
       GS_C_TICKS = 19,      // duration of the coarse pass over all barcodes (0: not measured yet)
       GS_COUNT_FINE = 20,   // barcodes the fine / the coarse guard flagged in the finished E-step (GS_UNKNOWN: not evaluated); the guard of the pass
                             // that did not run sees one barcode in 8 (estep_epilogue.h: GUARD_ALT_SAMPLE): its count is that estimate
       GS_COUNT_COARSE = 21,
       GS_CAPABLE = 22,      // the current E-step evaluates the coarse guard too (the problem has a coarse pass)
       GS_COARSE_STEPS = 23, // E-steps that took the coarse pass since the last reset
       GS_PROBES = 24,       // E-steps that ran another level than the cheapest one to have it timed again, since the last reset
       GS_STREAK = 25,       // E-steps in a row on the current level without such a re-probe (k_guard_begin: GUARD_PROBE_STREAK)
       GS_WORDS = 26 };
// A pass is timed only when it runs.  A time taken once under other conditions - the first E-step of a call on a device that had idled
// runs at a fraction of its clocks - would stand for as long as the pass is not chosen, and it is not chosen because of that time: a
// 250-iteration call begun 0.5 s after the last one took the fine pass throughout, 1.88 instead of 1.10 ms per iteration (round 5,
// scripts/phase_timer_cost.py).  After GUARD_PROBE_STREAK E-steps in a row on one level the cheapest OTHER admissible level runs once if
// its standing price is below twice the running level's: at most 1.6 % of E-step time, nothing where the levels are further apart
// (coarse against fine at 200k x 100k x 64: 0.69 against 1.47 ms).
constexpr unsigned GUARD_PROBE_STREAK = 64;
constexpr unsigned GS_UNKNOWN = 0xFFFFFFFFu;
// The coarse pass (kernels.hip: k_estep_tiled_coarse) reads the genotype table as binary16, rounded to nearest: p' = p (1 + d), |d| <= 2^-11
// for every p >= 2^-14 (normal range; run_estep checks the clip), and forms a term as keep (p' + r), r = fl(floor / keep) with the slot tag in
// its low 4 bits: against the true term keep (p + floor / keep) the sum p' + r is off by at most 2^-11 (p') + 2^-24 + 2^-19 (r) + 2^-24 (its own
// rounding) relative - p, r >= 0 -, the reference's float32 term by 2 x 2^-24, so the logs differ by at most d / (1 - d) with
// d = 4.8828e-4 + 1.91e-6 + 4 x 6e-8: 4.9068e-4.  Then per call 3/4 of a float32 rounding of the 4-term product (4.5e-8) and 1/4 of v_log_f32's
// error on a product in [1e-16, 2^80) - within 2 ulp of a result below 128 in magnitude (tests/test_gpu_guarded.py): 1.53e-5 log2 units =
// 1.06e-5, a quarter of it per call: 2.7e-6.  (A product beyond float32 - calls with keep below 1e-9 - becomes inf / NaN: the guard flags NaN.)
// log2_keep (launch_build_coarse_stream) is a float64 sum of v_log_f32 results since round 6: the same constant for every option of a barcode,
// so its error cancels in the posteriors and in every difference of two logits; in a logit served by a coarse E-step (dmx_set_coarse_pass(2)) it
// is one more v_log_f32 error per call, 1.06e-5.  4.934e-4 + 1.06e-5 = 5.04e-4, rounded up (round 5: 4.94e-4, 0.1 % of headroom).
constexpr float GUARD_PER_CALL_COARSE = 5.1e-4f;
// The fine pass on the coarse pass's records (kernels.hip: k_estep_tiled_fine8; the tile-major stream released: dmx_set_lean_memory): float32 table,
// float64 sums, a term keep (p + r): 4 x 2^-24 relative against the reference's float32 term (r's division, the sum's rounding, the
// reference's two), the slot tag in one r of a block's 8 / cpg per batch (2^-19 of r <= 1.91e-6 of the term), the product's roundings and
// the mantissa's log as GUARD_PER_CALL prices them.  Every block of a barcode's batch carries exactly one tagged r and the guard counts
// padded calls, so the tag is charged to one call in 8 / cpg.
inline float guard_per_call_fine8(int cpg) { return 1.91e-6f / (float)(8 / cpg) + 4.0f * 6.0e-8f + 7.0e-8f; }
constexpr float GUARD_ACCUM_F32 = 6.0e-8f;  // 2^-24, rounded up: per float32 addition of the running sum (estep_epilogue.h)
constexpr int GUARD_SLOTS = 256;   // hashed counters behind the state words: barcodes flagged by a guard whose pass does not run (a direct
                                   // E-step: both; a fine one: the coarse guard's; a coarse one: the fine guard's) - a set per guard
constexpr int GUARD_QUEUES = 256;  // ... and behind those the lengths of the sub-queues (EstepArgs::guard_sub)
constexpr int GS_SLOTS_FINE = GS_WORDS, GS_SLOTS_COARSE = GS_WORDS + GUARD_SLOTS, GS_QUEUE_LEN = GS_WORDS + 2 * GUARD_SLOTS;
constexpr int GUARD_STATE_WORDS = GS_WORDS + 2 * GUARD_SLOTS + GUARD_QUEUES;
// between the fast launches and the exact launch of a guarded E-step: the sub-queues become the dense list, GS_COUNT their total
// length, GS_T_REDO the wall clock (a direct E-step: the list becomes every barcode, order_direct)
hipError_t launch_guard_compact(hipStream_t st, unsigned *state, const int *sub, unsigned sub_cap, int *list, const int *order_direct, long long B);
// capable: the coming E-step evaluates the coarse guard too; allow_coarse: ... and may take the coarse pass (nobody reads its logits)
hipError_t launch_guard_begin(hipStream_t st, unsigned *state, long long B, int K, int adaptive, int capable, int allow_coarse);
hipError_t launch_guard_stamp(hipStream_t st, unsigned *state, int which);  // state[which] = the device's wall clock (GS_T_REDO, GS_T_END)

constexpr int CALL_PAD_PAIRS = 64;     // readable neutral records behind the last barcode's row (pairs and call_rows)
constexpr int DICT_CAP = 8;            // distinct values per row the dictionary form handles (singlet runs)
constexpr int DICT_PAIR_CAP = 4;       // ... in doublet runs (10 pair values)
constexpr int DICT_LANE_K = 256;       // option tables up to this width take the lane-per-four-options dictionary kernel
__host__ __device__ inline int dict_code_pitch(int n) { return (n + 3) & ~3; }  // bytes between the rows of a code table of n codes per row

constexpr int DENSE_SLOTS = 1024;            // hashed counters of the dense-call statistic
hipError_t launch_sum_dense(hipStream_t st, unsigned long long *counters, unsigned *guard_state = nullptr);  // (also stamps GS_T_END)
constexpr int TILE_R_MAX = 9;                // barcodes per bin at most (LDS: 4 waves x 9 x 64 doubles = 18 KB per block)
constexpr long long TILE_BYTES = 1 << 20;    // genotype-table bytes per variant tile: a quarter of an XCD's 4 MB L2 (tolerance-mode E-step on
                                             // 200k x 100k x 64: 1.480 / 1.483 / 1.510 / 1.546 ms with tiles of 0.5 / 1 / 2 / 3 MB; DEMUXALOT_AMD_TILE_KB)
#ifndef DMX_TILE_MIN_BARCODES
#define DMX_TILE_MIN_BARCODES 8192
#endif
// below this there are fewer bins than wavefront slots.  (65 536 until the coarse pass: the FINE pass gains nothing from the schedule on smaller
// problems - 25k barcodes 0.221 against 0.234 ms, 10k 0.134 against 0.115 -, the coarse pass, which only exists on it, does: E-step of 10k / 25k /
// 50k barcodes x 100k SNPs x 64 genotypes 0.115 -> 0.091 / 0.234 -> 0.141 / 0.451 -> 0.251 ms; scripts/small_shard_variants.sh)
constexpr long long TILE_MIN_BARCODES = DMX_TILE_MIN_BARCODES;
#ifndef DMX_TILE_MIN_TABLE_BYTES
#define DMX_TILE_MIN_TABLE_BYTES (1 << 20)
#endif
// (8 MB until the coarse pass: a smaller table is L2-resident and the fine pass gains nothing from the schedule; the coarse pass, which only
// exists on it, does - every row still misses the 32 KB L1: 20k barcodes x 10k SNPs x 64 genotypes, a 5 MB table: E-step 0.141 -> 0.107 ms)
constexpr long long TILE_MIN_TABLE_BYTES = DMX_TILE_MIN_TABLE_BYTES;

struct MstepArgs {
    const int *order;               // [n_items] items by decreasing length (work distribution)
    const long long *item_start;    // [n_items] first CSC call of the item
    const int *item_len;            // [n_items] number of calls (<= item_calls_for(N))
    const uint2 *calls;             // [N] (compressed_cb, bits of 1 - p_base_wrong), variant-major
    const float *post;              // [B, K] posteriors (singlet columns 0..G-1 are read)
    const unsigned long long *nz;   // [B, ceil(G/64)] non-zero bitmap of the singlet posteriors
    bool wide;                      // never use 32-bit buffer offsets (dmx_set_mstep_wide_addresses)
    unsigned long long first_bytes; // 8 B
    const uint2 *first;             // [B] bitmap + posterior of the lowest non-zero singlet column (G <= 64), as the E-step wrote them
    double *partial;                // [n_items, G]
    // Variants of ONE work item (all but the hottest few hundred) need no combining pass: the item's wavefront rounds its
    // sums straight into the output table (out32 / out64: what k_mcombine would write; row prow[v] or v), and k_mcombine
    // only visits the variants of several items and those without calls.  item_variant == nullptr: every item leaves
    // its partial sums (the chunked exchange, whose output rows differ per chunk).
    const int *item_variant;        // nullable [n_items] variant of every item
    unsigned long long redo_cap;    // capacity of the redo queue (launch_mcombine: long variants from the front, the others from the back)
    bool tiles_done;                // the sums were written by k_mstep_tiles unless the dense regime's kernel ran (k_mcombine then only acts in that regime)
    const long long *item_ptr;      // [V + 1] first item of every variant
    const int *prow;                // nullable [V] row of every variant in the output table
    float *out32;                   // exactly one of the two (or none: item_variant == nullptr)
    double *out64;
    long long n_items;
    long long K;
    unsigned long long post_bytes;  // B * K * 4
    int G;
    int square;   // contribution_power == 2
    float power;  // otherwise
    // G <= 64: statistic of the last E-step (EstepArgs::dense_calls) and the number of (padded) E-step calls; the
    // call-parallel kernel runs when few calls are dense, the dense kernel otherwise.  dense_calls == nullptr (or a
    // posterior table beyond 4 GiB): always the call-parallel kernel.
    const unsigned long long *dense_calls;
    unsigned long long total_calls;
    // FIXED-POINT work-item form (fixed_shift_v != nullptr; G <= 64, no exact additions): the items add the integers rint(c 2^shift)
    // the tile-major kernel adds - shift = the variant's (MIncrArgs::shift_v) -, so their sums ARE the tile-major form's, bit for bit
    // (integer addition does not care who adds in which order), without the tile-major records: what a call too short to pay for
    // their sort runs, and what the incremental M-step (MIncrArgs) builds on there.  partial[] then holds 64-bit integers;
    // k_mcombine adds them, converts once and leaves the sums in fixed_acc64.  fixed_state (nullable): the incremental M-step's
    // state words - the launch stands back unless they ask for the full pass.
    const unsigned char *fixed_shift_v;
    unsigned long long *fixed_acc64;
    const unsigned *fixed_state;
    // incremental M-step of a variant-sharded rank (MIncrArgs::changed_map): the delta pass stands back when the changed barcodes are more
    // than an eighth of incr_total (2 per barcode row of the job); 0: total_calls (one context with the calls of its barcodes)
    unsigned long long incr_total;
};

// A posterior p <= 2^-80 contributes (p * keep)^2 = +0 exactly for |keep| <= 32 (|p * keep| <= 2^-75, and a
// float32 square below 2^-150 rounds to +0), so with contribution_power == 2 the M-step may treat it as absent.
// exp(-55) is about 2^-80: most barcodes then have ONE live posterior instead of one per genotype within 103
// logit units of the best.
constexpr float NZ_FLOOR_SQUARE = 8.271806125530277e-25f;  // 2^-80

// Longest run of one variant's calls handled by one wavefront.  Long items keep hot variants in few pieces (every
// extra piece makes more sums order-sensitive, see k_mcombine), short items keep the tail of the launch short:
// the length is chosen per problem so that the longest item is about a tenth of one wavefront's share of a full
// launch, between these bounds.
constexpr int MIN_ITEM_CALLS = 1024, MAX_ITEM_CALLS = 16384;
inline int item_calls_for(long long n_calls)
{
    long long len = MIN_ITEM_CALLS;
    while (len < MAX_ITEM_CALLS && len * 6000 < n_calls) len *= 2;
    return (int)len;
}

// P-step of variants [v_begin, v_begin + n_rows) (whole SNP groups); the result goes to row prow[v] of `prob`
// (padded multi-GPU layout) or row v when prow is null
hipError_t launch_probs_from_betas(hipStream_t st, const float *prior, const float *addition, const int *v2snp,
                                   const int *snp_ptr, const int *snp_vars, long long v_begin, long long n_rows, long long n_snps,
                                   int G, const int *prow, float lo, float hi, float *prob, unsigned short *prob16 = nullptr);
// the same from caller-supplied float64 betas (no addition): numpy divides float64 / float64 and rounds once
hipError_t launch_probs_from_betas_f64(hipStream_t st, const double *betas, const int *v2snp, const int *snp_ptr,
                                       const int *snp_vars, long long V, long long n_snps, int G, const int *prow, float lo,
                                       float hi, float *prob);
// sets flags[0] bit 0 when a value lies outside [0, 1] or is not finite
hipError_t launch_check_unit_range(hipStream_t st, const float *x, long long n, int *flags);
hipError_t launch_estep(hipStream_t st, const EstepArgs &a, bool pairs);
// the coarse pass's records from the tile-major stream (coarse_bin_ptr first, one block; then the stream); cpg = calls per gather: 1 for
// 65 .. 128 genotypes, 2 for 33 .. 64, 4 for 17 .. 32; zero_off = byte offset of the all-zero row behind the table
constexpr int coarse_calls_per_gather(int K) { return K > 64 ? 1 : K > 32 ? 2 : 4; }
constexpr int coarse_batches_per_record(int cpg) { return cpg; }  // (kernels.hip: CoarseShape<CPG>::BPR)
hipError_t launch_build_coarse_stream(hipStream_t st, const CallPair *stream, const long long *bin_ptr, long long n_bins, unsigned zero_off,
                                      int cpg, long long *coarse_bin_ptr, unsigned *out, const int *bin_rows, int R, double *log2_keep);
hipError_t launch_prob_to_half(hipStream_t st, const float *prob, long long rows, int G, unsigned short *out, const unsigned *skip);  // EstepArgs::prob16; *skip != 0: nothing
hipError_t launch_softmax_rows(hipStream_t st, const EstepArgs &a);  // rows left as logits by the option-tile launches
// dictionary form (estep_dict.hip): distinct values and codes of every row of `prob`; stat[0] = most distinct values
// in a row (DICT_CAP + 1: some row has more)
hipError_t launch_build_dict(hipStream_t st, const float *prob, long long rows, int G, float *dict, unsigned char *codes, unsigned *stat);
// the packed table of the dictionary-form kernel from the two arrays above; `distinct` as found by launch_build_dict
int dict_table_pitch(int distinct, int K, bool pairs);
hipError_t launch_pack_rows(hipStream_t st, const float *dict, const unsigned char *codes, const unsigned *opt_pairs, long long rows, int G,
                            int K, bool pairs, int distinct, unsigned char *table);
hipError_t launch_estep_dict(hipStream_t st, const EstepArgs &a, bool pairs);
hipError_t launch_estep_dict_block(hipStream_t st, const EstepArgs &a);  // doublet tables of more than DICT_LANE_K options
hipError_t launch_mstep(hipStream_t st, const MstepArgs &a);
// sums the item partials of variants [v0, v1) and redoes, in the reference's order, the sums whose float32
// rounding could depend on the order (redo: queue of a.redo_cap = (n_items / 2 + 1) * G entries, n_redo: its TWO counters -
// variants of more than EXACT_LONG calls are queued from the front, the others from the back)
// prow (nullable): row of every variant in the output tables (padded multi-GPU exchange buffer)
// vlist (nullable): the variants are entries [v0, v1) of this list instead of v0 .. v1 - 1 (chunks of the pipelined exchange)
// skip_single: the variants of one item were written by the M-step kernels themselves (MstepArgs::item_variant)
// Tile-major M-step (fixed-point sums, order-independent and bit-reproducible, but not the reference's float64 sum bit for bit:
// not for dmx_set_exact_additions), G <= 64, contribution_power > 0.  The variant axis is cut into tiles of at most
// `tv` variants; the M-step records are kept once more sorted by (tile, barcode row), so that a tile's calls read the barcode
// codes in ascending order (the item form's one gather per call out of a 1.6 MB table, 112 G requests/s to the L2s, was
// what bounded it).  One workgroup per tile: float64 accumulators [variants of the tile][G] in LDS, ds_add_f64.
struct MTileArgs {
    const uint2 *stream;   // x = barcode row | variant in tile << 24, y = bits of 1 - p_base_wrong
    const long long *ptr;  // [n_tiles + 1]
    const int *first;      // [n_tiles + 1] first variant of every tile
    const int *order;      // [n_tiles] tiles by decreasing number of calls
    const int *shift;      // [n_tiles] binary exponent of the tile's fixed-point grid (k_mstep_tiles: contributions are added as rint(c 2^shift))
    long long n_tiles;
    int tv;
    // incremental M-step (MIncrArgs below; both null: every M-step the full pass)
    unsigned long long *acc64;   // [V, G] the tiles' fixed-point sums, kept between M-steps
    const unsigned *incr_state;  // the current M-step's state words: the launch stands back unless they ask for the full pass
};
// Incremental M-step (kernels.hip: k_mincr_*).  The tile-major M-step adds INTEGERS - rint(c 2^shift) of every contribution
// c = (posterior x keep)^power -, so its sums can be UPDATED exactly: after a full pass has left them in acc64, an M-step only has
// to visit the barcodes whose posteriors changed where it matters - a posterior below 2^-26 contributes c 2^shift < 2^-2, i.e.
// exactly 0, so a change between two such values changes nothing - and add, per call and genotype, the difference of the new and the
// old integer.  On converged iterations that is a fraction of a percent of the calls (99 % of the barcodes sit at a posterior of
// exactly 1.0).  The result is the full pass's, bit for bit.  All decisions on the device: k_mincr_changes lists the changed barcodes
// and counts their calls; when the sums are not valid, the changed barcodes hold more than an eighth of the calls, or the posteriors
// are dense (dense_regime), the delta pass stands back and the full pass (k_mstep_tiles) runs instead.
struct MIncrArgs {
    unsigned *state;        // this M-step's state words (IS_*); `next`: the other set, prepared by k_mincr_finish for the next M-step
    unsigned *next;
    unsigned *counters;     // [0] full passes, [1] delta passes, [2] barcodes the last M-step found changed (since dmx_reset_timings)
    unsigned long long *acc64;     // [V, G]
    float *prev;                   // [B, G] singlet posteriors the sums in acc64 were formed from
    uint2 *prev_first;             // [B] ... and their codes (MstepArgs::first)
    int *list;                     // [B] changed barcodes
    unsigned char *touched;        // [V] variants whose sums the delta pass changed (their rows of the addition are converted again)
    const unsigned char *shift_v;  // [V] fixed-point exponent of the variant's tile (MTileArgs::shift)
    const CallPair *pairs;         // barcode-major call records + the table row (= variant) of every call
    const unsigned *call_rows;
    const long long *pair_ptr;
    long long B, V;
    float floor;            // mincr_floor(power)
    // Variant-sharded rank (round 6): B = the barcode rows of ALL ranks (post / first: the gathered tables, row stride G), pair_ptr / pairs /
    // call_rows null - a rank holds the barcode-major records of its own barcodes only.  The changed barcodes are flagged in changed_map
    // [B] and the delta pass is a MASKED WALK of the rank's variant-major records (k_mincr_delta_masked): a record whose barcode is
    // flagged adds the difference of its new and old integer contribution; k_mincr_finish then brings prev / prev_first of the listed
    // barcodes up to date and clears their flags.
    unsigned char *changed_map;
    // A rank that exchanges sums (sliced table): the records' table rows are the PADDED rows of the exchange layout; row_variant
    // [padded rows] brings them back to variants (acc64, shift_v, touched are per variant).  Null: table rows = variants.
    const int *row_variant;
    // ... or, with the slice's records sorted by barcode row too (dmx_ctx::d_slice_rec, build_slice_row_index): rec_ptr [B + 1], rec
    // {variant, bits(1 - e)} - the delta pass of one context (k_mincr_delta: a workgroup per changed barcode) on the rows of all ranks;
    // changed_map is then null, IS_CALLS counts the changed barcodes' real calls against MstepArgs::incr_total = the slice's calls.
    const uint2 *rec;
    const long long *rec_ptr;
};
enum { IS_N = 0,        // changed barcodes of this M-step
       IS_CALLS = 2,    // (64 bit, words 2 and 3) their (padded) calls
       IS_VALID = 4,    // acc64 / prev hold the previous M-step's sums and posteriors
       IS_STREAK = 5,   // full passes in a row that the changes asked for
       IS_SITOUT = 6,   // M-steps still to go without looking for changes (k_mincr_finish)
       IS_FORCE = 7,    // this M-step's delta pass builds the sums from nothing (acc64, prev and the addition zeroed by the host): never the full pass
       IS_WORDS = 8 };
// posteriors below this contribute exactly 0 on every tile's grid (shift <= 50): p^power 2^50 <= 2^-2 for p <= 2^(-52 / power) - 2^-26 for the
// reference's power of 2; powers for which that is not a normal float32: 0 (every bit counts)
inline float mincr_floor(float power) { return power * 126.0f > 52.0f ? exp2f(-52.0f / power) : 0.0f; }
hipError_t launch_mstep_incremental(hipStream_t st, const MstepArgs &a, const MTileArgs &t, const MIncrArgs &x);
// the same with the fixed-point work-item form as the full pass (MstepArgs::fixed_shift_v): no tile-major records needed
hipError_t launch_mstep_items_incremental(hipStream_t st, const MstepArgs &a, const MIncrArgs &x);
// ... of a variant-sharded rank: changes over the gathered tables -> masked walk of the slice's records | the tile-major full pass -> finish
hipError_t launch_mstep_incremental_sharded(hipStream_t st, const MstepArgs &a, const MTileArgs &t, const MIncrArgs &x);
constexpr int MTILE_LDS_BYTES = 64 * 1024;  // accumulators of a tile: with the 12 KB of dense-call queues, two workgroups of 1024 threads per CU
constexpr int MTILE_MAX_VARIANTS = 128;     // 7 bits of the record
hipError_t launch_mstep_tiles(hipStream_t st, const MstepArgs &a, const MTileArgs &t);

hipError_t launch_mcombine(hipStream_t st, const MstepArgs &a, const long long *item_ptr, long long v0, long long v1,
                           const int *prow, float *add32, double *add64, unsigned long long *redo, unsigned *n_redo,
                           const int *vlist = nullptr, bool skip_single = false);
hipError_t launch_store_slice(hipStream_t st, const void *slice, bool f64, long long v_begin, long long n_rows, int G, float *add);
// call_rows (nullable): the compact row array of the same records, rewritten as well
// estep_packed.hip: exact E-step of narrow doublet tables, several option slots per lane (K = 36: 8 lanes x 5 slots)
bool estep_packed_shape(int K, int G, int *lanes, int *slots);
hipError_t launch_estep_packed(hipStream_t st, const EstepArgs &a);
hipError_t launch_remap_row_offsets(hipStream_t st, CallPair *pairs, long long n_pairs, unsigned row_bytes, const int *new_rows,
                                    unsigned *call_rows);
hipError_t launch_f64_to_f32(hipStream_t st, const double *in, float *out, long long n);
hipError_t launch_add_f32(hipStream_t st, const float *a, const float *b, float *out, long long n);  // out = a + b, one float32 rounding (numpy's)
// one wavefront that keeps the stream busy for `ticks` of the constant-rate wall clock (emulated wire: dmx_comm_init_emulated)
hipError_t launch_delay(hipStream_t st, long long ticks);
// compact exchange of the posterior rows (kernels.hip: k_post_compact_build / k_post_reconstruct; dmx_exchange.cpp: gather_posteriors)
// (peers: emulated wire only, else nullptr - the other ranks' headers are cleared)
// (sent [B, G] / sent_multi [B]: what the receivers hold of this rank's rows with several live posteriors - only rows that differ are listed)
hipError_t launch_post_compact_build(hipStream_t st, const uint2 *first, const float *post, long long B, int G, unsigned cap, unsigned *block,
                                     float *sent, unsigned char *sent_multi, unsigned *peers, unsigned long long block_words, int nranks, int own);
// compact exchange of the genotype table (kernels.hip: k_prob_changes_build / k_prob_changes_apply; dmx_steps.cpp: run_pstep)
hipError_t launch_prob_changes_build(hipStream_t st, const float *slice, float *prev, long long rows, int G, unsigned cap, unsigned *block,
                                     unsigned *peers, unsigned long long block_words, int nranks, int own);
hipError_t launch_prob_changes_apply(hipStream_t st, float *table, const unsigned *blocks, unsigned long long block_words, long long slice_rows, int G,
                                     int nranks, int own, unsigned cap, unsigned short *table16);
// out_host_visible: [nranks + 1] - the counts, then `seq` (written last, system scope: the host polls it)
hipError_t launch_post_counts(hipStream_t st, const unsigned *blocks, unsigned long long block_words, int nranks, unsigned *out_host_visible, unsigned seq);
hipError_t launch_post_reconstruct(hipStream_t st, const uint2 *first_g, float *post_g, const unsigned *blocks, unsigned long long block_words,
                                   long long rows_pad, int G, int nranks, int own, unsigned cap, uint2 *seen);
hipError_t launch_f32_to_f64(hipStream_t st, const float *in, double *out, long long n);
hipError_t launch_prior_betas(hipStream_t st, const float *betas, float *bsum, const unsigned long long *n_mol,
                              const int *v2snp, const int *snp_ptr, const int *snp_vars, long long V, int G,
                              double default_prior, float *out);
// recomputes the M-step's bitmap / first-posterior table from the stored posteriors
hipError_t launch_rebuild_nz(hipStream_t st, const float *post, long long B, int K, int G, float nz_floor,
                             unsigned long long *nz, uint2 *first);
hipError_t launch_assign(hipStream_t st, const float *post, long long B, int K, int *best, float *best_p);
hipError_t launch_test_log(hipStream_t st, const float *in, float *out, long long n);
hipError_t launch_test_log_hot(hipStream_t st, const float *in, float *out, long long n);
hipError_t launch_test_exp(hipStream_t st, const float *in, float *out, long long n);
hipError_t launch_test_log2_hw(hipStream_t st, const float *in, float *out, long long n);
hipError_t launch_test_softmax(hipStream_t st, const float *in, float *out, long long rows, int cols);

}  // namespace dmx
