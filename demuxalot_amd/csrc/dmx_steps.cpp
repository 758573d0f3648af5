// dmx_steps.cpp -- the step drivers behind dmx_probs_from_betas / dmx_estep / dmx_mstep / dmx_em / dmx_run_iterations: which form of
// which kernel runs (dictionary / packed / tiled / coarse / direct E-step under the guard; work-item / fixed-point / tile-major /
// incremental M-step), their device-side state, and the entry points themselves.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <array>
#include <chrono>
#include <functional>

#include <algorithm>
#include <mutex>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "dmx_ctx.h"
#include "dmx_host.h"

using namespace dmx::host;

namespace dmx {
namespace host {

// Split rows of the tolerance / guarded E-step (kernels.h: EstepArgs::segs).  A wavefront walks its barcode's calls as a
// chain of memory latencies, so the longest row bounds a launch from below; rows with more CallPairs than half of what a
// wavefront slot of the chip gets on average (and at least 128) are cut into equal segments of whole 8-call groups.
// On the 200k-barcode bench workload nothing is cut (4 800 pairs per slot against rows of at most 2 000); on one rank's
// share of it on 8 GPUs (25k barcodes, 600 pairs per slot) the rows beyond 600 calls are.
int build_row_segments(dmx_ctx *c)
{
    c->n_segs = c->n_split = 0;
    const long long B = c->B;
    if (B == 0 || c->n_pairs == 0) return 0;
    if (!c->n_simd) {
        int cus = 0;
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
        c->n_simd = 4 * cus;
    }
    const long long slots = 8ll * std::max(1, c->n_simd);
    long long cap = std::max<long long>(128, c->n_pairs / (2 * slots));
    cap = (cap + 3) & ~3ll;
    std::vector<long long> pair_ptr((size_t)B + 1);
    std::vector<int> order((size_t)B);
    HIP_TRY(hipMemcpyAsync(pair_ptr.data(), c->d_pair_ptr, sizeof(long long) * (B + 1), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(order.data(), c->d_bc_order, sizeof(int) * B, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::vector<dmx::EstepSegment> segs;
    std::vector<int> first(1, 0);
    for (long long j = 0; j < B; j++) {  // `order` is sorted by decreasing length: the split rows are its first entries
        const int b = order[(size_t)j];
        const long long pairs = pair_ptr[(size_t)b + 1] - pair_ptr[(size_t)b];
        if (pairs <= cap) break;
        const long long pieces = (pairs + cap - 1) / cap;
        const long long groups = pairs / 4, per = (groups + pieces - 1) / pieces;  // whole 8-call groups per segment
        for (long long g0 = 0; g0 < groups; g0 += per)
            segs.push_back({b, (int)(4 * g0), (int)(4 * std::min(per, groups - g0)), 0});
        first.push_back((int)segs.size());
    }
    if (segs.empty()) return 0;
    // (the segments of one barcode stay adjacent and in order - split_first indexes them - and are of nearly equal length;
    // the barcodes come longest first, so the work list is roughly longest-first too)
    c->n_segs = (long long)segs.size();
    c->n_split = (long long)first.size() - 1;
    DMX_TRY(dev_alloc(c, &c->d_segs, segs.size()));
    DMX_TRY(dev_alloc(c, &c->d_split_first, first.size()));
    HIP_TRY(hipMemcpyAsync(c->d_segs, segs.data(), sizeof(dmx::EstepSegment) * segs.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_split_first, first.data(), sizeof(int) * first.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));  // locals
    return 0;
}

int ensure_options(dmx_ctx *c, int with_doublets, const float *penalties)
{
    const int G = c->G;
    const long long K = with_doublets ? (long long)G * (G + 1) / 2 : G;
    if (K > (1 << 24)) return fail(DMX_ERR_UNSUPPORTED, "too many options (%lld)", K);
    if (G > 1024)  // one lane holds at most 16 genotype accumulators (E- and M-step); the block form stages G rows in LDS
        return fail(DMX_ERR_UNSUPPORTED, "more than 1024 genotypes are not supported (G=%d)", G);
    if (K > c->cap_k) {
        dev_free(c, &c->d_pen, (size_t)c->cap_k);
        dev_free(c, &c->d_pairs, (size_t)c->cap_k);
        c->cap_k = 0;
        DMX_TRY(dev_alloc(c, &c->d_pen, (size_t)K));
        DMX_TRY(dev_alloc(c, &c->d_pairs, (size_t)K));
        c->cap_k = (int)K;
    }
    if (c->B * K > c->cap_bk) {
        dev_free(c, &c->d_logits, (size_t)c->cap_bk);
        dev_free(c, &c->d_post, (size_t)c->cap_bk);
        c->cap_bk = 0;
        DMX_TRY(dev_alloc(c, &c->d_logits, (size_t)(c->B * K)));
        DMX_TRY(dev_alloc(c, &c->d_post, (size_t)(c->B * K)));
        c->cap_bk = c->B * K;
    }
    // option k -> (g1, g2): singlets (g, g) first, then g1 < g2 row-major (demux.py:175-191)
    std::vector<unsigned> pairs((size_t)K);
    for (int g = 0; g < G; g++) pairs[g] = (unsigned)g | ((unsigned)g << 16);
    if (with_doublets) {
        size_t k = G;
        for (int g1 = 0; g1 < G; g1++)
            for (int g2 = g1 + 1; g2 < G; g2++) pairs[k++] = (unsigned)g1 | ((unsigned)g2 << 16);
    }
    HIP_TRY(hipMemcpyAsync(c->d_pairs, pairs.data(), sizeof(unsigned) * K, hipMemcpyHostToDevice, c->stream));
    // 2 x 3 blocks of the (g1, g2) triangle for the tolerance mode's workgroup-per-barcode kernel (kernels.hip: k_estep_pairblocks)
    std::vector<unsigned> blocks;
    if (with_doublets && K > 256) {
        constexpr int R1 = dmx::PAIRBLOCK_R1, R2 = dmx::PAIRBLOCK_R2;
        for (int i = 0; R1 * i < G; i++)
            for (int j = 0; R2 * j < G; j++)
                if (R2 * j + R2 - 1 >= R1 * i) blocks.push_back((unsigned)i | ((unsigned)j << 16));  // some g2 of the block is >= its smallest g1
    }
    c->n_pair_blocks = (int)blocks.size();
    if (c->n_pair_blocks > c->cap_pair_blocks) {
        dev_free(c, &c->d_pair_blocks, (size_t)c->cap_pair_blocks);
        c->cap_pair_blocks = 0;
        DMX_TRY(dev_alloc(c, &c->d_pair_blocks, blocks.size()));
        c->cap_pair_blocks = c->n_pair_blocks;
    }
    if (c->n_pair_blocks) HIP_TRY(hipMemcpyAsync(c->d_pair_blocks, blocks.data(), sizeof(unsigned) * blocks.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_pen, penalties, sizeof(float) * K, hipMemcpyHostToDevice, c->stream));
    DMX_TRY(dmx::ensure_sum_plan(c, K));
    if ((size_t)c->n_segs * (size_t)K > c->cap_seg_sums) {
        dev_free(c, &c->d_seg_sums, c->cap_seg_sums);
        c->cap_seg_sums = 0;
        DMX_TRY(dev_alloc(c, &c->d_seg_sums, (size_t)c->n_segs * (size_t)K));
        c->cap_seg_sums = (size_t)c->n_segs * (size_t)K;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));  // `pairs` is a local
    if ((int)K != c->K) c->have_post64 = false;  // the float64 results of dmx_estep_snp were laid out for another K
    c->K = (int)K;
    return 0;
}

int upload_prior_logits(dmx_ctx *c, const void *prior, int dtype)
{
    if (!prior) return 0;
    if (dtype != DMX_F32 && dtype != DMX_F64) return fail(DMX_ERR_INVALID, "prior_dtype must be DMX_F32 or DMX_F64");
    const size_t bytes = (size_t)c->B * c->K * (dtype == DMX_F64 ? 8 : 4);
    if (bytes > c->cap_prior) {
        if (c->d_prior_logits) {
            (void)hipFree(c->d_prior_logits);
            c->bytes -= (int64_t)c->cap_prior;
            c->d_prior_logits = nullptr;
            c->cap_prior = 0;
        }
        hipError_t e = hipMalloc(&c->d_prior_logits, bytes ? bytes : 1);
        if (e != hipSuccess) return fail(DMX_ERR_HIP, "hipMalloc(prior logits, %zu bytes): %s", bytes, hipGetErrorString(e));
        c->cap_prior = bytes;
        c->bytes += (int64_t)bytes;
    }
    HIP_TRY(hipMemcpyAsync(c->d_prior_logits, prior, bytes, hipMemcpyHostToDevice, c->stream));
    return 0;
}

// [V, G] float32 table in the layout of d_prob <-> dense host table
int copy_prob_out(dmx_ctx *c, float *dst)
{
    if (!dst) return 0;
    const int G = c->G;
    if (!c->sliced) return copy_out(c, dst, c->d_prob, (size_t)c->V * G);
    for (int r = 0; r < c->nranks; r++) {
        const long long rows = c->cut[r + 1] - c->cut[r];
        if (rows) HIP_TRY(hipMemcpyAsync(dst + c->cut[r] * G, c->d_prob + (size_t)r * c->slice_rows * G, sizeof(float) * rows * G, hipMemcpyDeviceToHost, c->stream));
    }
    return 0;
}

int copy_prob_in(dmx_ctx *c, const float *src)
{
    const int G = c->G;
    c->prob_prev_valid = false;  // (the table is the caller's now: what the ranks hold of each other's slices is no longer what they sent)
    if (!c->sliced) {
        HIP_TRY(hipMemcpyAsync(c->d_prob, src, sizeof(float) * c->V * G, hipMemcpyHostToDevice, c->stream));
        return 0;
    }
    for (int r = 0; r < c->nranks; r++) {
        const long long rows = c->cut[r + 1] - c->cut[r];
        if (rows) HIP_TRY(hipMemcpyAsync(c->d_prob + (size_t)r * c->slice_rows * G, src + c->cut[r] * G, sizeof(float) * rows * G, hipMemcpyHostToDevice, c->stream));
    }
    return 0;
}

// sliced mode: after an M-step only this rank's slice of d_add is current; assemble the whole table (collective:
// every rank must get here)
int ensure_full_addition(dmx_ctx *c)
{
    if (!c->add_partial) return 0;
    const int G = c->G, n = c->nranks;
    float *stage = (float *)c->d_exch;
    const size_t block = (size_t)c->slice_rows * G;
    const long long mine = c->cut[c->rank + 1] - c->cut[c->rank];
    if (mine) HIP_TRY(hipMemcpyAsync(stage + c->rank * block, c->d_add + c->cut[c->rank] * G, sizeof(float) * mine * G, hipMemcpyDeviceToDevice, c->stream));
    DMX_TRY(coll_all_gather(c, stage, block, "addition"));
    for (int k = 0; k < n; k++) {
        const long long rows = c->cut[k + 1] - c->cut[k];
        if (rows && k != c->rank) HIP_TRY(hipMemcpyAsync(c->d_add + c->cut[k] * G, stage + k * block, sizeof(float) * rows * G, hipMemcpyDeviceToDevice, c->stream));
    }
    // the exchange buffer's padding rows must be zero again before the next reduce-scatter
    HIP_TRY(hipMemsetAsync(c->d_exch, 0, c->exch_bytes, c->stream));
    c->add_partial = false;
    return 0;
}

// Whether the E-step behind a P-step with clip `lo` can take the coarse pass (kernels.hip: k_estep_tiled_coarse) - what run_estep asks
// again, of the table it finds.
bool coarse_capable(const dmx_ctx *c, int with_doublets, float lo)
{
    // (its records are there, or can be built from the tile-major stream: dmx_set_lean_memory releases that one behind the build)
    return c->coarse_pass && c->estep_mode == DMX_ESTEP_GUARDED && !with_doublets && c->K > 16 && c->K <= 128 && c->tiled_estep && c->n_bins > 0 &&
           (c->coarse_ready || c->d_tile_stream != nullptr) && lo >= 6.2e-5f && ((unsigned long long)c->prob_rows + 1ull) * (unsigned long long)c->G * 4ull < (1ull << 32);
}

// the table as binary16 + the all-zero row the padding calls gather (EstepArgs::prob16)
int ensure_prob16(dmx_ctx *c)
{
    const size_t need16 = ((size_t)c->prob_rows + 1) * c->G * 2;
    if (need16 > c->cap_prob16) {
        dev_free(c, &c->d_prob16, c->cap_prob16);
        c->cap_prob16 = 0;
        DMX_TRY(dev_alloc(c, &c->d_prob16, need16));
        c->cap_prob16 = need16;
        c->prob16_valid = false;
        HIP_TRY(hipMemsetAsync(c->d_prob16, 0, need16 * sizeof(unsigned short), c->stream));
    }
    return 0;
}

// with_half: the E-step behind this P-step may take the coarse pass - the kernel writes the table as binary16 too (one rank, whole
// table; a sliced run converts behind the all-gather of the slices: run_estep)
int run_pstep(dmx_ctx *c, float lo, float hi, bool with_addition, bool with_half)
{
    // A sliced run converts the whole table behind the all-gather of the slices (run_estep) - unless the slices travel as lists of
    // changed rows: then this rank's slice is written as binary16 here, the others' changed rows where they are applied, and the table
    // that was valid before stays so.
    const bool half_rows = with_half && c->sliced && c->prob_list_words != 0 && c->prob_prev_valid && c->prob16_valid && c->d_prob16 != nullptr;
    with_half = with_half && !c->sliced;
    if (with_half) DMX_TRY(ensure_prob16(c));
    c->prob16_valid = false;
    TimerSpan ev{nullptr, nullptr};
    SpanGuard ev_guard{c, &ev};
    timer_begin(c, DMX_T_PSTEP, &ev);
    const long long v0 = c->sliced ? c->cut[c->rank] : 0, v1 = c->sliced ? c->cut[c->rank + 1] : c->V;
    if (c->emulated && c->sliced && !c->emu_table_filled) {
        // emulated wire: nobody fills the other ranks' slices of genotype_prob; they hold the table without addition, so
        // that the E-step's rows are what an E-step sees (the posteriors decide which M-step kernel runs)
        for (int r = 0; r < c->nranks; r++)
            if (r != c->rank)
                HIP_TRY(dmx::launch_probs_from_betas(c->stream, c->d_prior, nullptr, c->d_v2snp, c->d_snp_ptr, c->d_snp_vars, c->cut[r],
                                                     c->cut[r + 1] - c->cut[r], -1LL, c->G, c->d_prow, lo, hi, c->d_prob));
        c->emu_table_filled = true;
    }
    HIP_TRY(dmx::launch_probs_from_betas(c->stream, c->d_prior, with_addition ? c->d_add : nullptr, c->d_v2snp,
                                         c->d_snp_ptr, c->d_snp_vars, v0, v1 - v0, c->sliced ? -1LL : (long long)c->S, c->G, c->d_prow, lo, hi,
                                         c->d_prob, (with_half || half_rows) ? c->d_prob16 : nullptr));
    c->prob16_valid = with_half;
    timer_end(c, DMX_T_PSTEP, ev);
    if (c->sliced) {  // everybody gets everybody's slice of genotype_prob
        timer_begin(c, DMX_T_ALLREDUCE, &ev);
        const size_t block = (size_t)c->slice_rows * c->G;
        float *mine = c->d_prob + (size_t)c->rank * block;
        int rc = 0;
        bool whole = true;
        if (c->prob_list_words != 0 && c->prob_prev_valid) {
            // Compact form (kernels.hip: k_prob_changes_build): only the rows of this rank's slice that changed since it sent them; every
            // rank reads every rank's count behind the all-gather of the lists - one host synchronisation - and all take the whole
            // slices when a list overflowed.  The receivers' copies are what was sent last, so the table keeps its bits.
            // (the lists are sized from the last exchange's counts, as the posteriors' are: dmx_exchange.cpp, gather_posteriors)
            const unsigned cap_now = std::max(1u, std::min(c->prob_cap_now, c->prob_list_cap));
            const size_t words_now = 4 + (size_t)cap_now * (size_t)(1 + c->G);
            HIP_TRY(dmx::launch_prob_changes_build(c->stream, mine, c->d_prob_prev, c->slice_rows, c->G, cap_now,
                                                   c->d_prob_list + (size_t)c->rank * words_now,
                                                   c->emulated ? c->d_prob_list : nullptr, (unsigned long long)words_now, c->nranks, c->rank));
            rc = coll_all_gather(c, (float *)c->d_prob_list, words_now, "changed rows of genotype_prob");
            if (rc == 0) {
                // (the listed rows are written while the host polls for the counts: should a list have overflowed, the whole slices overwrite them)
                const unsigned seq = ++c->list_seq;
                HIP_TRY(dmx::launch_post_counts(c->stream, c->d_prob_list, (unsigned long long)words_now, c->nranks, c->h_prob_counts, seq));
                HIP_TRY(dmx::launch_prob_changes_apply(c->stream, c->d_prob, c->d_prob_list, (unsigned long long)words_now, c->slice_rows, c->G,
                                                       c->nranks, c->rank, cap_now, half_rows ? (unsigned short *)c->d_prob16 : nullptr));
                DMX_TRY(wait_counts(c, c->h_prob_counts, c->nranks, seq));
                unsigned longest = 0;
                for (int r = 0; r < c->nranks; r++) longest = std::max(longest, c->h_prob_counts[r]);
                whole = longest > cap_now;
                if (std::getenv("DEMUXALOT_AMD_EXCHANGE_TRACE")) std::fprintf(stderr, "[table lists] longest %u capacity %u of %u\n", longest, cap_now, c->prob_list_cap);
                c->prob_cap_now = whole ? c->prob_list_cap : (unsigned)std::min<unsigned long long>(c->prob_list_cap, 4ull * longest + 512ull);
                if (!whole) {
                    c->prob_compact_taken++;
                    c->prob16_valid = half_rows;
                } else {
                    c->prob_compact_overflows++;
                }
            }
        } else if (c->prob_list_words != 0) {  // the first table of a layout (or behind a table somebody else wrote): what is sent now is what the others hold
            HIP_TRY(hipMemcpyAsync(c->d_prob_prev, mine, sizeof(float) * block, hipMemcpyDeviceToDevice, c->stream));
            c->prob_prev_valid = true;
        }
        if (rc == 0 && whole) rc = coll_all_gather(c, c->d_prob, block, "genotype_prob");
        timer_end(c, DMX_T_ALLREDUCE, ev);
        if (rc) return rc;
    }
    c->have_probs = true;
    c->p_clip_lo = lo;
    c->dict_candidate = !with_addition || c->add_is_zero;
    return 0;
}

// Dictionary form of the E-step (estep_dict.hip): distinct values per row of the current genotype table.  Returns
// the form to run in *form (DMX_FORM_DIRECT when some row does not fit or the form does not exist for the shape).
int prepare_dictionary(dmx_ctx *c, bool pairs, dmx::EstepArgs &a, int *form)
{
    *form = DMX_FORM_DIRECT;
    a.dict_n = 0;
    c->dict_distinct = 0;
    const bool wanted = c->dict_mode == 2 || (c->dict_mode == 1 && c->dict_candidate);
    if (!wanted || c->estep_mode == DMX_ESTEP_FAST || c->B == 0 || c->prob_rows == 0) return 0;  // (guarded: exact and faster)
    const int G = c->G;
    const long long K = c->K, rows = c->prob_rows;
    const bool block_form = pairs && K > dmx::DICT_LANE_K;  // wide doublet tables: workgroup per barcode
    if (a.call_rows == nullptr) return 0;  // (dmx_set_lean_memory: the form's row array was released)
    if (!block_form && (K > dmx::DICT_LANE_K || rows >= (1 << 24) || a.pairs_bytes == 0)) return 0;  // singlet tables beyond 256: the direct forms; 24-bit row x pitch; 32-bit record offsets
    if (block_form && (size_t)G * 72 + 9 * 1024 > 160 * 1024) return 0;  // the code rows of a chunk must fit the LDS
    if (G > 1024) return 0;  // widest k_build_dict instantiation (ensure_options refuses such runs anyway)
    if (c->dict_mode == 1 && !block_form) {
        // Where the lane form pays (measured): singlet runs with enough barcodes for several rounds of
        // wavefronts.  A launch of one round lasts as long as its longest barcode, whose calls this form walks in
        // batches with a memory latency each (20k x 10k x 64: 0.31 ms against 0.25 ms direct), and the 16 entry slots of
        // a doublet run leave two calls per barcode and batch (20k x 20k x 8 with doublets: 0.60 against 0.28 ms).
        const long long lanes = K <= 16 ? 4 : K <= 32 ? 8 : K <= 64 ? 16 : K <= 128 ? 32 : 64;
        if (pairs || c->B * lanes / 64 < 8192) return 0;
    }
    const size_t code_pitch = (size_t)dmx::dict_code_pitch(G);
    if ((size_t)rows > c->cap_dict_rows) {
        dev_free(c, &c->d_dict, c->cap_dict_rows * dmx::DICT_CAP);
        dev_free(c, &c->d_codes, c->cap_dict_rows * code_pitch);
        c->cap_dict_rows = 0;
        DMX_TRY(dev_alloc(c, &c->d_dict, (size_t)rows * dmx::DICT_CAP));
        DMX_TRY(dev_alloc(c, &c->d_codes, (size_t)rows * code_pitch));
        c->cap_dict_rows = (size_t)rows;
    }
    if (!c->d_dict_stat) DMX_TRY(dev_alloc(c, &c->d_dict_stat, (size_t)1));
    HIP_TRY(dmx::launch_build_dict(c->stream, c->d_prob, rows, G, c->d_dict, c->d_codes, c->d_dict_stat));
    unsigned distinct = 0;
    HIP_TRY(hipMemcpyAsync(&distinct, c->d_dict_stat, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->dict_distinct = (int)distinct;
    if (distinct == 0 || (int)distinct > (pairs ? dmx::DICT_PAIR_CAP : dmx::DICT_CAP)) return 0;
    if (block_form) {
        a.dict_n = (int)distinct;
        a.dict = c->d_dict;
        a.codes = c->d_codes;
        *form = DMX_FORM_DICT_BLOCK;
        return 0;
    }
    const size_t pitch = (size_t)dmx::dict_table_pitch((int)distinct, (int)K, pairs), need_bytes = (size_t)rows * pitch;
    if (need_bytes >= (1ull << 32)) return 0;  // buffer addressing
    if (need_bytes > c->cap_dtab) {
        dev_free(c, &c->d_dtab, c->cap_dtab);
        c->cap_dtab = 0;
        DMX_TRY(dev_alloc(c, &c->d_dtab, need_bytes));
        c->cap_dtab = need_bytes;
    }
    HIP_TRY(dmx::launch_pack_rows(c->stream, c->d_dict, c->d_codes, c->d_pairs, rows, G, (int)K, pairs, (int)distinct, c->d_dtab));
    a.dict_n = (int)distinct;
    a.dtab = c->d_dtab;
    a.dtab_pitch = (int)pitch;
    a.dtab_bytes = (unsigned)need_bytes;
    *form = DMX_FORM_DICT;
    return 0;
}

// logits_kept: somebody can read this E-step's logits (it is the last one of the call); else the next E-step of the same call
// overwrites them, and the guarded mode may take the coarse pass (kernels.hip: k_estep_tiled_coarse)
int run_estep(dmx_ctx *c, int with_doublets, bool with_prior, int prior_dtype, float power, bool logits_kept)
{
    dmx::EstepArgs a;
    a.pair_ptr = c->d_pair_ptr;
    a.order = c->d_bc_order;
    a.pairs = c->d_call_pairs;
    a.call_rows = c->d_call_rows;
    const unsigned long long rec_bytes = ((unsigned long long)c->n_pairs + dmx::CALL_PAD_PAIRS) * sizeof(dmx::CallPair);
    a.pairs_bytes = rec_bytes < (1ull << 32) ? (unsigned)rec_bytes : 0u;
    a.prob = c->d_prob;
    a.prob16 = nullptr;
    a.guard_accum = 0.0f;
    a.guard_alt_per_call = 0.0f;
    a.guard_alt_accum = 0.0f;
    a.guard_main_coarse = 0;
    a.opt_pairs = c->d_pairs;
    a.pair_blocks = with_doublets ? c->d_pair_blocks : nullptr;
    a.n_pair_blocks = with_doublets ? c->n_pair_blocks : 0;
    a.sum_plan = c->d_sum_plan;
    a.sum_plan_values = c->sum_plan_values;
    a.pen = c->d_pen;
    a.prior = with_prior ? c->d_prior_logits : nullptr;
    a.prior_dtype = prior_dtype;
    a.logits = c->d_logits;
    a.post = c->d_post;
    // variant-sharded M-step: the posteriors' codes / bitmaps / singlet columns go into this rank's block of the global tables
    const size_t row_base = c->mshard ? (size_t)c->rank * (size_t)c->rows_pad : 0;
    a.nz = c->mshard ? c->d_nz_g + row_base * ((c->G + 63) / 64) : c->d_nz;
    a.first = c->G <= 64 ? (c->mshard ? c->d_first_g + row_base : c->d_first) : nullptr;
    a.post_singlets = c->mshard ? c->d_post_g + row_base * c->G : nullptr;
    c->post_gathered = false;
    a.dense_calls = c->G <= 64 ? c->d_dense_calls : nullptr;
    // (the slots are zero: set at the install, left so by k_sum_dense at the end of every E-step that used them)
    c->dense_stat_valid = a.dense_calls != nullptr;
    a.nz_floor = power == 2.0f ? dmx::NZ_FLOOR_SQUARE : 0.0f;
    c->nz_floor = a.nz_floor;
    a.B = c->B;
    a.prob_bytes = (unsigned)((unsigned long long)c->prob_rows * c->G * 4ull);
    a.G = c->G;
    a.K = c->K;
    // Guarded mode: the tolerance-mode kernels wherever a lane-per-option one exists (estep_epilogue.h: estep_guard), the
    // exact mode for the workgroup-per-barcode shapes
    // (the workgroup-per-barcode forms - option tables beyond 1024, doublet tables beyond 256 - evaluate the guard in
    // k_softmax_rows from the logits alone, which does not cover prior logits: with a prior they run the exact mode)
    const bool block_shape = c->K > 1024 || (with_doublets && c->K > 256);
    const bool guarded = c->estep_mode == DMX_ESTEP_GUARDED && !(block_shape && with_prior);
    a.fast = c->estep_mode == DMX_ESTEP_FAST || guarded;
    a.guard = 0;
    a.guard_per_call = 7.0e-8f;  // estep_epilogue.h: GUARD_PER_CALL (launch_estep raises it for the form with pre-scaled rows)
    a.guard_count = c->d_guard_count;
    a.guard_list = c->d_guard_list;
    a.guard_sub = c->d_guard_sub;
    a.guard_sub_cap = c->guard_sub_cap;
    a.order_count = nullptr;
    a.direct = nullptr;
    a.order_direct = nullptr;
    a.segs = c->n_segs > 0 && c->K <= 1024 ? c->d_segs : nullptr;
    a.n_segs = c->n_segs;
    a.n_split = c->n_split;
    a.split_first = c->d_split_first;
    a.seg_sums = c->d_seg_sums;
    c->guard_ran = false;
    a.tiled = c->tiled_estep;
    a.n_bins = c->tiled_estep ? c->n_bins : 0;
    a.bin_rows_cap = c->bin_rows_cap;
    a.bin_order = c->d_bin_order;
    a.bin_rows = c->d_bin_rows;
    a.bin_ptr = c->d_bin_ptr;
    a.tile_stream = c->d_tile_stream;
    a.coarse_stream = nullptr;
    a.coarse_bin_ptr = nullptr;
    a.log2_keep = nullptr;
    a.n_long = 0;
    a.dict_n = 0;
    a.dtab = nullptr;
    a.dtab_bytes = 0;
    a.dtab_pitch = 0;
    a.dict = nullptr;
    a.codes = nullptr;
    if (c->lean_memory && with_doublets && c->d_tile_stream != nullptr && !c->coarse_ready) {
        // (dmx_set_lean_memory: the tile-major schedule is the singlet runs'; a run with doublets never reads its stream - 6.4 GB of configs[4])
        dev_free(c, &c->d_tile_stream, (size_t)c->n_pairs);
        a.tile_stream = nullptr;
    }
    TimerSpan ev{nullptr, nullptr};
    SpanGuard ev_guard{c, &ev};
    timer_begin(c, DMX_T_ESTEP, &ev);
    int form = DMX_FORM_DIRECT;
    // The dictionary form is exact and faster than the fine pass, but not than the COARSE pass (200k x 100k x 64: 1.1 ms with its
    // dictionary build against 0.75): an E-step whose logits nobody reads - the first of a dmx_em call of several iterations - takes
    // the coarse pass like the ones behind it (its records are built here instead of one E-step later).
    // dmx_set_lean_memory has released the tile-major stream: the tolerance kernels of this E-step walk the coarse pass's records where they
    // exist (k_estep_tiled_fine8: the float32 table, float64 sums), else the barcode-major ones (n_bins = 0 where they launch)
    const bool fine8 = c->d_tile_stream == nullptr && c->coarse_ready && !with_doublets && a.n_bins > 0 && c->K > 16 && c->K <= 128;
    const bool coarse_first = c->estep_mode == DMX_ESTEP_GUARDED && !logits_kept && c->guard_adaptive && coarse_capable(c, with_doublets, c->p_clip_lo) &&
                              a.n_bins > 0 && c->dict_mode == 1;
    if (!coarse_first) DMX_TRY(prepare_dictionary(c, with_doublets != 0, a, &form));  // part of the E-step's time
    if (form == DMX_FORM_DICT)
        HIP_TRY(dmx::launch_estep_dict(c->stream, a, with_doublets != 0));
    else if (form == DMX_FORM_DICT_BLOCK)
        HIP_TRY(dmx::launch_estep_dict_block(c->stream, a));
    else {
        // Several option slots per lane (estep_packed.hip) make a barcode's serial walk `slots` times longer, and a launch
        // lasts at least as long as its longest barcode.  So the barcodes with more calls than a third of what a SIMD
        // gets on average (counted by the repack) walk on 64 lanes inside the same launch; when that is more than an
        // eighth of them the problem is one of few, long rows and the direct form takes it.  20k x 20k x 8 with
        // doublets (longest row 3 500 calls): all packed 0.72 ms, split at 1 000 / 2 000 rows 0.32 / 0.30 ms, direct
        // 0.28 ms - the wavefronts of a launch that fits the chip at once stay where they were placed, the heaviest
        // 64-lane walks next to the heaviest packed ones; see DESIGN.md 4.1c.  Mode 2: every barcode packed; 3: the split
        // wherever the shape exists.
        int lanes = 0, slots = 0;
        // (guarded mode: where the packed form is taken it is exact AND faster than the tolerance-mode kernel on 64 lanes -
        // 200k x 20k x 8 with doublets: 1.94 against 2.08 ms -, so it runs as it is, without guard)
        bool packed = c->estep_packing && with_doublets && c->estep_mode != DMX_ESTEP_FAST && a.pairs_bytes && dmx::estep_packed_shape(a.K, a.G, &lanes, &slots);
        if (packed && c->estep_packing != 2) {
            const int k = lanes == 8 ? 0 : lanes == 16 ? 1 : 2;
            a.n_long = c->max_row_calls > 0 ? c->n_long_rows[k] : c->B;  // no statistic (host-packed problem): not packed
#ifdef DMX_EXPERIMENTS  // experiment builds only (make EXPERIMENTS=1)
            if (const char *e = std::getenv("DEMUXALOT_AMD_PACKED_LONG")) a.n_long = std::min<long long>(c->B, std::max(0ll, atoll(e)));
#endif
            if (c->estep_packing == 1 && 8 * a.n_long > c->B) packed = false;
            if (!packed) a.n_long = 0;
        }
        if (packed) {
            a.fast = 0;
            HIP_TRY(dmx::launch_estep_packed(c->stream, a));
            form = DMX_FORM_PACKED;
        } else if (guarded) {
            // fast kernels with the guard evaluated per barcode, then the exact kernel over the barcodes they queued (their
            // number is only known on the device: a launch sized for all of them, the wavefronts past the queue's end
            // return at once); the redo rewrites logits, posteriors, bitmaps and codes of those barcodes.  Adaptive
            // (kernels.hip: k_guard_begin): the passes are timed on the device, and an E-step for which pass + redo would cost
            // more than the exact kernel over every barcode runs that kernel directly - the fast kernels stand back.
            // The coarse pass (kernels.hip: k_estep_tiled_coarse; singlets, 17 .. 128 genotypes, the tile-major schedule, a P-step's
            // table whose clip keeps binary16 normal) is admissible when nobody can read this E-step's logits.  Which of coarse pass,
            // fine pass and the direct form runs is the device's choice (k_guard_begin): both fast launches are issued, the one
            // that is not taken stands back.
            const bool capable = coarse_capable(c, with_doublets, c->p_clip_lo) && a.n_bins > 0;
            const bool allow_coarse = capable && (!logits_kept || c->coarse_pass == 2);
            if (allow_coarse && !c->coarse_ready) {
                // once per problem, ahead of k_guard_begin's time stamp (not part of the pass the device times): the coarse pass's
                // records - 8 bytes per call where the tile-major stream has 16 - and the log2 of the keep factors per barcode
                const int cpg = dmx::coarse_calls_per_gather((int)c->K), bpr = dmx::coarse_batches_per_record(cpg);
                const size_t words = (((size_t)c->n_pairs / 4 + (size_t)c->n_bins * (bpr - 1)) / bpr + 1) * (size_t)(cpg * 16);
                DMX_TRY(dev_alloc(c, &c->d_coarse_stream, words));
                c->cap_coarse_stream = words;
                DMX_TRY(dev_alloc(c, &c->d_coarse_bin_ptr, (size_t)c->n_bins + 1));
                DMX_TRY(dev_alloc(c, &c->d_log2_keep, (size_t)c->B));
                // (the barcodes' sums of log2 keep come out of the same pass: every call's keep factor is read there once)
                HIP_TRY(dmx::launch_build_coarse_stream(c->stream, c->d_tile_stream, c->d_bin_ptr, c->n_bins, a.prob_bytes, cpg, c->d_coarse_bin_ptr, c->d_coarse_stream,
                                                        c->d_bin_rows, c->bin_rows_cap, c->d_log2_keep));
                c->coarse_ready = true;
                // dmx_set_lean_memory: the tile-major stream has done its last job (the block returns to the context's cache behind the build,
                // stream-ordered); the fine level is the barcode-major tolerance kernel from here on (below)
                if (c->lean_memory) {
                    dev_free(c, &c->d_tile_stream, (size_t)c->n_pairs);
                    // ... and the compact row array of the dictionary form with it (4 bytes per call): an E-step that keeps its logits on the
                    // prior table then runs the tolerance kernel too, the incremental M-step reads the rows from the records
                    dev_free(c, &c->d_call_rows, ((size_t)c->n_pairs + dmx::CALL_PAD_PAIRS) * 2);
                    a.call_rows = nullptr;
                }
            }
            if (allow_coarse) DMX_TRY(ensure_prob16(c));
            HIP_TRY(dmx::launch_guard_begin(c->stream, c->d_guard_count, c->B, c->K, c->guard_adaptive, capable, allow_coarse));
            a.guard = 1;
            if (fine8) a.guard_per_call = dmx::guard_per_call_fine8(dmx::coarse_calls_per_gather((int)c->K));  // (the coarse guard's estimate of the fine level reads it too)
            a.order_direct = c->d_bc_order;
            a.guard_main_coarse = 0;
            a.guard_alt_per_call = capable ? dmx::GUARD_PER_CALL_COARSE : 0.0f;
            a.guard_alt_accum = capable ? dmx::GUARD_ACCUM_F32 : 0.0f;
            if (allow_coarse) {
                if (!c->prob16_valid) {  // (the P-step of a dmx_em / dmx_run_iterations call has written it already)
                    // a sliced run whose slices travel as lists of changed rows keeps the binary16 table up to date row by row from here on
                    // (run_pstep): converted whatever level the device takes, so that the host knows it valid
                    const bool kept = c->sliced && c->prob_list_words != 0;
                    HIP_TRY(dmx::launch_prob_to_half(c->stream, c->d_prob, c->prob_rows, c->G, c->d_prob16, kept ? nullptr : c->d_guard_count + dmx::GS_SKIP_COARSE));
                    c->prob16_valid = kept;
                }
                dmx::EstepArgs coarse = a;
                coarse.prob16 = c->d_prob16;
                coarse.coarse_stream = c->d_coarse_stream;
                coarse.coarse_bin_ptr = c->d_coarse_bin_ptr;
                coarse.log2_keep = c->d_log2_keep;
                coarse.guard_per_call = dmx::GUARD_PER_CALL_COARSE;
                coarse.guard_accum = dmx::GUARD_ACCUM_F32;
                coarse.guard_main_coarse = 1;
                coarse.guard_alt_per_call = a.guard_per_call;
                coarse.guard_alt_accum = 0.0f;
                coarse.direct = c->d_guard_count + dmx::GS_SKIP_COARSE;
                HIP_TRY(dmx::launch_estep(c->stream, coarse, false));
            }
            a.direct = c->d_guard_count + dmx::GS_SKIP_FINE;
            a.tile_stream = c->d_tile_stream;
            if (c->d_tile_stream == nullptr) {  // (released: the fine level walks the coarse pass's records, or a barcode per wavefront the barcode-major ones)
                if (c->coarse_ready && !with_doublets && a.n_bins > 0 && c->K > 16 && c->K <= 128) {
                    a.coarse_stream = c->d_coarse_stream;
                    a.coarse_bin_ptr = c->d_coarse_bin_ptr;
                    a.log2_keep = c->d_log2_keep;
                    a.prob16 = nullptr;
                    a.guard_per_call = dmx::guard_per_call_fine8(dmx::coarse_calls_per_gather((int)c->K));  // (also where the release came with this E-step)
                } else {
                    a.n_bins = 0;
                }
            }
            HIP_TRY(dmx::launch_estep(c->stream, a, with_doublets != 0));
            a.coarse_stream = nullptr;  // (the redo below is the exact kernel's)
            HIP_TRY(dmx::launch_guard_compact(c->stream, c->d_guard_count, c->d_guard_sub, c->guard_sub_cap, c->d_guard_list, c->d_bc_order, c->B));
            dmx::EstepArgs redo = a;
            redo.direct = c->d_guard_count + dmx::GS_DIRECT;
            redo.fast = 0;
            redo.guard = 2;
            redo.n_bins = 0;
            redo.order = c->d_guard_list;
            redo.order_count = c->d_guard_count + dmx::GS_COUNT;
            HIP_TRY(dmx::launch_estep(c->stream, redo, with_doublets != 0));
            c->guard_rows_total += c->B;
            c->guard_ran = true;
        } else {
            if (c->d_tile_stream == nullptr) {
                if (fine8 && a.fast) {  // (the tolerance mode without the guard)
                    a.coarse_stream = c->d_coarse_stream;
                    a.coarse_bin_ptr = c->d_coarse_bin_ptr;
                    a.log2_keep = c->d_log2_keep;
                } else {
                    a.n_bins = 0;
                }
            }
            HIP_TRY(dmx::launch_estep(c->stream, a, with_doublets != 0));
        }
    }
    c->estep_form = form;
    if (a.dense_calls) HIP_TRY(dmx::launch_sum_dense(c->stream, c->d_dense_calls, c->guard_ran ? c->d_guard_count : nullptr));
    else if (c->guard_ran) HIP_TRY(dmx::launch_guard_stamp(c->stream, c->d_guard_count, dmx::GS_T_END));
    timer_end(c, DMX_T_ESTEP, ev);
    c->have_post = true;
    c->logits_readable = logits_kept;  // (an E-step nobody was to read the logits of may have taken the coarse pass: the device's choice)
    return 0;
}

int run_mstep(dmx_ctx *c, float power)
{
    const bool mshard = c->mshard;
    const size_t row_base = mshard ? (size_t)c->rank * (size_t)c->rows_pad : 0;
    const int Wn = (c->G + 63) / 64;
    dmx::MstepArgs a;
    a.order = c->d_item_order;
    a.item_start = c->d_item_start;
    a.item_len = c->d_item_len;
    a.calls = c->d_csc;
    // variant-sharded: the barcodes of all ranks (global rows), singlet posteriors only (row stride G)
    a.post = mshard ? c->d_post_g : c->d_post;
    a.nz = mshard ? c->d_nz_g : c->d_nz;
    a.first = mshard ? c->d_first_g : c->d_first;
    const unsigned long long rows = mshard ? (unsigned long long)c->rows_total : (unsigned long long)c->B;
    a.K = mshard ? c->G : c->K;
    a.first_bytes = 8ull * rows;
    a.wide = c->mstep_wide;
    a.post_bytes = rows * (unsigned long long)a.K * 4ull;
    a.partial = c->d_partial;
    a.n_items = c->n_items;
    a.G = c->G;
    a.square = (power == 2.0f);
    a.power = power;
    a.dense_calls = c->dense_stat_valid && a.post_bytes < (1ull << 32) ? c->d_dense_calls : nullptr;
    a.total_calls = 2ull * (unsigned long long)c->n_pairs;
    if (!a.square && c->nz_floor != 0.0f) {
        // the E-step assumed a squaring M-step: rebuild the bitmap with the exact `!= 0` rule
        HIP_TRY(dmx::launch_rebuild_nz(c->stream, c->d_post, c->B, c->K, c->G, 0.0f, (mshard ? c->d_nz_g : c->d_nz) + row_base * Wn,
                                       c->G <= 64 ? (mshard ? c->d_first_g : c->d_first) + row_base : nullptr));
        c->nz_floor = 0.0f;
        c->post_gathered = false;
    }
    DMX_TRY(gather_posteriors(c));
    c->add_is_zero = false;
    TimerSpan ev{nullptr, nullptr};
    SpanGuard ev_guard{c, &ev};
    const bool dist = c->attached();  // also with one rank: keeps the collective path testable on one GPU
    unsigned long long *redo = c->exact_additions ? c->d_redo : nullptr;
    a.item_variant = nullptr;
    a.item_ptr = c->d_item_ptr;
    a.prow = nullptr;
    a.out32 = nullptr;
    a.out64 = nullptr;
    const bool f64 = c->reduce_dtype == DMX_F64;
    // where k_mcombine writes: the variants of one work item are written there by the M-step kernels themselves
    a.item_variant = c->d_item_variant;
    a.redo_cap = c->cap_redo;
    a.fixed_shift_v = nullptr;
    a.fixed_acc64 = nullptr;
    a.fixed_state = nullptr;
    a.incr_total = 0ull;
    // tile-major form (kernels.h: MTileArgs): sums in any order, so not with the exact additions; built on first use
    a.tiles_done = false;
    dmx::MTileArgs tiles{};
    // Building the records (a sort of the calls: 2.6 ms on 200k x 100k x 64, where an M-step + combine then takes 0.34 instead of
    // 0.70 ms) pays from MSTEP_TILES_PAY M-steps on: taken when that many are still to come - in the running dmx_em /
    // dmx_run_iterations call, or as the caller announced (dmx_set_msteps_expected) -, or the problem has seen that many
    // already (somebody iterates call by call), or always (dmx_set_mstep_tiles(ctx, 2)).
    // Under the INCREMENTAL M-step only the full passes cost anything, and since round 6 the work items can make them with the tile
    // form's arithmetic (fixed_items below: 0.71 instead of 0.33 ms, no records): the records then pay only where full passes keep
    // coming - a workload whose posteriors keep moving.  So a context that can go incremental starts on the work items and reads the
    // device's count of full passes ONCE at its 4th, 16th and 64th M-step (a 4-byte download: the only host synchronisation of the
    // policy); three full passes in the first four M-steps, or half of them later, and the records are built as before.  A converging
    // 25-iteration call: M-steps 0.71 + 23 x 0.03 ms instead of 2.6 (build) + 0.33 + 23 x 0.03.
    constexpr int MSTEP_TILES_PAY = 8;
    const long long ahead = std::max<long long>(c->msteps_ahead, c->msteps_expected);
    // (a rank that exchanges SUMS - reduce-scatter of the partial sums of its own barcodes - is one context with all calls of its barcodes
    // too: its partial sums stay in the exchange buffer between two M-steps, the delta pass updates the rows it touched.  Not the all-reduce,
    // which sums in place.)
    const bool own_sums = !mshard && (!dist || c->sliced);
    const bool can_go_incremental = c->mstep_incremental && c->mstep_tiles == 1 && !c->exact_additions && c->G <= 64 && c->n_csc > 0 && power > 0.0f &&
                                    own_sums && c->d_call_pairs != nullptr && c->d_item_variant != nullptr;
    if (can_go_incremental && !c->incr_heavy && c->n_mt == 0 && c->d_incr_state != nullptr &&
        (c->msteps_done == 4 || c->msteps_done == 16 || c->msteps_done == 64)) {
        unsigned full_passes = 0;
        HIP_TRY(hipMemcpyAsync(&full_passes, c->d_incr_state + 2 * dmx::IS_WORDS + 3, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->incr_heavy = c->msteps_done == 4 ? full_passes >= 3u : 2ull * full_passes >= (unsigned long long)c->msteps_done;
    }
    const bool tiles_wanted = c->mstep_tiles == 2 || (c->mstep_tiles == 1 && (c->n_mt > 0 || ((ahead >= MSTEP_TILES_PAY || c->msteps_done >= MSTEP_TILES_PAY) &&
                                                                                            (!can_go_incremental || c->incr_heavy))));
    if (c->msteps_expected > 0) c->msteps_expected--;
    c->msteps_done++;
    if (!c->exact_additions && tiles_wanted && c->G <= 64 && c->n_csc > 0 && power > 0.0f) {  // (power > 0: contributions in [0, 1])
        if (!c->mt_tried) {
            HIP_TRY(hipStreamSynchronize(c->stream));  // (the build synchronises anyway; this makes its wall time its own)
            const auto t0 = std::chrono::steady_clock::now();
            DMX_TRY(dmx::build_mstep_tiles(c, mshard ? c->cut[c->rank] : 0, mshard ? c->cut[c->rank + 1] : c->V));
            c->mt_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
        if (c->n_mt > 0) {
            tiles.stream = c->d_mt_stream;
            tiles.ptr = c->d_mt_ptr;
            tiles.first = c->d_mt_first;
            tiles.order = c->d_mt_order;
            tiles.shift = c->d_mt_shift;
            tiles.n_tiles = c->n_mt;
            tiles.tv = c->mt_tv;
            a.tiles_done = true;
        }
    }
    if (!dist || mshard) {
        a.out32 = c->d_add;
    } else if (c->sliced) {
        a.prow = c->d_prow;
        if (f64) a.out64 = (double *)c->d_exch;
        else a.out32 = (float *)c->d_exch;
    } else {
        if (f64) a.out64 = c->d_add64;
        else a.out32 = c->d_add;
    }
    // Fixed-point WORK-ITEM form (kernels.h: MstepArgs::fixed_shift_v): where the tile-major records are not there - a call too short
    // to pay for their sort, learn_genotypes' default of 5 iterations among them - the work items add the tile-major form's integers
    // with the tile cut's exponents (plan_mstep_shifts: the host's cut, no sort), so that their sums are the tile-major form's bit for
    // bit and the incremental M-step builds on them: one full pass of 0.7 ms, then delta passes, instead of 0.7 ms per M-step.
    // (dmx_set_mstep_tiles(ctx, 0) or dmx_set_mstep_incremental(ctx, 0): the float64 work-item form, as before.)
    bool fixed_items = !a.tiles_done && c->mstep_tiles != 0 && c->mstep_incremental && !c->exact_additions && c->G <= 64 && c->n_csc > 0 &&
                       power > 0.0f && own_sums && c->d_call_pairs != nullptr && c->d_item_variant != nullptr;
    if (fixed_items) {
        DMX_TRY(dmx::plan_mstep_shifts(c));
        fixed_items = c->d_mt_shift_v != nullptr;
    }
    // Incremental form (kernels.h: MIncrArgs): one context with all calls of its barcodes, the tiles' per-variant exponents at hand.
    // ... or a variant-sharded rank with the tile-major records of its slice (round 6): the same sums over the barcodes of ALL ranks, the
    // changed barcodes found in the gathered tables, the delta pass a masked walk of the slice's variant-major records (MIncrArgs::changed_map).
    const bool sharded_incr = a.tiles_done && c->mstep_incremental == 1 && mshard && c->d_mt_shift_v != nullptr && a.out32 == c->d_add && c->G <= 64 &&
                              c->d_item_variant != nullptr && c->rows_total > 0;
    const bool incremental = sharded_incr || ((a.tiles_done || fixed_items) && c->mstep_incremental && own_sums && c->d_mt_shift_v != nullptr &&
                                              c->d_call_pairs != nullptr);
    const long long incr_rows = sharded_incr ? c->rows_total : c->B;
    dmx::MIncrArgs incr{};
    if (incremental) {
        if (!c->d_acc64) {
            c->incr_rows = incr_rows;
            DMX_TRY(dev_alloc(c, &c->d_acc64, (size_t)c->V * c->G));
            DMX_TRY(dev_alloc(c, &c->d_prev_post, (size_t)incr_rows * c->G));
            DMX_TRY(dev_alloc(c, &c->d_prev_first, (size_t)incr_rows));
            DMX_TRY(dev_alloc(c, &c->d_incr_list, (size_t)incr_rows));
            if (sharded_incr) {
                DMX_TRY(dmx::build_slice_row_index(c));  // (the slice's records by barcode row; without it: the masked walk and its byte map)
                if (c->d_slice_rec == nullptr) {
                    DMX_TRY(dev_alloc(c, &c->d_incr_map, (size_t)incr_rows));
                    HIP_TRY(hipMemsetAsync(c->d_incr_map, 0, (size_t)incr_rows, c->stream));
                }
            }
            DMX_TRY(dev_alloc(c, &c->d_incr_touched, (size_t)c->V));
            DMX_TRY(dev_alloc(c, &c->d_incr_state, (size_t)(3 * dmx::IS_WORDS)));  // two alternating sets + the counters
            HIP_TRY(hipMemsetAsync(c->d_incr_state, 0, sizeof(unsigned) * 3 * dmx::IS_WORDS, c->stream));
            HIP_TRY(hipMemsetAsync(c->d_incr_touched, 0, (size_t)c->V, c->stream));
            c->incr_valid = false;
        }
        if (!c->incr_valid || c->incr_power != power) {  // (nothing to build on: zeroed state words ask for the full pass)
            HIP_TRY(hipMemsetAsync(c->d_incr_state, 0, sizeof(unsigned) * 2 * dmx::IS_WORDS, c->stream));
            if (c->mstep_incremental == 2 && !dist) {  // (measurement: the sums built from nothing by the delta pass instead of the full pass)
                const unsigned on[2] = {1u, 1u};
                HIP_TRY(hipMemsetAsync(c->d_acc64, 0, sizeof(unsigned long long) * (size_t)c->V * c->G, c->stream));
                HIP_TRY(hipMemsetAsync(c->d_prev_post, 0, sizeof(float) * (size_t)incr_rows * c->G, c->stream));
                HIP_TRY(hipMemsetAsync(c->d_prev_first, 0xFF, sizeof(uint2) * (size_t)incr_rows, c->stream));
                HIP_TRY(hipMemsetAsync(c->d_add, 0, sizeof(float) * (size_t)c->V * c->G, c->stream));
                HIP_TRY(hipMemcpyAsync(c->d_incr_state + dmx::IS_VALID, &on[0], sizeof(unsigned), hipMemcpyHostToDevice, c->stream));
                HIP_TRY(hipMemcpyAsync(c->d_incr_state + dmx::IS_FORCE, &on[1], sizeof(unsigned), hipMemcpyHostToDevice, c->stream));
                HIP_TRY(hipStreamSynchronize(c->stream));
            }
            c->incr_parity = 0;
            c->incr_valid = true;
            c->incr_power = power;
        }
        incr.state = c->d_incr_state + c->incr_parity * dmx::IS_WORDS;
        incr.next = c->d_incr_state + (c->incr_parity ^ 1) * dmx::IS_WORDS;
        c->incr_parity ^= 1;
        incr.counters = c->d_incr_state + 2 * dmx::IS_WORDS;
        incr.acc64 = c->d_acc64;
        incr.prev = c->d_prev_post;
        incr.prev_first = c->d_prev_first;
        incr.list = c->d_incr_list;
        incr.touched = c->d_incr_touched;
        incr.shift_v = c->d_mt_shift_v;
        incr.pairs = sharded_incr ? nullptr : c->d_call_pairs;
        incr.call_rows = sharded_incr ? nullptr : c->d_call_rows;
        incr.pair_ptr = sharded_incr ? nullptr : c->d_pair_ptr;
        incr.changed_map = sharded_incr ? c->d_incr_map : nullptr;  // (null with the row index)
        incr.rec = sharded_incr ? c->d_slice_rec : nullptr;
        incr.rec_ptr = sharded_incr && c->d_slice_rec != nullptr ? c->d_slice_ptr : nullptr;
        incr.row_variant = !sharded_incr && c->sliced ? c->d_row_variant : nullptr;
        incr.B = incr_rows;
        incr.V = c->V;
        incr.floor = dmx::mincr_floor(power);
        a.incr_total = !sharded_incr ? 0ull : incr.rec_ptr != nullptr ? (unsigned long long)c->n_csc : 2ull * (unsigned long long)incr_rows;
        tiles.acc64 = c->d_acc64;
        tiles.incr_state = incr.state;
        if (fixed_items) {
            a.fixed_shift_v = c->d_mt_shift_v;
            a.fixed_acc64 = c->d_acc64;
            a.fixed_state = incr.state;
        }
        c->mstep_incr_launches++;
    } else {
        c->incr_valid = false;  // (another form writes the addition: the kept sums no longer describe it)
    }
    timer_begin(c, DMX_T_MSTEP, &ev);
    if (sharded_incr) HIP_TRY(dmx::launch_mstep_incremental_sharded(c->stream, a, tiles, incr));
    else if (incremental && fixed_items) HIP_TRY(dmx::launch_mstep_items_incremental(c->stream, a, incr));
    else if (incremental) HIP_TRY(dmx::launch_mstep_incremental(c->stream, a, tiles, incr));
    else if (a.tiles_done) HIP_TRY(dmx::launch_mstep_tiles(c->stream, a, tiles));
    else HIP_TRY(dmx::launch_mstep(c->stream, a));
    c->mstep_form = a.tiles_done ? 2 : (incremental && fixed_items ? 3 : 1);
    timer_end(c, DMX_T_MSTEP, ev);
    if (!dist) {
        timer_begin(c, DMX_T_MCOMBINE, &ev);
        HIP_TRY(dmx::launch_mcombine(c->stream, a, c->d_item_ptr, 0, c->V, nullptr, c->d_add, nullptr, redo, c->d_n_redo, nullptr, true));
        timer_end(c, DMX_T_MCOMBINE, ev);
        return 0;
    }
    if (mshard) {
        // this rank's variant slice, summed over the barcodes of all ranks: final, exact, nothing to reduce
        timer_begin(c, DMX_T_MCOMBINE, &ev);
        HIP_TRY(dmx::launch_mcombine(c->stream, a, c->d_item_ptr, c->cut[c->rank], c->cut[c->rank + 1], nullptr, c->d_add, nullptr, redo,
                                     c->d_n_redo, nullptr, true));
        timer_end(c, DMX_T_MCOMBINE, ev);
        c->add_partial = c->nranks > 1;
        return 0;
    }
    int rc = 0;
    if (c->sliced) {
        // partial sums straight into the padded exchange buffer, reduce-scatter, this rank's slice rounded into d_add
        timer_begin(c, DMX_T_MCOMBINE, &ev);
        HIP_TRY(dmx::launch_mcombine(c->stream, a, c->d_item_ptr, 0, c->V, c->d_prow, f64 ? nullptr : (float *)c->d_exch,
                                     f64 ? (double *)c->d_exch : nullptr, redo, c->d_n_redo, nullptr, true));
        timer_end(c, DMX_T_MCOMBINE, ev);
        timer_begin(c, DMX_T_ALLREDUCE, &ev);
        const size_t block = (size_t)c->slice_rows * c->G;
        rc = coll_reduce_scatter(c, c->d_exch, c->d_recv, block, f64, c->stream);
        if (rc == 0)
            HIP_TRY(dmx::launch_store_slice(c->stream, c->d_recv, f64, c->cut[c->rank], c->cut[c->rank + 1] - c->cut[c->rank], c->G, c->d_add));
        timer_end(c, DMX_T_ALLREDUCE, ev);
        if (rc) return rc;
        c->add_partial = c->nranks > 1;
        return 0;
    }
    // SNPs with scattered variants: all-reduce of the dense sums, P-step on every rank
    timer_begin(c, DMX_T_MCOMBINE, &ev);
    HIP_TRY(dmx::launch_mcombine(c->stream, a, c->d_item_ptr, 0, c->V, nullptr, f64 ? nullptr : c->d_add, f64 ? c->d_add64 : nullptr, redo,
                                 c->d_n_redo, nullptr, true));
    timer_end(c, DMX_T_MCOMBINE, ev);
    timer_begin(c, DMX_T_ALLREDUCE, &ev);
    const size_t cnt = (size_t)c->V * c->G;
    if (f64) {
        rc = coll_all_reduce(c, c->d_add64, cnt, true);
        if (rc == 0) HIP_TRY(dmx::launch_f64_to_f32(c->stream, c->d_add64, c->d_add, (long long)cnt));
    } else {
        rc = coll_all_reduce(c, c->d_add, cnt, false);
    }
    timer_end(c, DMX_T_ALLREDUCE, ev);
    return rc;
}

}  // namespace host
}  // namespace dmx

extern "C" {

int dmx_set_addition(dmx_ctx *c, const float *addition)
{
    DMX_TRY(bind(c));
    DMX_TRY(need(c, c->have_problem, "dmx_set_problem before dmx_set_addition"));
    const size_t vg = (size_t)c->V * c->G;
    c->incr_valid = false;  // (the addition is no longer the last M-step's: the incremental M-step starts over)
    if (addition) {
        HIP_TRY(hipMemcpyAsync(c->d_add, addition, sizeof(float) * vg, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    } else {
        HIP_TRY(hipMemsetAsync(c->d_add, 0, sizeof(float) * (vg ? vg : 1), c->stream));
    }
    c->add_is_zero = addition == nullptr;
    c->add_partial = false;
    return 0;
}

int dmx_probs_from_betas(dmx_ctx *c, float lo, float hi, float *prob_out)
{
    DMX_TRY(bind(c));
    DMX_TRY(need(c, c->have_problem && c->have_betas, "dmx_set_problem + dmx_set_betas before dmx_probs_from_betas"));
    DMX_TRY(run_pstep(c, lo, hi, true));
    DMX_TRY(copy_prob_out(c, prob_out));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int dmx_set_probs(dmx_ctx *c, const float *prob)
{
    DMX_TRY(bind(c));
    DMX_TRY(need(c, c->have_problem, "dmx_set_problem before dmx_set_probs"));
    if (!prob && c->V > 0) return fail(DMX_ERR_INVALID, "null prob table");
    if (c->V) DMX_TRY(copy_prob_in(c, prob));
    // the E-step's log is the hot-path form (finite argument >= 1e-4): a table with entries outside [0, 1]
    // (or NaN) is refused rather than answered with numbers that mean nothing
    HIP_TRY(hipMemsetAsync(c->d_best, 0, sizeof(int), c->stream));
    HIP_TRY(dmx::launch_check_unit_range(c->stream, c->d_prob, c->prob_rows * c->G, c->d_best));
    int flag = 0;
    HIP_TRY(hipMemcpyAsync(&flag, c->d_best, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (flag) return fail(DMX_ERR_INVALID, "genotype_prob has entries outside [0, 1] (or NaN)");
    c->have_probs = true;
    c->p_clip_lo = 0.0f;  // (a caller's table: entries may lie below binary16's normal range - no coarse pass)
    c->prob16_valid = false;
    c->dict_candidate = true;
    return 0;
}

int dmx_probs_from_betas_f64(dmx_ctx *c, const double *betas, float lo, float hi, float *prob_out)
{
    DMX_TRY(bind(c));
    DMX_TRY(need(c, c->have_problem, "dmx_set_problem before dmx_probs_from_betas_f64"));
    const size_t vg = (size_t)c->V * c->G;
    if (!betas && vg) return fail(DMX_ERR_INVALID, "null betas");
    double *d_b = nullptr;
    HIP_TRY(hipMalloc((void **)&d_b, (vg ? vg : 1) * sizeof(double)));
    hipError_t e = vg ? hipMemcpyAsync(d_b, betas, vg * sizeof(double), hipMemcpyHostToDevice, c->stream) : hipSuccess;
    if (e == hipSuccess)
        c->prob_prev_valid = false;  // (every rank computes the whole table here)
        e = dmx::launch_probs_from_betas_f64(c->stream, d_b, c->d_v2snp, c->d_snp_ptr, c->d_snp_vars, c->V, c->S, c->G, c->d_prow, lo, hi, c->d_prob);
    int rc_copy = 0;
    if (e == hipSuccess && vg) rc_copy = copy_prob_out(c, prob_out);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(d_b);
    if (e != hipSuccess) return fail(DMX_ERR_HIP, "P-step from float64 betas: %s", hipGetErrorString(e));
    if (rc_copy) return rc_copy;
    c->have_probs = true;
    c->p_clip_lo = lo;
    c->prob16_valid = false;
    c->dict_candidate = true;
    return 0;
}

int dmx_estep(dmx_ctx *c, int with_doublets, const float *penalties, const void *prior_logits, int prior_dtype,
              float *logits_out, float *probs_out)
{
    DMX_TRY(bind(c));
    DMX_TRY(need(c, c->have_problem && c->have_probs, "genotype probabilities (dmx_probs_from_betas / dmx_set_probs) before dmx_estep"));
    if (!penalties) return fail(DMX_ERR_INVALID, "null penalties");
    DMX_TRY(ensure_options(c, with_doublets, penalties));
    DMX_TRY(upload_prior_logits(c, prior_logits, prior_dtype));
    DMX_TRY(run_estep(c, with_doublets, prior_logits != nullptr, prior_dtype, 2.0f));
    const size_t bk = (size_t)c->B * c->K;
    DMX_TRY(copy_out(c, logits_out, c->d_logits, bk));
    DMX_TRY(copy_out(c, probs_out, c->d_post, bk));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int dmx_mstep(dmx_ctx *c, float power, float *addition_out)
{
    DMX_TRY(bind(c));
    DMX_TRY(need(c, c->have_problem && c->have_post, "dmx_estep before dmx_mstep"));
    DMX_TRY(run_mstep(c, power));
    if (addition_out) DMX_TRY(ensure_full_addition(c));  // collective when sliced: all ranks pass it, or none does
    DMX_TRY(copy_out(c, addition_out, c->d_add, (size_t)c->V * c->G));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int dmx_em(dmx_ctx *c, int n_iterations, float lo, float hi, int with_doublets, const float *penalties,
           const void *prior_logits, int prior_dtype, float power, float *logits_out, float *probs_out,
           float *addition_out)
{
    DMX_TRY(bind(c));
    DMX_TRY(need(c, c->have_problem && c->have_betas, "dmx_set_problem + dmx_set_betas before dmx_em"));
    if (n_iterations < 1) return fail(DMX_ERR_INVALID, "n_iterations must be >= 1");
    if (!penalties) return fail(DMX_ERR_INVALID, "null penalties");
    DMX_TRY(ensure_options(c, with_doublets, penalties));
    DMX_TRY(upload_prior_logits(c, prior_logits, prior_dtype));
    const size_t vg = (size_t)c->V * c->G;
    HIP_TRY(hipMemsetAsync(c->d_add, 0, sizeof(float) * (vg ? vg : 1), c->stream));  // demux.py:86
    c->incr_valid = false;
    c->add_is_zero = true;
    c->add_partial = false;
    const bool keep_last = c->logits_needed || logits_out != nullptr;  // (dmx_set_logits_needed)
    for (int it = 0; it < n_iterations; it++) {
        const bool kept = it + 1 == n_iterations && keep_last;  // somebody can read this E-step's logits
        DMX_TRY(run_pstep(c, lo, hi, true, !kept && it > 0 && coarse_capable(c, with_doublets, lo)));  // (iteration 0: the dictionary form)
        DMX_TRY(run_estep(c, with_doublets, it == 0 && prior_logits != nullptr, prior_dtype, power, kept));
        if (it + 1 < n_iterations) {  // the M-step after the last yield is dead
            c->msteps_ahead = n_iterations - 1 - it;
            const int rc_m = run_mstep(c, power);
            c->msteps_ahead = 0;
            if (rc_m) return rc_m;
        }
    }
    const size_t bk = (size_t)c->B * c->K;
    DMX_TRY(copy_out(c, logits_out, c->d_logits, bk));
    DMX_TRY(copy_out(c, probs_out, c->d_post, bk));
    DMX_TRY(ensure_full_addition(c));  // (collective when sliced) the slices of the last M-step, on every rank
    DMX_TRY(copy_out(c, addition_out, c->d_add, vg));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int dmx_run_iterations(dmx_ctx *c, int n_iterations, float lo, float hi, float power)
{
    DMX_TRY(bind(c));
    DMX_TRY(need(c, c->have_problem && c->have_betas && c->have_post && c->K > 0,
                 "dmx_estep or dmx_em (to fix the options) before dmx_run_iterations"));
    if (n_iterations < 0) return fail(DMX_ERR_INVALID, "negative n_iterations");
    const int with_doublets = c->K != c->G;
    for (int it = 0; it < n_iterations; it++) {
        const bool kept = it + 1 == n_iterations && c->logits_needed;  // somebody can read this E-step's logits
        DMX_TRY(run_pstep(c, lo, hi, true, !kept && coarse_capable(c, with_doublets, lo)));
        DMX_TRY(run_estep(c, with_doublets, false, DMX_F32, power, kept));
        c->msteps_ahead = n_iterations - it;
        const int rc_m = run_mstep(c, power);
        c->msteps_ahead = 0;
        if (rc_m) return rc_m;
    }
    return 0;
}

}  // extern "C"
