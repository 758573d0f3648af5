// dmx_api.cpp -- C ABI of libdemux_hip.so (include/demux_hip.h, include/demux_hip_debug.h): problem install (CSR / CSC derivation
// through csrc/repack_device.hip), switches and read-outs, beta tables, results, device self-tests.  The step drivers are in
// dmx_steps.cpp, the exchange in dmx_exchange.cpp, memory / timers / context life cycle in dmx_runtime.cpp.
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

extern "C" {

// validations shared by the two ways of installing a problem + SNP groups (variants of each SNP in
// increasing variant index = np.bincount order), built on the host: O(V)
static int begin_problem(dmx_ctx *c, int64_t B, int64_t V, int32_t G, const int32_t *v2snp, std::vector<int> &snp_ptr,
                         std::vector<int> &snp_vars, long long &S)
{
    if (B < 0 || V < 0 || G <= 0) return fail(DMX_ERR_INVALID, "bad problem sizes B=%lld V=%lld G=%d", (long long)B, (long long)V, G);
    if (G > 65535) return fail(DMX_ERR_UNSUPPORTED, "G=%d genotypes exceed the 16-bit option encoding", G);
    if (B >= (int64_t(1) << 31) || V >= (int64_t(1) << 31)) return fail(DMX_ERR_UNSUPPORTED, "B and V must fit int32");
    if ((long long)V * G * 4 >= (1LL << 32))
        return fail(DMX_ERR_UNSUPPORTED, "genotype table of %lld x %d floats exceeds the 4 GiB reachable by 32-bit row offsets",
                    (long long)V, G);
    if (V > 0 && !v2snp) return fail(DMX_ERR_INVALID, "null v2snp");
    HIP_TRY(hipStreamSynchronize(c->stream));
    release_problem(c);
    S = 0;
    for (int64_t v = 0; v < V; v++) {
        if (v2snp[v] < 0) return fail(DMX_ERR_INVALID, "v2snp[%lld] negative", (long long)v);
        S = std::max<long long>(S, (long long)v2snp[v] + 1);
    }
    snp_ptr.assign((size_t)S + 1, 0);
    snp_vars.assign((size_t)V, 0);
    for (int64_t v = 0; v < V; v++) snp_ptr[(size_t)v2snp[v] + 1]++;
    for (long long s = 0; s < S; s++) snp_ptr[s + 1] += snp_ptr[s];
    std::vector<int> cur(snp_ptr.begin(), snp_ptr.end() - 1);
    for (int64_t v = 0; v < V; v++) snp_vars[(size_t)cur[v2snp[v]]++] = (int)v;
    c->B = B;
    c->V = V;
    c->G = G;
    c->S = S;
    c->h_v2snp.assign(v2snp, v2snp + V);
    return 0;
}

// genotype tables and per-barcode outputs; called once the call layouts are on the device
static int finish_problem(dmx_ctx *c, const int32_t *v2snp, const std::vector<int> &snp_ptr, const std::vector<int> &snp_vars)
{
    const long long B = c->B, V = c->V, S = c->S;
    const int G = c->G;
    const size_t vg = (size_t)V * G;
    DMX_TRY(dev_alloc(c, &c->d_v2snp, (size_t)V));
    DMX_TRY(dev_alloc(c, &c->d_snp_ptr, (size_t)S + 1));
    DMX_TRY(dev_alloc(c, &c->d_snp_vars, (size_t)V));
    DMX_TRY(dev_alloc(c, &c->d_prior, vg));
    DMX_TRY(dev_alloc(c, &c->d_add, vg));
    DMX_TRY(dev_alloc(c, &c->d_add64, vg));
    DMX_TRY(dev_alloc(c, &c->d_partial, (size_t)c->n_items * G));
    c->cap_redo = ((size_t)c->n_items / 2 + 1) * (size_t)G;  // a variant queues at most G sums and only with >= 2 items
    DMX_TRY(dev_alloc(c, &c->d_redo, c->cap_redo));
    DMX_TRY(dev_alloc(c, &c->d_n_redo, (size_t)2));  // long variants, the others
    DMX_TRY(dev_alloc(c, &c->d_nz, (size_t)B * ((G + 63) / 64)));
    DMX_TRY(dev_alloc(c, &c->d_first, (size_t)B));
    DMX_TRY(dev_alloc(c, &c->d_dense_calls, (size_t)1 + dmx::DENSE_SLOTS));
    DMX_TRY(dev_alloc(c, &c->d_guard_count, (size_t)dmx::GUARD_STATE_WORDS));
    DMX_TRY(dev_alloc(c, &c->d_guard_list, (size_t)B));
    c->guard_sub_cap = (unsigned)((B + dmx::GUARD_QUEUES - 1) / dmx::GUARD_QUEUES);
    DMX_TRY(dev_alloc(c, &c->d_guard_sub, (size_t)dmx::GUARD_QUEUES * c->guard_sub_cap));
    HIP_TRY(hipMemsetAsync(c->d_guard_count, 0, dmx::GUARD_STATE_WORDS * sizeof(unsigned), c->stream));
    HIP_TRY(hipMemsetAsync(c->d_dense_calls, 0, sizeof(unsigned long long) * (1 + dmx::DENSE_SLOTS), c->stream));
    DMX_TRY(dev_alloc(c, &c->d_best, (size_t)B));
    DMX_TRY(dev_alloc(c, &c->d_bestp, (size_t)B));
    hipStream_t st = c->stream;
    if (V) {
        HIP_TRY(hipMemcpyAsync(c->d_v2snp, v2snp, sizeof(int) * V, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(c->d_snp_vars, snp_vars.data(), sizeof(int) * V, hipMemcpyHostToDevice, st));
    }
    HIP_TRY(hipMemcpyAsync(c->d_snp_ptr, snp_ptr.data(), sizeof(int) * (S + 1), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(c->d_add, 0, sizeof(float) * (vg ? vg : 1), st));
    if (B) HIP_TRY(hipMemsetAsync(c->d_nz, 0, sizeof(unsigned long long) * (size_t)B * ((G + 63) / 64), st));
    if (B) HIP_TRY(hipMemsetAsync(c->d_first, 0, sizeof(uint2) * (size_t)B, st));
    HIP_TRY(hipStreamSynchronize(st));  // host staging vectors die in the caller
    DMX_TRY(build_row_segments(c));
    DMX_TRY(layout_exchange(c));        // genotype_prob table (padded when a communicator is attached)
    c->msteps_done = 0;
    c->incr_heavy = false;
    c->have_problem = true;
    return 0;
}

int dmx_set_problem(dmx_ctx *c, int64_t B, int64_t V, int32_t G, int64_t N, const int32_t *variant_id,
                    const int32_t *cb, const float *p_wrong, const int32_t *v2snp)
{
    DMX_TRY(bind(c));
    if (N < 0) return fail(DMX_ERR_INVALID, "negative number of calls");
    if (N > 0 && (!variant_id || !cb || !p_wrong)) return fail(DMX_ERR_INVALID, "null call arrays");
    std::vector<int> snp_ptr, snp_vars;
    long long S = 0;
    DMX_TRY(begin_problem(c, B, V, G, v2snp, snp_ptr, snp_vars, S));
    c->N = N;
    // call records, work items and work lists are derived on the GPU (repack_device.hip)
    DMX_TRY(dmx::repack_on_device(c, variant_id, cb, p_wrong));
    return finish_problem(c, v2snp, snp_ptr, snp_vars);
}

int dmx_pack_and_set_problem(dmx_ctx *c, int64_t B, int64_t V, int32_t G, const int32_t *var_chrom, const int32_t *var_pos,
                             const uint8_t *var_base, const int32_t *v2snp, int64_t n_calls, const int32_t *call_chrom,
                             const int32_t *call_pos, const uint8_t *call_base, const int32_t *call_cb, const float *call_p,
                             int64_t *n_matched, int64_t *n_unique, int64_t *mol_per_variant)
{
    DMX_TRY(bind(c));
    if (n_calls < 0 || !n_matched || !n_unique) return fail(DMX_ERR_INVALID, "bad sizes or null counters");
    if (n_calls > 0 && (!call_chrom || !call_pos || !call_base || !call_cb || !call_p)) return fail(DMX_ERR_INVALID, "null call arrays");
    if (V > 0 && (!var_chrom || !var_pos || !var_base)) return fail(DMX_ERR_INVALID, "null variant arrays");
    std::vector<int> snp_ptr, snp_vars;
    long long S = 0;
    DMX_TRY(begin_problem(c, B, V, G, v2snp, snp_ptr, snp_vars, S));
    long long matched = 0, unique = 0;
    DMX_TRY(dmx::pack_on_device(c, V, var_chrom, var_pos, var_base, n_calls, call_chrom, call_pos, call_base, call_cb, call_p,
                                &matched, &unique, (long long *)mol_per_variant));
    *n_matched = matched;
    *n_unique = unique;
    return finish_problem(c, v2snp, snp_ptr, snp_vars);
}

int dmx_pack_containers_and_set_problem(dmx_ctx *c, int64_t B, int64_t V, int32_t G, const int32_t *var_chrom,
                                        const int32_t *var_pos, const uint8_t *var_base, const int32_t *v2snp,
                                        const dmx_call_container *containers, int32_t n_containers, int64_t *n_matched,
                                        int64_t *n_unique, int64_t *mol_per_variant)
{
    DMX_TRY(bind(c));
    if (n_containers < 0 || !n_matched || !n_unique) return fail(DMX_ERR_INVALID, "bad sizes or null counters");
    if (n_containers > 0 && !containers) return fail(DMX_ERR_INVALID, "null container list");
    for (int k = 0; k < n_containers; k++) {
        const dmx_call_container &p = containers[k];
        if (p.n_snp_calls < 0 || p.n_molecules < 0) return fail(DMX_ERR_INVALID, "container %d: negative size", k);
        if (p.n_snp_calls > 0 && (!p.snp_calls || !p.molecules || p.n_molecules == 0))
            return fail(DMX_ERR_INVALID, "container %d: calls without a molecule table", k);
    }
    if (V > 0 && (!var_chrom || !var_pos || !var_base)) return fail(DMX_ERR_INVALID, "null variant arrays");
    std::vector<int> snp_ptr, snp_vars;
    long long S = 0;
    DMX_TRY(begin_problem(c, B, V, G, v2snp, snp_ptr, snp_vars, S));
    long long matched = 0, unique = 0;
    DMX_TRY(dmx::pack_containers_on_device(c, V, var_chrom, var_pos, var_base, containers, n_containers, &matched, &unique,
                                           (long long *)mol_per_variant));
    *n_matched = matched;
    *n_unique = unique;
    return finish_problem(c, v2snp, snp_ptr, snp_vars);
}

int dmx_stage_containers(dmx_ctx *c, const dmx_call_container *containers, int32_t n_containers)
{
    DMX_TRY(bind(c));
    if (n_containers < 0 || (n_containers > 0 && !containers)) return fail(DMX_ERR_INVALID, "bad container list");
    for (int k = 0; k < n_containers; k++) {
        const dmx_call_container &p = containers[k];
        if (p.n_snp_calls < 0 || p.n_molecules < 0) return fail(DMX_ERR_INVALID, "container %d: negative size", k);
        if (p.n_snp_calls > 0 && (!p.snp_calls || !p.molecules || p.n_molecules == 0))
            return fail(DMX_ERR_INVALID, "container %d: calls without a molecule table", k);
    }
    return dmx::stage_containers_on_device(c, containers, n_containers);
}

int dmx_pack_staged_and_set_problem(dmx_ctx *c, int64_t B, int64_t V, int32_t G, const int32_t *var_chrom, const int32_t *var_pos,
                                    const uint8_t *var_base, const int32_t *v2snp, const int32_t *chrom_of_container,
                                    int32_t n_containers, int64_t *n_matched, int64_t *n_unique, int64_t *mol_per_variant)
{
    DMX_TRY(bind(c));
    if (!n_matched || !n_unique || n_containers < 0) return fail(DMX_ERR_INVALID, "bad sizes or null counters");
    if (V > 0 && (!var_chrom || !var_pos || !var_base)) return fail(DMX_ERR_INVALID, "null variant arrays");
    if (c->n_staged < 0) return fail(DMX_ERR_INVALID, "call order: dmx_stage_containers before dmx_pack_staged_and_set_problem");
    std::vector<int> snp_ptr, snp_vars;
    long long S = 0;
    DMX_TRY(begin_problem(c, B, V, G, v2snp, snp_ptr, snp_vars, S));
    long long matched = 0, unique = 0;
    DMX_TRY(dmx::pack_staged_on_device(c, V, var_chrom, var_pos, var_base, chrom_of_container, n_containers, &matched, &unique,
                                       (long long *)mol_per_variant));
    *n_matched = matched;
    *n_unique = unique;
    return finish_problem(c, v2snp, snp_ptr, snp_vars);
}

int dmx_get_packed_calls(dmx_ctx *c, int32_t *variant_id, int32_t *cb, float *p_wrong, int64_t *count)
{
    DMX_TRY(bind(c));
    DMX_TRY(need(c, c->have_problem && (c->d_u_variant || c->n_u == 0) && c->n_u == c->N,
                 "dmx_pack_and_set_problem before dmx_get_packed_calls"));
    const size_t n = (size_t)c->n_u;
    if (variant_id && n) HIP_TRY(hipMemcpyAsync(variant_id, c->d_u_variant, sizeof(int) * n, hipMemcpyDeviceToHost, c->stream));
    if (cb && n) HIP_TRY(hipMemcpyAsync(cb, c->d_u_cb, sizeof(int) * n, hipMemcpyDeviceToHost, c->stream));
    if (p_wrong && n) HIP_TRY(hipMemcpyAsync(p_wrong, c->d_u_p, sizeof(float) * n, hipMemcpyDeviceToHost, c->stream));
    if (count && n) HIP_TRY(hipMemcpyAsync(count, c->d_u_count, sizeof(long long) * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int dmx_set_exact_additions(dmx_ctx *c, int exact)
{
    if (!c) return fail(DMX_ERR_INVALID, "null context");
    c->exact_additions = exact != 0;
    return 0;
}

int dmx_set_estep_mode(dmx_ctx *c, int mode)
{
    if (!c) return fail(DMX_ERR_INVALID, "null context");
    if (mode != DMX_ESTEP_EXACT && mode != DMX_ESTEP_FAST && mode != DMX_ESTEP_GUARDED) return fail(DMX_ERR_INVALID, "unknown E-step mode %d", mode);
    c->estep_mode = mode;
    return 0;
}

static int read_guard_state(dmx_ctx *c, unsigned (&st)[dmx::GS_WORDS], long long *count, long long *count_fine = nullptr, long long *count_coarse = nullptr)
{
    std::vector<unsigned> all((size_t)dmx::GUARD_STATE_WORDS, 0u);
    if (c->d_guard_count) {
        HIP_TRY(hipMemcpyAsync(all.data(), c->d_guard_count, all.size() * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    for (int i = 0; i < dmx::GS_WORDS; i++) st[i] = all[(size_t)i];
    // the last E-step's flags: the guard of the pass that ran filled the queue, the other guard (a direct E-step: both) counted on its
    // hashed slots (k_guard_begin of the next E-step adds them up and clears them)
    long long hashed_fine = 0, hashed_coarse = 0;
    for (int i = 0; i < dmx::GUARD_SLOTS; i++) {
        hashed_fine += all[(size_t)dmx::GS_SLOTS_FINE + i];
        hashed_coarse += all[(size_t)dmx::GS_SLOTS_COARSE + i];
    }
    const unsigned level = st[dmx::GS_LEVEL];
    const long long fine = level == 1u ? (long long)st[dmx::GS_COUNT] : hashed_fine;
    const long long coarse = level == 0u ? (long long)st[dmx::GS_COUNT] : (st[dmx::GS_CAPABLE] ? hashed_coarse : -1);
    *count = level == 0u ? coarse : fine;  // of the pass that ran (direct: what the fine pass would have queued)
    if (count_fine) *count_fine = fine;
    if (count_coarse) *count_coarse = coarse;
    return 0;
}

int dmx_get_guard_stats(dmx_ctx *c, int64_t *redone_last, int64_t *redone_total, int64_t *rows_total)
{
    DMX_TRY(bind(c));
    unsigned st[dmx::GS_WORDS];
    long long count = 0;
    DMX_TRY(read_guard_state(c, st, &count));
    // barcodes the last guarded E-step computed with the exact kernel: its queue, or all of them when it ran direct
    const long long last = st[dmx::GS_DIRECT] ? (long long)st[dmx::GS_ROWS] : count;
    const long long total = (long long)(((unsigned long long)st[dmx::GS_TOTAL + 1] << 32) | st[dmx::GS_TOTAL]) + (st[dmx::GS_PENDING] ? last : 0);
    if (redone_last) *redone_last = c->guard_ran ? (int64_t)last : 0;
    if (redone_total) *redone_total = (int64_t)total;
    if (rows_total) *rows_total = (int64_t)c->guard_rows_total;
    return 0;
}

int dmx_set_mstep_incremental(dmx_ctx *c, int incremental)
{
    if (!c) return fail(DMX_ERR_INVALID, "null context");
    if (incremental < 0 || incremental > 2) return fail(DMX_ERR_INVALID, "incremental M-step: 0 off, 1 on, 2 on with the first sums built by the delta pass");
    if (incremental != c->mstep_incremental) c->incr_valid = false;
    c->mstep_incremental = incremental;
    return 0;
}

int dmx_get_mstep_incremental(dmx_ctx *c, int64_t *full_passes, int64_t *delta_passes, int64_t *barcodes_last_delta)
{
    DMX_TRY(bind(c));
    unsigned st[3 * dmx::IS_WORDS] = {};
    if (c->d_incr_state) {
        HIP_TRY(hipMemcpyAsync(st, c->d_incr_state, sizeof(st), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    if (full_passes) *full_passes = (int64_t)st[2 * dmx::IS_WORDS];
    if (delta_passes) *delta_passes = (int64_t)st[2 * dmx::IS_WORDS + 1];
    if (barcodes_last_delta) *barcodes_last_delta = st[2 * dmx::IS_WORDS + 2] == 0xFFFFFFFFu ? -1 : (int64_t)st[2 * dmx::IS_WORDS + 2];
    return 0;
}

int dmx_set_coarse_pass(dmx_ctx *c, int coarse)
{
    if (!c) return fail(DMX_ERR_INVALID, "null context");
    if (coarse < 0 || coarse > 2) return fail(DMX_ERR_INVALID, "coarse pass: 0 never, 1 where the logits are not read, 2 wherever the shape allows");
    c->coarse_pass = coarse;
    return 0;
}

int dmx_set_lean_memory(dmx_ctx *c, int lean)
{
    DMX_TRY(bind(c));
    c->lean_memory = lean != 0;
    if (c->lean_memory && c->coarse_ready) {  // the resident problem's coarse records exist: what run_estep would have released behind their build
        dev_free(c, &c->d_tile_stream, (size_t)c->n_pairs);
        dev_free(c, &c->d_call_rows, ((size_t)c->n_pairs + dmx::CALL_PAD_PAIRS) * 2);
    }
    return 0;
}

int dmx_set_guard_adaptive(dmx_ctx *c, int adaptive)
{
    if (!c) return fail(DMX_ERR_INVALID, "null context");
    c->guard_adaptive = adaptive != 0;
    return 0;
}

int dmx_get_guard_direct(dmx_ctx *c, int32_t *last_ran_direct, int64_t *direct_steps, int64_t *would_queue_last, double *fast_pass_ms,
                         double *exact_pass_ms)
{
    DMX_TRY(bind(c));
    unsigned st[dmx::GS_WORDS];
    long long count = 0;
    DMX_TRY(read_guard_state(c, st, &count));
    if (last_ran_direct) *last_ran_direct = c->guard_ran && st[dmx::GS_DIRECT] ? 1 : 0;
    if (direct_steps) *direct_steps = (int64_t)st[dmx::GS_DIRECT_STEPS];
    if (would_queue_last) *would_queue_last = c->guard_ran ? (int64_t)count : 0;
    int khz = 0;
    HIP_TRY(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device));
    const double ms_per_tick = khz > 0 ? 1.0 / (double)khz : 0.0;
    if (fast_pass_ms) *fast_pass_ms = st[dmx::GS_F_TICKS] * ms_per_tick;
    if (exact_pass_ms) *exact_pass_ms = (st[dmx::GS_E_MEASURED] ? 1.0 : -1.0) * st[dmx::GS_E_TICKS] * ms_per_tick;
    return 0;
}

int dmx_get_guard_levels(dmx_ctx *c, int32_t *level_last, int64_t *coarse_steps, int64_t *flagged_fine_last, int64_t *flagged_coarse_last,
                         double *coarse_pass_ms, double *fine_pass_ms, double *exact_pass_ms)
{
    DMX_TRY(bind(c));
    unsigned st[dmx::GS_WORDS];
    long long count = 0, fine = 0, coarse = 0;
    DMX_TRY(read_guard_state(c, st, &count, &fine, &coarse));
    if (level_last) *level_last = c->guard_ran ? (int32_t)st[dmx::GS_LEVEL] : -1;
    if (coarse_steps) *coarse_steps = (int64_t)st[dmx::GS_COARSE_STEPS];
    if (flagged_fine_last) *flagged_fine_last = c->guard_ran ? (int64_t)fine : 0;
    if (flagged_coarse_last) *flagged_coarse_last = c->guard_ran ? (int64_t)coarse : -1;
    int khz = 0;
    HIP_TRY(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device));
    const double ms_per_tick = khz > 0 ? 1.0 / (double)khz : 0.0;
    if (coarse_pass_ms) *coarse_pass_ms = st[dmx::GS_C_TICKS] * ms_per_tick;
    if (fine_pass_ms) *fine_pass_ms = st[dmx::GS_F_TICKS] * ms_per_tick;
    if (exact_pass_ms) *exact_pass_ms = (st[dmx::GS_E_MEASURED] ? 1.0 : -1.0) * st[dmx::GS_E_TICKS] * ms_per_tick;
    return 0;
}

int dmx_set_logits_needed(dmx_ctx *c, int needed)
{
    if (!c) return fail(DMX_ERR_INVALID, "null context");
    c->logits_needed = needed != 0;
    return 0;
}

int dmx_get_guard_probes(dmx_ctx *c, int64_t *probes, int64_t *streak)
{
    DMX_TRY(bind(c));
    unsigned st[dmx::GS_WORDS];
    long long count = 0, fine = 0, coarse = 0;
    DMX_TRY(read_guard_state(c, st, &count, &fine, &coarse));
    if (probes) *probes = (int64_t)st[dmx::GS_PROBES];
    if (streak) *streak = (int64_t)st[dmx::GS_STREAK];
    return 0;
}

int dmx_debug_set_pass_ms(dmx_ctx *c, double coarse_pass_ms, double fine_pass_ms, double exact_pass_ms)
{
    DMX_TRY(bind(c));
    if (!c->d_guard_count) return fail(DMX_ERR_INVALID, "call order: a guarded E-step before dmx_debug_set_pass_ms");
    int khz = 0;
    HIP_TRY(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device));
    const double ms[3] = {coarse_pass_ms, fine_pass_ms, exact_pass_ms};
    const int word[3] = {dmx::GS_C_TICKS, dmx::GS_F_TICKS, dmx::GS_E_TICKS};
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int i = 0; i < 3; i++) {
        if (ms[i] < 0.0) continue;
        const unsigned ticks = (unsigned)std::min(ms[i] * (double)khz, 1.0e9);
        HIP_TRY(hipMemcpy(c->d_guard_count + word[i], &ticks, sizeof(unsigned), hipMemcpyHostToDevice));
        if (i == 2) {
            const unsigned measured = ticks != 0u;
            HIP_TRY(hipMemcpy(c->d_guard_count + dmx::GS_E_MEASURED, &measured, sizeof(unsigned), hipMemcpyHostToDevice));
        }
    }
    return 0;
}

int dmx_set_estep_dictionary(dmx_ctx *c, int mode)
{
    if (!c) return fail(DMX_ERR_INVALID, "null context");
    if (mode < 0 || mode > 2) return fail(DMX_ERR_INVALID, "dictionary mode must be 0, 1 or 2");
    c->dict_mode = mode;
    return 0;
}

int dmx_set_estep_packing(dmx_ctx *c, int on)
{
    if (!c) return fail(DMX_ERR_INVALID, "null context");
    if (on < 0 || on > 3) return fail(DMX_ERR_INVALID, "packing mode must be 0 .. 3");
    c->estep_packing = on;
    return 0;
}

int dmx_get_estep_form(dmx_ctx *c, int32_t *form, int32_t *distinct_values)
{
    if (!c) return fail(DMX_ERR_INVALID, "null context");
    if (form) *form = c->estep_form;
    if (distinct_values) *distinct_values = c->dict_distinct;
    return 0;
}

int dmx_set_estep_schedule(dmx_ctx *c, int tiled)
{
    if (!c) return fail(DMX_ERR_INVALID, "null context");
    if (tiled < 0 || tiled > 2) return fail(DMX_ERR_INVALID, "schedule must be 0, 1 or 2");
    c->tiled_estep = tiled;
    return 0;
}

int dmx_get_redo_count(dmx_ctx *c, int64_t *count)
{
    DMX_TRY(bind(c));
    if (!count) return fail(DMX_ERR_INVALID, "null argument");
    unsigned n[2] = {0, 0};
    if (c->d_n_redo) {
        HIP_TRY(hipMemcpyAsync(n, c->d_n_redo, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    *count = (int64_t)n[0] + (int64_t)n[1];
    return 0;
}

int dmx_set_mstep_tiles(dmx_ctx *c, int enable)
{
    if (!c) return fail(DMX_ERR_INVALID, "null context");
    if (enable < 0 || enable > 2) return fail(DMX_ERR_INVALID, "dmx_set_mstep_tiles: 0 never, 1 when it pays, 2 always");
    c->mstep_tiles = enable;
    return 0;
}

int dmx_set_msteps_expected(dmx_ctx *c, int64_t n)
{
    if (!c) return fail(DMX_ERR_INVALID, "null context");
    c->msteps_expected = n > 0 ? (long long)n : 0;
    return 0;
}

int dmx_get_mstep_tiles_info(dmx_ctx *c, int32_t *built, double *build_ms)
{
    if (!c) return fail(DMX_ERR_INVALID, "null context");
    if (built) *built = c->n_mt > 0 ? 1 : 0;
    if (build_ms) *build_ms = c->n_mt > 0 ? c->mt_build_ms : 0.0;
    return 0;
}

int dmx_get_mstep_form(dmx_ctx *c, int32_t *form)
{
    if (!c || !form) return fail(DMX_ERR_INVALID, "null argument");
    *form = c->mstep_form;
    return 0;
}

int dmx_set_mstep_wide_addresses(dmx_ctx *c, int wide)
{
    if (!c) return fail(DMX_ERR_INVALID, "null context");
    c->mstep_wide = wide != 0;
    return 0;
}

int dmx_set_betas(dmx_ctx *c, const float *prior)
{
    DMX_TRY(bind(c));
    DMX_TRY(need(c, c->have_problem, "dmx_set_problem before dmx_set_betas"));
    if (!prior && c->V > 0) return fail(DMX_ERR_INVALID, "null betas");
    HIP_TRY(hipMemcpyAsync(c->d_prior, prior, sizeof(float) * c->V * c->G, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_betas = true;
    c->have_raw = false;  // (a prior table given as such: no raw betas behind it)
    return 0;
}

int dmx_set_prior_betas(dmx_ctx *c, const float *raw_betas, double default_prior, int add_data_prior,
                        const int64_t *mol_per_variant, float *prior_out)
{
    DMX_TRY(bind(c));
    DMX_TRY(need(c, c->have_problem, "a resident problem before dmx_set_prior_betas"));
    const long long V = c->V;
    const int G = c->G;
    if (V > 0 && !raw_betas) return fail(DMX_ERR_INVALID, "null betas");
    const size_t vg = (size_t)V * G;
    // the raw betas stay on the device: dmx_get_learnt_betas forms raw + addition there (demux.py:65)
    c->have_raw = false;
    if (!c->d_raw) DMX_TRY(dev_alloc(c, &c->d_raw, vg));
    float *d_raw = c->d_raw, *d_bsum = nullptr;
    HIP_TRY(hipMalloc((void **)&d_bsum, (size_t)(V ? V : 1) * sizeof(float)));
    int rc = 0;
    do {
        if (vg && hipMemcpyAsync(d_raw, raw_betas, vg * sizeof(float), hipMemcpyHostToDevice, c->stream) != hipSuccess) {
            rc = fail(DMX_ERR_HIP, "upload of the raw betas failed");
            break;
        }
        const unsigned long long *n_mol = nullptr;
        if (add_data_prior) {
            if (mol_per_variant) {  // counts supplied by the caller (problem installed with dmx_set_problem)
                dev_free(c, &c->d_mol, (size_t)V);
                if ((rc = dev_alloc(c, &c->d_mol, (size_t)V)) != 0) break;
                if (V && hipMemcpyAsync(c->d_mol, mol_per_variant, sizeof(long long) * V, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
                    rc = fail(DMX_ERR_HIP, "upload of the molecule counts failed");
                    break;
                }
            } else if (!c->d_mol) {
                rc = fail(DMX_ERR_INVALID, "add_data_prior needs molecule counts: pass them or install the problem with dmx_pack_and_set_problem");
                break;
            }
            n_mol = c->d_mol;
        }
        const hipError_t e = dmx::launch_prior_betas(c->stream, d_raw, d_bsum, n_mol, c->d_v2snp, c->d_snp_ptr, c->d_snp_vars, V, G,
                                    default_prior, c->d_prior);
        if (e != hipSuccess) {
            rc = fail(DMX_ERR_HIP, "prior betas kernel: %s", hipGetErrorString(e));
            break;
        }
        if (prior_out && vg && hipMemcpyAsync(prior_out, c->d_prior, vg * sizeof(float), hipMemcpyDeviceToHost, c->stream) != hipSuccess) {
            rc = fail(DMX_ERR_HIP, "download of the prior betas failed");
            break;
        }
        if (hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(DMX_ERR_HIP, "synchronisation failed");
    } while (false);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(d_bsum);
    if (rc == 0) c->have_betas = c->have_raw = true;
    return rc;
}

int dmx_get_prior_betas(dmx_ctx *c, float *out)
{
    DMX_TRY(bind(c));
    DMX_TRY(need(c, c->have_problem && c->have_betas, "prior betas (dmx_set_betas / dmx_set_prior_betas) before dmx_get_prior_betas"));
    DMX_TRY(copy_out(c, out, c->d_prior, (size_t)c->V * c->G));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int dmx_get_learnt_betas(dmx_ctx *c, float *out)
{
    DMX_TRY(bind(c));
    DMX_TRY(need(c, c->have_problem && c->have_betas && c->have_raw, "dmx_set_prior_betas (the raw betas) before dmx_get_learnt_betas"));
    if (!out && c->V > 0) return fail(DMX_ERR_INVALID, "null output");
    DMX_TRY(ensure_full_addition(c));  // collective when sliced
    const size_t vg = (size_t)c->V * c->G;
    if (vg == 0) return 0;
    float *d_sum = nullptr;
    DMX_TRY(dev_alloc(c, &d_sum, vg));
    hipError_t e = dmx::launch_add_f32(c->stream, c->d_raw, c->d_add, d_sum, (long long)vg);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_sum, vg * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    const hipError_t e2 = hipStreamSynchronize(c->stream);
    dev_free(c, &d_sum, vg);
    if (e != hipSuccess) return fail(DMX_ERR_HIP, "learnt betas: %s", hipGetErrorString(e));
    if (e2 != hipSuccess) return fail(DMX_ERR_HIP, "learnt betas: %s", hipGetErrorString(e2));
    return 0;
}

int dmx_get_logits(dmx_ctx *c, float *out)
{
    DMX_TRY(bind(c));
    DMX_TRY(need(c, c->have_post, "dmx_estep before dmx_get_logits"));
    DMX_TRY(need(c, c->logits_readable, "the last E-step ran with dmx_set_logits_needed(ctx, 0): its logits were not kept (dmx_estep computes them)"));
    DMX_TRY(copy_out(c, out, c->d_logits, (size_t)c->B * c->K));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int dmx_get_probs(dmx_ctx *c, float *out)
{
    DMX_TRY(bind(c));
    DMX_TRY(need(c, c->have_post, "dmx_estep before dmx_get_probs"));
    DMX_TRY(copy_out(c, out, c->d_post, (size_t)c->B * c->K));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int dmx_get_addition(dmx_ctx *c, float *out)
{
    DMX_TRY(bind(c));
    DMX_TRY(need(c, c->have_problem, "dmx_set_problem before dmx_get_addition"));
    DMX_TRY(ensure_full_addition(c));  // collective when sliced
    DMX_TRY(copy_out(c, out, c->d_add, (size_t)c->V * c->G));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int dmx_get_block(dmx_ctx *c, int what, int64_t b0, int64_t b1, int64_t k0, int64_t k1, float *out)
{
    DMX_TRY(bind(c));
    DMX_TRY(need(c, c->have_post, "dmx_estep before dmx_get_block"));
    if (what != DMX_LOGITS && what != DMX_PROBS) return fail(DMX_ERR_INVALID, "what must be DMX_LOGITS or DMX_PROBS");
    if (what == DMX_LOGITS)
        DMX_TRY(need(c, c->logits_readable, "the last E-step ran with dmx_set_logits_needed(ctx, 0): its logits were not kept (dmx_estep computes them)"));
    if (b0 < 0 || b1 < b0 || b1 > c->B || k0 < 0 || k1 < k0 || k1 > c->K)
        return fail(DMX_ERR_INVALID, "block [%lld,%lld) x [%lld,%lld) outside [0,%lld) x [0,%d)", (long long)b0, (long long)b1,
                    (long long)k0, (long long)k1, c->B, c->K);
    if (b1 == b0 || k1 == k0) return 0;
    if (!out) return fail(DMX_ERR_INVALID, "null output");
    const float *src = (what == DMX_LOGITS ? c->d_logits : c->d_post) + (size_t)b0 * c->K + k0;
    HIP_TRY(hipMemcpy2DAsync(out, (size_t)(k1 - k0) * 4, src, (size_t)c->K * 4, (size_t)(k1 - k0) * 4, (size_t)(b1 - b0),
                             hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int dmx_get_assignments(dmx_ctx *c, int32_t *best, float *best_p)
{
    DMX_TRY(bind(c));
    DMX_TRY(need(c, c->have_post, "dmx_estep before dmx_get_assignments"));
    HIP_TRY(dmx::launch_assign(c->stream, c->d_post, c->B, c->K, c->d_best, c->d_bestp));
    if (best) HIP_TRY(hipMemcpyAsync(best, c->d_best, sizeof(int) * c->B, hipMemcpyDeviceToHost, c->stream));
    if (best_p) HIP_TRY(hipMemcpyAsync(best_p, c->d_bestp, sizeof(float) * c->B, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

// ---- self tests -------------------------------------------------------------------
static int scratch(dmx_ctx *c, size_t bytes)
{
    if (bytes <= c->cap_scratch) return 0;
    if (c->d_scratch) (void)hipFree(c->d_scratch);
    c->d_scratch = nullptr;
    c->cap_scratch = 0;
    hipError_t e = hipMalloc(&c->d_scratch, bytes);
    if (e != hipSuccess) return fail(DMX_ERR_HIP, "hipMalloc(scratch %zu): %s", bytes, hipGetErrorString(e));
    c->cap_scratch = bytes;
    return 0;
}

static int unary_test(dmx_ctx *c, const float *in, float *out, int64_t n, int which, int64_t rows, int64_t cols)
{
    DMX_TRY(bind(c));
    if (n < 0 || (n > 0 && (!in || !out))) return fail(DMX_ERR_INVALID, "bad test buffers");
    if (n == 0) return 0;
    DMX_TRY(scratch(c, (size_t)n * 8));
    float *d_in = (float *)c->d_scratch, *d_out = d_in + n;
    HIP_TRY(hipMemcpyAsync(d_in, in, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    if (which == 0) HIP_TRY(dmx::launch_test_log(c->stream, d_in, d_out, n));
    if (which == 1) HIP_TRY(dmx::launch_test_exp(c->stream, d_in, d_out, n));
    if (which == 3) HIP_TRY(dmx::launch_test_log_hot(c->stream, d_in, d_out, n));
    if (which == 4) HIP_TRY(dmx::launch_test_log2_hw(c->stream, d_in, d_out, n));
    if (which == 2) HIP_TRY(dmx::launch_test_softmax(c->stream, d_in, d_out, rows, (int)cols));
    HIP_TRY(hipMemcpyAsync(out, d_out, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int dmx_test_logf(dmx_ctx *c, const float *in, float *out, int64_t n) { return unary_test(c, in, out, n, 0, 0, 0); }

int dmx_test_logf_hot(dmx_ctx *c, const float *in, float *out, int64_t n) { return unary_test(c, in, out, n, 3, 0, 0); }

int dmx_test_expf(dmx_ctx *c, const float *in, float *out, int64_t n) { return unary_test(c, in, out, n, 1, 0, 0); }

int dmx_test_log2_hw(dmx_ctx *c, const float *in, float *out, int64_t n) { return unary_test(c, in, out, n, 4, 0, 0); }

int dmx_test_softmax(dmx_ctx *c, const float *in, float *out, int64_t rows, int64_t cols)
{
    if (rows < 0 || cols <= 0 || cols > (1 << 24)) return fail(DMX_ERR_INVALID, "bad softmax test shape");
    return unary_test(c, in, out, rows * cols, 2, rows, cols);
}

}  // extern "C"
