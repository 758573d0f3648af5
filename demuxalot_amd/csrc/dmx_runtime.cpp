// dmx_runtime.cpp -- errors, the sum plan of np.sum, the contexts' device-block cache, phase timers, the life cycle of a context
// (C ABI: include/demux_hip.h; helpers shared with the other translation units: dmx_host.h).
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

// ------------------------------------------------------------------------------------
// error handling
// ------------------------------------------------------------------------------------
namespace dmx {
static thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}
}  // namespace dmx

namespace dmx {
namespace host {
const char *last_error() { return g_last_error.c_str(); }
}  // namespace host
}  // namespace dmx

namespace dmx {

// np.sum over a row of K values as a plan a whole workgroup executes (np_math.h: plan_sum_block): numpy's pairwise tree
// spelled out - the blocks of <= 128 elements, then the inner nodes level by level from the deepest, then the roots of
// the 8192-element chunks, which are added left to right.
int ensure_sum_plan(dmx_ctx *c, long long K)
{
    if (K == c->sum_plan_k) return 0;
    std::vector<int> leaves;                  // (start, length) of the blocks of <= 128 elements
    std::vector<int> roots;
    // values: [0, n_leaves) the leaf sums, then one value per inner node
    std::vector<std::array<int, 3>> inner;  // left, right, height (leaf = 0)
    std::function<std::pair<int, int>(int, int)> build = [&](int start, int n) -> std::pair<int, int> {  // (value, height)
        if (n <= 128) {
            leaves.push_back(start);
            leaves.push_back(n);
            return {(int)leaves.size() / 2 - 1, 0};
        }
        int half = n / 2;
        half -= half % 8;
        const auto l = build(start, half), r = build(start + half, n - half);
        inner.push_back({l.first, r.first, std::max(l.second, r.second) + 1});
        return {-(int)inner.size(), std::max(l.second, r.second) + 1};  // inner nodes: negative handles, resolved below
    };
    for (long long s0 = 0; s0 < K; s0 += 8192) roots.push_back(build((int)s0, (int)std::min<long long>(8192, K - s0)).first);
    const int n_leaves = (int)leaves.size() / 2;
    int max_h = 0;
    for (auto &nd : inner) max_h = std::max(max_h, nd[2]);
    // order the inner nodes by height (children before parents), remember where each went
    std::vector<int> place(inner.size());
    std::vector<int> level_off(1, 0);
    std::vector<int> ordered;
    for (int h = 1; h <= max_h; h++) {
        for (size_t i = 0; i < inner.size(); i++)
            if (inner[i][2] == h) {
                place[i] = n_leaves + (int)ordered.size() / 2;
                ordered.push_back((int)i);
                ordered.push_back(0);
            }
        level_off.push_back((int)ordered.size() / 2);
    }
    auto value_of = [&](int handle) { return handle >= 0 ? handle : place[(size_t)(-handle - 1)]; };
    std::vector<int> plan;
    plan.push_back(n_leaves);
    plan.push_back(max_h);
    plan.push_back((int)roots.size());
    for (int h = 0; h <= max_h; h++) plan.push_back(level_off[(size_t)h]);
    plan.insert(plan.end(), leaves.begin(), leaves.end());
    for (size_t q = 0; q < ordered.size(); q += 2) {
        const auto &nd = inner[(size_t)ordered[q]];
        plan.push_back(value_of(nd[0]));
        plan.push_back(value_of(nd[1]));
    }
    for (int r : roots) plan.push_back(value_of(r));
    dev_free(c, &c->d_sum_plan, c->cap_sum_plan);
    c->cap_sum_plan = plan.size();
    DMX_TRY(dev_alloc(c, &c->d_sum_plan, plan.size()));
    HIP_TRY(hipMemcpyAsync(c->d_sum_plan, plan.data(), sizeof(int) * plan.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->sum_plan_k = K;
    c->sum_plan_values = n_leaves + (int)inner.size();
    return 0;
}

}  // namespace dmx

size_t retired_trim(int device);

size_t ctx_cache_limit()
{
    static const size_t limit = [] {
        const char *e = std::getenv("DEMUXALOT_AMD_CACHE_GB");
        const double gb = e ? atof(e) : 24.0;
        return gb <= 0 ? (size_t)0 : (size_t)(gb * 1073741824.0);
    }();
    return limit;
}

// Blocks of destroyed contexts (their streams are idle by then), per device, for the contexts created later: a fresh
// context's first problem cost 0.45 s more than a re-used one's inside a process that had closed a large context before
// (bench.py's e2e part after the timed regions: 0.57 s against 0.11 s).
// Lock order: g_retired.lock (retired lists + registry of live contexts) before any context's cache_lock.
namespace {
struct RetiredBlocks {
    std::mutex lock;
    std::multimap<size_t, void *> idle[16];
    size_t bytes[16] = {};
    std::vector<dmx_ctx *> live[16];
};
RetiredBlocks g_retired;

size_t trim_locked(dmx_ctx *c, size_t keep_bytes)  // c->cache_lock held
{
    size_t freed = 0;
    while (!c->idle_blocks.empty() && c->idle_bytes > keep_bytes) {
        auto it = std::prev(c->idle_blocks.end());
        (void)hipFree(it->second);  // waits for the device: whatever was queued on the block is done
        c->idle_bytes -= it->first;
        freed += it->first;
        c->block_capacity.erase(it->second);
        c->idle_blocks.erase(it);
    }
    return freed;
}
}  // namespace

void ctx_register(dmx_ctx *c)
{
    if (c->device < 0 || c->device >= 16) return;
    std::lock_guard<std::mutex> guard(g_retired.lock);
    g_retired.live[c->device].push_back(c);
}

void ctx_unregister(dmx_ctx *c)
{
    if (c->device < 0 || c->device >= 16) return;
    std::lock_guard<std::mutex> guard(g_retired.lock);
    auto &v = g_retired.live[c->device];
    v.erase(std::remove(v.begin(), v.end(), c), v.end());
}

// Everything parked on a device goes back to the driver: the idle blocks of EVERY live context (pooled private contexts
// are unreachable from the API, and each may hold gigabytes) and the retired list.
size_t trim_device_caches(int device)
{
    if (device < 0 || device >= 16) return 0;
    std::lock_guard<std::mutex> guard(g_retired.lock);
    size_t freed = 0;
    for (dmx_ctx *other : g_retired.live[device]) {
        std::lock_guard<std::mutex> own(other->cache_lock);
        freed += trim_locked(other, 0);
    }
    for (auto &kv : g_retired.idle[device]) (void)hipFree(kv.second);
    g_retired.idle[device].clear();
    freed += g_retired.bytes[device];
    g_retired.bytes[device] = 0;
    return freed;
}

int ctx_malloc(dmx_ctx *c, void **p, size_t bytes)
{
    *p = nullptr;
    if (bytes == 0) bytes = 1;
    {
        // an idle block of this size, or up to an eighth (+ 64 KB) larger
        std::lock_guard<std::mutex> own(c->cache_lock);
        auto it = c->idle_blocks.lower_bound(bytes);
        if (it != c->idle_blocks.end() && it->first <= bytes + bytes / 8 + 65536) {
            *p = it->second;
            c->idle_bytes -= it->first;
            c->idle_blocks.erase(it);
            return 0;
        }
    }
    if (c->device >= 0 && c->device < 16) {
        std::lock_guard<std::mutex> guard(g_retired.lock);
        auto &pool = g_retired.idle[c->device];
        auto jt = pool.lower_bound(bytes);
        if (jt != pool.end() && jt->first <= bytes + bytes / 8 + 65536) {
            *p = jt->second;
            {
                std::lock_guard<std::mutex> own(c->cache_lock);
                c->block_capacity[*p] = jt->first;
            }
            g_retired.bytes[c->device] -= jt->first;
            pool.erase(jt);
            return 0;
        }
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {  // out of memory with blocks parked on this device - here, in sibling contexts, retired: give them back, try again
        (void)hipGetLastError();
        ctx_trim(c, 0);
        (void)trim_device_caches(c->device);
        e = hipMalloc(p, bytes);
    }
    if (e != hipSuccess) {
        *p = nullptr;
        return fail(DMX_ERR_HIP, "hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
    }
    std::lock_guard<std::mutex> own(c->cache_lock);
    c->block_capacity[*p] = bytes;
    return 0;
}

void ctx_free(dmx_ctx *c, void *p)
{
    if (!p) return;
    bool over = false;
    {
        std::lock_guard<std::mutex> own(c->cache_lock);
        auto it = c->block_capacity.find(p);
        if (it == c->block_capacity.end() || ctx_cache_limit() == 0) {
            if (it != c->block_capacity.end()) c->block_capacity.erase(it);
            (void)hipFree(p);
            return;
        }
        c->idle_blocks.emplace(it->second, p);
        c->idle_bytes += it->second;
        over = c->idle_bytes > ctx_cache_limit();
    }
    if (over) ctx_trim(c, ctx_cache_limit() / 2);
}

size_t retired_trim(int device)  // returns the bytes given back
{
    if (device < 0 || device >= 16) return 0;
    std::lock_guard<std::mutex> guard(g_retired.lock);
    for (auto &kv : g_retired.idle[device]) (void)hipFree(kv.second);
    g_retired.idle[device].clear();
    const size_t freed = g_retired.bytes[device];
    g_retired.bytes[device] = 0;
    return freed;
}

// dmx_destroy: the context's idle blocks (its stream has been waited for) go to the device's retired list, up to the
// cache limit; what does not fit is freed
void ctx_retire(dmx_ctx *c)
{
    if (c->device < 0 || c->device >= 16 || ctx_cache_limit() == 0) {
        ctx_trim(c, 0);
        return;
    }
    {
        std::lock_guard<std::mutex> guard(g_retired.lock);
        std::lock_guard<std::mutex> own(c->cache_lock);
        for (auto it = c->idle_blocks.begin(); it != c->idle_blocks.end();) {
            if (g_retired.bytes[c->device] + it->first > ctx_cache_limit()) {
                ++it;
                continue;
            }
            g_retired.idle[c->device].emplace(it->first, it->second);
            g_retired.bytes[c->device] += it->first;
            c->idle_bytes -= it->first;
            c->block_capacity.erase(it->second);
            it = c->idle_blocks.erase(it);
        }
    }
    ctx_trim(c, 0);
}

// hipFree (which waits for the device) of idle blocks, largest first, until at most keep_bytes stay parked
void ctx_trim(dmx_ctx *c, size_t keep_bytes)
{
    std::lock_guard<std::mutex> own(c->cache_lock);
    (void)trim_locked(c, keep_bytes);
}

namespace dmx {
namespace host {

int bind(dmx_ctx *c)
{
    if (!c) return fail(DMX_ERR_INVALID, "null ctx");
    HIP_TRY(hipSetDevice(c->device));
    c->boundary = nullptr;  // whatever this call enqueues first sits behind the last phase's stamp
    return 0;
}

TimerStamp *stamp_now(dmx_ctx *c)
{
    TimerStamp *s;
    if (!c->idle_stamps.empty()) {
        s = c->idle_stamps.back();
        c->idle_stamps.pop_back();
    } else {
        s = new TimerStamp;
        (void)hipEventCreateWithFlags(&s->ev, hipEventReleaseToDevice);
    }
    s->refs = 0;
    (void)hipEventRecord(s->ev, c->stream);
    return s;
}

void stamp_release(dmx_ctx *c, TimerStamp *s)
{
    if (--s->refs > 0) return;
    if (c->boundary == s) c->boundary = nullptr;
    c->idle_stamps.push_back(s);
}

void timer_begin(dmx_ctx *c, int slot, TimerSpan *ev)
{
    (void)slot;
    ev->first = ev->second = nullptr;
    if (!c->phase_timers) return;
    ev->first = c->boundary ? c->boundary : stamp_now(c);  // the phase before this one ended here: one barrier packet, not two
    ev->first->refs++;
    c->boundary = nullptr;
}

void timer_flush(dmx_ctx *c, int slot)
{
    TimerSlot &t = c->timers[slot];
    for (auto &ev : t.pending) {
        (void)hipEventSynchronize(ev.second->ev);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev.first->ev, ev.second->ev) == hipSuccess) t.ms += ms;
        stamp_release(c, ev.first);
        stamp_release(c, ev.second);
    }
    t.pending.clear();
}

void timer_end(dmx_ctx *c, int slot, TimerSpan &ev)
{
    TimerSlot &t = c->timers[slot];
    if (ev.first == nullptr) {  // (the timers were off when the phase began)
        t.launches++;
        return;
    }
    ev.second = stamp_now(c);
    ev.second->refs++;
    c->boundary = ev.second;
    t.pending.push_back(ev);
    t.launches++;
    t.timed++;
    ev.first = ev.second = nullptr;  // (handed to the slot: SpanGuard has nothing to give back)
    if (t.pending.size() >= 4096) timer_flush(c, slot);
}

// incremental M-step: the sums, the posteriors they were formed from, the work lists (run_mstep allocates them at first use)
void release_incremental(dmx_ctx *c)
{
    const size_t vg = (size_t)c->V * c->G;
    dev_free(c, &c->d_acc64, vg);
    const size_t rows = (size_t)(c->incr_rows > 0 ? c->incr_rows : c->B);
    dev_free(c, &c->d_prev_post, rows * c->G);
    dev_free(c, &c->d_prev_first, rows);
    dev_free(c, &c->d_incr_list, rows);
    dev_free(c, &c->d_incr_map, rows);
    dev_free(c, &c->d_slice_rec, (size_t)c->n_slice_rec);
    dev_free(c, &c->d_slice_ptr, rows + 1);
    c->n_slice_rec = 0;
    c->slice_index_tried = false;
    c->incr_rows = 0;
    dev_free(c, &c->d_incr_touched, (size_t)c->V);
    dev_free(c, &c->d_incr_state, (size_t)(3 * dmx::IS_WORDS));
    c->incr_valid = false;
}

// the coarse pass's records and constants (run_estep builds them at the problem's first admissible E-step)
void release_coarse_stream(dmx_ctx *c)
{
    dev_free(c, &c->d_coarse_stream, c->cap_coarse_stream);
    c->cap_coarse_stream = 0;
    dev_free(c, &c->d_coarse_bin_ptr, (size_t)c->n_bins + 1);
    dev_free(c, &c->d_log2_keep, (size_t)c->B);
    c->coarse_ready = false;
}

void release_problem(dmx_ctx *c)
{
    // the blocks released here are handed out again at once (ctx_malloc) to work ordered on c->stream: whatever the
    // other streams of the context still have queued on them must be done first (hipFree used to wait for the device)
    dev_free(c, &c->d_pair_ptr, (size_t)c->B + 1);
    dev_free(c, &c->d_call_pairs, (size_t)c->n_pairs + dmx::CALL_PAD_PAIRS);
    dev_free(c, &c->d_call_rows, ((size_t)c->n_pairs + dmx::CALL_PAD_PAIRS) * 2);
    dev_free(c, &c->d_tile_stream, (size_t)c->n_pairs);
    release_coarse_stream(c);
    release_incremental(c);
    c->n_pairs = 0;
    dmx::release_mstep_tiles(c);  // (its record stream is sized by n_csc)
    dev_free(c, &c->d_csc, (size_t)c->n_csc);
    c->n_csc = 0;
    dev_free(c, &c->d_item_start, (size_t)c->n_items);
    dev_free(c, &c->d_item_len, (size_t)c->n_items);
    dev_free(c, &c->d_item_ptr, (size_t)c->V + 1);
    dev_free(c, &c->d_item_variant, (size_t)c->n_items);
    dev_free(c, &c->d_bc_order, (size_t)c->B);
    dev_free(c, &c->d_bin_rows, (size_t)c->n_bins * c->bin_rows_cap);
    dev_free(c, &c->d_bin_order, (size_t)c->n_bins);
    dev_free(c, &c->d_bin_ptr, (size_t)c->n_bins + 1);
    c->n_bins = 0;
    c->n_tiles = c->bin_rows_cap = 0;
    dev_free(c, &c->d_item_order, (size_t)c->n_items);
    dev_free(c, &c->d_v2snp, (size_t)c->V);
    dev_free(c, &c->d_snp_ptr, (size_t)c->S + 1);
    dev_free(c, &c->d_snp_vars, (size_t)c->V);
    const size_t vg = (size_t)c->V * c->G;
    dev_free(c, &c->d_prior, vg);
    dev_free(c, &c->d_raw, vg);
    c->have_raw = false;
    dev_free(c, &c->d_add, vg);
    dev_free(c, &c->d_prob_list, c->prob_list_words * (size_t)std::max(1, c->nranks));
    dev_free(c, &c->d_prob_prev, (size_t)c->slice_rows * c->G);
    c->prob_list_words = 0;
    c->prob_list_cap = 0;
    c->prob_prev_valid = false;
    if (c->h_prob_counts) (void)hipHostFree(c->h_prob_counts);
    c->h_prob_counts = nullptr;
    dev_free(c, &c->d_prob, (size_t)c->prob_rows * c->G);
    dev_free(c, &c->d_prob16, c->cap_prob16);
    c->cap_prob16 = 0;
    c->prob16_valid = false;
    dev_free(c, &c->d_add64, vg);
    dev_free(c, &c->d_prow, (size_t)c->V);
    dev_free(c, &c->d_row_variant, (size_t)c->prob_rows);
    if (c->d_exch) {
        (void)hipFree(c->d_exch);
        c->bytes -= (int64_t)c->exch_bytes;
        c->d_exch = nullptr;
        c->exch_bytes = 0;
    }
    if (c->d_recv) {
        (void)hipFree(c->d_recv);
        c->bytes -= (int64_t)c->recv_bytes;
        c->d_recv = nullptr;
        c->recv_bytes = 0;
    }
    dev_free(c, &c->d_first_g, (size_t)c->rows_total);
    dev_free(c, &c->d_nz_g, (size_t)c->rows_total * ((c->G + 63) / 64));
    dev_free(c, &c->d_post_g, (size_t)c->rows_total * c->G);
    dev_free(c, &c->d_post_compact, c->post_compact_words * (size_t)std::max(1, c->nranks));
    dev_free(c, &c->d_post_seen, (size_t)c->rows_total);
    dev_free(c, &c->d_post_sent, (size_t)std::max<long long>(1, c->B) * c->G);
    dev_free(c, &c->d_post_sent_multi, (size_t)std::max<long long>(1, c->B));
    if (c->h_post_counts) (void)hipHostFree(c->h_post_counts);
    c->h_post_counts = nullptr;
    c->post_compact_words = 0;
    c->post_compact_cap = 0;
    c->mshard = c->post_gathered = c->emu_post_filled = false;
    c->rows_pad = c->rows_total = 0;
    c->sliced = c->add_partial = false;
    c->slice_rows = c->prob_rows = 0;
    c->cut.clear();
    c->h_v2snp.clear();
    c->h_col_ptr.clear();
    dev_free(c, &c->d_partial, (size_t)c->n_items * c->G);
    dev_free(c, &c->d_redo, c->cap_redo);
    dev_free(c, &c->d_n_redo, (size_t)2);
    c->cap_redo = 0;
    dev_free(c, &c->d_logits, (size_t)c->cap_bk);
    dev_free(c, &c->d_post, (size_t)c->cap_bk);
    c->cap_bk = 0;
    dev_free(c, &c->d_nz, (size_t)c->B * ((c->G + 63) / 64));
    dev_free(c, &c->d_first, (size_t)c->B);
    dev_free(c, &c->d_dense_calls, (size_t)1 + dmx::DENSE_SLOTS);
    c->dense_stat_valid = false;
    dev_free(c, &c->d_segs, (size_t)c->n_segs);
    dev_free(c, &c->d_split_first, (size_t)c->n_split + 1);
    dev_free(c, &c->d_seg_sums, c->cap_seg_sums);
    c->cap_seg_sums = 0;
    c->n_segs = c->n_split = 0;
    dev_free(c, &c->d_guard_count, (size_t)dmx::GUARD_STATE_WORDS);
    dev_free(c, &c->d_guard_list, (size_t)c->B);
    dev_free(c, &c->d_guard_sub, (size_t)dmx::GUARD_QUEUES * c->guard_sub_cap);
    c->guard_sub_cap = 0;
    c->guard_rows_total = 0;
    c->guard_ran = false;
    dev_free(c, &c->d_dict, c->cap_dict_rows * dmx::DICT_CAP);
    dev_free(c, &c->d_codes, c->cap_dict_rows * (size_t)dmx::dict_code_pitch(c->G));
    dev_free(c, &c->d_dtab, c->cap_dtab);
    dev_free(c, &c->d_dict_stat, (size_t)1);
    c->cap_dict_rows = c->cap_dtab = 0;
    c->dict_candidate = false;
    c->add_is_zero = true;
    c->estep_form = DMX_FORM_NONE;
    c->dict_distinct = 0;
    dev_free(c, &c->d_pen, (size_t)c->cap_k);
    dev_free(c, &c->d_pairs, (size_t)c->cap_k);
    dev_free(c, &c->d_pair_blocks, (size_t)c->cap_pair_blocks);
    c->cap_pair_blocks = c->n_pair_blocks = 0;
    dev_free(c, &c->d_sum_plan, c->cap_sum_plan);
    c->cap_sum_plan = 0;
    c->sum_plan_k = -1;
    c->cap_k = 0;
    if (c->d_prior_logits) {
        (void)hipFree(c->d_prior_logits);
        c->bytes -= (int64_t)c->cap_prior;
        c->d_prior_logits = nullptr;
        c->cap_prior = 0;
    }
    dev_free(c, &c->d_best, (size_t)c->B);
    dev_free(c, &c->d_bestp, (size_t)c->B);
    dev_free(c, &c->d_u_variant, (size_t)c->n_u);
    dev_free(c, &c->d_u_cb, (size_t)c->n_u);
    dev_free(c, &c->d_u_p, (size_t)c->n_u);
    dev_free(c, &c->d_u_count, (size_t)c->n_u);
    c->n_u = 0;
    dev_free(c, &c->d_mol, (size_t)c->V);
    dev_free(c, &c->d_mc_variant, (size_t)c->n_mc);
    dev_free(c, &c->d_mc_e, (size_t)c->n_mc);
    dev_free(c, &c->d_mc_start, (size_t)c->B + 1);
    c->n_mc = 0;
    c->mc_max_count = 0;
    dev_free(c, &c->d_logits64, c->cap_bk64);
    dev_free(c, &c->d_post64, c->cap_bk64);
    c->cap_bk64 = 0;
    c->have_post64 = false;
    c->have_problem = c->have_betas = c->have_probs = c->have_post = false;
    c->B = c->V = c->N = c->S = 0;
    c->G = c->K = 0;
    c->n_items = 0;
}

int copy_out(dmx_ctx *c, float *dst, const float *src, size_t count)
{
    if (!dst) return 0;
    HIP_TRY(hipMemcpyAsync(dst, src, count * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    return 0;
}

int need(dmx_ctx *c, bool cond, const char *what)
{
    (void)c;
    if (!cond) return fail(DMX_ERR_INVALID, "call order: %s", what);
    return 0;
}

}  // namespace host
}  // namespace dmx

extern "C" {

const char *dmx_last_error(void) { return dmx::g_last_error.c_str(); }

const char *dmx_version(void) { return "demux_hip 0.1 (gfx950)"; }

int dmx_device_count(int *count)
{
    if (!count) return fail(DMX_ERR_INVALID, "null count");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(DMX_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return 0;
}

int dmx_create(int device, dmx_ctx **out)
{
    if (!out) return fail(DMX_ERR_INVALID, "null out pointer");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(DMX_ERR_NO_DEVICE, "no HIP device visible: the demuxalot_amd hot path needs an MI355X (there is no CPU fallback)");
    if (device < 0 || device >= n) return fail(DMX_ERR_INVALID, "device %d out of range (%d visible)", device, n);
    HIP_TRY(hipSetDevice(device));
    dmx_ctx *c = new dmx_ctx();
    c->device = device;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail(DMX_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    ctx_register(c);
    *out = c;
    return 0;
}

int dmx_destroy(dmx_ctx *c)
{
    if (!c) return 0;
    ctx_unregister(c);
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    comm_destroy(c);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    release_problem(c);
    if (c->d_scratch) (void)hipFree(c->d_scratch);
    dmx::release_staged_calls(c);
    (void)hipDeviceSynchronize();  // the exchange stream too
    ctx_retire(c);
    c->boundary = nullptr;
    for (int slot = 0; slot < DMX_T_COUNT; slot++) timer_flush(c, slot);
    for (TimerStamp *s : c->idle_stamps) {
        (void)hipEventDestroy(s->ev);
        delete s;
    }
    (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

int dmx_synchronize(dmx_ctx *c)
{
    DMX_TRY(bind(c));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int dmx_release_problem(dmx_ctx *c)
{
    DMX_TRY(bind(c));
    HIP_TRY(hipStreamSynchronize(c->stream));
    release_problem(c);
    dmx::release_staged_calls(c);
    return 0;
}

int dmx_trim_device_caches(int device, int64_t *released_bytes)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return fail(DMX_ERR_INVALID, "device %d out of range", device);
    HIP_TRY(hipSetDevice(device));
    const size_t freed = trim_device_caches(device);
    if (released_bytes) *released_bytes = (int64_t)freed;
    return 0;
}

int dmx_trim_cache(dmx_ctx *c, int64_t *released_bytes)
{
    DMX_TRY(bind(c));
    size_t before = 0;
    {
        std::lock_guard<std::mutex> own(c->cache_lock);
        before = c->idle_bytes;
    }
    ctx_trim(c, 0);
    before += retired_trim(c->device);
    if (released_bytes) *released_bytes = (int64_t)before;
    return 0;
}

int dmx_set_phase_timers(dmx_ctx *c, int on)
{
    if (!c) return fail(DMX_ERR_INVALID, "null ctx");
    c->phase_timers = on != 0;
    c->boundary = nullptr;
    return 0;
}

int dmx_get_timings(dmx_ctx *c, double *ms, int64_t *launches)
{
    DMX_TRY(bind(c));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int s = 0; s < DMX_T_COUNT; s++) {
        timer_flush(c, s);
        // -1: launches ran in this slot, none of them between events (the phase timers were off: dmx_set_phase_timers)
        if (ms) ms[s] = (c->timers[s].launches > 0 && c->timers[s].timed == 0) ? -1.0 : c->timers[s].ms;
        if (launches) launches[s] = c->timers[s].launches;
    }
    return 0;
}

int dmx_reset_timings(dmx_ctx *c)
{
    DMX_TRY(bind(c));
    HIP_TRY(hipStreamSynchronize(c->stream));
    // the totals; the last E-step's own numbers stay (they decide how the next one runs: kernels.hip k_guard_begin)
    if (c->d_guard_count) HIP_TRY(hipMemsetAsync(c->d_guard_count + dmx::GS_PENDING, 0, 4 * sizeof(unsigned), c->stream));  // GS_PENDING, GS_DIRECT_STEPS, GS_TOTAL
    if (c->d_guard_count) HIP_TRY(hipMemsetAsync(c->d_guard_count + dmx::GS_COARSE_STEPS, 0, 2 * sizeof(unsigned), c->stream));  // + GS_PROBES
    if (c->d_incr_state) HIP_TRY(hipMemsetAsync(c->d_incr_state + 2 * dmx::IS_WORDS, 0, sizeof(unsigned) * 3, c->stream));  // (word 3: full passes since the install, kept)
    c->guard_rows_total = 0;
    for (int s = 0; s < DMX_T_COUNT; s++) {
        timer_flush(c, s);
        c->timers[s].ms = 0.0;
        c->timers[s].launches = 0;
        c->timers[s].timed = 0;
    }
    return 0;
}

int dmx_device_bytes(dmx_ctx *c, int64_t *bytes)
{
    if (!c || !bytes) return fail(DMX_ERR_INVALID, "null argument");
    *bytes = c->bytes;
    return 0;
}

}  // extern "C"
