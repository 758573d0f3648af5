// dmx_api.cpp -- C ABI of libdemux_hip.so (include/demux_hip.h): context, device memory,
// problem upload (CSR/CSC derivation), step drivers, RCCL all-reduce, timing.
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


// ------------------------------------------------------------------------------------
// RCCL, loaded on demand so that single-GPU use has no dependency on it
// ------------------------------------------------------------------------------------
namespace {
struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string path;
};
RcclApi g_rccl;

// Files of the process image whose name contains `needle` (/proc/self/maps), each once.
std::vector<std::string> mapped_files(const char *needle)
{
    std::vector<std::string> out;
    FILE *f = std::fopen("/proc/self/maps", "r");
    if (!f) return out;
    char line[4096];
    while (std::fgets(line, sizeof line, f)) {
        const char *path = std::strchr(line, '/');
        if (!path || !std::strstr(path, needle)) continue;
        std::string p(path);
        while (!p.empty() && (p.back() == '\n' || p.back() == ' ')) p.pop_back();
        if (std::find(out.begin(), out.end(), p) == out.end()) out.push_back(p);
    }
    std::fclose(f);
    return out;
}

// ONE HIP runtime per process.  libdemux_hip.so is linked against the ROCm installation's libamdhip64; the RCCL it
// hands its streams and buffers to must sit on the same runtime.  So RCCL is taken from the directory of the HIP
// runtime this library resolved (dladdr of hipGetDeviceCount), never from whatever copy a launcher happened to map (a
// process that imported torch carries torch's own librccl + libamdhip64 + libhsa-runtime64: streams of one runtime
// handed to collectives of the other is undefined, and round 2 did exactly that under `bench.py --gpus N`).
// A process with two HIP runtimes mapped is refused - loudly - unless DEMUXALOT_AMD_ALLOW_FOREIGN_RCCL=1.
// DEMUXALOT_AMD_RCCL=<path> overrides the library file.
int load_rccl()
{
    if (g_rccl.handle) return 0;
    const bool lenient = std::getenv("DEMUXALOT_AMD_ALLOW_FOREIGN_RCCL") && std::atoi(std::getenv("DEMUXALOT_AMD_ALLOW_FOREIGN_RCCL")) != 0;
    const std::vector<std::string> hips = mapped_files("libamdhip64");
    if (hips.size() > 1 && !lenient) {
        std::string all;
        for (const auto &h : hips) all += (all.empty() ? "" : ", ") + h;
        return fail(DMX_ERR_RCCL, "two HIP runtimes are mapped into this process (%s): a multi-rank worker must not import torch "
                                  "(use demuxalot_amd.plane.SocketControlPlane for the control plane); set "
                                  "DEMUXALOT_AMD_ALLOW_FOREIGN_RCCL=1 to run anyway", all.c_str());
    }
    std::string path;
    if (const char *forced = std::getenv("DEMUXALOT_AMD_RCCL")) {
        path = forced;
    } else {
        Dl_info info;
        if (dladdr((const void *)&hipGetDeviceCount, &info) && info.dli_fname && std::strchr(info.dli_fname, '/')) {
            path = info.dli_fname;
            path = path.substr(0, path.rfind('/')) + "/librccl.so.1";
        } else {
            path = "/opt/rocm/lib/librccl.so.1";
        }
    }
    void *h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(DMX_ERR_RCCL, "cannot load %s: %s", path.c_str(), dlerror());
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
    g_rccl.ReduceScatter = (decltype(g_rccl.ReduceScatter))dlsym(h, "ncclReduceScatter");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(h, "ncclAllGather");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
    g_rccl.GroupStart = (decltype(g_rccl.GroupStart))dlsym(h, "ncclGroupStart");
    g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))dlsym(h, "ncclGroupEnd");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.ReduceScatter || !g_rccl.AllGather ||
        !g_rccl.CommDestroy) {
        dlclose(h);
        return fail(DMX_ERR_RCCL, "%s lacks a required symbol", path.c_str());
    }
    // loading RCCL must not have brought a second runtime along either
    const std::vector<std::string> after = mapped_files("libamdhip64");
    if (after.size() > 1 && !lenient) {
        dlclose(h);
        return fail(DMX_ERR_RCCL, "%s depends on another HIP runtime (%s) than this library (%s)", path.c_str(), after.back().c_str(), after.front().c_str());
    }
    g_rccl.handle = h;
    g_rccl.path = path;
    return 0;
}
}  // namespace

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

namespace {

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

// A phase that leaves through an error return between timer_begin and timer_end (HIP_TRY / DMX_TRY) still holds a reference to
// its opening stamp: without this the stamp never returned to idle_stamps and its event was never destroyed.
struct SpanGuard {
    dmx_ctx *c;
    TimerSpan *ev;
    ~SpanGuard()
    {
        if (ev->first != nullptr) {
            stamp_release(c, ev->first);
            ev->first = nullptr;
        }
    }
};

// incremental M-step: the sums, the posteriors they were formed from, the work lists (run_mstep allocates them at first use)
static void release_incremental(dmx_ctx *c)
{
    const size_t vg = (size_t)c->V * c->G;
    dev_free(c, &c->d_acc64, vg);
    dev_free(c, &c->d_prev_post, (size_t)c->B * c->G);
    dev_free(c, &c->d_prev_first, (size_t)c->B);
    dev_free(c, &c->d_incr_list, (size_t)c->B);
    dev_free(c, &c->d_incr_touched, (size_t)c->V);
    dev_free(c, &c->d_incr_state, (size_t)(3 * dmx::IS_WORDS));
    c->incr_valid = false;
}

// the coarse pass's records and constants (run_estep builds them at the problem's first admissible E-step)
static void release_coarse_stream(dmx_ctx *c)
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
    dev_free(c, &c->d_prob, (size_t)c->prob_rows * c->G);
    dev_free(c, &c->d_prob16, c->cap_prob16);
    c->cap_prob16 = 0;
    c->prob16_valid = false;
    dev_free(c, &c->d_add64, vg);
    dev_free(c, &c->d_prow, (size_t)c->V);
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

const char *rccl_error(ncclResult_t r) { return g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"; }

// ------------------------------------------------------------------------------------
// Multi-GPU exchange.  Every EM iteration needs, on every rank, the genotype_prob table computed from
// prior + the sum over ranks of the per-rank beta additions.  Instead of all-reducing the [V, G] float64 sums and
// running the P-step on every rank (2 (n-1)/n x 8 bytes per entry on the wire, the P-step replicated), the variants
// are cut into one slice per rank at SNP boundaries:
//     reduce-scatter (float64 partial sums, or float32)  ->  rank r owns the summed addition of slice r
//     round to float32, P-step of slice r                ->  rank r owns genotype_prob of slice r
//     all-gather (float32 genotype_prob)                 ->  everybody has the table for the next E-step
// = (n-1)/n x (8 + 4) bytes per entry, the P-step done once.  ncclReduceScatter / ncclAllGather want equal,
// contiguous blocks, so the tables that travel (the exchange buffer of the M-step, genotype_prob) are kept in a
// padded row layout: slice r = rows [r * slice_rows, (r + 1) * slice_rows).  Only the E-step records (row byte
// offsets) and the kernels that write those two tables know about it (prow).  The full addition is assembled
// (all-gather of the float32 slices) only when a caller asks for it.
// Requires every SNP's variants to be contiguous in the variant numbering (they are when genotypes come from a
// VCF: genotypes.py:112-168); otherwise `sliced` is false and the exchange is the all-reduce + replicated P-step.
// ------------------------------------------------------------------------------------
// Variant slices of the exchange (host only): cut[r] = first variant of slice r, every cut at the first variant of
// a SNP; rows = the longest slice.  contiguous = every SNP id forms exactly one run of v2snp.
void exchange_slices(const int *v2snp, long long V, int n, std::vector<long long> &cut, long long &rows, bool &contiguous)
{
    contiguous = true;
    int max_snp = -1;
    for (long long v = 0; v < V; v++) max_snp = std::max(max_snp, v2snp[v]);
    std::vector<char> seen((size_t)max_snp + 1, 0);
    for (long long v = 0; v < V && contiguous; v++) {
        if (v > 0 && v2snp[v] == v2snp[v - 1]) continue;
        if (seen[v2snp[v]]) contiguous = false;
        seen[v2snp[v]] = 1;
    }
    cut.assign((size_t)n + 1, 0);
    cut[n] = V;
    for (int r = 1; r < n; r++) {
        long long v = V * r / n;
        while (v > 0 && v < V && v2snp[v] == v2snp[v - 1]) v--;  // back to the first variant of the SNP
        cut[r] = std::max(v, cut[r - 1]);
    }
    rows = 1;
    for (int r = 0; r < n; r++) rows = std::max(rows, cut[r + 1] - cut[r]);
}

// ---- the three collectives of the exchange: RCCL on the ctx stream, or the caller's over pinned host memory --------
int host_stage(dmx_ctx *c, size_t bytes)
{
    if (bytes <= c->h_stage_bytes) return 0;
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    c->h_stage = nullptr;
    c->h_stage_bytes = 0;
    HIP_TRY(hipHostMalloc(&c->h_stage, bytes, hipHostMallocDefault));
    c->h_stage_bytes = bytes;
    return 0;
}

// runs `op` on the caller's collectives: device [src, src + bytes_in) -> host stage at byte offset off_in, callback,
// host stage [off_out, off_out + bytes_out) -> device dst
int host_collective(dmx_ctx *c, int op, const void *src, size_t off_in, size_t bytes_in, void *dst, size_t off_out, size_t bytes_out,
                    size_t total_bytes, int64_t count, int dtype, const char *what, hipStream_t st)
{
    DMX_TRY(host_stage(c, total_bytes));
    char *h = (char *)c->h_stage;
    if (bytes_in) HIP_TRY(hipMemcpyAsync(h + off_in, src, bytes_in, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const int rc = c->host_coll(c->host_user, op, h, count, dtype);
    if (rc != 0) return fail(DMX_ERR_RCCL, "the caller's collective (%s) failed with %d", what, rc);
    if (bytes_out) HIP_TRY(hipMemcpyAsync(dst, h + off_out, bytes_out, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));  // the stage is reused by the next collective
    return 0;
}

// Emulated wire: a direct exchange on a fully connected node moves one block per peer link in either direction, all links
// at once: latency + block bytes / link rate, whatever the number of ranks (the blocks shrink with it).
int emulated_wire(dmx_ctx *c, size_t block_bytes, int rounds, hipStream_t st)
{
    if (c->nranks <= 1) return 0;
    // inside a group (coll_group_begin) the collectives are one launch: the latency is paid by the first only
    const double latency = c->in_group && c->group_paid ? 0.0 : c->emu_latency_us * 1e3;
    c->group_paid = true;
    const double ns = rounds * (latency + (double)block_bytes / c->emu_link_gbps);
    HIP_TRY(dmx::launch_delay(st, (long long)(ns * c->emu_ticks_per_ns)));
    return 0;
}

// Several collectives as one launch (ncclGroupStart / ncclGroupEnd); the host-staged and emulated backends run them one
// after the other.
void coll_group_begin(dmx_ctx *c)
{
    c->in_group = true;
    c->group_paid = false;
    if (c->comm && g_rccl.GroupStart) (void)g_rccl.GroupStart();
}

int coll_group_end(dmx_ctx *c)
{
    c->in_group = false;
    if (c->comm && g_rccl.GroupEnd) {
        ncclResult_t r = g_rccl.GroupEnd();
        if (r != ncclSuccess) return fail(DMX_ERR_RCCL, "ncclGroupEnd failed: %s", rccl_error(r));
    }
    return 0;
}

// recv[block] = sum over ranks of their send[rank * block ...]
int coll_reduce_scatter(dmx_ctx *c, const void *send, void *recv, size_t block, bool f64, hipStream_t st)
{
    const size_t elem = f64 ? 8 : 4;
    if (c->emulated) {  // the other ranks "send zeros": this rank's own block is the sum
        HIP_TRY(hipMemcpyAsync(recv, (const char *)send + (size_t)c->rank * block * elem, block * elem, hipMemcpyDeviceToDevice, st));
        return emulated_wire(c, block * elem, 1, st);
    }
    if (c->comm) {
        ncclResult_t r = g_rccl.ReduceScatter(send, recv, block, f64 ? ncclDouble : ncclFloat, ncclSum, c->comm, st);
        return r == ncclSuccess ? 0 : fail(DMX_ERR_RCCL, "ncclReduceScatter failed: %s", rccl_error(r));
    }
    const size_t total = block * elem * c->nranks;
    return host_collective(c, DMX_COLL_REDUCE_SCATTER, send, 0, total, recv, block * elem * c->rank, block * elem, total, (int64_t)block,
                           f64 ? DMX_F64 : DMX_F32, "reduce-scatter", st);
}

// float32 table of nranks blocks, this rank's block filled: everybody's blocks on return
int coll_all_gather(dmx_ctx *c, float *table, size_t block, const char *what)
{
    if (c->emulated) return emulated_wire(c, block * 4, 1, c->stream);  // the other slices keep what they hold
    if (c->comm) {
        ncclResult_t r = g_rccl.AllGather(table + c->rank * block, table, block, ncclFloat, c->comm, c->stream);
        return r == ncclSuccess ? 0 : fail(DMX_ERR_RCCL, "ncclAllGather (%s) failed: %s", what, rccl_error(r));
    }
    const size_t total = block * 4 * c->nranks, mine = block * 4 * c->rank;
    return host_collective(c, DMX_COLL_ALL_GATHER, table + c->rank * block, mine, block * 4, table, 0, total, total, (int64_t)block, DMX_F32, what, c->stream);
}

int coll_all_reduce(dmx_ctx *c, void *buf, size_t count, bool f64)
{
    if (c->emulated) return emulated_wire(c, count * (f64 ? 8 : 4) / (size_t)std::max(1, c->nranks), 2, c->stream);  // = reduce-scatter + all-gather
    if (c->comm) {
        ncclResult_t r = g_rccl.AllReduce(buf, buf, count, f64 ? ncclDouble : ncclFloat, ncclSum, c->comm, c->stream);
        return r == ncclSuccess ? 0 : fail(DMX_ERR_RCCL, "ncclAllReduce failed: %s", rccl_error(r));
    }
    const size_t total = count * (f64 ? 8 : 4);
    return host_collective(c, DMX_COLL_ALL_REDUCE, buf, 0, total, buf, 0, total, total, (int64_t)count, f64 ? DMX_F64 : DMX_F32, "all-reduce", c->stream);
}

// small host numbers of every rank, through the data-plane collective: out[r * count + i] = rank r's values[i] (each < 2^48)
int gather_numbers(dmx_ctx *c, const long long *values, int count, std::vector<long long> &out)
{
    const int n = c->nranks;
    std::vector<float> host((size_t)n * count * 3, 0.0f);  // three 16-bit digits per number: exact in float32
    for (int i = 0; i < count; i++)
        for (int d = 0; d < 3; d++) host[((size_t)c->rank * count + i) * 3 + d] = (float)((values[i] >> (16 * d)) & 0xFFFF);
    float *dev = nullptr;
    HIP_TRY(hipMalloc((void **)&dev, host.size() * sizeof(float)));
    int rc = 0;
    if (hipMemcpyAsync(dev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = fail(DMX_ERR_HIP, "upload failed");
    if (rc == 0) rc = coll_all_gather(c, dev, (size_t)count * 3, "sizes");
    if (rc == 0 && hipMemcpyAsync(host.data(), dev, host.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = fail(DMX_ERR_HIP, "download failed");
    if (rc == 0 && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(DMX_ERR_HIP, "synchronisation failed");
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(dev);
    if (rc) return rc;
    out.assign((size_t)n * count, 0);
    for (int r = 0; r < n; r++)
        for (int i = 0; i < count; i++) {
            long long v = 0;
            for (int d = 0; d < 3; d++) v |= (long long)host[((size_t)r * count + i) * 3 + d] << (16 * d);
            out[(size_t)r * count + i] = c->emulated ? values[i] : v;  // emulated wire: every rank is a copy of this one
        }
    return 0;
}

// ------------------------------------------------------------------------------------
// Multi-GPU, what is exchanged.  The E-step shards on barcodes.  Reducing the M-step's partial [V, G] sums over the ranks
// (round 3: reduce-scatter over variant slices) puts a DENSE table on the wire - 51 MB at 200k variants x 64 - although
// what the ranks really have to tell each other is sparse: the posteriors, ~1.05 live genotypes per barcode.  And it makes
// every rank walk all V variants with 1 / n of the calls each: items of ~50 calls at 8 ranks, where the M-step kernels run
// at half their rate, plus an unsharded combine pass.  So the M-step shards on VARIANTS instead:
//     set-up   every rank's variant-major call records travel once (all-gather); rank r keeps the calls of ITS variant
//              slice from the barcodes of ALL ranks (global barcode row = owner * rows_pad + barcode)
//     E-step   writes its barcodes' posterior codes / bitmaps / singlet posteriors into its block of three global tables
//     exchange all-gather of those three tables (8 + 8 W + 4 G bytes per barcode)
//     M-step   rank r sums slice r over all barcodes in the reference's order - float64, one rounding: the additions
//              are BIT-IDENTICAL to a single-GPU run whatever the number of ranks, nothing is added across ranks
//     P-step   of slice r, then the all-gather of genotype_prob as before
// ------------------------------------------------------------------------------------
// force: DEMUXALOT_AMD_EXCHANGE=variant.  Otherwise the exchange with fewer bytes on the wire per iteration is taken: what
// the M-step reads of ALL barcodes (4 G + 8 + 8 W bytes each: grows with the barcodes of the whole job) against the
// [V, G] partial sums of the reduce-scatter (fixed).  One 200k-barcode experiment over n GPUs: the posteriors (54 MB
// against 51 / 102 MB of float32 / float64 sums) - and the M-step then walks whole variants instead of 1 / n of each;
// n x 200k barcodes (weak scaling): the sums.  Every rank sees the same sizes and decides alike.
int shard_mstep_by_variant(dmx_ctx *c, bool force)
{
    const int n = c->nranks, G = c->G, W = (G + 63) / 64;
    hipStream_t st = c->stream;
    const long long mine[2] = {c->B, c->n_csc};
    std::vector<long long> all;
    DMX_TRY(gather_numbers(c, mine, 2, all));
    long long rows_pad = 1, calls_pad = 1;
    for (int r = 0; r < n; r++) {
        rows_pad = std::max(rows_pad, all[(size_t)2 * r]);
        calls_pad = std::max(calls_pad, all[(size_t)2 * r + 1]);
    }
    if (rows_pad * n >= (1LL << 31)) return fail(DMX_ERR_UNSUPPORTED, "%lld barcode rows over all ranks exceed int32", rows_pad * n);
    const double posterior_bytes = (double)rows_pad * n * (4.0 * G + 8.0 + 8.0 * W);
    const double sum_bytes = (double)c->V * G * (c->reduce_dtype == DMX_F64 ? 8.0 : 4.0);
    // at equal bytes the variant-sharded M-step is the faster exchange (0.69 against 0.92 ms per iteration at 8 ranks of the
    // 200k-barcode experiment, where the ratio is 1.07: whole variants instead of 1 / n of each, no combine pass)
    if (!force && posterior_bytes > 1.25 * sum_bytes) return 0;  // the reduce-scatter of the sums moves clearly less
    // the call records of every rank
    uint4 *wire = nullptr;
    const size_t wire_bytes = sizeof(uint4) * (size_t)calls_pad * n;
    hipError_t e = hipMalloc((void **)&wire, wire_bytes);
    if (e != hipSuccess) return fail(DMX_ERR_HIP, "hipMalloc of %zu bytes for the call records of all ranks failed: %s", wire_bytes, hipGetErrorString(e));
    int rc = 0;
    if (c->emulated) {  // emulated wire: the other ranks hold copies of this rank's calls (their barcodes other rows)
        for (int r = 0; r < n && rc == 0; r++) rc = dmx::wire_records_of(c, r * rows_pad, wire + (size_t)r * calls_pad, calls_pad);
        if (rc == 0) rc = emulated_wire(c, sizeof(uint4) * (size_t)calls_pad, 1, st);
    } else {
        rc = dmx::wire_records_of(c, c->rank * rows_pad, wire + (size_t)c->rank * calls_pad, calls_pad);
        if (rc == 0) rc = coll_all_gather(c, (float *)wire, (size_t)calls_pad * 4, "call records");
    }
    if (rc == 0) rc = dmx::install_mstep_records(c, wire, calls_pad * n, c->cut[c->rank], c->cut[c->rank + 1]);
    (void)hipStreamSynchronize(st);
    (void)hipFree(wire);
    if (rc) return rc;
    c->rows_pad = rows_pad;
    c->rows_total = rows_pad * n;
    DMX_TRY(dev_alloc(c, &c->d_first_g, (size_t)c->rows_total));
    DMX_TRY(dev_alloc(c, &c->d_nz_g, (size_t)c->rows_total * W));
    DMX_TRY(dev_alloc(c, &c->d_post_g, (size_t)c->rows_total * G));
    HIP_TRY(hipMemsetAsync(c->d_first_g, 0, sizeof(uint2) * (size_t)c->rows_total, st));
    HIP_TRY(hipMemsetAsync(c->d_nz_g, 0, sizeof(unsigned long long) * (size_t)c->rows_total * W, st));
    HIP_TRY(hipMemsetAsync(c->d_post_g, 0, sizeof(float) * (size_t)c->rows_total * G, st));
    c->mshard = true;
    c->post_gathered = false;
    c->emu_post_filled = false;
    return 0;
}

int layout_exchange(dmx_ctx *c)
{
    const long long V = c->V;
    const int G = c->G, n = c->attached() ? c->nranks : 1;
    hipStream_t st = c->stream;
    HIP_TRY(hipStreamSynchronize(st));
    if (c->d_prow) return fail(DMX_ERR_INVALID, "the resident problem is already laid out for a communicator: install it again");
    bool contiguous = true;
    long long rows = V;
    exchange_slices(c->h_v2snp.data(), V, n, c->cut, rows, contiguous);
    // DEMUXALOT_AMD_EXCHANGE=allreduce: the plain exchange (all-reduce of the float64 sums, P-step on every rank) whatever
    // the SNP layout - the fallback switch for the sliced exchange (reduce-scatter / sliced P-step / all-gather)
    const char *exchange = std::getenv("DEMUXALOT_AMD_EXCHANGE");
    const bool force_allreduce = exchange && std::strcmp(exchange, "allreduce") == 0;
    c->sliced = c->attached() && contiguous && V > 0 && !force_allreduce;
    if (!c->sliced) {
        c->cut.assign((size_t)n + 1, 0);
        c->cut[n] = V;
    } else {
        if ((unsigned long long)rows * n * G * 4ull >= (1ull << 32))
            return fail(DMX_ERR_UNSUPPORTED, "padded genotype table of %lld x %d floats exceeds the 4 GiB reachable by 32-bit row offsets",
                        rows * n, G);
    }
    c->slice_rows = c->sliced ? rows : V;
    const long long new_rows = c->sliced ? rows * n : V;
    if (new_rows != c->prob_rows || !c->d_prob) {
        dev_free(c, &c->d_prob, (size_t)c->prob_rows * G);
        c->prob_rows = new_rows;
        DMX_TRY(dev_alloc(c, &c->d_prob, (size_t)new_rows * G));
    }
    c->have_probs = false;
    c->emu_table_filled = false;
    HIP_TRY(hipMemsetAsync(c->d_prob, 0, sizeof(float) * (size_t)(new_rows ? new_rows * G : 1), st));
    if (c->sliced) {
        std::vector<int> prow((size_t)V);
        for (int r = 0; r < n; r++)
            for (long long v = c->cut[r]; v < c->cut[r + 1]; v++) prow[v] = (int)(r * rows + (v - c->cut[r]));
        DMX_TRY(dev_alloc(c, &c->d_prow, (size_t)V));
        HIP_TRY(hipMemcpyAsync(c->d_prow, prow.data(), sizeof(int) * V, hipMemcpyHostToDevice, st));
        HIP_TRY(dmx::launch_remap_row_offsets(st, c->d_call_pairs, c->n_pairs, (unsigned)G * 4u, c->d_prow, c->d_call_rows));
        if (c->d_tile_stream) HIP_TRY(dmx::launch_remap_row_offsets(st, c->d_tile_stream, c->n_pairs, (unsigned)G * 4u, c->d_prow, nullptr));
        release_coarse_stream(c);  // (its row offsets are the tile-major stream's: rebuilt at the next admissible E-step)
        const size_t elem = c->reduce_dtype == DMX_F64 ? 8 : 4;
        c->exch_bytes = (size_t)new_rows * G * 8;  // float64 sums of the reduce-scatter exchange; also the float32 staging of the addition gather
        c->recv_bytes = (size_t)rows * G * elem;
        HIP_TRY(hipMalloc(&c->d_exch, c->exch_bytes));
        c->bytes += (int64_t)c->exch_bytes;
        HIP_TRY(hipMalloc(&c->d_recv, c->recv_bytes));
        c->bytes += (int64_t)c->recv_bytes;
        HIP_TRY(hipMemsetAsync(c->d_exch, 0, c->exch_bytes, st));  // padding rows stay zero
        HIP_TRY(hipStreamSynchronize(st));                          // `prow` is a local
        // DEMUXALOT_AMD_EXCHANGE=reduce_scatter keeps the M-step on every rank's own barcodes and reduce-scatters the sums;
        // =variant shards the M-step on variants whatever the sizes; default: whichever moves fewer bytes per iteration
        const bool by_sums = exchange && std::strcmp(exchange, "reduce_scatter") == 0;
        const bool by_variant = exchange && std::strcmp(exchange, "variant") == 0;
        // (forced, it also runs with ONE rank: its collectives through a real one-rank RCCL communicator on a one-GPU test box)
        if ((n > 1 && !by_sums) || by_variant) DMX_TRY(shard_mstep_by_variant(c, by_variant));
    }
    c->add_partial = false;
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
static bool coarse_capable(const dmx_ctx *c, int with_doublets, float lo)
{
    return c->coarse_pass && c->estep_mode == DMX_ESTEP_GUARDED && !with_doublets && c->K > 16 && c->K <= 128 && c->tiled_estep && c->n_bins > 0 &&
           lo >= 6.2e-5f && ((unsigned long long)c->prob_rows + 1ull) * (unsigned long long)c->G * 4ull < (1ull << 32);
}

// the table as binary16 + the all-zero row the padding calls gather (EstepArgs::prob16)
static int ensure_prob16(dmx_ctx *c)
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
int run_pstep(dmx_ctx *c, float lo, float hi, bool with_addition, bool with_half = false)
{
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
                                         c->d_prob, with_half ? c->d_prob16 : nullptr));
    c->prob16_valid = with_half;
    timer_end(c, DMX_T_PSTEP, ev);
    if (c->sliced) {  // everybody gets everybody's slice of genotype_prob
        timer_begin(c, DMX_T_ALLREDUCE, &ev);
        const size_t block = (size_t)c->slice_rows * c->G;
        const int rc = coll_all_gather(c, c->d_prob, block, "genotype_prob");
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
int run_estep(dmx_ctx *c, int with_doublets, bool with_prior, int prior_dtype, float power, bool logits_kept = true)
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
    TimerSpan ev{nullptr, nullptr};
    SpanGuard ev_guard{c, &ev};
    timer_begin(c, DMX_T_ESTEP, &ev);
    int form = DMX_FORM_DIRECT;
    // The dictionary form is exact and faster than the fine pass, but not than the COARSE pass (200k x 100k x 64: 1.1 ms with its
    // dictionary build against 0.75): an E-step whose logits nobody reads - the first of a dmx_em call of several iterations - takes
    // the coarse pass like the ones behind it (its records are built here instead of one E-step later).
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
            }
            if (allow_coarse) DMX_TRY(ensure_prob16(c));
            HIP_TRY(dmx::launch_guard_begin(c->stream, c->d_guard_count, c->B, c->K, c->guard_adaptive, capable, allow_coarse));
            a.guard = 1;
            a.order_direct = c->d_bc_order;
            a.guard_main_coarse = 0;
            a.guard_alt_per_call = capable ? dmx::GUARD_PER_CALL_COARSE : 0.0f;
            a.guard_alt_accum = capable ? dmx::GUARD_ACCUM_F32 : 0.0f;
            if (allow_coarse) {
                if (!c->prob16_valid)  // (the P-step of a dmx_em / dmx_run_iterations call has written it already)
                    HIP_TRY(dmx::launch_prob_to_half(c->stream, c->d_prob, c->prob_rows, c->G, c->d_prob16, c->d_guard_count + dmx::GS_SKIP_COARSE));
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
            HIP_TRY(dmx::launch_estep(c->stream, a, with_doublets != 0));
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

// Variant-sharded M-step (shard_mstep_by_variant): everybody's posterior codes, bitmaps and singlet posteriors, gathered
// once per E-step.  Emulated wire: the other ranks' blocks hold a copy of this rank's first ones (what the M-step reads
// of them decides its work), never refreshed.
int gather_posteriors(dmx_ctx *c)
{
    if (!c->mshard || c->post_gathered) return 0;
    const int G = c->G, W = (G + 63) / 64;
    const size_t rows = (size_t)c->rows_pad;
    TimerSpan ev{nullptr, nullptr};
    SpanGuard ev_guard{c, &ev};
    timer_begin(c, DMX_T_ALLREDUCE, &ev);
    int rc = 0;
    if (c->emulated && !c->emu_post_filled) {
        for (int r = 0; r < c->nranks; r++) {
            if (r == c->rank) continue;
            HIP_TRY(hipMemcpyAsync(c->d_first_g + r * rows, c->d_first_g + c->rank * rows, sizeof(uint2) * rows, hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(c->d_nz_g + r * rows * W, c->d_nz_g + c->rank * rows * W, sizeof(unsigned long long) * rows * W, hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(c->d_post_g + r * rows * G, c->d_post_g + c->rank * rows * G, sizeof(float) * rows * G, hipMemcpyDeviceToDevice, c->stream));
        }
        c->emu_post_filled = true;
    }
    coll_group_begin(c);  // one launch for the three tables
    rc = coll_all_gather(c, (float *)c->d_first_g, rows * 2, "posterior codes");
    if (rc == 0) rc = coll_all_gather(c, (float *)c->d_nz_g, rows * W * 2, "posterior bitmaps");
    if (rc == 0) rc = coll_all_gather(c, c->d_post_g, rows * G, "singlet posteriors");
    const int rc_end = coll_group_end(c);
    if (rc == 0) rc = rc_end;
    timer_end(c, DMX_T_ALLREDUCE, ev);
    if (rc) return rc;
    c->post_gathered = true;
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
    // tile-major form (kernels.h: MTileArgs): sums in any order, so not with the exact additions; built on first use
    a.tiles_done = false;
    dmx::MTileArgs tiles{};
    // Building the records (a sort of the calls: 2.6 ms on 200k x 100k x 64, where an M-step + combine then takes 0.34 instead of
    // 0.70 ms) pays from MSTEP_TILES_PAY M-steps on: taken when that many are still to come - in the running dmx_em /
    // dmx_run_iterations call, or as the caller announced (dmx_set_msteps_expected) -, or the problem has seen that many
    // already (somebody iterates call by call), or always (dmx_set_mstep_tiles(ctx, 2)).
    constexpr int MSTEP_TILES_PAY = 8;
    const long long ahead = std::max<long long>(c->msteps_ahead, c->msteps_expected);
    const bool tiles_wanted = c->mstep_tiles == 2 || (c->mstep_tiles == 1 && (c->n_mt > 0 || ahead >= MSTEP_TILES_PAY ||
                                                                             c->msteps_done >= MSTEP_TILES_PAY));
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
                       power > 0.0f && !dist && !mshard && !c->sliced && a.out32 == c->d_add && c->d_call_rows != nullptr && c->d_item_variant != nullptr;
    if (fixed_items) {
        DMX_TRY(dmx::plan_mstep_shifts(c));
        fixed_items = c->d_mt_shift_v != nullptr;
    }
    // Incremental form (kernels.h: MIncrArgs): one context with all calls of its barcodes, the tiles' per-variant exponents at hand.
    const bool incremental = (a.tiles_done || fixed_items) && c->mstep_incremental && !dist && !c->sliced && c->d_mt_shift_v != nullptr &&
                             a.out32 == c->d_add && c->d_call_rows != nullptr;
    dmx::MIncrArgs incr{};
    if (incremental) {
        if (!c->d_acc64) {
            DMX_TRY(dev_alloc(c, &c->d_acc64, (size_t)c->V * c->G));
            DMX_TRY(dev_alloc(c, &c->d_prev_post, (size_t)c->B * c->G));
            DMX_TRY(dev_alloc(c, &c->d_prev_first, (size_t)c->B));
            DMX_TRY(dev_alloc(c, &c->d_incr_list, (size_t)c->B));
            DMX_TRY(dev_alloc(c, &c->d_incr_touched, (size_t)c->V));
            DMX_TRY(dev_alloc(c, &c->d_incr_state, (size_t)(3 * dmx::IS_WORDS)));  // two alternating sets + the counters
            HIP_TRY(hipMemsetAsync(c->d_incr_state, 0, sizeof(unsigned) * 3 * dmx::IS_WORDS, c->stream));
            HIP_TRY(hipMemsetAsync(c->d_incr_touched, 0, (size_t)c->V, c->stream));
            c->incr_valid = false;
        }
        if (!c->incr_valid || c->incr_power != power) {  // (nothing to build on: zeroed state words ask for the full pass)
            HIP_TRY(hipMemsetAsync(c->d_incr_state, 0, sizeof(unsigned) * 2 * dmx::IS_WORDS, c->stream));
            if (c->mstep_incremental == 2) {  // (measurement: the sums built from nothing by the delta pass instead of the full pass)
                const unsigned on[2] = {1u, 1u};
                HIP_TRY(hipMemsetAsync(c->d_acc64, 0, sizeof(unsigned long long) * (size_t)c->V * c->G, c->stream));
                HIP_TRY(hipMemsetAsync(c->d_prev_post, 0, sizeof(float) * (size_t)c->B * c->G, c->stream));
                HIP_TRY(hipMemsetAsync(c->d_prev_first, 0xFF, sizeof(uint2) * (size_t)c->B, c->stream));
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
        incr.pairs = c->d_call_pairs;
        incr.call_rows = c->d_call_rows;
        incr.pair_ptr = c->d_pair_ptr;
        incr.B = c->B;
        incr.V = c->V;
        incr.floor = dmx::mincr_floor(power);
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
    if (incremental && fixed_items) HIP_TRY(dmx::launch_mstep_items_incremental(c->stream, a, incr));
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

}  // namespace

// ------------------------------------------------------------------------------------
// ABI
// ------------------------------------------------------------------------------------
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
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
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

int dmx_exchange_slices(int64_t n_variants, const int32_t *v2snp, int32_t nranks, int64_t *cuts, int64_t *slice_rows,
                        int32_t *contiguous)
{
    if (n_variants < 0 || nranks < 1 || !cuts || (n_variants > 0 && !v2snp)) return fail(DMX_ERR_INVALID, "bad arguments");
    for (int64_t v = 0; v < n_variants; v++)
        if (v2snp[v] < 0) return fail(DMX_ERR_INVALID, "v2snp[%lld] negative", (long long)v);
    std::vector<long long> cut;
    long long rows = 0;
    bool contig = true;
    exchange_slices(v2snp, n_variants, nranks, cut, rows, contig);
    for (int r = 0; r <= nranks; r++) cuts[r] = cut[r];
    if (slice_rows) *slice_rows = rows;
    if (contiguous) *contiguous = contig ? 1 : 0;
    return 0;
}

int dmx_runtime_info(char *out, int64_t capacity)
{
    if (!out || capacity <= 0) return fail(DMX_ERR_INVALID, "null buffer");
    std::string text;
    for (const auto &h : mapped_files("libamdhip64")) text += "hip=" + h + "\n";
    for (const auto &h : mapped_files("librccl")) text += "rccl_mapped=" + h + "\n";
    text += "rccl_loaded=" + (g_rccl.handle ? g_rccl.path : std::string("")) + "\n";
    std::snprintf(out, (size_t)capacity, "%s", text.c_str());
    return 0;
}

int dmx_comm_unique_id(void *id_out)
{
    if (!id_out) return fail(DMX_ERR_INVALID, "null id buffer");
    DMX_TRY(load_rccl());
    static_assert(sizeof(ncclUniqueId) == DMX_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId id;
    ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return fail(DMX_ERR_RCCL, "ncclGetUniqueId failed (%d)", (int)r);
    std::memcpy(id_out, &id, sizeof id);
    return 0;
}

int dmx_comm_init(dmx_ctx *c, int rank, int nranks, const void *unique_id, int reduce_dtype)
{
    DMX_TRY(bind(c));
    if (nranks < 1 || rank < 0 || rank >= nranks || !unique_id) return fail(DMX_ERR_INVALID, "bad communicator arguments");
    if (reduce_dtype != DMX_F32 && reduce_dtype != DMX_F64) return fail(DMX_ERR_INVALID, "reduce_dtype must be DMX_F32 or DMX_F64");
    DMX_TRY(load_rccl());
    if (c->comm) {
        g_rccl.CommDestroy(c->comm);
        c->comm = nullptr;
    }
    c->host_coll = nullptr;
    c->emulated = false;
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof id);
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        c->comm = nullptr;
        return fail(DMX_ERR_RCCL, "ncclCommInitRank failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    }
    c->rank = rank;
    c->nranks = nranks;
    c->reduce_dtype = reduce_dtype;
    // a problem installed before the communicator gets its exchange layout now (the E-step records are rewritten
    // in place for the padded genotype table)
    if (c->have_problem) DMX_TRY(layout_exchange(c));
    return 0;
}

int dmx_comm_init_host(dmx_ctx *c, int rank, int nranks, dmx_host_collective collective, void *user, int reduce_dtype)
{
    DMX_TRY(bind(c));
    if (nranks < 1 || rank < 0 || rank >= nranks || !collective) return fail(DMX_ERR_INVALID, "bad communicator arguments");
    if (reduce_dtype != DMX_F32 && reduce_dtype != DMX_F64) return fail(DMX_ERR_INVALID, "reduce_dtype must be DMX_F32 or DMX_F64");
    if (c->comm) {
        g_rccl.CommDestroy(c->comm);
        c->comm = nullptr;
    }
    c->host_coll = collective;
    c->emulated = false;
    c->host_user = user;
    c->rank = rank;
    c->nranks = nranks;
    c->reduce_dtype = reduce_dtype;
    if (c->have_problem) DMX_TRY(layout_exchange(c));
    return 0;
}

int dmx_comm_init_emulated(dmx_ctx *c, int rank, int nranks, double link_gbytes_per_s, double latency_us, int reduce_dtype)
{
    DMX_TRY(bind(c));
    if (nranks < 1 || rank < 0 || rank >= nranks || !(link_gbytes_per_s > 0) || !(latency_us >= 0)) return fail(DMX_ERR_INVALID, "bad emulated communicator arguments");
    if (reduce_dtype != DMX_F32 && reduce_dtype != DMX_F64) return fail(DMX_ERR_INVALID, "reduce_dtype must be DMX_F32 or DMX_F64");
    if (c->comm) {
        g_rccl.CommDestroy(c->comm);
        c->comm = nullptr;
    }
    c->host_coll = nullptr;
    int khz = 0;
    HIP_TRY(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device));
    if (khz <= 0) return fail(DMX_ERR_UNSUPPORTED, "the device reports no wall clock rate");
    c->emu_ticks_per_ns = khz * 1e-6;
    c->emulated = true;
    c->emu_link_gbps = link_gbytes_per_s;
    c->emu_latency_us = latency_us;
    c->rank = rank;
    c->nranks = nranks;
    c->reduce_dtype = reduce_dtype;
    if (c->have_problem) DMX_TRY(layout_exchange(c));
    return 0;
}

int dmx_get_exchange_mode(dmx_ctx *c, int32_t *mode)
{
    if (!c || !mode) return fail(DMX_ERR_INVALID, "null argument");
    *mode = !c->attached() ? DMX_EXCHANGE_NONE : c->mshard ? DMX_EXCHANGE_VARIANT : c->sliced ? DMX_EXCHANGE_REDUCE_SCATTER : DMX_EXCHANGE_ALLREDUCE;
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
    if (c->d_incr_state) HIP_TRY(hipMemsetAsync(c->d_incr_state + 2 * dmx::IS_WORDS, 0, sizeof(unsigned) * dmx::IS_WORDS, c->stream));
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
