// dmx_exchange.cpp -- RCCL, loaded on demand, and the multi-GPU exchange: variant slices, the variant-sharded M-step's set-up,
// the collectives (RCCL / caller-provided / emulated wire), the all-gather of the posterior tables (include/demux_hip.h "Multi-GPU").
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
namespace host {

void comm_destroy(dmx_ctx *c)
{
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
}

static const char *rccl_error(ncclResult_t r) { return g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"; }

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
    if (c->in_group) {  // one launch: one delay, at the end of the group (coll_group_end; the group's collectives are on c->stream)
        c->group_ns += ns;
        return 0;
    }
    HIP_TRY(dmx::launch_delay(st, (long long)(ns * c->emu_ticks_per_ns)));
    return 0;
}

// The host's side of k_post_counts: the counts are in pinned memory once `seq` stands behind them.  Polling that word costs the host
// what the kernel takes to get there; hipStreamSynchronize costs an interrupt and a wake-up on top (15 us per exchange at 8 ranks), and
// the kernels enqueued behind k_post_counts - which do not depend on what the host decides - keep the GPU busy meanwhile.
int wait_counts(dmx_ctx *c, unsigned *counts, int n, unsigned seq)
{
    for (unsigned spins = 1;; spins++) {
        if (__atomic_load_n(&counts[n], __ATOMIC_ACQUIRE) == seq) return 0;
        if ((spins & 1023u) != 0) continue;
        const hipError_t e = hipStreamQuery(c->stream);
        if (e == hipErrorNotReady) continue;
        if (e != hipSuccess) return fail(DMX_ERR_HIP, "the exchange of the lists failed: %s", hipGetErrorString(e));
        HIP_TRY(hipStreamSynchronize(c->stream));  // (the stream has drained: the word is there, or the memory is not coherent - then it is now)
        if (__atomic_load_n(&counts[n], __ATOMIC_ACQUIRE) == seq) return 0;
        return fail(DMX_ERR_HIP, "the lists' lengths did not arrive");
    }
}

// Several collectives as one launch (ncclGroupStart / ncclGroupEnd); the host-staged and emulated backends run them one
// after the other.
void coll_group_begin(dmx_ctx *c)
{
    c->in_group = true;
    c->group_paid = false;
    c->group_ns = 0.0;
    if (c->comm && g_rccl.GroupStart) (void)g_rccl.GroupStart();
}

int coll_group_end(dmx_ctx *c)
{
    c->in_group = false;
    if (c->emulated && c->group_ns > 0.0) HIP_TRY(dmx::launch_delay(c->stream, (long long)(c->group_ns * c->emu_ticks_per_ns)));
    c->group_ns = 0.0;
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
    // Compact exchange of the posterior rows (gather_posteriors): a barcode with ONE live posterior - 82 % of them on the converged 200k x 100k x 64
    // experiment - is described by its 8-byte code; only the rows of the others - and of those only the ones that differ from what
    // was sent last (d_post_sent) - travel, in a list of at most rows_pad / 4 per rank (beyond that: the whole table, as until round 6).  G <= 64 (the codes exist).
    // DEMUXALOT_AMD_EXCHANGE_COMPACT=0 switches it off, =<n> sets the capacity to n rows (tests: the overflow path).
    c->post_compact_words = 0;
    c->post_compact_cap = 0;
    const char *compact = std::getenv("DEMUXALOT_AMD_EXCHANGE_COMPACT");
    const long long asked = compact ? atoll(compact) : -1;
    if (G <= 64 && asked != 0 && n > 1) {
        c->post_compact_cap = (unsigned)(asked > 0 ? std::min<long long>(asked, rows_pad) : std::max<long long>(64, rows_pad / 4));
        c->post_compact_words = 4 + (size_t)c->post_compact_cap * (size_t)(1 + G);
        c->post_cap_now = c->post_compact_cap;
        DMX_TRY(dev_alloc(c, &c->d_post_compact, c->post_compact_words * (size_t)n));
        HIP_TRY(hipMemsetAsync(c->d_post_compact, 0, sizeof(unsigned) * (c->post_compact_words * (size_t)n), st));
        DMX_TRY(dev_alloc(c, &c->d_post_seen, (size_t)c->rows_total));
        HIP_TRY(hipMemsetAsync(c->d_post_seen, 0xFF, sizeof(uint2) * (size_t)c->rows_total, st));
        DMX_TRY(dev_alloc(c, &c->d_post_sent, (size_t)std::max<long long>(1, c->B) * G));
        DMX_TRY(dev_alloc(c, &c->d_post_sent_multi, (size_t)std::max<long long>(1, c->B)));
        HIP_TRY(hipMemsetAsync(c->d_post_sent_multi, 0, (size_t)std::max<long long>(1, c->B), st));  // (nothing sent yet: every such row is listed)
        HIP_TRY(hipHostMalloc((void **)&c->h_post_counts, sizeof(unsigned) * (size_t)(n + 1), hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(c->h_post_counts, 0, sizeof(unsigned) * (size_t)(n + 1));
    }
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
        std::vector<int> row_variant((size_t)std::max<long long>(1, c->prob_rows), 0);
        for (long long v = 0; v < V; v++) row_variant[(size_t)prow[v]] = (int)v;
        DMX_TRY(dev_alloc(c, &c->d_row_variant, (size_t)c->prob_rows));
        HIP_TRY(hipMemcpyAsync(c->d_row_variant, row_variant.data(), sizeof(int) * (size_t)c->prob_rows, hipMemcpyHostToDevice, st));
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
        // Compact exchange of the table (run_pstep): the rows of a rank's slice that changed since it sent them, in a list of at most
        // slice_rows / 4 (beyond that: the whole slices, as until round 6).  DEMUXALOT_AMD_EXCHANGE_COMPACT=0 switches it off, =<n>: capacity n.
        {
            const char *compact = std::getenv("DEMUXALOT_AMD_EXCHANGE_COMPACT");
            const long long asked = compact ? atoll(compact) : -1;
            c->prob_list_words = 0;
            c->prob_list_cap = 0;
            c->prob_prev_valid = false;
            if (asked != 0 && n > 1 && rows > 0) {
                c->prob_list_cap = (unsigned)(asked > 0 ? std::min<long long>(asked, rows) : std::max<long long>(64, rows / 4));
                c->prob_list_words = 4 + (size_t)c->prob_list_cap * (size_t)(1 + G);
                c->prob_cap_now = c->prob_list_cap;
                DMX_TRY(dev_alloc(c, &c->d_prob_list, c->prob_list_words * (size_t)n));
                HIP_TRY(hipMemsetAsync(c->d_prob_list, 0, sizeof(unsigned) * (c->prob_list_words * (size_t)n), st));
                DMX_TRY(dev_alloc(c, &c->d_prob_prev, (size_t)rows * G));
                if (!c->h_prob_counts) {
                    HIP_TRY(hipHostMalloc((void **)&c->h_prob_counts, sizeof(unsigned) * (size_t)(n + 1), hipHostMallocMapped | hipHostMallocCoherent));
                    std::memset(c->h_prob_counts, 0, sizeof(unsigned) * (size_t)(n + 1));
                }
            }
        }
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
    const bool compact = c->post_compact_words != 0;
    // The lists of THIS exchange hold four times what the longest list of the last one held + 512 rows (at most rows / 4): every rank has read every
    // count, so every rank sizes alike - at convergence a list is 2 % of the rows, not 25 %.  A list that outgrows that overflows:
    // the whole table travels and the next lists are full-sized again.
    const unsigned cap_now = compact ? std::max(1u, std::min(c->post_cap_now, c->post_compact_cap)) : 0u;
    const size_t words_now = 4 + (size_t)cap_now * (size_t)(1 + G);
    if (compact)  // this rank's rows with several live posteriors, listed (the count may run beyond the capacity: overflow)
        HIP_TRY(dmx::launch_post_compact_build(c->stream, c->d_first_g + c->rank * rows, c->d_post_g + c->rank * rows * G, c->B, G, cap_now,
                                               c->d_post_compact + (size_t)c->rank * words_now, c->d_post_sent, c->d_post_sent_multi,
                                               c->emulated ? c->d_post_compact : nullptr, (unsigned long long)words_now, c->nranks, c->rank));
    coll_group_begin(c);  // one launch for the tables
    rc = coll_all_gather(c, (float *)c->d_first_g, rows * 2, "posterior codes");
    if (rc == 0) rc = coll_all_gather(c, (float *)c->d_nz_g, rows * W * 2, "posterior bitmaps");
    if (rc == 0 && compact) rc = coll_all_gather(c, (float *)c->d_post_compact, words_now, "listed posterior rows");
    if (rc == 0 && !compact) rc = coll_all_gather(c, c->d_post_g, rows * G, "singlet posteriors");
    int rc_end = coll_group_end(c);
    if (rc == 0) rc = rc_end;
    if (rc == 0 && compact) {
        // Every rank reads every rank's count: the same decision everywhere.  A small kernel writes the counts into pinned memory (a copy per
        // rank was 8 x 8 us at 8 ranks, one strided hipMemcpy2DAsync 70 us of runtime overhead) and the host polls for them (wait_counts) while
        // the rows are rebuilt: that launch does not wait for the decision - should a list have overflowed, the whole table overwrites what it wrote.
        const unsigned seq = ++c->list_seq;
        HIP_TRY(dmx::launch_post_counts(c->stream, c->d_post_compact, (unsigned long long)words_now, c->nranks, c->h_post_counts, seq));
        HIP_TRY(dmx::launch_post_reconstruct(c->stream, c->d_first_g, c->d_post_g, c->d_post_compact, (unsigned long long)words_now,
                                             (long long)rows, G, c->nranks, c->rank, cap_now, c->d_post_seen));
        DMX_TRY(wait_counts(c, c->h_post_counts, c->nranks, seq));
        unsigned longest = 0;
        for (int r = 0; r < c->nranks; r++) longest = std::max(longest, c->h_post_counts[r]);
        const bool overflow = longest > cap_now;
        if (std::getenv("DEMUXALOT_AMD_EXCHANGE_TRACE")) std::fprintf(stderr, "[posterior lists] longest %u capacity %u of %u\n", longest, cap_now, c->post_compact_cap);
        c->post_cap_now = overflow ? c->post_compact_cap : (unsigned)std::min<unsigned long long>(c->post_compact_cap, 4ull * longest + 512ull);
        if (overflow) {  // (dense posteriors: the first E-steps of a run that starts from uninformative genotypes)
            c->post_compact_overflows++;
            rc = coll_all_gather(c, c->d_post_g, rows * G, "singlet posteriors (the lists overflowed)");
            HIP_TRY(hipMemsetAsync(c->d_post_seen, 0xFF, sizeof(uint2) * (size_t)c->rows_total, c->stream));  // (the rows are the senders' own now)
            if (c->B > 0) {  // ... and what everybody holds of this rank's rows is what they are
                HIP_TRY(hipMemcpyAsync(c->d_post_sent, c->d_post_g + c->rank * rows * G, sizeof(float) * (size_t)c->B * G, hipMemcpyDeviceToDevice, c->stream));
                HIP_TRY(hipMemsetAsync(c->d_post_sent_multi, 1, (size_t)c->B, c->stream));
            }
        } else {
            c->post_compact_taken++;
        }
    }
    timer_end(c, DMX_T_ALLREDUCE, ev);
    if (rc) return rc;
    c->post_gathered = true;
    return 0;
}

}  // namespace host
}  // namespace dmx

extern "C" {

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

int dmx_get_exchange_compact(dmx_ctx *c, int64_t *taken, int64_t *overflows, int64_t *capacity_rows)
{
    if (!c) return fail(DMX_ERR_INVALID, "null ctx");
    if (taken) *taken = c->post_compact_taken;
    if (overflows) *overflows = c->post_compact_overflows;
    if (capacity_rows) *capacity_rows = c->post_compact_words ? (int64_t)c->post_compact_cap : 0;
    return 0;
}

int dmx_get_exchange_compact_table(dmx_ctx *c, int64_t *taken, int64_t *overflows, int64_t *capacity_rows)
{
    if (!c) return fail(DMX_ERR_INVALID, "null ctx");
    if (taken) *taken = c->prob_compact_taken;
    if (overflows) *overflows = c->prob_compact_overflows;
    if (capacity_rows) *capacity_rows = c->prob_list_words ? (int64_t)c->prob_list_cap : 0;
    return 0;
}

int dmx_get_exchange_mode(dmx_ctx *c, int32_t *mode)
{
    if (!c || !mode) return fail(DMX_ERR_INVALID, "null argument");
    *mode = !c->attached() ? DMX_EXCHANGE_NONE : c->mshard ? DMX_EXCHANGE_VARIANT : c->sliced ? DMX_EXCHANGE_REDUCE_SCATTER : DMX_EXCHANGE_ALLREDUCE;
    return 0;
}

}  // extern "C"
