// repack_device.hip -- GPU-side derivation of the kernels' layouts from the COO columns of the
// reference's `barcode_calls` (variant_id, compressed_cb, p_base_wrong; demuxalot/demux.py:290-300):
//
//   E-step records  barcode-major, calls of a barcode in input order, rows padded to 8 calls, two
//                   calls per 32-byte CallPair (kernels.h)
//   M-step records  variant-major, calls of a variant in input order, {compressed_cb, bits(1 - e)}
//   work items      runs of <= item_calls calls of one variant (dmx_ctx::item_calls); item_ptr per variant
//   work lists      barcodes / items by decreasing length
//
// "In input order" is what makes the float64 sums of the kernels run in np.bincount's order, so the
// two re-orderings are STABLE sorts (rocPRIM LSD radix sort of (key, input index) pairs; the one-time
// repack uses the library primitive, the per-iteration kernels are hand-written).  Row / column
// offsets are read off the sorted keys by binary search (no atomic counters).  Everything stays on
// the ctx stream.
#include <cstring>

#include <cstdlib>
#include <algorithm>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "dmx_ctx.h"

namespace dmx {
namespace {

// range check without atomics in the good case: a wave only touches memory when it saw a bad index
// bad[0..2] = flags (barcode, variant, p_base_wrong out of range), bad[3] = lowest offending call index.
// p_base_wrong must be a probability: the E-step's log assumes a finite argument >= 1e-4, which
// p * (1 - e) + max(e, 1e-4) is exactly when 0 <= e <= 1 (NaN fails both comparisons).
__global__ __launch_bounds__(256) void k_validate(const int *__restrict__ variant, const int *__restrict__ cb,
                                                  const float *__restrict__ p_wrong, long long N, long long B, long long V,
                                                  unsigned long long *__restrict__ bad)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < N;
    const bool bad_b = in && ((unsigned)cb[in ? i : 0] >= (unsigned long long)B);
    const bool bad_v = in && ((unsigned)variant[in ? i : 0] >= (unsigned long long)V);
    const float e = p_wrong[in ? i : 0];
    const bool bad_p = in && !(e >= 0.0f && e <= 1.0f);
    if (__ballot(bad_b || bad_v || bad_p) == 0ull) return;
    if (bad_b || bad_v || bad_p) {
        atomicMax(&bad[bad_b ? 0 : (bad_v ? 1 : 2)], 1ull);
        atomicMin(&bad[3], (unsigned long long)i);
    }
}

// start[k] = first position of key k in the sorted key array (lower bound), k = 0..n_keys (start[n_keys] = N):
// row / column offsets from the sorted order, no atomic counters
__global__ __launch_bounds__(256) void k_boundaries(const unsigned *__restrict__ sorted_keys, long long N, long long n_keys,
                                                    long long *__restrict__ start)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k > n_keys) return;
    long long lo = 0, hi = N;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if ((long long)sorted_keys[mid] < k) lo = mid + 1; else hi = mid;
    }
    start[k] = lo;
}

// from offsets: per-row padded pair counts (mode 0) or per-variant item counts (mode 1), and ~count
// (ascending sort of ~count = longest first) with the ids for the work lists
__global__ __launch_bounds__(256) void k_derive_counts(const long long *__restrict__ start, long long n, int mode,
                                                       int item_calls, long long *__restrict__ derived, unsigned *__restrict__ inv,
                                                       unsigned *__restrict__ ids)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long c = start[i + 1] - start[i];
    derived[i] = mode == 0 ? ((c + 7) / 8) * 4                  // CallPairs of a barcode row padded to 8 calls
                           : (c + item_calls - 1) / item_calls;  // work items of a variant
    if (inv) {
        inv[i] = ~(unsigned)c;
        ids[i] = (unsigned)i;
    }
}

// rows with more calls than each of 3 thresholds: inv_sorted holds ~calls, ascending (longest row first)
__global__ void k_count_longer(const unsigned *__restrict__ inv_sorted, long long n, unsigned t0, unsigned t1, unsigned t2,
                               long long *__restrict__ out)
{
    const unsigned threshold = threadIdx.x == 0 ? t0 : threadIdx.x == 1 ? t1 : t2;
    if (threadIdx.x > 2) return;
    long long lo = 0, hi = n;  // first index whose row has <= threshold calls
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (~inv_sorted[mid] > threshold) lo = mid + 1;
        else hi = mid;
    }
    out[threadIdx.x] = lo;
}

__global__ __launch_bounds__(256) void k_iota(unsigned *__restrict__ out, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (unsigned)i;
}

__global__ __launch_bounds__(256) void k_fill_neutral(CallPair *__restrict__ pairs, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    CallPair p;
    p.row_off[0] = p.row_off[1] = 0u;
    p.keep[0] = p.keep[1] = 0.0f;    // p*0 + 1 = 1, log(1) = +0: padding calls add nothing
    p.floor[0] = p.floor[1] = 1.0f;
    p.reserved[0] = p.reserved[1] = 0u;
    pairs[i] = p;
}

// s = position in the barcode-sorted order; perm[s] = input index of that call
__global__ __launch_bounds__(256) void k_build_pairs(const unsigned *__restrict__ sorted_cb,
                                                     const unsigned *__restrict__ perm,
                                                     const int *__restrict__ variant, const float *__restrict__ p_wrong,
                                                     const long long *__restrict__ row_start,
                                                     const long long *__restrict__ pair_ptr, long long N, unsigned G,
                                                     CallPair *__restrict__ pairs, unsigned *__restrict__ call_rows)
{
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= N) return;
    const unsigned b = sorted_cb[s];
    const unsigned i = perm[s];
    const long long j = s - row_start[b];  // position inside the barcode's row (input order)
    const float e = p_wrong[i];
    CallPair &pr = pairs[pair_ptr[b] + (j >> 1)];
    const int h = (int)(j & 1);
    pr.row_off[h] = (unsigned)variant[i] * G * 4u;   // byte offset of the variant's row in prob[V, G]
    call_rows[2 * (pair_ptr[b] + (j >> 1)) + h] = (unsigned)variant[i];  // the row itself, compact (estep_dict.hip)
    pr.keep[h] = 1.0f - e;                          // float32, numpy's `1 - e`
    pr.floor[h] = e > 1e-4f ? e : 1e-4f;            // numpy's `e.clip(1e-4)`
}

__global__ __launch_bounds__(256) void k_build_csc(const unsigned *__restrict__ perm, const int *__restrict__ cb,
                                                   const float *__restrict__ p_wrong, long long N,
                                                   uint2 *__restrict__ csc)
{
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= N) return;
    const unsigned i = perm[s];
    const float keep = 1.0f - p_wrong[i];  // the M-step only ever needs 1 - e
    csc[s] = make_uint2((unsigned)cb[i], __float_as_uint(keep));
}

__global__ __launch_bounds__(256) void k_build_items(const long long *__restrict__ col_ptr,
                                                     const long long *__restrict__ item_ptr, long long V,
                                                     int item_calls, long long *__restrict__ item_start, int *__restrict__ item_len,
                                                     unsigned *__restrict__ inv_len, unsigned *__restrict__ ids, int *__restrict__ item_variant)
{
    const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    long long it = item_ptr[v];
    for (long long s = col_ptr[v]; s < col_ptr[v + 1]; s += item_calls, it++) {
        const long long len = (col_ptr[v + 1] - s) < item_calls ? (col_ptr[v + 1] - s) : item_calls;
        item_start[it] = s;
        item_len[it] = (int)len;
        item_variant[it] = (int)v;
        inv_len[it] = ~(unsigned)len;  // ascending sort = longest first
        ids[it] = (unsigned)it;
    }
}

// ---- tile-major E-step schedule (kernels.hip: k_estep_tiled) ------------------------------------------------
// Bins of R barcodes with (nearly) equal numbers of 8-call groups: the barcodes, sorted by decreasing length,
// are dealt in R strata of n_bins rows; inside a stratum the longest row goes to the bin that is lightest so
// far (bins sorted by load before every stratum).
__global__ __launch_bounds__(256) void k_assign_stratum(const int *__restrict__ order, const unsigned *__restrict__ sorted_bins,
                                                        int stratum, int R, long long n_bins, long long B,
                                                        const long long *__restrict__ pair_ptr, int *__restrict__ bin_rows,
                                                        unsigned *__restrict__ loads, int *__restrict__ row_slot)
{
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_bins) return;
    const unsigned bin = sorted_bins[j];
    const long long idx = (long long)stratum * n_bins + j;
    int row = -1;
    if (idx < B) {
        row = order[idx];
        loads[bin] += (unsigned)((pair_ptr[row + 1] - pair_ptr[row]) >> 2);
        row_slot[row] = (int)(bin * R + stratum);
    }
    bin_rows[(size_t)bin * R + stratum] = row;
}

__global__ __launch_bounds__(256) void k_bin_keys(const unsigned *__restrict__ loads, long long n_bins, unsigned *__restrict__ inv,
                                                  unsigned *__restrict__ ids)
{
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_bins) return;
    inv[j] = ~loads[j];
    ids[j] = (unsigned)j;
}

// Tile of every 8-call group of a barcode, made monotone along the row (running maximum), so that "the groups of
// row r in tile t" is always a contiguous run of the row even when the caller's calls are not variant-sorted.
// One wavefront per barcode; MODE 0 counts the groups per (bin, tile, slot) cell, MODE 1 copies them into the
// bin-major stream: cell_start = first stream group of the cell, row_tile_start = groups of the row before the tile.
template <int MODE>
__global__ __launch_bounds__(256) void k_tile_groups(const CallPair *__restrict__ pairs, const long long *__restrict__ pair_ptr,
                                                     const int *__restrict__ row_slot, long long B, unsigned row_bytes,
                                                     unsigned tile_rows, int n_tiles, int R, unsigned *__restrict__ cnt,
                                                     const unsigned *__restrict__ cell_start,
                                                     const unsigned *__restrict__ row_tile_start, CallPair *__restrict__ stream)
{
    const int lane = threadIdx.x & 63;
    const long long b = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int slot = row_slot[b];
    const long long bin = slot / R;
    const int r = slot % R;
    const long long p0 = pair_ptr[b];
    const long long n_groups = (pair_ptr[b + 1] - p0) >> 2;
    unsigned carry = 0;  // largest tile seen in earlier groups of the row
    for (long long j0 = 0; j0 < n_groups; j0 += 64) {
        const long long j = j0 + lane;
        unsigned tile = j < n_groups ? pairs[p0 + 4 * j].row_off[0] / row_bytes / tile_rows : 0u;
        tile = max(tile, carry);
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {  // inclusive prefix maximum over the lanes
            const unsigned other = __shfl_up(tile, off);
            if (lane >= off) tile = max(tile, other);
        }
        carry = __shfl(tile, 63);
        if (j >= n_groups) continue;
        const size_t cell = ((size_t)bin * n_tiles + tile) * R + r;
        if (MODE == 0) {
            atomicAdd(&cnt[cell], 1u);
        } else {
            const size_t dst = (size_t)cell_start[cell] + (size_t)(j - row_tile_start[cell]);
            const uint4 *src4 = (const uint4 *)(pairs + p0 + 4 * j);
            uint4 *dst4 = (uint4 *)(stream + 4 * dst);
#pragma unroll
            for (int q = 0; q < 8; q++) {
                uint4 w = src4[q];
                if (q == 1) w.z = (unsigned)r;  // CallPair.reserved[0] of the group's first pair: accumulator slot
                dst4[q] = w;
            }
        }
    }
}

// row_tile_start[bin][t][r] = groups of the slot's barcode in tiles < t (one thread per (bin, slot))
__global__ __launch_bounds__(256) void k_row_tile_starts(const unsigned *__restrict__ cnt, long long n_bins, int n_tiles, int R,
                                                         unsigned *__restrict__ row_tile_start)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_bins * R) return;
    const long long bin = i / R;
    const int r = (int)(i % R);
    unsigned run = 0;
    for (int t = 0; t < n_tiles; t++) {
        const size_t cell = ((size_t)bin * n_tiles + t) * R + r;
        row_tile_start[cell] = run;
        run += cnt[cell];
    }
}

__global__ __launch_bounds__(256) void k_bin_ptr(const unsigned *__restrict__ cell_start, long long n_bins, size_t cells_per_bin,
                                                 long long total_groups, long long *__restrict__ bin_ptr)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_bins) return;
    bin_ptr[i] = i < n_bins ? (long long)cell_start[(size_t)i * cells_per_bin] : total_groups;
}

inline unsigned grid_for(long long n) { return (unsigned)((n + 255) / 256); }

inline unsigned bits_for(unsigned long long max_value)
{
    unsigned b = 1;
    while (b < 64 && (max_value >> b) != 0) b++;
    return b;
}

// scratch owner: frees everything on scope exit
// Temporaries of one repack, taken from and returned to the context's block cache (dmx_ctx.h: ctx_malloc).  They go
// back while the kernels that use them may still be queued: the context's next user of such a block runs behind them
// on the same stream.
struct Scratch {
    dmx_ctx *ctx;
    std::vector<void *> ptrs;
    explicit Scratch(dmx_ctx *c) : ctx(c) {}
    Scratch(const Scratch &) = delete;
    Scratch &operator=(const Scratch &) = delete;
    ~Scratch()
    {
        for (void *p : ptrs) ctx_free(ctx, p);
    }
    template <typename T>
    int get(T **out, size_t count)
    {
        void *p = nullptr;
        const int rc = ctx_malloc(ctx, &p, (count ? count : 1) * sizeof(T));
        if (rc) return rc;
        ptrs.push_back(p);
        *out = (T *)p;
        return 0;
    }
};

int sort_pairs(Scratch &sc, const unsigned *keys_in, unsigned *keys_out, const unsigned *vals_in, unsigned *vals_out,
               size_t n, unsigned end_bit, hipStream_t st)
{
    if (n == 0) return 0;
    size_t bytes = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, end_bit, st));
    char *tmp = nullptr;
    DMX_TRY(sc.get(&tmp, bytes));
    HIP_TRY(rocprim::radix_sort_pairs(tmp, bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, end_bit, st));
    return 0;
}

// out[0..n] = exclusive prefix sums of in[0..n) (out has n+1 entries; the last is the total)
int scan_with_total(Scratch &sc, const long long *in, long long *out, size_t n, hipStream_t st)
{
    HIP_TRY(hipMemsetAsync(out, 0, sizeof(long long), st));
    if (n == 0) return 0;
    size_t bytes = 0;
    HIP_TRY(rocprim::inclusive_scan(nullptr, bytes, in, out + 1, n, rocprim::plus<long long>(), st));
    char *tmp = nullptr;
    DMX_TRY(sc.get(&tmp, bytes));
    HIP_TRY(rocprim::inclusive_scan(tmp, bytes, in, out + 1, n, rocprim::plus<long long>(), st));
    return 0;
}

}  // namespace

// Core: COO columns already on the device (d_variant / d_cb / d_p, N = c->N entries).
static int repack_core(dmx_ctx *c, Scratch &sc, const int *d_variant, const int *d_cb, const float *d_p)
{
    const long long B = c->B, V = c->V, N = c->N;
    const int G = c->G;
    hipStream_t st = c->stream;
    c->item_calls = item_calls_for(N);
#ifdef DMX_EXPERIMENTS  // experiment builds only (make EXPERIMENTS=1)
    if (const char *forced = std::getenv("DEMUXALOT_AMD_ITEM_CALLS")) c->item_calls = std::max(64, std::atoi(forced));
#endif

    // the sorts below carry call indices as 32-bit values
    if (N >= (1LL << 32)) return fail(DMX_ERR_UNSUPPORTED, "%lld calls: one context holds fewer than 2^32 (shard the barcodes)", N);

    // ---- range check ----
    unsigned long long *bad = nullptr;
    DMX_TRY(sc.get(&bad, 4));
    const unsigned long long bad_init[4] = {0, 0, 0, ~0ull};
    HIP_TRY(hipMemcpyAsync(bad, bad_init, sizeof bad_init, hipMemcpyHostToDevice, st));
    if (N) hipLaunchKernelGGL(k_validate, dim3(grid_for(N)), dim3(256), 0, st, d_variant, d_cb, d_p, N, B, V, bad);
    unsigned long long h_bad[4];
    HIP_TRY(hipMemcpyAsync(h_bad, bad, sizeof h_bad, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (h_bad[0] || h_bad[1]) {
        int value = 0;
        (void)hipMemcpy(&value, (h_bad[0] ? d_cb : d_variant) + h_bad[3], sizeof(int), hipMemcpyDeviceToHost);
        return fail(DMX_ERR_INVALID, "%s[%llu]=%d outside [0,%lld)", h_bad[0] ? "compressed_cb" : "variant_id", h_bad[3],
                    value, h_bad[0] ? B : V);
    }
    if (h_bad[2]) {
        float value = 0.0f;
        (void)hipMemcpy(&value, d_p + h_bad[3], sizeof(float), hipMemcpyDeviceToHost);
        return fail(DMX_ERR_INVALID, "p_base_wrong[%llu]=%g outside [0,1]", h_bad[3], (double)value);
    }

    // ---- stable sorts of (key, input index); offsets from the sorted keys ----
    unsigned *iota = nullptr, *keys_b = nullptr, *perm_b = nullptr, *keys_v = nullptr, *perm_v = nullptr;
    DMX_TRY(sc.get(&iota, (size_t)N));
    DMX_TRY(sc.get(&keys_b, (size_t)N));
    DMX_TRY(sc.get(&perm_b, (size_t)N));
    DMX_TRY(sc.get(&keys_v, (size_t)N));
    DMX_TRY(sc.get(&perm_v, (size_t)N));
    if (N) hipLaunchKernelGGL(k_iota, dim3(grid_for(N)), dim3(256), 0, st, iota, N);
    DMX_TRY(sort_pairs(sc, (const unsigned *)d_cb, keys_b, iota, perm_b, (size_t)N, bits_for(B ? B - 1 : 0), st));
    DMX_TRY(sort_pairs(sc, (const unsigned *)d_variant, keys_v, iota, perm_v, (size_t)N, bits_for(V ? V - 1 : 0), st));
    long long *row_start = nullptr, *col_ptr = nullptr, *row_pairs = nullptr, *col_items = nullptr;
    DMX_TRY(sc.get(&row_start, (size_t)B + 1));
    DMX_TRY(sc.get(&col_ptr, (size_t)V + 1));
    DMX_TRY(sc.get(&row_pairs, (size_t)B));
    DMX_TRY(sc.get(&col_items, (size_t)V));
    hipLaunchKernelGGL(k_boundaries, dim3(grid_for(B + 1)), dim3(256), 0, st, keys_b, N, B, row_start);
    hipLaunchKernelGGL(k_boundaries, dim3(grid_for(V + 1)), dim3(256), 0, st, keys_v, N, V, col_ptr);
    unsigned *inv = nullptr, *ids = nullptr, *inv_sorted = nullptr;
    DMX_TRY(sc.get(&inv, (size_t)B));
    DMX_TRY(sc.get(&ids, (size_t)B));
    DMX_TRY(sc.get(&inv_sorted, (size_t)B));
    if (B) hipLaunchKernelGGL(k_derive_counts, dim3(grid_for(B)), dim3(256), 0, st, row_start, B, 0, 0, row_pairs, inv, ids);
    if (V)
        hipLaunchKernelGGL(k_derive_counts, dim3(grid_for(V)), dim3(256), 0, st, col_ptr, V, 1, c->item_calls, col_items, (unsigned *)nullptr,
                           (unsigned *)nullptr);
    DMX_TRY(dev_alloc(c, &c->d_pair_ptr, (size_t)B + 1));
    DMX_TRY(dev_alloc(c, &c->d_item_ptr, (size_t)V + 1));
    DMX_TRY(dev_alloc(c, &c->d_bc_order, (size_t)B));
    DMX_TRY(scan_with_total(sc, row_pairs, c->d_pair_ptr, (size_t)B, st));
    DMX_TRY(scan_with_total(sc, col_items, c->d_item_ptr, (size_t)V, st));
    DMX_TRY(sort_pairs(sc, inv, inv_sorted, ids, (unsigned *)c->d_bc_order, (size_t)B, 32, st));  // longest rows first
    long long n_pairs = 0, n_items = 0;
    unsigned longest = ~0u;  // ~calls of the longest row
    if (B) HIP_TRY(hipMemcpyAsync(&longest, inv_sorted, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    // how many rows are longer than a third of the calls a SIMD gets at 8 / 4 / 2 barcodes per wavefront (the packed
    // E-step lets those walk on 64 lanes: dmx_api.cpp, run_estep)
    long long h_longer[3] = {0, 0, 0}, *d_longer = nullptr;
    if (!c->n_simd) {
        int cus = 0;
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
        c->n_simd = 4 * cus;
    }
    for (int k = 0; k < 3; k++) c->long_row_calls[k] = N / (3 * (8 >> k) * (long long)std::max(1, c->n_simd));
    if (B) {
        DMX_TRY(sc.get(&d_longer, 3));
        hipLaunchKernelGGL(k_count_longer, dim3(1), dim3(64), 0, st, inv_sorted, B, (unsigned)std::min<long long>(c->long_row_calls[0], 0xFFFFFFFEll),
                           (unsigned)std::min<long long>(c->long_row_calls[1], 0xFFFFFFFEll), (unsigned)std::min<long long>(c->long_row_calls[2], 0xFFFFFFFEll), d_longer);
        HIP_TRY(hipMemcpyAsync(h_longer, d_longer, sizeof(h_longer), hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipMemcpyAsync(&n_pairs, c->d_pair_ptr + B, sizeof(long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&n_items, c->d_item_ptr + V, sizeof(long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (n_items >= (1LL << 31)) return fail(DMX_ERR_UNSUPPORTED, "too many M-step work items");
    c->n_pairs = n_pairs;
    c->n_items = n_items;
    c->max_row_calls = (long long)(~longest);
    for (int k = 0; k < 3; k++) c->n_long_rows[k] = h_longer[k];

    // barcode-major -> E-step records
    // (CALL_PAD_PAIRS neutral records behind the last row: the dictionary form reads whole super-batches)
    const long long padded_pairs = n_pairs + CALL_PAD_PAIRS;
    DMX_TRY(dev_alloc(c, &c->d_call_pairs, (size_t)padded_pairs));
    DMX_TRY(dev_alloc(c, &c->d_call_rows, (size_t)padded_pairs * 2));
    hipLaunchKernelGGL(k_fill_neutral, dim3(grid_for(padded_pairs)), dim3(256), 0, st, c->d_call_pairs, padded_pairs);
    HIP_TRY(hipMemsetAsync(c->d_call_rows, 0, sizeof(unsigned) * (size_t)padded_pairs * 2, st));
    if (N)
        hipLaunchKernelGGL(k_build_pairs, dim3(grid_for(N)), dim3(256), 0, st, keys_b, perm_b, d_variant, d_p, row_start,
                           c->d_pair_ptr, N, (unsigned)G, c->d_call_pairs, c->d_call_rows);
    // ---- tile-major E-step schedule, when the shape calls for it ----
    c->n_bins = 0;
    c->n_tiles = 0;
    c->bin_rows_cap = 0;
    if (G > 16 && G <= 128 && B >= TILE_MIN_BARCODES && V * (long long)G * 4 >= TILE_MIN_TABLE_BYTES) {
        // as many bins as wavefronts the chip holds, times a whole number of rounds: bins have equal work, so a
        // launch is that many rounds long, and a partly filled last round would cost a whole one
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, c->device));
        const long long wave_slots = (long long)prop.multiProcessorCount * 32;
        const long long rounds = (B + TILE_R_MAX * wave_slots - 1) / (TILE_R_MAX * wave_slots);
        const long long n_bins = rounds * wave_slots;
        const int R = (int)((B + n_bins - 1) / n_bins);  // <= TILE_R_MAX
#ifdef DMX_EXPERIMENTS  // experiment builds only (make EXPERIMENTS=1): DEMUXALOT_AMD_TILE_KB
        static const long long tile_bytes = [] {
            const char *e = std::getenv("DEMUXALOT_AMD_TILE_KB");
            return e && atoll(e) > 0 ? atoll(e) * 1024 : TILE_BYTES;
        }();
#else
        const long long tile_bytes = TILE_BYTES;
#endif
        const unsigned tile_rows = (unsigned)std::max<long long>(1, tile_bytes / ((long long)G * 4));
        const int n_tiles = (int)((V + tile_rows - 1) / tile_rows);
        const size_t cells = (size_t)n_bins * n_tiles * R;
        const long long total_groups = n_pairs >> 2;
        if (total_groups >= (1LL << 32)) return fail(DMX_ERR_UNSUPPORTED, "too many call groups for the tile-major schedule");
        DMX_TRY(dev_alloc(c, &c->d_bin_rows, (size_t)n_bins * R));
        DMX_TRY(dev_alloc(c, &c->d_bin_order, (size_t)n_bins));
        DMX_TRY(dev_alloc(c, &c->d_bin_ptr, (size_t)n_bins + 1));
        DMX_TRY(dev_alloc(c, &c->d_tile_stream, (size_t)n_pairs));
        c->n_bins = n_bins;  // (set before any failure below so that release_problem frees with the right sizes)
        c->n_tiles = n_tiles;
        c->bin_rows_cap = R;
        unsigned *loads = nullptr, *bin_ids = nullptr, *keys_tmp = nullptr, *sorted_bins = nullptr, *inv_l = nullptr;
        unsigned *cnt = nullptr, *cell_start = nullptr, *row_tile_start = nullptr;
        int *row_slot = nullptr;
        DMX_TRY(sc.get(&loads, (size_t)n_bins));
        DMX_TRY(sc.get(&bin_ids, (size_t)n_bins));
        DMX_TRY(sc.get(&keys_tmp, (size_t)n_bins));
        DMX_TRY(sc.get(&sorted_bins, (size_t)n_bins));
        DMX_TRY(sc.get(&inv_l, (size_t)n_bins));
        DMX_TRY(sc.get(&row_slot, (size_t)B));
        DMX_TRY(sc.get(&cnt, cells));
        DMX_TRY(sc.get(&cell_start, cells));
        DMX_TRY(sc.get(&row_tile_start, cells));
        HIP_TRY(hipMemsetAsync(loads, 0, sizeof(unsigned) * n_bins, st));
        hipLaunchKernelGGL(k_iota, dim3(grid_for(n_bins)), dim3(256), 0, st, bin_ids, n_bins);
        for (int stratum = 0; stratum < R; stratum++) {
            DMX_TRY(sort_pairs(sc, loads, keys_tmp, bin_ids, sorted_bins, (size_t)n_bins, 32, st));  // lightest bin first (stable)
            hipLaunchKernelGGL(k_assign_stratum, dim3(grid_for(n_bins)), dim3(256), 0, st, c->d_bc_order, sorted_bins, stratum, R, n_bins,
                               B, c->d_pair_ptr, c->d_bin_rows, loads, row_slot);
        }
        hipLaunchKernelGGL(k_bin_keys, dim3(grid_for(n_bins)), dim3(256), 0, st, loads, n_bins, inv_l, bin_ids);
        DMX_TRY(sort_pairs(sc, inv_l, keys_tmp, bin_ids, (unsigned *)c->d_bin_order, (size_t)n_bins, 32, st));  // heaviest bin first
        // groups per (bin, tile, slot) cell -> position of every cell / of every row's tile run in the bin-major stream
        HIP_TRY(hipMemsetAsync(cnt, 0, sizeof(unsigned) * cells, st));
        const dim3 row_grid((unsigned)((B + 3) / 4));
        hipLaunchKernelGGL(k_tile_groups<0>, row_grid, dim3(256), 0, st, c->d_call_pairs, c->d_pair_ptr, row_slot, B, (unsigned)G * 4u,
                           tile_rows, n_tiles, R, cnt, (const unsigned *)nullptr, (const unsigned *)nullptr, (CallPair *)nullptr);
        {
            size_t bytes = 0;
            HIP_TRY(rocprim::exclusive_scan(nullptr, bytes, cnt, cell_start, 0u, cells, rocprim::plus<unsigned>(), st));
            char *tmp;
            DMX_TRY(sc.get(&tmp, bytes));
            HIP_TRY(rocprim::exclusive_scan(tmp, bytes, cnt, cell_start, 0u, cells, rocprim::plus<unsigned>(), st));
        }
        hipLaunchKernelGGL(k_row_tile_starts, dim3(grid_for(n_bins * R)), dim3(256), 0, st, cnt, n_bins, n_tiles, R, row_tile_start);
        hipLaunchKernelGGL(k_bin_ptr, dim3(grid_for(n_bins + 1)), dim3(256), 0, st, cell_start, n_bins, (size_t)n_tiles * R, total_groups,
                           c->d_bin_ptr);
        hipLaunchKernelGGL(k_tile_groups<1>, row_grid, dim3(256), 0, st, c->d_call_pairs, c->d_pair_ptr, row_slot, B, (unsigned)G * 4u,
                           tile_rows, n_tiles, R, (unsigned *)nullptr, cell_start, row_tile_start, c->d_tile_stream);
    }

    // variant-major -> M-step records
    DMX_TRY(dev_alloc(c, &c->d_csc, (size_t)N));
    c->n_csc = N;
    if (N) hipLaunchKernelGGL(k_build_csc, dim3(grid_for(N)), dim3(256), 0, st, perm_v, d_cb, d_p, N, c->d_csc);

    // ---- work items and their length-sorted list ----
    DMX_TRY(dev_alloc(c, &c->d_item_start, (size_t)n_items));
    DMX_TRY(dev_alloc(c, &c->d_item_len, (size_t)n_items));
    DMX_TRY(dev_alloc(c, &c->d_item_order, (size_t)n_items));
    DMX_TRY(dev_alloc(c, &c->d_item_variant, (size_t)n_items));
    unsigned *inv_i = nullptr, *ids_i = nullptr, *keys_out = nullptr;
    DMX_TRY(sc.get(&inv_i, (size_t)n_items));
    DMX_TRY(sc.get(&ids_i, (size_t)n_items));
    DMX_TRY(sc.get(&keys_out, (size_t)n_items));
    if (V)
        hipLaunchKernelGGL(k_build_items, dim3(grid_for(V)), dim3(256), 0, st, col_ptr, c->d_item_ptr, V, c->item_calls,
                           c->d_item_start, c->d_item_len, inv_i, ids_i, c->d_item_variant);
    DMX_TRY(sort_pairs(sc, inv_i, keys_out, ids_i, (unsigned *)c->d_item_order, (size_t)n_items, 32, st));
    // the variant-major offsets stay on the host: what cuts the tiles of the tile-major M-step (build_mstep_tiles)
    c->h_col_ptr.resize((size_t)V + 1);
    HIP_TRY(hipMemcpyAsync(c->h_col_ptr.data(), col_ptr, sizeof(long long) * (V + 1), hipMemcpyDeviceToHost, st));

    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));  // scratch is released by the caller's Scratch
    return 0;
}

// ------------------------------------------------------------------------------------
// Multi-GPU: M-step records by VARIANT SLICE (dmx_api.cpp: shard_mstep_by_variant).  Every rank's variant-major call
// records travel once, at set-up; each rank keeps the calls of ITS variant slice - from the barcodes of all ranks, the
// barcode a global row r * rows_per_rank + b - and derives from them what repack_core derives for the M-step: the
// variant-major records, the work items and their length-sorted list.  The gathered buffer is rank-major and the ranks hold
// consecutive barcode ranges, so the stable sort by variant leaves every variant's calls rank after rank, each rank's in
// its own input order.  PRECONDITION of the bit-identity with one GPU: inside a variant the caller's calls ascend by barcode -
// what the reference's molecule_calls2barcode_calls (np.unique: demux.py:278-300), this library's packs and synth.py all
// produce; the order is then the ascending GLOBAL barcode order, i.e. np.bincount's (demux.py:113-118) over the whole
// experiment.  For calls handed to dmx_set_problem in another order one GPU sums a variant in the given order and n ranks
// rank after rank: the same float64 terms in another order, a float32 rounding tie at most.
// ------------------------------------------------------------------------------------
namespace {
// record of one call on the wire: {variant, global barcode row, bits of 1 - p_base_wrong, 1 (0: padding)}
__global__ __launch_bounds__(256) void k_wire_records(const uint2 *__restrict__ csc, const long long *__restrict__ item_start,
                                                      const int *__restrict__ item_len, const int *__restrict__ item_variant,
                                                      long long n_items, unsigned row_base, uint4 *__restrict__ out)
{
    // one wavefront per work item: its calls are consecutive in csc and belong to one variant
    const int lane = threadIdx.x & 63;
    const long long it = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (it >= n_items) return;
    const long long s0 = item_start[it];
    const unsigned v = (unsigned)item_variant[it];
    for (int i = lane; i < item_len[it]; i += 64) {
        const uint2 r = csc[s0 + i];
        out[s0 + i] = make_uint4(v, r.x + row_base, r.y, 1u);
    }
}

__global__ __launch_bounds__(256) void k_flag_slice(const uint4 *__restrict__ rec, long long n, unsigned v_lo, unsigned v_hi,
                                                    long long *__restrict__ flag)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (rec[i].w != 0u && rec[i].x >= v_lo && rec[i].x < v_hi) ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_compact_slice(const uint4 *__restrict__ rec, long long n, unsigned v_lo, unsigned v_hi,
                                                       const long long *__restrict__ pos, unsigned *__restrict__ variant,
                                                       uint2 *__restrict__ row_keep)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 r = rec[i];
    if (r.w != 0u && r.x >= v_lo && r.x < v_hi) {
        variant[pos[i]] = r.x;
        row_keep[pos[i]] = make_uint2(r.y, r.z);
    }
}

__global__ __launch_bounds__(256) void k_permute_records(const unsigned *__restrict__ perm, const uint2 *__restrict__ in, long long n,
                                                         uint2 *__restrict__ out)
{
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n) out[s] = in[perm[s]];
}
}  // namespace

// this rank's variant-major records in wire form (16 bytes per call), `capacity` entries, the tail zero (padding)
int wire_records_of(dmx_ctx *c, long long row_base, uint4 *d_out, long long capacity)
{
    hipStream_t st = c->stream;
    HIP_TRY(hipMemsetAsync(d_out, 0, sizeof(uint4) * (size_t)capacity, st));
    if (c->n_items)
        hipLaunchKernelGGL(k_wire_records, dim3(grid_for(c->n_items * 64)), dim3(256), 0, st, c->d_csc, c->d_item_start, c->d_item_len,
                           c->d_item_variant, c->n_items, (unsigned)row_base, d_out);
    HIP_TRY(hipGetLastError());
    return 0;
}

// the records of variants [v_lo, v_hi) among `n` gathered wire records become the context's M-step records
int install_mstep_records(dmx_ctx *c, const uint4 *d_rec, long long n, long long v_lo, long long v_hi)
{
    hipStream_t st = c->stream;
    const long long V = c->V;
    Scratch sc(c);
    long long *flag = nullptr, *pos = nullptr;
    DMX_TRY(sc.get(&flag, (size_t)n + 1));
    DMX_TRY(sc.get(&pos, (size_t)n + 1));
    if (n) hipLaunchKernelGGL(k_flag_slice, dim3(grid_for(n)), dim3(256), 0, st, d_rec, n, (unsigned)v_lo, (unsigned)v_hi, flag);
    DMX_TRY(scan_with_total(sc, flag, pos, (size_t)n, st));
    long long m = 0;
    HIP_TRY(hipMemcpyAsync(&m, pos + n, sizeof(long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (m >= (1LL << 32)) return fail(DMX_ERR_UNSUPPORTED, "%lld calls in one variant slice: one context holds fewer than 2^32", m);
    unsigned *variant = nullptr, *keys_v = nullptr, *iota = nullptr, *perm = nullptr;
    uint2 *row_keep = nullptr;
    DMX_TRY(sc.get(&variant, (size_t)m));
    DMX_TRY(sc.get(&keys_v, (size_t)m));
    DMX_TRY(sc.get(&iota, (size_t)m));
    DMX_TRY(sc.get(&perm, (size_t)m));
    DMX_TRY(sc.get(&row_keep, (size_t)m));
    if (n) hipLaunchKernelGGL(k_compact_slice, dim3(grid_for(n)), dim3(256), 0, st, d_rec, n, (unsigned)v_lo, (unsigned)v_hi, pos, variant, row_keep);
    if (m) hipLaunchKernelGGL(k_iota, dim3(grid_for(m)), dim3(256), 0, st, iota, m);
    DMX_TRY(sort_pairs(sc, variant, keys_v, iota, perm, (size_t)m, bits_for(V ? V - 1 : 0), st));
    // the old records and items go, the slice's come
    release_mstep_tiles(c);
    dev_free(c, &c->d_csc, (size_t)c->n_csc);
    dev_free(c, &c->d_item_start, (size_t)c->n_items);
    dev_free(c, &c->d_item_len, (size_t)c->n_items);
    dev_free(c, &c->d_item_order, (size_t)c->n_items);
    dev_free(c, &c->d_item_variant, (size_t)c->n_items);
    dev_free(c, &c->d_partial, (size_t)c->n_items * c->G);
    dev_free(c, &c->d_redo, c->cap_redo);
    c->n_items = 0;
    c->n_csc = 0;
    c->item_calls = item_calls_for(m);
    long long *col_ptr = nullptr, *col_items = nullptr;
    DMX_TRY(sc.get(&col_ptr, (size_t)V + 1));
    DMX_TRY(sc.get(&col_items, (size_t)V));
    hipLaunchKernelGGL(k_boundaries, dim3(grid_for(V + 1)), dim3(256), 0, st, keys_v, m, V, col_ptr);
    if (V)
        hipLaunchKernelGGL(k_derive_counts, dim3(grid_for(V)), dim3(256), 0, st, col_ptr, V, 1, c->item_calls, col_items, (unsigned *)nullptr,
                           (unsigned *)nullptr);
    DMX_TRY(scan_with_total(sc, col_items, c->d_item_ptr, (size_t)V, st));
    long long n_items = 0;
    HIP_TRY(hipMemcpyAsync(&n_items, c->d_item_ptr + V, sizeof(long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (n_items >= (1LL << 31)) return fail(DMX_ERR_UNSUPPORTED, "too many M-step work items");
    DMX_TRY(dev_alloc(c, &c->d_csc, (size_t)m));
    c->n_csc = m;
    if (m) hipLaunchKernelGGL(k_permute_records, dim3(grid_for(m)), dim3(256), 0, st, perm, row_keep, m, c->d_csc);
    c->n_items = n_items;
    DMX_TRY(dev_alloc(c, &c->d_item_start, (size_t)n_items));
    DMX_TRY(dev_alloc(c, &c->d_item_len, (size_t)n_items));
    DMX_TRY(dev_alloc(c, &c->d_item_order, (size_t)n_items));
    DMX_TRY(dev_alloc(c, &c->d_item_variant, (size_t)n_items));
    DMX_TRY(dev_alloc(c, &c->d_partial, (size_t)n_items * c->G));
    c->cap_redo = ((size_t)n_items / 2 + 1) * (size_t)c->G;
    DMX_TRY(dev_alloc(c, &c->d_redo, c->cap_redo));
    unsigned *inv_i = nullptr, *ids_i = nullptr, *keys_out = nullptr;
    DMX_TRY(sc.get(&inv_i, (size_t)n_items));
    DMX_TRY(sc.get(&ids_i, (size_t)n_items));
    DMX_TRY(sc.get(&keys_out, (size_t)n_items));
    if (V)
        hipLaunchKernelGGL(k_build_items, dim3(grid_for(V)), dim3(256), 0, st, col_ptr, c->d_item_ptr, V, c->item_calls, c->d_item_start,
                           c->d_item_len, inv_i, ids_i, c->d_item_variant);
    DMX_TRY(sort_pairs(sc, inv_i, keys_out, ids_i, (unsigned *)c->d_item_order, (size_t)n_items, 32, st));
    c->h_col_ptr.resize((size_t)V + 1);
    HIP_TRY(hipMemcpyAsync(c->h_col_ptr.data(), col_ptr, sizeof(long long) * (V + 1), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

namespace {
// one workgroup per work item: sort key (tile, barcode row) and the tile-major record of every call of the item
__global__ __launch_bounds__(256) void k_mtile_keys(const uint2 *__restrict__ csc, const long long *__restrict__ item_start,
                                                    const int *__restrict__ item_len, const int *__restrict__ item_variant,
                                                    const unsigned *__restrict__ tile_of, const unsigned *__restrict__ vin_of,
                                                    unsigned row_bits, unsigned *__restrict__ keys, uint2 *__restrict__ rec)
{
    const long long item = blockIdx.x;
    const long long s0 = item_start[item];
    const int n = item_len[item];
    const int v = item_variant[item];
    const unsigned tile = tile_of[v], vin = vin_of[v];
    for (int i = threadIdx.x; i < n; i += 256) {
        const uint2 d = csc[s0 + i];
        keys[s0 + i] = (tile << row_bits) | d.x;
        rec[s0 + i] = make_uint2(d.x | (vin << 24), d.y);
    }
}
}  // namespace

namespace {
// One wavefront per barcode: sort key (tile; `sentinel` for the padding calls) and tile-major record of every call of the row,
// taken from the barcode-major E-step records (row offset -> variant, keep).  Written at the call's position in the
// barcode-major order, so that a STABLE sort by the key alone leaves every tile's records in ascending barcode order.
// tile_vin[v] = tile of variant v << 7 | its index inside the tile: ONE gather per call (the two tables took two: 157 M L2 requests,
// which is what the kernel ran on)
__global__ __launch_bounds__(256) void k_mtile_from_rows(const CallPair *__restrict__ pairs, const long long *__restrict__ pair_ptr,
                                                         long long B, unsigned row_bytes, const unsigned *__restrict__ tile_vin,
                                                         unsigned sentinel, unsigned *__restrict__ keys, unsigned long long *__restrict__ rec)
{
    const int lane = threadIdx.x & 63;
    const long long b = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const long long p0 = pair_ptr[b];
    const long long n_pairs = pair_ptr[b + 1] - p0;
    for (long long j = lane; j < n_pairs; j += 64) {  // a lane takes a whole 32-byte record (two calls): coalesced in, coalesced out
        const uint4 lo = ((const uint4 *)(pairs + p0 + j))[0];  // row_off[2], keep[2]
        const uint4 hi = ((const uint4 *)(pairs + p0 + j))[1];  // floor[2], reserved[2]
        const unsigned row_off[2] = {lo.x, lo.y}, keep[2] = {lo.z, lo.w}, floor_bits[2] = {hi.x, hi.y};
        unsigned key[2];
        unsigned long long r[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            // the neutral calls that pad a row to 8 (keep 0, floor 1, row 0); a real call with p_base_wrong == 1 looks the same and
            // is dropped with them: it contributes (posterior x 0)^power = +0 to every sum
            const bool padding = keep[h] == 0u && floor_bits[h] == 0x3F800000u && row_off[h] == 0u;
            const unsigned tv = tile_vin[row_off[h] / row_bytes];
            key[h] = padding ? sentinel : tv >> 7;
            r[h] = (unsigned long long)((unsigned)b | ((tv & 127u) << 24)) | ((unsigned long long)keep[h] << 32);
        }
        *(uint2 *)(keys + 2 * (p0 + j)) = make_uint2(key[0], key[1]);
        *(ulonglong2 *)(rec + 2 * (p0 + j)) = make_ulonglong2(r[0], r[1]);
    }
}

__global__ __launch_bounds__(256) void k_tile_ptr(const unsigned *__restrict__ sorted_keys, long long n, long long n_tiles,
                                                  long long *__restrict__ ptr)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k > n_tiles) return;
    long long lo = 0, hi = n;  // first position whose key is >= k (the sentinel = n_tiles closes the last tile)
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if ((long long)sorted_keys[mid] < k) lo = mid + 1; else hi = mid;
    }
    ptr[k] = lo;
}

int sort_pairs64(Scratch &sc, const unsigned *keys_in, unsigned *keys_out, const unsigned long long *vals_in,
                 unsigned long long *vals_out, size_t n, unsigned end_bit, hipStream_t st)
{
    if (n == 0) return 0;
    size_t bytes = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, end_bit, st));
    char *tmp = nullptr;
    DMX_TRY(sc.get(&tmp, bytes));
    HIP_TRY(rocprim::radix_sort_pairs(tmp, bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, end_bit, st));
    return 0;
}
}  // namespace

void release_mstep_tiles(dmx_ctx *c)
{
    dev_free(c, (unsigned long long **)&c->d_mt_stream, (size_t)c->n_mt_stream);
    c->n_mt_stream = 0;
    dev_free(c, &c->d_mt_ptr, (size_t)c->n_mt + 1);
    dev_free(c, &c->d_mt_first, (size_t)c->n_mt + 1);
    dev_free(c, &c->d_mt_order, (size_t)c->n_mt);
    dev_free(c, &c->d_mt_shift, (size_t)c->n_mt);
    dev_free(c, &c->d_mt_shift_v, (size_t)c->V);
    c->incr_valid = false;
    c->n_mt = 0;
    c->mt_tv = 0;
    c->mt_tried = false;
    c->mt_shift_tried = false;
}

// Tiles of the tile-major M-step (kernels.h: MTileArgs): runs of at most tv variants and (where a run of variants allows) about
// cap calls, cut on the host from the work items' offsets - and with them the fixed-point exponent of every tile (k_mstep_tiles:
// contributions in [0, 1] are added as rint(c 2^shift); the sum of the longest variant's n contributions stays below 2^63, and
// c 2^shift below 2^51, the conversion's range).  false: the problem does not take the tile form (stay with the item form).
namespace {
struct TileCut {
    std::vector<int> tile_first;        // [n_mt + 1]
    std::vector<long long> tile_ptr;    // [n_mt + 1]
    std::vector<unsigned> tile_of, vin_of;  // [V]
    std::vector<int> tile_shift;        // [n_mt]
    int tv = 0;
    unsigned row_bits = 0;
    long long n_mt = 0;
};

bool cut_mstep_tiles(dmx_ctx *c, long long v_lo, long long v_hi, TileCut &t)
{
    const int G = c->G;
    const long long rows = c->mshard ? c->rows_total : c->B, m = c->n_csc;
    if (G < 1 || G > 64 || rows >= (1LL << 24) || m == 0 || m >= (1LL << 32) || v_hi <= v_lo) return false;
    const long long V = c->V;
    // first record of every variant: the repack left the variant-major offsets on the host (dmx_ctx::h_col_ptr)
    if ((long long)c->h_col_ptr.size() != V + 1 || c->h_col_ptr[(size_t)V] != m) return false;  // (not the resident records') stay with the item form
    const std::vector<long long> &first_call = c->h_col_ptr;
    if (first_call[(size_t)v_lo] != 0 || first_call[(size_t)v_hi] != m) return false;  // records outside the range: stay with the item form
    if (!c->n_simd) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess) return false;
        c->n_simd = 4 * cus;
    }
    t.tv = std::max(1, std::min<int>(dmx::MTILE_MAX_VARIANTS, dmx::MTILE_LDS_BYTES / (G * 8)));
    t.row_bits = bits_for(rows ? (unsigned long long)rows - 1 : 0);
    t.tile_of.assign((size_t)V, 0u);
    t.vin_of.assign((size_t)V, 0u);
    // at most a 1024th of the calls per tile (four tiles per CU and more: most tiles end at their 128 variants first; a tile costs
    // its 64 KB of accumulators zeroed and written back - 0.34 / 0.36 / 0.46 / 0.54 ms with caps of 1 / 2 / 3 / 6 per SIMD on
    // 200k x 100k x 64), as long as the sort key (tile, row) fits 32 bits
    long long cap = std::max<long long>(4096, m / std::max(1, c->n_simd));
    for (int attempt = 0;; attempt++) {
        t.tile_first.clear();
        t.tile_ptr.clear();
        long long v = v_lo;
        while (v < v_hi) {
            t.tile_first.push_back((int)v);
            t.tile_ptr.push_back(first_call[(size_t)v]);
            long long w = v + 1;  // a tile takes at least one variant, however many calls it has
            while (w < v_hi && w - v < t.tv && first_call[(size_t)w + 1] - first_call[(size_t)v] <= cap) w++;
            for (long long x = v; x < w; x++) {
                t.tile_of[(size_t)x] = (unsigned)(t.tile_first.size() - 1);
                t.vin_of[(size_t)x] = (unsigned)(x - v);
            }
            v = w;
        }
        const unsigned tile_bits = bits_for(t.tile_first.empty() ? 0 : (unsigned long long)t.tile_first.size() - 1);
        if (tile_bits + t.row_bits <= 32) break;
        if (cap >= m || attempt > 40) return false;  // even the widest tiles are too many for a 32-bit key: item form
        cap *= 2;
    }
    t.n_mt = (long long)t.tile_first.size();
    t.tile_first.push_back((int)v_hi);
    t.tile_ptr.push_back(m);
    t.tile_shift.resize((size_t)t.n_mt);
    for (long long i = 0; i < t.n_mt; i++) {
        long long longest = 1;
        for (long long v = t.tile_first[(size_t)i]; v < t.tile_first[(size_t)i + 1]; v++)
            longest = std::max(longest, first_call[(size_t)v + 1] - first_call[(size_t)v]);
        t.tile_shift[(size_t)i] = std::min(50, 62 - (int)bits_for((unsigned long long)longest));
    }
    return true;
}

// [V] exponent of every variant's tile, on the device (MIncrArgs::shift_v, MstepArgs::fixed_shift_v)
int upload_variant_shifts(dmx_ctx *c, const TileCut &t)
{
    const long long V = c->V;
    std::vector<unsigned char> shift_v((size_t)V);
    for (long long v = 0; v < V; v++) shift_v[(size_t)v] = (unsigned char)t.tile_shift[(size_t)t.tile_of[(size_t)v]];
    if (!c->d_mt_shift_v) DMX_TRY(dev_alloc(c, &c->d_mt_shift_v, (size_t)V));
    HIP_TRY(hipMemcpyAsync(c->d_mt_shift_v, shift_v.data(), (size_t)V, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));  // (the host vector)
    return 0;
}
}  // namespace

namespace {
// one wavefront per work item: its records as (barcode row, {variant, bits(1 - e)})
__global__ __launch_bounds__(256) void k_slice_rows(const uint2 *__restrict__ csc, const long long *__restrict__ item_start, const int *__restrict__ item_len,
                                                    const int *__restrict__ item_variant, long long n_items, unsigned *__restrict__ keys,
                                                    unsigned long long *__restrict__ vals)
{
    const long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= n_items) return;
    const long long s0 = item_start[item];
    const unsigned long long v = (unsigned long long)(unsigned)item_variant[item];
    for (int i = threadIdx.x & 63; i < item_len[item]; i += 64) {
        const uint2 d = csc[s0 + i];
        keys[s0 + i] = d.x;
        vals[s0 + i] = v | ((unsigned long long)d.y << 32);
    }
}
}  // namespace

// Variant-sharded rank, incremental M-step (kernels.h: MIncrArgs::rec): the delta pass adds the differences of the CHANGED barcodes' calls;
// a rank holds its slice's records variant-major only, and a masked walk of all of them (k_mincr_delta_masked) is a read of 8 bytes per
// call of the slice whatever changed - 32 us of an 8-rank iteration of 0.32 ms, 0.1 ms of a 2-rank one.  So the slice's records are
// sorted once more, by barcode row (stable: a row's calls stay in variant order), with a pointer per row of the whole job.
// Leaves d_slice_rec null where it does not apply (the caller stays with the masked walk).
int build_slice_row_index(dmx_ctx *c)
{
    if (c->d_slice_rec != nullptr || c->slice_index_tried) return 0;
    c->slice_index_tried = true;
    const char *env = std::getenv("DEMUXALOT_AMD_SLICE_INDEX");  // =0: the masked walk (tests; a rank short of memory)
    if (env && atoi(env) == 0) return 0;
    const long long m = c->n_csc, rows = c->rows_total;
    if (!c->mshard || m == 0 || rows == 0 || m >= (1LL << 32) || c->d_item_variant == nullptr) return 0;
    hipStream_t st = c->stream;
    Scratch sc(c);
    unsigned *keys = nullptr, *keys_out = nullptr;
    unsigned long long *vals = nullptr;
    DMX_TRY(sc.get(&keys, (size_t)m));
    DMX_TRY(sc.get(&keys_out, (size_t)m));
    DMX_TRY(sc.get(&vals, (size_t)m));
    hipLaunchKernelGGL(k_slice_rows, dim3((unsigned)((c->n_items + 3) / 4)), dim3(256), 0, st, c->d_csc, c->d_item_start, c->d_item_len,
                       c->d_item_variant, c->n_items, keys, vals);
    DMX_TRY(dev_alloc(c, &c->d_slice_rec, (size_t)m));
    c->n_slice_rec = m;
    DMX_TRY(sort_pairs64(sc, keys, keys_out, vals, (unsigned long long *)c->d_slice_rec, (size_t)m, bits_for((unsigned long long)(rows - 1)), st));
    DMX_TRY(dev_alloc(c, &c->d_slice_ptr, (size_t)rows + 1));
    hipLaunchKernelGGL(k_tile_ptr, dim3(grid_for(rows + 1)), dim3(256), 0, st, keys_out, m, rows, c->d_slice_ptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));  // (scratch)
    return 0;
}

// The exponents alone, without the tile-major records (their sort): what the fixed-point WORK-ITEM form adds with (MstepArgs::
// fixed_shift_v) - the same cut, so the same exponents the tile-major form uses should the records be built later.  Leaves
// d_mt_shift_v null when the problem does not take the tile cut (the caller then stays with the float64 item form).
int plan_mstep_shifts(dmx_ctx *c)
{
    if (c->d_mt_shift_v != nullptr || c->mt_shift_tried) return 0;
    c->mt_shift_tried = true;
    if (c->mshard) return 0;  // (a variant-sharded rank: the cut of its slice comes with the records, build_mstep_tiles)
    TileCut t;
    if (!cut_mstep_tiles(c, 0, c->V, t)) return 0;
    return upload_variant_shifts(c, t);
}

int build_mstep_tiles(dmx_ctx *c, long long v_lo, long long v_hi)
{
    release_mstep_tiles(c);
    c->mt_tried = true;
    TileCut cut;
    if (!cut_mstep_tiles(c, v_lo, v_hi, cut)) return 0;
    const int G = c->G;
    const long long m = c->n_csc;
    hipStream_t st = c->stream;
    const long long V = c->V;
    const int tv = cut.tv;
    const unsigned row_bits = cut.row_bits;
    std::vector<int> &tile_first = cut.tile_first;
    std::vector<long long> &tile_ptr = cut.tile_ptr;
    std::vector<unsigned> &tile_of = cut.tile_of, &vin_of = cut.vin_of;
    std::vector<int> &tile_shift = cut.tile_shift;
    const long long n_mt = cut.n_mt;
    std::vector<int> order((size_t)n_mt);
    for (long long i = 0; i < n_mt; i++) order[(size_t)i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
        return tile_ptr[(size_t)x + 1] - tile_ptr[(size_t)x] > tile_ptr[(size_t)y + 1] - tile_ptr[(size_t)y];
    });
    Scratch sc(c);
    unsigned *d_tile_of = nullptr, *d_vin_of = nullptr, *keys = nullptr, *keys_out = nullptr, *iota = nullptr, *perm = nullptr;
    uint2 *rec = nullptr;
    DMX_TRY(sc.get(&d_tile_of, (size_t)V));
    DMX_TRY(sc.get(&d_vin_of, (size_t)V));
    // One context holding all calls of its barcodes (no variant-sharded M-step, table rows = variants): the records come out
    // of the barcode-major E-step records by a STABLE sort on the tile alone - 11 key bits instead of tile + barcode row (29),
    // two radix passes instead of four, the records themselves as the sort's values instead of a gather behind it
    // (200k x 100k x 64: the build 4.1 -> ~1.6 ms).  A tile's lengths follow from the sorted keys.
    if (!c->mshard && !c->sliced && c->d_call_pairs != nullptr && v_lo == 0 && v_hi == V && c->n_pairs > 0 && 2 * c->n_pairs < (1LL << 32)) {
        const size_t n = (size_t)(2 * c->n_pairs);
        unsigned long long *vals = nullptr, *vals_out = nullptr;
        DMX_TRY(sc.get(&keys, n));
        DMX_TRY(sc.get(&keys_out, n));
        DMX_TRY(sc.get(&vals, n));
        DMX_TRY(upload_variant_shifts(c, cut));
        for (long long v = 0; v < V; v++) tile_of[(size_t)v] = (tile_of[(size_t)v] << 7) | vin_of[(size_t)v];  // (vin < 128 = MTILE_MAX_VARIANTS)
        HIP_TRY(hipMemcpyAsync(d_tile_of, tile_of.data(), sizeof(unsigned) * V, hipMemcpyHostToDevice, st));
        DMX_TRY(dev_alloc(c, &vals_out, n));
        c->d_mt_stream = (uint2 *)vals_out;
        c->n_mt_stream = (long long)n;
        hipLaunchKernelGGL(k_mtile_from_rows, dim3((unsigned)((c->B + 3) / 4)), dim3(256), 0, st, c->d_call_pairs, c->d_pair_ptr, c->B,
                           (unsigned)G * 4u, d_tile_of, (unsigned)n_mt, keys, vals);
        DMX_TRY(sort_pairs64(sc, keys, keys_out, vals, vals_out, n, bits_for((unsigned long long)n_mt), st));
        DMX_TRY(dev_alloc(c, &c->d_mt_ptr, (size_t)n_mt + 1));
        hipLaunchKernelGGL(k_tile_ptr, dim3(grid_for(n_mt + 1)), dim3(256), 0, st, keys_out, (long long)n, n_mt, c->d_mt_ptr);
        DMX_TRY(dev_alloc(c, &c->d_mt_first, (size_t)n_mt + 1));
        DMX_TRY(dev_alloc(c, &c->d_mt_order, (size_t)n_mt));
        DMX_TRY(dev_alloc(c, &c->d_mt_shift, (size_t)n_mt));
        HIP_TRY(hipMemcpyAsync(c->d_mt_shift, tile_shift.data(), sizeof(int) * n_mt, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(c->d_mt_first, tile_first.data(), sizeof(int) * (n_mt + 1), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(c->d_mt_order, order.data(), sizeof(int) * n_mt, hipMemcpyHostToDevice, st));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(st));  // host vectors, scratch
        c->n_mt = n_mt;
        c->mt_tv = tv;
        return 0;
    }
    // (the incremental M-step of a variant-sharded rank - kernels.h MIncrArgs::changed_map - or of a rank that exchanges sums: row_variant)
    DMX_TRY(upload_variant_shifts(c, cut));
    DMX_TRY(sc.get(&keys, (size_t)m));
    DMX_TRY(sc.get(&keys_out, (size_t)m));
    DMX_TRY(sc.get(&iota, (size_t)m));
    DMX_TRY(sc.get(&perm, (size_t)m));
    DMX_TRY(sc.get(&rec, (size_t)m));
    HIP_TRY(hipMemcpyAsync(d_tile_of, tile_of.data(), sizeof(unsigned) * V, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_vin_of, vin_of.data(), sizeof(unsigned) * V, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_mtile_keys, dim3((unsigned)c->n_items), dim3(256), 0, st, c->d_csc, c->d_item_start, c->d_item_len, c->d_item_variant,
                       d_tile_of, d_vin_of, row_bits, keys, rec);
    hipLaunchKernelGGL(k_iota, dim3(grid_for(m)), dim3(256), 0, st, iota, m);
    const unsigned tile_bits = bits_for(n_mt > 1 ? (unsigned long long)n_mt - 1 : 0);
    DMX_TRY(sort_pairs(sc, keys, keys_out, iota, perm, (size_t)m, std::max(1u, tile_bits + row_bits), st));
    DMX_TRY(dev_alloc(c, &c->d_mt_stream, (size_t)m));
    c->n_mt_stream = m;
    hipLaunchKernelGGL(k_permute_records, dim3(grid_for(m)), dim3(256), 0, st, perm, rec, m, c->d_mt_stream);
    DMX_TRY(dev_alloc(c, &c->d_mt_ptr, (size_t)n_mt + 1));
    DMX_TRY(dev_alloc(c, &c->d_mt_first, (size_t)n_mt + 1));
    DMX_TRY(dev_alloc(c, &c->d_mt_order, (size_t)n_mt));
    DMX_TRY(dev_alloc(c, &c->d_mt_shift, (size_t)n_mt));
    HIP_TRY(hipMemcpyAsync(c->d_mt_shift, tile_shift.data(), sizeof(int) * n_mt, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(c->d_mt_ptr, tile_ptr.data(), sizeof(long long) * (n_mt + 1), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(c->d_mt_first, tile_first.data(), sizeof(int) * (n_mt + 1), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(c->d_mt_order, order.data(), sizeof(int) * n_mt, hipMemcpyHostToDevice, st));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));  // host vectors, scratch
    c->n_mt = n_mt;
    c->mt_tv = tv;
    return 0;
}

int repack_on_device(dmx_ctx *c, const int32_t *h_variant, const int32_t *h_cb, const float *h_p)
{
    const long long N = c->N;
    hipStream_t st = c->stream;
    Scratch sc(c);
    int *d_variant = nullptr, *d_cb = nullptr;
    float *d_p = nullptr;
    DMX_TRY(sc.get(&d_variant, (size_t)N));
    DMX_TRY(sc.get(&d_cb, (size_t)N));
    DMX_TRY(sc.get(&d_p, (size_t)N));
    if (N) {
        HIP_TRY(hipMemcpyAsync(d_variant, h_variant, sizeof(int) * N, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_cb, h_cb, sizeof(int) * N, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_p, h_p, sizeof(float) * N, hipMemcpyHostToDevice, st));
    }
    return repack_core(c, sc, d_variant, d_cb, d_p);
}

// ------------------------------------------------------------------------------------
// Device pack: Demultiplexer.pack_calls' variant matching + molecule_calls2barcode_calls
// (demuxalot/demux.py:276-300, 332-365) on the GPU, feeding repack_core without a round trip:
//   1. 64-bit keys (chromosome, position, base) of the variants in a hash table -> one or two probes
//      per molecule call -> variant row or "no match";
//   2. order-preserving compaction of the matched calls (prefix sum of the match flags);
//   3. stable radix sort of (variant << 32 | barcode, input index);
//   4. one thread per run of equal keys multiplies the members' p_base_wrong in input order,
//      starting from 1.0f (np.multiply.at semantics) -> unique calls, variant-major.
// ------------------------------------------------------------------------------------
namespace {

__host__ __device__ inline unsigned long long variant_key(int chrom, int pos, unsigned char base)
{
    return ((unsigned long long)(unsigned)chrom << 35) | ((unsigned long long)(unsigned)pos << 3) | (unsigned long long)(base & 7);
}

// Variant matching through an open-addressing hash table of the variants' 64-bit keys (capacity a power of two >= 2 V, linear
// probing): one or two 16-byte probes per molecule call instead of an 18-step binary search through the sorted keys (k_match
// took 11.5 ms of the 25 ms device pack of 78 M calls, most of it the search's dependent loads and one atomic per call on the
// molecule counters of the variants - 72 000 of them on the hottest variant's; the counts now come from the sorted calls).
constexpr unsigned long long EMPTY_SLOT = ~0ull;

__device__ __forceinline__ unsigned long long slot_of(unsigned long long key, unsigned bits)
{
    return (key * 0x9E3779B97F4A7C15ull) >> (64u - bits);
}

__global__ __launch_bounds__(256) void k_table_insert(const int *chrom, const int *pos, const unsigned char *base, long long V,
                                                      unsigned bits, unsigned long long *tkeys, unsigned *trows)
{
    const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const unsigned long long key = variant_key(chrom[v], pos[v], base[v]), mask = (1ull << bits) - 1ull;
    for (unsigned long long h = slot_of(key, bits);; h = (h + 1ull) & mask) {
        const unsigned long long seen = atomicCAS(&tkeys[h], EMPTY_SLOT, key);
        if (seen == EMPTY_SLOT || seen == key) {
            atomicMin(&trows[h], (unsigned)v);  // rows that share a key (only a caller's own arrays could): the lowest, as the sorted search chose
            return;
        }
    }
}

__global__ __launch_bounds__(256) void k_match(const int *chrom, const int *pos, const unsigned char *base, const int *cb,
                                               long long n, const unsigned long long *tkeys, const unsigned *trows,
                                               unsigned bits, int *call_variant, unsigned *flag, int *bad)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long q = variant_key(chrom[i], pos[i], base[i]), mask = (1ull << bits) - 1ull;
    int v = -1;
    for (unsigned long long h = slot_of(q, bits);; h = (h + 1ull) & mask) {
        const unsigned long long k = tkeys[h];
        if (k == q) v = (int)trows[h];
        if (k == q || k == EMPTY_SLOT) break;
    }
    call_variant[i] = v;
    flag[i] = v >= 0 ? 1u : 0u;
    if (v >= 0 && cb[i] < 0) atomicMax(bad, 1);
}

// matched molecule calls per variant (np.bincount(molecule_calls['variant_id']) of demux.py:381) from the calls sorted by
// (variant << 32 | barcode): one thread per variant, two binary searches
__global__ __launch_bounds__(256) void k_variant_counts(const unsigned long long *sorted_keys, long long m, long long V,
                                                        unsigned long long *counts)
{
    const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    auto lower = [&](unsigned long long key) {
        long long lo = 0, hi = m;
        while (lo < hi) {
            const long long mid = (lo + hi) >> 1;
            if (sorted_keys[mid] < key) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    counts[v] = (unsigned long long)(lower((unsigned long long)(v + 1) << 32) - lower((unsigned long long)v << 32));
}

__global__ __launch_bounds__(256) void k_compact_keys(const int *call_variant, const int *cb, const unsigned *flag,
                                                      const unsigned *pos_excl, long long n, unsigned long long *keys,
                                                      unsigned *idx)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flag[i]) return;
    const unsigned o = pos_excl[i];
    keys[o] = ((unsigned long long)(unsigned)call_variant[i] << 32) | (unsigned)cb[i];
    idx[o] = (unsigned)i;
}

__global__ __launch_bounds__(256) void k_heads(const unsigned long long *keys, long long m, unsigned *head)
{
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < m) head[s] = (s == 0 || keys[s] != keys[s - 1]) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void k_products(const unsigned long long *keys, const unsigned *perm, const unsigned *head,
                                                  const unsigned *seg_incl, const float *call_p, long long m,
                                                  int *u_variant, int *u_cb, float *u_p, long long *u_count)
{
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= m || !head[s]) return;
    float p = 1.0f;
    long long t = s;
    do {
        p = p * call_p[perm[t]];  // float32, sequential in input order
        t++;
    } while (t < m && !head[t]);
    const unsigned u = seg_incl[s] - 1u;
    u_variant[u] = (int)(keys[s] >> 32);
    u_cb[u] = (int)(keys[s] & 0xFFFFFFFFull);
    u_p[u] = p;
    u_count[u] = t - s;
}

template <typename T>
int upload(Scratch &sc, T **dst, const T *src, size_t n, hipStream_t st)
{
    DMX_TRY(sc.get(dst, n));
    if (n) HIP_TRY(hipMemcpyAsync(*dst, src, sizeof(T) * n, hipMemcpyHostToDevice, st));
    return 0;
}

}  // namespace

// The calls are already on the device (flat arrays owned by `sc`).
static int pack_core(dmx_ctx *c, Scratch &sc, long long V, const int *var_chrom, const int *var_pos,
                     const unsigned char *var_base, long long n_calls, const int *d_cchrom, const int *d_cpos,
                     const unsigned char *d_cbase, const int *d_ccb, const float *d_cp, long long *n_matched,
                     long long *n_unique, long long *mol_per_variant)
{
    hipStream_t st = c->stream;
    if (n_calls >= (1LL << 32)) return fail(DMX_ERR_UNSUPPORTED, "more than 2^32 molecule calls in one batch");
    // 1. hash table of the variant keys
    int *d_vchrom, *d_vpos;
    unsigned char *d_vbase;
    DMX_TRY(upload(sc, &d_vchrom, var_chrom, (size_t)V, st));
    DMX_TRY(upload(sc, &d_vpos, var_pos, (size_t)V, st));
    DMX_TRY(upload(sc, &d_vbase, var_base, (size_t)V, st));
    const unsigned table_bits = std::max(6u, bits_for((unsigned long long)(2 * V)));
    const size_t table_slots = (size_t)1 << table_bits;
    unsigned long long *tkeys;
    unsigned *trows;
    DMX_TRY(sc.get(&tkeys, table_slots));
    DMX_TRY(sc.get(&trows, table_slots));
    HIP_TRY(hipMemsetAsync(tkeys, 0xFF, sizeof(unsigned long long) * table_slots, st));
    HIP_TRY(hipMemsetAsync(trows, 0xFF, sizeof(unsigned) * table_slots, st));
    if (V) hipLaunchKernelGGL(k_table_insert, dim3(grid_for(V)), dim3(256), 0, st, d_vchrom, d_vpos, d_vbase, V, table_bits, tkeys, trows);
    // 2. match + order-preserving compaction
    int *call_variant, *bad;
    unsigned *flag, *pos_excl;
    dev_free(c, &c->d_mol, (size_t)c->V);
    DMX_TRY(dev_alloc(c, &c->d_mol, (size_t)V));
    unsigned long long *d_mol = c->d_mol;
    DMX_TRY(sc.get(&call_variant, (size_t)n_calls));
    DMX_TRY(sc.get(&flag, (size_t)n_calls + 1));
    DMX_TRY(sc.get(&pos_excl, (size_t)n_calls + 1));
    DMX_TRY(sc.get(&bad, 1));
    HIP_TRY(hipMemsetAsync(bad, 0, sizeof(int), st));
    HIP_TRY(hipMemsetAsync(flag + n_calls, 0, sizeof(unsigned), st));
    if (n_calls)
        hipLaunchKernelGGL(k_match, dim3(grid_for(n_calls)), dim3(256), 0, st, d_cchrom, d_cpos, d_cbase, d_ccb, n_calls,
                           tkeys, trows, table_bits, call_variant, flag, bad);
    {
        size_t bytes = 0;
        HIP_TRY(rocprim::exclusive_scan(nullptr, bytes, flag, pos_excl, 0u, (size_t)n_calls + 1, rocprim::plus<unsigned>(), st));
        char *tmp;
        DMX_TRY(sc.get(&tmp, bytes));
        HIP_TRY(rocprim::exclusive_scan(tmp, bytes, flag, pos_excl, 0u, (size_t)n_calls + 1, rocprim::plus<unsigned>(), st));
    }
    unsigned h_matched = 0;
    int h_bad = 0;
    HIP_TRY(hipMemcpyAsync(&h_matched, pos_excl + n_calls, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&h_bad, bad, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (h_bad) return fail(DMX_ERR_INVALID, "negative barcode index among the matched calls");
    const long long m = h_matched;
    *n_matched = m;
    // 3. stable sort by (variant, barcode)
    unsigned long long *keys, *keys_sorted;
    unsigned *idx, *perm;
    DMX_TRY(sc.get(&keys, (size_t)m));
    DMX_TRY(sc.get(&keys_sorted, (size_t)m));
    DMX_TRY(sc.get(&idx, (size_t)m));
    DMX_TRY(sc.get(&perm, (size_t)m));
    if (n_calls)
        hipLaunchKernelGGL(k_compact_keys, dim3(grid_for(n_calls)), dim3(256), 0, st, call_variant, d_ccb, flag, pos_excl,
                           n_calls, keys, idx);
    // aggregate_on_snps wants the matched molecule calls themselves, grouped by (barcode, SNP)
    if (c->keep_molecule_calls) DMX_TRY(build_snp_groups(c, keys, idx, d_cp, m));
    if (m) {
        size_t bytes = 0;
        const unsigned end_bit = 32 + bits_for(V ? V - 1 : 0);
        HIP_TRY(rocprim::radix_sort_pairs(nullptr, bytes, keys, keys_sorted, idx, perm, (size_t)m, 0u, end_bit, st));
        char *tmp;
        DMX_TRY(sc.get(&tmp, bytes));
        HIP_TRY(rocprim::radix_sort_pairs(tmp, bytes, keys, keys_sorted, idx, perm, (size_t)m, 0u, end_bit, st));
    }
    // matched molecule calls per variant, from the sorted keys
    if (V) hipLaunchKernelGGL(k_variant_counts, dim3(grid_for(V)), dim3(256), 0, st, m ? keys_sorted : keys, m, V, d_mol);
    if (mol_per_variant && V) HIP_TRY(hipMemcpyAsync(mol_per_variant, d_mol, sizeof(long long) * V, hipMemcpyDeviceToHost, st));
    // 4. runs of equal keys -> products in input order
    unsigned *head, *seg_incl;
    DMX_TRY(sc.get(&head, (size_t)m));
    DMX_TRY(sc.get(&seg_incl, (size_t)m));
    long long n_u = 0;
    if (m) {
        hipLaunchKernelGGL(k_heads, dim3(grid_for(m)), dim3(256), 0, st, keys_sorted, m, head);
        size_t bytes = 0;
        HIP_TRY(rocprim::inclusive_scan(nullptr, bytes, head, seg_incl, (size_t)m, rocprim::plus<unsigned>(), st));
        char *tmp;
        DMX_TRY(sc.get(&tmp, bytes));
        HIP_TRY(rocprim::inclusive_scan(tmp, bytes, head, seg_incl, (size_t)m, rocprim::plus<unsigned>(), st));
        unsigned last = 0;
        HIP_TRY(hipMemcpyAsync(&last, seg_incl + (m - 1), sizeof(unsigned), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        n_u = last;
    }
    *n_unique = n_u;
    // unique calls stay resident in the ctx (dmx_get_packed_calls) and feed the layout derivation directly
    dev_free(c, &c->d_u_variant, (size_t)c->n_u);
    dev_free(c, &c->d_u_cb, (size_t)c->n_u);
    dev_free(c, &c->d_u_p, (size_t)c->n_u);
    dev_free(c, &c->d_u_count, (size_t)c->n_u);
    c->n_u = n_u;
    DMX_TRY(dev_alloc(c, &c->d_u_variant, (size_t)n_u));
    DMX_TRY(dev_alloc(c, &c->d_u_cb, (size_t)n_u));
    DMX_TRY(dev_alloc(c, &c->d_u_p, (size_t)n_u));
    DMX_TRY(dev_alloc(c, &c->d_u_count, (size_t)n_u));
    if (m)
        hipLaunchKernelGGL(k_products, dim3(grid_for(m)), dim3(256), 0, st, keys_sorted, perm, head, seg_incl, d_cp, m,
                           c->d_u_variant, c->d_u_cb, c->d_u_p, c->d_u_count);
    HIP_TRY(hipGetLastError());
    c->N = n_u;
    return repack_core(c, sc, c->d_u_variant, c->d_u_cb, c->d_u_p);
}

int pack_on_device(dmx_ctx *c, long long V, const int *var_chrom, const int *var_pos, const unsigned char *var_base,
                   long long n_calls, const int *call_chrom, const int *call_pos, const unsigned char *call_base,
                   const int *call_cb, const float *call_p, long long *n_matched, long long *n_unique,
                   long long *mol_per_variant)
{
    hipStream_t st = c->stream;
    Scratch sc(c);
    int *d_cchrom, *d_cpos, *d_ccb;
    unsigned char *d_cbase;
    float *d_cp;
    DMX_TRY(upload(sc, &d_cchrom, call_chrom, (size_t)n_calls, st));
    DMX_TRY(upload(sc, &d_cpos, call_pos, (size_t)n_calls, st));
    DMX_TRY(upload(sc, &d_cbase, call_base, (size_t)n_calls, st));
    DMX_TRY(upload(sc, &d_ccb, call_cb, (size_t)n_calls, st));
    DMX_TRY(upload(sc, &d_cp, call_p, (size_t)n_calls, st));
    return pack_core(c, sc, V, var_chrom, var_pos, var_base, n_calls, d_cchrom, d_cpos, d_cbase, d_ccb, d_cp, n_matched,
                     n_unique, mol_per_variant);
}

// ------------------------------------------------------------------------------------
// The same from the reference's own containers (CompressedSNPCalls, snp_counter.py:77-139): the packed numpy
// records are uploaded as they are and taken apart here, instead of being flattened field by field on the host
// (demux.py:332-358 does that with one fancy-indexing pass per field).
//   snp_calls record, 13 bytes: int32 molecule_index | int32 snp_position | uint8 base_index | float32 p_base_wrong
//   molecule record, 12 bytes:  int32 compressed_cb  | int32 compressed_ub | float32 p_group_misaligned
// ------------------------------------------------------------------------------------
namespace {

constexpr int SNP_CALL_BYTES = 13, MOLECULE_BYTES = 12;

__global__ __launch_bounds__(256) void k_flatten_container(const unsigned char *__restrict__ snp_calls, long long n,
                                                           const unsigned char *__restrict__ molecules, long long n_molecules,
                                                           int chrom, int *__restrict__ out_chrom, int *__restrict__ out_pos,
                                                           unsigned char *__restrict__ out_base, int *__restrict__ out_cb,
                                                           float *__restrict__ out_p, int *bad)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned char *r = snp_calls + i * SNP_CALL_BYTES;
    int mol, pos;
    float p;
    __builtin_memcpy(&mol, r, 4);
    __builtin_memcpy(&pos, r + 4, 4);
    __builtin_memcpy(&p, r + 9, 4);
    int cb = 0;
    if (mol < 0 || mol >= n_molecules)
        atomicOr(bad, 1);
    else
        __builtin_memcpy(&cb, molecules + (long long)mol * MOLECULE_BYTES, 4);
    out_chrom[i] = chrom;
    out_pos[i] = pos;
    out_base[i] = r[8];
    out_cb[i] = cb;
    out_p[i] = p;
}

}  // namespace

// chromosome numbers of staged calls: from the caller's provisional numbering (container order) to var_chrom's
__global__ __launch_bounds__(256) void k_remap_chrom(int *__restrict__ chrom, long long n, const int *__restrict__ table, int n_table,
                                                     int *__restrict__ bad)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int k = chrom[i];
    const int to = k >= 0 && k < n_table ? table[k] : -1;
    if (to < 0) atomicOr(bad, 1);
    chrom[i] = to;
}

void release_staged_calls(dmx_ctx *c)
{
    ctx_free(c, c->st_chrom);
    ctx_free(c, c->st_pos);
    ctx_free(c, c->st_base);
    ctx_free(c, c->st_cb);
    ctx_free(c, c->st_p);
    c->st_chrom = c->st_pos = c->st_cb = nullptr;
    c->st_base = nullptr;
    c->st_p = nullptr;
    c->n_staged = -1;
}

// Upload + field extraction of the containers' records (everything of the pack that does not need the variant keys):
// the flat call arrays stay with the context until pack_staged_on_device (or the next staging) takes them.
int stage_containers_on_device(dmx_ctx *c, const dmx_call_container *parts, int n_parts)
{
    hipStream_t st = c->stream;
    release_staged_calls(c);
    Scratch sc(c);
    long long n_calls = 0;
    for (int k = 0; k < n_parts; k++) n_calls += parts[k].n_snp_calls;
    const size_t n = (size_t)(n_calls ? n_calls : 1);
    DMX_TRY(ctx_malloc(c, (void **)&c->st_chrom, n * sizeof(int)));
    DMX_TRY(ctx_malloc(c, (void **)&c->st_pos, n * sizeof(int)));
    DMX_TRY(ctx_malloc(c, (void **)&c->st_base, n));
    DMX_TRY(ctx_malloc(c, (void **)&c->st_cb, n * sizeof(int)));
    DMX_TRY(ctx_malloc(c, (void **)&c->st_p, n * sizeof(float)));
    int *bad;
    DMX_TRY(sc.get(&bad, 1));
    HIP_TRY(hipMemsetAsync(bad, 0, sizeof(int), st));
    long long at = 0;
    for (int k = 0; k < n_parts; k++) {
        const dmx_call_container &part = parts[k];
        if (part.n_snp_calls == 0) continue;
        unsigned char *d_calls, *d_molecules;
        DMX_TRY(upload(sc, &d_calls, (const unsigned char *)part.snp_calls, (size_t)part.n_snp_calls * SNP_CALL_BYTES, st));
        DMX_TRY(upload(sc, &d_molecules, (const unsigned char *)part.molecules, (size_t)part.n_molecules * MOLECULE_BYTES, st));
        hipLaunchKernelGGL(k_flatten_container, dim3(grid_for(part.n_snp_calls)), dim3(256), 0, st, d_calls, part.n_snp_calls,
                           d_molecules, part.n_molecules, part.chrom, c->st_chrom + at, c->st_pos + at, c->st_base + at, c->st_cb + at,
                           c->st_p + at, bad);
        at += part.n_snp_calls;
    }
    HIP_TRY(hipGetLastError());
    int h_bad = 0;
    HIP_TRY(hipMemcpyAsync(&h_bad, bad, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));  // also: the caller's buffers are free to change from here on
    if (h_bad) {
        release_staged_calls(c);
        return fail(DMX_ERR_INVALID, "molecule_index outside the molecule table");
    }
    c->n_staged = n_calls;
    return 0;
}

// The rest of the pack on the staged calls.  chrom_table (nullable): var_chrom's number of the chromosome every staged
// call carries provisionally (-1: no variant on it - calls there are an error, demux.py:339-341, 359).
int pack_staged_on_device(dmx_ctx *c, long long V, const int *var_chrom, const int *var_pos, const unsigned char *var_base,
                          const int *chrom_table, int n_table, long long *n_matched, long long *n_unique, long long *mol_per_variant)
{
    if (c->n_staged < 0) return fail(DMX_ERR_INVALID, "call order: no staged call containers (dmx_stage_containers first)");
    hipStream_t st = c->stream;
    Scratch sc(c);
    const long long n_calls = c->n_staged;
    if (chrom_table && n_calls) {
        int *d_table, *bad;
        DMX_TRY(upload(sc, &d_table, chrom_table, (size_t)std::max(1, n_table), st));
        DMX_TRY(sc.get(&bad, 1));
        HIP_TRY(hipMemsetAsync(bad, 0, sizeof(int), st));
        hipLaunchKernelGGL(k_remap_chrom, dim3(grid_for(n_calls)), dim3(256), 0, st, c->st_chrom, n_calls, d_table, n_table, bad);
        int h_bad = 0;
        HIP_TRY(hipMemcpyAsync(&h_bad, bad, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (h_bad) {
            release_staged_calls(c);
            return fail(DMX_ERR_INVALID, "calls on a chromosome without variants");
        }
    }
    const int rc = pack_core(c, sc, V, var_chrom, var_pos, var_base, n_calls, c->st_chrom, c->st_pos, c->st_base, c->st_cb, c->st_p,
                             n_matched, n_unique, mol_per_variant);
    release_staged_calls(c);
    return rc;
}

int pack_containers_on_device(dmx_ctx *c, long long V, const int *var_chrom, const int *var_pos,
                              const unsigned char *var_base, const dmx_call_container *parts, int n_parts,
                              long long *n_matched, long long *n_unique, long long *mol_per_variant)
{
    DMX_TRY(stage_containers_on_device(c, parts, n_parts));
    return pack_staged_on_device(c, V, var_chrom, var_pos, var_base, nullptr, 0, n_matched, n_unique, mol_per_variant);
}

}  // namespace dmx
