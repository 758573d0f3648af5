// repack_device.hip -- GPU-side derivation of the kernels' layouts from the COO columns of the
// reference's `barcode_calls` (variant_id, compressed_cb, p_base_wrong; demuxalot/demux.py:290-300):
//
//   E-step records  barcode-major, calls of a barcode in input order, rows padded to 8 calls, two
//                   calls per 32-byte CallPair (kernels.h)
//   M-step records  variant-major, calls of a variant in input order, {compressed_cb, bits(1 - e)}
//   work items      runs of <= ITEM_CALLS calls of one variant; item_ptr per variant
//   work lists      barcodes / items by decreasing length
//
// "In input order" is what makes the float64 sums of the kernels run in np.bincount's order, so the
// two re-orderings are STABLE sorts (rocPRIM LSD radix sort of (key, input index) pairs; the one-time
// repack uses the library primitive, the per-iteration kernels are hand-written).  Counting is done
// with integer atomics (exact, order-free).  Everything stays on the ctx stream.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "dmx_ctx.h"

namespace dmx {
namespace {

__global__ __launch_bounds__(256) void k_count(const int *__restrict__ variant, const int *__restrict__ cb, long long N,
                                               long long B, long long V, unsigned *__restrict__ row_cnt,
                                               unsigned *__restrict__ col_cnt, int *__restrict__ bad)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int b = cb[i], v = variant[i];
    if (b < 0 || b >= B) {
        atomicMax(&bad[0], 1);
        atomicMin(&bad[2], (int)(i < 0x7fffffff ? i : 0x7fffffff));
        return;
    }
    if (v < 0 || v >= V) {
        atomicMax(&bad[1], 1);
        atomicMin(&bad[2], (int)(i < 0x7fffffff ? i : 0x7fffffff));
        return;
    }
    atomicAdd(&row_cnt[b], 1u);
    atomicAdd(&col_cnt[v], 1u);
}

// per-row padded pair counts, per-variant item counts, 64-bit copies of the raw counts
__global__ __launch_bounds__(256) void k_derive_counts(const unsigned *__restrict__ cnt, long long n, int mode,
                                                       long long *__restrict__ raw, long long *__restrict__ derived)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long c = cnt[i];
    raw[i] = c;
    derived[i] = mode == 0 ? ((c + 7) / 8) * 4            // CallPairs of a barcode row padded to 8 calls
                           : (c + ITEM_CALLS - 1) / ITEM_CALLS;  // work items of a variant
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
                                                     CallPair *__restrict__ pairs)
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
                                                     long long *__restrict__ item_start, int *__restrict__ item_len,
                                                     unsigned *__restrict__ inv_len, unsigned *__restrict__ ids)
{
    const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    long long it = item_ptr[v];
    for (long long s = col_ptr[v]; s < col_ptr[v + 1]; s += ITEM_CALLS, it++) {
        const long long len = (col_ptr[v + 1] - s) < ITEM_CALLS ? (col_ptr[v + 1] - s) : ITEM_CALLS;
        item_start[it] = s;
        item_len[it] = (int)len;
        inv_len[it] = ~(unsigned)len;  // ascending sort of ~len = longest first
        ids[it] = (unsigned)it;
    }
}

__global__ __launch_bounds__(256) void k_inv_counts(const unsigned *__restrict__ cnt, long long n,
                                                    unsigned *__restrict__ inv, unsigned *__restrict__ ids)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    inv[i] = ~cnt[i];
    ids[i] = (unsigned)i;
}

inline unsigned grid_for(long long n) { return (unsigned)((n + 255) / 256); }

inline unsigned bits_for(unsigned long long max_value)
{
    unsigned b = 1;
    while (b < 64 && (max_value >> b) != 0) b++;
    return b;
}

// scratch owner: frees everything on scope exit
struct Scratch {
    std::vector<void *> ptrs;
    ~Scratch()
    {
        for (void *p : ptrs) (void)hipFree(p);
    }
    template <typename T>
    int get(T **out, size_t count)
    {
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, (count ? count : 1) * sizeof(T));
        if (e != hipSuccess) return fail(DMX_ERR_HIP, "hipMalloc(scratch %zu bytes): %s", count * sizeof(T), hipGetErrorString(e));
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

int repack_on_device(dmx_ctx *c, const int32_t *h_variant, const int32_t *h_cb, const float *h_p)
{
    const long long B = c->B, V = c->V, N = c->N;
    const int G = c->G;
    hipStream_t st = c->stream;
    Scratch sc;

    // ---- upload the COO columns ----
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

    // ---- counts (integer atomics) + range check ----
    unsigned *row_cnt = nullptr, *col_cnt = nullptr;
    int *bad = nullptr;
    DMX_TRY(sc.get(&row_cnt, (size_t)B));
    DMX_TRY(sc.get(&col_cnt, (size_t)V));
    DMX_TRY(sc.get(&bad, 3));
    HIP_TRY(hipMemsetAsync(row_cnt, 0, sizeof(unsigned) * (B ? B : 1), st));
    HIP_TRY(hipMemsetAsync(col_cnt, 0, sizeof(unsigned) * (V ? V : 1), st));
    const int bad_init[3] = {0, 0, 0x7fffffff};
    HIP_TRY(hipMemcpyAsync(bad, bad_init, sizeof bad_init, hipMemcpyHostToDevice, st));
    if (N) hipLaunchKernelGGL(k_count, dim3(grid_for(N)), dim3(256), 0, st, d_variant, d_cb, N, B, V, row_cnt, col_cnt, bad);
    int h_bad[3];
    HIP_TRY(hipMemcpyAsync(h_bad, bad, sizeof h_bad, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (h_bad[0]) return fail(DMX_ERR_INVALID, "compressed_cb[%d]=%d outside [0,%lld)", h_bad[2], h_cb[h_bad[2]], B);
    if (h_bad[1]) return fail(DMX_ERR_INVALID, "variant_id[%d]=%d outside [0,%lld)", h_bad[2], h_variant[h_bad[2]], V);

    // ---- prefix sums: raw row / column starts, padded pair offsets, items per variant ----
    long long *row_raw = nullptr, *row_pairs = nullptr, *col_raw = nullptr, *col_items = nullptr, *row_start = nullptr,
              *col_ptr = nullptr;
    DMX_TRY(sc.get(&row_raw, (size_t)B));
    DMX_TRY(sc.get(&row_pairs, (size_t)B));
    DMX_TRY(sc.get(&col_raw, (size_t)V));
    DMX_TRY(sc.get(&col_items, (size_t)V));
    DMX_TRY(sc.get(&row_start, (size_t)B + 1));
    DMX_TRY(sc.get(&col_ptr, (size_t)V + 1));
    if (B) hipLaunchKernelGGL(k_derive_counts, dim3(grid_for(B)), dim3(256), 0, st, row_cnt, B, 0, row_raw, row_pairs);
    if (V) hipLaunchKernelGGL(k_derive_counts, dim3(grid_for(V)), dim3(256), 0, st, col_cnt, V, 1, col_raw, col_items);
    DMX_TRY(dev_alloc(c, &c->d_pair_ptr, (size_t)B + 1));
    DMX_TRY(dev_alloc(c, &c->d_item_ptr, (size_t)V + 1));
    DMX_TRY(scan_with_total(sc, row_raw, row_start, (size_t)B, st));
    DMX_TRY(scan_with_total(sc, row_pairs, c->d_pair_ptr, (size_t)B, st));
    DMX_TRY(scan_with_total(sc, col_raw, col_ptr, (size_t)V, st));
    DMX_TRY(scan_with_total(sc, col_items, c->d_item_ptr, (size_t)V, st));
    long long n_pairs = 0, n_items = 0;
    HIP_TRY(hipMemcpyAsync(&n_pairs, c->d_pair_ptr + B, sizeof(long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&n_items, c->d_item_ptr + V, sizeof(long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (n_items >= (1LL << 31)) return fail(DMX_ERR_UNSUPPORTED, "too many M-step work items");
    c->n_pairs = n_pairs;
    c->n_items = n_items;

    // ---- stable sorts of (key, input index) ----
    unsigned *iota = nullptr, *keys_sorted = nullptr, *perm = nullptr;
    DMX_TRY(sc.get(&iota, (size_t)N));
    DMX_TRY(sc.get(&keys_sorted, (size_t)N));
    DMX_TRY(sc.get(&perm, (size_t)N));
    if (N) hipLaunchKernelGGL(k_iota, dim3(grid_for(N)), dim3(256), 0, st, iota, N);

    // barcode-major -> E-step records
    DMX_TRY(dev_alloc(c, &c->d_call_pairs, (size_t)n_pairs));
    if (n_pairs) hipLaunchKernelGGL(k_fill_neutral, dim3(grid_for(n_pairs)), dim3(256), 0, st, c->d_call_pairs, n_pairs);
    DMX_TRY(sort_pairs(sc, (const unsigned *)d_cb, keys_sorted, iota, perm, (size_t)N, bits_for(B ? B - 1 : 0), st));
    if (N)
        hipLaunchKernelGGL(k_build_pairs, dim3(grid_for(N)), dim3(256), 0, st, keys_sorted, perm, d_variant, d_p, row_start,
                           c->d_pair_ptr, N, (unsigned)G, c->d_call_pairs);

    // variant-major -> M-step records
    DMX_TRY(dev_alloc(c, &c->d_csc, (size_t)N));
    DMX_TRY(sort_pairs(sc, (const unsigned *)d_variant, keys_sorted, iota, perm, (size_t)N, bits_for(V ? V - 1 : 0), st));
    if (N) hipLaunchKernelGGL(k_build_csc, dim3(grid_for(N)), dim3(256), 0, st, perm, d_cb, d_p, N, c->d_csc);

    // ---- work items and length-sorted work lists ----
    DMX_TRY(dev_alloc(c, &c->d_item_start, (size_t)n_items));
    DMX_TRY(dev_alloc(c, &c->d_item_len, (size_t)n_items));
    DMX_TRY(dev_alloc(c, &c->d_item_order, (size_t)n_items));
    DMX_TRY(dev_alloc(c, &c->d_bc_order, (size_t)B));
    unsigned *inv = nullptr, *ids = nullptr, *inv_sorted = nullptr;
    const size_t m = (size_t)(n_items > B ? n_items : B);
    DMX_TRY(sc.get(&inv, m));
    DMX_TRY(sc.get(&ids, m));
    DMX_TRY(sc.get(&inv_sorted, m));
    if (V)
        hipLaunchKernelGGL(k_build_items, dim3(grid_for(V)), dim3(256), 0, st, col_ptr, c->d_item_ptr, V, c->d_item_start,
                           c->d_item_len, inv, ids);
    // ~len has its variable bits only in the low bits_for(ITEM_CALLS) positions; the high bits are all ones
    DMX_TRY(sort_pairs(sc, inv, inv_sorted, ids, (unsigned *)c->d_item_order, (size_t)n_items, 32, st));
    if (B) hipLaunchKernelGGL(k_inv_counts, dim3(grid_for(B)), dim3(256), 0, st, row_cnt, B, inv, ids);
    DMX_TRY(sort_pairs(sc, inv, inv_sorted, ids, (unsigned *)c->d_bc_order, (size_t)B, 32, st));

    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));  // host source buffers and scratch are released on return
    return 0;
}

}  // namespace dmx
