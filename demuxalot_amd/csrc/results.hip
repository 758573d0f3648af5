// results.hip -- device-side reductions of the posterior matrix, so that the [B, K] result need not cross
// PCIe (4.3 GB per rank at 130k barcodes x 8256 options) when the caller only wants what users of the reference
// take from the DataFrame:
//   probs[probs.max(axis=1).gt(thr)].idxmax(axis=1)   examples/2-with-detection-of-new-SNPs.ipynb cell 14,
//                                                     demuxalot/snp_detection.py:166
//   probs[genotype_names].sum()                       same notebook, cells 19 / 21
//   the few best options of each barcode (singlet vs doublet calls)
#include <hip/hip_runtime.h>

#include "dmx_ctx.h"

namespace {

constexpr int TOP_MAX = 4;

// (value, column) ordering of the reductions: larger value first, lower column first among equals -- the first
// maximum, as DataFrame.idxmax / np.argmax return it; NaNs never win (as `v > best` is false for them).
__device__ __forceinline__ bool better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

// One wavefront per barcode: the TOP best options, best first.  Each lane keeps the TOP best of its own columns
// (sorted registers, unrolled insertion), then TOP rounds of a wave-wide arg-max pop the lanes' heads.
template <int TOP>
__global__ __launch_bounds__(256) void k_top_options(const float *__restrict__ post, long long B, int K, float threshold,
                                                     int *__restrict__ best, float *__restrict__ best_p,
                                                     unsigned long long *__restrict__ n_above)
{
    const int lane = threadIdx.x & 63;
    const long long b = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float *row = post + (size_t)b * K;
    float v[TOP];
    int ix[TOP];
#pragma unroll
    for (int t = 0; t < TOP; t++) {
        v[t] = -__builtin_inff();
        ix[t] = 0x7FFFFFFF;
    }
    for (int k = lane; k < K; k += 64) {
        float x = row[k];
        int xi = k;
        if (x != x) continue;  // NaN: never selected
#pragma unroll
        for (int t = 0; t < TOP; t++) {
            if (better(x, xi, v[t], ix[t])) {  // insert here, push the rest down
                const float tv = v[t];
                const int ti = ix[t];
                v[t] = x;
                ix[t] = xi;
                x = tv;
                xi = ti;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < TOP; r++) {
        float bv = v[0];
        int bi = ix[0];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float ov = __shfl_xor(bv, off);
            const int oi = __shfl_xor(bi, off);
            if (better(ov, oi, bv, bi)) {
                bv = ov;
                bi = oi;
            }
        }
        if (bi == ix[0] && bi != 0x7FFFFFFF) {  // the winning lane pops its head
#pragma unroll
            for (int t = 0; t + 1 < TOP; t++) {
                v[t] = v[t + 1];
                ix[t] = ix[t + 1];
            }
            v[TOP - 1] = -__builtin_inff();
            ix[TOP - 1] = 0x7FFFFFFF;
        }
        if (lane == 0) {
            const bool have = bi != 0x7FFFFFFF;
            if (TOP == 1) {
                // thresholded assignment: -1 unless the best posterior is strictly above the threshold (Series.gt)
                const bool ok = have && bv > threshold;
                best[b] = ok ? bi : -1;
                best_p[b] = have ? bv : __builtin_nanf("");
                if (ok && n_above) atomicAdd(n_above, 1ull);
            } else {
                best[(size_t)b * TOP + r] = have ? bi : -1;
                best_p[(size_t)b * TOP + r] = have ? bv : __builtin_nanf("");
            }
        }
    }
}

// Column sums over barcodes in two deterministic passes: float64 partial sums of row slabs, then the slabs in
// order.  (pandas adds float32 values row by row; the float64 sums here are at least as accurate.)
constexpr int SUM_SLABS = 512;
__global__ __launch_bounds__(256) void k_option_partial(const float *__restrict__ post, long long B, int K,
                                                        double *__restrict__ partial)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    const long long per = (B + SUM_SLABS - 1) / SUM_SLABS;
    const long long b0 = (long long)blockIdx.y * per;
    const long long b1 = b0 + per < B ? b0 + per : B;
    if (k >= K) return;
    double s = 0.0;
    for (long long b = b0; b < b1; b++) s += (double)post[(size_t)b * K + k];
    partial[(size_t)blockIdx.y * K + k] = s;
}

__global__ __launch_bounds__(256) void k_option_final(const double *__restrict__ partial, int K, double *__restrict__ sums)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    double s = 0.0;
    for (int j = 0; j < SUM_SLABS; j++) s += partial[(size_t)j * K + k];
    sums[k] = s;
}

int check_ready(dmx_ctx *c, const char *who)
{
    if (!c) return fail(DMX_ERR_INVALID, "null ctx");
    HIP_TRY(hipSetDevice(c->device));
    if (!c->have_post) return fail(DMX_ERR_INVALID, "call order: dmx_estep / dmx_em before %s", who);
    return 0;
}

template <typename T>
int scratch_alloc(T **p, size_t count)
{
    *p = nullptr;
    hipError_t e = hipMalloc((void **)p, (count ? count : 1) * sizeof(T));
    if (e != hipSuccess) return fail(DMX_ERR_HIP, "hipMalloc of %zu bytes failed: %s", count * sizeof(T), hipGetErrorString(e));
    return 0;
}

}  // namespace

extern "C" {

int dmx_get_assignments_above(dmx_ctx *c, float threshold, int32_t *best, float *best_p, int64_t *n_assigned)
{
    DMX_TRY(check_ready(c, "dmx_get_assignments_above"));
    unsigned long long *d_n = nullptr;
    DMX_TRY(scratch_alloc(&d_n, 1));
    int rc = 0;
    do {
        if (hipMemsetAsync(d_n, 0, sizeof(unsigned long long), c->stream) != hipSuccess) {
            rc = fail(DMX_ERR_HIP, "memset failed");
            break;
        }
        if (c->B > 0) {
            hipLaunchKernelGGL(k_top_options<1>, dim3((unsigned)((c->B + 3) / 4)), dim3(256), 0, c->stream, c->d_post, c->B, c->K,
                               threshold, c->d_best, c->d_bestp, d_n);
            if (hipGetLastError() != hipSuccess) {
                rc = fail(DMX_ERR_HIP, "assignment kernel launch failed");
                break;
            }
        }
        unsigned long long n = 0;
        hipError_t e = hipSuccess;
        if (best && c->B) e = hipMemcpyAsync(best, c->d_best, sizeof(int) * c->B, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess && best_p && c->B) e = hipMemcpyAsync(best_p, c->d_bestp, sizeof(float) * c->B, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&n, d_n, sizeof n, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) {
            rc = fail(DMX_ERR_HIP, "assignments: %s", hipGetErrorString(e));
            break;
        }
        if (n_assigned) *n_assigned = (int64_t)n;
    } while (false);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(d_n);
    return rc;
}

int dmx_get_top_options(dmx_ctx *c, int32_t k, int32_t *options, float *probs)
{
    DMX_TRY(check_ready(c, "dmx_get_top_options"));
    if (k < 1 || k > TOP_MAX) return fail(DMX_ERR_INVALID, "k must be 1..%d", TOP_MAX);
    if (c->B == 0) return 0;
    if (!options || !probs) return fail(DMX_ERR_INVALID, "null outputs");
    const size_t n = (size_t)c->B * k;
    int *d_i = nullptr;
    float *d_p = nullptr;
    DMX_TRY(scratch_alloc(&d_i, n));
    if (scratch_alloc(&d_p, n) != 0) {
        (void)hipFree(d_i);
        return DMX_ERR_HIP;
    }
    const dim3 grid((unsigned)((c->B + 3) / 4)), block(256);
    const float none = -__builtin_inff();
    switch (k) {
    case 1: hipLaunchKernelGGL(k_top_options<1>, grid, block, 0, c->stream, c->d_post, c->B, c->K, none, d_i, d_p, nullptr); break;
    case 2: hipLaunchKernelGGL(k_top_options<2>, grid, block, 0, c->stream, c->d_post, c->B, c->K, none, d_i, d_p, nullptr); break;
    case 3: hipLaunchKernelGGL(k_top_options<3>, grid, block, 0, c->stream, c->d_post, c->B, c->K, none, d_i, d_p, nullptr); break;
    default: hipLaunchKernelGGL(k_top_options<4>, grid, block, 0, c->stream, c->d_post, c->B, c->K, none, d_i, d_p, nullptr); break;
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(options, d_i, n * sizeof(int), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(probs, d_p, n * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(d_i);
    (void)hipFree(d_p);
    if (e != hipSuccess) return fail(DMX_ERR_HIP, "top options: %s", hipGetErrorString(e));
    return 0;
}

int dmx_get_option_sums(dmx_ctx *c, double *sums)
{
    DMX_TRY(check_ready(c, "dmx_get_option_sums"));
    if (!sums) return fail(DMX_ERR_INVALID, "null output");
    const int K = c->K;
    double *d_part = nullptr, *d_sums = nullptr;
    DMX_TRY(scratch_alloc(&d_part, (size_t)SUM_SLABS * K));
    if (scratch_alloc(&d_sums, (size_t)K) != 0) {
        (void)hipFree(d_part);
        return DMX_ERR_HIP;
    }
    const unsigned kb = (unsigned)((K + 255) / 256);
    hipLaunchKernelGGL(k_option_partial, dim3(kb, SUM_SLABS), dim3(256), 0, c->stream, c->d_post, c->B, K, d_part);
    hipLaunchKernelGGL(k_option_final, dim3(kb), dim3(256), 0, c->stream, d_part, K, d_sums);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(sums, d_sums, (size_t)K * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(d_part);
    (void)hipFree(d_sums);
    if (e != hipSuccess) return fail(DMX_ERR_HIP, "option sums: %s", hipGetErrorString(e));
    return 0;
}

}  // extern "C"
