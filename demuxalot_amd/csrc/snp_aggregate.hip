// snp_aggregate.hip -- the alternative E-step of the reference, Demultiplexer.aggregate_on_snps = True
// (demuxalot/demux.py:204-244), and the float64 M-step that goes with it.
//
// The reference regularises the likelihood per (barcode, SNP) pair ("bns"):
//   S[bns, k]  = sum over the MOLECULE calls m of the pair, in molecule_calls order, of  log(p_k[v_m] + e_m)
//                (float32 add, numpy float32 log, float64 accumulation, stored float32)          demux.py:226-228
//   x          = float32( float64(S) / count[bns] ** compensation )                              demux.py:232
//   y          = scipy log_softmax(x) over the options, float32                                  demux.py:234
//   z          = np.logaddexp(y, log(0.01 / K))            -> float64 from here on               demux.py:235-236
//   w          = scipy log_softmax(z), float64                                                   demux.py:237
//   logit[b,k] = sum of w over the barcode's pairs in (SNP-sorted) pair order, float64           demux.py:239-242
// and the posteriors are scipy's softmax of the float64 logits (no doublet penalties are added in this mode:
// the reference computes them and never uses them).  The M-step then runs on float64 posteriors:
//   add[v,g] = float32( sum_c ( post64[cb_c, g] * float64(1 - e_c) ) ** power )                  demux.py:113-118
//
// Everything up to y repeats numpy's float32 arithmetic exactly as the default E-step does (np_math.h).  z and w
// use float64 exp / log / log1p, for which numpy calls its own SIMD kernels or libm depending on the host CPU;
// the device uses the ROCm device library's: results agree with the reference to a few float64 ulps, not bit for bit.
//
// Layout: the matched molecule calls, stably sorted by (barcode, SNP) -- i.e. grouped by pair, pairs in the
// reference's FeatureLookup order (utils.py:207-262), molecule order inside a pair -- with a head flag on the first
// call of every pair.  One wavefront walks one barcode.
#include <cmath>
#include <cstring>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "dmx_ctx.h"
#include "np_math.h"

namespace {

using dmx::fail;

constexpr int SNP_MAX_OPTIONS = 256 * 33;  // doublets of 128 genotypes: 8256 options, about 100 KB of LDS per barcode

inline unsigned grid_for(long long n) { return (unsigned)((n + 255) / 256); }

// ---- layout -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pair_keys(const unsigned long long *__restrict__ vb_keys, const int *__restrict__ v2snp,
                                                   long long m, unsigned long long *__restrict__ keys, unsigned *__restrict__ idx)
{
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const unsigned long long k = vb_keys[j];  // variant << 32 | barcode
    keys[j] = ((k & 0xFFFFFFFFull) << 32) | (unsigned)v2snp[k >> 32];
    idx[j] = (unsigned)j;
}

__global__ __launch_bounds__(256) void k_pair_fill(const unsigned long long *__restrict__ keys_sorted, const unsigned *__restrict__ perm,
                                                   const unsigned long long *__restrict__ vb_keys, const unsigned *__restrict__ src_idx,
                                                   const float *__restrict__ src_p, long long m, int *__restrict__ variant,
                                                   float *__restrict__ e)
{
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= m) return;
    const unsigned j = perm[s];
    const bool head = s == 0 || keys_sorted[s] != keys_sorted[s - 1];
    variant[s] = (int)(vb_keys[j] >> 32) | (head ? (int)0x80000000 : 0);  // top bit: first call of a pair
    e[s] = src_p[src_idx ? src_idx[j] : j];
}

// first sorted call of every barcode (lower bound on the barcode half of the key), b = 0..B
__global__ __launch_bounds__(256) void k_barcode_starts(const unsigned long long *__restrict__ keys_sorted, long long m, long long B,
                                                        long long *__restrict__ start)
{
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b > B) return;
    long long lo = 0, hi = m;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if ((long long)(keys_sorted[mid] >> 32) < b) lo = mid + 1; else hi = mid;
    }
    start[b] = lo;
}

__global__ __launch_bounds__(256) void k_max_run(const int *__restrict__ variant, long long m, unsigned *__restrict__ longest)
{
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= m || variant[s] >= 0) return;  // heads only
    unsigned n = 1;
    for (long long t = s + 1; t < m && variant[t] >= 0; t++) n++;
    atomicMax(longest, n);
}

// ---- float64 helpers ----------------------------------------------------------------------------------------
__device__ __forceinline__ double logaddexp_f64(double x, double y)  // numpy's npy_logaddexp
{
    if (x == y) return x + 0.693147180559945309417232121458176568;
    const double t = x - y;
    if (t > 0) return x + log1p(exp(-t));
    if (t <= 0) return y + log1p(exp(t));
    return t;  // NaN
}

// np.sum of n float64 values in LDS, numpy's pairwise association (np_math.h, float64 flavour), by one wavefront
__device__ __forceinline__ double leaf_sum64(const double *a, int n, int lane)
{
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    }
    const int j = lane & 7;
    const int nfull = n - (n & 7);
    double r = a[j];
    for (int i = 8 + j; i < nfull; i += 8) r += a[i];
    for (int off = 1; off < 8; off <<= 1) {
        const int lo = __shfl_xor(__double2loint(r), off), hi = __shfl_xor(__double2hiint(r), off);
        r = r + __hiloint2double(hi, lo);
    }
    for (int i = nfull; i < n; i++) r += a[i];
    return r;
}

__device__ __forceinline__ double row_sum64(const double *a, int n, int lane)
{
    if (n <= 128) return leaf_sum64(a, n, lane);
    int st_start[16], st_len[16];
    unsigned char st_state[16];
    double st_left[16];
    int sp = 0;
    st_start[0] = 0;
    st_len[0] = n;
    st_state[0] = 0;
    double ret = 0.0;
    while (sp >= 0) {  // post-order walk of numpy's split tree (n <= 8192: one pairwise chunk)
        const int s = st_start[sp], len = st_len[sp];
        if (len <= 128) {
            ret = leaf_sum64(a + s, len, lane);
            sp--;
            continue;
        }
        int half = len / 2;
        half -= half % 8;
        if (st_state[sp] == 0) {
            st_state[sp] = 1;
            sp++;
            st_start[sp] = s;
            st_len[sp] = half;
            st_state[sp] = 0;
        } else if (st_state[sp] == 1) {
            st_left[sp] = ret;
            st_state[sp] = 2;
            sp++;
            st_start[sp] = s + half;
            st_len[sp] = len - half;
            st_state[sp] = 0;
        } else {
            ret = st_left[sp] + ret;
            sp--;
        }
    }
    return ret;
}

__device__ __forceinline__ double wave_max64(double v)
{
    for (int off = 1; off < 64; off <<= 1) {
        const int lo = __shfl_xor(__double2loint(v), off), hi = __shfl_xor(__double2hiint(v), off);
        v = fmax(v, __hiloint2double(hi, lo));
    }
    return v;
}

// ---- E-step -------------------------------------------------------------------------------------------------
struct SnpArgs {
    const long long *bc_start;  // [B+1] first sorted molecule call of every barcode
    const int *variant;         // [m] variant row, top bit = first call of a (barcode, SNP) pair
    const float *e;             // [m] p_base_wrong of the molecule call
    const float *prob;          // genotype_prob, padded row layout
    const int *prow;            // nullable: padded row of every variant
    const unsigned *opt_pairs;  // [K] g1 | g2 << 16
    const int *sum_plan;        // np.sum over K values as a workgroup plan (np_math.h: plan_sum_block); block kernel only
    int sum_plan_values;
    const double *count_pow;    // [max_count + 1] count ** compensation as numpy computes it
    const void *prior;          // nullable [B, K] prior logits
    int prior_dtype;
    double *logits, *post;      // [B, K] float64
    double log_bad;             // np.log(0.01 / K)
    long long B;
    int G, K;
};

// A = options per lane (K <= 64 A).  LDS: K float32 + K float64 per wavefront (numpy-ordered sums).
template <int A, bool PAIRS>
__global__ __launch_bounds__(256) void k_estep_snp(SnpArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int K = a.K, G = a.G;
    double *sh64 = (double *)smem + (size_t)wave * K;
    float *sh32 = (float *)((double *)smem + (size_t)4 * K) + (size_t)wave * K;
    const long long b = (long long)blockIdx.x * 4 + wave;
    if (b >= a.B) return;

    unsigned g1[A], g2[A];
    bool valid[A];
#pragma unroll
    for (int s = 0; s < A; s++) {
        const int k = lane + 64 * s;
        valid[s] = k < K;
        const unsigned pr = a.opt_pairs[valid[s] ? k : K - 1];
        g1[s] = pr & 0xFFFFu;
        g2[s] = pr >> 16;
    }
    double logit[A], acc[A];
#pragma unroll
    for (int s = 0; s < A; s++) logit[s] = acc[s] = 0.0;
    int count = 0;

    // one (barcode, SNP) pair is complete: regularise it and add it to the barcode's logits
    auto finish = [&]() {
        const double div = a.count_pow[count];
        float x[A];
        float mx = -__builtin_inff();
#pragma unroll
        for (int s = 0; s < A; s++) {
            const float s32 = (float)acc[s];             // utils.py:35-36: float32(0 + float64 sum)
            x[s] = (float)((double)s32 / div);           // demux.py:232 (float32 /= float64)
            if (valid[s]) mx = fmaxf(mx, x[s]);
        }
        for (int off = 1; off < 64; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        if (!(fabsf(mx) < __builtin_inff())) mx = 0.0f;  // scipy: x_max[~isfinite(x_max)] = 0
        float tmp[A];
#pragma unroll
        for (int s = 0; s < A; s++) {
            tmp[s] = x[s] - mx;
            if (valid[s]) sh32[lane + 64 * s] = npm::exp_f32(tmp[s]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const float tot = npm::row_sum_wave(sh32, K, lane);
        const float lse = npm::log_f32<true, false>(__shfl(tot, 0));
        double z[A];
        double mx2 = -__builtin_inf();
#pragma unroll
        for (int s = 0; s < A; s++) {
            const float y = tmp[s] - lse;                // float32 log_softmax
            z[s] = logaddexp_f64((double)y, a.log_bad);  // demux.py:236
            if (valid[s]) mx2 = fmax(mx2, z[s]);
        }
        mx2 = wave_max64(mx2);
        if (!(fabs(mx2) < __builtin_inf())) mx2 = 0.0;
        double t2[A];
#pragma unroll
        for (int s = 0; s < A; s++) {
            t2[s] = z[s] - mx2;
            if (valid[s]) sh64[lane + 64 * s] = exp(t2[s]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        double tot2 = row_sum64(sh64, K, lane);
        tot2 = __hiloint2double(__shfl(__double2hiint(tot2), 0), __shfl(__double2loint(tot2), 0));
        const double lse2 = log(tot2);
#pragma unroll
        for (int s = 0; s < A; s++) {
            logit[s] += t2[s] - lse2;                    // np.bincount over the barcode's pairs, in pair order
            acc[s] = 0.0;
        }
        count = 0;
        __builtin_amdgcn_wave_barrier();
    };

    const long long c0 = a.bc_start[b], c1 = a.bc_start[b + 1];
    for (long long c = c0; c < c1; c++) {
        const int tagged = a.variant[c];
        if (tagged < 0 && c > c0) finish();
        const int v = tagged & 0x7FFFFFFF;
        const float e = a.e[c];
        const float *row = a.prob + (size_t)(a.prow ? a.prow[v] : v) * G;
#pragma unroll
        for (int s = 0; s < A; s++) {
            float p = row[g1[s]];
            if (PAIRS) p = (p + row[g2[s]]) * 0.5f;
            acc[s] += (double)npm::log_f32<true, false>(p + e);  // demux.py:227: np.log(p + p_base_wrong)
        }
        count++;
    }
    if (c1 > c0) finish();

    // logits (+ prior at the first iteration), scipy softmax in float64
    double mx = -__builtin_inf();
#pragma unroll
    for (int s = 0; s < A; s++) {
        const int k = lane + 64 * s;
        if (a.prior && valid[s]) {
            const size_t o = (size_t)b * K + k;
            logit[s] += a.prior_dtype == DMX_F32 ? (double)((const float *)a.prior)[o] : ((const double *)a.prior)[o];
        }
        if (valid[s]) mx = fmax(mx, logit[s]);
    }
    mx = wave_max64(mx);
    double ex[A];
#pragma unroll
    for (int s = 0; s < A; s++) {
        ex[s] = exp(logit[s] - mx);
        if (valid[s]) sh64[lane + 64 * s] = ex[s];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    double tot = row_sum64(sh64, K, lane);
    tot = __hiloint2double(__shfl(__double2hiint(tot), 0), __shfl(__double2loint(tot), 0));
#pragma unroll
    for (int s = 0; s < A; s++) {
        const int k = lane + 64 * s;
        if (!valid[s]) continue;
        const size_t o = (size_t)b * K + k;
        a.logits[o] = logit[s];
        a.post[o] = ex[s] / tot;
    }
}

// More than 1024 options (doublets of 45+ genotypes): one WORKGROUP walks one barcode, thread t holds options
// t + 256 s.  Same arithmetic as k_estep_snp; the row maxima go through LDS, the numpy-ordered sums are done by
// wavefront 0 over the LDS row.  LDS: K float64 + K float32.
template <int A>
__global__ __launch_bounds__(256) void k_estep_snp_block(SnpArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ double red64[5];
    __shared__ float red32[5];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.K, G = a.G;
    double *sh64 = (double *)smem;
    double *val64 = sh64 + K;                       // scratch of the sum plan
    float *sh32 = (float *)(val64 + a.sum_plan_values);
    float *val32 = sh32 + K;
    const long long b = blockIdx.x;

    unsigned pr[A];
    bool valid[A];
#pragma unroll
    for (int s = 0; s < A; s++) {
        const int k = tid + 256 * s;
        valid[s] = k < K;
        pr[s] = a.opt_pairs[valid[s] ? k : K - 1];
    }
    double logit[A], acc[A];
#pragma unroll
    for (int s = 0; s < A; s++) logit[s] = acc[s] = 0.0;
    int count = 0;

    auto block_max32 = [&](float v) {
        for (int off = 1; off < 64; off <<= 1) v = fmaxf(v, __shfl_xor(v, off));
        if (lane == 0) red32[wave] = v;
        __syncthreads();
        return fmaxf(fmaxf(red32[0], red32[1]), fmaxf(red32[2], red32[3]));
    };
    auto block_max64 = [&](double v) {
        v = wave_max64(v);
        if (lane == 0) red64[wave] = v;
        __syncthreads();
        return fmax(fmax(red64[0], red64[1]), fmax(red64[2], red64[3]));
    };
    auto block_sum32 = [&]() {  // np.sum of sh32[0..K)
        __syncthreads();
        return npm::plan_sum_block<float>(sh32, a.sum_plan, val32, tid);
    };
    auto block_sum64 = [&]() {
        __syncthreads();
        return npm::plan_sum_block<double>(sh64, a.sum_plan, val64, tid);
    };

    auto finish = [&]() {
        const double div = a.count_pow[count];
        float tmp[A];
        float mx = -__builtin_inff();
#pragma unroll
        for (int s = 0; s < A; s++) {
            const float s32 = (float)acc[s];
            tmp[s] = (float)((double)s32 / div);
            if (valid[s]) mx = fmaxf(mx, tmp[s]);
        }
        mx = block_max32(mx);
        if (!(fabsf(mx) < __builtin_inff())) mx = 0.0f;
#pragma unroll
        for (int s = 0; s < A; s++) {
            tmp[s] = tmp[s] - mx;
            if (valid[s]) sh32[tid + 256 * s] = npm::exp_f32(tmp[s]);
        }
        const float lse = npm::log_f32<true, false>(block_sum32());
        double t2[A];
        double mx2 = -__builtin_inf();
#pragma unroll
        for (int s = 0; s < A; s++) {
            const float y = tmp[s] - lse;
            t2[s] = logaddexp_f64((double)y, a.log_bad);
            if (valid[s]) mx2 = fmax(mx2, t2[s]);
        }
        mx2 = block_max64(mx2);
        if (!(fabs(mx2) < __builtin_inf())) mx2 = 0.0;
#pragma unroll
        for (int s = 0; s < A; s++) {
            t2[s] = t2[s] - mx2;
            if (valid[s]) sh64[tid + 256 * s] = exp(t2[s]);
        }
        const double lse2 = log(block_sum64());
#pragma unroll
        for (int s = 0; s < A; s++) {
            logit[s] += t2[s] - lse2;
            acc[s] = 0.0;
        }
        count = 0;
    };

    const long long c0 = a.bc_start[b], c1 = a.bc_start[b + 1];
    for (long long c = c0; c < c1; c++) {
        const int tagged = a.variant[c];
        if (tagged < 0 && c > c0) finish();
        const int v = tagged & 0x7FFFFFFF;
        const float e = a.e[c];
        const float *row = a.prob + (size_t)(a.prow ? a.prow[v] : v) * G;
#pragma unroll
        for (int s = 0; s < A; s++) {
            const float p = (row[pr[s] & 0xFFFFu] + row[pr[s] >> 16]) * 0.5f;  // g1 == g2 for the singlets: (p + p) / 2 = p
            acc[s] += (double)npm::log_f32<true, false>(p + e);
        }
        count++;
    }
    if (c1 > c0) finish();

    double mx = -__builtin_inf();
#pragma unroll
    for (int s = 0; s < A; s++) {
        const int k = tid + 256 * s;
        if (a.prior && valid[s]) {
            const size_t o = (size_t)b * K + k;
            logit[s] += a.prior_dtype == DMX_F32 ? (double)((const float *)a.prior)[o] : ((const double *)a.prior)[o];
        }
        if (valid[s]) mx = fmax(mx, logit[s]);
    }
    mx = block_max64(mx);
    double ex[A];
#pragma unroll
    for (int s = 0; s < A; s++) {
        ex[s] = exp(logit[s] - mx);
        if (valid[s]) sh64[tid + 256 * s] = ex[s];
    }
    const double tot = block_sum64();
#pragma unroll
    for (int s = 0; s < A; s++) {
        const int k = tid + 256 * s;
        if (!valid[s]) continue;
        const size_t o = (size_t)b * K + k;
        a.logits[o] = logit[s];
        a.post[o] = ex[s] / tot;
    }
}

// ---- M-step on float64 posteriors -----------------------------------------------------------------------------
// one wavefront per variant walks all its calls in order (items in order = CSC order), lane g (+64 s) = genotype
template <int A, bool SQUARE>
__global__ __launch_bounds__(256) void k_mstep_f64(const long long *__restrict__ item_ptr, const long long *__restrict__ item_start,
                                                   const int *__restrict__ item_len, const uint2 *__restrict__ calls,
                                                   const double *__restrict__ post, long long V, int G, long long K, double power,
                                                   float *__restrict__ add, double *__restrict__ sums)
{
    const int lane = threadIdx.x & 63;
    const long long v = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= V) return;
    double acc[A];
#pragma unroll
    for (int s = 0; s < A; s++) acc[s] = 0.0;
    for (long long it = item_ptr[v]; it < item_ptr[v + 1]; it++) {
        const long long first = item_start[it];
        const int n = item_len[it];
        for (int i = 0; i < n; i++) {
            const uint2 d = calls[first + i];
            const double keep = (double)__uint_as_float(d.y);  // float32 (1 - e), promoted as numpy does
#pragma unroll
            for (int s = 0; s < A; s++) {
                const int g = lane + 64 * s;
                if (g >= G) continue;
                double cterm = post[(size_t)d.x * K + g] * keep;
                cterm = SQUARE ? cterm * cterm : pow(cterm, power);
                acc[s] += cterm;
            }
        }
    }
#pragma unroll
    for (int s = 0; s < A; s++) {
        const int g = lane + 64 * s;
        if (g < G) {
            add[v * G + g] = (float)acc[s];
            if (sums) sums[v * G + g] = acc[s];  // the unrounded float64 sum (barcode-sharded runs add these over ranks)
        }
    }
}

template <int A>
int launch_snp(dmx_ctx *c, const SnpArgs &a, bool pairs)
{
    const size_t bytes = (size_t)4 * a.K * (sizeof(double) + sizeof(float));
    const dim3 grid((unsigned)((a.B + 3) / 4)), block(256);
    if (a.B == 0) return 0;
    if (pairs) {
        HIP_TRY(hipFuncSetAttribute((const void *)k_estep_snp<A, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        hipLaunchKernelGGL((k_estep_snp<A, true>), grid, block, bytes, c->stream, a);
    } else {
        HIP_TRY(hipFuncSetAttribute((const void *)k_estep_snp<A, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        hipLaunchKernelGGL((k_estep_snp<A, false>), grid, block, bytes, c->stream, a);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int A>
int launch_snp_block(dmx_ctx *c, const SnpArgs &a)
{
    const size_t bytes = ((size_t)a.K + (size_t)a.sum_plan_values) * (sizeof(double) + sizeof(float));
    HIP_TRY(hipFuncSetAttribute((const void *)k_estep_snp_block<A>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    if (a.B) hipLaunchKernelGGL((k_estep_snp_block<A>), dim3((unsigned)a.B), dim3(256), bytes, c->stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int A>
void launch_m64(dmx_ctx *c, double power, double *sums)
{
    const dim3 grid((unsigned)((c->V + 3) / 4)), block(256);
    if (power == 2.0)
        hipLaunchKernelGGL((k_mstep_f64<A, true>), grid, block, 0, c->stream, c->d_item_ptr, c->d_item_start, c->d_item_len, c->d_csc,
                           c->d_post64, c->V, c->G, (long long)c->K, power, c->d_add, sums);
    else
        hipLaunchKernelGGL((k_mstep_f64<A, false>), grid, block, 0, c->stream, c->d_item_ptr, c->d_item_start, c->d_item_len, c->d_csc,
                           c->d_post64, c->V, c->G, (long long)c->K, power, c->d_add, sums);
}

}  // namespace

namespace dmx {

// matched molecule calls -> (barcode, SNP)-grouped layout.  vb_keys[j] = variant << 32 | barcode of call j (molecule
// order); p of call j = src_p[src_idx ? src_idx[j] : j].
int build_snp_groups(dmx_ctx *c, const unsigned long long *vb_keys, const unsigned *src_idx, const float *src_p, long long m)
{
    hipStream_t st = c->stream;
    dev_free(c, &c->d_mc_variant, (size_t)c->n_mc);
    dev_free(c, &c->d_mc_e, (size_t)c->n_mc);
    dev_free(c, &c->d_mc_start, (size_t)c->B + 1);
    c->n_mc = m;
    c->mc_max_count = 0;
    DMX_TRY(dev_alloc(c, &c->d_mc_variant, (size_t)m));
    DMX_TRY(dev_alloc(c, &c->d_mc_e, (size_t)m));
    DMX_TRY(dev_alloc(c, &c->d_mc_start, (size_t)c->B + 1));
    int *d_v2snp = nullptr;
    unsigned long long *keys = nullptr, *keys_sorted = nullptr;
    unsigned *idx = nullptr, *perm = nullptr, *longest = nullptr;
    char *tmp = nullptr;
    int rc = 0;
    auto alloc = [&](void **p, size_t bytes) {
        if (rc == 0 && hipMalloc(p, bytes ? bytes : 1) != hipSuccess) rc = fail(DMX_ERR_HIP, "hipMalloc of %zu bytes failed", bytes);
    };
    alloc((void **)&d_v2snp, sizeof(int) * (size_t)c->V);
    alloc((void **)&keys, sizeof(unsigned long long) * (size_t)m);
    alloc((void **)&keys_sorted, sizeof(unsigned long long) * (size_t)m);
    alloc((void **)&idx, sizeof(unsigned) * (size_t)m);
    alloc((void **)&perm, sizeof(unsigned) * (size_t)m);
    alloc((void **)&longest, sizeof(unsigned));
    do {
        if (rc) break;
        hipError_t e = hipSuccess;
        if (c->V) e = hipMemcpyAsync(d_v2snp, c->h_v2snp.data(), sizeof(int) * (size_t)c->V, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemsetAsync(longest, 0, sizeof(unsigned), st);
        if (e != hipSuccess) { rc = fail(DMX_ERR_HIP, "snp groups: %s", hipGetErrorString(e)); break; }
        if (m) {
            hipLaunchKernelGGL(k_pair_keys, dim3(grid_for(m)), dim3(256), 0, st, vb_keys, d_v2snp, m, keys, idx);
            size_t bytes = 0;
            e = rocprim::radix_sort_pairs(nullptr, bytes, keys, keys_sorted, idx, perm, (size_t)m, 0u, 64u, st);
            if (e == hipSuccess) alloc((void **)&tmp, bytes);
            if (rc) break;
            if (e == hipSuccess) e = rocprim::radix_sort_pairs(tmp, bytes, keys, keys_sorted, idx, perm, (size_t)m, 0u, 64u, st);
            if (e != hipSuccess) { rc = fail(DMX_ERR_HIP, "snp groups sort: %s", hipGetErrorString(e)); break; }
            hipLaunchKernelGGL(k_pair_fill, dim3(grid_for(m)), dim3(256), 0, st, keys_sorted, perm, vb_keys, src_idx, src_p, m,
                               c->d_mc_variant, c->d_mc_e);
            hipLaunchKernelGGL(k_max_run, dim3(grid_for(m)), dim3(256), 0, st, c->d_mc_variant, m, longest);
        }
        hipLaunchKernelGGL(k_barcode_starts, dim3(grid_for(c->B + 1)), dim3(256), 0, st, keys_sorted, m, c->B, c->d_mc_start);
        unsigned h_longest = 0;
        e = hipMemcpyAsync(&h_longest, longest, sizeof(unsigned), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) { rc = fail(DMX_ERR_HIP, "snp groups: %s", hipGetErrorString(e)); break; }
        c->mc_max_count = h_longest;
    } while (false);
    (void)hipStreamSynchronize(st);
    for (void *p : {(void *)d_v2snp, (void *)keys, (void *)keys_sorted, (void *)idx, (void *)perm, (void *)longest, (void *)tmp})
        if (p) (void)hipFree(p);
    return rc;
}

}  // namespace dmx

extern "C" {

int dmx_set_keep_molecule_calls(dmx_ctx *c, int keep)
{
    if (!c) return fail(DMX_ERR_INVALID, "null context");
    c->keep_molecule_calls = keep != 0;
    return 0;
}

int dmx_set_molecule_calls(dmx_ctx *c, int64_t n, const int32_t *variant_id, const int32_t *compressed_cb, const float *p_base_wrong)
{
    if (!c) return fail(DMX_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    if (!c->have_problem) return fail(DMX_ERR_INVALID, "call order: a resident problem before dmx_set_molecule_calls");
    if (n < 0 || n >= (1LL << 32) || (n > 0 && (!variant_id || !compressed_cb || !p_base_wrong))) return fail(DMX_ERR_INVALID, "bad molecule calls");
    std::vector<unsigned long long> keys((size_t)n);
    for (int64_t j = 0; j < n; j++) {
        if (variant_id[j] < 0 || variant_id[j] >= c->V || compressed_cb[j] < 0 || compressed_cb[j] >= c->B)
            return fail(DMX_ERR_INVALID, "molecule call %lld outside the problem", (long long)j);
        keys[(size_t)j] = ((unsigned long long)(unsigned)variant_id[j] << 32) | (unsigned)compressed_cb[j];
    }
    unsigned long long *d_keys = nullptr;
    float *d_p = nullptr;
    HIP_TRY(hipMalloc((void **)&d_keys, sizeof(unsigned long long) * (size_t)(n ? n : 1)));
    if (hipMalloc((void **)&d_p, sizeof(float) * (size_t)(n ? n : 1)) != hipSuccess) {
        (void)hipFree(d_keys);
        return fail(DMX_ERR_HIP, "hipMalloc failed");
    }
    int rc = 0;
    if (n && (hipMemcpyAsync(d_keys, keys.data(), sizeof(unsigned long long) * n, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
              hipMemcpyAsync(d_p, p_base_wrong, sizeof(float) * n, hipMemcpyHostToDevice, c->stream) != hipSuccess))
        rc = fail(DMX_ERR_HIP, "upload of the molecule calls failed");
    if (rc == 0) rc = dmx::build_snp_groups(c, d_keys, nullptr, d_p, n);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(d_keys);
    (void)hipFree(d_p);
    return rc;
}

int dmx_get_max_pair_count(dmx_ctx *c, int64_t *max_count)
{
    if (!c || !max_count) return fail(DMX_ERR_INVALID, "null argument");
    if (!c->d_mc_start) return fail(DMX_ERR_INVALID, "call order: molecule calls (dmx_set_keep_molecule_calls + a device pack, or dmx_set_molecule_calls) first");
    *max_count = (int64_t)c->mc_max_count;
    return 0;
}

int dmx_estep_snp(dmx_ctx *c, int with_doublets, const double *count_pow, int64_t n_count_pow, const void *prior_logits,
                  int prior_dtype, double *logits_out, double *probs_out)
{
    if (!c) return fail(DMX_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    if (!c->have_problem || !c->have_probs) return fail(DMX_ERR_INVALID, "call order: genotype probabilities before dmx_estep_snp");
    if (!c->d_mc_start) return fail(DMX_ERR_INVALID, "call order: molecule calls (dmx_set_keep_molecule_calls + a device pack, or dmx_set_molecule_calls) first");
    if (!count_pow || n_count_pow <= (int64_t)c->mc_max_count) return fail(DMX_ERR_INVALID, "count_pow must cover counts 0..%u", c->mc_max_count);
    if (prior_logits && prior_dtype != DMX_F32 && prior_dtype != DMX_F64) return fail(DMX_ERR_INVALID, "prior_dtype must be DMX_F32 or DMX_F64");
    const int G = c->G;
    const long long K = with_doublets ? (long long)G * (G + 1) / 2 : G;
    if (K > SNP_MAX_OPTIONS) return fail(DMX_ERR_UNSUPPORTED, "aggregate_on_snps supports up to %d options (K=%lld)", SNP_MAX_OPTIONS, K);
    const size_t bk = (size_t)c->B * K;
    if (bk > c->cap_bk64) {
        dev_free(c, &c->d_logits64, c->cap_bk64);
        dev_free(c, &c->d_post64, c->cap_bk64);
        c->cap_bk64 = 0;
        DMX_TRY(dev_alloc(c, &c->d_logits64, bk));
        DMX_TRY(dev_alloc(c, &c->d_post64, bk));
        c->cap_bk64 = bk;
    }
    std::vector<unsigned> pairs((size_t)K);
    for (int g = 0; g < G; g++) pairs[g] = (unsigned)g | ((unsigned)g << 16);
    if (with_doublets) {
        size_t k = G;
        for (int g1 = 0; g1 < G; g1++)
            for (int g2 = g1 + 1; g2 < G; g2++) pairs[k++] = (unsigned)g1 | ((unsigned)g2 << 16);
    }
    unsigned *d_pairs = nullptr;
    double *d_pow = nullptr;
    void *d_prior = nullptr;
    HIP_TRY(hipMalloc((void **)&d_pairs, sizeof(unsigned) * K));
    int rc = 0;
    do {
        if (hipMalloc((void **)&d_pow, sizeof(double) * n_count_pow) != hipSuccess) { rc = fail(DMX_ERR_HIP, "hipMalloc failed"); break; }
        const size_t prior_bytes = prior_logits ? bk * (prior_dtype == DMX_F64 ? 8 : 4) : 0;
        if (prior_bytes && hipMalloc(&d_prior, prior_bytes) != hipSuccess) { rc = fail(DMX_ERR_HIP, "hipMalloc failed"); break; }
        hipError_t e = hipMemcpyAsync(d_pairs, pairs.data(), sizeof(unsigned) * K, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_pow, count_pow, sizeof(double) * n_count_pow, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess && prior_bytes) e = hipMemcpyAsync(d_prior, prior_logits, prior_bytes, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) { rc = fail(DMX_ERR_HIP, "dmx_estep_snp uploads: %s", hipGetErrorString(e)); break; }
        SnpArgs a;
        a.bc_start = c->d_mc_start;
        a.variant = c->d_mc_variant;
        a.e = c->d_mc_e;
        a.prob = c->d_prob;
        a.prow = c->d_prow;
        a.opt_pairs = d_pairs;
        a.count_pow = d_pow;
        a.prior = d_prior;
        a.prior_dtype = prior_dtype;
        a.logits = c->d_logits64;
        a.post = c->d_post64;
        a.log_bad = std::log(0.01 / (double)K);
        a.sum_plan = nullptr;
        a.sum_plan_values = 0;
        if (K > 1024) {
            rc = dmx::ensure_sum_plan(c, K);
            if (rc) break;
            a.sum_plan = c->d_sum_plan;
            a.sum_plan_values = c->sum_plan_values;
        }
        a.B = c->B;
        a.G = G;
        a.K = (int)K;
        const bool pairs_on = with_doublets != 0;
        if (K <= 64) rc = launch_snp<1>(c, a, pairs_on);
        else if (K <= 128) rc = launch_snp<2>(c, a, pairs_on);
        else if (K <= 256) rc = launch_snp<4>(c, a, pairs_on);
        else if (K <= 512) rc = launch_snp<8>(c, a, pairs_on);
        else if (K <= 1024) rc = launch_snp<16>(c, a, pairs_on);
        else if (K <= 256 * 6) rc = launch_snp_block<6>(c, a);  // K > 1024 is always the doublet table
        else if (K <= 256 * 9) rc = launch_snp_block<9>(c, a);
        else if (K <= 256 * 12) rc = launch_snp_block<12>(c, a);
        else if (K <= 256 * 17) rc = launch_snp_block<17>(c, a);
        else if (K <= 256 * 24) rc = launch_snp_block<24>(c, a);
        else rc = launch_snp_block<33>(c, a);
        if (rc) break;
        if (logits_out && bk) e = hipMemcpyAsync(logits_out, c->d_logits64, sizeof(double) * bk, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess && probs_out && bk) e = hipMemcpyAsync(probs_out, c->d_post64, sizeof(double) * bk, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { rc = fail(DMX_ERR_HIP, "dmx_estep_snp: %s", hipGetErrorString(e)); break; }
        // the float32 logits / posteriors of dmx_estep (and the M-step's bitmaps) were laid out for the previous K
        if ((int)K != c->K) c->have_post = false;
        c->K = (int)K;
        c->have_post64 = true;
    } while (false);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(d_pairs);
    if (d_pow) (void)hipFree(d_pow);
    if (d_prior) (void)hipFree(d_prior);
    return rc;
}

static int mstep_f64(dmx_ctx *c, double contribution_power, float *addition_out, double *sums_out)
{
    if (!c) return fail(DMX_ERR_INVALID, "null context");
    if (c->mshard)
        return fail(DMX_ERR_UNSUPPORTED, "the float64 M-step of aggregate_on_snps runs on a context without a communicator (its sums are added "
                                         "over the ranks by the caller: dmx_mstep_f64_sums)");
    HIP_TRY(hipSetDevice(c->device));
    if (!c->have_problem || !c->have_post64) return fail(DMX_ERR_INVALID, "call order: dmx_estep_snp before dmx_mstep_f64");
    if (c->attached()) return fail(DMX_ERR_UNSUPPORTED, "the float64 M-step does not run the device-side exchange: barcode-sharded "
                                                       "aggregate_on_snps runs add dmx_mstep_f64_sums over ranks on the host");
    const int G = c->G;
    double *sums = sums_out ? c->d_add64 : nullptr;
    if (G <= 64) launch_m64<1>(c, contribution_power, sums);
    else if (G <= 128) launch_m64<2>(c, contribution_power, sums);
    else if (G <= 256) launch_m64<4>(c, contribution_power, sums);
    else if (G <= 512) launch_m64<8>(c, contribution_power, sums);
    else launch_m64<16>(c, contribution_power, sums);
    HIP_TRY(hipGetLastError());
    c->add_partial = false;
    c->add_is_zero = false;
    const size_t vg = (size_t)c->V * G;
    if (addition_out && vg) HIP_TRY(hipMemcpyAsync(addition_out, c->d_add, sizeof(float) * vg, hipMemcpyDeviceToHost, c->stream));
    if (sums_out && vg) HIP_TRY(hipMemcpyAsync(sums_out, c->d_add64, sizeof(double) * vg, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int dmx_mstep_f64(dmx_ctx *c, double contribution_power, float *addition_out) { return mstep_f64(c, contribution_power, addition_out, nullptr); }

int dmx_mstep_f64_sums(dmx_ctx *c, double contribution_power, double *sums_out)
{
    if (!sums_out) return fail(DMX_ERR_INVALID, "null output");
    return mstep_f64(c, contribution_power, nullptr, sums_out);
}

}  // extern "C"
