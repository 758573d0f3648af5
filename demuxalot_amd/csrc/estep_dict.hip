// estep_dict.hip -- dictionary form of the E-step (demuxalot/demux.py:246-265) for genotype tables whose rows hold few
// DISTINCT probabilities.
//
// Before the first M-step - predict_posteriors, and iteration 0 of learn_genotypes - a row of genotype_prob is
// beta / sum(beta) of betas that the importers wrote from a handful of values (genotypes.py:147-164: 0, s/2, s and
// 0.1 x mean for donors without a call), so a row of G probabilities holds 2..4 distinct float32 values, and the
// (p1 + p2) * 0.5 of demux.py:190 at most 10.  The direct kernels (kernels.hip) evaluate numpy's float32 log once
// per (call, option): G (or K = G (G + 1) / 2) times per call.  Here it is evaluated once per (call, DISTINCT value):
//
//   k_build_dict        per table row: the distinct values (first-occurrence order, at most DICT_CAP) and, per
//                       genotype, the index of its value (as a byte offset into a row of float64 logs)
//   k_build_pair_codes  per (row, option): the index of the option's (p1 + p2) * 0.5 among the <= 10 pair values
//   k_estep_dict        one wavefront per barcode.  Phase A: lane (call j, entry e) computes
//                       f64(np.log(dict[v_j][e] * keep_j + floor_j)) - the SAME float32 operations on the SAME
//                       operands as the direct form - and parks it in LDS.  Phase B: lane k (option k) reads
//                       lp[j][code[v_j][k]] and adds it to its float64 accumulator, call after call in the barcode's
//                       order.  Every accumulator therefore receives exactly the addends of the direct form in
//                       exactly its order: logits and posteriors are bit-identical.
//   k_estep_dict_block  the same for wide doublet tables (K > 256): one 256-thread workgroup per barcode and tile of
//                       options, the singlet codes of a chunk of calls staged in LDS, four calls per dword.
//
// Per call and wavefront the work drops from ~68 VALU issue cycles (15.5 instructions of numpy's log per lane) to one
// byte gather + one ds_read_b64 + one v_add_f64 per option slot, plus 1/16th (4 entries) .. 1/6th (10 pair entries)
// of a log.  The host tries the form whenever the table was computed without a beta addition and takes it when every
// row fits (dmx_api.cpp: run_estep); after the first M-step rows are all-distinct and the direct kernels run.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "estep_epilogue.h"
#include "kernels.h"
#include "np_math.h"

namespace dmx {

// ------------------------------------------------------------------------------------
// Dictionary of one table row per wavefront.  Equality is equality of bit patterns (so +0 and -0, which log to the
// same value anyway, may take two entries).  stat[0] = max over rows of the number of distinct values
// (DICT_CAP + 1: some row has more).  Unused entries repeat entry 0 so that every entry logs to a finite number.
// ------------------------------------------------------------------------------------
template <int A>
__global__ __launch_bounds__(256) void k_build_dict(const float *__restrict__ prob, long long rows, int G,
                                                    float *__restrict__ dict, unsigned char *__restrict__ codes,
                                                    unsigned *__restrict__ stat)
{
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;  // wave-uniform
    unsigned x[A], code[A];
    unsigned long long rem[A];
#pragma unroll
    for (int s = 0; s < A; s++) {
        const int g = lane + 64 * s;
        const bool valid = g < G;
        x[s] = valid ? __float_as_uint(prob[(size_t)r * G + g]) : 0u;
        rem[s] = __ballot(valid);
        code[s] = 0u;
    }
    unsigned mine = 0u, first = 0u;
    int d = 0;
    for (; d < DICT_CAP; d++) {
        unsigned val = 0u;
        bool found = false;
#pragma unroll
        for (int s = 0; s < A; s++) {
            if (!found && rem[s] != 0ull) {  // wave-uniform
                val = (unsigned)__builtin_amdgcn_readlane((int)x[s], __builtin_ctzll(rem[s]));
                found = true;
            }
        }
        if (!found) break;
#pragma unroll
        for (int s = 0; s < A; s++) {
            const unsigned long long m = __ballot(x[s] == val) & rem[s];
            if ((m >> lane) & 1ull) code[s] = (unsigned)d;
            rem[s] &= ~m;
        }
        if (lane == d) mine = val;
        if (d == 0) first = val;
    }
    bool overflow = false;
#pragma unroll
    for (int s = 0; s < A; s++) overflow = overflow || rem[s] != 0ull;
    if (lane < DICT_CAP) dict[(size_t)r * DICT_CAP + lane] = __uint_as_float(lane < d ? mine : first);
#pragma unroll
    for (int s = 0; s < A; s++) {
        const int g = lane + 64 * s;
        if (g < G) codes[(size_t)r * G + g] = (unsigned char)(code[s] * 8u);
    }
    const unsigned n = overflow ? (unsigned)DICT_CAP + 1u : (unsigned)d;
    // one shared maximum: after the first rows it is at its final value and nobody writes any more
    if (lane == 0 && n > __hip_atomic_load(stat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(stat, n);
}

// Pair values of a row with D <= 4 distinct singlet values: entry t(lo, hi), lo <= hi, holds (dict[lo] + dict[hi]) * 0.5
// (float32 addition is commutative, so the order of an option's two genotypes does not matter).
//   t = 0..3: lo 0, hi 0..3;  4..6: lo 1, hi 1..3;  7..8: lo 2, hi 2..3;  9: (3, 3)
__host__ __device__ __forceinline__ unsigned pair_entry(unsigned c1, unsigned c2)
{
    const unsigned lo = c1 < c2 ? c1 : c2, hi = c1 < c2 ? c2 : c1;
    return lo * 4u - lo * (lo - 1u) / 2u + (hi - lo);
}
__device__ __forceinline__ unsigned pair_entry_lo(unsigned t) { return (unsigned)(0x3221110000ull >> (4u * t)) & 15u; }
__device__ __forceinline__ unsigned pair_entry_hi(unsigned t) { return (unsigned)(0x3323213210ull >> (4u * t)) & 15u; }

// ocodes[row, k] = 8 * pair_entry(code(g1), code(g2)) for every option k = (g1, g2) (singlets: g1 == g2)
__global__ __launch_bounds__(256) void k_build_pair_codes(const unsigned char *__restrict__ codes, const unsigned *__restrict__ opt_pairs,
                                                          long long rows, int G, int K, unsigned char *__restrict__ ocodes)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * K) return;
    const long long r = i / K;
    const unsigned pr = opt_pairs[(int)(i - r * K)];
    const unsigned c1 = codes[(size_t)r * G + (pr & 0xFFFFu)] >> 3, c2 = codes[(size_t)r * G + (pr >> 16)] >> 3;
    ocodes[i] = (unsigned char)(8u * pair_entry(c1, c2));
}

// ------------------------------------------------------------------------------------
// Lane-per-option form: one 64-thread workgroup (= one wavefront, so that the LDS address of a log IS its code) per
// barcode; option k in lane k & 63, slot k >> 6 (A slots).
//   NE      dictionary entries per call: 4 or 8 (singlets), 10 (pairs of <= 4 values)
//   CB      = 64 / NE calls whose logs one 64-lane pass computes; the packed log handles two passes at once, so a
//           super-batch of SB = 2 CB calls is produced at a time into one of two LDS buffers
//   PU      calls whose option codes are in flight together (prefetch unit), PU | CB
// Pipeline per super-batch i: the records of i + 2 and the dictionary rows of i + 1 are requested, the codes of the
// next unit are requested before a unit is consumed, the logs of i + 1 are computed after the units of i.
// ------------------------------------------------------------------------------------
template <int NE, int A, bool PAIRS>
struct DictShape {
    static constexpr int CB = 64 / NE;
    static constexpr int SB = 2 * CB;
    static constexpr int PU_WANT = A <= 2 ? 16 : A <= 4 ? 8 : A <= 8 ? 4 : 2;
    static constexpr int PU = CB % PU_WANT == 0 ? PU_WANT : CB;  // 16 / 8 / 6 calls for A <= 2
    static constexpr int NU = SB / PU;                            // units per super-batch (even)
};

struct DictRec {
    unsigned row;   // table row of the call
    float keep, flo;
};

template <int NE, int A, bool PAIRS>
__global__ __launch_bounds__(64) void k_estep_dict(EstepArgs a)
{
    using S = DictShape<NE, A, PAIRS>;
    constexpr int CB = S::CB, SB = S::SB, PU = S::PU, NU = S::NU;
    static_assert(NU % 2 == 0 && CB % PU == 0, "units tile the half batches");
    __shared__ __attribute__((aligned(16))) double sh_lp[2][SB * NE];
    const int lane = threadIdx.x;
    const int K = a.K;
    int kk[A];
    bool valid[A];
    unsigned voff[A];
#pragma unroll
    for (int s = 0; s < A; s++) {
        const int k = lane + 64 * s;
        valid[s] = k < K;
        kk[s] = valid[s] ? k : K - 1;
        voff[s] = (unsigned)kk[s];
    }
    double acc[A];
#pragma unroll
    for (int s = 0; s < A; s++) acc[s] = 0.0;

    const long long b = a.order[blockIdx.x];
    const long long pbeg = a.pair_ptr[b];
    const int n = 2 * (int)(a.pair_ptr[b + 1] - pbeg);  // calls incl. neutral padding, multiple of 8
    if (n > 0) {
        const unsigned *__restrict__ words = (const unsigned *)(a.pairs + pbeg);
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc((void *)a.ocodes, 0, (int)a.ocode_bytes, 0x00020000);
        const unsigned pitch = (unsigned)a.ocode_pitch;
        // phase A: this lane's (call, entry); lanes past CB * NE (NE = 10: 60..63) repeat a valid pair
        const int ja = (lane / NE) < CB ? lane / NE : CB - 1;
        const unsigned e = (unsigned)(lane % NE);
        const unsigned d1 = PAIRS ? pair_entry_lo(e) : e, d2 = PAIRS ? pair_entry_hi(e) : e;
        const int nsb = (n + SB - 1) / SB;

        auto load_rec = [&](int sb, int half) {
            int ci = sb * SB + half * CB + ja;
            ci = ci < n ? ci : n - 1;  // past the row: the last call again (computed, never consumed)
            const int w = (ci >> 1) * 8 + (ci & 1);
            DictRec r;
            r.row = words[w + 6];
            r.keep = __uint_as_float(words[w + 2]);
            r.flo = __uint_as_float(words[w + 4]);
            return r;
        };
        auto load_p = [&](const DictRec &r) {
            const float *row = a.dict + (size_t)r.row * DICT_CAP;
            float p = row[d1];
            if (PAIRS) p = (p + row[d2]) * 0.5f;  // demux.py:190
            return p;
        };
        // logs of super-batch sb; calls past the row's end park +0 (adding +0 leaves a sum unchanged: the sums
        // start at +0 and never become -0), so that phase B needs no per-call test
        auto produce = [&](int buf, int sb, const DictRec &rx, const DictRec &ry, float px, float py) {
            npm::f32x2 t;
            t.x = px * rx.keep;
            t.y = py * ry.keep;
            t.x = t.x + rx.flo;
            t.y = t.y + ry.flo;
            const npm::f32x2 lp = npm::log_f32_hot2(t);
            const int cix = sb * SB + ja, ciy = cix + CB;
            sh_lp[buf][ja * NE + e] = cix < n ? (double)lp.x : 0.0;
            sh_lp[buf][(CB + ja) * NE + e] = ciy < n ? (double)lp.y : 0.0;
        };
        // option codes of unit u of a super-batch whose records are (rx, ry): call q of the unit is call
        // u * PU + q of the super-batch = lane ((u * PU + q) % CB) * NE of rx (first half) or ry
        auto load_codes = [&](unsigned (&c)[PU][A], const DictRec &rx, const DictRec &ry, int u) {
#pragma unroll
            for (int q = 0; q < PU; q++) {
                const int jj = u * PU + q;
                const int srow = __builtin_amdgcn_readlane((int)(jj < CB ? rx.row : ry.row), (jj % CB) * NE);
                const int soff = (int)((unsigned)srow * pitch);
#pragma unroll
                for (int s = 0; s < A; s++)
                    c[q][s] = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rsrc, (int)voff[s], soff, 0);
            }
        };
        auto consume = [&](const unsigned (&c)[PU][A], int buf, int sb, int u) {
            if (sb * SB + u * PU >= n) return;  // wave-uniform: the whole unit lies past the row
            const char *base = (const char *)&sh_lp[buf][0];
            double v[PU][A];
#pragma unroll
            for (int q = 0; q < PU; q++)
#pragma unroll
                for (int s = 0; s < A; s++) v[q][s] = *(const double *)(base + (c[q][s] + (unsigned)((u * PU + q) * NE * 8)));
#pragma unroll
            for (int q = 0; q < PU; q++)  // call order
#pragma unroll
                for (int s = 0; s < A; s++) acc[s] += v[q][s];
        };

        DictRec cx = load_rec(0, 0), cy = load_rec(0, 1);  // records of the super-batch being consumed
        DictRec nx = cx, ny = cy;                         // ... of the next one
        if (nsb > 1) {
            nx = load_rec(1, 0);
            ny = load_rec(1, 1);
        }
        produce(0, 0, cx, cy, load_p(cx), load_p(cy));
        unsigned c0[PU][A], c1[PU][A];
        load_codes(c0, cx, cy, 0);
        // one super-batch; BUF is a compile-time constant so that every LDS address is code + immediate
        auto step = [&](auto buf_tag, int sb) {
            constexpr int BUF = decltype(buf_tag)::value;
            const bool more = sb + 1 < nsb;  // wave-uniform
            float px = 0.0f, py = 0.0f;
            DictRec fx = nx, fy = ny;  // records of super-batch sb + 2
            if (more) {
                px = load_p(nx);
                py = load_p(ny);
                if (sb + 2 < nsb) {
                    fx = load_rec(sb + 2, 0);
                    fy = load_rec(sb + 2, 1);
                }
            }
            __builtin_amdgcn_wave_barrier();  // the logs of this super-batch were stored by other lanes
#pragma unroll
            for (int u = 0; u < NU; u += 2) {
                load_codes(c1, cx, cy, u + 1);
                consume(c0, BUF, sb, u);
                if (u + 2 < NU)
                    load_codes(c0, cx, cy, u + 2);
                else if (more)
                    load_codes(c0, nx, ny, 0);
                consume(c1, BUF, sb, u + 1);
            }
            if (more) produce(BUF ^ 1, sb + 1, nx, ny, px, py);
            cx = nx;
            cy = ny;
            nx = fx;
            ny = fy;
        };
        for (int sb = 0; sb < nsb; sb += 2) {
            step(std::integral_constant<int, 0>{}, sb);
            if (sb + 1 < nsb) step(std::integral_constant<int, 1>{}, sb + 1);
        }
    }
    estep_epilogue<64, A>(a, b, true, acc, kk, valid, lane, lane, 0, n);
}

template <int A>
static void launch_build_dict(hipStream_t st, const float *prob, long long rows, int G, float *dict, unsigned char *codes, unsigned *stat)
{
    hipLaunchKernelGGL((k_build_dict<A>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, prob, rows, G, dict, codes, stat);
}

hipError_t launch_build_dict(hipStream_t st, const float *prob, long long rows, int G, float *dict, unsigned char *codes, unsigned *stat)
{
    hipError_t e = hipMemsetAsync(stat, 0, sizeof(unsigned), st);
    if (e != hipSuccess || rows == 0) return e;
    if (G <= 64) launch_build_dict<1>(st, prob, rows, G, dict, codes, stat);
    else if (G <= 128) launch_build_dict<2>(st, prob, rows, G, dict, codes, stat);
    else if (G <= 256) launch_build_dict<4>(st, prob, rows, G, dict, codes, stat);
    else if (G <= 512) launch_build_dict<8>(st, prob, rows, G, dict, codes, stat);
    else if (G <= 1024) launch_build_dict<16>(st, prob, rows, G, dict, codes, stat);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_build_pair_codes(hipStream_t st, const unsigned char *codes, const unsigned *opt_pairs, long long rows, int G, int K,
                                   unsigned char *ocodes)
{
    const long long n = rows * K;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_build_pair_codes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, codes, opt_pairs, rows, G, K, ocodes);
    return hipGetLastError();
}

template <int NE, bool PAIRS>
static hipError_t launch_dict_ne(hipStream_t st, const EstepArgs &a)
{
    const dim3 grid((unsigned)a.B), block(64);
    const int K = a.K;
    if (K <= 64) hipLaunchKernelGGL((k_estep_dict<NE, 1, PAIRS>), grid, block, 0, st, a);
    else if (K <= 128) hipLaunchKernelGGL((k_estep_dict<NE, 2, PAIRS>), grid, block, 0, st, a);
    else if (K <= 256) hipLaunchKernelGGL((k_estep_dict<NE, 4, PAIRS>), grid, block, 0, st, a);
    else if constexpr (PAIRS) return hipErrorInvalidValue;  // wider doublet tables: the workgroup-per-barcode form
    else if (K <= 512) hipLaunchKernelGGL((k_estep_dict<NE, 8, false>), grid, block, 0, st, a);
    else if (K <= 1024) hipLaunchKernelGGL((k_estep_dict<NE, 16, false>), grid, block, 0, st, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// a.dict_n distinct values per row at most (dict_form_fits says whether this form exists for the problem)
hipError_t launch_estep_dict(hipStream_t st, const EstepArgs &a, bool pairs)
{
    if (a.B == 0) return hipSuccess;
    if (pairs) return launch_dict_ne<10, true>(st, a);
    return a.dict_n <= 4 ? launch_dict_ne<4, false>(st, a) : launch_dict_ne<8, false>(st, a);
}

}  // namespace dmx
