// estep_epilogue.h -- what the E-step kernels of kernels.hip and estep_dict.hip share: the 8-byte barcode code of the
// M-step, numpy's pairwise row sum over register-resident values, and the softmax epilogue of the lane-per-option forms.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "np_math.h"

namespace dmx {

static __device__ __forceinline__ double shfl_xor_f64(double v, int mask)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, mask);
    hi = __shfl_xor(hi, mask);
    return __hiloint2double(hi, lo);
}

// What the M-step needs to know of a barcode whose row has at most NZ_CODE live posteriors, in 8 bytes
// (EstepArgs::first, written by the E-step epilogues): x = bits of the posterior of the lowest live genotype,
// y = count of live genotypes (7 bits) | the first four live genotypes (6 bits each).  One 8-byte gather per call
// from a 1.6 MB table (200k barcodes) is what the call-parallel part runs on; the 64-bit bitmap is only read for
// the dense calls.  [The M-step runs on its loads, not on its arithmetic: bitmap (8 B) + posterior
// (4 B) from two tables took 0.85 ms - the L2s at 65 % of their request rate -, one gather 0.70.]
constexpr int NZ_CODE = 4;
__device__ __forceinline__ uint2 nz_code(unsigned long long live, float first_posterior)
{
    unsigned code = (unsigned)__popcll(live);
#pragma unroll
    for (int t = 0; t < NZ_CODE; t++) {
        if (live) code |= (unsigned)__builtin_ctzll(live) << (7 + 6 * t);
        live &= live - 1ull;
    }
    return make_uint2(code & 127u ? __float_as_uint(first_posterior) : 0u, code);
}

// ------------------------------------------------------------------------------------
// helpers for the in-register softmax of the lane-per-option kernels: option k of a lane group of L lanes lives in
// lane (group base + k % L), register slot k / L.
// ------------------------------------------------------------------------------------
template <int L>
constexpr int log2_lanes()
{
    static_assert(L == 4 || L == 8 || L == 16 || L == 32 || L == 64, "lane groups are powers of two");
    return L == 4 ? 2 : L == 8 ? 3 : L == 16 ? 4 : L == 32 ? 5 : 6;
}

template <int L, int A>
static __device__ __forceinline__ float reg_elem(const float (&x)[A], int i, int gbase)
{
    float v = 0.0f;
#pragma unroll
    for (int a = 0; a < A; a++) {
        const float t = __shfl(x[a], gbase + (i & (L - 1)));
        v = ((i >> log2_lanes<L>()) == a) ? t : v;
    }
    return v;
}

// numpy pairwise block (n <= 128) over elements [start, start+n) held in registers.
// Lane groups are 8-aligned whenever n >= 8 can occur (several slots per lane only in groups of 8 lanes or more), so
// (lane & 7) indexes the 8 partial sums and the xor butterflies stay inside the group.
template <int L, int A>
static __device__ __forceinline__ float reg_block_sum(const float (&x)[A], int start, int n, int lane, int gbase)
{
    static_assert(A == 1 || L >= 8, "the 8 partial sums of numpy's pairwise block need 8 lanes");
    if (n < 8) {
        float res = 0.0f;
        for (int i = 0; i < n; i++) res += reg_elem<L, A>(x, start + i, gbase);
        return res;
    }
    const int j = lane & 7;
    const int nfull = n - (n & 7);
    float r = reg_elem<L, A>(x, start + j, gbase);
    for (int i = 8; i < nfull; i += 8) r += reg_elem<L, A>(x, start + i + j, gbase);
    r = r + __shfl_xor(r, 1);
    r = r + __shfl_xor(r, 2);
    r = r + __shfl_xor(r, 4);
    for (int i = nfull; i < n; i++) r += reg_elem<L, A>(x, start + i, gbase);
    return r;
}

static __device__ __forceinline__ int pw_half(int n)
{
    int h = n / 2;
    return h - (h % 8);
}

// np.sum of K <= 1024 register-resident elements (one 8192-element numpy chunk): the pairwise split
// tree walked iteratively, leaves (<= 128 elements) summed by reg_block_sum.  Uniform control flow.
template <int L, int A>
static __device__ __forceinline__ float reg_row_sum(const float (&x)[A], int K, int lane, int gbase)
{
    if (K <= 128) return reg_block_sum<L, A>(x, 0, K, lane, gbase);
    // explicit post-order traversal; depth <= 4 for K <= 1024
    int st_start[6], st_len[6], st_state[6];
    float st_left[6];
    int sp = 0;
    st_start[0] = 0;
    st_len[0] = K;
    st_state[0] = 0;
    float ret = 0.0f;
    while (sp >= 0) {
        const int s0 = st_start[sp], len = st_len[sp];
        if (len <= 128) {
            ret = reg_block_sum<L, A>(x, s0, len, lane, gbase);
            sp--;
            continue;
        }
        const int half = pw_half(len);
        if (st_state[sp] == 0) {
            st_state[sp] = 1;
            sp++;
            st_start[sp] = s0;
            st_len[sp] = half;
            st_state[sp] = 0;
        } else if (st_state[sp] == 1) {
            st_left[sp] = ret;
            st_state[sp] = 2;
            sp++;
            st_start[sp] = s0 + half;
            st_len[sp] = len - half;
            st_state[sp] = 0;
        } else {
            ret = st_left[sp] + ret;
            sp--;
        }
    }
    return ret;
}

template <int L>
static __device__ __forceinline__ int group_max_over_wave(int v)
{
#pragma unroll
    for (int off = L; off < 64; off <<= 1) v = max(v, __shfl_xor(v, off));
    return v;
}

// Operands are kept PAIR-MAJOR: element [q][s] holds calls 2q (.x) and 2q+1 (.y) of option slot s in
// one 64-bit register pair, i.e. exactly the packed operand - no register shuffling between the
// gather and the packed instructions (with call-major arrays the compiler staged them through LDS).
template <int A, bool PAIRS, int H>
static __device__ __forceinline__ void estep_terms(const npm::f32x2 (&p1)[H][A], const npm::f32x2 (&p2)[H][A],
                                                   const npm::f32x2 (&keep)[H], const npm::f32x2 (&flo)[H],
                                                   double (&acc)[A], int n_slots)
{
#pragma unroll
    for (int q = 0; q < H; q++) {
#pragma unroll
        for (int s = 0; s < A; s++) {
            if (A > 1 && s >= n_slots) continue;  // wave-uniform: slot entirely past the last option
            npm::f32x2 p = p1[q][s];
            if (PAIRS) p = (p + p2[q][s]) * 0.5f;
            npm::f32x2 t = p * keep[q];
            t = t + flo[q];
            const npm::f32x2 lp = npm::log_f32_hot2(t);
            acc[s] += (double)lp.x;  // call order preserved: 2q before 2q+1
            acc[s] += (double)lp.y;
        }
    }
}

// One batch of H call pairs (2H calls) of a wave-uniform record stream: keep / floor stay in SGPRs, the genotype
// rows are gathered with the row offset as the buffer load's scalar offset.
template <int A, int H, bool PAIRS>
struct RecBatch {
    npm::f32x2 p1[H][A], p2[H][A], keep[H], flo[H];
};

template <int A, int H, bool PAIRS>
static __device__ __forceinline__ void load_batch(RecBatch<A, H, PAIRS> &x, const CallPair *__restrict__ recs, int j,
                                                  __amdgpu_buffer_rsrc_t rsrc, const unsigned (&o1)[A], const unsigned (&o2)[A],
                                                  int n_slots)
{
    j = __builtin_amdgcn_readfirstlane(j);  // wave-uniform by construction; says so to the compiler (scalar loads)
#pragma unroll
    for (int q = 0; q < H; q++) {
        const CallPair r = recs[j + q];
        x.keep[q] = npm::f32x2{r.keep[0], r.keep[1]};
        x.flo[q] = npm::f32x2{r.floor[0], r.floor[1]};
#pragma unroll
        for (int s = 0; s < A; s++) {
            if (A > 1 && s >= n_slots) continue;
            x.p1[q][s].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)o1[s], (int)r.row_off[0], 0));
            x.p1[q][s].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)o1[s], (int)r.row_off[1], 0));
            if (PAIRS) {
                x.p2[q][s].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)o2[s], (int)r.row_off[0], 0));
                x.p2[q][s].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)o2[s], (int)r.row_off[1], 0));
            }
        }
    }
}

template <int L>
static __device__ __forceinline__ unsigned long long group_mask()
{
    return L >= 64 ? ~0ull : (1ull << (L & 63)) - 1ull;
}

// ------------------------------------------------------------------------------------
// Guarded mode (DMX_ESTEP_GUARDED, include/demux_hip.h).  The tolerance-mode kernels leave, per option k of a barcode
// with n (padded) calls, S'_k = their float64 sum of log terms.  Against the reference's S_k (float64 sum of numpy's
// float32 logs of the same float32 terms t_i in [1e-4, 1 + 1.0001e-4]):
//   |S_true - S_k|  <= RHO sum_i |log t_i| <= RHO (|S_true| + 2.1e-4 n)   numpy's float32 log is within RHO = 2.73e-7
//                      relative of the true log for every float32 in [1e-4, 2.0002] (exhaustive: tests/test_oracle_npsimd.py);
//                      terms above 1 are below 1.0001e-4 + 2^-23, hence the second summand
//   |S'_k - S_true| <= (n / 8) (7 x 2^-24 + 2 ulp(log2) ln 2) <= 6.3e-8 n   one float32 rounding per multiplication of
//                      the 8-term product (no underflow: P >= 1e-32), v_log_f32 of a mantissa in [0.5, 1) within 2 ulp
//                      (exhaustive on the device: tests/test_gpu_guarded.py), exponents exact, float64 accumulation
//   logit: both sides round pen + S to float32 (2^-24 relative each) and, with a prior, the sum with it once more.
// D_k is the sum of the three with the constants rounded up; D = max_k D_k.  With every logit off by at most D,
// p'_k / p_k and (1 - p'_k) / (1 - p_k) lie in [e^-2D, e^2D], so |p'_k - p_k| <= min(p_k, 1 - p_k) (e^2D - 1); the
// float32 evaluation of the softmax (numpy's exp, pairwise sum, division: a few ulp of p on either side) gets the
// 2e-6 between GUARD_TOL and the contract's 1e-5.  The argmax is the reference's when no other logit is within 2 D
// of the largest.  Returns true (uniform over the lane group) when the barcode must be redone exactly.
// ------------------------------------------------------------------------------------
constexpr float GUARD_RHO = 3.0e-7f;
constexpr float GUARD_POSITIVE_TERM = 2.1e-4f;
constexpr float GUARD_PER_CALL = 7.0e-8f;
// k_estep_pairblocks stages the rows PRE-SCALED, u_g = p_g (keep / 2), and forms a term as (u_g1 + u_g2) + floor: two packed
// additions where (p_g1 + p_g2) (keep / 2) + floor takes an addition, a multiplication and an addition (the kernel is VALU-bound:
// 85 % busy, profiles/r5_pmc_pairblocks.txt).  That is no longer the reference's own float32 term: u_g1, u_g2 and their sum carry
// one rounding each where the reference's (p_g1 + p_g2) and its product with keep / 2 carry two, so the part of the term that
// depends on p differs by at most 3 x 2^-24 relative, the term (>= that part) by at most that plus one more rounding of the sum
// with the floor: 4 x 2^-24 = 2.4e-7 relative per term = 2.4e-7 in its log, on top of GUARD_PER_CALL.
constexpr float GUARD_PER_CALL_PRESCALED = 3.1e-7f;
constexpr float GUARD_LOGIT_ROUNDING = 1.2e-7f;  // 2 x 2^-24 (reference and here) with the conversions' slack
constexpr int GUARD_ALT_SAMPLE = 8;  // one barcode in 8 is shown to the guard of the pass that does not run (estep_epilogue)
constexpr float GUARD_TOL = 8.0e-6f;
constexpr float GUARD_TOL_WIDE = 6.0e-6f;  // rows of more than 1024 options (k_softmax_rows): numpy's pairwise sum has three more
                                           // levels there, so the float32 evaluation of the softmax gets 4e-6 instead of 2e-6

template <int L, int A>
static __device__ __forceinline__ bool estep_guard(const float (&dev)[A], const float (&lg)[A], const float (&post)[A],
                                                   const bool (&valid)[A], float mx, int gbase)
{
    float dmax = 0.0f;
    bool bad = false;
#pragma unroll
    for (int s = 0; s < A; s++) {
        if (!valid[s]) continue;
        bad |= !(dev[s] < 40.0f);  // also NaN; e^{2D} stays finite below
        dmax = fmaxf(dmax, dev[s]);
    }
#pragma unroll
    for (int off = 1; off < L; off <<= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, off));
    const float x2 = 2.0f * dmax;
    // >= e^{2D} - 1: x (1 + x) while 2 D < 0.5 (every form but the coarse pass, whose D grows by 2^-9 per call: k_estep_tiled_coarse),
    // else the hardware's 2^x (within 1 ulp) of an exponent rounded up, with a margin of 2e-5 relative
    const float e2 = dmax < 0.25f ? x2 * (1.0f + x2) * 1.000001f : __builtin_amdgcn_exp2f(x2 * 1.4426953f) * 1.00002f - 1.0f;
    const float near = mx - (x2 + 2.4e-7f * fabsf(mx)) * 1.000001f;  // logits at or above this could be the reference's argmax
    int close = 0;
#pragma unroll
    for (int s = 0; s < A; s++) {
        const float m = fminf(post[s], 1.0f - post[s]);
        bad |= valid[s] && !(m * e2 <= GUARD_TOL);
        const unsigned long long bal = __ballot(valid[s] && !(lg[s] < near));
        close += __popcll(L == 64 ? bal : (bal >> gbase) & group_mask<L>());
    }
    const unsigned long long any_bad = __ballot(bad);
    return close != 1 || (L == 64 ? any_bad : (any_bad >> gbase) & group_mask<L>()) != 0ull;
}

// The guard's bookkeeping for one barcode (one lane of its lane group acts): queued for the exact redo (a.guard == 1: appended to
// the sub-queue of its residue class, k_guard_compact makes the dense list of them), or - in the exact kernels of an E-step
// that runs direct (a.guard == 2) - only counted, on hashed counters (200k atomics on one address took 0.65 ms;
// k_guard_begin adds the slots up).
static __device__ __forceinline__ void guard_note(const EstepArgs &a, long long b, bool counting, bool coarse_set = false)
{
    if (counting) {
        atomicAdd(a.guard_count + (coarse_set ? GS_SLOTS_COARSE : GS_SLOTS_FINE) + (int)(b & (GUARD_SLOTS - 1)), 1u);
    } else {
        const unsigned q = (unsigned)(b & (GUARD_QUEUES - 1));  // (at most ceil(B / GUARD_QUEUES) barcodes share a queue)
        a.guard_sub[(size_t)q * a.guard_sub_cap + atomicAdd(a.guard_count + GS_QUEUE_LEN + q, 1u)] = (int)b;
    }
}
// A fast kernel of a guarded E-step that runs direct (kernels.h: EstepArgs::direct) stands back: k_guard_compact, between it
// and the exact launch, then lists ALL barcodes (longest rows first) in the queue, so that the exact launch walks every barcode
// with the very code that walks the queue.  Nothing is stored here, and the exact kernels choose no pointer at run time: a
// store ahead of the kernels' uniform loads (records, offsets), or a selected base pointer, turned those scalar loads into
// vector loads - +40 % on the exact kernel's time, 2 x on the 64-lane fast kernel's.  Returns true when the caller has to return.
static __device__ __forceinline__ bool guard_stand_back(const EstepArgs &a)
{
    return a.guard == 1 && a.direct != nullptr && *a.direct != 0u;
}
// whether this launch evaluates the guard: the fast kernels of a guarded E-step always, its exact kernels when the E-step runs direct
static __device__ __forceinline__ bool guard_active(const EstepArgs &a)
{
    return a.guard == 1 || (a.guard == 2 && a.direct != nullptr && *a.direct != 0u);
}

// Epilogue of the lane-per-option forms: penalties, optional prior, softmax as scipy evaluates it, the M-step's bitmap.
// acc[s] = float64 sum of the log terms of option kk[s] = li + L * s of barcode b (one lane group of L lanes per barcode).
// The guard (a.guard; estep_guard above) is evaluated by the tolerance-mode kernels of a guarded E-step and, for the count
// alone, by its exact kernels when the E-step runs direct (kernels.h: EstepArgs::direct).
template <int L, int A, bool GUARD = false>
static __device__ __forceinline__ void estep_epilogue(const EstepArgs &a, long long b, bool live, const double (&acc)[A],
                                                      const int (&kk)[A], const bool (&valid)[A], int lane, int li, int gbase,
                                                      int row_calls, float accum_extra = 0.0f)
{
    const int K = a.K;
    float lg[A], x[A];
    float mx = -__builtin_inff();
#pragma unroll
    for (int s = 0; s < A; s++) {
        const double t = (double)a.pen[kk[s]] + acc[s];
        float l = (float)t;
        if (a.prior) {
            const size_t o = (size_t)b * K + kk[s];
            if (a.prior_dtype == DMX_F32)
                l = l + ((const float *)a.prior)[o];
            else
                l = (float)((double)l + ((const double *)a.prior)[o]);
        }
        lg[s] = l;
        mx = valid[s] ? fmaxf(mx, l) : mx;
    }
#pragma unroll
    for (int off = 1; off < L; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
#pragma unroll
    for (int s = 0; s < A; s++) x[s] = npm::exp_f32(lg[s] - mx);
    const float tot = reg_row_sum<L, A>(x, K, lane, gbase);
    const int W = (a.G + 63) >> 6;
    unsigned long long mine = 0ull;  // L < 64: the group's bitmap of live singlet posteriors, over all slots
    float post[A];
#pragma unroll
    for (int s = 0; s < A; s++) {
        post[s] = x[s] / tot;
        if (live && valid[s]) {
            const size_t o = (size_t)b * K + kk[s];
            a.logits[o] = lg[s];
            a.post[o] = post[s];
            if (a.post_singlets != nullptr && kk[s] < a.G) a.post_singlets[(size_t)b * a.G + kk[s]] = post[s];
        }
        // non-zero bitmap of the singlet columns (the M-step skips exact zeros: (0*keep)^2 = +0)
        const unsigned long long bal = __ballot(live && valid[s] && (li + L * s) < a.G && !(post[s] <= a.nz_floor));
        if (L == 64) {
            if (lane == 0 && s < W) a.nz[(size_t)b * W + s] = bal;
            if (s == 0) mine = bal;
        } else if (L * s < 64) {
            mine |= ((bal >> gbase) & group_mask<L>()) << ((L * s) & 63);
        }
    }
    if (L < 64 && live && li == 0) a.nz[(size_t)b] = mine;
    bool redo = false;  // the exact kernel takes this barcode again (it rewrites everything written here)
    if (a.guard && guard_active(a)) {  // (uniform)
        const bool counting = a.guard == 2;
        float dev[A];  // bound on |logit - reference logit| per option
        const float n = (float)row_calls;
        auto bound = [&](float per_call, float accum) {
#pragma unroll
            for (int s = 0; s < A; s++) {
                const float l0 = (float)((double)a.pen[kk[s]] + acc[s]);
                dev[s] = GUARD_RHO * (fabsf((float)acc[s]) + GUARD_POSITIVE_TERM * n) + per_call * (n + 8.0f) + GUARD_LOGIT_ROUNDING * fabsf(l0);
                // a sum of n / 8 log2 values accumulated in float32: every addition rounds to 2^-24 of a partial sum, and the partial
                // sums stay below |sum| + 2 x (the positive terms: < 1.5e-4 log2 units per call) - here in natural-log units
                // (accum_extra: what the caller's partial sums may exceed the total by, beyond that)
                if (accum != 0.0f) dev[s] += accum * (0.125f * n + 2.0f) * (fabsf((float)acc[s]) + 3.0e-4f * n + accum_extra);
                if (a.prior) dev[s] += GUARD_LOGIT_ROUNDING * fabsf(lg[s]);
            }
        };
        bound(a.guard_per_call, a.guard_accum);
        const bool flagged = estep_guard<L, A>(dev, lg, post, valid, mx, gbase);
        if (flagged && live && li == 0) guard_note(a, b, counting, a.guard_main_coarse != 0);
        redo = flagged && !counting;
        // What the OTHER pass's guard would say, so that the device can price that pass (k_guard_begin): a SAMPLE - the barcodes
        // b % 8 == 0 of a lane group's row, each counted 8 times - since the second evaluation costs a short row's kernel 13 %
        // (200k x 50 calls: fine pass 0.351 -> 0.398 ms with every barcode evaluated twice).  L < 64: the groups of a wavefront
        // hold different barcodes, so the evaluation stays uniform there and only the count is sampled.
        if (a.guard_alt_per_call != 0.0f && (L < 64 || (b & (GUARD_ALT_SAMPLE - 1)) == 0)) {
            bound(a.guard_alt_per_call, a.guard_alt_accum);
            const bool flagged_alt = estep_guard<L, A>(dev, lg, post, valid, mx, gbase);
            if (flagged_alt && live && li == 0 && (b & (GUARD_ALT_SAMPLE - 1)) == 0)
                atomicAdd(a.guard_count + (a.guard_main_coarse == 0 ? GS_SLOTS_COARSE : GS_SLOTS_FINE) + (int)((b / GUARD_ALT_SAMPLE) & (GUARD_SLOTS - 1)), (unsigned)GUARD_ALT_SAMPLE);
        }
    }
    if (a.first) {
        // what the M-step's call-parallel part needs of this barcode, 8 bytes (nz_code): ONE gather per call there,
        // from a table small enough to stay in L2
        const int g0 = mine ? __builtin_ctzll(mine) : 0;  // the lowest live genotype: lane g0 % L, slot g0 / L
        float p0 = post[0];
#pragma unroll
        for (int s = 1; s < A; s++) p0 = (g0 >> log2_lanes<L>()) == s ? post[s] : p0;
        if (live && li == (g0 & (L - 1))) a.first[b] = nz_code(mine, p0);
        // statistic for the M-step's choice of kernel (G <= 64): calls whose barcode has more than 4 live posteriors
        if (a.dense_calls && live && !redo && li == 0 && __popcll(mine) > 4)
            atomicAdd(a.dense_calls + 1 + (b & (DENSE_SLOTS - 1)), (unsigned long long)row_calls);
    }
}

}  // namespace dmx
