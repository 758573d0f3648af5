// kernels.hip -- gfx950 kernels of the Demultiplexer EM hot path.
//
//   k_probs_from_betas   P-step, one lane group per SNP group   demuxalot/demux.py:267-274
//   k_estep_direct       E-step + softmax, one lane group (4..64 lanes) per barcode, options on lanes,
//                        up to 16 options per lane (K <= 1024; doublet tables up to 512)   demux.py:246-265, :101/:152
//   k_estep_tiled        the same for singlet tables of 33..128 genotypes on many barcodes, bins of barcodes walked
//                        variant tile by variant tile (tolerance mode)
//   k_estep_block        E-step logits, one 256-thread workgroup per barcode and tile of options, genotype
//                        rows staged in LDS (wider option tables)
//   k_softmax_rows       softmax, bitmaps and barcode codes of the rows k_estep_block left as logits
//   k_mstep_calls / k_mstep_dense (G <= 64, few / many live posteriors per barcode), k_mstep (G > 64),
//   k_mcombine, k_mstep_exact
//                        M-step (variant-major, no atomics)      demux.py:113-118
//   k_assign             per-barcode argmax of the posterior
//
// Numerics: float32 element-wise work in the reference's operation order, float64
// accumulation (np.bincount), one float32 rounding; log/exp/sum as numpy evaluates them
// (np_math.h).  Built with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include <type_traits>
#include <utility>
#include "kernels.h"
#include "np_math.h"
#include "estep_epilogue.h"

namespace dmx {

// ------------------------------------------------------------------------------------
// P-step.  One thread per (variant, genotype).  The per-SNP denominator is the float64 sum
// of beta over the SNP's variants in increasing variant index (np.bincount order).
// ------------------------------------------------------------------------------------
// T = float: the EM path (betas + addition, both float32).  T = double: caller-supplied float64 betas
// (numpy then divides float64 by float64 and rounds once to float32; no addition).
// L lanes share one variant row (genotypes li, li + L, ...); the row of a SNP's FIRST variant does the whole group, so
// that every beta is read once and the denominator is formed once per SNP (two variants per SNP: half the rows exit).
template <typename T, int L>
__global__ __launch_bounds__(256) void k_probs_from_betas(const T *__restrict__ prior,
                                                          const T *__restrict__ addition,
                                                          const int *__restrict__ v2snp,
                                                          const int *__restrict__ snp_ptr,
                                                          const int *__restrict__ snp_vars, long long v_begin,
                                                          long long n_rows, long long n_snps, int G,
                                                          const int *__restrict__ prow, float clip_lo, float clip_hi,
                                                          float *__restrict__ prob, unsigned short *__restrict__ prob16)
{
    // prob16 (nullable): the table once more, rounded to binary16 at the float32 table's row offsets (EstepArgs::prob16: the coarse
    // pass of the E-step that follows; as k_prob_to_half writes it).
    // n_snps >= 0: the whole table, one lane group per SNP 0 .. n_snps - 1.  Otherwise variants [v_begin, v_begin +
    // n_rows) (whole SNP groups: a rank's slice), one lane group per variant, the SNP's first variant working.
    // prob row of variant v = prow[v] (padded multi-GPU layout) or v.
    const int lane = threadIdx.x & 63, li = lane & (L - 1);
    const long long r = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 / L) + lane / L;
    if (r >= (n_snps >= 0 ? n_snps : n_rows)) return;
    const int snp = n_snps >= 0 ? (int)r : v2snp[v_begin + r];
    const int j0 = snp_ptr[snp], j1 = snp_ptr[snp + 1];
    if (n_snps < 0 && snp_vars[j0] != v_begin + r) return;
    constexpr int HELD = 4;  // the first variants of the group: their loads are issued together, their betas stay in registers
    const int cnt = j1 - j0;
    long long w[HELD];
#pragma unroll
    for (int q = 0; q < HELD; q++) w[q] = q < cnt ? (long long)snp_vars[j0 + q] : -1;
    for (int g = li; g < G; g += L) {
        T held[HELD];
#pragma unroll
        for (int q = 0; q < HELD; q++) {
            held[q] = (T)0;
            if (w[q] >= 0) {
                const long long o = w[q] * G + g;
                held[q] = addition ? prior[o] + addition[o] : prior[o];  // float32 add, demux.py:90
            }
        }
        double den = 0.0;
#pragma unroll
        for (int q = 0; q < HELD; q++) den += (double)held[q];  // increasing variant index (np.bincount order); absent: + 0
        for (int j = j0 + HELD; j < j1; j++) {
            const long long o = (long long)snp_vars[j] * G + g;
            den += (double)(addition ? prior[o] + addition[o] : prior[o]);
        }
        den = fmax(den, 1e-7);
#pragma unroll
        for (int q = 0; q < HELD; q++) {
            if (w[q] < 0) continue;
            float p = (float)((double)held[q] / den);
            p = fminf(fmaxf(p, clip_lo), clip_hi);  // ndarray.clip(lo, hi) = minimum(maximum(x, lo), hi)
            const long long row = prow ? (long long)prow[w[q]] : w[q];
            prob[row * G + g] = p;
            if (prob16) prob16[row * (2 * G) + g] = __builtin_bit_cast(unsigned short, (_Float16)p);
        }
        for (int j = j0 + HELD; j < j1; j++) {
            const long long v_j = snp_vars[j];
            const long long o = v_j * G + g;
            const T beta = addition ? prior[o] + addition[o] : prior[o];
            float p = (float)((double)beta / den);
            p = fminf(fmaxf(p, clip_lo), clip_hi);
            const long long row = prow ? (long long)prow[v_j] : v_j;
            prob[row * G + g] = p;
            if (prob16) prob16[row * (2 * G) + g] = __builtin_bit_cast(unsigned short, (_Float16)p);
        }
    }
}

// ------------------------------------------------------------------------------------
// Regularised prior betas (demuxalot/demux.py:367-388):
//   scale[v] = 1 + [data] n_mol[v] / (sum_snp n_mol + 100) + f64(bsum[v]) / (sum_snp f64(bsum) + 100)
//   beta'[v,g] = beta[v,g] + f32(scale[v] * default_prior)
// bsum[v] = np.sum(beta[v,:]) in float32 with numpy's pairwise association; the per-SNP sums run over the
// SNP's variants in increasing variant index in float64 (np.bincount order).
// Two small kernels: row sums (one wavefront per variant), then one thread per (variant, genotype).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_beta_rowsum(const float *__restrict__ betas, long long V, int G,
                                                     float *__restrict__ bsum)
{
    const int lane = threadIdx.x & 63;
    const long long v = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= V) return;
    const float tot = npm::row_sum_wave(betas + (size_t)v * G, G, lane);
    if (lane == 0) bsum[v] = tot;
}

__global__ __launch_bounds__(256) void k_prior_betas(const float *__restrict__ betas, const float *__restrict__ bsum,
                                                     const unsigned long long *__restrict__ n_mol,  // nullable
                                                     const int *__restrict__ v2snp, const int *__restrict__ snp_ptr,
                                                     const int *__restrict__ snp_vars, long long V, int G,
                                                     double default_prior, float *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V * G) return;
    const long long v = i / G;
    const int snp = v2snp[v];
    double mol_snp = 0.0, beta_snp = 0.0;
    for (int j = snp_ptr[snp]; j < snp_ptr[snp + 1]; j++) {
        const int u = snp_vars[j];
        if (n_mol) mol_snp += (double)n_mol[u];
        beta_snp += (double)bsum[u];
    }
    double scale = 1.0;
    if (n_mol) scale = scale + (double)n_mol[v] / (mol_snp + 100.0);
    scale = scale + (double)bsum[v] / (beta_snp + 100.0);
    out[i] = betas[i] + (float)(scale * default_prior);
}

// value of lane (group base + i) for every lane of the group; i is wave-uniform
template <int L, typename T>
static __device__ __forceinline__ T group_bcast(T v, int i, int gbase)
{
    static_assert(sizeof(T) == 4, "32-bit payloads");
    if (L == 64) {
        const int r = __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), i);  // -> SGPR
        return __builtin_bit_cast(T, r);
    }
    return __shfl(v, gbase + i);
}

// ------------------------------------------------------------------------------------
// E-step, direct form (K <= 1024).  A wavefront is cut into 64/L lane groups; every group owns
// one barcode and walks its calls IN ORDER (so the float64 sum has the reference's bincount
// association), lane l of the group accumulates option l (+64*s).  Barcodes are handed out from
// a row-length-sorted list, so the groups of a wave (and neighbouring waves) finish together.
//   L == 64: the whole wave works on one barcode, so the call records are wave-uniform: they are
//            fetched with scalar loads (s_load_dwordx8 per pair of calls) and keep/floor reach
//            the packed multiplies/adds as SGPR pairs - no per-call VALU work besides the terms.
//   L <  64: lane i of a group fetches the fields of call c0+i, the group broadcasts them call by
//            call through ds_bpermute.
// Per call the group reads G*4 contiguous bytes of the prob table (global_load_dword, SGPR base).
// ------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------
// Tolerance mode of the E-step (dmx_set_estep_mode(ctx, DMX_ESTEP_FAST)).  The contract of the path is
// "assignments identical, posteriors within 1e-5" (BASELINE.json north_star); the default mode above pays
// ~68 VALU issue cycles per term to repeat numpy's float32 log bit for bit.  Here the terms of 8 consecutive
// calls are MULTIPLIED in float32 and one hardware log2 is taken per 8 calls:
//     sum_c log(t_c) = ln2 * sum_chunks [ exponent(P) + log2(mantissa(P)) ],   P = prod of the chunk's 8 terms.
// t_c >= 1e-4, so P >= 1e-32 never underflows; t_c <= 2, so P <= 256.  The mantissa logs (in [-1, 0]) are
// accumulated in float64, the exponents as integers.  7 roundings of relative size 2^-24 and one 1-ulp log2 per
// 8 terms are smaller than the rounding noise numpy's own float32 log leaves in the reference's terms.
// ------------------------------------------------------------------------------------
struct FastAcc {
    double mant;  // sum of log2(mantissa) of the flushed chunk products
    int expo;     // sum of their binary exponents
};

template <int A, bool PAIRS, int H>
static __device__ __forceinline__ void estep_products(const npm::f32x2 (&p1)[H][A], const npm::f32x2 (&p2)[H][A],
                                                      const npm::f32x2 (&keep)[H], const npm::f32x2 (&flo)[H],
                                                      float (&prod)[A], int n_slots)
{
#pragma unroll
    for (int q = 0; q < H; q++) {
#pragma unroll
        for (int s = 0; s < A; s++) {
            if (A > 1 && s >= n_slots) continue;
            npm::f32x2 p = p1[q][s];
            if (PAIRS) p = (p + p2[q][s]) * 0.5f;
            npm::f32x2 t = p * keep[q];
            t = t + flo[q];
            prod[s] = (prod[s] * t.x) * t.y;
        }
    }
}

template <int A>
static __device__ __forceinline__ void estep_flush(float (&prod)[A], FastAcc (&acc)[A], int n_slots)
{
#pragma unroll
    for (int s = 0; s < A; s++) {
        if (A > 1 && s >= n_slots) continue;
        const float m = __builtin_amdgcn_frexp_mantf(prod[s]);  // [0.5, 1); NaN stays NaN
        acc[s].expo += __builtin_amdgcn_frexp_expf(prod[s]);
        acc[s].mant += (double)__builtin_amdgcn_logf(m);        // v_log_f32 = log2
        prod[s] = 1.0f;
    }
}

// ------------------------------------------------------------------------------------
// Tolerance mode with doublets, 64 lanes per barcode: an option (g1, g2) needs p[g1] and p[g2] of every call, and as two
// per-lane gathers per call and option slot it was the address unit, not arithmetic, that the kernel ran on (20k x 20k x 8
// with doublets, K = 36: 16 gathers per 8 calls).  The G <= GPAD probabilities of a call's row are contiguous, so ONE
// load fetches the rows of RPL = 64 / GPAD calls - lane l: call l / GPAD of the load, genotype l % GPAD - and every
// lane picks its two values out of the wavefront with ds_bpermute (the LDS crossbar, no memory involved).  Same values,
// same arithmetic as the gather form: bit-identical sums.
//   A = 1 (K <= 64: G <= 10), A = 2 (K <= 128: G <= 15): GPAD 16;  A <= 8 (K <= 512: G <= 31): GPAD 32
// ------------------------------------------------------------------------------------
template <int A>
struct PairRowShape {
    static constexpr int GPAD = A <= 2 ? 16 : A <= 8 ? 32 : 64;
    static constexpr int RPL = 64 / GPAD;  // calls per row load
};

template <int H, int NL>
struct PairRowBatch {
    float raw[NL];  // load i: the rows of calls i * RPL .. of the batch
    npm::f32x2 keep[H], flo[H];
};

template <int A, int H, int NL>
static __device__ __forceinline__ void load_pair_rows(PairRowBatch<H, NL> &x, const CallPair *__restrict__ recs, int j,
                                                      __amdgpu_buffer_rsrc_t rsrc, int lane)
{
    constexpr int GPAD = PairRowShape<A>::GPAD, RPL = PairRowShape<A>::RPL;
    static_assert(NL * RPL == 2 * H, "whole row loads per batch");
    j = __builtin_amdgcn_readfirstlane(j);
    unsigned ro[2 * H];
#pragma unroll
    for (int q = 0; q < H; q++) {
        const CallPair r = recs[j + q];
        x.keep[q] = npm::f32x2{r.keep[0], r.keep[1]};
        x.flo[q] = npm::f32x2{r.floor[0], r.floor[1]};
        ro[2 * q] = r.row_off[0];
        ro[2 * q + 1] = r.row_off[1];
    }
    const unsigned goff = (unsigned)(lane % GPAD) * 4u;  // (genotypes past G: bytes of the next row or past the table, never used)
#pragma unroll
    for (int i = 0; i < NL; i++) {
        unsigned sel = ro[i * RPL + RPL - 1];
#pragma unroll
        for (int c = RPL - 2; c >= 0; c--) sel = lane < (c + 1) * GPAD ? ro[i * RPL + c] : sel;
        x.raw[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(sel + goff), 0, 0));
    }
}

// o1 / o2: byte offsets of the option's two genotypes inside a row = 4 x their lane inside a call's GPAD lanes
template <int A, int H, int NL>
static __device__ __forceinline__ void pair_row_products(const PairRowBatch<H, NL> &x, const unsigned (&o1)[A], const unsigned (&o2)[A],
                                                         float (&prod)[A], int n_slots)
{
    constexpr int GPAD = PairRowShape<A>::GPAD, RPL = PairRowShape<A>::RPL;
    auto pick = [&](unsigned off, int call) {
        return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((int)(off + (unsigned)((call % RPL) * GPAD * 4)),
                                                                      __builtin_bit_cast(int, x.raw[call / RPL])));
    };
#pragma unroll
    for (int q = 0; q < H; q++) {
#pragma unroll
        for (int s = 0; s < A; s++) {
            if (A > 1 && s >= n_slots) continue;
            const npm::f32x2 p1{pick(o1[s], 2 * q), pick(o1[s], 2 * q + 1)};
            const npm::f32x2 p2{pick(o2[s], 2 * q), pick(o2[s], 2 * q + 1)};
            const npm::f32x2 p = (p1 + p2) * 0.5f;
            npm::f32x2 t = p * x.keep[q];
            t = t + x.flo[q];
            prod[s] = (prod[s] * t.x) * t.y;
        }
    }
}

// ------------------------------------------------------------------------------------
// Tolerance mode, K <= 64 singlets (one accumulator per lane): the software pipeline of the record stream.
// With ~10 VALU cycles per term there is nothing to hide latency behind, and the chain
//     scalar load of the call records (misses the scalar cache: the 1.3 GB stream is read once, from HBM)
//     -> row gathers (their addresses come from the records) -> arithmetic
// is paid in full for every batch of 8 calls (measured: 1.72 ms, the same as the gathers alone in round 1).  Here
// the records travel through the VECTOR memory path instead: a batch of 8 calls is 32 dwords, loaded by one
// coalesced 128-byte load (lane l holds dword l), 7 batches ahead of their use; row offsets, keep and floor are then
// moved to SGPRs with v_readlane (24 per batch); the gathers of batches k+1 .. k+3 are in flight while batch k is consumed.
//   on_group(k, tag)  called before batch k is consumed with the slot tag of its records (tile-major form)
// ------------------------------------------------------------------------------------
struct FastBatch {
    npm::f32x2 p[4];  // gathered genotype probabilities of the batch's 8 calls (call 2q in .x, 2q+1 in .y)
};
// Experiment knobs of the row gathers (make EXPERIMENTS=1 CXXEXTRA="-DDMX_ROW_AUX=2"; profiles/r5_estep_gather_experiments.txt):
//   DMX_ROW_AUX       cache-policy bits of the gathers' buffer loads: 0 default, 2 nt, 16 sc1, 17 sc0 sc1 (the last three bypass the L1)
//   DMX_GATHER_DEPTH  batches of 8 gathers a wavefront keeps in flight + 1 (4: the shipped pipeline)
#ifndef DMX_ROW_AUX
#define DMX_ROW_AUX 0
#endif
#ifndef DMX_GATHER_DEPTH
#define DMX_GATHER_DEPTH 4
#endif
#ifndef DMX_PAIRBLOCK_SERIAL_PRODUCT
#define DMX_PAIRBLOCK_SERIAL_PRODUCT 0  // 1: k_estep_pairblocks multiplies the two terms of a pair of calls one after the other (until round 5)
#endif

// HALF (tables of 17 .. 32 genotypes): a 256-byte gather holds the rows of TWO calls - lanes 0 .. 31 take the even call of
// a pair, lanes 32 .. 63 the odd one (the row offset, keep and floor of a lane's call selected per half with v_cndmask) -,
// so a batch of 8 calls is 4 gathers and every lane multiplies the 4 terms of its half; the two halves' sums are added
// when the barcode is finished.  Half the gathers per call of the one-call-per-gather form.
template <bool HALF, typename OnGroup>
static __device__ __forceinline__ void fast_walk_single(const CallPair *__restrict__ recs, int n_batches,
                                                        __amdgpu_buffer_rsrc_t rsrc, unsigned lane_off, int lane, float &prod,
                                                        FastAcc &facc, unsigned *ring, OnGroup on_group)
{
    // The records of a batch reach every lane through LDS: the batch's 32 dwords (one per lane, loaded batches ahead)
    // are written to one of four 128-byte slots of the wavefront and read back as broadcast reads - row offsets when the
    // gathers are issued, keep / floor when the batch is consumed.  [Until round 4 the fields were moved to SGPRs with 24
    // v_readlane per 8 calls, ~9 issue cycles each: 0.9 ms of the kernel's 1.45 - it was bound by those, not by the
    // gathers: a variant with HALF the gathers per call (below) took just as long.]
    const bool hi = lane >= 32;
    if (n_batches <= 0) return;
    const unsigned *__restrict__ words = (const unsigned *)recs;
    const int l32 = lane & 31;
    auto fetch = [&](int k) {  // dword (lane % 32) of batch k; batches past the end re-read the last one
        const int kk = k < n_batches ? k : n_batches - 1;
        return (int)__builtin_nontemporal_load(&words[(size_t)kk * 32 + l32]);  // read once: keep the table rows in L2
    };
    auto issue = [&](int w, FastBatch &g, int slot) {  // records -> LDS slot; row offsets back -> the batch's gathers in flight
        unsigned *rec = ring + slot * 32;
        rec[l32] = (unsigned)w;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint2 ro = *(const uint2 *)(rec + 8 * q);
            if constexpr (HALF) {
                g.p[q].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(lane_off + (hi ? ro.y : ro.x)), 0, DMX_ROW_AUX));
            } else {
                g.p[q].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(lane_off + ro.x), 0, DMX_ROW_AUX));
                g.p[q].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(lane_off + ro.y), 0, DMX_ROW_AUX));
            }
        }
    };
    auto consume = [&](int k, const FastBatch &g, int slot) {
        const unsigned *rec = ring + slot * 32;
        on_group(k, __builtin_amdgcn_readfirstlane((int)rec[6]));
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const npm::f32x2 keep = *(const npm::f32x2 *)(rec + 8 * q + 2);
            const npm::f32x2 flo = *(const npm::f32x2 *)(rec + 8 * q + 4);
            if constexpr (HALF) {
                float t = g.p[q].x * (hi ? keep.y : keep.x);
                t = t + (hi ? flo.y : flo.x);
                prod = prod * t;
            } else {
                npm::f32x2 t = g.p[q] * keep;
                t = t + flo;
                prod = (prod * t.x) * t.y;
            }
        }
        const float m = __builtin_amdgcn_frexp_mantf(prod);
        facc.expo += __builtin_amdgcn_frexp_expf(prod);
        facc.mant += (double)__builtin_amdgcn_logf(m);
        prod = 1.0f;
    };
#if DMX_GATHER_DEPTH != 4
    {   // the same pipeline with a ring of DMX_GATHER_DEPTH batches (experiment builds; `ring` holds that many record slots)
        constexpr int D = DMX_GATHER_DEPTH;
        int w[D];
        FastBatch g[D];
#pragma unroll
        for (int j = 0; j < D; j++) w[j] = fetch(j);
#pragma unroll
        for (int j = 0; j < D - 1; j++) {
            issue(w[j], g[j], j);
            w[j] = fetch(D + j);
        }
        for (int k = 0; k < n_batches; k += D) {
            bool done = false;
#pragma unroll
            for (int u = 0; u < D; u++) {
                if (done) continue;
                constexpr int dummy = 0;
                (void)dummy;
                const int slot = (u + D - 1) % D;
                issue(w[slot], g[slot], slot);
                w[slot] = fetch(k + u + 2 * D - 1);
                consume(k + u, g[u], u);
                done = k + u + 1 >= n_batches;
            }
        }
        return;
    }
#endif
    // records of batches k+4 .. k+7 on their way, gathers of batches k+1 .. k+3 in flight while batch k is consumed;
    // the loop is unrolled by 4 so that every ring slot is a fixed register (and a fixed LDS slot)
    int w0 = fetch(0), w1 = fetch(1), w2 = fetch(2), w3 = fetch(3);
    FastBatch g0, g1, g2, g3;
    issue(w0, g0, 0);
    w0 = fetch(4);
    issue(w1, g1, 1);
    w1 = fetch(5);
    issue(w2, g2, 2);
    w2 = fetch(6);
    // ONE exit, the remainder peeled: with a `break` after every step the compiler unifies the exits into a block that also carries the
    // back edge, the wait-count analysis sees the loop's head reached from states in which a register's load was the last one issued,
    // and puts s_waitcnt vmcnt(0) there - the pipeline drained once per trip of 4 batches (until round 5).
    int k = 0;
    for (; k + 4 <= n_batches; k += 4) {
        issue(w3, g3, 3);
        w3 = fetch(k + 7);
        consume(k, g0, 0);
        issue(w0, g0, 0);
        w0 = fetch(k + 8);
        consume(k + 1, g1, 1);
        issue(w1, g1, 1);
        w1 = fetch(k + 9);
        consume(k + 2, g2, 2);
        issue(w2, g2, 2);
        w2 = fetch(k + 10);
        consume(k + 3, g3, 3);
    }
    const int rem = n_batches - k;
    if (rem > 0) {  // (the gathers of the last three batches are in flight)
        consume(k, g0, 0);
        if (rem > 1) {
            consume(k + 1, g1, 1);
            if (rem > 2) consume(k + 2, g2, 2);
        }
    }
}

template <int L, int A, bool PAIRS, int U, bool FAST>
__global__ __launch_bounds__(256) void k_estep_direct(EstepArgs a)
{
    static_assert(A == 1 || L == 64, "several accumulators per lane only with 64 lanes per call");
    static_assert(L % U == 0 && U % 2 == 0 && U <= 8, "rows are padded to 8 calls; chunks must tile them");
    constexpr int CPW = 64 / L;
    const int lane = threadIdx.x & 63;
    const int li = lane % L;
    const int gbase = lane - li;
    const int K = a.K;

    unsigned o1[A], o2[A];  // byte offsets of this lane's genotype(s) inside a prob row
    int kk[A];
    bool valid[A];
#pragma unroll
    for (int s = 0; s < A; s++) {
        const int k = li + 64 * s;
        valid[s] = k < K;
        kk[s] = valid[s] ? k : K - 1;
        if (PAIRS) {
            const unsigned pr = a.opt_pairs[kk[s]];
            o1[s] = (pr & 0xFFFFu) * 4u;
            o2[s] = (pr >> 16) * 4u;
        } else {
            o1[s] = (unsigned)kk[s] * 4u;
            o2[s] = o1[s];
        }
    }
    double acc[A];
    FastAcc facc[A];
    float prod[A];
#pragma unroll
    for (int s = 0; s < A; s++) {
        acc[s] = 0.0;
        facc[s].mant = 0.0;
        facc[s].expo = 0;
        prod[s] = 1.0f;
    }
    const char *__restrict__ prob = (const char *)a.prob;
    const int n_slots = (K + 63) >> 6;  // register slots that hold at least one option (A may be larger)

    long long b;
    bool live;
    int row_calls = 0;  // calls of this lane group's barcode (incl. padding)
    // Guarded E-step that runs direct (kernels.h: EstepArgs::direct): the fast kernels stand back, the exact launch walks
    // every barcode instead of the queue
    if (FAST && guard_stand_back(a)) return;
    // rows of `order` to walk: all B, or as many as the guarded E-step queued (known on the device only)
    const long long n_rows = a.order_count ? (long long)min((unsigned long long)*a.order_count, (unsigned long long)a.B) : a.B;
    long long seg_of_wave = -1;  // split rows (FAST, 64 lanes): the segment this wavefront walks
    if constexpr (L == 64) {
        // ---- wave-uniform path: everything about the row lives in SGPRs ----
        const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        long long slot = (long long)blockIdx.x * 4 + wave;
        live = true;
        long long pbeg;
        int npairs;
        long long seg = -1;  // >= 0: this wavefront walks one segment of a split barcode and leaves its sums
        if (FAST && a.segs != nullptr) {
            // work list: the segments of the n_split longest barcodes, then the other barcodes whole
            if (slot < a.n_segs) {
                seg = slot;
                const EstepSegment sg = a.segs[slot];
                b = sg.barcode;
                pbeg = a.pair_ptr[b] + sg.first_pair;
                npairs = sg.n_pairs;
            } else {
                slot = slot - a.n_segs + a.n_split;
                if (slot >= n_rows) return;
                b = a.order[slot];
                pbeg = a.pair_ptr[b];
                npairs = (int)(a.pair_ptr[b + 1] - pbeg);
            }
        } else {
            if (slot >= n_rows) return;
            b = a.order[slot];
            pbeg = a.pair_ptr[b];
            npairs = (int)(a.pair_ptr[b + 1] - pbeg);
        }
        row_calls = 2 * npairs;
        seg_of_wave = seg;
        const CallPair *__restrict__ recs = a.pairs + pbeg;
        // buffer addressing: descriptor base = prob table, voffset = this lane's genotype (VGPR, fixed),
        // soffset = the call's row offset straight from the scalar load -> no VALU work per load
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc((void *)a.prob, 0, (int)a.prob_bytes, 0x00020000);
        constexpr int H = U / 2;
        if constexpr (FAST) {
            // The tolerance mode has little arithmetic per call, so the row gathers' latency is what it waits for:
            // two batches in flight (ping-pong), the gathers of batch i+1 issued before batch i is consumed.  Reads past
            // the row's last batch are redirected to it (valid records, results unused).
            if constexpr (PAIRS && A <= 8) {
                // rows loaded whole, the options' two genotypes picked out of the wavefront (load_pair_rows)
                constexpr int NL = 2 * H / PairRowShape<A>::RPL;
                // two batches in flight, as below.  PMC (scripts/pmc_doublet_rows.sh, 20k x 20k x 8 with doublets, 0.175 ms):
                // VALU 50 % busy, LDS array 65 % (4 cycles per pick), 6.6 wavefronts per SIMD.  Tried: four batches in flight -
                // no change; no scalar memory in the loop (picks and scalar loads share lgkmcnt, and scalar loads return out
                // of order): records as one dword per lane, keep / floor out of it by v_readlane, stages spread over four
                // iterations with the picks of the next batch issued before the arithmetic of this one - 0.196 ms, the 16
                // v_readlane per 8 calls cost more than the waits they remove.
                if (npairs > 0) {
                    PairRowBatch<H, NL> x, y;
                    load_pair_rows<A, H, NL>(x, recs, 0, rsrc, lane);
                    for (int j0 = 0; j0 < npairs; j0 += 2 * H) {
                        const int j1 = j0 + H;
                        load_pair_rows<A, H, NL>(y, recs, min(j1, npairs - H), rsrc, lane);
                        pair_row_products<A, H, NL>(x, o1, o2, prod, n_slots);
                        if (((j0 + H) & 3) == 0) estep_flush<A>(prod, facc, n_slots);  // every 8 calls (rows are padded to 8)
                        if (j1 >= npairs) break;
                        load_pair_rows<A, H, NL>(x, recs, min(j1 + H, npairs - H), rsrc, lane);
                        pair_row_products<A, H, NL>(y, o1, o2, prod, n_slots);
                        if (((j1 + H) & 3) == 0) estep_flush<A>(prod, facc, n_slots);
                    }
                }
            } else
            if (npairs > 0) {
                RecBatch<A, H, PAIRS> x, y;
                load_batch<A, H, PAIRS>(x, recs, 0, rsrc, o1, o2, n_slots);
                for (int j0 = 0; j0 < npairs; j0 += 2 * H) {
                    const int j1 = j0 + H;
                    load_batch<A, H, PAIRS>(y, recs, min(j1, npairs - H), rsrc, o1, o2, n_slots);
                    estep_products<A, PAIRS, H>(x.p1, x.p2, x.keep, x.flo, prod, n_slots);
                    if (((j0 + H) & 3) == 0) estep_flush<A>(prod, facc, n_slots);  // every 8 calls (rows are padded to 8)
                    if (j1 >= npairs) break;
                    load_batch<A, H, PAIRS>(x, recs, min(j1 + H, npairs - H), rsrc, o1, o2, n_slots);
                    estep_products<A, PAIRS, H>(y.p1, y.p2, y.keep, y.flo, prod, n_slots);
                    if (((j1 + H) & 3) == 0) estep_flush<A>(prod, facc, n_slots);
                }
            }
        } else {
            for (int j0 = 0; j0 < npairs; j0 += H) {
                RecBatch<A, H, PAIRS> x;
                load_batch<A, H, PAIRS>(x, recs, j0, rsrc, o1, o2, n_slots);
                estep_terms<A, PAIRS, H>(x.p1, x.p2, x.keep, x.flo, acc, n_slots);
            }
        }
    } else {
        if (((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * CPW >= n_rows) return;  // the whole wavefront past the list
        const long long slot = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * CPW + lane / L;
        live = slot < n_rows;
        b = a.order[live ? slot : n_rows - 1];
        const long long pbeg = a.pair_ptr[b];
        const int n = live ? 2 * (int)(a.pair_ptr[b + 1] - pbeg) : 0;  // calls incl. padding, multiple of 8
        row_calls = n;
        const unsigned *__restrict__ words = (const unsigned *)(a.pairs + pbeg);
        const int nmax = group_max_over_wave<L>(n);
        if constexpr (FAST) {
            // Tolerance arithmetic: a few instructions per term, so what the loop waits for is memory.  Two stages in
            // flight: the record fields of the NEXT chunk of L calls (one call per lane), and the row gathers of the NEXT
            // batch of U calls, both requested before the current batch is consumed (200k x 100k x 32: 1.30 -> see
            // DESIGN.md 4.1).  Batches alternate between two fixed register sets (compile-time unrolled).
            constexpr int H = U / 2;
            struct Batch {
                npm::f32x2 p1[H][A], p2[H][A], keep[H], flo[H];
            };
            auto load_fields = [&](int c0, unsigned &ro, float &kp, float &fl) {
                const int ci = c0 + li;
                const bool mine = ci < n;
                const int w = mine ? (ci >> 1) * 8 + (ci & 1) : 0;
                ro = 0u;
                kp = 0.0f;
                fl = 1.0f;
                if (mine) {
                    ro = words[w];
                    kp = __uint_as_float(words[w + 2]);
                    fl = __uint_as_float(words[w + 4]);
                }
            };
            auto issue = [&](Batch &x, unsigned ro_v, float keep_v, float floor_v, int i0) {
#pragma unroll
                for (int q = 0; q < H; q++) {
                    const unsigned ro0 = group_bcast<L>(ro_v, i0 + 2 * q, gbase);
                    const unsigned ro1 = group_bcast<L>(ro_v, i0 + 2 * q + 1, gbase);
                    x.keep[q].x = group_bcast<L>(keep_v, i0 + 2 * q, gbase);
                    x.keep[q].y = group_bcast<L>(keep_v, i0 + 2 * q + 1, gbase);
                    x.flo[q].x = group_bcast<L>(floor_v, i0 + 2 * q, gbase);
                    x.flo[q].y = group_bcast<L>(floor_v, i0 + 2 * q + 1, gbase);
#pragma unroll
                    for (int s = 0; s < A; s++) {
                        x.p1[q][s].x = *(const float *)(prob + (ro0 + o1[s]));
                        x.p1[q][s].y = *(const float *)(prob + (ro1 + o1[s]));
                        if (PAIRS) {
                            x.p2[q][s].x = *(const float *)(prob + (ro0 + o2[s]));
                            x.p2[q][s].y = *(const float *)(prob + (ro1 + o2[s]));
                        }
                    }
                }
            };
            auto consume = [&](const Batch &x, int pos) {
                estep_products<A, PAIRS, H>(x.p1, x.p2, x.keep, x.flo, prod, n_slots);
                if (((pos + U) & 7) == 0) estep_flush<A>(prod, facc, n_slots);
            };
            unsigned ro_c, ro_n = 0u;
            float keep_c, keep_n = 0.0f, floor_c, floor_n = 1.0f;
            load_fields(0, ro_c, keep_c, floor_c);
            for (int c0 = 0; c0 < nmax; c0 += L) {
                if (c0 + L < nmax) load_fields(c0 + L, ro_n, keep_n, floor_n);
                const int cnt = (nmax - c0) < L ? (nmax - c0) : L;  // a multiple of U (rows are padded to 8 calls)
                Batch x, y;
                issue(x, ro_c, keep_c, floor_c, 0);
#pragma unroll
                for (int i0 = 0; i0 < L; i0 += 2 * U) {
                    if (i0 >= cnt) break;
                    if (i0 + U < cnt) issue(y, ro_c, keep_c, floor_c, i0 + U);
                    consume(x, c0 + i0);
                    if (i0 + U >= cnt) break;
                    if (i0 + 2 * U < cnt) issue(x, ro_c, keep_c, floor_c, i0 + 2 * U);
                    consume(y, c0 + i0 + U);
                }
                ro_c = ro_n;
                keep_c = keep_n;
                floor_c = floor_n;
            }
        } else
        for (int c0 = 0; c0 < nmax; c0 += L) {
            // fields of call c0 + li of this group's row; groups that are done feed neutral calls
            const int ci = c0 + li;
            const bool mine = ci < n;
            const int w = mine ? (ci >> 1) * 8 + (ci & 1) : 0;
            unsigned ro_v = 0u;
            float keep_v = 0.0f, floor_v = 1.0f;
            if (mine) {
                ro_v = words[w];
                keep_v = __uint_as_float(words[w + 2]);
                floor_v = __uint_as_float(words[w + 4]);
            }
            const int cnt = (nmax - c0) < L ? (nmax - c0) : L;
            for (int i0 = 0; i0 < cnt; i0 += U) {
                constexpr int H = U / 2;
                npm::f32x2 p1[H][A], p2[H][A], keep[H], flo[H];
#pragma unroll
                for (int q = 0; q < H; q++) {
                    const unsigned ro0 = group_bcast<L>(ro_v, i0 + 2 * q, gbase);
                    const unsigned ro1 = group_bcast<L>(ro_v, i0 + 2 * q + 1, gbase);
                    keep[q].x = group_bcast<L>(keep_v, i0 + 2 * q, gbase);
                    keep[q].y = group_bcast<L>(keep_v, i0 + 2 * q + 1, gbase);
                    flo[q].x = group_bcast<L>(floor_v, i0 + 2 * q, gbase);
                    flo[q].y = group_bcast<L>(floor_v, i0 + 2 * q + 1, gbase);
#pragma unroll
                    for (int s = 0; s < A; s++) {
                        p1[q][s].x = *(const float *)(prob + (ro0 + o1[s]));
                        p1[q][s].y = *(const float *)(prob + (ro1 + o1[s]));
                        if (PAIRS) {
                            p2[q][s].x = *(const float *)(prob + (ro0 + o2[s]));
                            p2[q][s].y = *(const float *)(prob + (ro1 + o2[s]));
                        }
                    }
                }
                if constexpr (FAST) {
                    estep_products<A, PAIRS, H>(p1, p2, keep, flo, prod, n_slots);
                    if (((c0 + i0 + U) & 7) == 0) estep_flush<A>(prod, facc, n_slots);
                } else {
                    estep_terms<A, PAIRS, H>(p1, p2, keep, flo, acc, n_slots);
                }
            }
        }
    }
    if constexpr (FAST) {
        if constexpr (L == 64) {
            if (seg_of_wave >= 0) {  // a segment of a split barcode: its sums, for k_estep_join
#pragma unroll
                for (int s = 0; s < A; s++)
                    if (valid[s]) a.seg_sums[(size_t)seg_of_wave * K + kk[s]] = facc[s].mant + (double)facc[s].expo;
                return;
            }
        }
        const double LN2 = 0.693147180559945309417232121458176568;
#pragma unroll
        for (int s = 0; s < A; s++) acc[s] = (facc[s].mant + (double)facc[s].expo) * LN2;
    }

    estep_epilogue<L, A, FAST>(a, b, live, acc, kk, valid, lane, li, gbase, row_calls);
}

// Tolerance / guarded mode, split rows: one wavefront per split barcode adds the sums its segments left (in segment
// order: a fixed association) and runs the epilogue of the lane-per-option forms.
template <int A>
__global__ __launch_bounds__(256) void k_estep_join(EstepArgs a)
{
    const int lane = threadIdx.x & 63;
    const long long j = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= a.n_split) return;
    if (a.guard && a.direct != nullptr && *a.direct != 0u) return;  // the E-step runs direct: the exact launch does every barcode
    const long long b = a.order[j];
    const int K = a.K;
    int kk[A];
    bool valid[A];
    double acc[A];
#pragma unroll
    for (int s = 0; s < A; s++) {
        const int k = lane + 64 * s;
        valid[s] = k < K;
        kk[s] = valid[s] ? k : K - 1;
        acc[s] = 0.0;
    }
    for (int sg = a.split_first[j]; sg < a.split_first[j + 1]; sg++)
#pragma unroll
        for (int s = 0; s < A; s++) acc[s] += a.seg_sums[(size_t)sg * K + kk[s]];
#pragma unroll
    for (int s = 0; s < A; s++) acc[s] *= 0.693147180559945309417232121458176568;
    estep_epilogue<64, A, true>(a, b, true, acc, kk, valid, lane, lane, 0, 2 * (int)(a.pair_ptr[b + 1] - a.pair_ptr[b]));
}

// ------------------------------------------------------------------------------------
// E-step, tile-major form (singlets, 33 <= K <= 128, many barcodes, genotype table larger than an XCD's L2).
// The direct form gathers one genotype row (K * 4 bytes) per call from a table of V rows that no L2 holds (51 MB at
// 200k x 64; an XCD's L2 is 4 MB): its 8192 resident wavefronts are at 8192 unrelated places of the variant axis, so
// more than half of the row reads miss L2 and cross the fabric (measured: 7 GB per launch for 0.8 GB of algorithmic
// bytes).  Here the variant axis is cut into TILES of ~1 MB of table (kernels.h: TILE_BYTES), and every wavefront owns a BIN of up to
// TILE_R_MAX barcodes which it walks tile by tile: the calls of barcode 0 that fall into tile 0, of barcode 1 in tile
// 0, ... then tile 1, and so on (rows are variant-sorted, so that is a sequential walk of every row, resumed once per
// tile).  The bins hold equal numbers of calls and there are (rounds x resident wavefronts) of them, all wavefronts
// run the same instruction stream at the same rate, so at any moment the resident wavefronts of an XCD are working in
// the same one or two tiles, which its L2 holds.  No synchronisation is involved: the alignment is statistical and
// only affects speed.
// The repack materialises that walk: tile_stream holds the call records of a bin in exactly the order the wavefront
// consumes them (so the kernel reads one contiguous record stream, like the direct form), each 8-call group tagged
// with the barcode slot it belongs to.  The float64 accumulator of the current slot lives in registers; on a slot
// change it is swapped with the slot's copy in LDS.  Every barcode's calls are still added strictly in order, so
// the sums are bit-identical to the direct form's.
// ------------------------------------------------------------------------------------
#ifndef DMX_TILED_WAVES
#define DMX_TILED_WAVES 0  // experiment builds: wavefronts per SIMD the register allocation of k_estep_tiled aims at (0: the compiler's choice, 67 VGPRs = 7)
#endif
template <int A, bool FAST, bool HALF = false>
__global__ __launch_bounds__(256)
#if DMX_TILED_WAVES
__attribute__((amdgpu_waves_per_eu(DMX_TILED_WAVES, DMX_TILED_WAVES)))
#endif
void k_estep_tiled(EstepArgs a)
{
    static_assert(!HALF || (A == 1 && FAST), "two calls per gather: tolerance arithmetic, tables of at most 32 genotypes");
    __shared__ double sh_acc[4][TILE_R_MAX][A][64];
    __shared__ unsigned sh_rec[4][DMX_GATHER_DEPTH * 32];  // fast_walk_single: four batches of records per wavefront
    const int lane = threadIdx.x & 63;
    const int K = a.K;
    const int R = a.bin_rows_cap;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long slot_id = (long long)blockIdx.x * 4 + wave;
    if (FAST && guard_stand_back(a)) return;  // the E-step runs direct (EstepArgs::direct)
    if (slot_id >= a.n_bins) return;
    const long long bin = a.bin_order[slot_id];

    unsigned o1[A];
    int kk[A];
    bool valid[A];
#pragma unroll
    for (int s = 0; s < A; s++) {
        const int k = lane + 64 * s;
        valid[s] = k < K;
        kk[s] = valid[s] ? k : K - 1;
        o1[s] = (unsigned)kk[s] * 4u;
        if (HALF) o1[s] = (unsigned)min(lane & 31, K - 1) * 4u;  // both halves of the wavefront: genotype lane % 32
    }
    const int n_slots = (K + 63) >> 6;
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int s = 0; s < A; s++) sh_acc[wave][r][s][lane] = 0.0;

    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.prob, 0, (int)a.prob_bytes, 0x00020000);
    const long long g0 = a.bin_ptr[bin];
    const int n_pairs = (int)(a.bin_ptr[bin + 1] - g0) * 4;
    const CallPair *__restrict__ recs = a.tile_stream + g0 * 4;
    double acc[A];
    FastAcc facc[A];
    float prod[A];
#pragma unroll
    for (int s = 0; s < A; s++) {
        acc[s] = 0.0;
        facc[s].mant = 0.0;
        facc[s].expo = 0;
        prod[s] = 1.0f;
    }
    int cur = 0;  // slot whose accumulator is in registers
    constexpr int H = A == 1 ? 4 : 2;  // pairs per load batch; a group = 4 pairs = 8 calls
    const unsigned zero[A] = {};
    // a batch starting a group (j % 4 == 0) carries the group's slot tag; on a slot change the accumulators are
    // swapped through LDS (wave-uniform branch)
    auto enter = [&](int j, int tag) {
        if ((j & 3) != 0 || tag == cur) return;
#pragma unroll
        for (int s = 0; s < A; s++) {
            if (FAST) acc[s] += facc[s].mant + (double)facc[s].expo;  // the parked value is in log2 units
            sh_acc[wave][cur][s][lane] = acc[s];
            acc[s] = sh_acc[wave][tag][s][lane];
            facc[s].mant = 0.0;
            facc[s].expo = 0;
        }
        cur = tag;
    };
    if constexpr (FAST && A == 1) {
        fast_walk_single<HALF>(recs, n_pairs >> 2, rsrc, o1[0], lane, prod[0], facc[0], sh_rec[wave], [&](int, int tag) { enter(0, tag); });
    } else if constexpr (FAST) {
        // two batches in flight (see k_estep_direct); reads past the bin's last batch are redirected to it
        if (n_pairs > 0) {
            RecBatch<A, H, false> x, y;
            load_batch<A, H, false>(x, recs, 0, rsrc, o1, zero, n_slots);
            int tag_x = (int)recs[0].reserved[0], tag_y = 0;
            for (int j0 = 0; j0 < n_pairs; j0 += 2 * H) {
                const int j1 = j0 + H, jy = min(j1, n_pairs - H);
                load_batch<A, H, false>(y, recs, jy, rsrc, o1, zero, n_slots);
                tag_y = (int)recs[jy].reserved[0];
                enter(j0, tag_x);
                estep_products<A, false, H>(x.p1, x.p2, x.keep, x.flo, prod, n_slots);
                if (((j0 + H) & 3) == 0) estep_flush<A>(prod, facc, n_slots);
                if (j1 >= n_pairs) break;
                const int jx = min(j1 + H, n_pairs - H);
                load_batch<A, H, false>(x, recs, jx, rsrc, o1, zero, n_slots);
                tag_x = (int)recs[jx].reserved[0];
                enter(j1, tag_y);
                estep_products<A, false, H>(y.p1, y.p2, y.keep, y.flo, prod, n_slots);
                if (((j1 + H) & 3) == 0) estep_flush<A>(prod, facc, n_slots);
            }
        }
    } else {
        for (int j0 = 0; j0 < n_pairs; j0 += H) {
            RecBatch<A, H, false> x;
            load_batch<A, H, false>(x, recs, j0, rsrc, o1, zero, n_slots);
            enter(j0, (int)recs[j0].reserved[0]);
            estep_terms<A, false, H>(x.p1, x.p2, x.keep, x.flo, acc, n_slots);
        }
    }
#pragma unroll
    for (int s = 0; s < A; s++) {
        if (FAST) acc[s] += facc[s].mant + (double)facc[s].expo;
        sh_acc[wave][cur][s][lane] = acc[s];
    }
    for (int r = 0; r < R; r++) {
        const int row = a.bin_rows[bin * R + r];
        if (row < 0) continue;
        double out[A];
#pragma unroll
        for (int s = 0; s < A; s++) {
            out[s] = sh_acc[wave][r][s][lane];
            if (HALF) out[s] = out[s] + shfl_xor_f64(out[s], 32);  // even calls (lanes 0 .. 31) + odd calls (lanes 32 .. 63)
            if (FAST) out[s] *= 0.693147180559945309417232121458176568;
        }
        estep_epilogue<64, A, FAST>(a, (long long)row, true, out, kk, valid, lane, lane, 0,
                              2 * (int)(a.pair_ptr[row + 1] - a.pair_ptr[row]));
    }
}

// ------------------------------------------------------------------------------------
// E-step, COARSE pass of the tile-major form (guarded mode, singlets, 33 .. 64 genotypes; EM iterations whose logits nobody reads).
// k_estep_tiled<1, true> sits on the L1's data path: every call gathers a 256-byte row that misses the L1 (8 clocks of fill + read per
// call, 6.9 measured: DESIGN.md 4.1), with the LDS (its records' broadcast reads, 6.5 clocks per call) and the VALU (4.6) close behind.
// What the guard needs of a fast pass is only a PROVABLE bound D on the logits' deviation: a barcode whose second posterior is below
// 1e-6 keeps its result under D = 0.2 just as under D = 1.5e-4, and on a workload of separable donors that is 99.6 % of the barcodes.
// So this pass reads the genotype table as binary16 - 128-byte rows, one L1 line per call, relative error 2^-11 per term, priced per
// call by the guard (estep_epilogue.h; EstepArgs::guard_per_call) - and takes everything else off the paths that were next in line:
//   records   a batch's 32 dwords are loaded as 8 bytes per lane by EACH row of 16 lanes (lane i of a row: dwords 2i, 2i + 1) and reach
//             the arithmetic as DPP row broadcasts inside the consuming instructions (v_mul_f32_dpp .. row_newbcast) - no LDS, no readlane;
//   sums      the log2 of a batch's 8-term product (v_log_f32 of the product itself: its exponent is part of the result) is added in
//             float32 (EstepArgs::guard_accum), parked in LDS as float32.
// The barcodes it cannot prove go to the exact redo like those of every guarded pass.
// ------------------------------------------------------------------------------------
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

static __device__ __forceinline__ unsigned row_bcast(unsigned v, int n)  // lane n of the caller's row of 16 lanes (n: compile-time after unrolling)
{
#define DMX_BC(N) case N: return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x150 + N, 0xf, 0xf, true);  // (bound_ctrl: no `old` value to prepare)
    switch (n) {
        DMX_BC(0) DMX_BC(1) DMX_BC(2) DMX_BC(3) DMX_BC(4) DMX_BC(5) DMX_BC(6) DMX_BC(7)
        DMX_BC(8) DMX_BC(9) DMX_BC(10) DMX_BC(11) DMX_BC(12) DMX_BC(13) DMX_BC(14) DMX_BC(15)
    }
#undef DMX_BC
    return v;
}

struct CoarseSum {
    float lo, hi;  // sums of log2 for genotypes 2 i and 2 i + 1 (i = lane % lanes per call) over the calls of this lane's block
};

template <typename F, int... I>
static __device__ __forceinline__ void for_each_int(std::integer_sequence<int, I...>, F f)  // f(integral_constant<int, 0>) ... in order
{
    (f(std::integral_constant<int, I>{}), ...);
}

// Shape of the coarse pass for CPG calls per 256-byte gather: 1 (65 .. 128 genotypes: a call's binary16 row is the whole gather), 2 (33 ..
// 64 genotypes: 128 bytes, 32 lanes) or 4 (17 .. 32 genotypes: 64 bytes, 16 lanes).  Lanes of BLOCK j = lane / (64 / CPG) take call
// q CPG + j of a group's gather q.
template <int CPG>
struct CoarseShape {
    static constexpr int LPC = 64 / CPG;  // lanes per call
    static constexpr int GPB = 8 / CPG;   // gathers per batch (= one group of 8 calls of one barcode)
    static constexpr int BPR = 8 / GPB;   // batches per record: a block's record is 16 dwords = BPR x (GPB offsets, GPB r)
    static constexpr int DD = 4;          // records in registers
    static constexpr int T = DD * BPR;    // batches per trip of the unrolled loop
    static constexpr int GA = CPG == 4 ? 7 : 3;  // batches whose gathers are in flight while one is consumed (24 / 12 / 14 gathers)
    static constexpr int DG = GA + 1;
    static_assert(T % DG == 0, "ring slots must be fixed registers");
};

// The coarse pass's own record stream (launch_build_coarse_stream): per call a row offset and r = floor / keep -
//     term = p keep + floor = keep (p + r):   log2 of a barcode's product = sum_c log2(p_c + r_c) + sum_c log2 keep_c,
// the second sum the same for every option (EstepArgs::log2_keep, added in the epilogue): ONE addition per term instead of a
// multiplication and an addition, 8 bytes per call instead of 16.  A RECORD is CPG blocks of 16 dwords and covers BPR batches; block j,
// batch s of the record:  dwords s 2 GPB + q: row offset of call q CPG + j,  s 2 GPB + GPB + q: its r  (CPG = 2: the even calls' block
// [A: off0 off2 off4 off6 r0 r2 r4 r6 | B: ...] and the odd calls').  Lane i of a row of 16 lanes loads dword i of its block: one
// dword per lane for BPR batches, and every field reaches the arithmetic as a DPP row broadcast.  The slot tag of a batch sits in the
// low 4 bits of its r0 (2^-19 of r: priced).  Padding calls - and calls with keep = 0 - gather the all-zero row behind the table with
// r = floor: p + r = floor exactly.  Every bin starts at a record (coarse_bin_ptr).
// F32 (k_estep_tiled_fine8): the same walk on the float32 table - a gather is 8 bytes per lane, the sums p + r are plain additions, and a
// batch's product goes into a float64 sum through its mantissa and exponent (FineSum), as the fine pass on the tile-major stream does.
struct FineSum {
    double lo, hi;   // sums of log2 of the products' mantissas
    int elo, ehi;    // ... and of their exponents (exact)
};
template <int CPG, bool F32, typename Sum, typename OnGroup>
static __device__ __forceinline__ void coarse_walk(const unsigned *__restrict__ stream, int n_batches, __amdgpu_buffer_rsrc_t rsrc,
                                                   unsigned lane_off, int lane, Sum &lacc, OnGroup on_group)
{
    using S = CoarseShape<CPG>;
    constexpr int GPB = S::GPB, BPR = S::BPR, DD = S::DD, T = S::T, GA = S::GA, DG = S::DG;
    if (n_batches <= 0) return;
    const int n_rec = (n_batches + BPR - 1) / BPR;
    const unsigned *__restrict__ words = stream + ((lane / S::LPC) * 16 + (lane & 15));
    auto fetch = [&](int d) {  // (records past the end re-read the last one)
        const int dc = d < n_rec ? d : n_rec - 1;
        return __builtin_nontemporal_load(&words[(size_t)dc * (CPG * 16)]);
    };
    struct Gathers {
        std::conditional_t<F32, u32x2_t, unsigned> h[GPB];  // gather q: probabilities of genotypes 2 i, 2 i + 1 of this block's call of gather q (binary16 / float32)
    };
    auto issue = [&](unsigned w, auto sel, Gathers &g) {
        constexpr int B0 = decltype(sel)::value * 2 * GPB;
#pragma unroll
        for (int q = 0; q < GPB; q++) {
            if constexpr (F32) g.h[q] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(row_bcast(w, B0 + q) + lane_off), 0, 0);
            else g.h[q] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(row_bcast(w, B0 + q) + lane_off), 0, 0);
        }
    };
    auto consume = [&](int k, unsigned w, auto sel, const Gathers &g) {
        constexpr int R0 = decltype(sel)::value * 2 * GPB + GPB;
        on_group(k, __builtin_amdgcn_readlane((int)w, R0) & 15);
        // p + r of the lane's two genotypes: the binary16 probability goes into ONE v_fma_mix_f32 each (p x 1.0 + r, p converted inside
        // the instruction: the float32 rounding of v_cvt_f32_f16 + v_add_f32, bit for bit), r as one DPP row broadcast per call - three
        // instructions per call and lane where conversion, conversion, addition, addition were four (round 6: the pass is VALU-bound)
        auto sums = [&](auto q, float &s_lo, float &s_hi) {
            constexpr int Q = decltype(q)::value;
            const float r = __uint_as_float(row_bcast(w, R0 + Q));
            if constexpr (F32) {
                s_lo = __uint_as_float(g.h[Q].x) + r;
                s_hi = __uint_as_float(g.h[Q].y) + r;
            } else {
                const unsigned h = g.h[Q];  // (named here: an asm operand does not capture for the lambda)
                asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(s_lo) : "v"(h), "v"(r));
                asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(s_hi) : "v"(h), "v"(r));
            }
        };
        float prod_lo, prod_hi;
        sums(std::integral_constant<int, 0>{}, prod_lo, prod_hi);
        for_each_int(std::make_integer_sequence<int, GPB - 1>{}, [&](auto qm1) {  // (a product of GPB sums: at least 1e-32 for GPB = 8; beyond
            float s_lo, s_hi;                                                       // float32 only with several keep factors near 2^-24 - NaN, queued)
            sums(std::integral_constant<int, decltype(qm1)::value + 1>{}, s_lo, s_hi);
            prod_lo = prod_lo * s_lo;
            prod_hi = prod_hi * s_hi;
        });
        if constexpr (F32) {  // (k_estep_tiled: the mantissa's log2 is within 2 ulp of a value in [-1, 0), the exponent is exact)
            lacc.elo += __builtin_amdgcn_frexp_expf(prod_lo);
            lacc.ehi += __builtin_amdgcn_frexp_expf(prod_hi);
            lacc.lo += (double)__builtin_amdgcn_logf(__builtin_amdgcn_frexp_mantf(prod_lo));
            lacc.hi += (double)__builtin_amdgcn_logf(__builtin_amdgcn_frexp_mantf(prod_hi));
        } else {
            lacc.lo += __builtin_amdgcn_logf(prod_lo);  // v_log_f32 = log2 of a product of GPB sums p + r
            lacc.hi += __builtin_amdgcn_logf(prod_hi);
        }
    };
    unsigned w[DD];
    Gathers g[DG];
#pragma unroll
    for (int j = 0; j < DD; j++) w[j] = fetch(j);
    auto prologue = [&](auto u) { issue(w[(decltype(u)::value / BPR) % DD], std::integral_constant<int, decltype(u)::value % BPR>{}, g[decltype(u)::value % DG]); };
    // step U of a trip starting at batch k: the gathers of batch k + U + GA issued, batch k + U consumed; behind the last batch of a record
    // its register takes the record DD ahead
    auto step = [&](int k, auto u) {
        constexpr int U = decltype(u)::value;
        issue(w[((U + GA) / BPR) % DD], std::integral_constant<int, (U + GA) % BPR>{}, g[(U + GA) % DG]);
        consume(k + U, w[(U / BPR) % DD], std::integral_constant<int, U % BPR>{}, g[U % DG]);
        if constexpr (U % BPR == BPR - 1) w[(U / BPR) % DD] = fetch((k + U) / BPR + DD);
    };
    for_each_int(std::make_integer_sequence<int, GA>{}, prologue);
    // ONE exit and the remainder peeled: with a `break` after every step the compiler unifies the exits into a block that also carries
    // the back edge, the wait-count analysis then sees the loop's head reached from states in which a register's load was the last
    // one issued, and puts s_waitcnt vmcnt(0) there - the pipeline drained once per trip.
    int k = 0;
    for (; k + T <= n_batches; k += T)
        for_each_int(std::make_integer_sequence<int, T>{}, [&](auto u) { step(k, u); });
    const int rem = n_batches - k;  // (k is a multiple of T: the registers hold what a trip's start expects)
    auto tail = [&](auto self, auto u) -> void {
        constexpr int U = decltype(u)::value;
        if constexpr (U < T - 1) {
            if (rem > U) {
                step(k, u);
                self(self, std::integral_constant<int, U + 1>{});
            }
        }
    };
    tail(tail, std::integral_constant<int, 0>{});
}

template <int CPG>
__global__ __launch_bounds__(256) void k_estep_tiled_coarse(EstepArgs a)
{
    __shared__ CoarseSum sh_acc[4][TILE_R_MAX][64];
    const int lane = threadIdx.x & 63;
    const int K = a.K;
    const int R = a.bin_rows_cap;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long slot_id = (long long)blockIdx.x * 4 + wave;
    if (guard_stand_back(a)) return;  // another level runs (EstepArgs::direct)
    if (slot_id >= a.n_bins) return;
    const long long bin = a.bin_order[slot_id];
    constexpr int A = CPG == 1 ? 2 : 1;  // option slots per lane of the epilogue: option k = lane + 64 s
    int kk[A];
    bool valid[A];
#pragma unroll
    for (int s = 0; s < A; s++) {
        valid[s] = lane + 64 * s < K;
        kk[s] = valid[s] ? lane + 64 * s : K - 1;
    }
    for (int r = 0; r < R; r++) sh_acc[wave][r][lane] = CoarseSum{0.0f, 0.0f};
    // (the table's extent + the all-zero row behind it)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.prob16, 0, (int)(a.prob_bytes + (unsigned)a.G * 4u), 0x00020000);
    const int n_batches = (int)(a.bin_ptr[bin + 1] - a.bin_ptr[bin]);
    const unsigned *__restrict__ stream = a.coarse_stream + a.coarse_bin_ptr[bin] * (CPG * 16);
    CoarseSum lacc{0.0f, 0.0f};
    int cur = 0;  // slot whose sums are in registers, and where it parks them (kept as an address: one v_lshl_add per change of slot, not two)
    CoarseSum *cur_at = &sh_acc[wave][0][lane];
    // genotypes 2 i, 2 i + 1 of the lane's call: one dword of the binary16 row (a pair past G reads into the unused half of the row)
    coarse_walk<CPG, false>(stream, n_batches, rsrc, (unsigned)(lane & (CoarseShape<CPG>::LPC - 1)) * 4u, lane, lacc, [&](int, int tag) {
        if (tag == cur) return;  // (wave-uniform)
        *cur_at = lacc;
        cur_at = &sh_acc[wave][tag][lane];
        lacc = *cur_at;
        cur = tag;
    });
    *cur_at = lacc;
    for (int r = 0; r < R; r++) {
        const int row = a.bin_rows[bin * R + r];
        if (row < 0) continue;
        CoarseSum v = sh_acc[wave][r][lane];
#pragma unroll
        for (int off = 32; off >= CoarseShape<CPG>::LPC; off >>= 1) {  // the blocks' shares of the barcode's calls
            v.lo += __shfl_xor(v.lo, off);
            v.hi += __shfl_xor(v.hi, off);
        }
        double out[A];
        const double lk = a.log2_keep[row];
#pragma unroll
        for (int s = 0; s < A; s++) {  // option k = lane + 64 s: genotype k of lane k / 2
            const float lo = __shfl(v.lo, (lane >> 1) + 32 * s), hi = __shfl(v.hi, (lane >> 1) + 32 * s);
            out[s] = ((double)((lane & 1) ? hi : lo) + lk) * 0.693147180559945309417232121458176568;
        }
        // (the float32 partial sums are those of log2(p + r) <= -log2 keep + 1.5e-4: up to 3 |log2_keep| beyond the total's magnitude)
        estep_epilogue<64, A, true>(a, (long long)row, true, out, kk, valid, lane, lane, 0, 2 * (int)(a.pair_ptr[row + 1] - a.pair_ptr[row]),
                                    3.0f * 0.6931472f * fabsf((float)lk) * 1.000001f);
    }
}

// ------------------------------------------------------------------------------------
// E-step, FINE pass on the coarse pass's records (dmx_set_lean_memory: the tile-major stream of k_estep_tiled is gone).  The same walk
// of the same 8-byte records, the float32 table, float64 sums: a term is keep (p + r) with r = fl(floor / keep) and the slot tag in the
// low 4 bits of one r in GPB - against the reference's float32 term (two roundings) the sum p + r is off by at most 2^-24 (r) + 2^-24
// (its own rounding) + 2 x 2^-24 relative, one term in GPB by 2^-19 more (the tag), then the product's roundings and the mantissa's
// log as in k_estep_tiled (estep_epilogue.h: GUARD_PER_CALL): guard_per_call_fine8.  The sum of log2 keep (EstepArgs::log2_keep, v_log_f32
// results summed in float64) is the same for every option: it cancels in the posteriors the guard proves; a logit this pass serves
// carries its error, at most 2 ulp of |log2 keep| per call (8e-8 for keep in [0.5, 1]).
// ------------------------------------------------------------------------------------
struct FinePark {
    double lo, hi;
};
template <int CPG>
__global__ __launch_bounds__(256) void k_estep_tiled_fine8(EstepArgs a)
{
    __shared__ FinePark sh_acc[4][TILE_R_MAX][64];
    const int lane = threadIdx.x & 63;
    const int K = a.K;
    const int R = a.bin_rows_cap;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long slot_id = (long long)blockIdx.x * 4 + wave;
    if (guard_stand_back(a)) return;  // another level runs (EstepArgs::direct)
    if (slot_id >= a.n_bins) return;
    const long long bin = a.bin_order[slot_id];
    constexpr int A = CPG == 1 ? 2 : 1;
    int kk[A];
    bool valid[A];
#pragma unroll
    for (int s = 0; s < A; s++) {
        valid[s] = lane + 64 * s < K;
        kk[s] = valid[s] ? lane + 64 * s : K - 1;
    }
    for (int r = 0; r < R; r++) sh_acc[wave][r][lane] = FinePark{0.0, 0.0};
    // (the padding calls' offset is one row behind the table: beyond num_records, a buffer load returns 0 - p + r = r)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.prob, 0, (int)a.prob_bytes, 0x00020000);
    const int n_batches = (int)(a.bin_ptr[bin + 1] - a.bin_ptr[bin]);
    const unsigned *__restrict__ stream = a.coarse_stream + a.coarse_bin_ptr[bin] * (CPG * 16);
    FineSum lacc{0.0, 0.0, 0, 0};
    int cur = 0;
    FinePark *cur_at = &sh_acc[wave][0][lane];
    auto park = [&]() { *cur_at = FinePark{lacc.lo + (double)lacc.elo, lacc.hi + (double)lacc.ehi}; };
    coarse_walk<CPG, true>(stream, n_batches, rsrc, (unsigned)(lane & (CoarseShape<CPG>::LPC - 1)) * 8u, lane, lacc, [&](int, int tag) {
        if (tag == cur) return;  // (wave-uniform)
        park();
        cur_at = &sh_acc[wave][tag][lane];
        const FinePark v = *cur_at;
        lacc = FineSum{v.lo, v.hi, 0, 0};
        cur = tag;
    });
    park();
    for (int r = 0; r < R; r++) {
        const int row = a.bin_rows[bin * R + r];
        if (row < 0) continue;
        FinePark v = sh_acc[wave][r][lane];
#pragma unroll
        for (int off = 32; off >= CoarseShape<CPG>::LPC; off >>= 1) {  // the blocks' shares of the barcode's calls
            v.lo += shfl_xor_f64(v.lo, off);
            v.hi += shfl_xor_f64(v.hi, off);
        }
        double out[A];
        const double lk = a.log2_keep[row];
#pragma unroll
        for (int s = 0; s < A; s++) {  // option k = lane + 64 s: genotype k of lane k / 2
            const double lo = __shfl(v.lo, (lane >> 1) + 32 * s), hi = __shfl(v.hi, (lane >> 1) + 32 * s);
            out[s] = (((lane & 1) ? hi : lo) + lk) * 0.693147180559945309417232121458176568;
        }
        estep_epilogue<64, A, true>(a, (long long)row, true, out, kk, valid, lane, lane, 0, 2 * (int)(a.pair_ptr[row + 1] - a.pair_ptr[row]));
    }
}

// ---- the coarse pass's record stream and per-barcode constants (built once per problem, at its first admissible E-step) ----
// first record of every bin: bins hold whole records of bpr batches
__global__ __launch_bounds__(1024) void k_coarse_bin_ptr(const long long *__restrict__ bin_ptr, long long n_bins, int bpr, long long *__restrict__ out)
{
    __shared__ long long part[1024];
    const long long per = (n_bins + 1023) / 1024, b0 = (long long)threadIdx.x * per, b1 = b0 + per < n_bins ? b0 + per : n_bins;
    long long mine = 0;
    for (long long b = b0; b < b1; b++) mine += (bin_ptr[b + 1] - bin_ptr[b] + bpr - 1) / bpr;
    part[threadIdx.x] = mine;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const long long add = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    long long at = part[threadIdx.x] - mine;
    for (long long b = b0; b < b1; b++) {
        out[b] = at;
        at += (bin_ptr[b + 1] - bin_ptr[b] + bpr - 1) / bpr;
    }
    if (threadIdx.x == 1023) out[n_bins] = part[1023];
}

// one wavefront per bin: the bin's groups of the tile-major stream (4 CallPairs = 8 calls each) -> its records (coarse_walk), and -
// every call's keep factor passes through exactly one lane here - the bin's barcodes' sums of log2 keep (EstepArgs::log2_keep;
// a pass of its own over the barcode-major records until round 6: 0.3 ms of a cold 5-iteration call).  The logarithm is the
// hardware's (v_log_f32: keep in [2^-24, 1]), summed in float64: the constant is the same for every option of a barcode, so it
// cancels in the posteriors; the logits of a coarse E-step carry it, within 1e-7 per call of the float64 logarithm's.
template <int CPG>
__global__ __launch_bounds__(256) void k_build_coarse_stream(const CallPair *__restrict__ stream, const long long *__restrict__ bin_ptr,
                                                             const long long *__restrict__ coarse_bin_ptr, long long n_bins, unsigned zero_off,
                                                             unsigned *__restrict__ out, const int *__restrict__ bin_rows, int R,
                                                             double *__restrict__ log2_keep)
{
    using S = CoarseShape<CPG>;
    __shared__ double sh_lk[4][TILE_R_MAX][64];
    const long long bin = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bin >= n_bins) return;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    for (int r = 0; r < R; r++) sh_lk[wave][r][lane] = 0.0;
    const long long g0 = bin_ptr[bin];
    const int n_groups = (int)(bin_ptr[bin + 1] - g0), n_rec = (n_groups + S::BPR - 1) / S::BPR;
    unsigned *dst = out + coarse_bin_ptr[bin] * (CPG * 16);
    for (int i = lane; i < n_rec * CPG * 16; i += 64) {
        const int rec = i / (CPG * 16), w = i % (CPG * 16), blk = w >> 4, d = w & 15;
        const int batch = rec * S::BPR + d / (2 * S::GPB), f = d % (2 * S::GPB), q = f % S::GPB, call = q * CPG + blk;
        unsigned off = zero_off;
        float r = 1.0f;
        unsigned tag = 0u;
        if (batch < n_groups) {
            const CallPair *grp = stream + (g0 + batch) * 4;
            const CallPair p = grp[call >> 1];
            const int half = call & 1;
            tag = grp[0].reserved[0];
            const float keep = p.keep[half], flo = p.floor[half];
            if (keep > 0.0f) {
                off = p.row_off[half];
                r = flo / keep;  // keep = fl(1 - e) >= 2^-24, floor <= 1: r < 2^25, a product of 4 sums below 2^100
                if (f >= S::GPB && (int)(tag & 15u) < R) sh_lk[wave][tag & 15u][lane] += (double)__builtin_amdgcn_logf(keep);  // (the lane that writes this call's r)
            } else if (keep == 0.0f) {
                r = flo;  // p keep + floor = floor: the all-zero row
            } else {
                r = __builtin_nanf("");  // a negative or NaN keep factor (p_base_wrong beyond 1): NaN sums, the guard queues the barcode
            }
        } else if (n_groups > 0) {
            tag = stream[(g0 + n_groups - 1) * 4].reserved[0];  // the padding batches stay in the last group's slot
        }
        unsigned word = off;
        if (f >= S::GPB) {
            word = __float_as_uint(r);
            if (f == S::GPB) word = (word & ~15u) | (tag & 15u);  // r0: the batch's slot tag in its low 4 bits (every block carries it, block 0's is read)
        }
        dst[i] = word;
    }
    for (int r = 0; r < R; r++) {
        double v = sh_lk[wave][r][lane];
        for (int off = 32; off > 0; off >>= 1) v += shfl_xor_f64(v, off);
        const int row = bin_rows[bin * R + r];
        if (lane == 0 && row >= 0) log2_keep[row] = v;
    }
}

// float32 genotype table -> binary16, round to nearest even, at the float32 table's row offsets (EstepArgs::prob16)
__global__ __launch_bounds__(256) void k_prob_to_half(const float *__restrict__ prob, long long n, unsigned short *__restrict__ out, int G,
                                                      const unsigned *__restrict__ skip)
{
    if (skip != nullptr && *skip != 0u) return;  // the coarse pass stands back (GS_SKIP_COARSE)
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long row = i / G;
    const int g = (int)(i - row * G);
    out[row * (2 * G) + g] = __builtin_bit_cast(unsigned short, (_Float16)prob[i]);
}


// ------------------------------------------------------------------------------------
// E-step, block form (K > 1024).  One 256-thread workgroup per barcode, option k = s*256 + tid.
// A chunk of C calls (C a multiple of 8 = the row padding) is staged in LDS TRANSPOSED:
// sh_t[g][c], row stride C+2 dwords, so that the probabilities of genotype g for the call pair
// (c, c+1) are one aligned 8-byte word = exactly the packed operand of the two-term log.  Every
// thread then walks its options with two ds_read_b64 per term pair.  Consecutive lanes hold
// consecutive pairs (g1, g2): g1 is (nearly) wave-uniform -> LDS broadcast; g2 is consecutive ->
// lane stride (C+2) dwords, conflict-free inside each 32-lane service group of ds_read_b64.
// ------------------------------------------------------------------------------------
// K > 33 * 256 options (doublets of more than 129 genotypes): the options are cut into tiles of 33 * 256, each
// tile is one launch that only leaves its logits (TILED), and k_softmax_rows finishes the rows.
template <int A, bool FAST>
__global__ __launch_bounds__(256) void k_estep_block(EstepArgs a, int C, int k_base)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (FAST && guard_stand_back(a)) return;  // the guarded E-step runs direct: EstepArgs::direct
    if (a.order_count != nullptr && blockIdx.x >= *a.order_count) return;  // the exact redo of a guarded E-step: the queued barcodes only
    const long long b = a.order[blockIdx.x];
    const int K = a.K, G = a.G;
    const int CS = C + 2;
    // LDS carve: sh_t [G*CS] f32 | keep [C] | floor [C] | row offsets [C]
    float *sh_t = (float *)smem;
    float *sh_keep = sh_t + (size_t)G * CS;
    float *sh_floor = sh_keep + C;
    unsigned *sh_off = (unsigned *)(sh_floor + C);

    unsigned a1[A], a2[A];  // LDS dword index of the (g1, .) and (g2, .) rows of this thread's options
#pragma unroll
    for (int s = 0; s < A; s++) {
        const int k = k_base + s * 256 + tid;
        const unsigned pr = a.opt_pairs[k < K ? k : K - 1];
        a1[s] = (pr & 0xFFFFu) * (unsigned)CS;
        a2[s] = (pr >> 16) * (unsigned)CS;
    }
    double acc[A];
    int acc_e[A];          // FAST: binary exponents of the flushed products
    float prod[A];         // FAST: running product of the terms since the last flush
#pragma unroll
    for (int s = 0; s < A; s++) {
        acc[s] = 0.0;
        acc_e[s] = 0;
        prod[s] = 1.0f;
    }

    const unsigned *__restrict__ words = (const unsigned *)(a.pairs + a.pair_ptr[b]);
    const int n_calls = 2 * (int)(a.pair_ptr[b + 1] - a.pair_ptr[b]);  // incl. neutral padding, multiple of 8
    for (int pos = 0; pos < n_calls; pos += C) {
        const int n = (n_calls - pos) < C ? (n_calls - pos) : C;  // multiple of 8
        __syncthreads();
        if (tid < n) {
            const int ci = pos + tid;
            const int w = (ci >> 1) * 8 + (ci & 1);
            sh_off[tid] = words[w];  // byte offset of the prob row
            // HALF the keep factor: ((p1 + p2) * 0.5) * keep and (p1 + p2) * (0.5 * keep) round the same real number once (the
            // halvings are exact: p1 + p2 >= 2 clip, keep = 1 - e is 0 or >= 2^-24), so one multiplication per term less
            sh_keep[tid] = 0.5f * __uint_as_float(words[w + 2]);
            sh_floor[tid] = __uint_as_float(words[w + 4]);
        }
        __syncthreads();
        // wave w stages calls w, w+4, ...: coalesced row read, transposed LDS write
        for (int c = wave; c < n; c += 4) {
            const char *row = (const char *)a.prob + sh_off[c];
            for (int g = lane; g < G; g += 64) sh_t[g * CS + c] = *(const float *)(row + (unsigned)g * 4u);
        }
        __syncthreads();
        for (int c = 0; c < n; c += 2) {
            const npm::f32x2 keep2 = *(const npm::f32x2 *)(sh_keep + c);
            const npm::f32x2 flo2 = *(const npm::f32x2 *)(sh_floor + c);
#pragma unroll
            for (int s = 0; s < A; s++) {
                if (k_base + s * 256 + wave * 64 >= K) continue;  // wave-uniform: this wave's slot lies past the last option
                const npm::f32x2 pa = *(const npm::f32x2 *)(sh_t + a1[s] + c);
                const npm::f32x2 pb = *(const npm::f32x2 *)(sh_t + a2[s] + c);
                npm::f32x2 t = (pa + pb) * keep2;  // keep2 = half the keep factors
                t = t + flo2;
                if constexpr (FAST) {  // see estep_products / estep_flush: one log2 per 8 calls
                    prod[s] = (prod[s] * t.x) * t.y;
                    if ((c & 7) == 6) {
                        acc_e[s] += __builtin_amdgcn_frexp_expf(prod[s]);
                        acc[s] += (double)__builtin_amdgcn_logf(__builtin_amdgcn_frexp_mantf(prod[s]));
                        prod[s] = 1.0f;
                    }
                } else {
                    const npm::f32x2 lp = npm::log_f32_hot2(t);
                    acc[s] += (double)lp.x;
                    acc[s] += (double)lp.y;
                }
            }
        }
    }
    if constexpr (FAST) {
        const double LN2 = 0.693147180559945309417232121458176568;
#pragma unroll
        for (int s = 0; s < A; s++) acc[s] = (acc[s] + (double)acc_e[s]) * LN2;
    }
    __syncthreads();

    // logits of this tile of options; the softmax over complete rows is k_softmax_rows' (fused into this kernel it
    // cost 40 more VGPRs, i.e. a wavefront per SIMD: 12.2 ms against 10.4 ms on 20k x 20k x 64 with doublets)
#pragma unroll
    for (int s = 0; s < A; s++) {
        const int k = k_base + s * 256 + tid;
        if (k < K) {
            const double t = (double)a.pen[k] + acc[s];
            float l = (float)t;
            if (a.prior) {
                const size_t o = (size_t)b * K + k;
                if (a.prior_dtype == DMX_F32)
                    l = l + ((const float *)a.prior)[o];
                else
                    l = (float)((double)l + ((const double *)a.prior)[o]);
            }
            a.logits[(size_t)b * K + k] = l;
        }
    }
}

// Softmax of complete logit rows of any length (after the tiled block form): one workgroup per barcode, the
// exponentials parked in the posterior row itself, numpy-ordered sum by wave 0, bitmap of the singlet columns.
// STAGED: the row goes to LDS (K floats) once, the workgroup forms numpy's pairwise sum of the exponentials there, and
// the posteriors are written once.  (Through global memory - the form kept for rows that do not fit, K > 37 000 - the
// sum is a chain of dependent 4-byte loads by one wavefront: 44 ms on 130k barcodes x 8256 options.)
// ------------------------------------------------------------------------------------
// Tolerance / guarded mode, wide doublet tables: the option triangle in 2 x 3 BLOCKS.  k_estep_block gives every thread a
// run of consecutive options, i.e. per pair of calls two LDS reads (the rows of g1 and of g2) for every option - the LDS
// array, not arithmetic, is what the tolerance arithmetic then runs on (configs[4]: 88 ms where the multiplications are
// worth 35).  Here a thread takes the options (g1, g2) of g1 in {2i, 2i+1} x g2 in {3j, 3j+1, 3j+2}: five row reads for six
// options.  Blocks on the diagonal carry the singlets (g, g) - ((p + p) * 0.5 = p exactly) - and options with g1 > g2 that
// are computed and dropped.  Round 5 (profiles/r5_pmc_pairblocks.txt: VALU 85 % busy, LDS 31 %; the packed float32 operations occupy the
// VALU twice, so packing more of them saved instructions and no time): (1) rows staged pre-scaled - a term is two packed
// additions, not add / multiply / add; (2) 8 calls per trip, unrolled: no flush test and branch per option and pair of calls, the
// LDS reads of a whole group in flight; (3) the register allocation held to 64 VGPRs = 8 wavefronts per SIMD (amdgpu_waves_per_eu;
// 76 bytes of scratch per lane): unrolled at the compiler's own 106 VGPRs (4 wavefronts) the E-step of 130k x 650k x 128 took 72.4 ms,
// at 84 (5) 72.0, at 80 (6) 56.4, at 72 (7) 58.5, at 64 (8) 54.7 - against 65.4 not unrolled and 68.4 in round 4; blocks of 2 x 4 /
// 3 x 3 / 3 x 4 options at 8 wavefronts: 105 / 166 / 248 ms (144 - 248 bytes of scratch).
// Same staging and products of 8 as k_estep_block<., true> - but the rows are staged pre-scaled by
// keep / 2, a term is two additions (estep_epilogue.h: GUARD_PER_CALL_PRESCALED); the rows' softmax and the guard are
// k_softmax_rows'.
// ------------------------------------------------------------------------------------
#ifndef DMX_PAIRBLOCK_WAVES
#define DMX_PAIRBLOCK_WAVES 8  // wavefronts per SIMD the register allocation of k_estep_pairblocks aims at (0: the compiler's choice); see below
#endif
template <int R1, int R2, int THREADS>
__global__ __launch_bounds__(THREADS)
#if DMX_PAIRBLOCK_WAVES
__attribute__((amdgpu_waves_per_eu(DMX_PAIRBLOCK_WAVES, DMX_PAIRBLOCK_WAVES)))
#endif
void k_estep_pairblocks(EstepArgs a, int C, int blk_base)
{
    constexpr int NO = R1 * R2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (guard_stand_back(a)) return;  // the guarded E-step runs direct (EstepArgs::direct)
    const long long b = a.order[blockIdx.x];
    const int K = a.K, G = a.G;
    const int CS = C + 2;
    float *sh_t = (float *)smem;  // [G][CS] transposed rows of the chunk's calls
    float *sh_keep = sh_t + (size_t)G * CS;
    float *sh_floor = sh_keep + C;
    unsigned *sh_off = (unsigned *)(sh_floor + C);

    const int blk = blk_base + tid;
    const bool mine = blk < a.n_pair_blocks;
    const bool wave_active = blk_base + wave * 64 < a.n_pair_blocks;  // (uniform)
    const unsigned ij = a.pair_blocks[mine ? blk : a.n_pair_blocks - 1];
    const int g1_0 = R1 * (int)(ij & 0xFFFFu), g2_0 = R2 * (int)(ij >> 16);
    unsigned r1[R1], r2[R2];  // LDS dword index of the rows (clamped: options past G are dropped at the end)
#pragma unroll
    for (int x = 0; x < R1; x++) r1[x] = (unsigned)min(g1_0 + x, G - 1) * (unsigned)CS;
#pragma unroll
    for (int y = 0; y < R2; y++) r2[y] = (unsigned)min(g2_0 + y, G - 1) * (unsigned)CS;
    double acc[NO];
    int acc_e[NO];
    // running product of the terms since the last flush, the even calls in .x and the odd ones in .y: ONE packed multiplication
    // per pair of calls (v_pk_mul_f32) instead of two dependent ones; the halves are multiplied when the 8 terms are flushed -
    // the same seven roundings per 8 terms in another association (the guard's bound counts roundings, not their order)
    npm::f32x2 prod[NO];
#pragma unroll
    for (int o = 0; o < NO; o++) {
        acc[o] = 0.0;
        acc_e[o] = 0;
        prod[o] = npm::f32x2{1.0f, 1.0f};
    }
    const unsigned *__restrict__ words = (const unsigned *)(a.pairs + a.pair_ptr[b]);
    const int n_calls = 2 * (int)(a.pair_ptr[b + 1] - a.pair_ptr[b]);  // incl. neutral padding, multiple of 8
    for (int pos = 0; pos < n_calls; pos += C) {
        const int n = (n_calls - pos) < C ? (n_calls - pos) : C;  // multiple of 8
        __syncthreads();
        if (tid < n) {
            const int ci = pos + tid;
            const int w = (ci >> 1) * 8 + (ci & 1);
            sh_off[tid] = words[w];
            sh_keep[tid] = 0.5f * __uint_as_float(words[w + 2]);  // half the keep factor: see k_estep_block
            sh_floor[tid] = __uint_as_float(words[w + 4]);
        }
        __syncthreads();
        for (int c = wave; c < n; c += THREADS / 64) {  // rows staged pre-scaled: u_g = p_g (keep / 2)  (GUARD_PER_CALL_PRESCALED)
            const char *row = (const char *)a.prob + sh_off[c];
            const float half_keep = sh_keep[c];
            for (int g = lane; g < G; g += 64) sh_t[g * CS + c] = *(const float *)(row + (unsigned)g * 4u) * half_keep;
        }
        __syncthreads();
        if (!wave_active) continue;
        // 8 calls per trip (rows are padded to 8 calls, the chunks hold whole groups): four pairs of calls unrolled - no flush
        // test and no branch per option and pair, the LDS reads of a whole group in flight -, then one flush per option
        for (int c = 0; c < n; c += 8) {
#pragma unroll
            for (int q = 0; q < 8; q += 2) {
                const npm::f32x2 flo2 = *(const npm::f32x2 *)(sh_floor + c + q);
                npm::f32x2 pa[R1], pb[R2];
#pragma unroll
                for (int x = 0; x < R1; x++) pa[x] = *(const npm::f32x2 *)(sh_t + r1[x] + c + q);
#pragma unroll
                for (int y = 0; y < R2; y++) pb[y] = *(const npm::f32x2 *)(sh_t + r2[y] + c + q);
#pragma unroll
                for (int x = 0; x < R1; x++)
#pragma unroll
                    for (int y = 0; y < R2; y++) {
                        const npm::f32x2 t = (pa[x] + pb[y]) + flo2;
                        const int o = R2 * x + y;
#if DMX_PAIRBLOCK_SERIAL_PRODUCT
                        prod[o].x = (prod[o].x * t.x) * t.y;
#else
                        prod[o] = prod[o] * t;
#endif
                    }
            }
#pragma unroll
            for (int o = 0; o < NO; o++) {
                const float whole = prod[o].x * prod[o].y;
                acc_e[o] += __builtin_amdgcn_frexp_expf(whole);
                acc[o] += (double)__builtin_amdgcn_logf(__builtin_amdgcn_frexp_mantf(whole));
                prod[o] = npm::f32x2{1.0f, 1.0f};
            }
        }
    }
    if (!mine) return;
    const double LN2 = 0.693147180559945309417232121458176568;
#pragma unroll
    for (int x = 0; x < R1; x++)
#pragma unroll
        for (int y = 0; y < R2; y++) {
            const int g1 = g1_0 + x, g2 = g2_0 + y;
            if (g1 > g2 || g2 >= G) continue;
            // option index: singlets (g, g) first, then g1 < g2 row-major (dmx_api.cpp: ensure_options)
            const int k = g1 == g2 ? g1 : G + g1 * (2 * G - g1 - 1) / 2 + (g2 - g1 - 1);
            const double t = (double)a.pen[k] + (acc[R2 * x + y] + (double)acc_e[R2 * x + y]) * LN2;
            float l = (float)t;
            if (a.prior) {
                const size_t o = (size_t)b * K + k;
                if (a.prior_dtype == DMX_F32)
                    l = l + ((const float *)a.prior)[o];
                else
                    l = (float)((double)l + ((const double *)a.prior)[o]);
            }
            a.logits[(size_t)b * K + k] = l;
        }
}

template <bool STAGED>
__global__ __launch_bounds__(256) void k_softmax_rows(EstepArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float sh_row[];
    __shared__ float sh_red[8];
    __shared__ float sh_guard[12];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // rows: all B, or the barcodes the guarded pass queued (their number is known on the device only); a guarded E-step that
    // runs direct (EstepArgs::direct): the pass after the fast option tiles stands back (the queue then holds every barcode)
    if (a.guard == 1 && a.direct != nullptr && *a.direct != 0u) return;
    if (a.order_count != nullptr && blockIdx.x >= *a.order_count) return;
    const long long b = a.order_count != nullptr ? (long long)a.order[blockIdx.x] : (long long)blockIdx.x;
    const int K = a.K, G = a.G;
    const float *__restrict__ lg = a.logits + (size_t)b * K;
    float *post = a.post + (size_t)b * K;
    float mx = -__builtin_inff();
    // the logits are read once, eight independent loads in flight per thread (one at a time, a row of 8256 options
    // took 150 us: 33 load latencies per pass)
    for (int k0 = tid; k0 < K; k0 += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = k0 + 256 * u < K ? lg[k0 + 256 * u] : -__builtin_inff();
#pragma unroll
        for (int u = 0; u < 8; u++) {
            mx = fmaxf(mx, v[u]);
            if (STAGED && k0 + 256 * u < K) sh_row[k0 + 256 * u] = v[u];
        }
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if (lane == 0) sh_red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(sh_red[0], sh_red[1]), fmaxf(sh_red[2], sh_red[3]));
    float *x = STAGED ? sh_row : post;
    for (int k = tid; k < K; k += 256) x[k] = npm::exp_f32((STAGED ? sh_row[k] : lg[k]) - mx);
    __threadfence_block();
    __syncthreads();
    if constexpr (STAGED) {
        // np.sum(x) by the whole workgroup along the plan of the host (EstepArgs::sum_plan: numpy's pairwise tree spelled
        // out): the blocks of <= 128 elements by groups of 8 lanes, 32 at a time, then the inner nodes level by level,
        // then the 8192-element chunks left to right.  One wavefront walking the tree with an explicit stack took
        // 180 us per row of 8256 options - 30 ms of the 257 ms E-step of 130k x 650k x 128 with doublets.
        const float tot = npm::plan_sum_block<float>(x, a.sum_plan, sh_row + K, tid);
        if (tid == 0) sh_red[4] = tot;
    } else if (wave == 0) {
        const float tot = npm::row_sum_wave(x, K, lane);
        if (lane == 0) sh_red[4] = tot;
    }
    __syncthreads();
    const float tot = sh_red[4];
    const int W = (G + 63) >> 6;
    // Guarded mode after the tolerance-mode option tiles (estep_epilogue.h: estep_guard, same bound): the sums themselves
    // are gone, |S_k| <= |logit_k| + |pen_k| stands in (no prior logits in this mode: run_estep).
    bool redo = false;
    if (a.guard && guard_active(a)) {
        const bool counting = a.guard == 2;  // the exact tiles of a direct E-step: what the guard would have queued is only counted
        const float n = (float)(2 * (a.pair_ptr[b + 1] - a.pair_ptr[b]));
        float dmax = 0.0f;
        bool bad = false;
        for (int k = tid; k < K; k += 256) {
            const float l = lg[k];
            const float d = GUARD_RHO * (fabsf(l) + fabsf(a.pen[k]) + GUARD_POSITIVE_TERM * n) + a.guard_per_call * (n + 8.0f) + GUARD_LOGIT_ROUNDING * fabsf(l);
            bad |= !(d < 0.25f);
            dmax = fmaxf(dmax, d);
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, off));
        if (lane == 0) sh_guard[wave] = dmax;
        __syncthreads();
        dmax = fmaxf(fmaxf(sh_guard[0], sh_guard[1]), fmaxf(sh_guard[2], sh_guard[3]));
        const float x2 = 2.0f * dmax;
        const float e2 = x2 * (1.0f + x2) * 1.000001f;
        const float near = mx - (x2 + 2.4e-7f * fabsf(mx)) * 1.000001f;
        int close = 0;
        for (int k = tid; k < K; k += 256) {
            const float p = x[k] / tot;
            bad |= !(fminf(p, 1.0f - p) * e2 <= (K > 1024 ? GUARD_TOL_WIDE : GUARD_TOL));
            close += !(lg[k] < near) ? 1 : 0;
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) close += __shfl_xor(close, off);
        const bool any_bad = __ballot(bad) != 0ull;
        if (lane == 0) {
            sh_guard[4 + wave] = (float)close;
            sh_guard[8 + wave] = any_bad ? 1.0f : 0.0f;
        }
        __syncthreads();
        const float total_close = sh_guard[4] + sh_guard[5] + sh_guard[6] + sh_guard[7];
        const bool flagged = total_close != 1.0f || (sh_guard[8] + sh_guard[9] + sh_guard[10] + sh_guard[11]) != 0.0f;
        if (flagged && tid == 0) guard_note(a, b, counting);
        redo = flagged && !counting;
    }
    for (int k0 = 0; k0 < K; k0 += 256) {  // uniform trip count: the ballots below need whole waves
        const int k = k0 + tid;
        float p = 0.0f;
        if (k < K) {
            p = x[k] / tot;
            post[k] = p;
            if (a.post_singlets != nullptr && k < G) a.post_singlets[(size_t)b * G + k] = p;
        }
        if (k0 < G) {
            const unsigned long long bal = __ballot(k < G && !(p <= a.nz_floor));
            const int word = (k0 >> 6) + wave;
            if (lane == 0 && word < W) a.nz[(size_t)b * W + word] = bal;
            if (a.first && word == 0 && lane == (bal ? __builtin_ctzll(bal) : 0)) {
                a.first[b] = nz_code(bal, p);
                if (a.dense_calls && !redo && __popcll(bal) > 4)
                    atomicAdd(a.dense_calls + 1 + (b & (DENSE_SLOTS - 1)), (unsigned long long)(2 * (a.pair_ptr[b + 1] - a.pair_ptr[b])));
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// M-step.  Work item = a run of <= item_calls (kernels.h) consecutive CSC calls of one variant, owned by one
// wavefront that walks it in CSC (= reference bincount) order with float64 accumulation and leaves
// one float64 partial per item.  Items are handed out from a length-sorted list.
// The E-step leaves a per-barcode bitmap of the posteriors that can contribute (kernels.h:
// NZ_FLOOR_SQUARE); everything else adds exactly +0 and is never loaded.
//
// Genotype-per-lane form (G > 64; A = ceil(G / 64) accumulators per lane): 64 calls per chunk are
// loaded one per lane, then handled call by call: v_readlane -> SGPRs, the call's bitmap word becomes
// the EXEC mask of the row gather (inverse ballot), the row address is an SGPR soffset of a buffer load.
// ------------------------------------------------------------------------------------
// pair-major operands, as in the E-step: [q][s] = calls 2q (.x) and 2q+1 (.y) of genotype slot s
template <int A, int H, bool SQUARE>
static __device__ __forceinline__ void mstep_terms(const npm::f32x2 (&p)[H][A], const npm::f32x2 (&keep)[H], float power,
                                                   double (&acc)[A])
{
#pragma unroll
    for (int q = 0; q < H; q++) {
#pragma unroll
        for (int s = 0; s < A; s++) {
            npm::f32x2 c = p[q][s] * keep[q];  // v_pk_mul_f32: calls 2q and 2q+1
            if (SQUARE) {
                c = c * c;
            } else {
                c.x = powf(c.x, power);
                c.y = powf(c.y, power);
            }
            acc[s] += (double)c.x;  // call order preserved
            acc[s] += (double)c.y;
        }
    }
}

// Where the sums of work item `item` go: straight into the output table when its variant has no other item (the value
// k_mcombine would form is 0.0 + acc = acc), else into the item's partial row.
static __device__ __forceinline__ void mstep_store(const MstepArgs &a, long long item, int g, double acc)
{
    if (a.item_variant != nullptr) {
        const long long v = a.item_variant[item];
        if (a.item_ptr[v + 1] - a.item_ptr[v] == 1) {
            const size_t o = (size_t)(a.prow ? (long long)a.prow[v] : v) * a.G + g;
            if (a.out32) a.out32[o] = (float)acc;
            else a.out64[o] = acc;
            return;
        }
    }
    a.partial[(size_t)item * a.G + g] = acc;
}

template <int A, int U, bool SQUARE, bool SMALL>
__global__ __launch_bounds__(256) void k_mstep(MstepArgs a)
{
    static_assert(64 % U == 0, "chunk must be a multiple of the unroll");
    const int lane = threadIdx.x & 63;
    const int G = a.G;
    const int W = (G + 63) >> 6;
    double acc[A];
#pragma unroll
    for (int s = 0; s < A; s++) acc[s] = 0.0;

    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long slot = (long long)blockIdx.x * 4 + wave;
    if (slot >= a.n_items) return;
    const long long item = a.order[slot];
    const int n = a.item_len[item];
    const uint2 *__restrict__ calls = a.calls + a.item_start[item];
    // row offsets fit the 32-bit soffset of a buffer load when the posterior table is < 4 GiB
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)a.post, 0, SMALL ? (int)a.post_bytes : 0, 0x00020000);
    unsigned voff[A];
#pragma unroll
    for (int s = 0; s < A; s++) voff[s] = (unsigned)(lane + 64 * s) * 4u;

    // Software pipeline: records two chunks ahead, bitmaps one chunk ahead.
    auto load_records = [&](int c0) {
        const int ci = c0 + lane;
        uint2 d = make_uint2(0u, 0u);  // keep bits 0 -> keep = +0: padding adds (p*0)^power = +0
        if (ci < n) d = calls[ci];
        return d;
    };
    auto load_bitmap = [&](int c0, uint2 d, int s) {
        unsigned long long m = 0ull;
        if (c0 + lane < n && s < W) m = a.nz[(size_t)d.x * W + s];
        return m;
    };
    uint2 d_cur = load_records(0);
    uint2 d_nxt = load_records(64);
    unsigned long long m_cur[A], m_nxt[A];
#pragma unroll
    for (int s = 0; s < A; s++) m_cur[s] = load_bitmap(0, d_cur, s);
    for (int c0 = 0; c0 < n; c0 += 64) {
        const uint2 d_nn = load_records(c0 + 128);
#pragma unroll
        for (int s = 0; s < A; s++) m_nxt[s] = load_bitmap(c0 + 64, d_nxt, s);
        const float keep_v = __uint_as_float(d_cur.y);
        const int cnt = (n - c0) < 64 ? (n - c0) : 64;
        for (int i0 = 0; i0 < cnt; i0 += U) {
            constexpr int H = U / 2;
            npm::f32x2 p[H][A], keep[H];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const unsigned cb = (unsigned)__builtin_amdgcn_readlane((int)d_cur.x, i0 + u);
                const float kp = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, keep_v), i0 + u));
                if (u & 1) keep[u >> 1].y = kp; else keep[u >> 1].x = kp;
                const unsigned long long row = (unsigned long long)cb * (unsigned long long)a.K * 4ull;
#pragma unroll
                for (int s = 0; s < A; s++) {
                    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)m_cur[s], i0 + u);
                    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(m_cur[s] >> 32), i0 + u);
                    const unsigned long long m = ((unsigned long long)hi << 32) | lo;
                    float v = 0.0f;
                    if (__builtin_amdgcn_inverse_ballot_w64(m)) {  // EXEC := bitmap
                        if (SMALL)
                            v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                              rsrc, (int)voff[s], (int)(unsigned)row, 0));
                        else
                            v = *(const float *)((const char *)a.post + row + voff[s]);
                    }
                    if (u & 1) p[u >> 1][s].y = v; else p[u >> 1][s].x = v;
                }
            }
            mstep_terms<A, H, SQUARE>(p, keep, a.power, acc);
        }
        d_cur = d_nxt;
        d_nxt = d_nn;
#pragma unroll
        for (int s = 0; s < A; s++) m_cur[s] = m_nxt[s];
    }
#pragma unroll
    for (int s = 0; s < A; s++) {
        const int g = lane + 64 * s;
        if (g < G) mstep_store(a, item, g, acc[s]);
    }
}

// ------------------------------------------------------------------------------------
// M-step, call-parallel form (G <= 64).  The genotype-per-lane form above spends ~14 VALU
// instructions per call although almost every call has ONE posterior that can contribute (200k x
// 100k x 64 workload: 94.5% of the calls).  Here a wavefront still owns one work item and lane g still owns
// the float64 accumulator of genotype g, but a chunk of 64 calls is first turned into per-genotype
// QUEUES in LDS: val[r][g] = the r-th contribution (in call order) to genotype g.  Lane g then adds
// val[0][g], val[1][g], ... -- exactly the addends of the sequential walk in exactly its order
// (np.bincount order), so the sums stay bit-identical, in max-queue-length iterations of 3
// instructions instead of 64 iterations of 8.
//   queue position of call i for genotype g = number of calls j < i of the chunk with bit g set
//                                           = popcount(colmask[g] & lanes below i),
// colmask = the transposed 64 x 64 bit matrix of the calls' non-zero bitmaps:
//   * sparse calls (<= NZ_S non-zeros) are handled one per lane: lane i ORs its bit into colmask[g]
//     in LDS (ds_or_b64) for each of its genotypes, later gathers post[cb_i, g], squares and stores;
//   * dense calls (uninformative barcodes, all genotypes alive) are handled one at a time by the
//     whole wavefront, lane g fetching post[cb_i, g] from the coalesced row.
// The queues hold R entries; a chunk that could overflow them is processed R calls at a time.
// ------------------------------------------------------------------------------------
// counters[0] = sum of the DENSE_SLOTS hashed counters the E-step epilogues add to (one address would serialise
// 200k atomics: +0.65 ms measured); the slots are left at zero for the next E-step (a memset per E-step was two more
// launches: 15 us of every EM iteration)
__global__ __launch_bounds__(256) void k_sum_dense(unsigned long long *counters, unsigned *guard_state)
{
    __shared__ unsigned long long part[4];
    if (guard_state != nullptr && threadIdx.x == 0) guard_state[GS_T_END] = (unsigned)wall_clock64();  // the E-step's kernels are done
    unsigned long long s = 0;
    for (int i = threadIdx.x; i < DENSE_SLOTS; i += 256) {
        s += counters[1 + i];
        counters[1 + i] = 0ull;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned lo = __shfl_down((unsigned)s, off), hi = __shfl_down((unsigned)(s >> 32), off);
        s += ((unsigned long long)hi << 32) | lo;
    }
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) counters[0] = part[0] + part[1] + part[2] + part[3];
}

// Between two E-steps of the guarded mode: books the E-step that has finished (its queue length, or - run direct - what the
// guard would have queued, summed from the hashed counters; the durations of its two passes from the stamps their kernels
// left), and decides how the next one runs: direct when the fast pass + the exact redo of the queued share costs more than
// the exact kernel over every barcode (kernels.h: EstepArgs::direct).  One workgroup; replaces the memset of the queue length.
__global__ __launch_bounds__(GUARD_SLOTS) void k_guard_begin(unsigned *st, unsigned B, unsigned K, int adaptive, int capable, int allow_coarse)
{
    static_assert(GUARD_QUEUES == GUARD_SLOTS, "one thread per slot and per sub-queue");
    __shared__ unsigned part[2][GUARD_SLOTS / 64];
    unsigned v = st[GS_SLOTS_FINE + threadIdx.x], vc = st[GS_SLOTS_COARSE + threadIdx.x];
    st[GS_SLOTS_FINE + threadIdx.x] = 0u;
    st[GS_SLOTS_COARSE + threadIdx.x] = 0u;
    st[GS_QUEUE_LEN + threadIdx.x] = 0u;  // the sub-queues of the coming E-step are empty
    for (int off = 32; off > 0; off >>= 1) {
        v += __shfl_down(v, off);
        vc += __shfl_down(vc, off);
    }
    if ((threadIdx.x & 63) == 0) {
        part[0][threadIdx.x >> 6] = v;
        part[1][threadIdx.x >> 6] = vc;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const unsigned was = st[GS_LEVEL], rows = st[GS_ROWS], was_capable = st[GS_CAPABLE];
    const bool was_direct = was == 2u;
    unsigned hashed_fine = 0u, hashed_coarse = 0u;
    for (int i = 0; i < GUARD_SLOTS / 64; i++) {
        hashed_fine += part[0][i];
        hashed_coarse += part[1][i];
    }
    // what the two guards flagged in the finished E-step: the guard of the pass that ran filled the queue, the other one (a direct
    // E-step: both) counted on its hashed slots
    const unsigned queued = was_direct ? 0u : st[GS_COUNT];
    const unsigned count_fine = was == 1u ? queued : hashed_fine;
    const unsigned count_coarse = was == 0u ? queued : (was_capable ? hashed_coarse : GS_UNKNOWN);
    const unsigned count = was == 0u ? count_coarse : count_fine;  // of the pass that ran (direct: what the fine pass would have queued)
    if (st[GS_PENDING]) {
        unsigned long long total = ((unsigned long long)st[GS_TOTAL + 1] << 32) | st[GS_TOTAL];
        total += was_direct ? rows : count;
        st[GS_TOTAL] = (unsigned)total;
        st[GS_TOTAL + 1] = (unsigned)(total >> 32);
    }
    const bool same = st[GS_VALID] && rows == B && st[GS_K] == K;
    if (!same) st[GS_F_TICKS] = st[GS_C_TICKS] = st[GS_E_TICKS] = st[GS_E_MEASURED] = st[GS_O_TICKS] = 0u;
    unsigned level = allow_coarse ? 0u : 1u, streak = 0u;
    if (same) {
        // durations of the finished E-step's passes (32-bit wall clock differences: modular, intervals far below the wrap)
        const unsigned d_fast = st[GS_T_REDO] - st[GS_T_FAST], d_redo = st[GS_T_END] - st[GS_T_REDO];
        if (d_fast < (1u << 30) && d_redo < (1u << 30)) {
            if (was_direct) {
                st[GS_E_TICKS] = d_redo > 0u ? d_redo : 1u;
                st[GS_E_MEASURED] = 1u;
            } else {
                st[was == 0u ? GS_C_TICKS : GS_F_TICKS] = d_fast > 0u ? d_fast : 1u;
                if (1000ull * count < rows) st[GS_O_TICKS] = d_redo;  // an (almost) empty queue: what the exact launch costs by itself
                if (!st[GS_E_MEASURED] && 20ull * count >= rows) {  // the redo's time over its share of the barcodes
                    const unsigned own = st[GS_O_TICKS];
                    const double e = (double)(d_redo > own ? d_redo - own : 1u) * (double)rows / (double)count;
                    st[GS_E_TICKS] = e < 1.0e9 ? (unsigned)e + 1u : 1000000000u;
                }
            }
        }
        st[GS_COUNT_FINE] = count_fine;
        st[GS_COUNT_COARSE] = count_coarse;
        // The coming E-step: the cheapest of  C + f_coarse E  (if admissible),  F + f_fine E  and  E, each pass's time as the device
        // measured it.  A pass that has not run yet is priced from the other one at the ratio they have at 200k x 100k x 64 (C =
        // 0.63 F) - no E-step is spent on finding out: the last E-step of every call takes the fine pass anyway (its logits are
        // read) and measures F, the first admissible one the coarse pass -; without any E (no E-step has queued 5 % of its
        // barcodes) the redo is priced at 1.8 F for the choice between the two passes only - the direct form needs a measured or
        // estimated E.  3 % of hysteresis.
        double F = (double)st[GS_F_TICKS], C = (double)st[GS_C_TICKS];
        const double E = (double)st[GS_E_TICKS];
        if (adaptive && (F > 0.0 || C > 0.0)) {
            if (F == 0.0) F = C / 0.63;
            if (C == 0.0) C = 0.63 * F;
            const double e_redo = E > 0.0 ? E : 1.8 * F;
            const double f_fine = (double)count_fine / (double)rows;
            double cost[3];
            cost[0] = allow_coarse && count_coarse != GS_UNKNOWN ? C + (double)count_coarse / (double)rows * e_redo : 1.0e300;
            cost[1] = F + f_fine * e_redo;
            cost[2] = E > 0.0 ? E : 1.0e300;
            if (was <= 2u) cost[was] *= 0.97;  // (the level that ran stays unless another one is 3 % cheaper)
            level = cost[0] <= cost[1] ? 0u : 1u;
            if (cost[2] < cost[level]) level = 2u;
            // times of the passes that do not run go stale (kernels.h: GUARD_PROBE_STREAK)
            streak = level == was ? st[GS_STREAK] + 1u : 0u;
            if (streak >= GUARD_PROBE_STREAK) {
                unsigned other = 3u;
                double best = 2.0 * cost[level];
                for (unsigned l = 0; l < 3u; l++)
                    if (l != level && cost[l] < best) {
                        best = cost[l];
                        other = l;
                    }
                if (other < 3u) {
                    level = other;
                    st[GS_PROBES] += 1u;
                }
                streak = 0u;
            }
        }
    } else {
        st[GS_COUNT_FINE] = st[GS_COUNT_COARSE] = GS_UNKNOWN;
    }
    st[GS_LEVEL] = level;
    st[GS_DIRECT] = level == 2u;
    st[GS_SKIP_COARSE] = level != 0u;
    st[GS_SKIP_FINE] = level != 1u;
    st[GS_DIRECT_STEPS] += level == 2u;
    st[GS_COARSE_STEPS] += level == 0u;
    st[GS_STREAK] = streak;
    st[GS_CAPABLE] = capable != 0;
    st[GS_COUNT] = 0u;
    st[GS_ROWS] = B;
    st[GS_K] = K;
    st[GS_VALID] = 1u;
    st[GS_PENDING] = 1u;
    st[GS_T_FAST] = st[GS_T_REDO] = st[GS_T_END] = (unsigned)wall_clock64();
}

hipError_t launch_guard_begin(hipStream_t st, unsigned *state, long long B, int K, int adaptive, int capable, int allow_coarse)
{
    hipLaunchKernelGGL(k_guard_begin, dim3(1), dim3(GUARD_SLOTS), 0, st, state, (unsigned)B, (unsigned)K, adaptive, capable, allow_coarse);
    return hipGetLastError();
}

// Between the fast launches and the exact launch of a guarded E-step: block q copies sub-queue q behind the sub-queues before
// it (their lengths added up by every block for itself: 256 words), block 0 leaves the total and the wall clock.  An E-step
// that runs direct: the list becomes every barcode (EstepArgs::order_direct).
__global__ __launch_bounds__(256) void k_guard_compact(unsigned *st, const int *__restrict__ sub, unsigned cap, int *__restrict__ list,
                                                       const int *__restrict__ order_direct, unsigned B)
{
    __shared__ unsigned before[4], all[4];
    const unsigned q = blockIdx.x;
    if (st[GS_DIRECT] != 0u) {  // (uniform) the fast kernels stood back: the exact launch gets every barcode, longest rows first
        for (unsigned i = q * 256u + threadIdx.x; i < B; i += GUARD_QUEUES * 256u) list[i] = order_direct[i];
        if (q == 0 && threadIdx.x == 0) {
            st[GS_COUNT] = B;
            st[GS_T_REDO] = (unsigned)wall_clock64();
        }
        return;
    }
    const unsigned len = st[GS_QUEUE_LEN + threadIdx.x];  // GUARD_QUEUES == 256 threads
    unsigned lo = threadIdx.x < q ? len : 0u, total = len;
    for (int off = 32; off > 0; off >>= 1) {
        lo += __shfl_down(lo, off);
        total += __shfl_down(total, off);
    }
    if ((threadIdx.x & 63) == 0) {
        before[threadIdx.x >> 6] = lo;
        all[threadIdx.x >> 6] = total;
    }
    __syncthreads();
    const unsigned offset = before[0] + before[1] + before[2] + before[3];
    const unsigned mine = st[GS_QUEUE_LEN + q];
    for (unsigned i = threadIdx.x; i < mine; i += 256) list[offset + i] = sub[(size_t)q * cap + i];
    if (q == 0 && threadIdx.x == 0) {
        st[GS_COUNT] = all[0] + all[1] + all[2] + all[3];
        st[GS_T_REDO] = (unsigned)wall_clock64();
    }
}

hipError_t launch_guard_compact(hipStream_t st, unsigned *state, const int *sub, unsigned sub_cap, int *list, const int *order_direct, long long B)
{
    hipLaunchKernelGGL(k_guard_compact, dim3(GUARD_QUEUES), dim3(256), 0, st, state, sub, sub_cap, list, order_direct, (unsigned)B);
    return hipGetLastError();
}

// The wall clock between two launches of a guarded E-step (GS_T_REDO between its fast and its exact pass, GS_T_END behind
// them), as a kernel of its own: a store at the top of the E-step kernels themselves - thread 0 of block 0 stamping its
// start - makes every load behind it "possibly clobbered", and the exact kernel's uniform loads (records, offsets) turned
// from scalar into vector loads: +40 % on its time.
__global__ void k_guard_stamp(unsigned *st, int which) { st[which] = (unsigned)wall_clock64(); }

hipError_t launch_guard_stamp(hipStream_t st, unsigned *state, int which)
{
    hipLaunchKernelGGL(k_guard_stamp, dim3(1), dim3(1), 0, st, state, which);
    return hipGetLastError();
}

hipError_t launch_sum_dense(hipStream_t st, unsigned long long *counters, unsigned *guard_state)
{
    hipLaunchKernelGGL(k_sum_dense, dim3(1), dim3(256), 0, st, counters, guard_state);
    return hipGetLastError();
}

// Which of the two G <= 64 kernels runs is decided on the device, from the statistic the E-step left: both are
// launched, one returns at once.  dense: more than a quarter of the (padded) calls belong to barcodes with more than
// 4 live posteriors (uninformative genotypes, first iterations of a run that starts from barcode labels).
static __device__ __forceinline__ bool dense_regime(const MstepArgs &a)
{
    return a.dense_calls != nullptr && 4ull * *a.dense_calls > a.total_calls;
}

// whether this M-step runs the full pass (k_mstep_tiles, or the fixed-point work-item form) instead of the delta pass: see MIncrArgs
static __device__ __forceinline__ bool incr_full(const unsigned *state, const MstepArgs &a)
{
    const unsigned long long calls = ((unsigned long long)state[IS_CALLS + 1] << 32) | state[IS_CALLS];
    if (state[IS_FORCE] != 0u) return dense_regime(a);
    return state[IS_VALID] == 0u || 8ull * calls > (a.incr_total ? a.incr_total : a.total_calls) || dense_regime(a);
}

// c in [0, 1] -> rint(c 2^shift) as an integer (k_mstep_tiles, k_mincr_delta and the fixed-point work-item form add the same
// integers): adding 1.5 x 2^52 leaves the rounded value (ties to even) in the low bits of the sum's mantissa (c 2^shift < 2^51)
static __device__ __forceinline__ unsigned long long fixed_of(float c, int shift)
{
    constexpr double MAGIC = 6755399441055744.0;
    const double d = __builtin_ldexp((double)c, shift) + MAGIC;
    return (unsigned long long)(__double_as_longlong(d) - __double_as_longlong(MAGIC));
}

// Where the integer sums of work item `item` go (fixed-point work-item form, MstepArgs::fixed_shift_v): a variant of ONE item is
// finished here - its sums into fixed_acc64, converted once into the addition, as k_mstep_tiles writes them -, the others leave
// their integers in the item's partial row for k_mcombine.
// (an integer sum into the output table: one conversion - float32, or float64 on the way into a float64 exchange; prow: the padded rows
// of the multi-GPU exchange buffer)
static __device__ __forceinline__ void mstep_out_fixed(const MstepArgs &a, long long v, int g, unsigned long long acc, int shift)
{
    const size_t o = (size_t)(a.prow ? (long long)a.prow[v] : v) * a.G + g;
    const double sum = __builtin_ldexp((double)(long long)acc, -shift);
    if (a.out32) a.out32[o] = (float)sum;
    else a.out64[o] = sum;
}

static __device__ __forceinline__ void mstep_store_fixed(const MstepArgs &a, long long item, int g, unsigned long long acc, int shift)
{
    const long long v = a.item_variant[item];
    if (a.item_ptr[v + 1] - a.item_ptr[v] == 1) {
        mstep_out_fixed(a, v, g, acc, shift);
        a.fixed_acc64[(size_t)v * a.G + g] = acc;
        return;
    }
    a.partial[(size_t)item * a.G + g] = __longlong_as_double((long long)acc);
}

// ------------------------------------------------------------------------------------
// M-step, dense form (G <= 64, most posteriors alive): the call-parallel form below handles a call with more than 4
// live posteriors one at a time through its queues (4.2 ms on 200k x 100k x 64 with uniform posteriors); here a
// wavefront simply walks its work item in order, gathers the whole posterior row of every call (lane g = genotype g,
// row offset as the buffer load's scalar offset), and adds (p keep)^2 in float64.  Posteriors at or below the
// contribution floor add exactly +0, so the sums are bit-identical to the sparse form's.
// ------------------------------------------------------------------------------------
template <bool SQUARE>
__global__ __launch_bounds__(256) void k_mstep_dense(MstepArgs a)
{
    if (!dense_regime(a)) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // (a grid that strides over the items: most launches find the regime sparse and return - with one block per four items
    // that alone took 12 us of every EM iteration)
    for (long long slot = (long long)blockIdx.x * 4 + wave; slot < a.n_items; slot += (long long)gridDim.x * 4) {
        const long long item = a.order[slot];
        const int n = a.item_len[item];
        const uint2 *__restrict__ calls = a.calls + a.item_start[item];
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.post, 0, (int)a.post_bytes, 0x00020000);
        const unsigned row_bytes = (unsigned)a.K * 4u;
        const unsigned voff = (unsigned)(lane < a.G ? lane : 0) * 4u;
        double acc = 0.0;
        auto records = [&](int c0) {
            uint2 d = make_uint2(0u, 0u);  // padding: keep bits 0 -> (p * 0)^power = +0
            if (c0 + lane < n) d = calls[c0 + lane];
            return d;
        };
        // row gathers in flight per wavefront: 8 -> 2.52 ms, 16 -> 2.42 ms, 32 -> 2.43 ms on 200k x 100k x 64 with uniform
        // posteriors (the fabric-side gather rate is the limit, not their latency)
        constexpr int DENSE_ROWS = 16;
        uint2 d_cur = records(0);
        for (int c0 = 0; c0 < n; c0 += 64) {
            const uint2 d_nxt = records(c0 + 64);
            const int cnt = (n - c0) < 64 ? (n - c0) : 64;
            for (int i0 = 0; i0 < cnt; i0 += DENSE_ROWS) {
                float p[DENSE_ROWS];
    #pragma unroll
                for (int u = 0; u < DENSE_ROWS; u++) {
                    const unsigned cb = (unsigned)__builtin_amdgcn_readlane((int)d_cur.x, (i0 + u) & 63);
                    p[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, (int)(cb * row_bytes), 0));
                }
    #pragma unroll
                for (int u = 0; u < DENSE_ROWS; u++) {
                    const float keep = __builtin_bit_cast(float, __builtin_amdgcn_readlane((int)d_cur.y, (i0 + u) & 63));
                    float c = p[u] * keep;
                    c = SQUARE ? c * c : powf(c, a.power);
                    acc += (double)c;
                }
            }
            d_cur = d_nxt;
        }
        if (lane < a.G) mstep_store(a, item, lane, acc);
    }
}

// BUF: the three per-lane loads of the call-parallel part (records, barcode code, extra posteriors) as raw buffer
// loads with 32-bit offsets, inactive lanes pointed out of range (a buffer load past num_records returns 0 and makes
// no request): no EXEC regions, nothing merged after a load (the compiler otherwise parks an s_waitcnt vmcnt(0)
// behind the first conditional posterior gather of every chunk).  Needs the posterior table below 4 GiB and
// barcode indices below 2^24 (launcher).
template <bool SQUARE, int R, int D, bool BUF, bool FIXED = false>
__global__ __launch_bounds__(256) void k_mstep_calls(MstepArgs a)
{
    if (dense_regime(a)) return;
    if (FIXED && a.fixed_state != nullptr && !incr_full(a.fixed_state, a)) return;  // the delta pass updates the sums (k_mincr_delta)
    constexpr int NZ_S = NZ_CODE;  // "sparse" call: at most this many non-zero posteriors
    typedef unsigned long long u64;
    __shared__ float sh_val[4][R * 64];
    __shared__ u64 sh_colmask[4][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long slot = (long long)blockIdx.x * 4 + wave;
    if (slot >= a.n_items) return;
    const long long item = a.order[slot];
    const int n = a.item_len[item];
    const uint2 *__restrict__ calls = a.calls + a.item_start[item];
    float *val = sh_val[wave];
    u64 *colmask = sh_colmask[wave];
    const int G = a.G;
    const long long K = a.K;
    const u64 bit = 1ull << lane;
    const u64 below = bit - 1ull;
    double acc = 0.0;
    u64 acc_fixed = 0ull;  // FIXED: the sum of rint(c 2^shift), shift = the variant's (uniform: an item belongs to one variant)
    const int shift = FIXED ? __builtin_amdgcn_readfirstlane((int)a.fixed_shift_v[a.item_variant[item]]) : 0;
    auto accumulate = [&](float v) {
        if constexpr (FIXED) acc_fixed += fixed_of(v, shift);
        else acc += (double)v;
    };

    auto power_of = [&](float c) { return SQUARE ? c * c : powf(c, a.power); };
    auto genotype = [](unsigned code, int t) { return (int)((code >> (7 + 6 * t)) & 63u); };
    auto lane_u64 = [&](u64 v, int i) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, i);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), i);
        return ((u64)hi << 32) | lo;
    };
    constexpr unsigned OOB = 0xFFFFFFFFu;
    const __amdgpu_buffer_rsrc_t r_calls = __builtin_amdgcn_make_buffer_rsrc((void *)calls, 0, BUF ? n * 8 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_first = __builtin_amdgcn_make_buffer_rsrc((void *)a.first, 0, BUF ? (int)a.first_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_post = __builtin_amdgcn_make_buffer_rsrc((void *)a.post, 0, BUF ? (int)a.post_bytes : 0, 0x00020000);
    const unsigned row_bytes = (unsigned)K * 4u;
    auto load_records = [&](int c0) {
        if constexpr (BUF) {  // past the item's end: zeros (keep bits 0)
            const auto r = __builtin_amdgcn_raw_buffer_load_b64(r_calls, lane * 8, c0 * 8, 0);
            return make_uint2(r[0], r[1]);
        } else {
            uint2 d = make_uint2(0u, 0u);
            if (c0 + lane < n) d = calls[c0 + lane];
            return d;
        }
    };
    // {first live posterior, code} of this lane's barcode; padding lanes: no live genotype
    auto load_code = [&](int c0, uint2 d) {
        if constexpr (BUF) {
            const auto q = __builtin_amdgcn_raw_buffer_load_b64(r_first, (int)(c0 + lane < n ? d.x * 8u : OOB), 0, 0);
            return make_uint2(q[0], q[1]);
        } else {
            uint2 r = make_uint2(0u, 0u);
            if (c0 + lane < n) r = a.first[(size_t)d.x];
            return r;
        }
    };
    // the <= NZ_S posteriors of this lane's call; the lowest one came with the code.
    // Every p[t] is written by exactly one load, so that nothing waits for it here.
    auto load_sparse = [&](uint2 fc, uint2 d, float (&p)[NZ_S]) {
        const int nnz = (int)(fc.y & 127u);
        const bool sparse = nnz >= 1 && nnz <= NZ_S;
        p[0] = sparse ? __uint_as_float(fc.x) : 0.0f;
        if constexpr (BUF) {  // straight-line: a load with every lane out of range costs its issue only
            const unsigned base = __umul24(d.x, row_bytes);
#pragma unroll
            for (int t = 1; t < NZ_S; t++) {
                const unsigned off = sparse && t < nnz ? base + 4u * (unsigned)genotype(fc.y, t) : OOB;
                p[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_post, (int)off, 0, 0));
            }
        } else {
            const bool more = __any(sparse && nnz > 1);  // (uniform) most chunks hold single-posterior calls only
            const float *__restrict__ row = a.post + (size_t)d.x * K;
#pragma unroll
            for (int t = 1; t < NZ_S; t++) {
                p[t] = 0.0f;
                if (!more) continue;
                if (sparse && t < nnz) p[t] = row[genotype(fc.y, t)];
            }
        }
    };
    // Dense calls (more than NZ_S live posteriors; 3.6 % of the calls of the 200k x 100k x 64 workload once it has
    // converged, i.e. two or three in nine chunks out of ten) are handled one at a time by the whole wavefront, lane
    // g taking post[cb, g].  Their rows come from the 51 MB posterior table, i.e. over the fabric: fetched inside
    // the iteration that needs them, that latency was what the kernel ran on.  So the rows of the first DP dense
    // calls of a chunk are requested one iteration ahead, unmasked (a dead posterior is simply not used), and the
    // bitmaps of the dense lanes with them (per-lane gather, other lanes out of range).
    const __amdgpu_buffer_rsrc_t r_nz = __builtin_amdgcn_make_buffer_rsrc((void *)a.nz, 0, BUF ? (int)a.first_bytes : 0, 0x00020000);
    auto load_dense_bitmap = [&](uint2 fc, uint2 d) {
        const bool dense = (int)(fc.y & 127u) > NZ_S;
        if constexpr (BUF) {
            const auto q = __builtin_amdgcn_raw_buffer_load_b64(r_nz, (int)(dense ? d.x * 8u : OOB), 0, 0);
            return ((u64)q[1] << 32) | q[0];
        } else {
            u64 m = 0ull;
            if (dense) m = a.nz[(size_t)d.x];
            return m;
        }
    };
    constexpr int DP = 2;  // dense rows requested one iteration ahead (registers: 2 DP)
    auto prefetch_dense_rows = [&](u64 dense, uint2 d, float (&q)[DP]) {
#pragma unroll
        for (int j = 0; j < DP; j++) {
            const bool have = dense != 0ull;  // uniform
            const unsigned cb = (unsigned)__builtin_amdgcn_readlane((int)d.x, have ? __builtin_ctzll(dense) : 0);
            dense &= dense - 1ull;
            if constexpr (BUF) {
                q[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_post, (int)(have && lane < G ? lane * 4u : OOB),
                                                                                     (int)(cb * row_bytes), 0));
            } else {
                q[j] = 0.0f;
                if (have && lane < G) q[j] = a.post[(size_t)cb * K + lane];
            }
        }
    };
    // lane g's posterior of the dense call held by lane i, fetched now (calls past the first D of a chunk)
    auto load_dense = [&](uint2 d, u64 m, int i) {
        const unsigned cb = (unsigned)__builtin_amdgcn_readlane((int)d.x, i);
        float v = 0.0f;
        if (__builtin_amdgcn_inverse_ballot_w64(lane_u64(m, i))) v = a.post[(size_t)cb * K + lane];
        return v;
    };
    auto load_dense_rows = [&](u64 dense, uint2 d, u64 m, float (&q)[D]) {
#pragma unroll
        for (int j = 0; j < D; j++) {
            q[j] = 0.0f;
            if (dense) {
                q[j] = load_dense(d, m, __builtin_ctzll(dense));
                dense &= dense - 1ull;
            }
        }
        // These rows are waited for HERE, on this (rare) path, whatever the lanes' masks: their first use is inside
        // lane-conditional code, and a load the compiler believes pending on SOME path out of here made it drain the
        // memory pipeline (s_waitcnt vmcnt(0)) in the middle of every chunk's processing - where the next chunks'
        // loads have just been issued.
#pragma unroll
        for (int j = 0; j < D; j++) asm volatile("" : "+v"(q[j]));
    };

#pragma unroll
    for (int r = 0; r < R; r++) val[r * 64 + lane] = 0.0f;
    // software pipeline: records three chunks ahead, codes two, the extra posteriors of the sparse calls and the
    // first rows of the dense ones one
    uint2 d0 = load_records(0), d1 = load_records(64), d2 = load_records(128);
    uint2 fc0 = load_code(0, d0), fc1 = load_code(64, d1);
    u64 dense0 = __ballot((int)(fc0.y & 127u) > NZ_S);
    float ps0[NZ_S], q0[DP];
    load_sparse(fc0, d0, ps0);
    u64 bm0 = load_dense_bitmap(fc0, d0);
    prefetch_dense_rows(dense0, d0, q0);
    for (int c0 = 0; c0 < n; c0 += 64) {
        const uint2 d3 = load_records(c0 + 192);
        const uint2 fc2 = load_code(c0 + 128, d2);
        const u64 dense1 = __ballot((int)(fc1.y & 127u) > NZ_S);
        float ps1[NZ_S], q1[DP];
        load_sparse(fc1, d1, ps1);
        const u64 bm1 = load_dense_bitmap(fc1, d1);
        prefetch_dense_rows(dense1, d1, q1);

        const float keep = __uint_as_float(d0.y);
        const unsigned code = fc0.y;
        const int nnz = (int)(code & 127u);
        const int mine_all = nnz <= NZ_S ? nnz : 0;  // queue entries this lane's (sparse) call makes
        // one dense call: lane g appends its contribution to queue g
        auto put_dense = [&](int i, float q, u64 &cm) {
            const float kp = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, keep), i));
            const bool alive = __builtin_amdgcn_inverse_ballot_w64(lane_u64(bm0, i));
            const u64 call_bit = 1ull << i;
            const int pos = __popcll(cm & (call_bit - 1ull));
            const float c = power_of(q * kp);
            if (alive) {
                val[pos * 64 + lane] = c;
                cm |= call_bit;
            }
        };
        auto transpose = [&](int mine) {
            colmask[lane] = 0ull;
#pragma unroll
            for (int t = 0; t < NZ_S; t++) {
                if (t > 0 && !__any(t < mine)) break;
                if (t < mine) atomicOr(&colmask[genotype(code, t)], bit);
            }
            return colmask[lane];
        };
        auto drain = [&](u64 cm) {
            static_assert(R % 4 == 0, "the drain reads four queue rows at a time");
            // every queue entry is zero outside [put, drain): what is read is cleared again, so that the rows past a
            // lane's count need no select (absent entries add +0.0, which changes nothing)
            const int cnt = __popcll(cm);
            for (int r = 0; __any(r < cnt); r += 4) {
                const float v0 = val[(r + 0) * 64 + lane];
                const float v1 = val[(r + 1) * 64 + lane];
                const float v2 = val[(r + 2) * 64 + lane];
                const float v3 = val[(r + 3) * 64 + lane];
                val[(r + 0) * 64 + lane] = 0.0f;
                val[(r + 1) * 64 + lane] = 0.0f;
                val[(r + 2) * 64 + lane] = 0.0f;
                val[(r + 3) * 64 + lane] = 0.0f;
                accumulate(v0);
                accumulate(v1);
                accumulate(v2);
                accumulate(v3);
            }
        };
        auto put_sparse = [&](int mine) {
#pragma unroll
            for (int t = 0; t < NZ_S; t++) {
                if (t > 0 && !__any(t < mine)) break;
                if (t < mine) {
                    const int g = genotype(code, t);
                    const int pos = __popcll(colmask[g] & below);
                    val[pos * 64 + g] = power_of(ps0[t] * keep);
                }
            }
        };
        auto put_dense_calls = [&](u64 todo, u64 &cm, bool prefetched) {
            if (prefetched) {  // the first DP dense calls of the chunk: their rows are here already
#pragma unroll
                for (int j = 0; j < DP; j++) {
                    if (todo) {
                        put_dense(__builtin_ctzll(todo), q0[j], cm);
                        todo &= todo - 1ull;
                    }
                }
            }
            while (todo) {  // rows fetched here, D at a time
                float q[D];
                load_dense_rows(todo, d0, bm0, q);
#pragma unroll
                for (int j = 0; j < D; j++) {
                    if (todo) {
                        put_dense(__builtin_ctzll(todo), q[j], cm);
                        todo &= todo - 1ull;
                    }
                }
            }
        };

        u64 cm = transpose(mine_all);
        if (!__any(__popcll(cm) + __popcll(dense0) > R)) {
            // ---- the whole chunk fits the queues ----
            put_dense_calls(dense0, cm, true);
            colmask[lane] = cm;
            put_sparse(mine_all);
            drain(cm);
        } else {
            // ---- the queues could overflow: fewer calls at a time (R calls always fit) ----
            int first_lane = 0, width = 32;
            while (first_lane < 64) {
                const u64 range = ((1ull << width) - 1ull) << first_lane;
                const int mine = (range & bit) ? mine_all : 0;
                cm = transpose(mine);
                if (width > R && __any(__popcll(cm) + __popcll(dense0 & range) > R)) {
                    width >>= 1;
                    continue;
                }
                put_dense_calls(dense0 & range, cm, false);
                colmask[lane] = cm;
                put_sparse(mine);
                drain(cm);
                first_lane += width;
            }
        }
        d0 = d1; d1 = d2; d2 = d3;
        fc0 = fc1; fc1 = fc2;
        dense0 = dense1;
        bm0 = bm1;
#pragma unroll
        for (int j = 0; j < DP; j++) q0[j] = q1[j];
#pragma unroll
        for (int t = 0; t < NZ_S; t++) ps0[t] = ps1[t];
    }
    if (lane < G) {
        if constexpr (FIXED) mstep_store_fixed(a, item, lane, acc_fixed, shift);
        else mstep_store(a, item, lane, acc);
    }
}

// ------------------------------------------------------------------------------------
// M-step, tile-major form (kernels.h: MTileArgs; G <= 64; not for the exact additions).
// The item form above walks one variant's calls at a time, and a variant's calls belong to barcodes ~500 apart: one 8-byte
// gather of the barcode's code per call, 64 different cache lines per wavefront load - 78.7 M requests to the L2s on
// 200k x 100k x 64, which is what it ran on (0.68 ms whatever was done to its arithmetic).  Here the records of a TILE of
// up to 128 variants are sorted by barcode, so that the 64 calls of a wavefront load belong to ~64 barcodes out of a
// window of ~130 consecutive ones: the code gather touches 8-16 lines instead of 64.  What the order costs - the
// contributions of a variant no longer arrive together - is paid in LDS: the workgroup keeps accumulators
// [variants of the tile][G] there and every lane adds its call's contribution(s) with an LDS atomic.
// FIXED-POINT accumulators: the reference's np.bincount is a fixed sequential float64 sum and bit-reproducible; float64
// atomics arriving in any order are not (a sum next to a float32 rounding tie could round either way from run to run: what
// this kernel did until round 5).  Every contribution c = (posterior x keep)^power is a float32 in [0, 1]; it is added as the
// 64-bit INTEGER rint(c x 2^s), s = the tile's shift = min(50, 62 - ceil(log2(calls of its longest variant + 1))) (no overflow:
// a variant has at most one call per barcode), so the sum does not depend on the order of the additions at all: the same bits
// on every run, on any number of wavefronts.  Contributions of 2^(23 - s) and more (2^-27 = 7.5e-9 at s = 50: every live
// posterior above 1e-4) are on that grid EXACTLY, so for all but the sums of nothing but dead posteriors the integer sum is
// the exact real sum, rounded once to float64 and once to float32 - the reference's sequential float64 sum carries its own
// 1e-16 relative rounding noise, so the two differ at a float32 rounding tie at most; in general
//     |addition - reference| <= one float32 ulp  +  n 2^-(s + 1)   (n calls of the variant; 1.8e-13 for 400 calls).
//   sparse calls (<= NZ_CODE live posteriors: the code holds the genotypes and the first posterior) one per lane,
//   dense calls one at a time by the whole wavefront, lane g taking post[row, g].
// The workgroup of a tile writes its rows of the output itself: no partial sums, no combining pass.
// ------------------------------------------------------------------------------------
// 1024 threads and 64 + 12 KB of LDS per workgroup: two workgroups = 32 wavefronts per CU (56 VGPRs).  Measured on 200k x 100k x
// 64: 512 threads 0.46 ms, 1024 threads 0.34; 2 / 3 / 4 / 6 / 8 chunks in flight per wavefront 0.50 / 0.35 / 0.34 / 0.45 / 0.44
// (beyond 64 VGPRs half the wavefronts); dense rows 8 / 16 / 32 at a time 0.36 / 0.34 / 0.47; non-temporal record loads: no change.
// Requesting the next round's records before working on this one's: 70 VGPRs, or 64 with spills - 0.40 ms.
// All loads as raw buffer loads with out-of-range masking (no EXEC regions, no 64-bit address arithmetic: 52 VGPRs, a tenth
// fewer instructions): 0.34 ms as well (0.32 against 0.30 at 32 genotypes) - PMC: VALU 64 %, address unit 69 %, LDS 49 % busy.
constexpr int MTILE_THREADS = 1024;
constexpr int MTILE_QUEUE = 96;  // dense calls a wavefront parks before it takes their rows (64 + the flush threshold)

template <bool SQUARE>
__global__ __launch_bounds__(MTILE_THREADS) void k_mstep_tiles(MstepArgs a, MTileArgs t)
{
    if (dense_regime(a)) return;
    if (t.incr_state != nullptr && !incr_full(t.incr_state, a)) return;  // the delta pass updates the sums (k_mincr_delta)
    extern __shared__ __attribute__((aligned(16))) unsigned long long mt_acc[];
    __shared__ unsigned mt_queue[MTILE_THREADS / 64][2][MTILE_QUEUE];  // per wavefront: barcode row | variant in tile << 24, keep bits of its parked dense calls
    const int tile = t.order[blockIdx.x];
    const int v0 = t.first[tile], nv = t.first[tile + 1] - v0;
    const long long beg = t.ptr[tile], end = t.ptr[tile + 1];
    const int G = a.G;
    const long long K = a.K;
    for (int i = threadIdx.x; i < nv * G; i += MTILE_THREADS) mt_acc[i] = 0ull;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int shift = __builtin_amdgcn_readfirstlane(t.shift[tile]);
    unsigned *q_rec = mt_queue[wave][0], *q_keep = mt_queue[wave][1];
    int queued = 0;  // (uniform)
    auto power_of = [&](float c) { return SQUARE ? c * c : powf(c, a.power); };
    // c in [0, 1] -> rint(c 2^shift) as an integer: adding 1.5 x 2^52 leaves the rounded value (ties to even) in the low bits
    // of the sum's mantissa (c 2^shift < 2^51)
    constexpr double MAGIC = 6755399441055744.0;
    auto add = [&](int index, float c) {
        const double d = __builtin_ldexp((double)c, shift) + MAGIC;
        const unsigned long long q = (unsigned long long)(__double_as_longlong(d) - __double_as_longlong(MAGIC));
        __hip_atomic_fetch_add(&mt_acc[index], q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    // The parked dense calls, 16 at a time: lane g takes post[row, g] of every one of them (16 row loads in flight: they come
    // from the posterior table, i.e. over the fabric, and one at a time their latency was the whole kernel), then adds.  A
    // posterior at or below the contribution floor contributes exactly +0 (as in k_mstep_dense): no bitmap is read.
    auto flush = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        constexpr int DB = 16;
        for (int q0 = 0; q0 < queued; q0 += DB) {
            float p[DB];
#pragma unroll
            for (int i = 0; i < DB; i++) {
                p[i] = 0.0f;
                if (q0 + i < queued && lane < G) p[i] = a.post[(size_t)(q_rec[q0 + i] & 0xFFFFFFu) * K + lane];
            }
#pragma unroll
            for (int i = 0; i < DB; i++)
                if (q0 + i < queued && lane < G) add((int)(q_rec[q0 + i] >> 24) * G + lane, power_of(p[i] * __uint_as_float(q_keep[q0 + i])));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        queued = 0;
    };
    constexpr int UN = 4;  // chunks of 64 calls in flight per wavefront: records, then codes, then the rare extra posteriors
    for (long long c0 = beg + (long long)wave * (64 * UN); c0 < end; c0 += (long long)(MTILE_THREADS / 64) * 64 * UN) {
        uint2 rec[UN], code[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const long long i = c0 + 64 * u + lane;
            rec[u] = make_uint2(0u, 0u);
            if (i < end) rec[u] = t.stream[i];
        }
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const long long i = c0 + 64 * u + lane;
            code[u] = make_uint2(0u, 0u);  // no live genotype
            if (i < end) code[u] = a.first[rec[u].x & 0xFFFFFFu];
        }
        float extra[UN][NZ_CODE - 1];  // the second .. fourth live posterior of the sparse calls (5 % of them have any)
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const int nnz = (int)(code[u].y & 127u);
            const float *__restrict__ post_row = a.post + (size_t)(rec[u].x & 0xFFFFFFu) * K;
#pragma unroll
            for (int j = 1; j < NZ_CODE; j++) {
                extra[u][j - 1] = 0.0f;
                if (nnz <= NZ_CODE && j < nnz) extra[u][j - 1] = post_row[(code[u].y >> (7 + 6 * j)) & 63u];
            }
        }
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const unsigned row = rec[u].x & 0xFFFFFFu;
            const int base = (int)(rec[u].x >> 24) * G;
            const float keep = __uint_as_float(rec[u].y);
            const int nnz = (int)(code[u].y & 127u);
            if (nnz >= 1 && nnz <= NZ_CODE) {
                add(base + (int)((code[u].y >> 7) & 63u), power_of(__uint_as_float(code[u].x) * keep));
#pragma unroll
                for (int j = 1; j < NZ_CODE; j++)
                    if (j < nnz) add(base + (int)((code[u].y >> (7 + 6 * j)) & 63u), power_of(extra[u][j - 1] * keep));
            }
            const unsigned long long dense = __ballot(nnz > NZ_CODE);
            if (dense) {  // (uniform)
                if (queued > MTILE_QUEUE - 64) flush();
                if (nnz > NZ_CODE) {
                    const int at = queued + __popcll(dense & ((1ull << lane) - 1ull));
                    q_rec[at] = rec[u].x;
                    q_keep[at] = rec[u].y;
                }
                queued += __popcll(dense);
            }
        }
    }
    if (queued) flush();
    __syncthreads();
    for (int i = threadIdx.x; i < nv * G; i += MTILE_THREADS) {
        const int r = (int)((unsigned)i / (unsigned)G), g = i - r * G;
        const long long v = v0 + r;
        const size_t o = (size_t)(a.prow ? (long long)a.prow[v] : v) * G + g;
        const double sum = __builtin_ldexp((double)(long long)mt_acc[i], -shift);  // one rounding to float64 (sums beyond 2^53 grid units)
        if (a.out32) a.out32[o] = (float)sum;
        else a.out64[o] = sum;
        if (t.acc64) t.acc64[(size_t)v * G + g] = mt_acc[i];  // (incremental M-step: the sums themselves, for the delta passes to come)
    }
}

// ---- incremental M-step (kernels.h: MIncrArgs) ----
constexpr unsigned MINCR_COOLDOWN = 8;  // M-steps that do not look for changes after two full passes the changes asked for (k_mincr_finish)
// a posterior pair that makes a difference to the sums: other bits, and not both below the grid's floor (NaN: a difference)
static __device__ __forceinline__ bool mincr_differs(float now, float before, float floor)
{
    return __float_as_uint(now) != __float_as_uint(before) && !(fmaxf(now, before) < floor);
}

// A lane per barcode: its 8-byte code (nz_code: first live posterior, count, first live genotypes) against the code the sums were
// formed with - a barcode with ONE live posterior then and now and the same code has not changed where it matters (99 % of them on
// converged iterations: 16 bytes per barcode instead of two 256-byte rows).  The others' rows are compared by the whole wavefront, a
// barcode at a time (lane = genotype).
__global__ __launch_bounds__(256) void k_mincr_changes(MstepArgs a, MIncrArgs x)
{
    if (x.state[IS_VALID] == 0u || dense_regime(a)) return;  // the full pass is coming
    // (16 barcodes per wavefront: the row comparisons of a wavefront are one after the other, so more wavefronts = more of them in flight;
    // 64 per wavefront where the last M-step found next to nothing changed was tried: 5 us slower at 200 000 barcodes, not faster)
    const int lane = threadIdx.x & 63;
    const long long b = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + lane;
    bool look = false;
    if (lane < 16 && b < x.B) {
        const uint2 now = a.first[b], before = x.prev_first[b];
        look = !((now.y & 127u) == 1u && now.x == before.x && now.y == before.y);
    }
    for (unsigned long long m = __ballot(look); m != 0ull; m &= m - 1ull) {  // (uniform)
        const long long bb = b - lane + __builtin_ctzll(m);
        bool differs = false;
        if (lane < a.G) differs = mincr_differs(a.post[(size_t)bb * a.K + lane], x.prev[(size_t)bb * a.G + lane], x.floor);
        const bool changed = __ballot(differs) != 0ull;
        if (lane == 0) {
            if (changed) {
                x.list[atomicAdd(x.state + IS_N, 1u)] = (int)bb;
                atomicAdd((unsigned long long *)(x.state + IS_CALLS), x.pair_ptr  ? 2ull * (unsigned long long)(x.pair_ptr[bb + 1] - x.pair_ptr[bb])
                                                                    : x.rec_ptr ? (unsigned long long)(x.rec_ptr[bb + 1] - x.rec_ptr[bb])
                                                                                : 2ull);
                if (x.changed_map) x.changed_map[bb] = 1;
            } else {
                x.prev_first[bb] = a.first[bb];  // (nothing that matters changed: the next M-step need not look at its rows again)
            }
        }
    }
}

// the changed barcodes, a wavefront each (the wavefronts stride over the list): lane = call, per genotype that differs the difference of
// the new and the old integer contribution, added to the sums with device-scope atomics
template <bool SQUARE>
__global__ __launch_bounds__(256) void k_mincr_delta(MstepArgs a, MIncrArgs x)
{
    if (incr_full(x.state, a)) return;
    const int lane = threadIdx.x & 63;
    const unsigned n = x.state[IS_N];
    const int G = a.G;
    constexpr double MAGIC = 6755399441055744.0;  // (k_mstep_tiles: the same conversion)
    auto quant = [&](float c, int shift) {
        const double d = __builtin_ldexp((double)(SQUARE ? c * c : powf(c, a.power)), shift) + MAGIC;
        return (unsigned long long)(__double_as_longlong(d) - __double_as_longlong(MAGIC));
    };
    // a workgroup per changed barcode (the workgroups stride over the list), its four wavefronts taking the chunks of 64 calls in turn:
    // a launch lasts as long as its longest barcode
    const int wave = threadIdx.x >> 6;
    for (unsigned i = blockIdx.x; i < n; i += gridDim.x) {
        const long long b = x.list[i];
        float now = 0.0f, before = 0.0f;
        if (lane < G) {
            now = a.post[(size_t)b * a.K + lane];
            before = x.prev[(size_t)b * G + lane];
        }
        const unsigned long long mask = __ballot(lane < G && mincr_differs(now, before, x.floor));
        const bool indexed = x.rec_ptr != nullptr;  // (uniform) a variant-sharded rank: the slice's calls of global barcode row b
        const long long p0 = indexed ? x.rec_ptr[b] : x.pair_ptr[b];
        const int n_calls = indexed ? (int)(x.rec_ptr[b + 1] - p0) : 2 * (int)(x.pair_ptr[b + 1] - p0);
        for (int c0 = 64 * wave; c0 < n_calls; c0 += 256) {
            const int ci = c0 + lane;
            const bool mine = ci < n_calls;
            float keep = 0.0f;
            unsigned row = 0u;
            if (mine && indexed) {
                const uint2 d = x.rec[p0 + ci];
                row = d.x;
                keep = __uint_as_float(d.y);
            } else if (mine) {
                keep = x.pairs[p0 + (ci >> 1)].keep[ci & 1];
                // (the compact row array, or - dmx_set_lean_memory has released it - the record's byte offset into the table)
                row = x.call_rows != nullptr ? x.call_rows[2 * p0 + ci] : x.pairs[p0 + (ci >> 1)].row_off[ci & 1] / (4u * (unsigned)G);
                if (x.row_variant != nullptr) row = (unsigned)x.row_variant[row];
            }
            const int shift = mine ? (int)x.shift_v[row] : 0;
            for (unsigned long long m = mask; m != 0ull; m &= m - 1ull) {  // (uniform)
                const int g = __builtin_ctzll(m);
                const float pn = __shfl(now, g), po = __shfl(before, g);
                const unsigned long long qn = quant(pn * keep, shift), qo = quant(po * keep, shift);
                if (mine && qn != qo) {
                    __hip_atomic_fetch_add(&x.acc64[(size_t)row * G + g], qn - qo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    x.touched[row] = 1;
                }
            }
        }
        __syncthreads();  // (every wavefront has read the old row)
        if (wave == 0) {
            if (lane < G) x.prev[(size_t)b * G + lane] = now;
            if (lane == 0) x.prev_first[b] = a.first[b];
        }
    }
}

// Variant-sharded rank: the delta pass as a MASKED WALK of the rank's variant-major records (a wavefront per work item: one variant's
// calls from the barcodes of all ranks).  A lane per record tests its barcode's flag; the flagged records are taken one after the other by
// the whole wavefront, lane = genotype: new and old posterior row (the gathered table / prev), the difference of the two integer
// contributions added to the variant's sums with device-scope atomics.  On converged iterations 1 % of the records are flagged and the
// pass is a read of the records (8 bytes per call of the slice) and of a byte map that stays in the L2.
template <bool SQUARE>
__global__ __launch_bounds__(256) void k_mincr_delta_masked(MstepArgs a, MIncrArgs x)
{
    if (incr_full(x.state, a)) return;
    const int lane = threadIdx.x & 63;
    const int G = a.G;
    auto quant = [&](float c, int shift) { return fixed_of(SQUARE ? c * c : powf(c, a.power), shift); };
    for (long long slot = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); slot < a.n_items; slot += (long long)gridDim.x * 4) {
        const long long item = a.order[slot];
        const int n = a.item_len[item];
        const uint2 *__restrict__ calls = a.calls + a.item_start[item];
        const long long v = a.item_variant[item];
        const int shift = (int)x.shift_v[v];
        bool touched = false;
        for (int c0 = 0; c0 < n; c0 += 64) {
            uint2 d = make_uint2(0u, 0u);
            bool flagged = false;
            if (c0 + lane < n) {
                d = calls[c0 + lane];
                flagged = x.changed_map[d.x] != 0;
            }
            for (unsigned long long m = __ballot(flagged); m != 0ull; m &= m - 1ull) {  // (uniform)
                const int src = __builtin_ctzll(m);
                const unsigned row = (unsigned)__shfl((int)d.x, src);
                const float keep = __uint_as_float((unsigned)__shfl((int)d.y, src));
                if (lane < G) {
                    const float now = a.post[(size_t)row * a.K + lane], before = x.prev[(size_t)row * G + lane];
                    if (mincr_differs(now, before, x.floor)) {
                        const unsigned long long qn = quant(now * keep, shift), qo = quant(before * keep, shift);
                        if (qn != qo) {
                            __hip_atomic_fetch_add(&x.acc64[(size_t)v * G + lane], qn - qo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            touched = true;
                        }
                    }
                }
            }
        }
        if (__ballot(touched) != 0ull && lane == 0) x.touched[v] = 1;
    }
}

// behind the delta pass: the rows of the addition whose sums changed (a thread per variant); behind a full pass: the posteriors and
// codes it summed, for the next M-step to compare with.  Prepares the next M-step's state words.
__global__ __launch_bounds__(256) void k_mincr_finish(MstepArgs a, MIncrArgs x)
{
    const bool full = incr_full(x.state, a);
    const int G = a.G;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    // A workload whose posteriors keep changing (sibling donors, few calls per barcode) pays for the comparison and gets the full pass
    // anyway: after two full passes in a row that the CHANGES asked for, the next MINCR_COOLDOWN M-steps do not look (no valid sums: the
    // full pass, no snapshot behind it), then the sums are kept again (IS_STREAK, IS_SITOUT of the state words).
    const bool too_many = full && x.state[IS_VALID] != 0u && !dense_regime(a);
    const unsigned streak = too_many ? x.state[IS_STREAK] + 1u : 0u;
    const unsigned sitting_out = x.state[IS_SITOUT];
    const bool keep_next = full ? (!dense_regime(a) && streak < 2u && sitting_out <= 1u) : true;  // whether the next M-step finds valid sums
    if (full && keep_next) {
        for (long long i = tid; i < x.B * G; i += stride) {
            const long long b = i / G;
            x.prev[i] = a.post[(size_t)b * a.K + (i - b * G)];
        }
        for (long long b = tid; b < x.B; b += stride) x.prev_first[b] = a.first[b];
    } else if (!full) {
        const int lane = threadIdx.x & 63;
        for (long long v0 = (tid - lane); v0 < x.V; v0 += stride) {  // a wavefront per 64 variants: their flags, then the touched rows, lane = genotype
            const long long v = v0 + lane;
            const bool hit = v < x.V && x.touched[v] != 0;
            if (hit) x.touched[v] = 0;
            for (unsigned long long m = __ballot(hit); m != 0ull; m &= m - 1ull) {
                const long long vv = v0 + __builtin_ctzll(m);
                const int shift = (int)x.shift_v[vv];
                if (lane < G) mstep_out_fixed(a, vv, lane, x.acc64[(size_t)vv * G + lane], shift);
            }
        }
    }
    if (!full && x.changed_map != nullptr) {  // masked walk: the listed barcodes' rows and codes become what the sums now hold
        const unsigned n = x.state[IS_N];
        const int lane = threadIdx.x & 63;
        for (long long i = (tid - lane) / 64; i < (long long)n; i += stride / 64) {  // (a wavefront per listed barcode)
            const long long b = x.list[i];
            if (lane < G) x.prev[(size_t)b * G + lane] = a.post[(size_t)b * a.K + lane];
            if (lane == 0) {
                x.prev_first[b] = a.first[b];
                x.changed_map[b] = 0;
            }
        }
    } else if (full && x.changed_map != nullptr) {  // (flags set by a comparison whose delta pass stood back)
        for (long long b = tid; b < x.B; b += stride) x.changed_map[b] = 0;
    }
    if (tid == 0) {
        x.counters[full ? 0 : 1] += 1u;
        // full passes of the sparse regime's kernels since the state was allocated (not reset by dmx_reset_timings: run_mstep's policy reads
        // it - the tile-major records do nothing for the dense regime's kernel)
        if (full && !dense_regime(a)) x.counters[3] += 1u;
        x.counters[2] = x.state[IS_VALID] ? x.state[IS_N] : 0xFFFFFFFFu;  // barcodes this M-step found changed (no valid sums: not looked for)
        x.next[IS_N] = 0u;
        x.next[IS_CALLS] = x.next[IS_CALLS + 1] = 0u;
        x.next[IS_FORCE] = 0u;
        x.next[IS_VALID] = keep_next ? 1u : 0u;  // (dense regime: k_mstep_dense did the work, the sums are not the tiles')
        x.next[IS_STREAK] = streak >= 2u ? 0u : streak;
        x.next[IS_SITOUT] = streak >= 2u ? MINCR_COOLDOWN : (sitting_out > 0u ? sitting_out - 1u : 0u);
    }
}

// one wavefront per (barcode, 64 genotypes): the bitmap and first-posterior table as the E-step writes them
__global__ __launch_bounds__(256) void k_rebuild_nz(const float *__restrict__ post, long long B, int K, int G,
                                                    float nz_floor, unsigned long long *__restrict__ nz,
                                                    uint2 *__restrict__ first)
{
    const int lane = threadIdx.x & 63;
    const int W = (G + 63) >> 6;
    const long long word = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (word >= B * W) return;
    const long long b = word / W;
    const int g = (int)(word % W) * 64 + lane;
    float p = 0.0f;
    if (g < G) p = post[(size_t)b * K + g];
    const unsigned long long bal = __ballot(g < G && !(p <= nz_floor));
    if (lane == 0) nz[word] = bal;
    if (first && W == 1 && lane == (bal ? __builtin_ctzll(bal) : 0))
        first[b] = nz_code(bal, p);
}

// Sums the item partials of each variant in item order; writes float32 (single GPU) or the float64 total
// that goes into the all-reduce.
// A variant with more calls than the work-item length has several items, and adding their float64 partials is not the
// reference's single left-to-right float64 sum: the two can differ in the last bits, which matters exactly when
// the total sits next to a float32 rounding boundary -- and it often sits ON one (a sum of a few hundred float32
// squares of similar size is an exact float64 whose bits below the float32 precision are ...1000 about once in
// as many entries as it has addends), where a contribution of 1e-20 decides the rounding.  In exact mode (dmx_set_exact_additions; `redo`
// non-null) the combined sum is only accepted when no float32 rounding boundary lies within the error bound of
// either summation order
//     |S_seq - S_comb| <= 2 n u S   (non-negative addends, n calls, u = 2^-53; 4 n u S is used),
// and the (variant, genotype) pair is otherwise queued for k_mstep_exact, which redoes that one sum in order.
constexpr int EXACT_WAVES = 16;
constexpr int EXACT_SPAN = 1024;  // calls per wavefront and segment (LDS: EXACT_WAVES * EXACT_SPAN floats = 64 KB)
// Variants of more calls than this are redone by a whole workgroup (k_mstep_exact), the others by one wavefront each: a rank
// of a variant-sharded run holds 1 / n of the variants in work items 1 / n as long, so that thousands of medium-sized
// variants - not a few dozen hot ones - have sums at a float32 tie (7 289 per M-step at 8 ranks of 200k x 100k x 64).
constexpr long long EXACT_LONG = 8 * EXACT_SPAN;
__global__ __launch_bounds__(256) void k_mcombine(const double *__restrict__ partial,
                                                  const long long *__restrict__ item_ptr,
                                                  const long long *__restrict__ item_start,
                                                  const int *__restrict__ item_len, long long v0, long long v1, int G,
                                                  const int *__restrict__ prow, float *__restrict__ add32,
                                                  double *__restrict__ add64, unsigned long long *__restrict__ redo,
                                                  unsigned *__restrict__ n_redo, unsigned long long redo_cap,
                                                  const int *__restrict__ vlist, bool skip_single,
                                                  const unsigned long long *__restrict__ dense_calls, unsigned long long total_calls,
                                                  bool only_dense, const unsigned char *__restrict__ fixed_shift_v,
                                                  unsigned long long *__restrict__ fixed_acc64, const unsigned *__restrict__ fixed_state)
{
    // only_dense: k_mstep_tiles has written the sums, unless the dense regime's kernel (partial sums per item) took the launch
    const bool dense = dense_calls != nullptr && 4ull * *dense_calls > total_calls;
    if (only_dense && !dense) return;
    // fixed-point work-item form (MstepArgs::fixed_shift_v): the partials are 64-bit integers, unless the dense regime's kernel left
    // float64 ones; when the incremental M-step's delta pass updated the sums (fixed_state) the partials are stale: nothing to combine
    const bool fixed = fixed_shift_v != nullptr && !dense;
    if (fixed && fixed_state != nullptr) {
        const unsigned long long calls = ((unsigned long long)fixed_state[IS_CALLS + 1] << 32) | fixed_state[IS_CALLS];
        const bool full = fixed_state[IS_FORCE] != 0u ? false : (fixed_state[IS_VALID] == 0u || 8ull * calls > total_calls);
        if (!full) return;
    }
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (v1 - v0) * G; i += (long long)gridDim.x * blockDim.x) {
        long long row;
        int g;
        if ((v1 - v0) * G < (1ll << 31)) {  // (uniform) the 64-bit division is a hundred instructions
            row = (unsigned)i / (unsigned)G;
            g = (int)((unsigned)i - (unsigned)row * (unsigned)G);
        } else {
            row = i / G;
            g = (int)(i % G);
        }
        const long long v = vlist ? (long long)vlist[v0 + row] : v0 + row;  // vlist: entries [v0, v1) of a list of variants
        const long long it0 = item_ptr[v], it1 = item_ptr[v + 1];
        if (skip_single && it1 - it0 == 1) continue;  // written by the item's own wavefront (mstep_store)
        if (fixed) {  // integer partial sums: any order, one conversion (as k_mstep_tiles writes a tile's rows); a variant without calls: 0
            unsigned long long q = 0ull;
            for (long long it = it0; it < it1; it++) q += (unsigned long long)__double_as_longlong(partial[(size_t)it * G + g]);
            const long long o = (prow ? (long long)prow[v] : v) * G + g;
            const double sum = __builtin_ldexp((double)(long long)q, -(int)fixed_shift_v[v]);
            if (add64) add64[o] = sum;
            if (add32) add32[o] = (float)sum;
            fixed_acc64[v * G + g] = q;
            continue;
        }
        double s = 0.0;
        for (long long it = it0; it < it1; it++) s += partial[(size_t)it * G + g];
        const long long o = (prow ? (long long)prow[v] : v) * G + g;  // prow: padded rows of the multi-GPU exchange buffer
        if (add64) add64[o] = s;
        if (add32) add32[o] = (float)s;
        if (redo && it1 - it0 > 1 && s > 0.0) {
            const long long n = item_start[it1 - 1] + item_len[it1 - 1] - item_start[it0];
            const double bound = 4.0 * (double)n * 1.1102230246251565e-16 * s;
            const float f = (float)s;
            const double lo = 0.5 * ((double)f + (double)nextafterf(f, 0.0f));  // rounding boundary towards zero
            const double hi = 0.5 * ((double)f + (double)nextafterf(f, __builtin_inff()));
            if (!(s - bound > lo && s + bound < hi)) {
                const unsigned long long entry = ((unsigned long long)v << 16) | (unsigned long long)g;
                if (n > EXACT_LONG) redo[atomicAdd(&n_redo[0], 1u)] = entry;       // a workgroup per sum
                else redo[redo_cap - 1ull - atomicAdd(&n_redo[1], 1u)] = entry;    // a wavefront per sum
            }
        }
    }
}

// The queued sums, redone exactly as the reference does them: all calls of the variant, in order, into ONE float64
// accumulator.  A workgroup of EXACT_WAVES wavefronts per queued (variant, genotype) pair (persistent: the queue
// length is only known on the device).  The variant is walked in segments of EXACT_WAVES x EXACT_SPAN calls: each
// wavefront collects the non-zero contributions of its EXACT_SPAN calls to the genotype in LDS (call order), then
// wavefront 0 adds the lists in order.  The gathers -- the expensive part -- run in parallel; only the additions
// are serial.
template <bool SQUARE>
__global__ __launch_bounds__(64 * EXACT_WAVES) void k_mstep_exact(MstepArgs a, const long long *__restrict__ item_ptr,
                                                                  const unsigned long long *__restrict__ redo,
                                                                  const unsigned *__restrict__ n_redo,
                                                                  const int *__restrict__ prow,
                                                                  float *__restrict__ add32, double *__restrict__ add64)
{
    __shared__ float sh_c[EXACT_WAVES][EXACT_SPAN];
    __shared__ int sh_cnt[EXACT_WAVES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float *mine = sh_c[wave];
    const int W = (a.G + 63) >> 6;
    // calls [lo, hi) of a variant (at most EXACT_SPAN): this wavefront's list of the non-zero contributions to genotype g,
    // in call order; returns their number.
    // The calls as three rounds of independent loads - all records, then all bitmap words, then all posteriors - instead
    // of a chunk-by-chunk chain (the kernel is a chain of load latencies: 0.11 ms -> 0.04 ms on 200k x 100k x 64).
    auto collect = [&](const uint2 *__restrict__ calls, long long lo, long long hi, int g) {
        int cnt = 0;
        constexpr int CH = EXACT_SPAN / 64;
        uint2 d[CH];
        unsigned long long word[CH];
        float p[CH];
#pragma unroll
        for (int j = 0; j < CH; j++) {
            d[j] = make_uint2(0u, 0u);
            if (lo + 64 * j + lane < hi) d[j] = calls[lo + 64 * j + lane];
        }
#pragma unroll
        for (int j = 0; j < CH; j++) {
            word[j] = 0ull;
            if (lo + 64 * j + lane < hi) word[j] = a.nz[(size_t)d[j].x * W + (g >> 6)];
        }
#pragma unroll
        for (int j = 0; j < CH; j++) {
            p[j] = 0.0f;
            if ((word[j] >> (g & 63)) & 1ull) p[j] = a.post[(size_t)d[j].x * a.K + g];
        }
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const bool live = (word[j] >> (g & 63)) & 1ull;
            float c = p[j] * __uint_as_float(d[j].y);
            c = SQUARE ? c * c : powf(c, a.power);
            const unsigned long long bal = __ballot(live);
            if (live) mine[cnt + __popcll(bal & ((1ull << lane) - 1ull))] = c;
            cnt += __popcll(bal);
        }
        return cnt;
    };
    // the list of one wavefront added to ONE accumulator in order (every lane carries the same sum)
    auto add_list = [&](const float *list, int cw, double acc) {
        for (int base = 0; base < cw; base += 64) {
            const float x = base + lane < cw ? list[base + lane] : 0.0f;
            const int m = (cw - base) < 64 ? (cw - base) : 64;
            for (int i = 0; i < m; i++)
                acc += (double)__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), i));
        }
        return acc;
    };
    struct Entry {
        long long v, n;
        int g;
        const uint2 *calls;
    };
    auto entry_of = [&](unsigned long long entry) {
        Entry e;
        e.v = (long long)(entry >> 16);
        e.g = (int)(entry & 0xFFFFull);
        const long long it0 = item_ptr[e.v], it1 = item_ptr[e.v + 1];
        const long long first = a.item_start[it0];
        e.n = a.item_start[it1 - 1] + a.item_len[it1 - 1] - first;
        e.calls = a.calls + first;
        return e;
    };
    auto store = [&](const Entry &e, double acc) {
        const long long o = (prow ? (long long)prow[e.v] : e.v) * a.G + e.g;
        if (add32) add32[o] = (float)acc;
        if (add64) add64[o] = acc;
    };

    // ---- long variants (front of the queue): a workgroup per sum, segments of EXACT_WAVES x EXACT_SPAN calls ----
    const unsigned count_long = n_redo[0];
    for (unsigned q = blockIdx.x; q < count_long; q += gridDim.x) {
        const Entry e = entry_of(redo[q]);
        double acc = 0.0;  // meaningful in wavefront 0
        for (long long seg = 0; seg < e.n; seg += (long long)EXACT_WAVES * EXACT_SPAN) {
            const long long lo = seg + (long long)wave * EXACT_SPAN;
            const long long hi = lo + EXACT_SPAN < e.n ? lo + EXACT_SPAN : e.n;
            const int cnt = collect(e.calls, lo, hi, e.g);
            if (lane == 0) sh_cnt[wave] = cnt;
            __syncthreads();
            if (wave == 0)
                for (int w = 0; w < EXACT_WAVES; w++) acc = add_list(sh_c[w], sh_cnt[w], acc);
            __syncthreads();
        }
        if (threadIdx.x == 0) store(e, acc);
    }
    // ---- the others (back of the queue): a wavefront per sum, its own list in LDS, no workgroup barrier ----
    const unsigned count_short = n_redo[1];
    for (unsigned q = blockIdx.x * EXACT_WAVES + wave; q < count_short; q += gridDim.x * EXACT_WAVES) {
        const Entry e = entry_of(redo[a.redo_cap - 1ull - q]);
        double acc = 0.0;
        for (long long lo = 0; lo < e.n; lo += EXACT_SPAN) {
            const int cnt = collect(e.calls, lo, lo + EXACT_SPAN < e.n ? lo + EXACT_SPAN : e.n, e.g);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the list was written by other lanes of this wavefront
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            acc = add_list(mine, cnt, acc);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // read before the next segment overwrites it
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0) store(e, acc);
    }
}

__global__ __launch_bounds__(256) void k_f64_to_f32(const double *__restrict__ in, float *__restrict__ out, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}

// Multi-GPU: this rank's reduced slice (rows [0, n_rows) of the padded layout, float64 or float32 sums) rounded
// into rows [v_begin, v_begin + n_rows) of the dense float32 addition.
template <typename T>
__global__ __launch_bounds__(256) void k_store_slice(const T *__restrict__ slice, long long v_begin, long long n_rows, int G,
                                                     float *__restrict__ add)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_rows * G) add[v_begin * G + i] = (float)slice[i];
}

// Multi-GPU: the E-step records address the genotype table by byte offset of the variant's row; when the table
// changes from the dense to the padded layout the offsets are rewritten in place (new_rows[v] = padded row of
// variant v; neutral padding calls point at row 0, which stays row 0).
__global__ __launch_bounds__(256) void k_remap_row_offsets(CallPair *__restrict__ pairs, long long n_pairs, unsigned row_bytes,
                                                           const int *__restrict__ new_rows, unsigned *__restrict__ call_rows)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pairs) return;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const unsigned row = (unsigned)new_rows[pairs[i].row_off[h] / row_bytes];
        pairs[i].row_off[h] = row * row_bytes;
        if (call_rows) call_rows[2 * i + h] = row;  // the compact row array of the dictionary form (estep_dict.hip)
    }
}

// Emulated wire: holds the stream for `ticks` of the constant-rate wall clock (one wavefront, asleep most of the time).
__global__ __launch_bounds__(64) void k_delay(long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

hipError_t launch_delay(hipStream_t st, long long ticks)
{
    if (ticks <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, st, ticks);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_f32_to_f64(const float *__restrict__ in, double *__restrict__ out, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (double)in[i];
}

// per-barcode argmax of the posterior (first maximum, like DataFrame.idxmax / np.argmax)
__global__ __launch_bounds__(256) void k_assign(const float *__restrict__ post, long long B, int K,
                                                int *__restrict__ best, float *__restrict__ best_p)
{
    const int lane = threadIdx.x & 63;
    const long long b = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float *row = post + (size_t)b * K;
    float bv = -__builtin_inff();
    int bi = 0x7FFFFFFF;
    for (int k = lane; k < K; k += 64) {
        const float v = row[k];
        if (v > bv) {
            bv = v;
            bi = k;
        }
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float ov = __shfl_xor(bv, off);
        const int oi = __shfl_xor(bi, off);
        if (ov > bv || (ov == bv && oi < bi)) {
            bv = ov;
            bi = oi;
        }
    }
    if (lane == 0) {
        best[b] = bi;
        best_p[b] = bv;
    }
}

// ------------------------------------------------------------------------------------
// self-test kernels for the numpy-exact float32 building blocks
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_test_log(const float *in, float *out, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = npm::log_f32<true>(in[i]);
}

__global__ __launch_bounds__(256) void k_test_log_hot(const float *in, float *out, long long n)
{
    // the packed two-argument form the E-step kernels use; element pairs (2i, 2i+1)
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i >= n) return;
    const bool has2 = i + 1 < n;
    const npm::f32x2 v = {in[i], has2 ? in[i + 1] : 1.0f};
    const npm::f32x2 r = npm::log_f32_hot2(v);
    out[i] = r.x;
    if (has2) out[i + 1] = r.y;
}

// the hardware log2 of the tolerance / guarded modes (v_log_f32), for the exhaustive accuracy check of its mantissa range
__global__ __launch_bounds__(256) void k_test_log2_hw(const float *in, float *out, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_logf(in[i]);
}

__global__ __launch_bounds__(256) void k_test_exp(const float *in, float *out, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = npm::exp_f32(in[i]);
}

// one wave per row, row staged through global memory (in -> out in place semantics)
__global__ __launch_bounds__(64) void k_test_softmax(const float *in, float *out, long long rows, int cols)
{
    const long long r = blockIdx.x;
    const int lane = threadIdx.x;
    const float *x = in + (size_t)r * cols;
    float *y = out + (size_t)r * cols;
    float mx = -__builtin_inff();
    for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, x[c]);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    for (int c = lane; c < cols; c += 64) y[c] = npm::exp_f32(x[c] - mx);
    __threadfence_block();
    __syncthreads();
    const float tot = npm::row_sum_wave(y, cols, lane);
    __syncthreads();
    for (int c = lane; c < cols; c += 64) y[c] = y[c] / tot;
}

// ------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------
static inline unsigned blocks_for(long long n, int per_block) { return (unsigned)((n + per_block - 1) / per_block); }

template <typename T>
static hipError_t launch_pstep(hipStream_t st, const T *prior, const T *addition, const int *v2snp, const int *snp_ptr,
                               const int *snp_vars, long long v_begin, long long n_rows, long long n_snps, int G, const int *prow,
                               float lo, float hi, float *prob, unsigned short *prob16 = nullptr)
{
    if (n_rows * G == 0) return hipSuccess;
    const long long groups = n_snps >= 0 ? n_snps : n_rows;
#define PSTEP(L)                                                                                                          \
    hipLaunchKernelGGL((k_probs_from_betas<T, L>), dim3(blocks_for(groups, 4 * (64 / L))), dim3(256), 0, st, prior, addition, \
                       v2snp, snp_ptr, snp_vars, v_begin, n_rows, n_snps, G, prow, lo, hi, prob, prob16)
    // Lanes per variant row: a lane group walks the chain SNP -> its variants -> their rows -> stores, and what the kernel waits on is
    // that chain (58 VGPRs: 8 wavefronts per SIMD as it is), so wide tables take HALF a row per pass of a lane group's loop - two SNPs in
    // flight per wavefront: 64 genotypes 0.054 -> 0.045 ms on 200 000 variants (a quarter of a row: 0.048), 32 genotypes 0.032 -> 0.028
    // (scripts/pstep_lanes.sh)
    if (G <= 4) PSTEP(4);
    else if (G <= 8) PSTEP(8);
    else if (G <= 16) PSTEP(16);
    else if (G <= 32) PSTEP(16);
    else PSTEP(32);
#undef PSTEP
    return hipGetLastError();
}

// n_snps >= 0: [v_begin, v_begin + n_rows) is the whole table and the SNPs are numbered 0 .. n_snps - 1
hipError_t launch_probs_from_betas(hipStream_t st, const float *prior, const float *addition, const int *v2snp,
                                   const int *snp_ptr, const int *snp_vars, long long v_begin, long long n_rows, long long n_snps,
                                   int G, const int *prow, float lo, float hi, float *prob, unsigned short *prob16)
{
    return launch_pstep<float>(st, prior, addition, v2snp, snp_ptr, snp_vars, v_begin, n_rows, n_snps, G, prow, lo, hi, prob, prob16);
}

hipError_t launch_probs_from_betas_f64(hipStream_t st, const double *betas, const int *v2snp, const int *snp_ptr,
                                       const int *snp_vars, long long V, long long n_snps, int G, const int *prow, float lo, float hi,
                                       float *prob)
{
    return launch_pstep<double>(st, betas, (const double *)nullptr, v2snp, snp_ptr, snp_vars, 0LL, V, n_snps, G, prow, lo, hi, prob);
}

// flags[0] |= 1 when any of the n values is outside [0, 1] or not finite (caller-supplied probability tables)
__global__ __launch_bounds__(256) void k_check_unit_range(const float *__restrict__ x, long long n, int *__restrict__ flags)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool bad = i < n && !(x[i < n ? i : 0] >= 0.0f && x[i < n ? i : 0] <= 1.0f);
    if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flags, 1);
}

hipError_t launch_build_coarse_stream(hipStream_t st, const CallPair *stream, const long long *bin_ptr, long long n_bins, unsigned zero_off,
                                      int cpg, long long *coarse_bin_ptr, unsigned *out, const int *bin_rows, int R, double *log2_keep)
{
    if (n_bins == 0) return hipSuccess;
    hipLaunchKernelGGL(k_coarse_bin_ptr, dim3(1), dim3(1024), 0, st, bin_ptr, n_bins, cpg, coarse_bin_ptr);  // (CoarseShape<CPG>::BPR == CPG)
    static_assert(CoarseShape<1>::BPR == 1 && CoarseShape<2>::BPR == 2 && CoarseShape<4>::BPR == 4, "coarse_batches_per_record");
    if (cpg == 1)
        hipLaunchKernelGGL(k_build_coarse_stream<1>, dim3(blocks_for(n_bins, 4)), dim3(256), 0, st, stream, bin_ptr, coarse_bin_ptr, n_bins, zero_off, out, bin_rows, R, log2_keep);
    else if (cpg == 2)
        hipLaunchKernelGGL(k_build_coarse_stream<2>, dim3(blocks_for(n_bins, 4)), dim3(256), 0, st, stream, bin_ptr, coarse_bin_ptr, n_bins, zero_off, out, bin_rows, R, log2_keep);
    else
        hipLaunchKernelGGL(k_build_coarse_stream<4>, dim3(blocks_for(n_bins, 4)), dim3(256), 0, st, stream, bin_ptr, coarse_bin_ptr, n_bins, zero_off, out, bin_rows, R, log2_keep);
    return hipGetLastError();
}

hipError_t launch_prob_to_half(hipStream_t st, const float *prob, long long rows, int G, unsigned short *out, const unsigned *skip)
{
    const long long n = rows * G;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_prob_to_half, dim3(blocks_for(n, 256)), dim3(256), 0, st, prob, n, out, G, skip);
    return hipGetLastError();
}

hipError_t launch_check_unit_range(hipStream_t st, const float *x, long long n, int *flags)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_check_unit_range, dim3(blocks_for(n, 256)), dim3(256), 0, st, x, n, flags);
    return hipGetLastError();
}

template <int L, int A, int U>
static void launch_direct(hipStream_t st, const EstepArgs &a_in, bool pairs)
{
    EstepArgs a = a_in;
    // the doublet tolerance kernel loads a call's row whole into PairRowShape<A>::GPAD lanes: every table of G (G + 1) / 2
    // options that reaches this A has G <= GPAD; any other option table takes the exact kernel (always admissible)
    const bool fast = a.fast && (!(pairs && L == 64 && A <= 8) || a.G <= PairRowShape<A>::GPAD);
    const bool split = L == 64 && fast && a.segs != nullptr && a.n_segs > 0 && a.order_count == nullptr;
    if (!split) a.segs = nullptr;
    const dim3 grid(blocks_for(split ? a.n_segs + (a.B - a.n_split) : a.B, 4 * (64 / L))), block(256);
    if (fast) {
        if (pairs)
            hipLaunchKernelGGL((k_estep_direct<L, A, true, U, true>), grid, block, 0, st, a);
        else
            hipLaunchKernelGGL((k_estep_direct<L, A, false, U, true>), grid, block, 0, st, a);
        if constexpr (L == 64) {
            if (split) hipLaunchKernelGGL((k_estep_join<A>), dim3(blocks_for(a.n_split, 4)), block, 0, st, a);
        }
    } else {
        if (pairs)
            hipLaunchKernelGGL((k_estep_direct<L, A, true, U, false>), grid, block, 0, st, a);
        else
            hipLaunchKernelGGL((k_estep_direct<L, A, false, U, false>), grid, block, 0, st, a);
    }
}

template <int A>
static hipError_t launch_block(hipStream_t st, const EstepArgs &a, int k_base)
{
    int C = (16384 / (4 * a.G)) & ~7;  // calls staged per chunk: multiple of the 8-call row padding
    C = C < 8 ? 8 : (C > 128 ? 128 : C);
    size_t bytes = (size_t)(C + 2) * a.G * 4 + (size_t)C * 12;
    bytes = (bytes + 15) & ~size_t(15);
    // Tolerance mode (a.fast): with 33 accumulators per thread the running products' registers halved the occupancy and
    // the mode did not pay (344 ms against 299 ms on 130k x 650k x 128 with doublets); with tiles of at most 17 it does
    // (212 ms against 257 ms).
    const void *kernel = a.fast ? (const void *)k_estep_block<A, true> : (const void *)k_estep_block<A, false>;
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    if (a.fast)
        hipLaunchKernelGGL((k_estep_block<A, true>), dim3((unsigned)a.B), dim3(256), bytes, st, a, C, k_base);
    else
        hipLaunchKernelGGL((k_estep_block<A, false>), dim3((unsigned)a.B), dim3(256), bytes, st, a, C, k_base);
    return hipGetLastError();
}

template <int A>
static void launch_tiled(hipStream_t st, const EstepArgs &a)
{
    const dim3 grid(blocks_for(a.n_bins, 4)), block(256);
    if (a.fast && a.prob16 == nullptr && a.coarse_stream != nullptr) {  // the fine pass on the coarse pass's records (the tile-major stream was released)
        if constexpr (A == 2)
            hipLaunchKernelGGL(k_estep_tiled_fine8<1>, grid, block, 0, st, a);
        else if (a.K > 32)
            hipLaunchKernelGGL(k_estep_tiled_fine8<2>, grid, block, 0, st, a);
        else
            hipLaunchKernelGGL(k_estep_tiled_fine8<4>, grid, block, 0, st, a);
        return;
    }
    if constexpr (A == 1) {
        if (a.fast && a.prob16 != nullptr) {  // the coarse pass (guarded mode only: dmx_api.cpp: run_estep)
            if (a.K > 32)
                hipLaunchKernelGGL(k_estep_tiled_coarse<2>, grid, block, 0, st, a);
            else
                hipLaunchKernelGGL(k_estep_tiled_coarse<4>, grid, block, 0, st, a);
            return;
        }
        if (a.fast && a.K <= 32) {  // two calls per gather
            hipLaunchKernelGGL((k_estep_tiled<1, true, true>), grid, block, 0, st, a);
            return;
        }
    }
    if constexpr (A == 2) {
        if (a.fast && a.prob16 != nullptr) {  // the coarse pass for 65 .. 128 genotypes: one call per gather
            hipLaunchKernelGGL(k_estep_tiled_coarse<1>, grid, block, 0, st, a);
            return;
        }
    }
    if (a.fast)
        hipLaunchKernelGGL((k_estep_tiled<A, true>), grid, block, 0, st, a);
    else
        hipLaunchKernelGGL((k_estep_tiled<A, false>), grid, block, 0, st, a);
}

hipError_t launch_estep(hipStream_t st, const EstepArgs &a, bool pairs)
{
    if (a.B == 0) return hipSuccess;
    const int K = a.K;
    // tile-major schedule (built by the repack for large singlet problems).  It pays in the tolerance mode, whose
    // time is the row gathers (1.50 ms against 1.72 ms on 200k x 100k x 64: L2 hit rate 44 % -> 68 %); the exact mode
    // is bound by its arithmetic and only pays the schedule's overhead (2.94 against 2.70 ms), so it keeps one
    // barcode per wavefront unless the schedule is forced (a.tiled == 2: tests).
    if (a.n_bins > 0 && !pairs && K <= 128 && ((a.fast && K > 16) || (a.tiled == 2 && K > 32))) {
        if (K <= 64) launch_tiled<1>(st, a);
        else launch_tiled<2>(st, a);
        return hipGetLastError();
    }
    if (K <= 256) {
        if (K <= 4) launch_direct<4, 1, 4>(st, a, pairs);
        else if (K <= 8) launch_direct<8, 1, 8>(st, a, pairs);
        else if (K <= 16) launch_direct<16, 1, 8>(st, a, pairs);
        else if (K <= 32) launch_direct<32, 1, 8>(st, a, pairs);
        else if (K <= 64) launch_direct<64, 1, 8>(st, a, pairs);
        else if (K <= 128) launch_direct<64, 2, 4>(st, a, pairs);
        else launch_direct<64, 4, 2>(st, a, pairs);
        return hipGetLastError();
    }
    // Doublet tables of more than 512 options (256 in the tolerance mode) already go to the workgroup-per-barcode form
    // below: with 16 accumulators + two row offsets per lane the lane-per-option form is down to 2 (1) waves per SIMD
    // (20k x 20k x 32 with doublets, K = 528: 3.50 -> 2.97 ms, tolerance mode 4.54 -> 1.49 ms; at K = 496 it is still
    // ahead in the exact mode, 2.44 against 2.65 ms, and behind in the tolerance mode, 2.04 against 1.20 ms).
    const bool to_block = pairs && (K > 512 || (a.fast && K > 256));
    if (K <= 1024 && !to_block) {  // register-resident up to 16 options per lane; slots past K are skipped wave-uniformly
        if (K <= 512) launch_direct<64, 8, 2>(st, a, pairs);
        else launch_direct<64, 16, 2>(st, a, pairs);
        return hipGetLastError();
    }
    if (!pairs) return hipErrorInvalidValue;  // K = G > 1024 singlets: not supported (checked by the caller)
    // K > 1024: the options in tiles of up to 17 per thread, one launch of k_estep_block per tile leaving its logits,
    // then the softmax over complete rows.  What this form runs on is registers per thread, i.e. resident wavefronts:
    //   * the softmax fused into a single launch costs 40 VGPRs (3 waves per SIMD instead of 4): 12.2 ms against
    //     10.4 ms on 20k x 20k x 64 with doublets (K = 2080);
    //   * one launch with 33 accumulators per thread (236 VGPRs, 2 waves per SIMD) took 297 ms on 130k x 650k x 128 with
    //     doublets (K = 8256); two launches of 17 take 257 ms although every tile stages the barcode's genotype rows
    //     again; three of 12: 264 ms.
    // (Tiles of 65 accumulators per thread -- 385 VGPRs plus SGPR spills -- ended in GPU memory faults that narrower
    // tiles of the same source do not show: profiles/r2_block_tile65_experiment.txt.)
    if (a.fast && pairs && a.pair_blocks != nullptr && a.n_pair_blocks > 0 && a.order_count == nullptr) {
        EstepArgs prescaled = a;
        prescaled.guard_per_call = GUARD_PER_CALL_PRESCALED;
        // tolerance arithmetic: 2 x 3 blocks of the option triangle, 256 blocks per launch (k_estep_pairblocks)
        const bool big = a.n_pair_blocks >= 1024;         // enough blocks for 512 threads to share the staging of a chunk
        int C = ((big ? 32768 : 16384) / (4 * a.G)) & ~7;  // calls staged per chunk (twice as many for 512 threads: 69.3 -> 67.9 ms at K = 8256)
        C = C < 8 ? 8 : (C > 128 ? 128 : C);
        size_t bytes = (size_t)(C + 2) * a.G * 4 + (size_t)C * 12;
        bytes = (bytes + 15) & ~size_t(15);
        // 512 threads share the staging of a chunk where there are blocks for them (K = 8256: 1 450 blocks, 74.7 -> 69.7 ms;
        // K = 528: 121 blocks, 1.24 ms with 256 threads against 1.84); 1024 threads: 103 ms
        auto launch = [&](auto kernel, int threads) {
            const hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            if (e != hipSuccess) return e;
            for (int blk_base = 0; blk_base < a.n_pair_blocks; blk_base += threads)
                hipLaunchKernelGGL(kernel, dim3((unsigned)a.B), dim3(threads), bytes, st, prescaled, C, blk_base);
            return hipSuccess;
        };
        const hipError_t e = big ? launch(k_estep_pairblocks<PAIRBLOCK_R1, PAIRBLOCK_R2, 512>, 512)
                                                     : launch(k_estep_pairblocks<PAIRBLOCK_R1, PAIRBLOCK_R2, 256>, 256);
        if (e != hipSuccess) return e;
        return launch_softmax_rows(st, prescaled);
    }
    const int need = (K + 255) / 256;
    int tile = need <= 2 ? 2 : need <= 4 ? 4 : need <= 6 ? 6 : need <= 8 ? 8 : need <= 12 ? 12 : need <= 17 ? 17 : need <= 24 ? 12 : 17;
    // the tolerance mode carries a running product and an exponent per option besides the accumulator: tiles of 6
    // keep it at 4+ waves per SIMD (K = 8256: 170 ms with tiles of 17, 103 with 12, 87 with 6, 98 with 4; the exact
    // mode does not care: 214 / 220 / 218 / 230 ms)
    if (a.fast && need > 6) tile = 6;
    for (int k_base = 0; k_base < K; k_base += tile * 256) {
        const hipError_t e = tile == 2 ? launch_block<2>(st, a, k_base) : tile == 4 ? launch_block<4>(st, a, k_base) : tile == 6 ? launch_block<6>(st, a, k_base) : tile == 8 ? launch_block<8>(st, a, k_base)
                           : tile == 12 ? launch_block<12>(st, a, k_base) : launch_block<17>(st, a, k_base);
        if (e != hipSuccess) return e;
    }
    return launch_softmax_rows(st, a);
}

// softmax, bitmaps and barcode codes of complete logit rows (after the option-tile launches of the workgroup-per-barcode forms)
hipError_t launch_softmax_rows(hipStream_t st, const EstepArgs &a)
{
    if (a.B == 0) return hipSuccess;
    const size_t row_bytes = ((size_t)a.K + (size_t)a.sum_plan_values) * 4;  // the row + the values of the sum plan
    if (row_bytes <= 150 * 1024 && a.sum_plan) {
        const hipError_t e = hipFuncSetAttribute((const void *)k_softmax_rows<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)row_bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_softmax_rows<true>, dim3((unsigned)a.B), dim3(256), row_bytes, st, a);
    } else {
        hipLaunchKernelGGL(k_softmax_rows<false>, dim3((unsigned)a.B), dim3(256), 0, st, a);
    }
    return hipGetLastError();
}

template <int A, int U>
static void launch_m(hipStream_t st, const MstepArgs &a)
{
    const dim3 grid(blocks_for(a.n_items, 4));
    const bool small = a.post_bytes < (1ull << 32);
    if (a.square && small)
        hipLaunchKernelGGL((k_mstep<A, U, true, true>), grid, dim3(256), 0, st, a);
    else if (a.square)
        hipLaunchKernelGGL((k_mstep<A, U, true, false>), grid, dim3(256), 0, st, a);
    else if (small)
        hipLaunchKernelGGL((k_mstep<A, U, false, true>), grid, dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((k_mstep<A, U, false, false>), grid, dim3(256), 0, st, a);
}

hipError_t launch_mstep(hipStream_t st, const MstepArgs &a)
{
    if (a.n_items == 0) return hipSuccess;
    const int G = a.G;
    if (G <= 64) {
        // call-parallel form: 16-entry queues (8 waves per SIMD), dense rows fetched 4 at a time
        const dim3 grid(blocks_for(a.n_items, 4));
        // 32-bit offsets: posterior table below 4 GiB, barcode index below 2^24 (v_mul_u32_u24), row below 2^24 bytes
        const bool buf = !a.wide && a.post_bytes < (1ull << 32) && a.first_bytes < (8ull << 24) && a.K < (1 << 22);
#define MSTEP_CALLS(SQ)                                                                                            \
    do {                                                                                                           \
        if (a.fixed_shift_v != nullptr) {                                                                          \
            if (buf) hipLaunchKernelGGL((k_mstep_calls<SQ, 16, 4, true, true>), grid, dim3(256), 0, st, a);        \
            else hipLaunchKernelGGL((k_mstep_calls<SQ, 16, 4, false, true>), grid, dim3(256), 0, st, a);           \
        } else if (buf) hipLaunchKernelGGL((k_mstep_calls<SQ, 16, 4, true>), grid, dim3(256), 0, st, a);           \
        else hipLaunchKernelGGL((k_mstep_calls<SQ, 16, 4, false>), grid, dim3(256), 0, st, a);                     \
    } while (0)
        if (a.square)
            MSTEP_CALLS(true);
        else
            MSTEP_CALLS(false);
#undef MSTEP_CALLS
        if (a.dense_calls) {  // the dense regime's kernel; exactly one of the two does the work (dense_regime)
            const dim3 dense_grid(std::min(blocks_for(a.n_items, 4), 2048u));
            if (a.square)
                hipLaunchKernelGGL((k_mstep_dense<true>), dense_grid, dim3(256), 0, st, a);
            else
                hipLaunchKernelGGL((k_mstep_dense<false>), dense_grid, dim3(256), 0, st, a);
        }
        return hipGetLastError();
    }
    if (G <= 128) launch_m<2, 4>(st, a);
    else if (G <= 256) launch_m<4, 2>(st, a);
    else if (G <= 512) launch_m<8, 2>(st, a);
    else if (G <= 1024) launch_m<16, 2>(st, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// changes -> delta pass | full pass (exactly one of them works) -> conversion / snapshot; the dense regime's kernel as in launch_mstep_tiles
hipError_t launch_mstep_incremental(hipStream_t st, const MstepArgs &a, const MTileArgs &t, const MIncrArgs &x)
{
    if (t.n_tiles == 0 || x.B == 0) return hipSuccess;
    hipLaunchKernelGGL(k_mincr_changes, dim3(blocks_for(x.B, 64)), dim3(256), 0, st, a, x);
    if (a.square)
        hipLaunchKernelGGL((k_mincr_delta<true>), dim3(4096), dim3(256), 0, st, a, x);
    else
        hipLaunchKernelGGL((k_mincr_delta<false>), dim3(4096), dim3(256), 0, st, a, x);
    const hipError_t e = launch_mstep_tiles(st, a, t);  // (stands back unless the full pass is due; + the dense regime's kernel)
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_mincr_finish, dim3(2048), dim3(256), 0, st, a, x);
    return hipGetLastError();
}

// the same sequence with the fixed-point work-item form as the full pass (no tile-major records): changes -> delta pass | the items'
// full pass (exactly one of them works; k_mcombine, launched by the caller behind this, stands back with it) -> conversion / snapshot
hipError_t launch_mstep_items_incremental(hipStream_t st, const MstepArgs &a, const MIncrArgs &x)
{
    if (a.n_items == 0 || x.B == 0 || a.fixed_shift_v == nullptr) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_mincr_changes, dim3(blocks_for(x.B, 64)), dim3(256), 0, st, a, x);
    if (a.square)
        hipLaunchKernelGGL((k_mincr_delta<true>), dim3(4096), dim3(256), 0, st, a, x);
    else
        hipLaunchKernelGGL((k_mincr_delta<false>), dim3(4096), dim3(256), 0, st, a, x);
    const hipError_t e = launch_mstep(st, a);  // (stands back unless the full pass is due; + the dense regime's kernel)
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_mincr_finish, dim3(2048), dim3(256), 0, st, a, x);
    return hipGetLastError();
}

hipError_t launch_mstep_incremental_sharded(hipStream_t st, const MstepArgs &a, const MTileArgs &t, const MIncrArgs &x)
{
    if (t.n_tiles == 0 || x.B == 0 || (x.changed_map == nullptr && x.rec_ptr == nullptr)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_mincr_changes, dim3(blocks_for(x.B, 64)), dim3(256), 0, st, a, x);
    const dim3 grid(std::min(blocks_for(a.n_items, 4), 8192u));
    if (x.rec_ptr != nullptr) {  // the slice's records by barcode row: the changed barcodes' calls only
        if (a.square)
            hipLaunchKernelGGL((k_mincr_delta<true>), dim3(4096), dim3(256), 0, st, a, x);
        else
            hipLaunchKernelGGL((k_mincr_delta<false>), dim3(4096), dim3(256), 0, st, a, x);
    } else if (a.n_items) {
        if (a.square)
            hipLaunchKernelGGL((k_mincr_delta_masked<true>), grid, dim3(256), 0, st, a, x);
        else
            hipLaunchKernelGGL((k_mincr_delta_masked<false>), grid, dim3(256), 0, st, a, x);
    }
    const hipError_t e = launch_mstep_tiles(st, a, t);  // (stands back unless the full pass is due; + the dense regime's kernel)
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_mincr_finish, dim3(2048), dim3(256), 0, st, a, x);
    return hipGetLastError();
}

hipError_t launch_mstep_tiles(hipStream_t st, const MstepArgs &a, const MTileArgs &t)
{
    if (t.n_tiles == 0) return hipSuccess;
    const size_t lds = (size_t)t.tv * a.G * sizeof(unsigned long long);
    if (a.square)
        hipLaunchKernelGGL((k_mstep_tiles<true>), dim3((unsigned)t.n_tiles), dim3(MTILE_THREADS), lds, st, a, t);
    else
        hipLaunchKernelGGL((k_mstep_tiles<false>), dim3((unsigned)t.n_tiles), dim3(MTILE_THREADS), lds, st, a, t);
    if (a.dense_calls && a.n_items) {  // the dense regime's kernel; exactly one of the two does the work (dense_regime)
        const dim3 grid(std::min(blocks_for(a.n_items, 4), 2048u));
        if (a.square)
            hipLaunchKernelGGL((k_mstep_dense<true>), grid, dim3(256), 0, st, a);
        else
            hipLaunchKernelGGL((k_mstep_dense<false>), grid, dim3(256), 0, st, a);
    }
    return hipGetLastError();
}

hipError_t launch_mcombine(hipStream_t st, const MstepArgs &a, const long long *item_ptr, long long v0, long long v1,
                           const int *prow, float *add32, double *add64, unsigned long long *redo, unsigned *n_redo, const int *vlist,
                           bool skip_single)
{
    if ((v1 - v0) * a.G <= 0) return hipSuccess;
    if (redo) {
        const hipError_t e = hipMemsetAsync(n_redo, 0, 2 * sizeof(unsigned), st);
        if (e != hipSuccess) return e;
    }
    // (after k_mstep_tiles the pass only acts in the dense regime: a small grid that strides, so that standing back costs nothing)
    const unsigned full_grid = blocks_for((v1 - v0) * a.G, 256);
    hipLaunchKernelGGL(k_mcombine, dim3(a.tiles_done ? std::min(full_grid, 1024u) : full_grid), dim3(256), 0, st, a.partial, item_ptr,
                       a.item_start, a.item_len, v0, v1, a.G, prow, add32, add64, redo, n_redo, a.redo_cap, vlist, skip_single,
                       a.dense_calls, a.total_calls, a.tiles_done, a.fixed_shift_v, a.fixed_acc64, a.fixed_state);
    if (!redo) return hipGetLastError();
    // exact mode: the sums that must be redone in the reference's order (see k_mcombine)
    const dim3 grid(512), block(64 * EXACT_WAVES);
    if (a.square)
        hipLaunchKernelGGL((k_mstep_exact<true>), grid, block, 0, st, a, item_ptr, redo, n_redo, prow, add32, add64);
    else
        hipLaunchKernelGGL((k_mstep_exact<false>), grid, block, 0, st, a, item_ptr, redo, n_redo, prow, add32, add64);
    return hipGetLastError();
}

// ---- compact exchange of the posterior rows (variant-sharded M-step, G <= 64; dmx_exchange.cpp: gather_posteriors) ----
// What the M-step reads of a barcode with ONE live posterior is in its 8-byte code (nz_code); its 256-byte row need not travel: the
// receivers rebuild it - the code's posterior at the code's genotype, zeros elsewhere; a posterior that is not live contributes
// exactly +0 to every sum (NZ_FLOOR_SQUARE; with another contribution_power "live" means non-zero), so the additions keep their bits.
// Rows with several live posteriors (1 - 15 % of the barcodes) travel in a list: block = {rows listed (beyond `cap`: overflow - the
// caller falls back to the all-gather of the whole table), 3 words of padding, cap entries of (row, G floats)}.
// (peers, emulated wire only: nobody fills the other ranks' blocks; they list nothing, wherever this exchange's block size puts their
// headers.  A "last workgroup moves the count into the header" scheme instead of the header's memset was tried: one same-address
// device-scope atomic per workgroup, 25 - 35 ns each when they arrive together - 115 us for the 3 125 workgroups of a table slice.)
__device__ __forceinline__ void clear_peer_headers(unsigned *__restrict__ peers, unsigned long long block_words, int nranks, int own)
{
    if (peers == nullptr || blockIdx.x != 0 || threadIdx.x != 0) return;
    for (int r = 0; r < nranks; r++)
        if (r != own) peers[(size_t)r * block_words] = 0u;
}

// sent / sent_multi: what the receivers hold of this rank's rows with several live posteriors - a row is listed only when it differs from
// that (17.8 % of the barcodes of the converged 200k x 100k x 64 experiment have several live posteriors, and next to none of them moves);
// a row the receivers rebuild from its code is no longer described by what was sent.  16 rows per wavefront: the row comparisons of a
// wavefront are one after the other.
__global__ __launch_bounds__(256) void k_post_compact_build(const uint2 *__restrict__ first, const float *__restrict__ post, long long B, int G,
                                                            unsigned cap, unsigned *__restrict__ block, float *__restrict__ sent,
                                                            unsigned char *__restrict__ sent_multi, unsigned *__restrict__ peers,
                                                            unsigned long long block_words, int nranks, int own)
{
    clear_peer_headers(peers, block_words, nranks, own);
    const int lane = threadIdx.x & 63;
    const long long b0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    const long long b = b0 + lane;
    const bool in = lane < 16 && b < B;
    const bool multi = in && (first[in ? b : 0].y & 127u) != 1u;
    const bool was = in && sent_multi[in ? b : 0] != 0;
    if (in && !multi && was) sent_multi[b] = 0;
    const unsigned long long m = __ballot(multi), was_m = __ballot(multi && was);
    if (!m) return;
    unsigned long long dm = 0ull;  // the rows to list (uniform)
    for (unsigned long long mm = m; mm != 0ull; mm &= mm - 1ull) {
        const int src = __builtin_ctzll(mm);
        bool same = false;
        if ((was_m >> src) & 1ull) {
            const size_t o = (size_t)(b0 + src) * G + lane;
            const bool differs = lane < G && __float_as_uint(post[lane < G ? o : 0]) != __float_as_uint(sent[lane < G ? o : 0]);
            same = __ballot(differs) == 0ull;
        }
        if (!same) dm |= 1ull << src;
    }
    if (!dm) return;
    unsigned base = 0;
    if (lane == 0) base = atomicAdd(&block[0], (unsigned)__popcll(dm));
    base = (unsigned)__shfl((int)base, 0);
    for (unsigned i = 0; dm != 0ull; dm &= dm - 1ull, i++) {  // (uniform)
        const long long row = b0 + __builtin_ctzll(dm);
        const unsigned at = base + i;
        if (at >= cap) break;  // (overflow: the whole table travels, and `sent` becomes a copy of it)
        unsigned *e = block + 4 + (size_t)at * (size_t)(1 + G);
        if (lane == 0) {
            e[0] = (unsigned)row;
            sent_multi[row] = 1;
        }
        if (lane < G) {
            const float v = post[(size_t)row * G + lane];
            e[1 + lane] = __float_as_uint(v);
            sent[(size_t)row * G + lane] = v;
        }
    }
}

// the other ranks' rows: a wavefront per row with at most one live posterior (from its code), then a wavefront per listed row
// (`seen`: the code every row of post_g was last rebuilt from - a row whose code has not changed since is what it should be already:
// 99 % of the barcodes of a converged experiment keep a posterior of exactly 1.0 for the same donor; 0xFF..: unknown.)
// A lane per row compares; the rows that changed are rebuilt by the whole wavefront, one after the other.
__global__ __launch_bounds__(256) void k_post_reconstruct(const uint2 *__restrict__ first_g, float *__restrict__ post_g, const unsigned *__restrict__ blocks,
                                                          unsigned long long block_words, long long rows_pad, int G, int nranks, int own, unsigned cap,
                                                          uint2 *__restrict__ seen)
{
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long n_rows = rows_pad * nranks, row_waves = (n_rows + 63) / 64;
    if (wave < row_waves) {
        const long long row = wave * 64 + lane;
        bool rebuild = false;
        uint2 code = make_uint2(0u, 0u);
        if (row < n_rows && row / rows_pad != own) {
            code = first_g[row];
            const uint2 before = seen[row];
            const bool single = (code.y & 127u) <= 1u;
            rebuild = single && !(code.x == before.x && code.y == before.y);
            if (rebuild) seen[row] = code;
            else if (!single) seen[row] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);  // (a listed row: rewritten below, every time)
        }
        for (unsigned long long m = __ballot(rebuild); m != 0ull; m &= m - 1ull) {  // (uniform)
            const int src = __builtin_ctzll(m);
            const unsigned cx = (unsigned)__shfl((int)code.x, src), cy = (unsigned)__shfl((int)code.y, src);
            if (lane < G)
                post_g[(size_t)(wave * 64 + src) * G + lane] = ((cy & 127u) == 1u && lane == (int)((cy >> 7) & 63u)) ? __uint_as_float(cx) : 0.0f;
        }
        return;
    }
    const long long e_id = wave - row_waves;
    const long long r = e_id / cap;
    const unsigned at = (unsigned)(e_id - r * cap);
    if (r >= nranks || r == own) return;
    const unsigned *block = blocks + (size_t)r * block_words;
    if (at >= block[0]) return;
    const unsigned *e = block + 4 + (size_t)at * (size_t)(1 + G);
    const long long row = (long long)e[0];
    if (row < rows_pad && lane < G) post_g[((size_t)r * rows_pad + row) * G + lane] = __uint_as_float(e[1 + lane]);
}

// ---- compact exchange of the genotype table (sliced P-step; dmx_steps.cpp: run_pstep) ----
// Between two EM iterations most rows of genotype_prob keep their bits (27 % change at the second iteration of the 200k x 100k x 64
// experiment, 6 % at the third, 1 % at the fifth).  A rank lists the rows of ITS slice that differ from what it sent last (`prev`,
// brought up to date here) - block = {rows listed (beyond `cap`: overflow - the whole slices travel), 3 words of padding, cap entries of
// (row in the slice, G floats)} - and the receivers, whose copies of the slice are what was sent last, write the listed rows.
__global__ __launch_bounds__(256) void k_prob_changes_build(const float *__restrict__ slice, float *__restrict__ prev, long long rows, int G,
                                                            unsigned cap, unsigned *__restrict__ block, unsigned *__restrict__ peers,
                                                            unsigned long long block_words, int nranks, int own)
{
    clear_peer_headers(peers, block_words, nranks, own);
    const int lane = threadIdx.x & 63;
    const int W = (G + 63) >> 6;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    bool differs = false;
    for (int s = 0; s < W; s++) {
        const int g = lane + 64 * s;
        if (g < G) differs = differs || __float_as_uint(slice[(size_t)row * G + g]) != __float_as_uint(prev[(size_t)row * G + g]);
    }
    if (__ballot(differs) == 0ull) return;  // (uniform)
    unsigned at = 0;
    if (lane == 0) at = atomicAdd(&block[0], 1u);
    at = (unsigned)__shfl((int)at, 0);
    unsigned *e = at < cap ? block + 4 + (size_t)at * (size_t)(1 + G) : nullptr;
    if (e != nullptr && lane == 0) e[0] = (unsigned)row;
    for (int s = 0; s < W; s++) {
        const int g = lane + 64 * s;
        if (g < G) {
            const float v = slice[(size_t)row * G + g];
            prev[(size_t)row * G + g] = v;
            if (e != nullptr) e[1 + g] = __float_as_uint(v);
        }
    }
}

// the other ranks' listed rows into this rank's copy of their slices (a wavefront per entry)
// (table16, nullable: the binary16 table of the coarse pass, kept up to date row by row instead of converted as a whole behind the exchange)
__global__ __launch_bounds__(256) void k_prob_changes_apply(float *__restrict__ table, const unsigned *__restrict__ blocks, unsigned long long block_words,
                                                            long long slice_rows, int G, int nranks, int own, unsigned cap, unsigned short *__restrict__ table16)
{
    const int lane = threadIdx.x & 63;
    const long long e_id = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long r = e_id / cap;
    const unsigned at = (unsigned)(e_id - r * cap);
    if (r >= nranks || r == own) return;
    const unsigned *block = blocks + (size_t)r * block_words;
    if (at >= block[0]) return;
    const unsigned *e = block + 4 + (size_t)at * (size_t)(1 + G);
    const long long row = (long long)e[0];
    if (row >= slice_rows) return;
    for (int g = lane; g < G; g += 64) {
        const float v = __uint_as_float(e[1 + g]);
        table[((size_t)r * slice_rows + row) * G + g] = v;
        if (table16 != nullptr) table16[((size_t)r * slice_rows + row) * (size_t)(2 * G) + g] = __builtin_bit_cast(unsigned short, (_Float16)v);
    }
}

hipError_t launch_prob_changes_build(hipStream_t st, const float *slice, float *prev, long long rows, int G, unsigned cap, unsigned *block,
                                     unsigned *peers, unsigned long long block_words, int nranks, int own)
{
    hipError_t e = hipMemsetAsync(block, 0, 4 * sizeof(unsigned), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_prob_changes_build, dim3(blocks_for(rows > 0 ? rows : 1, 4)), dim3(256), 0, st, slice, prev, rows, G, cap, block, peers, block_words,
                       nranks, own);
    return hipGetLastError();
}

hipError_t launch_prob_changes_apply(hipStream_t st, float *table, const unsigned *blocks, unsigned long long block_words, long long slice_rows, int G,
                                     int nranks, int own, unsigned cap, unsigned short *table16)
{
    const long long waves = (long long)cap * nranks;
    if (waves == 0) return hipSuccess;
    hipLaunchKernelGGL(k_prob_changes_apply, dim3(blocks_for(waves, 4)), dim3(256), 0, st, table, blocks, block_words, slice_rows, G, nranks, own, cap, table16);
    return hipGetLastError();
}

// the lists' lengths of all ranks into host-visible memory (one small kernel: a strided 4-byte copy per rank cost 70 us of runtime overhead),
// then `seq` behind them (system-scope release): the host polls that word instead of synchronising with the stream (dmx_exchange.cpp: wait_counts)
__global__ void k_post_counts(const unsigned *__restrict__ blocks, unsigned long long block_words, int nranks, unsigned *__restrict__ out, unsigned seq)
{
    if ((int)threadIdx.x < nranks) out[threadIdx.x] = blocks[(size_t)threadIdx.x * block_words];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(out + nranks, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

hipError_t launch_post_counts(hipStream_t st, const unsigned *blocks, unsigned long long block_words, int nranks, unsigned *out_host_visible, unsigned seq)
{
    if (nranks > 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_post_counts, dim3(1), dim3(nranks <= 64 ? 64 : 1024), 0, st, blocks, block_words, nranks, out_host_visible, seq);
    return hipGetLastError();
}

hipError_t launch_post_compact_build(hipStream_t st, const uint2 *first, const float *post, long long B, int G, unsigned cap, unsigned *block,
                                     float *sent, unsigned char *sent_multi, unsigned *peers, unsigned long long block_words, int nranks, int own)
{
    hipError_t e = hipMemsetAsync(block, 0, 4 * sizeof(unsigned), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_post_compact_build, dim3(blocks_for(B > 0 ? B : 1, 64)), dim3(256), 0, st, first, post, B, G, cap, block, sent, sent_multi, peers,
                       block_words, nranks, own);
    return hipGetLastError();
}

hipError_t launch_post_reconstruct(hipStream_t st, const uint2 *first_g, float *post_g, const unsigned *blocks, unsigned long long block_words,
                                   long long rows_pad, int G, int nranks, int own, unsigned cap, uint2 *seen)
{
    const long long waves = (rows_pad * nranks + 63) / 64 + (long long)cap * nranks;
    if (waves == 0) return hipSuccess;
    hipLaunchKernelGGL(k_post_reconstruct, dim3(blocks_for(waves, 4)), dim3(256), 0, st, first_g, post_g, blocks, block_words, rows_pad, G, nranks, own, cap, seen);
    return hipGetLastError();
}

hipError_t launch_store_slice(hipStream_t st, const void *slice, bool f64, long long v_begin, long long n_rows, int G, float *add)
{
    if (n_rows * G == 0) return hipSuccess;
    if (f64)
        hipLaunchKernelGGL(k_store_slice<double>, dim3(blocks_for(n_rows * G, 256)), dim3(256), 0, st, (const double *)slice, v_begin, n_rows, G, add);
    else
        hipLaunchKernelGGL(k_store_slice<float>, dim3(blocks_for(n_rows * G, 256)), dim3(256), 0, st, (const float *)slice, v_begin, n_rows, G, add);
    return hipGetLastError();
}

hipError_t launch_remap_row_offsets(hipStream_t st, CallPair *pairs, long long n_pairs, unsigned row_bytes, const int *new_rows,
                                    unsigned *call_rows)
{
    if (n_pairs == 0) return hipSuccess;
    hipLaunchKernelGGL(k_remap_row_offsets, dim3(blocks_for(n_pairs, 256)), dim3(256), 0, st, pairs, n_pairs, row_bytes, new_rows, call_rows);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_add_f32(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ out, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}

hipError_t launch_add_f32(hipStream_t st, const float *a, const float *b, float *out, long long n)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_add_f32, dim3(blocks_for(n, 256)), dim3(256), 0, st, a, b, out, n);
    return hipGetLastError();
}

hipError_t launch_f64_to_f32(hipStream_t st, const double *in, float *out, long long n)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_f64_to_f32, dim3(blocks_for(n, 256)), dim3(256), 0, st, in, out, n);
    return hipGetLastError();
}

hipError_t launch_f32_to_f64(hipStream_t st, const float *in, double *out, long long n)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_f32_to_f64, dim3(blocks_for(n, 256)), dim3(256), 0, st, in, out, n);
    return hipGetLastError();
}

hipError_t launch_prior_betas(hipStream_t st, const float *betas, float *bsum, const unsigned long long *n_mol,
                              const int *v2snp, const int *snp_ptr, const int *snp_vars, long long V, int G,
                              double default_prior, float *out)
{
    if (V * G == 0) return hipSuccess;
    hipLaunchKernelGGL(k_beta_rowsum, dim3(blocks_for(V, 4)), dim3(256), 0, st, betas, V, G, bsum);
    hipLaunchKernelGGL(k_prior_betas, dim3(blocks_for(V * G, 256)), dim3(256), 0, st, betas, bsum, n_mol, v2snp, snp_ptr,
                       snp_vars, V, G, default_prior, out);
    return hipGetLastError();
}

hipError_t launch_rebuild_nz(hipStream_t st, const float *post, long long B, int K, int G, float nz_floor,
                             unsigned long long *nz, uint2 *first)
{
    const long long words = B * ((G + 63) / 64);
    if (words == 0) return hipSuccess;
    hipLaunchKernelGGL(k_rebuild_nz, dim3(blocks_for(words, 4)), dim3(256), 0, st, post, B, K, G, nz_floor, nz, first);
    return hipGetLastError();
}

hipError_t launch_assign(hipStream_t st, const float *post, long long B, int K, int *best, float *best_p)
{
    if (B == 0) return hipSuccess;
    hipLaunchKernelGGL(k_assign, dim3(blocks_for(B, 4)), dim3(256), 0, st, post, B, K, best, best_p);
    return hipGetLastError();
}

hipError_t launch_test_log(hipStream_t st, const float *in, float *out, long long n)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_test_log, dim3(blocks_for(n, 256)), dim3(256), 0, st, in, out, n);
    return hipGetLastError();
}

hipError_t launch_test_log_hot(hipStream_t st, const float *in, float *out, long long n)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_test_log_hot, dim3(blocks_for(n, 512)), dim3(256), 0, st, in, out, n);
    return hipGetLastError();
}

hipError_t launch_test_log2_hw(hipStream_t st, const float *in, float *out, long long n)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_test_log2_hw, dim3(blocks_for(n, 256)), dim3(256), 0, st, in, out, n);
    return hipGetLastError();
}

hipError_t launch_test_exp(hipStream_t st, const float *in, float *out, long long n)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_test_exp, dim3(blocks_for(n, 256)), dim3(256), 0, st, in, out, n);
    return hipGetLastError();
}

hipError_t launch_test_softmax(hipStream_t st, const float *in, float *out, long long rows, int cols)
{
    if (rows == 0) return hipSuccess;
    hipLaunchKernelGGL(k_test_softmax, dim3((unsigned)rows), dim3(64), 0, st, in, out, rows, cols);
    return hipGetLastError();
}

}  // namespace dmx
