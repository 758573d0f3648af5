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

#include <algorithm>
#include <cstdlib>
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
// ROWS rows per wavefront, their loads issued together (one row per wavefront left the launch bound by the rate at
// which wavefronts start and by one load latency each: 0.13 ms for a 51 MB table).
template <int A, int ROWS>
__global__ __launch_bounds__(256) void k_build_dict(const float *__restrict__ prob, long long rows, int G, int pitch,
                                                    float *__restrict__ dict, unsigned char *__restrict__ codes,
                                                    unsigned *__restrict__ stat)
{
    const int lane = threadIdx.x & 63;
    const long long r0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS;
    if (r0 >= rows) return;  // wave-uniform
    unsigned xs[ROWS][A];
#pragma unroll
    for (int i = 0; i < ROWS; i++) {
        const long long r = r0 + i < rows ? r0 + i : rows - 1;
#pragma unroll
        for (int s = 0; s < A; s++) {
            const int g = lane + 64 * s;
            xs[i][s] = g < G ? __float_as_uint(prob[(size_t)r * G + g]) : 0u;
        }
    }
    unsigned worst = 0;
#pragma unroll
    for (int i = 0; i < ROWS; i++) {
        const long long r = r0 + i;
        if (r >= rows) break;  // wave-uniform
        unsigned code[A];
        unsigned long long rem[A];
#pragma unroll
        for (int s = 0; s < A; s++) {
            rem[s] = __ballot(lane + 64 * s < G);
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
                    val = (unsigned)__builtin_amdgcn_readlane((int)xs[i][s], __builtin_ctzll(rem[s]));
                    found = true;
                }
            }
            if (!found) break;
#pragma unroll
            for (int s = 0; s < A; s++) {
                const unsigned long long m = __ballot(xs[i][s] == val) & rem[s];
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
            if (g < pitch) codes[(size_t)r * pitch + g] = (unsigned char)(g < G ? code[s] * 8u : 0u);
        }
        worst = max(worst, overflow ? (unsigned)DICT_CAP + 1u : (unsigned)d);
    }
    // one shared maximum: after the first rows it is at its final value and nobody writes any more
    if (lane == 0 && worst > __hip_atomic_load(stat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(stat, worst);
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

// The table the E-step kernel reads: one row per row of genotype_prob,
//     [DW float32 dictionary values][LB lanes x CBY bytes: the codes of the lane's four options, CBITS bits each]
// padded to a multiple of 16 bytes - 32 bytes per row for 64 genotypes with <= 4 values, so that 200 000 rows are
// 6.4 MB where genotype_prob is 51 MB: the row gathers of the direct form cross the fabric for half of their 128-byte
// lines (no L2 holds 51 MB), and so did a byte per option (12.8 MB) + 32 bytes of values (6.4 MB) per row.
//   NE = 4:  DW 4, 2-bit codes (the option's value);  NE = 8: DW 8, 4-bit codes;
//   NE = 16 (doublets): DW 4, 4-bit codes = pair_entry of the option's two genotypes
template <int NE>
struct DictRow {
    static constexpr int DW = NE == 8 ? 8 : 4;
    static constexpr int CBITS = NE == 4 ? 2 : 4;
    static constexpr int CBY = CBITS / 2;  // bytes of codes per lane (four options)
    static __host__ __device__ int pitch(int LB) { return (DW * 4 + LB * CBY + 15) & ~15; }
};

template <int NE, bool PAIRS>
__global__ __launch_bounds__(256) void k_pack_rows(const float *__restrict__ dict, const unsigned char *__restrict__ codes, int code_pitch,
                                                   const unsigned *__restrict__ opt_pairs, long long rows, int K, int LB, int pitch,
                                                   unsigned char *__restrict__ table)
{
    using R = DictRow<NE>;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // (row, lane)
    if (i >= rows * LB) return;
    const long long r = i / LB;
    const int li = (int)(i - r * LB);
    unsigned packed = 0;
#pragma unroll
    for (int o = 0; o < 4; o++) {
        const int k = 4 * li + o;
        unsigned code = 0;
        if (k < K) {
            if (PAIRS) {
                const unsigned pr = opt_pairs[k];
                code = pair_entry(codes[(size_t)r * code_pitch + (pr & 0xFFFFu)] >> 3, codes[(size_t)r * code_pitch + (pr >> 16)] >> 3);
            } else {
                code = codes[(size_t)r * code_pitch + k] >> 3;
            }
        }
        packed |= code << (R::CBITS * o);
    }
    unsigned char *row = table + (size_t)r * pitch;
    if (R::CBY == 1) row[R::DW * 4 + li] = (unsigned char)packed;
    else ((unsigned short *)(row + R::DW * 4))[li] = (unsigned short)packed;
    if (li < R::DW) ((float *)row)[li] = dict[(size_t)r * DICT_CAP + li];
    if (R::DW > LB && li == 0)  // fewer lanes than values (K <= 16 with 8 values)
        for (int d = LB; d < R::DW; d++) ((float *)row)[d] = dict[(size_t)r * DICT_CAP + d];
}

// ------------------------------------------------------------------------------------
// Lane-per-four-options form (K <= 256).  What bounds a dictionary E-step is not arithmetic but the number of vector
// memory instructions (a wave64 gather occupies the CU's address unit for ~8 cycles whatever its width) and LDS
// reads, so every lane owns FOUR consecutive options: one dword gather brings their four codes, and a wavefront of
// 64 lanes serves NB = 64 / LB barcodes at once (LB = lanes per barcode, 4 LB >= K).
//   NE      entry slots per call in LDS: 4 or 8 (singlets), 16 (pairs of <= 4 values: 10 entries used)
//   PC      = 64 / NE calls whose logs one 64-lane pass computes; the packed log does two passes at once
//   CBB     = 2 PC / NB calls per barcode and batch
// Phase A, lane (slot u of the batch, entry e): u -> (barcode u / CBB, call u % CBB of its batch); the lane loads that
//   call's record, gathers its dictionary value(s), computes f64(np.log(p * keep + floor)) and parks it at
//   [buffer][barcode][call][entry] in LDS (+0 for calls past the barcode's row: adding +0 leaves a sum unchanged, the
//   sums start at +0 and never become -0).
// Phase B, lane (barcode ga, option quad li): per call of the batch one dword of codes (the call's code-row offset
//   comes from the group's lanes by ds_bpermute) and, per option, ds_read_b64 at (barcode base | code byte) + an
//   immediate, v_add_f64 - call after call in the barcode's order.
// The barcodes of a wavefront are neighbours in the length-sorted work list; the loop runs to the longest.
// ABL != 0: timing ablations (DEMUXALOT_AMD_DICT_ABLATE; results are meaningless): 1 no code gathers, 2 no LDS
// lookups, 3 no logs, 4 no record / dictionary loads inside the loop, 5 = 4 and no code gathers;
// DEMUXALOT_AMD_DICT_BLOCKS=n launches the first n workgroups only (the longest barcodes)
// ------------------------------------------------------------------------------------
struct DictRec {
    unsigned row;   // table row of the call
    float keep, flo;
    bool inside;    // the call exists (lies inside its barcode's row)
};

// LDS addresses are computed as integers (barcode base | code byte); these turn them into pointers
typedef const __attribute__((address_space(3))) double *LdsF64;
typedef __attribute__((address_space(3))) double *LdsF64W;
__device__ __forceinline__ LdsF64 lds_f64(unsigned byte_address) { return (LdsF64)(unsigned long long)byte_address; }
__device__ __forceinline__ LdsF64W lds_f64w(unsigned byte_address) { return (LdsF64W)(unsigned long long)byte_address; }

template <int NE, int LB, bool PAIRS, int ABL = 0>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(LB <= 16 ? 4 : 3))) void k_estep_dictq(EstepArgs a)
{
    constexpr int NB = 64 / LB;
    constexpr int PC = 64 / NE;
    constexpr int CBB = 2 * PC / NB;
    static_assert(CBB >= 1 && CBB * NB == 2 * PC, "a batch = two 64-lane passes of phase A");
    // LDS bytes per barcode and buffer: the batch's logs plus one entry row of skew, so that the NB barcodes' regions start
    // NE * 2 banks apart (a power-of-two stride put them on the same banks: 36 % of the LDS cycles were conflicts); a
    // multiple of NE * 8, so that base | code (< NE * 8) is base + code
    constexpr int STRIDE = CBB * NE * 8 + (NB > 1 ? NE * 8 : 0);
    constexpr int BUFSZ = NB * STRIDE;
    constexpr int NEV = PAIRS ? 10 : NE;                       // entries that exist
    constexpr int LE = 4 * LB < 64 ? 4 * LB : 64;              // epilogue: lanes per barcode,
    constexpr int AE = 4 * LB / LE;                            // register slots per lane,
    constexpr int GPC = 64 / LE;                               // barcodes per epilogue pass
    constexpr int STAGE = NB * 4 * LB * 8;                     // bytes of the accumulator hand-over
    __shared__ __attribute__((aligned(128))) unsigned char sh[2 * BUFSZ > STAGE ? 2 * BUFSZ : STAGE];
    const unsigned sh_off = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char *)sh;
    const int lane = threadIdx.x;
    const int K = a.K;

    // ---- phase B identity: barcode slot ga, option quad li ----
    const int li = lane % LB, ga = lane / LB, gbase = lane - li;
    const long long slotB = (long long)blockIdx.x * NB + ga;
    const bool liveB = slotB < a.B;
    const long long bB = a.order[liveB ? slotB : a.B - 1];
    const long long pbegB = a.pair_ptr[bB];
    const int nB = liveB ? 2 * (int)(a.pair_ptr[bB + 1] - pbegB) : 0;  // calls incl. neutral padding, multiple of 8
    int nmax = nB;
#pragma unroll
    for (int off = LB; off < 64; off <<= 1) nmax = max(nmax, __shfl_xor(nmax, off));
    nmax = __builtin_amdgcn_readfirstlane(nmax);
    double acc[4] = {0.0, 0.0, 0.0, 0.0};

    if (nmax > 0) {
        using R = DictRow<NE>;
        const unsigned pitch = (unsigned)a.dtab_pitch;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.dtab, 0, (int)a.dtab_bytes, 0x00020000);
        const unsigned quad_off = (unsigned)(R::DW * 4 + li * R::CBY);  // this lane's codes inside a table row
        // records and rows through buffer loads (32-bit offsets: the arrays are below 4 GiB, checked by the host)
        const __amdgpu_buffer_rsrc_t rsrc_rec = __builtin_amdgcn_make_buffer_rsrc((void *)a.pairs, 0, (int)a.pairs_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrc_row = __builtin_amdgcn_make_buffer_rsrc((void *)a.call_rows, 0, (int)(a.pairs_bytes / 4u), 0x00020000);
        const unsigned rowsB = (unsigned)pbegB * 8u;  // byte offset of this barcode's rows in call_rows
        const int limB = max(nB - 1, 0);
        const unsigned baseB = sh_off + (unsigned)(ga * STRIDE);

        // ---- phase A identity: two (barcode, call of the batch) per lane, one per pass ----
        const int tA = lane / NE;
        const unsigned eslot = (unsigned)(lane % NE), e = eslot < (unsigned)NEV ? eslot : (unsigned)NEV - 1u;
        const unsigned d1 = PAIRS ? pair_entry_lo(e) : e, d2 = PAIRS ? pair_entry_hi(e) : e;
        const int ux = tA, uy = PC + tA;
        const int ax = ux / CBB, qx = ux % CBB, ay = uy / CBB, qy = uy % CBB;
        const unsigned pbegX = (unsigned)__shfl((int)pbegB, ax * LB), pbegY = (unsigned)__shfl((int)pbegB, ay * LB);  // < 2^27 pairs
        const int nX = __shfl(nB, ax * LB), nY = __shfl(nB, ay * LB);
        const int limX = max(nX - 1, 0), limY = max(nY - 1, 0);
        const unsigned ldsX = sh_off + (unsigned)(ax * STRIDE + qx * NE * 8) + eslot * 8u, ldsY = sh_off + (unsigned)(ay * STRIDE + qy * NE * 8) + eslot * 8u;

        auto load_rec = [&](unsigned pbeg, int ci, int n, int lim) {
            DictRec r;
            r.inside = ci < n;
            ci = min(ci, lim);  // past the row: the last call again (an empty row: whatever record follows; +0 is parked)
            const unsigned woff = pbeg * 32u + (unsigned)((ci >> 1) * 32 + (ci & 1) * 4);
            r.row = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsrc_row, (int)(pbeg * 8u + (unsigned)ci * 4u), 0, 0);
            r.keep = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_rec, (int)woff + 8, 0, 0));
            r.flo = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_rec, (int)woff + 16, 0, 0));
            return r;
        };
        auto load_p = [&](const DictRec &r) {
            const unsigned off = __umul24(r.row, pitch);
            float p = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(off + d1 * 4u), 0, 0));
            if (PAIRS) p = (p + __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(off + d2 * 4u), 0, 0))) * 0.5f;  // demux.py:190
            return p;
        };
        auto produce = [&](int buf, const DictRec &rx, const DictRec &ry, float px, float py) {
            npm::f32x2 t;
            t.x = px * rx.keep;
            t.y = py * ry.keep;
            t.x = t.x + rx.flo;
            t.y = t.y + ry.flo;
            const npm::f32x2 l2 = ABL == 3 ? t : npm::log_f32_hot2(t);
            *lds_f64w(ldsX + (unsigned)(buf * BUFSZ)) = rx.inside ? (double)l2.x : 0.0;
            *lds_f64w(ldsY + (unsigned)(buf * BUFSZ)) = ry.inside ? (double)l2.y : 0.0;
        };
        // code-row offsets of this barcode's calls of batch k, one per lane of the group (lane li: call li % CBB)
        auto load_offsets = [&](int k) {
            const unsigned row = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsrc_row, (int)(rowsB + (unsigned)min(k * CBB + li % CBB, limB) * 4u), 0, 0);
            return __umul24(row, pitch);
        };
        auto load_codes = [&](unsigned (&c)[CBB], unsigned offs) {
#pragma unroll
            for (int q = 0; q < CBB; q++) {
                const unsigned off = (unsigned)__builtin_amdgcn_ds_bpermute(4 * (gbase + q), (int)offs);
                if (ABL == 1) c[q] = off & 0xFFu;
                else if (R::CBY == 1) c[q] = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rsrc, (int)(off + quad_off), 0, 0);
                else c[q] = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rsrc, (int)(off + quad_off), 0, 0);
            }
        };
        auto consume = [&](const unsigned (&c)[CBB], auto buf_tag) {
            constexpr int BUF = decltype(buf_tag)::value;
#pragma unroll
            for (int q = 0; q < CBB; q++) {  // call order
                double v[4];
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    const unsigned addr = baseB | (((c[q] >> (R::CBITS * o)) & ((1u << R::CBITS) - 1u)) << 3);
                    v[o] = ABL == 2 ? (double)(int)addr : *lds_f64(addr + (unsigned)(BUF * BUFSZ + q * NE * 8));
                }
#pragma unroll
                for (int o = 0; o < 4; o++) acc[o] += v[o];
                // one call's lookups in flight: left alone the compiler issues all 4 CBB reads (2 VGPRs each) before
                // the first addition and sinks the additions behind the loop's branch; registers are what this kernel
                // runs out of (resident wavefronts hide its memory latency)
                {
#pragma unroll
                    for (int o = 0; o < 4; o++) asm volatile("" : "+v"(acc[o]));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };

        // Software pipeline, two batches deep: what a wavefront waits for is memory latency (a gather that misses L2
        // takes a few thousand cycles under load), and with 4-5 resident wavefronts per SIMD every request must be
        // two batches of everybody's work old before its data is needed.
        //   entering step k:  cc0 / cc1 = codes of batches k / k + 1, offs2 = code-row offsets of k + 2,
        //                     (rA, pA) = records and dictionary values of k + 1, rB = records of k + 2,
        //                     LDS buffer k & 1 = logs of batch k
        const int nbatch = (nmax + CBB - 1) / CBB;
        auto load_recs = [&](int k, DictRec &rx, DictRec &ry) {
            if (ABL >= 4) k = 0;
            rx = load_rec(pbegX, k * CBB + qx, nX, limX);
            ry = load_rec(pbegY, k * CBB + qy, nY, limY);
        };
        // Batch-indexed state lives in register sets selected by (batch mod 3) / (batch mod 2) at compile time - the
        // loop is unrolled six-fold - because MOVING a register that an outstanding load will write means waiting
        // for that load (a rotating-register version spent every step's end in s_waitcnt vmcnt(0)).
        //   codes[j], recs[j]: batch = j mod 3        pv[j], offs[j]: batch = j mod 2
        unsigned codes[3][CBB];
        DictRec recx[3], recy[3];
        float pvx[2], pvy[2];
        unsigned offs[2];
        load_recs(0, recx[0], recy[0]);
        produce(0, recx[0], recy[0], load_p(recx[0]), load_p(recy[0]));
        load_codes(codes[0], load_offsets(0));
        load_codes(codes[1], load_offsets(1));
        offs[0] = load_offsets(2);
        load_recs(1, recx[1], recy[1]);
        pvx[1] = load_p(recx[1]);
        pvy[1] = load_p(recy[1]);
        load_recs(2, recx[2], recy[2]);
        // step k (S = k mod 6): requests records of k + 3, offsets of k + 3, dictionary values of k + 2, codes of k + 2;
        // consumes the codes and logs of k; parks the logs of k + 1
        auto step = [&](auto s_tag, int k) {
            constexpr int S = decltype(s_tag)::value;
            if (ABL < 4) {
                load_recs(k + 3, recx[S % 3], recy[S % 3]);
                offs[(S + 1) % 2] = load_offsets(k + 3);
                pvx[S % 2] = load_p(recx[(S + 2) % 3]);
                pvy[S % 2] = load_p(recy[(S + 2) % 3]);
            }
            if (ABL != 5) load_codes(codes[(S + 2) % 3], offs[S % 2]);
            __builtin_amdgcn_wave_barrier();  // the logs of this batch were parked by other lanes
            consume(codes[S % 3], std::integral_constant<int, S % 2>{});
            if (k + 1 < nbatch) produce((S + 1) % 2, recx[(S + 1) % 3], recy[(S + 1) % 3], pvx[(S + 1) % 2], pvy[(S + 1) % 2]);  // wave-uniform
        };
        for (int k = 0; k < nbatch; k += 6) {
            step(std::integral_constant<int, 0>{}, k);
            if (k + 1 < nbatch) step(std::integral_constant<int, 1>{}, k + 1);
            if (k + 2 < nbatch) step(std::integral_constant<int, 2>{}, k + 2);
            if (k + 3 < nbatch) step(std::integral_constant<int, 3>{}, k + 3);
            if (k + 4 < nbatch) step(std::integral_constant<int, 4>{}, k + 4);
            if (k + 5 < nbatch) step(std::integral_constant<int, 5>{}, k + 5);
        }
    }

    // ---- hand the sums over to the epilogue's layout (option k of a barcode in lane k & 63, slot k >> 6) ----
    __builtin_amdgcn_wave_barrier();
    {
        const LdsF64W stage = lds_f64w(sh_off + (unsigned)((ga * 4 * LB + 4 * li) * 8));
#pragma unroll
        for (int o = 0; o < 4; o++) stage[o] = acc[o];
    }
    __builtin_amdgcn_wave_barrier();
    const int le = lane % LE, ge = lane / LE;
    int kk[AE];
    bool valid[AE];
#pragma unroll
    for (int s = 0; s < AE; s++) {
        const int k = le + 64 * s;
        valid[s] = k < K;
        kk[s] = valid[s] ? k : K - 1;
    }
#pragma unroll
    for (int r = 0; r < NB / GPC; r++) {
        const int slot = r * GPC + ge;  // barcode slot of this lane in this pass
        const long long b = __shfl(bB, slot * LB);
        const int n_calls = __shfl(nB, slot * LB);
        const bool live = __shfl((int)liveB, slot * LB) != 0;
        if (GPC == 1 && !live) continue;  // wave-uniform (the 64-lane epilogue does not mask its bitmap stores)
        double sums[AE];
#pragma unroll
        for (int s = 0; s < AE; s++) sums[s] = *lds_f64(sh_off + (unsigned)((slot * 4 * LB + kk[s]) * 8));
        estep_epilogue<LE, AE>(a, b, live, sums, kk, valid, lane, le, lane - le, n_calls);
    }
}

template <int A>
static void launch_build_dict(hipStream_t st, const float *prob, long long rows, int G, float *dict, unsigned char *codes, unsigned *stat)
{
    constexpr int ROWS = A <= 2 ? 8 : A <= 4 ? 4 : 2;
    hipLaunchKernelGGL((k_build_dict<A, ROWS>), dim3((unsigned)((rows + 4 * ROWS - 1) / (4 * ROWS))), dim3(256), 0, st, prob, rows, G,
                       dict_code_pitch(G), dict, codes, stat);
}

// codes: [rows, dict_code_pitch(G)]
hipError_t launch_build_dict(hipStream_t st, const float *prob, long long rows, int G, float *dict, unsigned char *codes, unsigned *stat)
{
    hipError_t e = hipMemsetAsync(stat, 0, sizeof(unsigned), st);
    if (e != hipSuccess || rows == 0) return e;
    if (G <= 64 - 3) launch_build_dict<1>(st, prob, rows, G, dict, codes, stat);  // the pitch's tail is written by lanes >= G
    else if (G <= 128 - 3) launch_build_dict<2>(st, prob, rows, G, dict, codes, stat);
    else if (G <= 256 - 3) launch_build_dict<4>(st, prob, rows, G, dict, codes, stat);
    else if (G <= 512 - 3) launch_build_dict<8>(st, prob, rows, G, dict, codes, stat);
    else if (G <= 1024) launch_build_dict<16>(st, prob, rows, G, dict, codes, stat);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// Wide doublet tables (K > DICT_LANE_K, <= 4 distinct values per row): one 256-thread workgroup per barcode and tile of
// A * 256 options, like k_estep_block (kernels.hip) - but where that kernel evaluates numpy's log per (call, option),
// this one evaluates the 16 logs of a call's value pairs (c1, c2) once, by 16 threads, and every option looks its log up:
//   sh_lp[c][c1 * 4 + c2]   float64 log(((d[c1] + d[c2]) * 0.5) * keep_c + floor_c) of the chunk's call c (demux.py:190, 261)
//   sh_code[g][c]           byte 8 * (index of genotype g's value in the dictionary of call c's row), 8 calls per 8 bytes
// An option (g1, g2) adds, call after call, sh_lp[c][code[g1][c] * 4 + code[g2][c]]: per eight calls two ds_read_b64 of
// codes, two v_lshl_add (the eight byte offsets (c1 * 4 + c2) * 8 at once), and per call one byte extraction, one
// ds_read_b64, one v_add_f64 - against ~68 VALU issue cycles of log in the direct form.  Same float32 operations on
// the same operands, same addition order: bit-identical logits.
// ------------------------------------------------------------------------------------
constexpr int DB_C = 64;                   // calls per chunk
constexpr int DB_CODE_STRIDE = DB_C + 8;   // bytes between the code rows of two genotypes: 18 words, so that the 8-byte reads
                                           // of 32 lanes with consecutive second genotypes hit 32 different bank pairs

template <int A>
__global__ __launch_bounds__(256) void k_estep_dict_block(EstepArgs a, int k_base)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long b = a.order[blockIdx.x];
    const int K = a.K, G = a.G;
    // LDS carve: logs [C][16] f64 | rows [C] | keep [C] | floor [C] | codes [G][DB_CODE_STRIDE] bytes
    double *sh_lp = (double *)smem;
    unsigned *sh_row = (unsigned *)(sh_lp + DB_C * 16);
    float *sh_keep = (float *)(sh_row + DB_C);
    float *sh_floor = sh_keep + DB_C;
    unsigned char *sh_code = (unsigned char *)(sh_floor + DB_C);
    // The kernel's only LDS is this dynamic array, so the logs start at LDS address 0 and a lookup's address is (code byte) +
    // an immediate; checked rather than assumed.
    if ((unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char *)smem != 0u) __builtin_trap();
    constexpr unsigned lp_off = 0u;

    unsigned a1[A], a2[A];  // LDS byte address of the code rows of this thread's options' genotypes
#pragma unroll
    for (int s = 0; s < A; s++) {
        const int k = k_base + s * 256 + tid;
        const unsigned pr = a.opt_pairs[k < K ? k : K - 1];
        a1[s] = (pr & 0xFFFFu) * (unsigned)DB_CODE_STRIDE;
        a2[s] = (pr >> 16) * (unsigned)DB_CODE_STRIDE;
    }
    double acc[A];
#pragma unroll
    for (int s = 0; s < A; s++) acc[s] = 0.0;

    const unsigned *__restrict__ words = (const unsigned *)(a.pairs + a.pair_ptr[b]);
    const unsigned *__restrict__ rows = a.call_rows + 2 * a.pair_ptr[b];
    const int n_calls = 2 * (int)(a.pair_ptr[b + 1] - a.pair_ptr[b]);  // incl. neutral padding, multiple of 8
    const int code_pitch = dict_code_pitch(G);
    const int e = tid & 15, cq = tid >> 4;  // phase A: pair entry and call (cq, cq + 16, cq + 32, cq + 48)
    const unsigned c1 = (unsigned)e >> 2, c2 = (unsigned)e & 3u;
    for (int pos = 0; pos < n_calls; pos += DB_C) {
        const int n = (n_calls - pos) < DB_C ? (n_calls - pos) : DB_C;  // multiple of 8
        __syncthreads();
        if (tid < n) {
            const int ci = pos + tid;
            const int w = (ci >> 1) * 8 + (ci & 1);
            sh_row[tid] = rows[ci];
            sh_keep[tid] = __uint_as_float(words[w + 2]);
            sh_floor[tid] = __uint_as_float(words[w + 4]);
        }
        __syncthreads();
        // codes: wave w stages calls w, w + 4, ...: one coalesced read of the row's G bytes, transposed byte stores
        for (int c = wave; c < n; c += 4) {
            const unsigned char *row = a.codes + (size_t)sh_row[c] * code_pitch;
            for (int g = lane; g < G; g += 64) sh_code[g * DB_CODE_STRIDE + c] = row[g];
        }
        // logs: 16 pair entries of every call, two calls per packed log
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int cx = cq + 32 * h, cy = cx + 16;
            if (cx < n) {  // n is a multiple of 8: cy < n need not hold
                const int cyc = cy < n ? cy : cx;
                const float *dx = a.dict + (size_t)sh_row[cx] * DICT_CAP, *dy = a.dict + (size_t)sh_row[cyc] * DICT_CAP;
                npm::f32x2 t;
                t.x = ((dx[c1] + dx[c2]) * 0.5f) * sh_keep[cx];
                t.y = ((dy[c1] + dy[c2]) * 0.5f) * sh_keep[cyc];
                t.x = t.x + sh_floor[cx];
                t.y = t.y + sh_floor[cyc];
                const npm::f32x2 l2 = npm::log_f32_hot2(t);
                sh_lp[cx * 16 + e] = (double)l2.x;
                if (cy < n) sh_lp[cy * 16 + e] = (double)l2.y;
            }
        }
        __syncthreads();
        // The chunk's eight groups of eight calls are unrolled so that every LDS address is (extracted byte) + an
        // immediate: one v_bfe_u32 per lookup.  (With the group offset in a register the compiler spent a v_add_u32_sdwa
        // per lookup - a slow form on this chip: the kernel sat at 100 % VALU busy with 3.5 instructions per term.)
#pragma unroll
        for (int c0 = 0; c0 < DB_C; c0 += 8) {
            if (c0 >= n) break;  // workgroup-uniform
            // (slots past the last option - the tail of the last tile - walk option K - 1 and are not stored: a test per
            // slot would cut the straight-line code into A serial chains of two LDS latencies.)  The codes of slot s + 1
            // are requested before the lookups of slot s.
            uint2 w1 = *(const uint2 *)(sh_code + a1[0] + c0), w2 = *(const uint2 *)(sh_code + a2[0] + c0);
#pragma unroll
            for (int s = 0; s < A; s++) {
                const unsigned lo = (w1.x << 2) + w2.x, hi = (w1.y << 2) + w2.y;  // bytes: (c1 * 4 + c2) * 8 <= 120, no carries
                if (s + 1 < A) {
                    w1 = *(const uint2 *)(sh_code + a1[s + 1] + c0);
                    w2 = *(const uint2 *)(sh_code + a2[s + 1] + c0);
                }
                double v[8];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    unsigned idx;  // (asm: the compiler's own selection is v_add_u32_sdwa / v_mov_b32_sdwa, slow forms here)
                    asm("v_bfe_u32 %0, %1, %2, 8" : "=v"(idx) : "v"(q < 4 ? lo : hi), "n"(8 * (q & 3)));
                    v[q] = *lds_f64(lp_off + idx + (unsigned)((c0 + q) * 128));
                }
#pragma unroll
                for (int q = 0; q < 8; q++) acc[s] += v[q];  // call order
            }
        }
    }
    __syncthreads();
    // logits of this tile of options; the softmax over complete rows is k_softmax_rows' (kernels.hip)
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

template <int A>
static hipError_t launch_dict_block(hipStream_t st, const EstepArgs &a, int k_base)
{
    const size_t bytes = (size_t)DB_C * 16 * 8 + (size_t)DB_C * 12 + (size_t)a.G * DB_CODE_STRIDE;
    hipError_t e = hipFuncSetAttribute((const void *)k_estep_dict_block<A>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_estep_dict_block<A>), dim3((unsigned)a.B), dim3(256), bytes, st, a, k_base);
    return hipGetLastError();
}

// doublet tables of more than DICT_LANE_K options: option tiles, then the softmax over complete rows
hipError_t launch_estep_dict_block(hipStream_t st, const EstepArgs &a)
{
    if (a.B == 0) return hipSuccess;
    const int K = a.K, need = (K + 255) / 256;
    int tile = need <= 2 ? 2 : need <= 4 ? 4 : need <= 8 ? 8 : need <= 12 ? 12 : need <= 17 ? 17 : need <= 24 ? 12 : 17;
#ifdef DMX_EXPERIMENTS  // tuning aid of experiment builds (make EXPERIMENTS=1); never in the library build() ships
    static const int forced = std::getenv("DEMUXALOT_AMD_DICT_TILE") ? std::atoi(std::getenv("DEMUXALOT_AMD_DICT_TILE")) : 0;
    if (forced == 2 || forced == 4 || forced == 8 || forced == 12 || forced == 17) tile = forced;
#endif
    for (int k_base = 0; k_base < K; k_base += tile * 256) {
        const hipError_t e = tile == 2 ? launch_dict_block<2>(st, a, k_base) : tile == 4 ? launch_dict_block<4>(st, a, k_base)
                           : tile == 8 ? launch_dict_block<8>(st, a, k_base) : tile == 12 ? launch_dict_block<12>(st, a, k_base)
                                                                                            : launch_dict_block<17>(st, a, k_base);
        if (e != hipSuccess) return e;
    }
    return launch_softmax_rows(st, a);
}

static int dict_lanes(int K, bool pairs)  // lanes per barcode of the E-step kernel
{
    return K <= 16 && !pairs ? 4 : K <= 32 ? 8 : K <= 64 ? 16 : K <= 128 ? 32 : 64;
}

// distinct: what launch_build_dict found (<= DICT_CAP; <= DICT_PAIR_CAP for doublet runs); returns the row pitch
int dict_table_pitch(int distinct, int K, bool pairs)
{
    const int LB = dict_lanes(K, pairs);
    return pairs ? DictRow<16>::pitch(LB) : distinct <= 4 ? DictRow<4>::pitch(LB) : DictRow<8>::pitch(LB);
}

hipError_t launch_pack_rows(hipStream_t st, const float *dict, const unsigned char *codes, const unsigned *opt_pairs, long long rows, int G,
                            int K, bool pairs, int distinct, unsigned char *table)
{
    const int LB = dict_lanes(K, pairs), pitch = dict_table_pitch(distinct, K, pairs);
    const long long n = rows * LB;
    if (n == 0) return hipSuccess;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipError_t e = hipMemsetAsync(table, 0, (size_t)rows * pitch, st);
    if (e != hipSuccess) return e;
    if (pairs) hipLaunchKernelGGL((k_pack_rows<16, true>), grid, block, 0, st, dict, codes, dict_code_pitch(G), opt_pairs, rows, K, LB, pitch, table);
    else if (distinct <= 4) hipLaunchKernelGGL((k_pack_rows<4, false>), grid, block, 0, st, dict, codes, dict_code_pitch(G), opt_pairs, rows, K, LB, pitch, table);
    else hipLaunchKernelGGL((k_pack_rows<8, false>), grid, block, 0, st, dict, codes, dict_code_pitch(G), opt_pairs, rows, K, LB, pitch, table);
    return hipGetLastError();
}

template <int NE, int LB, bool PAIRS>
static void launch_dictq(hipStream_t st, const EstepArgs &a)
{
    constexpr int NB = 64 / LB;
    long long max_blocks = 1LL << 40;
#ifdef DMX_EXPERIMENTS
    // Timing ablations and partial launches of experiment builds (make EXPERIMENTS=1; results are meaningless): an
    // environment variable must never be able to make the shipped library return wrong posteriors, so build() compiles
    // none of this (profiles/r3_pmc_dictq_em_200k_100k_64.txt was taken with such a build).
    static const int ablate = std::getenv("DEMUXALOT_AMD_DICT_ABLATE") ? std::atoi(std::getenv("DEMUXALOT_AMD_DICT_ABLATE")) : 0;
    static const long long forced_blocks = std::getenv("DEMUXALOT_AMD_DICT_BLOCKS") ? std::atoll(std::getenv("DEMUXALOT_AMD_DICT_BLOCKS")) : (1LL << 40);
    max_blocks = forced_blocks;
    if constexpr (NE == 4 && LB == 16 && !PAIRS) {
        const dim3 grid_a((unsigned)std::min<long long>(max_blocks, (a.B + NB - 1) / NB)), block_a(64);
        if (ablate == 1) {
            hipLaunchKernelGGL((k_estep_dictq<4, 16, false, 1>), grid_a, block_a, 0, st, a);
            return;
        }
        if (ablate == 2) {
            hipLaunchKernelGGL((k_estep_dictq<4, 16, false, 2>), grid_a, block_a, 0, st, a);
            return;
        }
        if (ablate == 3) {
            hipLaunchKernelGGL((k_estep_dictq<4, 16, false, 3>), grid_a, block_a, 0, st, a);
            return;
        }
        if (ablate == 4) {
            hipLaunchKernelGGL((k_estep_dictq<4, 16, false, 4>), grid_a, block_a, 0, st, a);
            return;
        }
        if (ablate == 5) {
            hipLaunchKernelGGL((k_estep_dictq<4, 16, false, 5>), grid_a, block_a, 0, st, a);
            return;
        }
    }
#endif
    const dim3 grid((unsigned)std::min<long long>(max_blocks, (a.B + NB - 1) / NB)), block(64);
    hipLaunchKernelGGL((k_estep_dictq<NE, LB, PAIRS, 0>), grid, block, 0, st, a);
}

template <int NE, bool PAIRS>
static hipError_t launch_dict_ne(hipStream_t st, const EstepArgs &a)
{
    const int K = a.K;
    if (K > DICT_LANE_K) return hipErrorInvalidValue;
    const int LB = dict_lanes(K, PAIRS);
    if (LB == 4) {
        if constexpr (!PAIRS) launch_dictq<NE, 4, false>(st, a);
    } else if (LB == 8) launch_dictq<NE, 8, PAIRS>(st, a);
    else if (LB == 16) launch_dictq<NE, 16, PAIRS>(st, a);
    else if (LB == 32) launch_dictq<NE, 32, PAIRS>(st, a);
    else launch_dictq<NE, 64, PAIRS>(st, a);
    return hipGetLastError();
}

// a.dict_n distinct values per row at most (dict_form_fits says whether this form exists for the problem)
hipError_t launch_estep_dict(hipStream_t st, const EstepArgs &a, bool pairs)
{
    if (a.B == 0) return hipSuccess;
    if (pairs) return launch_dict_ne<16, true>(st, a);
    return a.dict_n <= 4 ? launch_dict_ne<4, false>(st, a) : launch_dict_ne<8, false>(st, a);
}

}  // namespace dmx
