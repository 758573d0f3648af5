// estep_packed.hip -- exact E-step for NARROW DOUBLET TABLES: several option slots per lane.
//
// Reference: demuxalot/demux.py:246-265 (compute_barcode_logits_using_barcode_calls) over the options of
// demux.py:175-191 (singlets first, then the pairs g1 < g2 with (p1 + p2) * 0.5).
//
// The direct form (kernels.hip: k_estep_direct) gives every option a lane of a power-of-two lane group: K = 36 options
// (8 genotypes with doublets: BASELINE.json configs[1]) occupy 36 of 64 lanes, and the kernel is bound by VALU issue
// (SQ_ACTIVE_INST_VALU 88 % of the launch, profiles/r3_pmc_direct_small.txt), so 44 % of what it issues is thrown away.
// Here a lane group has L = 8 / 16 / 32 lanes and every lane A = 3 or 5 option slots (option k = lane k % L, slot
// k / L): K = 36 -> 8 lanes x 5 slots, 8 barcodes per wavefront, 36 of 40 slots busy.
//
//   * Few wavefronts: 20 000 barcodes at 8 per wavefront are 2 500 wavefronts, two or three per SIMD, and every one
//     of them is a serial walk over its rows' calls - no other wavefront hides its loads.  So the walk is software-
//     pipelined by BLOCKS of 8 records (16 calls): the records of block k + 2 and the genotype rows of block k + 1 are
//     requested before block k is evaluated.  [A first version with records two and rows one 2-call step ahead ran at
//     0.67 ms on 20k x 20k x 8 against the direct form's 0.27: every step waited for its row.]
//   * Records: lane i of a group holds record 8 k + i of the block (three 8-byte loads per lane and block), row
//     offsets, keep and floor reach the group by ds_bpermute (lane addresses fixed at kernel start).
//   * Genotype rows through the lanes: G <= L in every shape this kernel takes (K = G (G + 1) / 2 <= L A), so lane i of
//     the group loads p[i] of the call's row once, and the two operands of every option come by ds_bpermute: one v_add
//     per CALL for the row address instead of two per TERM.
//   * Split launch: a barcode's calls are added in order, so its walk is serial and A slots per lane make it A times
//     longer; the first EstepArgs::n_long barcodes of the length-sorted list (the repack counts the rows with more calls
//     than a third of what a SIMD gets on average) therefore take a 64-lane wavefront each, in the first blocks of the
//     same launch (walk_on_64_lanes).  60k x 20k x 8 with doublets: 0.86 ms direct, 0.88 ms all packed, 0.67 ms split.
//   * The float64 sums take the same float32 terms in the same order as the direct form: bit-identical.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "np_math.h"
#include "estep_epilogue.h"

namespace dmx {

#ifndef PACKED_BLOCK_RECORDS
#define PACKED_BLOCK_RECORDS 8  // records (pairs of calls) per block of the software pipeline: 4 or 8 (rows are padded to 4); 4 saves 14 VGPRs and is 1-5 % slower
#endif

// One barcode on the whole wavefront, option k on lane k % 64, slot k / 64: the walk of k_estep_direct<64, A64, true>
// (records through the scalar cache, rows gathered with the row offset as the buffer load's scalar offset).  For the
// longest barcodes of a launch: a barcode's calls are added in order, so its walk is serial, and with A option slots
// per lane it is A times longer than here.
template <int A64>
static __device__ __forceinline__ void walk_on_64_lanes(const EstepArgs &a, long long slot)
{
    const int lane = threadIdx.x & 63;
    const int K = a.K;
    unsigned o1[A64], o2[A64];
    int kk[A64];
    bool valid[A64];
#pragma unroll
    for (int s = 0; s < A64; s++) {
        const int k = lane + 64 * s;
        valid[s] = k < K;
        kk[s] = valid[s] ? k : K - 1;
        const unsigned pr = a.opt_pairs[kk[s]];
        o1[s] = (pr & 0xFFFFu) * 4u;
        o2[s] = (pr >> 16) * 4u;
    }
    double acc[A64];
#pragma unroll
    for (int s = 0; s < A64; s++) acc[s] = 0.0;
    // The order list, the row offsets and the call records are read through the CONSTANT address space: with the packed
    // path in the same kernel the compiler no longer proves these (wave-uniform, never written) loads unclobbered,
    // fetched the records with vector loads and wrapped every row gather in a v_readfirstlane loop (0.60 ms for 20k x
    // 20k x 8 against the 0.28 ms of the same walk in k_estep_direct).
    typedef const __attribute__((address_space(4))) int *const_int_ptr;
    typedef const __attribute__((address_space(4))) long long *const_i64_ptr;
    typedef const __attribute__((address_space(4))) CallPair *const_rec_ptr;
    const long long b = ((const_int_ptr)(uintptr_t)a.order)[slot];
    const long long pbeg = ((const_i64_ptr)(uintptr_t)a.pair_ptr)[b];
    const int npairs = (int)(((const_i64_ptr)(uintptr_t)a.pair_ptr)[b + 1] - pbeg);
    const const_rec_ptr recs = (const_rec_ptr)(uintptr_t)(a.pairs + pbeg);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.prob, 0, (int)a.prob_bytes, 0x00020000);
    constexpr int H = A64 == 1 ? 4 : 2;  // rows are padded to 4 pairs
    const int n_slots = (K + 63) >> 6;
    auto load = [&](RecBatch<A64, H, true> &x, int j) {
        j = __builtin_amdgcn_readfirstlane(j);
#pragma unroll
        for (int q = 0; q < H; q++) {
            const unsigned ro0 = recs[j + q].row_off[0], ro1 = recs[j + q].row_off[1];
            x.keep[q] = npm::f32x2{recs[j + q].keep[0], recs[j + q].keep[1]};
            x.flo[q] = npm::f32x2{recs[j + q].floor[0], recs[j + q].floor[1]};
#pragma unroll
            for (int s = 0; s < A64; s++) {
                if (A64 > 1 && s >= n_slots) continue;
                x.p1[q][s].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)o1[s], (int)ro0, 0));
                x.p1[q][s].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)o1[s], (int)ro1, 0));
                x.p2[q][s].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)o2[s], (int)ro0, 0));
                x.p2[q][s].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)o2[s], (int)ro1, 0));
            }
        }
    };
    // Two batches in flight (this kernel holds 4 wavefronts per SIMD, k_estep_direct 8: the gathers of batch i + 1 are
    // issued before batch i is evaluated); a read past the row's last batch is redirected to it and not evaluated.
    if (npairs > 0) {
        RecBatch<A64, H, true> x, y;
        load(x, 0);
        for (int j0 = 0; j0 < npairs; j0 += 2 * H) {
            const int j1 = j0 + H;
            load(y, min(j1, npairs - H));
            __builtin_amdgcn_sched_barrier(0);
            estep_terms<A64, true, H>(x.p1, x.p2, x.keep, x.flo, acc, n_slots);
            if (j1 >= npairs) break;
            load(x, min(j1 + H, npairs - H));
            __builtin_amdgcn_sched_barrier(0);
            estep_terms<A64, true, H>(y.p1, y.p2, y.keep, y.flo, acc, n_slots);
        }
    }
    estep_epilogue<64, A64>(a, b, true, acc, kk, valid, lane, lane, 0, 2 * npairs);
}

template <int L, int A>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_estep_packed(EstepArgs a)
{
    // the first a.n_long barcodes of the length-sorted list: one per wavefront (blocks 0 .. ceil(n_long / 4) - 1)
    const long long long_blocks = (a.n_long + 3) >> 2;
    if ((long long)blockIdx.x < long_blocks) {
        const long long slot64 = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        if (slot64 < a.n_long) walk_on_64_lanes<(L * A + 63) / 64>(a, slot64);
        return;
    }

    static_assert(L == 8 || L == 16 || L == 32, "lane groups of 8, 16 or 32");
    static_assert(A >= 2 && A <= 5, "option slots per lane");
    constexpr int CPW = 64 / L;
    const int lane = threadIdx.x & 63;
    const int li = lane & (L - 1);
    const int gbase = lane - li;
    const int K = a.K;

    int kk[A], sel1[A], sel2[A];  // ds_bpermute addresses (lane * 4) of the option's two genotypes inside the group
    bool valid[A];
#pragma unroll
    for (int s = 0; s < A; s++) {
        const int k = li + L * s;
        valid[s] = k < K;
        kk[s] = valid[s] ? k : K - 1;
        const unsigned pr = a.opt_pairs[kk[s]];
        sel1[s] = (gbase + (int)(pr & 0xFFFFu)) * 4;
        sel2[s] = (gbase + (int)(pr >> 16)) * 4;
    }
    double acc[A];
#pragma unroll
    for (int s = 0; s < A; s++) acc[s] = 0.0;

    const long long slot = a.n_long + (((long long)blockIdx.x - long_blocks) * 4 + (threadIdx.x >> 6)) * CPW + lane / L;
    const bool live = slot < a.B;
    const long long b = a.order[live ? slot : a.B - 1];
    const long long pbeg = a.pair_ptr[b];
    const int npairs = live ? (int)(a.pair_ptr[b + 1] - pbeg) : 0;  // multiple of 4
    const int nmax = group_max_over_wave<L>(npairs);

    // Records: lane i of a group (i mod BR) holds record BR * block + i of the group's row; a block = BR records.
    // Past the end of the row: the neutral record behind the last row (keep 0, floor 1: log(p * 0 + 1) = +0).
    const unsigned neutral = (unsigned)a.pair_ptr[a.B] * 32u;
    const unsigned row_begin = (unsigned)pbeg * 32u;
    const char *__restrict__ recs = (const char *)a.pairs;
    const char *__restrict__ prob = (const char *)a.prob;
    const unsigned my_col = (unsigned)(li < a.G ? li : a.G - 1) * 4u;  // the genotype this lane fetches of every row
    constexpr int BR = PACKED_BLOCK_RECORDS;
    const int l8 = li & (BR - 1);
    int bcast[BR];  // ds_bpermute address of the group's lane that holds record r of a block
#pragma unroll
    for (int r = 0; r < BR; r++) bcast[r] = (gbase + r) * 4;

    struct Rec {
        uint2 ro, keep, fl;
    };
    auto load_block = [&](int first) {  // records first .. first + BR - 1 of every group, one per lane
        const int idx = first + l8;
        const unsigned off = idx < npairs ? row_begin + (unsigned)idx * 32u : neutral;
        Rec r;
        r.ro = *(const uint2 *)(recs + off);
        r.keep = *(const uint2 *)(recs + off + 8);
        r.fl = *(const uint2 *)(recs + off + 16);
        return r;
    };
    auto load_rows = [&](const Rec &blk, float (&rows)[2 * BR]) {  // p[my genotype] of the 2 BR calls of a block
#pragma unroll
        for (int r = 0; r < BR; r++) {
            const unsigned ro0 = (unsigned)__builtin_amdgcn_ds_bpermute(bcast[r], (int)blk.ro.x);
            const unsigned ro1 = (unsigned)__builtin_amdgcn_ds_bpermute(bcast[r], (int)blk.ro.y);
            rows[2 * r] = *(const float *)(prob + (ro0 + my_col));
            rows[2 * r + 1] = *(const float *)(prob + (ro1 + my_col));
        }
    };
    auto terms = [&](const Rec &blk, const float (&rows)[2 * BR], int r) {
        npm::f32x2 p1[1][A], p2[1][A], keep[1], flo[1];
        keep[0].x = __int_as_float(__builtin_amdgcn_ds_bpermute(bcast[r], (int)blk.keep.x));
        keep[0].y = __int_as_float(__builtin_amdgcn_ds_bpermute(bcast[r], (int)blk.keep.y));
        flo[0].x = __int_as_float(__builtin_amdgcn_ds_bpermute(bcast[r], (int)blk.fl.x));
        flo[0].y = __int_as_float(__builtin_amdgcn_ds_bpermute(bcast[r], (int)blk.fl.y));
        const int r0 = __float_as_int(rows[2 * r]), r1 = __float_as_int(rows[2 * r + 1]);
#pragma unroll
        for (int s = 0; s < A; s++) {
            p1[0][s].x = __int_as_float(__builtin_amdgcn_ds_bpermute(sel1[s], r0));
            p1[0][s].y = __int_as_float(__builtin_amdgcn_ds_bpermute(sel1[s], r1));
            p2[0][s].x = __int_as_float(__builtin_amdgcn_ds_bpermute(sel2[s], r0));
            p2[0][s].y = __int_as_float(__builtin_amdgcn_ds_bpermute(sel2[s], r1));
        }
        estep_terms<A, true, 1>(p1, p2, keep, flo, acc, A);
    };

    if (nmax > 0) {
        // block k of the walk: the records of block k + 2 requested, the rows of block k + 1 requested (their offsets
        // broadcast from the lanes that hold its records), the 8 x 2 calls of block k evaluated.  What is requested at
        // the start of a block is first needed one block (8 records x A slots x ~140 cycles) later, which is what
        // a SIMD with two or three wavefronts of this kernel needs: it cannot count on other wavefronts to hide a load.
        Rec cur = load_block(0), nxt = load_block(BR);
        float rows[2 * BR], rows_nxt[2 * BR];
        load_rows(cur, rows);
        for (int j = 0; j < nmax; j += BR) {
            const Rec far = load_block(j + 2 * BR);
            load_rows(nxt, rows_nxt);
            __builtin_amdgcn_sched_barrier(0);  // the loads are issued here, not where their values are first used
#pragma unroll
            for (int r = 0; r < BR; r++) terms(cur, rows, r);
            __builtin_amdgcn_sched_barrier(0);
            cur = nxt;
            nxt = far;
#pragma unroll
            for (int i = 0; i < 2 * BR; i++) rows[i] = rows_nxt[i];
        }
    }
    estep_epilogue<L, A>(a, b, live, acc, kk, valid, lane, li, gbase, 2 * npairs);
}

// Lane-group shape for a doublet table of K options over G genotypes; false: the direct form is as good or better.
bool estep_packed_shape(int K, int G, int *lanes, int *slots)
{
    static const int shapes[][2] = {{8, 3}, {8, 5}, {16, 3}, {16, 5}, {32, 3}, {32, 5}};
    int direct = 4;  // slots the direct form spends on a row: next power of two up to 64, then multiples of 64 x {1, 2, 4, 8, 16}
    while (direct < K && direct < 64) direct <<= 1;
    while (direct < K) direct <<= 1;
    int best = direct, bl = 0, ba = 0;
    for (const auto &sh : shapes) {
        const int cap = sh[0] * sh[1];
        if (cap >= K && G <= sh[0] && cap < best) {
            best = cap;
            bl = sh[0];
            ba = sh[1];
        }
    }
    if (!bl) return false;
    *lanes = bl;
    *slots = ba;
    return true;
}

template <int L, int A>
static void launch_packed(hipStream_t st, const EstepArgs &a)
{
    const long long rest = a.B - a.n_long;
    const unsigned blocks = (unsigned)((a.n_long + 3) / 4 + (rest + 4 * (64 / L) - 1) / (4 * (64 / L)));
    hipLaunchKernelGGL((k_estep_packed<L, A>), dim3(blocks), dim3(256), 0, st, a);
}

hipError_t launch_estep_packed(hipStream_t st, const EstepArgs &a)
{
    if (a.B == 0) return hipSuccess;
    int L = 0, A = 0;
    if (a.fast || a.pairs_bytes == 0 || a.n_long < 0 || a.n_long > a.B || !estep_packed_shape(a.K, a.G, &L, &A)) return hipErrorInvalidValue;
    if (L == 8 && A == 3) launch_packed<8, 3>(st, a);
    else if (L == 8) launch_packed<8, 5>(st, a);
    else if (L == 16 && A == 3) launch_packed<16, 3>(st, a);
    else if (L == 16) launch_packed<16, 5>(st, a);
    else if (L == 32 && A == 3) launch_packed<32, 3>(st, a);
    else launch_packed<32, 5>(st, a);
    return hipGetLastError();
}

}  // namespace dmx
