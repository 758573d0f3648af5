// host_stubs.cpp -- link-time stand-ins for the GPU side of libdemux_hip.so, for the SANITIZER build of the host shim
// (make asan -> ../libdemux_host_asan.so: dmx_api.cpp + pack_host.cpp + this file, g++ -fsanitize=address,undefined).
// Never part of the shipped library and never a fallback: the kernels' launchers do NOTHING here, so every result is
// meaningless.  What the build is for: running the 2 000 lines of host logic that carry the C ABI's error contract -
// argument validation, call order, the block cache, np.sum plans, variant slices and the padded layouts
// of the multi-GPU exchange, host-staged collectives - under AddressSanitizer and UBSan on the CPU
// (tests/test_host_sanitizers.py; the contract replaces the reference's asserts at
// demux.py:78,81,98,135,160,317,359,374).  "Device memory" is plain malloc memory, so that every copy the shim makes
// into or out of a device buffer is bounds-checked by the sanitizer as well.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

#include "dmx_ctx.h"

// ------------------------------------------------------------------------------------
// HIP runtime
// ------------------------------------------------------------------------------------
extern "C" {

hipError_t hipGetDeviceCount(int *count)
{
    *count = 1;
    return hipSuccess;
}
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipDeviceGetAttribute(int *value, hipDeviceAttribute_t, int)
{
    *value = 100000;  // wall clock rate in kHz
    return hipSuccess;
}
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hip error (host stub)"; }

hipError_t hipMalloc(void **p, size_t bytes)
{
    // a cap far below a GPU's memory: requests beyond it fail like an exhausted device (exercises the retry path)
    if (bytes > (size_t(1) << 31)) {
        *p = nullptr;
        return hipErrorOutOfMemory;
    }
    *p = std::malloc(bytes ? bytes : 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void *p)
{
    std::free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned) { return hipMalloc(p, bytes); }
hipError_t hipHostFree(void *p) { return hipFree(p); }

hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind, hipStream_t)
{
    if (bytes) std::memmove(dst, src, bytes);
    return hipSuccess;
}
hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind) { return hipMemcpyAsync(dst, src, bytes, kind, nullptr); }
hipError_t hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t)
{
    for (size_t r = 0; r < height; r++) std::memmove((char *)dst + r * dpitch, (const char *)src + r * spitch, width);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *dst, int value, size_t bytes, hipStream_t)
{
    if (bytes) std::memset(dst, value, bytes);
    return hipSuccess;
}

static int g_stream_tag, g_event_tag;
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned)
{
    *s = (hipStream_t)&g_stream_tag;
    return hipSuccess;
}
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int)
{
    *s = (hipStream_t)&g_stream_tag;
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest)
{
    *least = 0;
    *greatest = -2;
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t *e)
{
    *e = (hipEvent_t)&g_event_tag;
    return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t)
{
    *ms = 0.0f;
    return hipSuccess;
}

}  // extern "C"

// ------------------------------------------------------------------------------------
// kernel launchers: nothing runs (see the header comment).  The two whose output the host reads say "nothing found".
// ------------------------------------------------------------------------------------
namespace dmx {

hipError_t launch_sum_dense(hipStream_t, unsigned long long *, unsigned *) { return hipSuccess; }
hipError_t launch_probs_from_betas(hipStream_t, const float *, const float *, const int *, const int *, const int *, long long, long long,
                                   long long, int, const int *, float, float, float *, unsigned short *) { return hipSuccess; }
hipError_t launch_probs_from_betas_f64(hipStream_t, const double *, const int *, const int *, const int *, long long, long long, int,
                                       const int *, float, float, float *) { return hipSuccess; }
hipError_t launch_check_unit_range(hipStream_t, const float *, long long, int *) { return hipSuccess; }
hipError_t launch_estep(hipStream_t, const EstepArgs &, bool) { return hipSuccess; }
hipError_t launch_softmax_rows(hipStream_t, const EstepArgs &) { return hipSuccess; }
hipError_t launch_build_dict(hipStream_t, const float *, long long, int, float *, unsigned char *, unsigned *stat)
{
    *stat = DICT_CAP + 1;  // "some row has too many distinct values": the direct form
    return hipSuccess;
}
int dict_table_pitch(int, int, bool) { return 32; }
hipError_t launch_pack_rows(hipStream_t, const float *, const unsigned char *, const unsigned *, long long, int, int, bool, int,
                            unsigned char *) { return hipSuccess; }
hipError_t launch_estep_dict(hipStream_t, const EstepArgs &, bool) { return hipSuccess; }
hipError_t launch_estep_dict_block(hipStream_t, const EstepArgs &) { return hipSuccess; }
hipError_t launch_mstep(hipStream_t, const MstepArgs &) { return hipSuccess; }
hipError_t launch_mstep_incremental(hipStream_t, const MstepArgs &, const MTileArgs &, const MIncrArgs &) { return hipSuccess; }
hipError_t launch_mstep_items_incremental(hipStream_t, const MstepArgs &, const MIncrArgs &) { return hipSuccess; }
hipError_t launch_mstep_incremental_sharded(hipStream_t, const MstepArgs &, const MTileArgs &, const MIncrArgs &) { return hipSuccess; }
hipError_t launch_mstep_tiles(hipStream_t, const MstepArgs &, const MTileArgs &) { return hipSuccess; }
hipError_t launch_guard_begin(hipStream_t, unsigned *, long long, int, int, int, int) { return hipSuccess; }
hipError_t launch_prob_to_half(hipStream_t, const float *, long long, int, unsigned short *, const unsigned *) { return hipSuccess; }
hipError_t launch_build_coarse_stream(hipStream_t, const CallPair *, const long long *, long long, unsigned, int, long long *, unsigned *, const int *, int, double *) { return hipSuccess; }
hipError_t launch_guard_stamp(hipStream_t, unsigned *, int) { return hipSuccess; }
hipError_t launch_guard_compact(hipStream_t, unsigned *, const int *, unsigned, int *, const int *, long long) { return hipSuccess; }
hipError_t launch_mcombine(hipStream_t, const MstepArgs &, const long long *, long long, long long, const int *, float *, double *,
                           unsigned long long *, unsigned *, const int *, bool) { return hipSuccess; }
hipError_t launch_store_slice(hipStream_t, const void *, bool, long long, long long, int, float *) { return hipSuccess; }
bool estep_packed_shape(int, int, int *, int *) { return false; }
hipError_t launch_estep_packed(hipStream_t, const EstepArgs &) { return hipSuccess; }
hipError_t launch_remap_row_offsets(hipStream_t, CallPair *pairs, long long n_pairs, unsigned row_bytes, const int *new_rows, unsigned *call_rows)
{
    // this one is cheap enough to do for real: it indexes new_rows with what the records hold
    for (long long i = 0; i < n_pairs; i++)
        for (int h = 0; h < 2; h++) {
            const unsigned row = (unsigned)new_rows[pairs[i].row_off[h] / row_bytes];
            pairs[i].row_off[h] = row * row_bytes;
            if (call_rows) call_rows[2 * i + h] = row;
        }
    return hipSuccess;
}
hipError_t launch_f64_to_f32(hipStream_t, const double *, float *, long long) { return hipSuccess; }
hipError_t launch_add_f32(hipStream_t, const float *, const float *, float *, long long) { return hipSuccess; }
hipError_t launch_f32_to_f64(hipStream_t, const float *, double *, long long) { return hipSuccess; }
hipError_t launch_delay(hipStream_t, long long) { return hipSuccess; }
hipError_t launch_post_compact_build(hipStream_t, const uint2 *, const float *, long long, int, unsigned cap, unsigned *block, float *, unsigned char *,
                                     unsigned *, unsigned long long, int, int)
{
    block[0] = cap + 1;  // (no kernels here: "overflow", so that the host logic takes the whole-table path it can execute)
    return hipSuccess;
}
hipError_t launch_prob_changes_build(hipStream_t, const float *, float *, long long, int, unsigned cap, unsigned *block, unsigned *,
                                     unsigned long long, int, int)
{
    block[0] = cap + 1;  // (no kernels here: "overflow" - the whole slices travel, which the host logic can execute)
    return hipSuccess;
}
hipError_t launch_prob_changes_apply(hipStream_t, float *, const unsigned *, unsigned long long, long long, int, int, int, unsigned, unsigned short *) { return hipSuccess; }
hipError_t launch_post_counts(hipStream_t, const unsigned *blocks, unsigned long long block_words, int nranks, unsigned *out, unsigned seq)
{
    for (int r = 0; r < nranks; r++) out[r] = blocks[(size_t)r * block_words];
    out[nranks] = seq;
    return hipSuccess;
}
hipError_t launch_post_reconstruct(hipStream_t, const uint2 *, float *, const unsigned *, unsigned long long, long long, int, int, int, unsigned, uint2 *) { return hipSuccess; }
hipError_t launch_prior_betas(hipStream_t, const float *, float *, const unsigned long long *, const int *, const int *, const int *,
                              long long, int, double, float *) { return hipSuccess; }
hipError_t launch_rebuild_nz(hipStream_t, const float *, long long, int, int, float, unsigned long long *, uint2 *) { return hipSuccess; }
hipError_t launch_assign(hipStream_t, const float *, long long, int, int *, float *) { return hipSuccess; }
hipError_t launch_test_log(hipStream_t, const float *, float *, long long) { return hipSuccess; }
hipError_t launch_test_log_hot(hipStream_t, const float *, float *, long long) { return hipSuccess; }
hipError_t launch_test_log2_hw(hipStream_t, const float *, float *, long long) { return hipSuccess; }
hipError_t launch_test_exp(hipStream_t, const float *, float *, long long) { return hipSuccess; }
hipError_t launch_test_softmax(hipStream_t, const float *, float *, long long, int) { return hipSuccess; }

// ------------------------------------------------------------------------------------
// The layouts repack_device.hip derives on the GPU, derived here with the same definitions (csrc/kernels.h, the
// kernels of repack_device.hip) so that what dmx_api.cpp does with them - sizes, the variant slices of the
// exchange, the row remap - walks real structures.
// ------------------------------------------------------------------------------------
int repack_on_device(dmx_ctx *c, const int32_t *variant, const int32_t *cb, const float *p)
{
    const long long B = c->B, V = c->V, N = c->N;
    const int G = c->G;
    c->item_calls = item_calls_for(N);
    for (long long i = 0; i < N; i++) {
        if ((unsigned)cb[i] >= (unsigned long long)B) return fail(DMX_ERR_INVALID, "compressed_cb[%lld]=%d outside [0,%lld)", i, cb[i], B);
        if ((unsigned)variant[i] >= (unsigned long long)V) return fail(DMX_ERR_INVALID, "variant_id[%lld]=%d outside [0,%lld)", i, variant[i], V);
        if (!(p[i] >= 0.0f && p[i] <= 1.0f)) return fail(DMX_ERR_INVALID, "p_base_wrong[%lld]=%g outside [0,1]", i, (double)p[i]);
    }
    std::vector<long long> perm_b((size_t)N), perm_v((size_t)N);
    std::iota(perm_b.begin(), perm_b.end(), 0LL);
    std::iota(perm_v.begin(), perm_v.end(), 0LL);
    std::stable_sort(perm_b.begin(), perm_b.end(), [&](long long x, long long y) { return cb[x] < cb[y]; });
    std::stable_sort(perm_v.begin(), perm_v.end(), [&](long long x, long long y) { return variant[x] < variant[y]; });
    std::vector<long long> row_start((size_t)B + 1, 0), col_ptr((size_t)V + 1, 0);
    for (long long i = 0; i < N; i++) {
        row_start[(size_t)cb[i] + 1]++;
        col_ptr[(size_t)variant[i] + 1]++;
    }
    for (long long b = 0; b < B; b++) row_start[(size_t)b + 1] += row_start[(size_t)b];
    for (long long v = 0; v < V; v++) col_ptr[(size_t)v + 1] += col_ptr[(size_t)v];
    std::vector<long long> pair_ptr((size_t)B + 1, 0), item_ptr((size_t)V + 1, 0);
    for (long long b = 0; b < B; b++) pair_ptr[(size_t)b + 1] = pair_ptr[(size_t)b] + ((row_start[(size_t)b + 1] - row_start[(size_t)b] + 7) / 8) * 4;
    for (long long v = 0; v < V; v++)
        item_ptr[(size_t)v + 1] = item_ptr[(size_t)v] + (col_ptr[(size_t)v + 1] - col_ptr[(size_t)v] + c->item_calls - 1) / c->item_calls;
    c->n_pairs = pair_ptr[(size_t)B];
    c->n_items = item_ptr[(size_t)V];
    c->max_row_calls = 0;
    std::vector<int> bc_order((size_t)B);
    std::iota(bc_order.begin(), bc_order.end(), 0);
    std::stable_sort(bc_order.begin(), bc_order.end(), [&](int x, int y) {
        return row_start[(size_t)x + 1] - row_start[(size_t)x] > row_start[(size_t)y + 1] - row_start[(size_t)y];
    });
    DMX_TRY(dev_alloc(c, &c->d_pair_ptr, (size_t)B + 1));
    DMX_TRY(dev_alloc(c, &c->d_item_ptr, (size_t)V + 1));
    DMX_TRY(dev_alloc(c, &c->d_bc_order, (size_t)B));
    std::memcpy(c->d_pair_ptr, pair_ptr.data(), sizeof(long long) * ((size_t)B + 1));
    std::memcpy(c->d_item_ptr, item_ptr.data(), sizeof(long long) * ((size_t)V + 1));
    if (B) std::memcpy(c->d_bc_order, bc_order.data(), sizeof(int) * (size_t)B);
    const long long padded_pairs = c->n_pairs + CALL_PAD_PAIRS;
    DMX_TRY(dev_alloc(c, &c->d_call_pairs, (size_t)padded_pairs));
    DMX_TRY(dev_alloc(c, &c->d_call_rows, (size_t)padded_pairs * 2));
    for (long long i = 0; i < padded_pairs; i++) {
        CallPair &pr = c->d_call_pairs[i];
        pr.row_off[0] = pr.row_off[1] = 0u;
        pr.keep[0] = pr.keep[1] = 0.0f;
        pr.floor[0] = pr.floor[1] = 1.0f;
        pr.reserved[0] = pr.reserved[1] = 0u;
    }
    std::memset(c->d_call_rows, 0, sizeof(unsigned) * (size_t)padded_pairs * 2);
    for (long long s = 0; s < N; s++) {
        const long long i = perm_b[(size_t)s];
        const long long b = cb[i], j = s - row_start[(size_t)b];
        CallPair &pr = c->d_call_pairs[pair_ptr[(size_t)b] + (j >> 1)];
        const int h = (int)(j & 1);
        pr.row_off[h] = (unsigned)variant[i] * (unsigned)G * 4u;
        c->d_call_rows[2 * (pair_ptr[(size_t)b] + (j >> 1)) + h] = (unsigned)variant[i];
        pr.keep[h] = 1.0f - p[i];
        pr.floor[h] = p[i] > 1e-4f ? p[i] : 1e-4f;
    }
    c->n_bins = 0;
    c->n_tiles = c->bin_rows_cap = 0;
    DMX_TRY(dev_alloc(c, &c->d_csc, (size_t)N));
    c->n_csc = N;
    for (long long s = 0; s < N; s++) {
        const long long i = perm_v[(size_t)s];
        const float keep = 1.0f - p[i];
        unsigned bits;
        std::memcpy(&bits, &keep, 4);
        c->d_csc[s] = make_uint2((unsigned)cb[i], bits);
    }
    DMX_TRY(dev_alloc(c, &c->d_item_start, (size_t)c->n_items));
    DMX_TRY(dev_alloc(c, &c->d_item_len, (size_t)c->n_items));
    DMX_TRY(dev_alloc(c, &c->d_item_order, (size_t)c->n_items));
    DMX_TRY(dev_alloc(c, &c->d_item_variant, (size_t)c->n_items));
    for (long long v = 0; v < V; v++) {
        long long it = item_ptr[(size_t)v];
        for (long long s = col_ptr[(size_t)v]; s < col_ptr[(size_t)v + 1]; s += c->item_calls, it++) {
            c->d_item_variant[it] = (int)v;
            c->d_item_start[it] = s;
            c->d_item_len[it] = (int)std::min<long long>(c->item_calls, col_ptr[(size_t)v + 1] - s);
        }
    }
    std::vector<int> order((size_t)c->n_items);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return c->d_item_len[x] > c->d_item_len[y]; });
    if (c->n_items) std::memcpy(c->d_item_order, order.data(), sizeof(int) * (size_t)c->n_items);
    return 0;
}

// multi-GPU, M-step records by variant slice: the same derivation as repack_device.hip's, on the host
int wire_records_of(dmx_ctx *c, long long row_base, uint4 *out, long long capacity)
{
    std::memset(out, 0, sizeof(uint4) * (size_t)capacity);
    for (long long it = 0; it < c->n_items; it++)
        for (int i = 0; i < c->d_item_len[it]; i++) {
            const long long s = c->d_item_start[it] + i;
            out[s] = make_uint4((unsigned)c->d_item_variant[it], c->d_csc[s].x + (unsigned)row_base, c->d_csc[s].y, 1u);
        }
    return 0;
}

// the tile-major M-step records are a device-side layout: the host build stays with the work-item form
int build_mstep_tiles(dmx_ctx *c, long long, long long)
{
    c->mt_tried = true;
    return 0;
}
int build_slice_row_index(dmx_ctx *c)
{
    c->slice_index_tried = true;
    return 0;
}
int plan_mstep_shifts(dmx_ctx *c)
{
    c->mt_shift_tried = true;  // (no exponents: the float64 work-item form)
    return 0;
}
void release_mstep_tiles(dmx_ctx *c)
{
    c->n_mt = 0;
    c->mt_tried = false;
    c->mt_shift_tried = false;
}

int install_mstep_records(dmx_ctx *c, const uint4 *rec, long long n, long long v_lo, long long v_hi)
{
    const long long V = c->V;
    std::vector<long long> keep;
    for (long long i = 0; i < n; i++)
        if (rec[i].w != 0u && rec[i].x >= (unsigned)v_lo && rec[i].x < (unsigned)v_hi) keep.push_back(i);
    std::stable_sort(keep.begin(), keep.end(), [&](long long x, long long y) { return rec[x].x < rec[y].x; });
    const long long m = (long long)keep.size();
    dev_free(c, &c->d_csc, (size_t)c->n_csc);
    dev_free(c, &c->d_item_start, (size_t)c->n_items);
    dev_free(c, &c->d_item_len, (size_t)c->n_items);
    dev_free(c, &c->d_item_order, (size_t)c->n_items);
    dev_free(c, &c->d_item_variant, (size_t)c->n_items);
    dev_free(c, &c->d_partial, (size_t)c->n_items * c->G);
    dev_free(c, &c->d_redo, c->cap_redo);
    c->item_calls = item_calls_for(m);
    std::vector<long long> col_ptr((size_t)V + 1, 0), item_ptr((size_t)V + 1, 0);
    for (long long i : keep) col_ptr[(size_t)rec[i].x + 1]++;
    for (long long v = 0; v < V; v++) col_ptr[(size_t)v + 1] += col_ptr[(size_t)v];
    for (long long v = 0; v < V; v++)
        item_ptr[(size_t)v + 1] = item_ptr[(size_t)v] + (col_ptr[(size_t)v + 1] - col_ptr[(size_t)v] + c->item_calls - 1) / c->item_calls;
    c->n_items = item_ptr[(size_t)V];
    c->n_csc = m;
    std::memcpy(c->d_item_ptr, item_ptr.data(), sizeof(long long) * ((size_t)V + 1));
    DMX_TRY(dev_alloc(c, &c->d_csc, (size_t)m));
    for (long long s = 0; s < m; s++) c->d_csc[s] = make_uint2(rec[keep[(size_t)s]].y, rec[keep[(size_t)s]].z);
    DMX_TRY(dev_alloc(c, &c->d_item_start, (size_t)c->n_items));
    DMX_TRY(dev_alloc(c, &c->d_item_len, (size_t)c->n_items));
    DMX_TRY(dev_alloc(c, &c->d_item_order, (size_t)c->n_items));
    DMX_TRY(dev_alloc(c, &c->d_item_variant, (size_t)c->n_items));
    DMX_TRY(dev_alloc(c, &c->d_partial, (size_t)c->n_items * c->G));
    c->cap_redo = ((size_t)c->n_items / 2 + 1) * (size_t)c->G;
    DMX_TRY(dev_alloc(c, &c->d_redo, c->cap_redo));
    for (long long v = 0; v < V; v++) {
        long long it = item_ptr[(size_t)v];
        for (long long s = col_ptr[(size_t)v]; s < col_ptr[(size_t)v + 1]; s += c->item_calls, it++) {
            c->d_item_variant[it] = (int)v;
            c->d_item_start[it] = s;
            c->d_item_len[it] = (int)std::min<long long>(c->item_calls, col_ptr[(size_t)v + 1] - s);
        }
    }
    std::vector<int> order((size_t)c->n_items);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return c->d_item_len[x] > c->d_item_len[y]; });
    if (c->n_items) std::memcpy(c->d_item_order, order.data(), sizeof(int) * (size_t)c->n_items);
    return 0;
}

// the device pack and the staged containers have no host twin in this build
int pack_on_device(dmx_ctx *, long long, const int *, const int *, const unsigned char *, long long, const int *, const int *,
                   const unsigned char *, const int *, const float *, long long *, long long *, long long *)
{
    return fail(DMX_ERR_UNSUPPORTED, "device pack: not part of the host sanitizer build");
}
int stage_containers_on_device(dmx_ctx *, const dmx_call_container *, int) { return fail(DMX_ERR_UNSUPPORTED, "staging: not part of the host sanitizer build"); }
int pack_staged_on_device(dmx_ctx *, long long, const int *, const int *, const unsigned char *, const int *, int, long long *, long long *,
                          long long *)
{
    return fail(DMX_ERR_UNSUPPORTED, "device pack: not part of the host sanitizer build");
}
void release_staged_calls(dmx_ctx *) {}
int pack_containers_on_device(dmx_ctx *, long long, const int *, const int *, const unsigned char *, const dmx_call_container *, int,
                              long long *, long long *, long long *)
{
    return fail(DMX_ERR_UNSUPPORTED, "device pack: not part of the host sanitizer build");
}

}  // namespace dmx
