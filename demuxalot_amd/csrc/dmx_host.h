// dmx_host.h -- host-side helpers of libdemux_hip.so shared by its translation units (not part of the ABI):
//   dmx_runtime.cpp   errors, the contexts' device-block cache, phase timers, context life cycle
//   dmx_exchange.cpp  RCCL (loaded on demand), the multi-GPU exchange and its set-up
//   dmx_steps.cpp     the step drivers: P-step, E-step (forms, guard levels, coarse pass), M-step (forms, incremental state)
//   dmx_api.cpp       problem install, switches, read-outs, results
#pragma once
#include <string>
#include <vector>

#include "dmx_ctx.h"

namespace dmx {
namespace host {

int bind(dmx_ctx *c);
TimerStamp *stamp_now(dmx_ctx *c);
void stamp_release(dmx_ctx *c, TimerStamp *s);
void timer_begin(dmx_ctx *c, int slot, TimerSpan *ev);
void timer_flush(dmx_ctx *c, int slot);
void timer_end(dmx_ctx *c, int slot, TimerSpan &ev);
void release_incremental(dmx_ctx *c);
void release_coarse_stream(dmx_ctx *c);
void release_problem(dmx_ctx *c);
int build_row_segments(dmx_ctx *c);
int ensure_options(dmx_ctx *c, int with_doublets, const float *penalties);
int upload_prior_logits(dmx_ctx *c, const void *prior, int dtype);
int copy_out(dmx_ctx *c, float *dst, const float *src, size_t count);
int need(dmx_ctx *c, bool cond, const char *what);
void exchange_slices(const int *v2snp, long long V, int n, std::vector<long long> &cut, long long &rows, bool &contiguous);
int host_stage(dmx_ctx *c, size_t bytes);
int host_collective(dmx_ctx *c, int op, const void *src, size_t off_in, size_t bytes_in, void *dst, size_t off_out, size_t bytes_out, size_t total_bytes, int64_t count, int dtype, const char *what, hipStream_t st);
int emulated_wire(dmx_ctx *c, size_t block_bytes, int rounds, hipStream_t st);
int wait_counts(dmx_ctx *c, unsigned *counts, int n, unsigned seq);
void coll_group_begin(dmx_ctx *c);
int coll_group_end(dmx_ctx *c);
int coll_reduce_scatter(dmx_ctx *c, const void *send, void *recv, size_t block, bool f64, hipStream_t st);
int coll_all_gather(dmx_ctx *c, float *table, size_t block, const char *what);
int coll_all_reduce(dmx_ctx *c, void *buf, size_t count, bool f64);
int gather_numbers(dmx_ctx *c, const long long *values, int count, std::vector<long long> &out);
int shard_mstep_by_variant(dmx_ctx *c, bool force);
int layout_exchange(dmx_ctx *c);
int copy_prob_out(dmx_ctx *c, float *dst);
int copy_prob_in(dmx_ctx *c, const float *src);
int ensure_full_addition(dmx_ctx *c);
bool coarse_capable(const dmx_ctx *c, int with_doublets, float lo);
int ensure_prob16(dmx_ctx *c);
int run_pstep(dmx_ctx *c, float lo, float hi, bool with_addition, bool with_half = false);
int prepare_dictionary(dmx_ctx *c, bool pairs, dmx::EstepArgs &a, int *form);
int run_estep(dmx_ctx *c, int with_doublets, bool with_prior, int prior_dtype, float power, bool logits_kept = true);
int gather_posteriors(dmx_ctx *c);
int run_mstep(dmx_ctx *c, float power);

// A phase that leaves through an error return between timer_begin and timer_end (HIP_TRY / DMX_TRY) still holds a reference to
// its opening stamp: without this the stamp never returned to idle_stamps and its event was never destroyed.
struct SpanGuard {
    dmx_ctx *c;
    TimerSpan *ev;
    ~SpanGuard()
    {
        if (ev->first != nullptr) {
            stamp_release(c, ev->first);
            ev->first = nullptr;
        }
    }
};

// dmx_exchange.cpp: the communicator of a context that is being destroyed (RCCL's, if one was created)
void comm_destroy(dmx_ctx *c);
// the message of the last failure on this thread (dmx_last_error)
const char *last_error();

}  // namespace host
}  // namespace dmx
