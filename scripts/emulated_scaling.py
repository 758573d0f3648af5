"""Strong / weak scaling of the EM iteration with an EMULATED wire, on ONE GPU.

No multi-GPU node is available to the build, so what CAN be measured is measured: for n = 1, 2, 4, 8 this script
installs exactly what rank 0 of an n-rank run holds - its barcode range (strong: BASELINE.json configs[3], one 200k-barcode
experiment cut into n ranges with equal numbers of calls; weak: the whole 200k-barcode workload per rank), the padded
variant slices, the sliced P-step - and runs the timed EM iterations with the three collectives replaced by a
device-side copy of the rank's own block plus a one-wavefront kernel that holds the stream for the modelled wire time
(dmx_comm_init_emulated: latency + block bytes / link rate per collective, a direct exchange over a fully connected
node).  Every kernel, copy and stream dependency of the real exchange is executed; only the bytes on the wire are
replaced by their modelled duration.  Output: one JSON object (per n: ms per iteration, kernel times, time inside the
exchange, speed-up against n = 1) - profiles/r4_emulated_scaling.json is a run of it.

    python scripts/emulated_scaling.py [--link-gbps 50] [--latency-us 10] [--steps 20] [--modes guarded,exact]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from demuxalot_amd import Demultiplexer, synth  # noqa: E402
from demuxalot_amd.device import DeviceContext  # noqa: E402
from demuxalot_amd.distributed import partition_barcodes  # noqa: E402


def region(ctx, steps, warmup):
    """ms per iteration from a region WITHOUT phase timers (their events are barrier packets on the queue: 4 - 5 phase boundaries of
    3 - 5 us each weigh on a 0.3 ms iteration), the per-phase breakdown from a second region with them."""
    ctx.run_iterations(warmup, 0.01)
    ctx.synchronize()
    ctx.set_phase_timers(False)
    t0 = time.perf_counter()
    ctx.run_iterations(steps, 0.01)
    ctx.synchronize()
    elapsed = time.perf_counter() - t0
    ctx.set_phase_timers(True); ctx.reset_timings()
    t0 = time.perf_counter()
    ctx.run_iterations(steps, 0.01)
    ctx.synchronize()
    elapsed_timed = time.perf_counter() - t0
    timers = ctx.timings()
    ctx.set_phase_timers(False)
    return {'ms_per_step': 1e3 * elapsed / steps, 'ms_per_step_with_phase_timers': 1e3 * elapsed_timed / steps,
            'kernel_ms': {k: round(v['ms'] / max(1, steps), 4) for k, v in timers.items()}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--link-gbps', type=float, default=50.0, help='per peer link and direction (xGMI: ~50-60 GB/s achieved)')
    ap.add_argument('--latency-us', type=float, default=10.0, help='per collective')
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--ranks', default='1,2,4,8')
    ap.add_argument('--modes', default='guarded')
    ap.add_argument('--kinds', default='strong,weak')
    ap.add_argument('--exchange', default='', help="DEMUXALOT_AMD_EXCHANGE: '' (default: variant-sharded M-step) | reduce_scatter | allreduce")
    ap.add_argument('--workload', default='em_200k_100k_64', help='a workload of bench.py; with --kinds weak the whole workload is one rank\'s '
                    'share (em_130k_650k_128_doublets = one rank of BASELINE.json configs[4] on 8 GPUs)')
    ap.add_argument('--incremental', type=int, default=1, help="0: every M-step a full pass, as bench.py's headline regions run it (the library's default is 1)")
    args = ap.parse_args()
    import bench
    B, S, G, dp, seed = bench.WORKLOADS[args.workload]
    whole = synth.generate(B, S, G, doublets=dp > 0, seed=seed)
    betas = whole.prior_betas(add_data_prior=False)
    pen = Demultiplexer._doublet_penalties(G, dp)
    counts = np.bincount(whole.compressed_cb, minlength=B)
    out = {'workload': args.workload, 'link_gbytes_per_s': args.link_gbps, 'latency_us': args.latency_us, 'steps': args.steps,
           'exchange': args.exchange or 'variant-sharded M-step (default)', 'incremental_mstep': bool(args.incremental), 'runs': []}
    if args.exchange:
        os.environ['DEMUXALOT_AMD_EXCHANGE'] = args.exchange
    base = {}
    for mode in args.modes.split(','):
        for kind in args.kinds.split(','):
            for n in (int(x) for x in args.ranks.split(',')):
                wire = 'f64' if mode == 'exact' else 'f32'  # as bench.py chooses
                if kind == 'strong' and n > 1:
                    bounds = partition_barcodes(counts, n)
                    lo, hi = int(bounds[0]), int(bounds[1])
                    v, cb, e = whole.subset_barcodes(lo, hi)
                    problem = synth.SyntheticProblem(hi - lo, S, G, whole.v2snp, whole.raw_betas, v, cb, e, whole.truth[lo:hi])
                else:
                    problem = whole
                ctx = DeviceContext(0)
                try:
                    ctx.set_estep_mode(mode)
                    ctx.set_exact_additions(mode == 'exact')
                    ctx.set_mstep_tiles('always')  # as bench.py: the tile-major records are built during the warm-up
                    ctx.set_mstep_incremental(bool(args.incremental))
                    if n > 1:
                        ctx.comm_init_emulated(0, n, args.link_gbps, args.latency_us, reduce_dtype=wire)
                    ctx.set_problem(problem.n_barcodes, problem.n_variants, G, problem.variant_id, problem.compressed_cb, problem.p_base_wrong, problem.v2snp)
                    ctx.set_betas(betas)
                    ctx.set_addition(None)
                    ctx.probs_from_betas(0.01, fetch=False)
                    ctx.estep(pen, with_doublets=dp > 0, fetch_logits=False, fetch_probs=False)
                    r = region(ctx, args.steps, args.warmup)
                    r['compact_exchange (taken, overflowed, capacity rows)'] = ctx.exchange_compact()
                    r['compact_exchange of the table (taken, overflowed, capacity rows)'] = ctx.exchange_compact_table()
                finally:
                    ctx.close()
                total_barcodes = B if kind == 'strong' else B * n
                r.update(mode=mode, scaling=kind, n=n, wire=wire if n > 1 else None, barcodes_per_rank=problem.n_barcodes, calls_per_rank=problem.n_calls,
                         barcodes_per_s=total_barcodes / (r['ms_per_step'] * 1e-3))
                if n == 1:
                    base[(mode, kind)] = r['barcodes_per_s']
                ref = base.get((mode, kind)) or base.get((mode, 'strong')) or base.get((mode, 'weak'))
                if ref:
                    r['speedup_vs_1'] = r['barcodes_per_s'] / ref
                    r['efficiency'] = r['speedup_vs_1'] / n
                compute = sum(v for k, v in r['kernel_ms'].items() if k != 'allreduce')
                r['compute_ms'] = round(compute, 4)
                r['exchange_ms'] = r['kernel_ms']['allreduce']
                out['runs'].append(r)
                print(f"{mode:8s} {kind:6s} n={n}: {r['ms_per_step']:.3f} ms/it  compute {compute:.3f}  exchange {r['exchange_ms']:.3f}  "
                      f"speed-up {r.get('speedup_vs_1', 1):.2f}  {r['kernel_ms']}", file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
