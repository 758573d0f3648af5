#!/usr/bin/env python
"""bench.py -- EM-iteration throughput of the Demultiplexer hot path on MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one EM iteration of learn_genotypes on device-resident inputs: P-step (beta ->
probability), E-step + softmax, M-step (+ RCCL all-reduce of the beta addition when N > 1).
Workload (BASELINE.json metric): 200k barcodes x 100k SNPs x 64 genotypes per GPU, synthetic
(demuxalot_amd/synth.py).  With N GPUs (--scaling weak, the default) every rank owns its
own 200k-barcode shard of an N x 200k-barcode experiment; the beta tables are replicated and the per-rank
beta additions are exchanged every iteration (reduce-scatter over variant slices, P-step on the owned slice,
all-gather of genotype_prob: include/demux_hip.h "Multi-GPU").  --scaling strong is BASELINE.json configs[3]
as written: ONE 200k-barcode experiment cut into N barcode ranges with equal numbers of calls.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (E-step kernel,
HIP-event timed on the stream it runs on) and `cpu_baseline` (the numpy oracle, one core, on a
bounded barcode sub-sample; rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (barcodes per GPU, SNPs, genotypes, doublet_prior, generator seed)
    'em_200k_100k_64': (200_000, 100_000, 64, 0.0, 1237),   # BASELINE.json metric / configs[3] shape
    'em_200k_100k_32': (200_000, 100_000, 32, 0.0, 1236),   # configs[2]
    'predict_20k_20k_8': (20_000, 20_000, 8, 0.35, 1235),   # configs[1]
    'predict_200k_20k_8': (200_000, 20_000, 8, 0.35, 1245),  # configs[1]'s option table (K = 36) with enough barcodes for the packed form
    'predict_200k_20k_12': (200_000, 20_000, 12, 0.35, 1246),  # K = 78: 16 lanes x 5 slots
    'predict_60k_20k_8': (60_000, 20_000, 8, 0.35, 1247),    # in between: the longest rows on 64 lanes, the rest packed
    'em_20k_10k_64': (20_000, 10_000, 64, 0.0, 77),         # quick check
    'em_200k_4k_64': (200_000, 4_000, 64, 0.0, 78),         # diagnostic: 2 MB genotype table (every row gather hits L2)
    'predict_20k_20k_32_doublets': (20_000, 20_000, 32, 0.25, 1240),   # K = 528: workgroup-per-barcode kernel
    'predict_20k_20k_64_doublets': (20_000, 20_000, 64, 0.25, 1243),   # K = 2080: doublets of 64 genotypes
    'predict_5k_20k_128_doublets': (5_000, 20_000, 128, 0.25, 1241),   # K = 8256 (configs[4] option count)
    'em_130k_650k_128_doublets': (130_000, 650_000, 128, 0.25, 1242),  # one rank's share of configs[4] (1M x 650k x 128 on 8 GPUs)
}


def algorithmic_bytes(B, V, G, K, N):
    """compulsory HBM bytes, every input read once and every output written once."""
    e = 8 * N + 8 * (B + 1) + 4 * V * G + 8 * B * K      # calls (variant, p_wrong), row_ptr (int64), prob table, logits + posteriors
    m = 8 * N + 8 * (V + 1) + 4 * B * G + 4 * V * G      # calls (cb, p_wrong), col_ptr, singlet posteriors, addition
    p = 12 * V * G + 4 * V                               # prior + addition in, prob out, v2snp
    return dict(estep=e, mstep=m, pstep=p, iteration=e + m + p)


def _oracle_iteration(demux_oracle, problem, betas, doublet_prior, n_sample):
    G = problem.n_genotypes
    v, cb, e = problem.subset_barcodes(0, n_sample)
    t0 = time.perf_counter()
    prob = demux_oracle.probs_from_betas(problem.v2snp, betas, 0.01)
    logits = demux_oracle.barcode_logits(v, cb, e, prob, n_sample, doublet_prior)
    post = demux_oracle.softmax_rows(logits)
    demux_oracle.beta_addition(v, cb, e, post, problem.n_variants, G)
    return time.perf_counter() - t0, len(v), logits, post


def cpu_baseline(problem, betas, doublet_prior, target_seconds=15.0):
    """The numpy oracle (same passes as the reference: K column passes + bincount; G passes for the
    M-step) on ONE core -- the reference path is single-threaded -- on the first barcodes of the
    workload. A small probe calibrates the sample so that the timed run takes ~10-30 s."""
    from oracle import demux_oracle
    n_probe = min(problem.n_barcodes, 1000)
    t_probe, calls_probe, _, _ = _oracle_iteration(demux_oracle, problem, betas, doublet_prior, n_probe)
    # the probe pays the fixed O(V*G) P-step too; scale only the per-call part
    t_fixed = 0.0
    if problem.n_barcodes > n_probe:
        t0 = time.perf_counter()
        demux_oracle.probs_from_betas(problem.v2snp, betas, 0.01)
        t_fixed = time.perf_counter() - t0
    per_barcode = max(1e-7, (t_probe - t_fixed) / n_probe)
    n_sample = int(max(n_probe, min(problem.n_barcodes, (target_seconds - t_fixed) / per_barcode)))
    dt, n_calls, logits, post = _oracle_iteration(demux_oracle, problem, betas, doublet_prior, n_sample)
    return dict(value=n_sample / dt, unit='barcodes/s', cores=1, kind='port',
                sample=f'first {n_sample} barcodes ({n_calls} calls) of the workload, full V and G, '
                       f'one EM iteration in {dt:.1f} s, numpy single-threaded like the reference',
                host_cores=os.cpu_count()), logits, post, n_sample


# VALU issue model of the exact-mode E-step term (direct kernels, K <= 1024), from the instruction count of the main
# loop of k_estep_direct<64,1,false,8,false> per two terms (ISA: 124 VALU per 8 calls = 15.5 per term: 17 packed
# float32, 6 plain integer, 2 v_cvt_f32_i32, 2 v_rcp_f32, 2 v_cvt_f64_f32, 2 v_add_f64) and the issue costs measured by
# scripts/valu_issue_bench.hip on this chip with >= 2 waves per SIMD (profiles/r2_valu_issue_bench.txt): plain 2.4
# cycles, packed float32 / float64 / conversions 4.4, transcendental 8.3.
VALU_CYCLES_PER_TERM = (17 * 4.4 + 6 * 2.4 + 2 * 4.4 + 2 * 8.3 + 2 * 4.4 + 2 * 4.4) / 2
N_SIMD, PEAK_CLOCK_HZ = 1024, 2.4e9


def roofline(workload, ab, e_ms, m_ms, timers, N, G, K, form='direct'):
    """The contract's HBM figures for the dominant kernel (the E-step) on ALGORITHMIC bytes, plus what actually binds
    it: VALU issue of the N*K numpy-exact float32 log terms."""
    achieved = ab['estep'] / (e_ms * 1e-3) / 1e9
    per_launch = lambda name: timers[name]['ms'] / max(1, timers[name]['launches'])
    terms_per_s = N * K / (e_ms * 1e-3)
    peak_terms = N_SIMD * PEAK_CLOCK_HZ * 64 / VALU_CYCLES_PER_TERM
    kernel = 'k_estep_block' if K > 1024 or (K > G and K > 512) else 'k_estep_direct'  # kernels.hip: launch_estep
    if form == 'packed':
        kernel = 'k_estep_packed'  # estep_packed.hip: narrow doublet tables, several option slots per lane
    return {
        'bound': 'valu-issue',
        'kernel': kernel, 'achieved': achieved, 'peak': 8000.0, 'unit': 'GB/s', 'frac': achieved / 8000.0,
        'traffic': measured_traffic(workload, kernel),
        'algorithmic_bytes_per_launch': ab['estep'],
        'iteration_bytes': ab['iteration'],
        'iteration_frac': ab['iteration'] / (1e-3 * (e_ms + m_ms + per_launch('pstep') + per_launch('mcombine'))) / 1e9 / 8000.0,
        'delivered_gather_GBps': (N * 4 * G) / (e_ms * 1e-3) / 1e9,
        'valu': {'log_terms_per_s': terms_per_s, 'peak_terms_per_s': peak_terms, 'frac': terms_per_s / peak_terms,
                 'cycles_per_term_model': VALU_CYCLES_PER_TERM,
                 'note': 'exact-mode term = numpy float32 log repeated operation for operation + float64 accumulate; issue costs '
                         'from profiles/r2_valu_issue_bench.txt; peak at the nominal 2.4 GHz (the kernel sustains ~2.0-2.1 GHz); '
                         'K > 1024 (k_estep_block) adds two LDS reads per term pair and option'},
        'note': 'frac is the HBM fraction the contract asks for (algorithmic bytes / time / 8 TB/s); the kernel is bound by '
                'VALU issue (valu.frac), not by HBM: see DESIGN.md 4',
    }


def measured_traffic(workload, kernel):
    """HBM-side bytes per launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE +
    WRITE_SIZE, see profiles/README.md); None when no profile of this workload/kernel is recorded."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        table = json.load(open(path))
        return table[workload][kernel]['bytes_per_launch']
    except (OSError, KeyError, ValueError):
        return None


def live_counters(args, kernel):
    """Measured now, by child runs of this script under `rocprofv3 --kernel-trace --pmc` (one counter per pass, as
    MI355X_MICROARCH.md prescribes): HBM-side bytes per launch of `kernel` (FETCH_SIZE + WRITE_SIZE, both reported in KiB;
    the row gathers are 4-byte-per-lane reads, not the 16-byte streams with the documented 2x under-count) and the clock
    the kernel sustains (GRBM_GUI_ACTIVE summed over the 8 XCDs / 8 / the kernel's duration in the same pass).
    {} when the profiler is not available; a missing key when a pass fails - the caller keeps the figures committed
    under profiles/ for those."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which('rocprofv3') is None:
        return {'skipped': 'rocprofv3 not found'}
    # never nest profilers: a bench run that is itself under rocprofv3 keeps to the committed figures
    nested = [k for k in os.environ if k.startswith(('ROCP_TOOL', 'ROCPROFILER_', 'ROCPROFV3_'))]
    if nested or 'rocprof' in os.environ.get('LD_PRELOAD', ''):
        return {'skipped': f'this run is itself under a profiler ({nested or "LD_PRELOAD"})'}
    out_dir = tempfile.mkdtemp(prefix='bench_pmc_', dir='/tmp')
    child = [sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', args.workload, '--steps', '2', '--warmup', '1',
             '--no-cpu-baseline', '--no-fast-mode', '--no-live-traffic', '--no-e2e'] + (['--flat-genotypes'] if args.flat_genotypes else [])
    env = dict(os.environ, TMPDIR='/tmp')
    found = {}
    try:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE', 'GRBM_GUI_ACTIVE'):
            where = os.path.join(out_dir, counter)
            try:
                subprocess.run(['rocprofv3', '--kernel-trace', '--pmc', counter, '--output-format', 'csv', '-d', where, '--'] + child,
                               env=env, cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=180, check=True)
            except (OSError, subprocess.SubprocessError) as exc:
                found[f'{counter}_failed'] = (getattr(exc, 'stderr', b'') or b'').decode(errors='replace')[-300:] or str(exc)
                continue
            values, dispatches = [], set()
            for path in glob.glob(os.path.join(where, '**', '*counter_collection.csv'), recursive=True):
                for row in csv.DictReader(open(path)):
                    if kernel in row['Kernel_Name'] and row['Counter_Name'] == counter:
                        values.append(float(row['Counter_Value']))
                        dispatches.add(row['Dispatch_Id'])
            if not values:  # e.g. another E-step kernel ran: no figure rather than a wrong one
                continue
            found[counter] = sum(values) / len(dispatches)
            if counter == 'GRBM_GUI_ACTIVE':
                spans = []
                for path in glob.glob(os.path.join(where, '**', '*kernel_trace.csv'), recursive=True):
                    for row in csv.DictReader(open(path)):
                        if kernel in row['Kernel_Name']:
                            spans.append(float(row['End_Timestamp']) - float(row['Start_Timestamp']))
                if spans:
                    found['clock_ghz'] = found[counter] / 8.0 / (sum(spans) / len(spans))  # cycles per XCD / ns
        if 'FETCH_SIZE' in found and 'WRITE_SIZE' in found:
            found['traffic'] = 1024.0 * (found['FETCH_SIZE'] + found['WRITE_SIZE'])
        return found
    except (OSError, KeyError, ValueError):
        return found
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)


def e2e_timing(problem, doublet_prior, n_iterations=5):
    """Wall time of the drop-in calls themselves (demux.py:35-66, 120-156: the reference's containers in, DataFrames out)
    on this workload, and where it goes.  SURVEY.md 8d asks for barcodes/s with and without the D2H of the [B, K] results:
    `on_device=True` keeps them on the GPU behind a DevicePosteriors (assignments computed there)."""
    import pandas as pd
    from demuxalot_amd import Demultiplexer, synth
    from demuxalot_amd.demux import _option_names, _pack_on_device
    from demuxalot_amd.device import DeviceContext
    t0 = time.perf_counter()
    calls, genotypes, handler = synth.as_objects(problem)
    t_objects = time.perf_counter() - t0
    B = handler.n_barcodes
    n_molecule_calls = int(sum(c.n_snp_calls for c in calls.values()))
    Demultiplexer.predict_posteriors(calls, genotypes, handler, doublet_prior=doublet_prior)  # warm-up: context, allocations

    def timed(fn):
        t = time.perf_counter()
        out = fn()
        return time.perf_counter() - t, out

    t_predict, (logits_df, probs_df) = timed(lambda: Demultiplexer.predict_posteriors(calls, genotypes, handler, doublet_prior=doublet_prior))
    t_predict_dev, dev = timed(lambda: Demultiplexer.predict_posteriors(calls, genotypes, handler, doublet_prior=doublet_prior, on_device=True))
    t_assign_dev, assigned = timed(lambda: dev.assignments(0.9))
    t_assign_host, assigned_host = timed(lambda: probs_df[probs_df.max(axis=1).gt(0.9)].idxmax(axis=1))
    same = bool(assigned.index.equals(assigned_host.index) and (assigned.values == assigned_host.values).all())
    dev.close()
    t_learn, (_learnt, last_df) = timed(lambda: Demultiplexer.learn_genotypes(calls, genotypes, handler, n_iterations=n_iterations, doublet_prior=0.))
    t_learn_dev, (_learnt2, dev2) = timed(lambda: Demultiplexer.learn_genotypes(calls, genotypes, handler, n_iterations=n_iterations, doublet_prior=0., on_device=True))
    dev2.close()
    # the same steps by hand, for the split
    pen = Demultiplexer._doublet_penalties(genotypes.n_genotypes, doublet_prior)
    from demuxalot_amd.device import acquire_private_context, release_private_context
    ctx = acquire_private_context()  # what the calls themselves take: a pooled context, its device blocks re-used
    try:
        t_pack, _ = timed(lambda: (_pack_on_device(calls, genotypes, B, False, fetch_betas=False, ctx=ctx), ctx.synchronize()))
        ctx.set_addition(None)
        t_pe, _ = timed(lambda: (ctx.probs_from_betas(0.01, fetch=False), ctx.estep(pen, with_doublets=doublet_prior > 0, fetch_logits=False, fetch_probs=False), ctx.synchronize()))
        t_d2h, (lg, pr) = timed(lambda: (ctx.get_logits(), ctx.get_probs()))
        columns = _option_names(genotypes.genotype_names, doublet_prior)
        t_frames, _ = timed(lambda: (pd.DataFrame(lg, index=list(handler.ordered_barcodes), columns=columns),
                                     pd.DataFrame(pr, index=list(handler.ordered_barcodes), columns=columns)))
        pen0 = Demultiplexer._doublet_penalties(genotypes.n_genotypes, 0.)
        t_em, _ = timed(lambda: (ctx.em(n_iterations, 0.01, pen0, with_doublets=False, fetch_logits=False, fetch_probs=False), ctx.synchronize()))
    finally:
        release_private_context(ctx)
    return {
        'workload_molecule_calls': n_molecule_calls, 'barcodes': B,
        'predict_posteriors_s': t_predict, 'predict_barcodes_per_s': B / t_predict,
        'predict_posteriors_on_device_s': t_predict_dev, 'predict_on_device_barcodes_per_s': B / t_predict_dev,
        'assignments_on_device_s': t_assign_dev, 'assignments_with_pandas_s': t_assign_host, 'assignments_identical': same,
        f'learn_genotypes_{n_iterations}it_s': t_learn, f'learn_genotypes_{n_iterations}it_on_device_s': t_learn_dev,
        'split_s': {'flatten_upload_device_pack_prior': t_pack, 'pstep_estep': t_pe, 'd2h_logits_and_posteriors': t_d2h,
                    'two_dataframes': t_frames, f'em_{n_iterations}_iterations_fused': t_em},
        'build_objects_s (synthetic generator, not part of a call)': t_objects,
        'note': 'wall time of the Python entry points on the containers of the whole workload (one molecule per call); the reference '
                'spends ~3.5-4 us per molecule call in pack_calls alone (SURVEY.md 8a9)',
    }


def _lib_device_count():
    from demuxalot_amd import _lib
    return _lib.device_count()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default='em_200k_100k_64', choices=sorted(WORKLOADS))
    ap.add_argument('--reduce-dtype', default='auto', choices=['auto', 'f64', 'f32'],
                    help='wire format of the reduce-scatter of the beta additions: float64 partial sums (results independent of the '
                         'number of ranks up to float32 rounding ties) or float32 (half the bytes; posteriors stay within the 1e-5 '
                         'contract: tests/test_gpu_ranks_on_one_gpu.py).  auto = f64 up to 2 ranks, f32 from 4 on, where the exchange '
                         'is what strong scaling runs into (DESIGN.md 5)')
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'],
                    help='weak: the workload per GPU; strong: the workload in total, barcodes sharded over the GPUs')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-fast-mode', action='store_true', help='skip the second timed region (tolerance-mode E-step)')
    ap.add_argument('--no-live-traffic', action='store_true',
                    help='roofline.traffic from profiles/pmc_traffic.json instead of two rocprofv3 --pmc child runs')
    ap.add_argument('--no-e2e', action='store_true', help='skip the end-to-end timing of the drop-in calls (containers in, DataFrames out)')
    ap.add_argument('--flat-genotypes', action='store_true',
                    help='worst case of the M-step: all-equal betas, so every posterior is 1/G and every call contributes to '
                         'every genotype (the start-from-assignment scenario of tests/test_synthetic.py:200-239 before any label '
                         'is used); the additions are then equal for all genotypes, so the posteriors stay uniform iteration after iteration')
    ap.add_argument('--force-dist', action='store_true',
                    help='testing aid: run the multi-rank control/collective path even with one rank')
    ap.add_argument('--host-plane', action='store_true',
                    help='the per-iteration exchange staged through host memory over the control plane instead of RCCL '
                         '(dmx_comm_init_host): several ranks on ONE GPU, hosts without a usable RCCL fabric')
    args = ap.parse_args()

    # gloo and RCCL print banners on stdout; stdout must carry the one JSON line only, so fd 1 is pointed
    # at stderr for the whole run and the line is written to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.reduce_dtype == 'auto':
        args.reduce_dtype = 'f32' if world >= 4 else 'f64'
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run'

    # Control plane (rendezvous, RCCL unique id, barrier, max-reduce of the wall time): plain sockets
    # (demuxalot_amd/plane.py).  torch.distributed.run is only the LAUNCHER: the workers never import torch, whose
    # ROCm wheels carry their own libamdhip64 / librccl - two HIP runtimes in one process is what the library refuses.
    # The data-plane collectives are RCCL inside libdemux_hip.so (or, with --host-plane, staged through the plane).
    plane = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        from demuxalot_amd.plane import SocketControlPlane
        plane = SocketControlPlane(rank, world, os.environ.get('MASTER_ADDR', '127.0.0.1'), host_collectives=args.host_plane)

    from demuxalot_amd import Demultiplexer, synth
    from demuxalot_amd.device import DeviceContext

    B, S, G, dp, seed = WORKLOADS[args.workload]
    B_total = B  # barcodes of the whole job: per GPU x GPUs (weak) or the workload's own count (strong)
    t_gen = time.perf_counter()
    if args.scaling == 'strong' and world > 1:
        from demuxalot_amd.distributed import partition_barcodes
        whole = synth.generate(B, S, G, doublets=dp > 0, seed=seed)  # the same experiment on every rank
        betas = whole.prior_betas(add_data_prior=False)
        bounds = partition_barcodes(np.bincount(whole.compressed_cb, minlength=B), world)
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        v_s, cb_s, e_s = whole.subset_barcodes(lo, hi)
        problem = synth.SyntheticProblem(hi - lo, S, G, whole.v2snp, whole.raw_betas, v_s, cb_s, e_s, whole.truth[lo:hi])
        del whole
        B = hi - lo
    else:
        problem = synth.generate(B, S, G, doublets=dp > 0, seed=seed, seed_calls=seed * 1000 + rank)
        betas = problem.prior_betas(add_data_prior=False)  # identical on every rank
        B_total = B * world
    if args.flat_genotypes:
        betas = np.ones_like(betas)
    t_gen = time.perf_counter() - t_gen
    V, N = problem.n_variants, problem.n_calls
    pen = Demultiplexer._doublet_penalties(G, dp)
    K = len(pen)

    ctx = DeviceContext(local_rank % max(1, _lib_device_count()))
    runtimes, rccl_fallback = None, None
    if use_dist and args.host_plane:
        ctx.comm_init_host(rank, world, plane.host_collective, reduce_dtype=args.reduce_dtype)
    elif use_dist:
        # RCCL communicator; when creating it fails on some rank (no usable fabric, library missing), every rank learns
        # so over the control plane and the run goes on with the exchange staged through host memory - said in the line
        unique_id, why = None, ''
        if rank == 0:
            try:
                unique_id = DeviceContext.new_unique_id()
            except Exception as exc:  # noqa: BLE001
                unique_id, why = b'', f'{type(exc).__name__}: {exc}'
        unique_id = plane.broadcast_bytes(unique_id)  # empty: rank 0 has no RCCL
        try:
            if not unique_id:
                raise RuntimeError('rank 0 could not create an RCCL unique id ' + why)
            ctx.comm_init(rank, world, unique_id, reduce_dtype=args.reduce_dtype)
            ok = True
        except Exception as exc:  # noqa: BLE001
            ok, why = False, f'{type(exc).__name__}: {exc}'
        ok, why = plane.all_ok(ok, why)
        if not ok:
            print(f'[bench] RCCL communicator not created ({why}); exchange staged through host memory', file=sys.stderr, flush=True)
            ctx.close()
            ctx = DeviceContext(local_rank % max(1, _lib_device_count()))
            ctx.comm_init_host(rank, world, plane._host_collective, reduce_dtype=args.reduce_dtype)
            args.host_plane = True
            rccl_fallback = why
    if use_dist:
        from demuxalot_amd import _lib
        runtimes = _lib.runtime_info()
        assert len(runtimes['hip']) == 1, f'more than one HIP runtime mapped: {runtimes}'
        assert 'torch' not in sys.modules
    t_up = time.perf_counter()
    ctx.set_problem(B, V, G, problem.variant_id, problem.compressed_cb, problem.p_base_wrong, problem.v2snp)
    ctx.set_betas(betas)
    t_up = time.perf_counter() - t_up

    # first pass fixes the options and gives the outputs used for the sanity check below
    ctx.set_addition(None)
    ctx.probs_from_betas(0.01, fetch=False)
    logits0, probs0 = ctx.estep(pen, with_doublets=dp > 0)

    def barrier():
        if plane is not None:
            plane.barrier()

    ctx.run_iterations(args.warmup, 0.01)
    ctx.synchronize()
    ctx.reset_timings()
    barrier()
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.run_iterations(args.steps, 0.01)
    ctx.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if plane is not None:
        elapsed = plane.max_float64(elapsed)
    timers = ctx.timings()
    em_form = ctx.estep_form()[0]  # what the E-steps of the timed iterations ran (direct | packed)
    guard_stats = ctx.guard_stats()  # (last, total, rows) of the guarded E-steps of the timed region

    # the tolerance-mode E-step (dmx_set_estep_mode: assignments identical, posteriors within the contract's 1e-5)
    # with the fast summation mode, timed the same way on the same resident problem; the default (bit-exact)
    # mode above stays the headline `value`
    fast = None
    if not args.no_fast_mode:
        ctx.set_estep_mode('fast')
        ctx.set_exact_additions(False)
        ctx.set_addition(None)
        ctx.probs_from_betas(0.01, fetch=False)
        _lf, probs_fast = ctx.estep(pen, with_doublets=dp > 0)
        ctx.run_iterations(args.warmup, 0.01)
        ctx.synchronize()
        ctx.reset_timings()
        barrier()
        ctx.synchronize()
        t0f = time.perf_counter()
        ctx.run_iterations(args.steps, 0.01)
        ctx.synchronize()
        barrier()
        elapsed_fast = time.perf_counter() - t0f
        if plane is not None:
            elapsed_fast = plane.max_float64(elapsed_fast)
        timers_fast = ctx.timings()
        fast = dict(value=B_total * args.steps / elapsed_fast, ms_per_step=1e3 * elapsed_fast / args.steps,
                    em_iterations_per_s=args.steps / elapsed_fast,
                    kernel_ms={k: (v['ms'] / max(1, v['launches'])) for k, v in timers_fast.items()},
                    vs_exact_first_pass=dict(
                        argmax_identical=bool(np.array_equal(probs_fast.argmax(1), probs0.argmax(1))),
                        max_abs_posterior_diff=float(np.abs(probs_fast - probs0).max())))
        ctx.set_estep_mode('exact')
        ctx.set_exact_additions(os.environ.get('DEMUXALOT_AMD_EXACT_ADDITIONS', '1') not in ('0', ''))
        ctx.reset_timings()

    # predict_posteriors throughput on the same resident problem (P + E only, no beta addition: demux.py:120-156),
    # rank-local.  The genotype table is then the importers' (a handful of distinct values per row), which is the
    # case the dictionary form of the exact E-step exists for (csrc/estep_dict.hip); timed with the form on (default)
    # and off, same bits either way.
    predict = {}
    n_pred = max(3, args.steps // 2)
    for mode in ('auto', 'never'):
        ctx.set_estep_dictionary(mode)
        ctx.set_addition(None)
        ctx.probs_from_betas(0.01, fetch=False)
        ctx.estep(pen, with_doublets=dp > 0, fetch_logits=False, fetch_probs=False)  # untimed first pass
        ctx.synchronize()
        ctx.reset_timings()
        t1 = time.perf_counter()
        for _ in range(n_pred):
            ctx.probs_from_betas(0.01, fetch=False)
            ctx.estep(pen, with_doublets=dp > 0, fetch_logits=False, fetch_probs=False)
        ctx.synchronize()
        seconds = (time.perf_counter() - t1) / n_pred
        t_pred = ctx.timings()
        form, distinct = ctx.estep_form()
        predict[mode] = dict(seconds=seconds, form=form, distinct_values_per_row=distinct,
                             estep_ms=t_pred['estep']['ms'] / max(1, t_pred['estep']['launches']),
                             pstep_ms=t_pred['pstep']['ms'] / max(1, t_pred['pstep']['launches']))
    ctx.set_estep_dictionary('auto')
    predict_s = predict['auto']['seconds']

    if rank == 0:
        ab = algorithmic_bytes(B, V, G, K, N)
        e_ms = timers['estep']['ms'] / max(1, timers['estep']['launches'])
        m_ms = timers['mstep']['ms'] / max(1, timers['mstep']['launches'])
        achieved = ab['estep'] / (e_ms * 1e-3) / 1e9
        out = {
            'metric': f'EM iterations/sec + barcodes demuxed/sec, {WORKLOADS[args.workload][0] // 1000}k bc x {S // 1000}k SNP x {G} gt '
                      + ('per GPU' if args.scaling == 'weak' else f'in total over {world} GPUs') +
                      ': value = barcodes/s through full learn_genotypes EM iterations (P-step + E-step + softmax + '
                      'M-step [+ exchange]) = barcodes x em_iterations_per_s; predict-only rate in predict_barcodes_per_s',
            'value': B_total * args.steps / elapsed,
            'unit': 'barcodes/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None,
            'dtype': 'f32 terms, f64 accumulate', 'data': 'synthetic',
            'config': {'workload': args.workload, 'barcodes_total': B_total, 'barcodes_per_gpu': B, 'snps': S, 'variants': V, 'genotypes': G,
                       'options': K, 'calls_per_gpu': N, 'doublet_prior': dp,
                       'summation': 'fast (DEMUXALOT_AMD_EXACT_ADDITIONS=0)' if os.environ.get('DEMUXALOT_AMD_EXACT_ADDITIONS', '1') in ('0', '') else 'exact: additions bit-identical to the reference (default)',
                       'parallelism': f'barcode shards x{world}' + (f', {"host-staged" if args.host_plane else "RCCL"} reduce-scatter {args.reduce_dtype} + all-gather f32 of variant slices' if use_dist else ''),
                       'runtimes': runtimes, **({'rccl_fallback': rccl_fallback} if rccl_fallback else {})},
            'em_iterations_per_s': args.steps / elapsed,
            'predict_barcodes_per_s': B_total / predict_s,
            'predict': {'dictionary_form': dict(predict['auto'], estep_hbm_frac=algorithmic_bytes(B, V, G, K, N)['estep'] / (predict['auto']['estep_ms'] * 1e-3) / 8e12),
                        'direct_form': dict(predict['never'], estep_hbm_frac=algorithmic_bytes(B, V, G, K, N)['estep'] / (predict['never']['estep_ms'] * 1e-3) / 8e12),
                        'note': 'P-step + E-step on the table without beta addition (predict_posteriors, EM iteration 0); '
                                'estep_ms includes building the dictionary'},
            'kernel_ms': {k: (v['ms'] / max(1, v['launches'])) for k, v in timers.items()},
            'exchange_ms_per_step': timers['allreduce']['ms'] / max(1, args.steps),
            'roofline': roofline(args.workload, ab, e_ms, m_ms, timers, N, G, K, em_form),
            'guard': {'barcodes_redone_exactly': guard_stats[1], 'barcode_rows': guard_stats[2],
                      'fraction': guard_stats[1] / max(1, guard_stats[2])},
            'setup_s': {'generate': t_gen, 'upload': t_up},
            'fast_mode': fast,
        }
        if fast is not None:
            e_fast = fast['kernel_ms']['estep']
            fast['estep_hbm_frac'] = ab['estep'] / (e_fast * 1e-3) / 1e9 / 8000.0
            fast['delivered_gather_GBps'] = (N * 4 * G) / (e_fast * 1e-3) / 1e9
        out['roofline']['traffic_source'] = 'profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE of an earlier run)'
        if os.environ.get('DEMUXALOT_AMD_ESTEP_SCHEDULE', 'auto') == 'tiled' and out['roofline']['kernel'] == 'k_estep_direct':
            out['roofline']['kernel'] = 'k_estep_tiled'
            out['roofline']['traffic'], out['roofline']['traffic_source'] = None, 'none (no committed figure for this kernel)'
        if world == 1 and not args.no_live_traffic:
            live = live_counters(args, out['roofline']['kernel'])
            if 'traffic' in live:
                out['roofline']['traffic'] = live['traffic']
                out['roofline']['traffic_source'] = 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, child runs of this command (live)'
            if 'traffic' not in live or 'clock_ghz' not in live:
                out['roofline']['live_counters'] = {k: v for k, v in live.items() if k == 'skipped' or k.endswith('_failed')}
            if 'clock_ghz' in live:  # the VALU issue fraction at the clock the kernel actually holds
                valu = out['roofline']['valu']
                valu['sustained_clock_ghz'] = live['clock_ghz']
                valu['frac_at_sustained_clock'] = valu['frac'] * PEAK_CLOCK_HZ / (live['clock_ghz'] * 1e9)
                valu['clock_source'] = 'rocprofv3 --pmc GRBM_GUI_ACTIVE / 8 XCDs / kernel duration, child run of this command (live)'
        if world == 1 and not args.no_e2e and not args.flat_genotypes:
            ctx.close()  # the end-to-end calls bring their own contexts; free this one's 4 GB first
            out['e2e'] = e2e_timing(problem, dp)
        if world == 1 and not args.no_cpu_baseline:
            base, ref_logits, ref_post, n_s = cpu_baseline(problem, betas, dp)
            out['cpu_baseline'] = base
            out['cpu_baseline']['speedup_vs_gpu_value'] = out['value'] / base['value']
            # sanity: the GPU rows of the sampled barcodes equal the oracle's
            out['parity_on_sample'] = {
                'argmax_identical': bool(np.array_equal(ref_post.argmax(1), probs0[:n_s].argmax(1))),
                'max_abs_posterior_diff': float(np.abs(ref_post - probs0[:n_s]).max()),
                'logits_bitwise_equal': bool(np.array_equal(ref_logits.view(np.uint32), logits0[:n_s].view(np.uint32))),
            }
        else:
            out['cpu_baseline'] = None
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + '\n').encode())
    if plane is not None:
        plane.barrier()
        plane.close()


if __name__ == '__main__':
    main()
