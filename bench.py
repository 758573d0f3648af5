#!/usr/bin/env python
"""bench.py -- EM-iteration throughput of the Demultiplexer hot path on MI355X.

    python bench.py                                   # 1 GPU
    python bench.py --gpus N --steps K --warmup W     # N GPUs of one node: starts its own N rank processes
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W     # the same under a launcher (RANK / WORLD_SIZE from the environment)

One "step" = one EM iteration of learn_genotypes on device-resident inputs: P-step (beta -> probability), E-step +
softmax, M-step (+ the exchange of the beta additions when N > 1).  Workload (BASELINE.json metric): 200k barcodes x
100k SNPs x 64 genotypes, synthetic (demuxalot_amd/synth.py).

N > 1, `scaling`: the headline `value` is BASELINE.json configs[3] AS WRITTEN - the ONE 200k-barcode experiment cut into
N barcode ranges with equal numbers of calls ("strong"); the `weak` object of the same line is the workload per GPU
(every rank a 200k-barcode shard of an N x 200k-barcode experiment).  The experiment is generated once (rank 0, handed to
the other ranks through /dev/shm).  The per-iteration exchange runs inside libdemux_hip.so (include/demux_hip.h
"Multi-GPU"): the M-step sharded on variants (all-gather of what it reads of every barcode; additions bit-identical to one
GPU) or, where that would move more bytes (weak scaling), the reduce-scatter of the per-rank sums; then the P-step on the
owned variant slice and the all-gather of genotype_prob.

E-step arithmetic: the library's default, the GUARDED mode (contract of the path proven per barcode, the rest redone
bit-exactly: include/demux_hip.h DMX_ESTEP_GUARDED); `exact_mode` = the same timed region with every logit / posterior
/ addition bit-identical to the reference.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel of the timed
iterations, HIP-event timed on the stream it runs on, live rocprofv3 counters) and `cpu_baseline` (the numpy oracle,
one core, on a bounded barcode sub-sample; N = 1 only).
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
    # name: (barcodes, SNPs, genotypes, doublet_prior, generator seed)
    'em_200k_100k_64': (200_000, 100_000, 64, 0.0, 1237),   # BASELINE.json metric / configs[3] shape
    'em_200k_100k_32': (200_000, 100_000, 32, 0.0, 1236),   # configs[2]
    'predict_20k_20k_8': (20_000, 20_000, 8, 0.35, 1235),   # configs[1]
    'predict_200k_20k_8': (200_000, 20_000, 8, 0.35, 1245),  # configs[1]'s option table (K = 36) with enough barcodes for the packed form
    'predict_200k_20k_12': (200_000, 20_000, 12, 0.35, 1246),  # K = 78: 16 lanes x 5 slots
    'predict_60k_20k_8': (60_000, 20_000, 8, 0.35, 1247),    # in between: the longest rows on 64 lanes, the rest packed
    'em_20k_10k_64': (20_000, 10_000, 64, 0.0, 77),         # quick check
    'em_25k_100k_64': (25_000, 100_000, 64, 0.0, 1237),     # the size of one rank's share of configs[3] on 8 GPUs
    'em_10k_100k_64': (10_000, 100_000, 64, 0.0, 1237),     # a single-sample experiment: the tile-major schedule's smallest sizes
    'em_50k_100k_64': (50_000, 100_000, 64, 0.0, 1237),
    'em_200k_4k_64': (200_000, 4_000, 64, 0.0, 78),         # diagnostic: 2 MB genotype table (every row gather hits L2)
    'predict_20k_20k_32_doublets': (20_000, 20_000, 32, 0.25, 1240),   # K = 528: workgroup-per-barcode kernel
    'predict_20k_20k_64_doublets': (20_000, 20_000, 64, 0.25, 1243),   # K = 2080: doublets of 64 genotypes
    'predict_5k_20k_128_doublets': (5_000, 20_000, 128, 0.25, 1241),   # K = 8256 (configs[4] option count)
    'em_130k_650k_128_doublets': (130_000, 650_000, 128, 0.25, 1242),  # one rank's share of configs[4] (1M x 650k x 128 on 8 GPUs)
    'em_1M_650k_128_doublets': (1_000_000, 650_000, 128, 0.25, 1242),  # configs[4] AS WRITTEN, the whole experiment on one GPU (K = 8256, ~4e8 calls)
}
SHARDED_GENERATOR_FROM = 500_000   # barcodes from which the experiment is generated in 16 shards by a pool of threads (synth.generate_sharded)

LOG_ISSUE_PEAK = 19.7e12   # v_log_f32 issue peak of the chip, 256 CUs x 4 SIMDs x 64 lanes / 8 cycles x 2.4 GHz
L2_GATHER_ALL_HIT_GBPS = 15800.0  # 256-byte rows out of L2 through buffer_load_dword: the same loop on a 2 MB table
N_SIMD, N_XCD = 1024, 8


def algorithmic_bytes(B, V, G, K, N):
    """compulsory HBM bytes, every input read once and every output written once."""
    e = 8 * N + 8 * (B + 1) + 4 * V * G + 8 * B * K      # calls (variant, p_wrong), row_ptr (int64), prob table, logits + posteriors
    m = 8 * N + 8 * (V + 1) + 4 * B * G + 4 * V * G      # calls (cb, p_wrong), col_ptr, singlet posteriors, addition
    p = 12 * V * G + 4 * V                               # prior + addition in, prob out, v2snp
    return dict(estep=e, mstep=m, pstep=p, iteration=e + m + p)


# ---------------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` with no launcher
# ---------------------------------------------------------------------------------------------------------
def phase(name):
    """Where this rank is, for the watchdog of self_launch (one small file per rank; a no-op outside a self-launched run)."""
    directory = os.environ.get('DEMUXALOT_BENCH_STATUS_DIR')
    if not directory:
        return
    try:
        with open(os.path.join(directory, f'rank{os.environ.get("RANK", "0")}'), 'w') as f:
            f.write(f'{name} (since {time.strftime("%H:%M:%S")})')
    except OSError:
        pass
    # test hook (tests/test_bench_watchdog_cpu.py): this rank stops here for good, as a rank hung in a collective would
    if os.environ.get('DEMUXALOT_BENCH_STALL') == f'{os.environ.get("RANK", "0")}:{name}':
        while True:
            time.sleep(3600)


def self_launch(args):
    """The parent of an N-rank run: it never touches the GPU (no HIP call, no library load) - it starts N fresh rank
    processes of this script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relays rank 0's one JSON line, and
    exits non-zero when any rank does.  WATCHDOG: the ranks report their phase in small files; when the deadline
    (--deadline seconds for the whole run) passes, or a rank has died and the others do not follow within 30 s (they sit in a
    rendezvous or a collective that will never complete), the parent kills the rank processes it started - those PIDs, nothing
    else -, prints every rank's last phase and exits non-zero: the first real multi-rank RCCL run that hangs costs its
    deadline, not the driver's."""
    import shutil
    import socket
    import subprocess
    import tempfile
    with socket.socket() as s:  # a free port NUMBER: the control plane's rendezvous is a file keyed by it
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    status_dir = tempfile.mkdtemp(prefix='demuxalot_bench_status_')
    base = dict(os.environ, WORLD_SIZE=str(args.gpus), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                TORCHELASTIC_RUN_ID=f'bench{os.getpid()}', HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'),
                DEMUXALOT_BENCH_STATUS_DIR=status_dir)
    procs = []
    out_path = os.path.join(status_dir, 'rank0.stdout')
    with open(out_path, 'wb') as rank0_out:
        for rank in range(args.gpus):
            env = dict(base, RANK=str(rank), LOCAL_RANK=str(rank))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, cwd=ROOT,
                                          stdout=rank0_out if rank == 0 else subprocess.DEVNULL))
    deadline = time.monotonic() + args.deadline
    first_death, verdict = None, None
    while True:
        codes = [pr.poll() for pr in procs]
        if all(code is not None for code in codes):
            break
        now = time.monotonic()
        if any(code not in (None, 0) for code in codes) and first_death is None:
            first_death = now
        if now > deadline:
            verdict = f'deadline of {args.deadline:.0f} s passed'
        elif first_death is not None and now > first_death + 30:
            verdict = 'a rank failed and the others did not finish within 30 s'
        if verdict:
            break
        time.sleep(0.2)
    if verdict:
        for pr in procs:  # exactly the processes started above
            if pr.poll() is None:
                pr.kill()
        codes = [pr.wait() for pr in procs]

    def last_phase(rank):
        try:
            return open(os.path.join(status_dir, f'rank{rank}')).read()
        except OSError:
            return 'no phase reported (died before main, or never started)'
    sys.stdout.write(open(out_path, 'rb').read().decode(errors='replace'))
    sys.stdout.flush()
    failed = verdict is not None or any(codes)
    if failed:
        print(f'[bench] {verdict or "a rank exited non-zero"}; exit codes {codes}', file=sys.stderr)
        for rank in range(args.gpus):
            print(f'[bench]   rank {rank}: exit code {codes[rank]}, last phase: {last_phase(rank)}', file=sys.stderr)
        sys.stderr.flush()
    shutil.rmtree(status_dir, ignore_errors=True)
    return 1 if failed else 0


# ---------------------------------------------------------------------------------------------------------
# the experiment: generated once per job
# ---------------------------------------------------------------------------------------------------------
_FIELDS = ('v2snp', 'raw_betas', 'variant_id', 'compressed_cb', 'p_base_wrong', 'truth')


def save_problem(directory, problem):
    os.makedirs(directory, exist_ok=True)
    for name in _FIELDS:
        np.save(os.path.join(directory, name + '.npy'), getattr(problem, name))
    with open(os.path.join(directory, 'shape.json.tmp'), 'w') as f:
        json.dump([problem.n_barcodes, problem.n_snps, problem.n_genotypes], f)
    os.replace(os.path.join(directory, 'shape.json.tmp'), os.path.join(directory, 'shape.json'))  # last: marks the set complete


def load_problem(directory):
    from demuxalot_amd import synth
    B, S, G = json.load(open(os.path.join(directory, 'shape.json')))
    arrays = {name: np.load(os.path.join(directory, name + '.npy'), mmap_mode='r') for name in _FIELDS}
    return synth.SyntheticProblem(B, S, G, np.asarray(arrays['v2snp']), np.asarray(arrays['raw_betas']), arrays['variant_id'],
                                  arrays['compressed_cb'], arrays['p_base_wrong'], np.asarray(arrays['truth']))


def shared_directory():
    """Where rank 0 leaves the generated experiment for the other ranks: memory-backed, named after the job."""
    root = '/dev/shm' if os.path.isdir('/dev/shm') and os.access('/dev/shm', os.W_OK) else '/tmp'
    job = f'{os.environ.get("MASTER_PORT", "0")}_{os.environ.get("TORCHELASTIC_RUN_ID", "none")}'
    return os.path.join(root, 'demuxalot_bench_' + ''.join(ch if ch.isalnum() else '_' for ch in job))


def get_problem(args, rank, world, plane):
    """(problem, seconds, where it came from).  DEMUXALOT_BENCH_PROBLEM=<dir>: a set saved by an earlier run (the profiler
    child runs of live_counters); otherwise rank 0 generates, and with several ranks shares it through /dev/shm."""
    from demuxalot_amd import synth
    B, S, G, dp, seed = WORKLOADS[args.workload]
    t0 = time.perf_counter()
    cached = os.environ.get('DEMUXALOT_BENCH_PROBLEM', '')
    if cached and os.path.exists(os.path.join(cached, 'shape.json')):
        problem = load_problem(cached)
        assert (problem.n_barcodes, problem.n_snps, problem.n_genotypes) == (B, S, G), 'cached problem of another workload'
        return problem, time.perf_counter() - t0, 'cache'
    def generate():
        if B >= SHARDED_GENERATOR_FROM:
            return synth.generate_sharded(B, S, G, n_shards=16, doublets=dp > 0, seed=seed)
        return synth.generate(B, S, G, doublets=dp > 0, seed=seed)
    if world == 1:
        return generate(), time.perf_counter() - t0, 'generated'
    directory = shared_directory()
    problem = None
    if rank == 0:
        problem = generate()
        save_problem(directory, problem)
    plane.barrier()
    if rank != 0:
        problem = load_problem(directory)
    return problem, time.perf_counter() - t0, 'generated on rank 0, shared through ' + directory


def shard_of(problem, rank, world):
    """This rank's barcode range of the experiment (equal numbers of calls), barcode indices re-based."""
    from demuxalot_amd import synth
    from demuxalot_amd.distributed import partition_barcodes
    bounds = partition_barcodes(np.bincount(problem.compressed_cb, minlength=problem.n_barcodes), world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    v, cb, e = problem.subset_barcodes(lo, hi)
    return synth.SyntheticProblem(hi - lo, problem.n_snps, problem.n_genotypes, problem.v2snp, problem.raw_betas, v, cb, e,
                                  problem.truth[lo:hi])


# ---------------------------------------------------------------------------------------------------------
# CPU baseline
# ---------------------------------------------------------------------------------------------------------
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
    n_options = problem.n_genotypes * (problem.n_genotypes + 1) // 2 if doublet_prior > 0 else problem.n_genotypes
    n_probe = min(problem.n_barcodes, max(50, min(1000, 64_000 // n_options)))  # (the oracle makes one pass per option)
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
                sample=f'first {n_sample} barcodes ({n_calls} calls), one EM iteration, {dt:.1f} s',
                sample_note='the first barcodes of the workload with the full V and G; numpy single-threaded like the reference',
                host_cores=os.cpu_count()), logits, post, n_sample


# ---------------------------------------------------------------------------------------------------------
# roofline of the dominant kernel
# ---------------------------------------------------------------------------------------------------------
def live_counters(args, problem_dir):
    """Measured now, by child runs of this script under `rocprofv3 --kernel-trace --pmc` (one counter per pass, as
    MI355X_MICROARCH.md prescribes), for the E-step kernel the timed iterations spend most time in: its name and average
    duration under the tracer, HBM-side bytes per launch (FETCH_SIZE + WRITE_SIZE, both reported in KiB; the row gathers
    are 4-byte-per-lane reads, not the 16-byte streams with the documented 2x under-count), the clock it sustains
    (GRBM_GUI_ACTIVE summed over the 8 XCDs / 8 / its duration in the same pass) and how busy the VALUs are
    (SQ_ACTIVE_INST_VALU, in quad-cycles, x 4 / 1024 SIMDs / the cycles of the launch).  The children load the experiment
    this run saved instead of generating it again.  {'skipped': why} when the profiler is not available."""
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
    # The child makes the timed call and nothing around it (--timed-only: no clock warm-up spins, no other region): install, the first
    # E-step (dictionary form, ONE launch), 2 + 12 + 12 iterations - so the k_estep_* kernel with the largest total IS the timed
    # iterations' (round 5 picked the warm-up's k_estep_dictq: 56 spin E-steps ahead of every region)
    child = [sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', args.workload, '--steps', '12', '--warmup', '2', '--timed-only']
    child += ['--flat-genotypes'] if args.flat_genotypes else []
    env = dict(os.environ, TMPDIR='/tmp', DEMUXALOT_BENCH_PROBLEM=problem_dir)
    found = {}
    try:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE', 'GRBM_GUI_ACTIVE', 'SQ_ACTIVE_INST_VALU'):
            where = os.path.join(out_dir, counter)
            try:
                subprocess.run(['rocprofv3', '--kernel-trace', '--pmc', counter, '--output-format', 'csv', '-d', where, '--'] + child,
                               env=env, cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=240, check=True)
            except (OSError, subprocess.SubprocessError) as exc:
                found[f'{counter}_failed'] = (getattr(exc, 'stderr', b'') or b'').decode(errors='replace')[-300:] or str(exc)
                continue
            spans = {}  # kernel name -> durations; the dominant E-step kernel = largest total among the k_estep_* ones
            for path in glob.glob(os.path.join(where, '**', '*kernel_trace.csv'), recursive=True):
                for row in csv.DictReader(open(path)):
                    if 'k_estep' in row['Kernel_Name']:
                        spans.setdefault(row['Kernel_Name'], []).append(float(row['End_Timestamp']) - float(row['Start_Timestamp']))
            if not spans:
                continue
            kernel = max(spans, key=lambda k: sum(spans[k]))
            found.setdefault('kernel', kernel.split('(')[0].replace('void dmx::', ''))
            found.setdefault('kernel_launches_in_the_child', len(spans[kernel]))
            values, dispatches = [], set()
            for path in glob.glob(os.path.join(where, '**', '*counter_collection.csv'), recursive=True):
                for row in csv.DictReader(open(path)):
                    if row['Kernel_Name'] == kernel and row['Counter_Name'] == counter:
                        values.append(float(row['Counter_Value']))
                        dispatches.add(row['Dispatch_Id'])
            if not values:
                continue
            found[counter] = sum(values) / len(dispatches)
            ns = sum(spans[kernel]) / len(spans[kernel])
            if counter == 'GRBM_GUI_ACTIVE':
                found['kernel_ns_under_tracer'] = ns
                found['cycles_per_launch'] = found[counter] / N_XCD
                found['clock_ghz'] = found['cycles_per_launch'] / ns
            if counter == 'SQ_ACTIVE_INST_VALU':
                found['valu_ns'] = ns
        if 'FETCH_SIZE' in found and 'WRITE_SIZE' in found:
            found['traffic'] = 1024.0 * (found['FETCH_SIZE'] + found['WRITE_SIZE'])
        if 'SQ_ACTIVE_INST_VALU' in found and 'clock_ghz' in found:
            found['valu_busy'] = found['SQ_ACTIVE_INST_VALU'] * 4.0 / N_SIMD / (found['clock_ghz'] * found['valu_ns'])
        return found
    except (OSError, KeyError, ValueError) as exc:
        found['error'] = repr(exc)
        return found
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)


def roofline(ab, e_ms, N, G, K, live, coarse_share=0.0, passes=None, ms_per_step=None):
    """The contract's HBM figures for the E-step of the timed iterations on ALGORITHMIC bytes, the ops roofline of
    SURVEY.md 8d (log terms against the v_log_f32 issue peak), and what the kernel actually runs on: the genotype-row
    gather out of L2 (guarded / tolerance mode) or VALU issue (exact mode) - `valu.busy` from the live counters."""
    achieved = ab['estep'] / (e_ms * 1e-3) / 1e9
    terms_per_s = N * K / (e_ms * 1e-3)
    delivered = (N * (4 - 2 * coarse_share) * G) / (e_ms * 1e-3) / 1e9   # (the coarse pass gathers binary16 rows)
    out = {
        'bound': 'hbm', 'kernel': live.get('kernel'), 'achieved': achieved, 'peak': 8000.0, 'unit': 'GB/s', 'frac': achieved / 8000.0,
        'traffic': live.get('traffic'),
        'traffic_source': 'rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE per launch, child runs of this command (live)' if 'traffic' in live else None,
        'algorithmic_bytes_per_launch': ab['estep'], 'estep_ms': e_ms,
        'estep_ms_source': 'HIP events on the library\'s stream around the E-step phase (dmx_set_phase_timers), average over the `steps` E-steps of the timed call '
                           'made once more behind the timed window with the events on (ms_per_step_with_phase_timers); the timed window itself runs without them, as a default call does',
        'ops': {'terms_per_s': terms_per_s, 'peak': LOG_ISSUE_PEAK, 'frac': terms_per_s / LOG_ISSUE_PEAK,
                'note': 'SURVEY.md 8d: N x K log terms per E-step against the v_log_f32 issue peak (19.7 T/s); the guarded / '
                        'tolerance mode takes one hardware log2 per 8 terms, the exact mode repeats numpy\'s float32 log (16 VALU '
                        'instructions per term)'},
        'l2_gather': {'delivered_GBps': delivered, 'all_hit_rate_GBps': L2_GATHER_ALL_HIT_GBPS, 'frac': delivered / L2_GATHER_ALL_HIT_GBPS,
                      'coarse_pass_share_of_the_esteps': coarse_share,
                      'note': 'N x 4G bytes (coarse pass: N x 2G) of genotype rows per E-step, gathered out of L2 / Infinity Cache (the table fits no L2); '
                              'the all-hit rate is the same loop on an L2-resident table (DESIGN.md 4.1)'},
        'valu': {k: live[k] for k in ('valu_busy', 'clock_ghz', 'cycles_per_launch', 'SQ_ACTIVE_INST_VALU', 'kernel_ns_under_tracer') if k in live},
        'note': 'frac = algorithmic bytes / E-step time / 8 TB/s as the contract defines it; the E-step re-reads a 256-byte table '
                'row per call, so it is bounded by the L2 gather rate (l2_gather.frac), not by HBM: DESIGN.md 4',
    }
    out['kernel_ns_under_tracer'] = live.get('kernel_ns_under_tracer')
    out['valu_busy'] = live.get('valu_busy')
    if ms_per_step:
        out['iteration_frac'] = ab['iteration'] / (ms_per_step * 1e-3) / 8e12   # all three phases' algorithmic bytes / the whole step / 8 TB/s
    # the kernel the counters describe must be the one the timed E-steps ran (estep_passes of the timed region)
    passes = passes or {}
    # (only the coarse pass has ONE kernel name; the fine pass is k_estep_tiled, k_estep_direct<.., true>, k_estep_block or k_estep_pairblocks by shape)
    expected = 'k_estep_tiled_coarse' if 2 * passes.get('coarse', 0) > passes.get('of', 1) else None
    out['kernel_expected'] = expected
    if expected and live.get('kernel'):
        out['kernel_is_the_timed_one'] = expected in live['kernel']
        if not out['kernel_is_the_timed_one']:
            print(f'[bench] live counters describe {live["kernel"]}, the timed E-steps ran {expected}*: traffic / valu dropped from the line', file=sys.stderr)
            out['traffic'] = out['kernel_ns_under_tracer'] = out['valu_busy'] = None
            out['live_counters'] = {'error': f'counters of {live["kernel"]}, timed kernel {expected}'}
    elif expected and not out['kernel']:
        out['kernel'] = expected + ' (by estep_passes; no live trace)'
    elif live.get('kernel') and passes.get('coarse', 0) > 0 and 'coarse' not in live['kernel'] and 2 * passes.get('coarse', 0) <= passes.get('of', 1):
        pass  # (a minority of coarse E-steps: the dominant kernel is the other level's)
    problems = {k: v for k, v in live.items() if k in ('skipped', 'error') or k.endswith('_failed')}
    if problems:
        out['live_counters'] = problems
    return out


def e2e_timing(problem, doublet_prior, n_iterations=5):
    """Wall time of the drop-in calls themselves (demux.py:35-66, 120-156: the reference's containers in, DataFrames out)
    on this workload, and where it goes.  SURVEY.md 8d asks for barcodes/s with and without the D2H of the [B, K] results:
    `on_device=True` keeps them on the GPU behind a DevicePosteriors (assignments computed there)."""
    import pandas as pd
    from demuxalot_amd import Demultiplexer, synth
    from demuxalot_amd.demux import _option_names, _pack_on_device
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

    def forget():  # what the front-end keeps between calls on the same inputs (demux.py: _pack_on_device): as for inputs never seen
        from demuxalot_amd.device import get_context
        get_context()._resident_key = None
        genotypes._amd_variant_keys = None
        handler._amd_index = None

    forget()
    t_predict, (logits_df, probs_df) = timed(lambda: Demultiplexer.predict_posteriors(calls, genotypes, handler, doublet_prior=doublet_prior))
    t_predict_again, _ = timed(lambda: Demultiplexer.predict_posteriors(calls, genotypes, handler, doublet_prior=doublet_prior))
    t_learn_after, _ = timed(lambda: Demultiplexer.learn_genotypes(calls, genotypes, handler, n_iterations=n_iterations, doublet_prior=0.))
    forget()
    t_predict_dev, dev = timed(lambda: Demultiplexer.predict_posteriors(calls, genotypes, handler, doublet_prior=doublet_prior, on_device=True))
    t_assign_dev, assigned = timed(lambda: dev.assignments(0.9))
    t_assign_host, assigned_host = timed(lambda: probs_df[probs_df.max(axis=1).gt(0.9)].idxmax(axis=1))
    same = bool(assigned.index.equals(assigned_host.index) and (assigned.values == assigned_host.values).all())
    dev.close()
    forget()
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
        # the EM call exactly as learn_genotypes makes it (demuxalot_amd/demux.py): no logits wanted, posteriors and addition left on the
        # device (the learnt table is formed there: dmx_get_learnt_betas); round 5 timed it with the 51 MB addition downloaded
        ctx.set_logits_needed(False)
        t_em, _ = timed(lambda: (ctx.em(n_iterations, 0.01, pen0, with_doublets=False, fetch_logits=False, fetch_probs=False, fetch_addition=False), ctx.synchronize()))
        ctx.set_logits_needed(True)
    finally:
        release_private_context(ctx)
    return {
        'workload_molecule_calls': n_molecule_calls, 'barcodes': B,
        'predict_posteriors_s': t_predict, 'predict_barcodes_per_s': B / t_predict,
        'predict_posteriors_same_inputs_again_s': t_predict_again, f'learn_genotypes_{n_iterations}it_after_predict_s': t_learn_after,
        'predict_posteriors_on_device_s': t_predict_dev, 'predict_on_device_barcodes_per_s': B / t_predict_dev,
        'assignments_on_device_s': t_assign_dev, 'assignments_with_pandas_s': t_assign_host, 'assignments_identical': same,
        f'learn_genotypes_{n_iterations}it_s': t_learn, f'learn_genotypes_{n_iterations}it_on_device_s': t_learn_dev,
        'split_s': {'flatten_upload_device_pack_prior': t_pack, 'pstep_estep': t_pe, 'd2h_logits_and_posteriors': t_d2h,
                    'two_dataframes': t_frames, f'em_{n_iterations}_iterations_fused': t_em},
        'build_objects_s (synthetic generator, not part of a call)': t_objects,
        'note': 'wall time of the Python entry points on the containers of the whole workload (one molecule per call); the reference '
                'spends ~3.5-4 us per molecule call in pack_calls alone (SURVEY.md 8a9).  predict_posteriors_s / learn_genotypes_*_s: inputs '
                'the front-end has not seen (warm context); *_again / *_after_predict: the same containers and genotypes again - the packed '
                'problem is still resident, the key arrays of var2varid are kept on the genotypes object',
    }


def hard_workload(args, ctx, whole, betas, pen, dp):
    """The default mode's worst cases next to the exact mode on the SAME shape (include/demux_hip.h: dmx_set_guard_adaptive).
    `siblings_50_calls`: 50 calls per barcode instead of 400 and donors in sibling pairs (half of the SNPs identical), so
    that a fifth of the barcodes keep some posterior between 0.03 and 0.97.  `identical_twins`: the workload's own calls
    on a genotype table whose donors 2j and 2j + 1 are identical - the two best logits of EVERY barcode tie, the guard can
    prove nothing, the fast pass is wasted on all of them: the worst case there is.  Three timed regions each: the default
    mode (adaptive: E-steps for which fast pass + redo costs more than the exact kernel run it on every barcode), the
    same with the adaptation off (fast pass + exact redo of the queued barcodes every time), and the exact mode."""
    from demuxalot_amd import synth
    B, S, G, _dp, seed = WORKLOADS[args.workload]
    t0 = time.perf_counter()
    siblings = synth.generate(B, S, G, calls_per_barcode=50, doublets=dp > 0, seed=seed + 5000, sibling_pairs=True)
    t_gen = time.perf_counter() - t0
    twins = betas.copy()
    twins[:, 1:2 * (G // 2):2] = twins[:, 0:2 * (G // 2):2]
    out = {}
    for variant, problem, table in (('siblings_50_calls', siblings, siblings.prior_betas(add_data_prior=False)), ('identical_twins', whole, twins)):
        ctx.set_problem(problem.n_barcodes, problem.n_variants, G, problem.variant_id, problem.compressed_cb, problem.p_base_wrong, problem.v2snp)
        ctx.set_betas(table)
        res = {'calls': problem.n_calls}
        for name, mode, adaptive in (('default', None, True), ('default_without_adaptation', None, False), ('exact', 'exact', True)):
            ctx.apply_environment()
            if mode:
                ctx.set_estep_mode(mode)
                ctx.set_exact_additions(True)
            ctx.set_guard_adaptive(adaptive)
            ctx.set_addition(None)
            ctx.probs_from_betas(0.01, fetch=False)
            ctx.set_estep_dictionary('never')  # (the first pass of an EM run takes the dictionary form: exact in every mode)
            ctx.estep(pen, with_doublets=dp > 0, fetch_logits=False, fetch_probs=False)
            first_queued = ctx.guard_state()[2]
            if args.mstep == 'auto':
                ctx.set_msteps_expected(args.warmup + args.steps)
            region = timed_region(ctx, None, args.steps, args.warmup, spin=lambda: ctx.estep(pen, with_doublets=dp > 0, fetch_logits=False, fetch_probs=False))
            _direct, direct_steps, would, fast_ms, exact_ms = ctx.guard_state()
            res[name] = {'ms_per_step': region['ms_per_step'], 'kernel_ms': region['kernel_ms'], 'guard': region['guard'], 'estep_passes': region['estep_passes'], 'mstep_passes': region['mstep_passes'],
                         'first_estep_queued_fraction': first_queued / problem.n_barcodes, 'esteps_run_direct': direct_steps, 'steps': args.steps,
                         'last_estep_would_queue_fraction': would / problem.n_barcodes,
                         'device_timed_ms': {'fast_pass_all_barcodes': fast_ms, 'exact_kernel_all_barcodes': abs(exact_ms),
                                             'exact_kernel_measured_by_a_direct_estep': exact_ms > 0}}
        res['default_over_exact'] = res['default']['ms_per_step'] / res['exact']['ms_per_step']
        res['estep_default_over_exact'] = res['default']['kernel_ms']['estep'] / res['exact']['kernel_ms']['estep']
        out[variant] = res
    out['siblings_50_calls'].update(calls_per_barcode=50, sibling_pairs=True, generated_s=t_gen)
    ctx.set_estep_dictionary('auto')
    ctx.set_guard_adaptive(True)
    ctx.apply_environment()
    out['note'] = ('guarded E-step = fast pass F + exact redo of the queued fraction f = F + f E against the exact kernel\'s E; both passes are timed '
                   'on the device and E-steps for which F + f E > E run the exact kernel on every barcode (bit-identical to the reference '
                   'there): include/demux_hip.h dmx_set_guard_adaptive')
    return out


def _lib_device_count():
    from demuxalot_amd import _lib
    return _lib.device_count()


CLOCK_WARMUP_MS = 60.0


def clock_warmup(ctx, spin, ms=CLOCK_WARMUP_MS):
    """Device work for `ms` milliseconds that leaves the EM where it is (`spin` enqueues the E-step on the current table once more:
    the same posteriors again).  The device lowers its clocks within 10 ms of idling and takes about 30 ms of work to raise them
    again (scripts/phase_timer_cost.py: E-step 1.03 -> 0.90 ms, M-step 0.37 -> 0.33 ms over the first 30 ms after an idle): behind
    the host-side pauses between this script's regions, W = 5 warm-up iterations (6 ms) end inside that ramp, and a timed region
    of 23 ms would report the ramp instead of the rate the device sustains.  The figure after an idle is measured too
    (`after_idle` of the line)."""
    if spin is None or ms <= 0:
        return 0
    t0, n = time.perf_counter(), 0
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(8):
            spin()
        ctx.synchronize()
        n += 8
    return n


def timed_region(ctx, plane, steps, warmup, spin=None, idle_s=0.0, snapshot=False):
    """(`spin`: clock_warmup ahead of the warm-up iterations - NOT used by the headline region, which is the contract's W warm-up
    iterations and nothing else; `idle_s`: the device left idle that long ahead of them instead; `snapshot`: the end state of the
    region - best option and its posterior of every barcode, all singlet posteriors - fetched behind both windows for parity_timed.)
    W untimed iterations, then exactly K timed ones bracketed by barrier + device synchronisation on both sides;
    the MAX over ranks of the wall time.  The timed call runs as a default call does, without the phase timers (an event record
    is a barrier packet of its own: 6 us at each of the iteration's four phase boundaries); kernel_ms comes from the SAME call made
    once more behind the timed window with the timers on, and its wall time is reported beside (ms_per_step_with_phase_timers)."""
    def barrier():
        if plane is not None:
            plane.barrier()
    n_spin = 0
    if idle_s > 0:
        ctx.synchronize()
        time.sleep(idle_s)
    else:
        n_spin = clock_warmup(ctx, spin)
    ctx.run_iterations(warmup, 0.01)
    ctx.synchronize()
    ctx.reset_timings()
    ctx.set_phase_timers(False)
    barrier()
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.run_iterations(steps, 0.01)
    ctx.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if plane is not None:
        elapsed = plane.max_float64(elapsed)
    stats = (ctx.guard_stats(), ctx.guard_levels(), ctx.mstep_incremental())
    ctx.set_phase_timers(True)
    ctx.reset_timings()
    barrier()
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.run_iterations(steps, 0.01)
    ctx.synchronize()
    barrier()
    elapsed_timers = time.perf_counter() - t0
    if plane is not None:
        elapsed_timers = plane.max_float64(elapsed_timers)
    timers = ctx.timings()
    ctx.set_phase_timers(False)
    end_state = None
    if snapshot:
        end_state = dict(best=ctx.get_assignments()[0], probs=ctx.get_block('probs', 0, ctx.B, 0, min(ctx.K, 128)), iterations=warmup + 2 * steps)
    (_redone, redone_total, rows), levels, (m_full, m_delta, m_changed) = stats
    return {'elapsed': elapsed, 'end_state': end_state, 'ms_per_step_with_phase_timers': 1e3 * elapsed_timers / steps, 'clock_warmup_esteps': n_spin,
            'mstep_passes': ({'incremental_mstep': True, 'full': m_full, 'delta': m_delta, 'of': steps, 'barcodes_changed_in_the_last_mstep': m_changed}
                             if m_full + m_delta > 0 else {'incremental_mstep': False, 'note': 'every M-step recomputes every sum'}), 'ms_per_step': 1e3 * elapsed / steps, 'em_iterations_per_s': steps / elapsed,
            'estep_passes': {'coarse': levels['coarse_steps'], 'of': steps, 'last': {0: 'coarse', 1: 'fine', 2: 'direct'}.get(levels['level'], 'not guarded'),
                             'device_timed_ms': {'coarse_pass': levels['coarse_pass_ms'], 'fine_pass': levels['fine_pass_ms'], 'exact_kernel': abs(levels['exact_pass_ms'])},
                             'last_estep_flagged': {'fine_guard': levels['flagged_fine'], 'coarse_guard': levels['flagged_coarse']}},
            'kernel_ms': {k: (v['ms'] / max(1, v['launches'])) for k, v in timers.items()},
            'exchange_ms_per_step': timers['allreduce']['ms'] / max(1, steps),
            'guard': {'barcodes_redone_exactly': redone_total, 'barcode_rows': rows, 'fraction': redone_total / max(1, rows)}}

# ---------------------------------------------------------------------------------------------------------
# the ONE line on stdout: compact, bounded; everything else goes to a sidecar file
# ---------------------------------------------------------------------------------------------------------
LINE_LIMIT = 4000   # bytes; the driver keeps the last 8 000 characters of stdout (round 5's 20 kB line was cut and never parsed)
LINE_KEYS = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data',
             'config', 'roofline', 'cpu_baseline')


def _sig(x, digits=6):
    """Floats to `digits` significant figures (the line is for reading and parsing, the sidecar keeps everything)."""
    if isinstance(x, float):
        return float(f'{x:.{digits}g}')
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def compact_line(full, details_path=None):
    """The contract's JSON line out of the full result: the contract's keys, `roofline` and `cpu_baseline` as the task statement
    defines them, the three scalars the round-5 verdict asks for beside them and the parity checks of this very run.  Regions, hard
    workloads, end-to-end splits and notes stay in the sidecar (`details`).  Raises when the line would not fit LINE_LIMIT: a line
    the driver cannot parse is a round without a measurement."""
    roof, base = full.get('roofline') or {}, full.get('cpu_baseline')
    cfg = full['config']
    line = {
        'metric': full['metric'], 'value': full['value'], 'unit': full['unit'], 'n_gpus': full['n_gpus'], 'steps': full['steps'], 'warmup': full['warmup'],
        'ms_per_step': full['ms_per_step'], 'higher_is_better': True, 'scaling': full['scaling'], 'vs_baseline': None,
        'dtype': full['dtype'], 'data': full['data'],
        'config': {k: cfg[k] for k in ('workload', 'barcodes_total', 'barcodes_per_gpu', 'snps', 'variants', 'genotypes', 'options', 'calls_per_gpu',
                                       'doublet_prior', 'estep_mode', 'mstep_form', 'parallelism') if k in cfg},
        'em_iterations_per_s': full.get('em_iterations_per_s'),
        'kernel_ms': full.get('kernel_ms'),
        'estep_passes': {k: v for k, v in (full.get('estep_passes') or {}).items() if k in ('coarse', 'of', 'last')},
        'guard_redone_fraction': (full.get('guard') or {}).get('fraction'),
        'roofline': {k: roof.get(k) for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'algorithmic_bytes_per_launch', 'estep_ms',
                                              'kernel_ns_under_tracer', 'valu_busy', 'iteration_frac')},
        'cpu_baseline': ({k: base.get(k) for k in ('value', 'unit', 'cores', 'host_cores', 'kind', 'sample', 'speedup_vs_gpu_value')} if base else None),
        'exact_mode_ms_per_step': (full.get('exact_mode') or {}).get('ms_per_step'),
        'after_idle_ms_per_step': (full.get('after_idle') or {}).get('ms_per_step'),
        'clocks_warm_ms_per_step': (full.get('clocks_warm') or {}).get('ms_per_step'),
        'default_call_5it_ms_per_iteration': full.get('default_call_5it_ms_per_iteration'),
        'cold_start_5it_ms_per_iteration': (full.get('cold_start_5it') or {}).get('ms_per_step'),
        'predict_barcodes_per_s': full.get('predict_barcodes_per_s'),
        'device_bytes_per_call': (full.get('device_bytes') or {}).get('per_call_without_the_results'),
        'lean_memory': ({k: full['lean_memory'].get(k) for k in ('device_bytes_per_call', 'ms_per_step')} if full.get('lean_memory') else None),
        'parity_timed': full.get('parity_timed'),
        'parity_on_sample': full.get('parity_on_sample'),
    }
    if cfg.get('rccl_fallback'):
        line['config']['rccl_fallback'] = str(cfg['rccl_fallback'])[:120]
    if full.get('weak'):
        line['weak'] = {k: full['weak'].get(k) for k in ('value', 'ms_per_step', 'barcodes_total', 'exchange_ms_per_step')}
    if full.get('exchange_ms_per_step'):
        line['exchange_ms_per_step'] = full['exchange_ms_per_step']
    if 'live_counters' in roof:
        line['roofline']['live_counters'] = {k: str(v)[:80] for k, v in roof['live_counters'].items()}
    if details_path:
        line['details'] = details_path
    line = {k: v for k, v in _sig(line).items() if v is not None or k in LINE_KEYS}
    text = json.dumps(line, separators=(', ', ': '))
    missing = [k for k in LINE_KEYS if k not in line]
    if missing or len(text) >= LINE_LIMIT:
        raise AssertionError(f'bench line unusable: {len(text)} bytes (limit {LINE_LIMIT}), missing keys {missing}')
    return text


def write_details(full, workload, world):
    """The full result (every region, the hard workloads, the end-to-end split, the notes) as a sidecar file: gpurun_out/ of the
    repository when it can be written (it travels back from the GPU box), the temporary directory otherwise."""
    name = f'bench_details_{workload}_n{world}.json'
    for directory in (os.path.join(ROOT, 'gpurun_out'), os.environ.get('TMPDIR', '/tmp')):
        try:
            os.makedirs(directory, exist_ok=True)
            path = os.path.join(directory, name)
            with open(path, 'w') as f:
                json.dump(full, f, indent=1)
            return os.path.relpath(path, ROOT) if path.startswith(ROOT) else path
        except OSError:
            continue
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default='em_200k_100k_64', choices=sorted(WORKLOADS))
    ap.add_argument('--reduce-dtype', default='auto', choices=['auto', 'f64', 'f32'],
                    help='wire format of the reduce-scatter of the beta additions: float64 partial sums (results independent of the '
                         'number of ranks up to float32 rounding ties) or float32 (half the bytes; posteriors stay within the 1e-5 '
                         'contract: tests/test_gpu_ranks_on_one_gpu.py).  auto = f32 (two GPUs share ONE link, so the bytes matter most '
                         'there: DESIGN.md 5); f64 when DEMUXALOT_AMD_ESTEP=exact.  The variant-sharded M-step exchanges no sums at all')
    ap.add_argument('--scaling', default='both', choices=['both', 'weak', 'strong'],
                    help='N > 1.  strong: the workload in total, barcodes sharded over the GPUs (BASELINE.json configs[3] as written); '
                         'weak: the workload per GPU; both (default): strong is the headline value, weak a sub-object of the line')
    ap.add_argument('--mstep', default='auto', choices=['auto', 'tiles', 'items'],
                    help='M-step form of the timed regions.  auto (default): the LIBRARY decides, as it does for a learn_genotypes call - it is '
                         'told how many iterations this run makes (warm-up + steps: dmx_set_msteps_expected, what a front-end looping over EM '
                         'iterations knows) and builds the tile-major records at the first M-step when 8 or more are to come; the build then '
                         'falls into the warm-up and the line reports it (mstep_records_build_ms, ms_per_step_incl_record_build).  tiles / '
                         'items: the form forced')
    ap.add_argument('--incremental-mstep', action='store_true',
                    help='the headline regions with the library\'s default M-step (incremental: dmx_set_mstep_incremental(1)) instead of a full pass per iteration')
    ap.add_argument('--deadline', type=float, default=1500.0,
                    help='N > 1 without a launcher: seconds after which the parent kills the rank processes it started, reports every '
                         'rank\'s last phase and exits non-zero (a rank hung in a collective must not cost the caller its own limit)')
    ap.add_argument('--timed-only', action='store_true', help='the timed region of the default mode only (profiler child runs)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-exact-mode', action='store_true', help='skip the second timed region (bit-exact E-step and additions)')
    ap.add_argument('--fast-mode', action='store_true', help='a third timed region: the tolerance-mode E-step without the guard')
    ap.add_argument('--no-fast-mode', action='store_true', help=argparse.SUPPRESS)  # older command lines (the region is opt-in now)
    ap.add_argument('--no-live-traffic', action='store_true', help='no rocprofv3 --pmc child runs (roofline.traffic / valu stay empty)')
    ap.add_argument('--no-e2e', action='store_true', help='skip the end-to-end timing of the drop-in calls (containers in, DataFrames out)')
    ap.add_argument('--no-hard-workload', action='store_true', help='skip the default mode\'s worst case (few calls per barcode, sibling donors) next to the exact mode')
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
    if args.timed_only:
        args.no_cpu_baseline = args.no_exact_mode = args.no_live_traffic = args.no_e2e = args.no_hard_workload = True

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args))

    # gloo and RCCL print banners on stdout; stdout must carry the one JSON line only, so fd 1 is pointed
    # at stderr for the whole run and the line is written to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    phase('start')
    if args.reduce_dtype == 'auto':
        args.reduce_dtype = 'f64' if os.environ.get('DEMUXALOT_AMD_ESTEP', '') == 'exact' else 'f32'
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'

    # Control plane (rendezvous, RCCL unique id, barrier, max-reduce of the wall time): plain sockets
    # (demuxalot_amd/plane.py).  A launcher only starts the processes: the workers never import torch, whose
    # ROCm wheels carry their own libamdhip64 / librccl - two HIP runtimes in one process is what the library refuses.
    # The data-plane collectives are RCCL inside libdemux_hip.so (or, with --host-plane, staged through the plane).
    plane = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        from demuxalot_amd.plane import SocketControlPlane
        phase('control plane rendezvous')
        plane = SocketControlPlane(rank, world, os.environ.get('MASTER_ADDR', '127.0.0.1'), host_collectives=args.host_plane)

    from demuxalot_amd import Demultiplexer
    from demuxalot_amd import device as dmx_device
    from demuxalot_amd.device import DeviceContext

    B_workload, S, G, dp, _seed = WORKLOADS[args.workload]
    phase('generating / loading the experiment')
    whole, t_gen, problem_source = get_problem(args, rank, world, plane)
    betas = whole.prior_betas(add_data_prior=False)  # identical on every rank
    if args.flat_genotypes:
        betas = np.ones_like(betas)
    V = whole.n_variants
    pen = Demultiplexer._doublet_penalties(G, dp)
    K = len(pen)

    phase('creating the device context')
    ctx = DeviceContext(local_rank % max(1, _lib_device_count()))
    default_mode = os.environ.get('DEMUXALOT_AMD_ESTEP', '') or dmx_device.DEFAULT_ESTEP_MODE
    runtimes, rccl_fallback = None, None
    if use_dist and args.host_plane:
        ctx.comm_init_host(rank, world, plane.host_collective, reduce_dtype=args.reduce_dtype)
    elif use_dist:
        # RCCL communicator; when creating it fails on some rank (no usable fabric, library missing), every rank learns
        # so over the control plane and the run goes on with the exchange staged through host memory - said in the line
        phase('RCCL communicator (ncclCommInitRank)')
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

    def install(problem):
        t = time.perf_counter()
        ctx.set_problem(problem.n_barcodes, V, G, problem.variant_id, problem.compressed_cb, problem.p_base_wrong, problem.v2snp)
        ctx.set_betas(betas)
        ctx.set_addition(None)
        ctx.probs_from_betas(0.01, fetch=False)
        return time.perf_counter() - t

    # The M-step form is the library's own choice (include/demux_hip.h: dmx_set_mstep_tiles): it builds the tile-major records
    # (a 2.6 ms sort of the calls; an M-step then takes 0.34 instead of 0.70 ms) at the first M-step that has 8 or more still to
    # come.  This run tells it how many it will make - warm-up + steps, as learn_genotypes(n_iterations=...) does through
    # dmx_em - so the build falls into the warm-up; its cost is reported next to the steady-state figure, and the line
    # carries the work-item form's figure too (`work_item_mstep`: what runs of fewer than 8 M-steps take).
    if args.mstep == 'tiles':
        ctx.set_mstep_tiles('always')
    elif args.mstep == 'items':
        ctx.set_mstep_tiles('never')
    # The headline regions recompute EVERY sum in EVERY iteration (dmx_set_mstep_incremental(0)): the library's default - the incremental
    # M-step, which keeps the integer sums of the previous iteration and only adds the differences for the barcodes whose posteriors changed
    # (same bits) - is timed right behind them and reported as `with_incremental_mstep`, not as `value`: on this synthetic experiment the EM
    # has converged by the timed iterations, so that an incremental step has next to nothing left to do.
    ctx.set_mstep_incremental(args.incremental_mstep)
    # The timed call is what learn_genotypes makes: it returns the last iteration's posteriors, no logits (demux.py:65-66), and says so
    # (dmx_set_logits_needed(0)) - the last E-step of the call is then one whose logits nobody reads, like the 19 before it.  The same region
    # with the last E-step's logits kept (what a caller of the C ABI gets by default): `with_logits_of_the_last_estep`.
    ctx.set_logits_needed(False)
    _apply_environment = ctx.apply_environment

    def apply_environment_for_the_bench():  # (the extra regions reset the modes in between: the M-step stays as chosen here)
        _apply_environment()
        ctx.set_mstep_incremental(args.incremental_mstep)
        ctx.set_logits_needed(False)
    ctx.apply_environment = apply_environment_for_the_bench

    def estep_again():  # (clock_warmup: the E-step on the table as it stands - the EM does not move)
        ctx.estep(pen, with_doublets=dp > 0, fetch_logits=False, fetch_probs=False)

    # ---- the timed regions: configs[3] as written first (strong: this rank's barcode range), then the workload per GPU ----
    kinds = ['strong', 'weak'] if args.scaling == 'both' else [args.scaling]
    if world == 1:
        kinds = kinds[:1]
    regions, t_up = {}, 0.0
    logits0 = probs0 = best0 = None
    problem = whole
    for kind in kinds:
        problem = shard_of(whole, rank, world) if (kind == 'strong' and world > 1) else whole
        phase(f'{kind}: installing the problem (set-up collectives)')
        t_up += install(problem)
        keep_first = world == 1 and not args.timed_only
        ctx.estep(pen, with_doublets=dp > 0, fetch_logits=False, fetch_probs=False)  # fixes the options
        if keep_first:  # the first rows of the first pass (never the whole [B, K] matrices: 2 x 33 GB at 1M x 8256)
            n_keep = int(min(problem.n_barcodes, max(64, (256 << 20) // (4 * K))))
            logits0, probs0 = ctx.get_block('logits', 0, n_keep), ctx.get_block('probs', 0, n_keep)
            best0 = ctx.get_assignments()[0]
        if args.mstep == 'auto':
            ctx.set_msteps_expected(args.warmup + args.steps)
        phase(f'{kind}: warm-up + timed region (per-iteration collectives)')
        region = timed_region(ctx, plane, args.steps, args.warmup, snapshot=world == 1 and not args.timed_only)
        built, build_ms = ctx.mstep_tiles_info()
        region['mstep_records_build_ms'] = build_ms if built else 0.0
        region['device_bytes'] = ctx.device_bytes()
        barcodes_total = B_workload if (kind == 'strong' or world == 1) else B_workload * world
        region.update(value=barcodes_total * args.steps / region['elapsed'], barcodes_total=barcodes_total,
                      barcodes_per_gpu=problem.n_barcodes, calls_per_gpu=problem.n_calls, exchange=ctx.exchange_mode())
        regions[kind] = region
    ctx_mstep_form = ctx.mstep_form()
    head_kind = kinds[0]
    head = regions[head_kind]
    N, B = problem.n_calls, problem.n_barcodes   # of the problem that is resident now (n = 1: the whole workload)

    # ---- the same timed region with the bit-exact E-step and additions (same resident problem) ----
    extra_modes = {}
    wanted = ([] if args.no_exact_mode or default_mode == 'exact' else ['exact']) + (['fast'] if args.fast_mode else [])
    for mode in wanted:
        phase(f'{mode} mode: timed region')
        ctx.set_estep_mode(mode)
        ctx.set_exact_additions(mode == 'exact')
        ctx.set_addition(None)
        ctx.probs_from_betas(0.01, fetch=False)
        ctx.estep(pen, with_doublets=dp > 0, fetch_logits=False, fetch_probs=False)
        probs_mode = ctx.get_block('probs', 0, len(probs0)) if (world == 1 and probs0 is not None) else None
        best_mode = ctx.get_assignments()[0] if probs_mode is not None else None
        region = timed_region(ctx, plane, args.steps, args.warmup, snapshot=world == 1 and mode == 'exact')
        region['value'] = regions[kinds[-1]]['barcodes_total'] * args.steps / region['elapsed']
        region['scaling'] = kinds[-1]
        if probs_mode is not None:
            region['first_pass_vs_default_mode'] = dict(
                argmax_identical_all_barcodes=bool(np.array_equal(best_mode, best0)),
                max_abs_posterior_diff_first_rows=float(np.abs(probs_mode - probs0).max()), rows_compared=len(probs0))
        extra_modes[mode] = region
    # What the bench times, checked: the end state of the default-mode region (best option and posteriors of EVERY barcode after
    # W + 2K iterations from the prior: warm-up, timed window, the window again with the phase timers) against the end state of the
    # exact-mode region after the same iterations from the same start - north_star's contract on the outputs of the EM loop.
    parity_timed = None
    end_default, end_exact = regions[kinds[-1]].get('end_state'), (extra_modes.get('exact') or {}).get('end_state')
    if end_default is not None and end_exact is not None:
        diff = np.abs(end_default['probs'] - end_exact['probs'])
        parity_timed = {'argmax_identical': bool(np.array_equal(end_default['best'], end_exact['best'])),
                        'assignments_differing': int((end_default['best'] != end_exact['best']).sum()),
                        'max_abs_posterior_diff': float(diff.max()), 'within_1e-5': bool(diff.max() <= 1e-5),
                        'barcodes': int(len(end_default['best'])), 'em_iterations': int(end_default['iterations']), 'against': 'exact_mode region, same start'}
    ctx.apply_environment()
    work_item_region = None
    took_tiles = ctx_mstep_form == 'tiles'
    if plane is not None:  # every rank or none (the region has barriers)
        took_tiles = plane.all_ok(took_tiles, 'work-item M-step on this rank')[0]
    if args.mstep != 'items' and not args.timed_only and took_tiles:
        ctx.set_mstep_tiles('never')
        ctx.set_addition(None)
        ctx.probs_from_betas(0.01, fetch=False)
        ctx.estep(pen, with_doublets=dp > 0, fetch_logits=False, fetch_probs=False)
        work_item_region = timed_region(ctx, plane, args.steps, args.warmup)
        work_item_region['value'] = regions[kinds[-1]]['barcodes_total'] * args.steps / work_item_region['elapsed']
        work_item_region['scaling'] = kinds[-1]
        ctx.set_mstep_tiles('always' if args.mstep == 'tiles' else 'auto')

    incr_mstep_region = None
    if world == 1 and not args.timed_only and not args.incremental_mstep and ctx_mstep_form == 'tiles' and default_mode != 'exact':  # (one rank: the region has barriers)
        phase('default mode with the incremental M-step (the library default): timed region')
        ctx.set_mstep_incremental(True)
        ctx.set_addition(None)
        ctx.probs_from_betas(0.01, fetch=False)
        ctx.estep(pen, with_doublets=dp > 0, fetch_logits=False, fetch_probs=False)
        incr_mstep_region = timed_region(ctx, plane, args.steps, args.warmup)
        incr_mstep_region['value'] = regions[kinds[-1]]['barcodes_total'] * args.steps / incr_mstep_region['elapsed']
        incr_mstep_region['scaling'] = kinds[-1]
        ctx.set_mstep_incremental(False)

    fine_only_region = None
    if world == 1 and not args.timed_only and default_mode == 'guarded' and regions[kinds[-1]]['estep_passes']['coarse'] > 0:  # (one rank: the region has barriers)
        phase('default mode without the coarse pass: timed region')
        ctx.set_coarse_pass(False)
        ctx.set_addition(None)
        ctx.probs_from_betas(0.01, fetch=False)
        ctx.estep(pen, with_doublets=dp > 0, fetch_logits=False, fetch_probs=False)
        fine_only_region = timed_region(ctx, plane, args.steps, args.warmup)
        fine_only_region['value'] = regions[kinds[-1]]['barcodes_total'] * args.steps / fine_only_region['elapsed']
        fine_only_region['scaling'] = kinds[-1]
        ctx.set_coarse_pass(True)

    logits_kept_region = None
    if world == 1 and not args.timed_only and default_mode == 'guarded':
        phase('default mode, logits of the last E-step kept: timed region')
        ctx.set_logits_needed(True)
        ctx.set_addition(None)
        ctx.probs_from_betas(0.01, fetch=False)
        ctx.estep(pen, with_doublets=dp > 0, fetch_logits=False, fetch_probs=False)
        logits_kept_region = timed_region(ctx, plane, args.steps, args.warmup)
        logits_kept_region['value'] = regions[kinds[-1]]['barcodes_total'] * args.steps / logits_kept_region['elapsed']
        ctx.set_logits_needed(False)

    clocks_warm_region = None
    if world == 1 and not args.timed_only:
        phase('default mode behind a clock warm-up: timed region')
        ctx.set_addition(None)
        ctx.probs_from_betas(0.01, fetch=False)
        ctx.estep(pen, with_doublets=dp > 0, fetch_logits=False, fetch_probs=False)
        clocks_warm_region = timed_region(ctx, plane, args.steps, args.warmup, spin=estep_again)
        clocks_warm_region['value'] = regions[kinds[-1]]['barcodes_total'] * args.steps / clocks_warm_region['elapsed']

    # What a learn_genotypes(n_iterations=5) call (the reference's default, demux.py:38) costs per iteration when it starts from the
    # prior: the dictionary-form first E-step, the stream / record builds, the first guard decisions - none of which the steady-state
    # `value` contains.  One dmx_em call exactly as demuxalot_amd/demux.py makes it, three times, the median.
    cold_region = None
    if world == 1 and not args.timed_only:
        phase('cold start: 5-iteration calls from the prior')
        _apply_environment()   # the library's own defaults: incremental M-step, form chosen for a 5-iteration call
        ctx.set_logits_needed(False)
        times = []
        for _ in range(3):
            install(problem)   # a fresh problem: no records, streams or pass prices of an earlier call
            ctx.synchronize()
            t0 = time.perf_counter()
            ctx.em(5, 0.01, pen, with_doublets=dp > 0, fetch_logits=False, fetch_probs=False, fetch_addition=False)
            ctx.synchronize()
            times.append(time.perf_counter() - t0)
        levels = ctx.guard_levels()
        cold_region = {'ms_per_step': 1e3 * sorted(times)[1] / 5, 'calls_ms': [1e3 * t for t in times], 'mstep_form': ctx.mstep_form(),
                       'coarse_steps_of_the_last_call': levels['coarse_steps'],
                       'note': 'dmx_em(5 iterations) from the prior on a freshly installed problem (every record / stream build inside the call), results left on the device: median of three calls / 5'}
        ctx.apply_environment()

    after_idle_region = None
    if world == 1 and not args.timed_only:
        phase('default mode on a device that has idled: timed region')
        ctx.set_addition(None)
        ctx.probs_from_betas(0.01, fetch=False)
        ctx.estep(pen, with_doublets=dp > 0, fetch_logits=False, fetch_probs=False)
        if args.mstep == 'auto':   # (the cold-start region installed the problem anew: announce the run length as the headline region does)
            ctx.set_msteps_expected(args.warmup + args.steps)
        after_idle_region = timed_region(ctx, plane, args.steps, args.warmup, idle_s=0.5)
        after_idle_region['value'] = regions[kinds[-1]]['barcodes_total'] * args.steps / after_idle_region['elapsed']

    # predict_posteriors throughput on the same resident problem (P + E only, no beta addition: demux.py:120-156),
    # rank-local.  The genotype table is then the importers' (a handful of distinct values per row), which is the
    # case the dictionary form of the exact E-step exists for (csrc/estep_dict.hip); timed with the form on (default)
    # and off, same bits either way.
    predict = {}
    if not args.timed_only:
        n_pred = max(3, args.steps // 2)
        for mode in ('auto', 'never'):
            ctx.set_estep_dictionary(mode)
            ctx.set_addition(None)
            ctx.probs_from_betas(0.01, fetch=False)
            ctx.estep(pen, with_doublets=dp > 0, fetch_logits=False, fetch_probs=False)  # untimed first pass
            ctx.synchronize()
            ctx.set_phase_timers(True)
            ctx.reset_timings()
            t1 = time.perf_counter()
            for _ in range(n_pred):
                ctx.probs_from_betas(0.01, fetch=False)
                ctx.estep(pen, with_doublets=dp > 0, fetch_logits=False, fetch_probs=False)
            ctx.synchronize()
            seconds = (time.perf_counter() - t1) / n_pred
            t_pred = ctx.timings()
            ctx.set_phase_timers(False)
            form, distinct = ctx.estep_form()
            pe_ms = t_pred['estep']['ms'] / max(1, t_pred['estep']['launches'])
            predict['dictionary_form' if mode == 'auto' else 'direct_form'] = dict(
                seconds=seconds, form=form, distinct_values_per_row=distinct, estep_ms=pe_ms,
                pstep_ms=t_pred['pstep']['ms'] / max(1, t_pred['pstep']['launches']),
                estep_hbm_frac=algorithmic_bytes(B, V, G, K, N)['estep'] / (pe_ms * 1e-3) / 8e12)
        ctx.set_estep_dictionary('auto')
        predict['note'] = ('P-step + E-step on the table without beta addition (predict_posteriors, EM iteration 0); estep_ms includes '
                           'building the dictionary; where the dictionary form runs the pass is bit-identical to the reference in every mode but `fast`')

    # dmx_set_lean_memory (default off): the fine pass's copy of the E-step records and the dictionary form's row array released - the
    # footprint and the same timed region on what is left.  The last region on this problem: the release is for good.
    lean_region = None
    if world == 1 and not args.timed_only and default_mode == 'guarded' and dp == 0:
        phase('lean memory: timed region')
        ctx.set_addition(None)
        ctx.probs_from_betas(0.01, fetch=False)
        ctx.estep(pen, with_doublets=False, fetch_logits=False, fetch_probs=False)
        if args.mstep == 'auto':
            ctx.set_msteps_expected(args.warmup + args.steps)
        bytes_before = ctx.device_bytes()
        ctx.set_lean_memory(True)
        lean = timed_region(ctx, plane, args.steps, args.warmup)
        lean_bytes = ctx.device_bytes()
        ctx.set_lean_memory(False)
        if lean_bytes < bytes_before:  # (a shape without the coarse pass has nothing to release)
            lean_region = {'ms_per_step': lean['ms_per_step'], 'estep_passes': lean.get('estep_passes'),
                           'device_bytes_per_call': (lean_bytes - 8 * B * K) / max(1, N), 'device_bytes': lean_bytes,
                           'note': 'dmx_set_lean_memory(1) / DEMUXALOT_AMD_LEAN=1: the same timed region after the tile-major E-step stream and the dictionary form\'s row '
                                   'array were released (E-steps that keep their logits then run the tolerance kernel on the barcode-major records)'}

    hard = None
    if world == 1 and not args.no_hard_workload and not args.flat_genotypes and N <= 200_000_000:
        phase('hard workloads')
        hard = hard_workload(args, ctx, whole, betas, pen, dp)

    if rank == 0:
        ab = algorithmic_bytes(B, V, G, K, N)
        e_ms = regions[kinds[-1]]['kernel_ms']['estep']
        mode_notes = {'guarded': ': contract of the path proven per barcode and E-step, the rest redone bit-exactly (include/demux_hip.h '
                                 'DMX_ESTEP_GUARDED); E-steps whose logits nobody can read - all but the last of a call: estep_passes - may take the '
                                 'coarse pass (genotype table as binary16, dmx_set_coarse_pass), the timed region is ONE dmx_run_iterations call of '
                                 '`steps` iterations made as learn_genotypes makes its call: no logits returned (dmx_set_logits_needed(0)), so the last E-step '
                                 'is such an E-step too; exact_mode = everything bit-identical to the reference',
                      'exact': ': logits, posteriors and additions bit-identical to the reference',
                      'fast': ': tolerance mode without the guard'}
        mode_short = {'guarded': 'guarded (1e-5 / arg-max contract proven per barcode, else exact redo; coarse f16-table pass where no logits are read)',
                      'exact': 'exact (bit-identical to the reference)', 'fast': 'fast (tolerance mode, no guard)'}
        out = {
            'metric': f'EM iterations/sec + barcodes demuxed/sec, {B_workload // 1000}k bc x {S // 1000}k SNP x {G} gt',
            'metric_note': ('in total' + (f', barcodes sharded over {world} GPUs' if world > 1 else '') if head_kind == 'strong' else 'per GPU') +
                           ': value = barcodes/s through full learn_genotypes EM iterations (P-step + E-step + softmax + '
                           'M-step [+ exchange]) = barcodes x em_iterations_per_s; predict-only rate in predict_barcodes_per_s',
            'value': head['value'],
            'unit': 'barcodes/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': head['ms_per_step'],
            'higher_is_better': True, 'scaling': head_kind, 'vs_baseline': None,
            'dtype': 'f32 terms, f64 sums' + (' (guarded mode: coarse pass reads the table as f16 and sums in f32; exact redo where the contract is not proven)'
                                              if default_mode == 'guarded' else ''), 'data': 'synthetic',
            'config': {'workload': args.workload, 'barcodes_total': head['barcodes_total'], 'barcodes_per_gpu': head['barcodes_per_gpu'],
                       'snps': S, 'variants': V, 'genotypes': G, 'options': K, 'calls_per_gpu': head['calls_per_gpu'], 'doublet_prior': dp,
                       'estep_mode': mode_short[default_mode], 'estep_mode_note': default_mode + mode_notes[default_mode],
                       'mstep_form': f'{ctx_mstep_form} (--mstep {args.mstep})',
                       'parallelism': f'barcode shards x{world}' + (
                           {'variant': ', variant-sharded M-step: all-gather of posteriors + genotype_prob slices',
                            'reduce_scatter': f', reduce-scatter {args.reduce_dtype} of partial sums + all-gather of genotype_prob slices',
                            'allreduce': f', all-reduce {args.reduce_dtype} of partial sums'}.get(head.get('exchange'), '')
                           + (' (host-staged)' if args.host_plane else ' (RCCL)') if use_dist else ''),
                       'runtimes': runtimes, **({'rccl_fallback': rccl_fallback} if rccl_fallback else {})},
            'em_iterations_per_s': head['em_iterations_per_s'],
            'device_bytes': {'total': head['device_bytes'], 'per_call': head['device_bytes'] / max(1, head['calls_per_gpu']),
                             'results_logits_and_posteriors': 8 * head['barcodes_per_gpu'] * K,
                             'per_call_without_the_results': (head['device_bytes'] - 8 * head['barcodes_per_gpu'] * K) / max(1, head['calls_per_gpu'])},
            'mstep_records_build_ms': head['mstep_records_build_ms'],
            'ms_per_step_incl_record_build': head['ms_per_step'] + head['mstep_records_build_ms'] / max(1, args.warmup + args.steps),
            'kernel_ms': head['kernel_ms'],
            'ms_per_step_with_phase_timers': head['ms_per_step_with_phase_timers'],
            'kernel_ms_note': 'per phase, HIP events on the library\'s stream (dmx_set_phase_timers): the timed call made once more behind the timed window with the events on - an event '
                              'record is a barrier packet of its own, 6 us at each of the four phase boundaries, which a default call does not pay and `value` does not contain',
            'exchange_ms_per_step': head['exchange_ms_per_step'],
            'guard': head['guard'],
            'estep_passes': head['estep_passes'],
            'mstep_passes': head['mstep_passes'],
            'setup_s': {'problem': t_gen, 'problem_source': problem_source, 'upload': t_up},
        }
        if 'weak' in regions and head_kind != 'weak':
            out['weak'] = {k: regions['weak'][k] for k in ('value', 'ms_per_step', 'em_iterations_per_s', 'barcodes_total', 'barcodes_per_gpu',
                                                           'calls_per_gpu', 'kernel_ms', 'exchange_ms_per_step', 'guard', 'exchange')}
            out['weak']['note'] = ('the workload per GPU: every rank holds a 200k-barcode shard (a copy of the generated one) of an '
                                   'N x 200k-barcode experiment')
        for mode, region in extra_modes.items():
            out[f'{mode}_mode'] = {k: v for k, v in region.items() if k not in ('elapsed', 'end_state')}
        if work_item_region:
            out['work_item_mstep'] = {k: v for k, v in work_item_region.items() if k not in ('elapsed', 'end_state')}
            out['work_item_mstep']['note'] = ('the same timed region with the work-item M-step, the form of runs with fewer than 8 M-steps '
                                              'ahead (the tile-major records cost a 2.6 ms sort of the calls to build)')
        if incr_mstep_region:
            out['with_incremental_mstep'] = {k: v for k, v in incr_mstep_region.items() if k not in ('elapsed', 'end_state')}
            out['with_incremental_mstep']['note'] = ('the same timed region as a DEFAULT call of the library runs it (dmx_set_mstep_incremental(1)): after one full pass the tile-major '
                                                     'M-step\'s integer sums stay on the device and an M-step only adds the differences for the barcodes whose posteriors changed where it '
                                                     'matters - the full pass\'s bits (tests/test_gpu_mstep_tiles.py); mstep_passes says how many M-steps did what.  Not the headline: the '
                                                     'synthetic experiment has converged by the timed iterations (the 6th to 25th of the run), so the incremental step has next to nothing '
                                                     'left to do; `value` recomputes every sum in every iteration')
        if logits_kept_region:
            out['with_logits_of_the_last_estep'] = {k: v for k, v in logits_kept_region.items() if k not in ('elapsed', 'end_state')}
            out['with_logits_of_the_last_estep']['note'] = ('the same timed region with dmx_set_logits_needed(1), the C ABI\'s default: the last E-step of the call takes the fine pass '
                                                            '(float32 table), so that its logits are the ones the contract describes; `value` is the call learn_genotypes makes, which returns posteriors only')
        if clocks_warm_region:
            out['clocks_warm'] = {k: clocks_warm_region[k] for k in ('value', 'ms_per_step', 'em_iterations_per_s', 'ms_per_step_with_phase_timers', 'kernel_ms', 'clock_warmup_esteps')}
            out['clocks_warm']['note'] = ('the same timed region behind %g ms of E-steps on the standing table (no EM progress): ' % CLOCK_WARMUP_MS
                                          + clock_warmup.__doc__.split('\n\n')[0].replace('\n    ', ' ') + '  NOT the headline: `value` is W warm-up iterations + K timed ones, nothing else')
        if cold_region:
            out['cold_start_5it'] = cold_region
        if lean_region:
            out['lean_memory'] = lean_region
        if after_idle_region:
            out['after_idle'] = {k: after_idle_region[k] for k in ('value', 'ms_per_step', 'em_iterations_per_s', 'ms_per_step_with_phase_timers', 'kernel_ms')}
            out['after_idle']['note'] = ('the same timed region - W warm-up iterations, K timed ones - begun 0.5 s after the device was last busy, without the clock warm-up: what the first '
                                         'short call on an idle device sees (the clocks come up over its first 30 ms)')
        if fine_only_region:
            out['without_coarse_pass'] = {k: v for k, v in fine_only_region.items() if k not in ('elapsed', 'end_state')}
            out['without_coarse_pass']['note'] = ('the same timed region with dmx_set_coarse_pass(0): every E-step the fine pass on the float32 table (the default mode of round 4; '
                                                  'what the LAST E-step of every call runs in any case - its logits are the ones a caller can read)')
        if parity_timed:
            out['parity_timed'] = parity_timed
        if hard:
            out['hard_workload'] = hard
        if predict:
            out['predict'] = predict
            out['predict_barcodes_per_s'] = B / predict['dictionary_form']['seconds']
        live = {}
        if world == 1 and not args.no_live_traffic:
            cache = os.environ.get('DEMUXALOT_BENCH_PROBLEM', '')
            own = not cache
            if own:
                cache = shared_directory() + f'_pmc{os.getpid()}'
                save_problem(cache, whole)
            try:
                live = live_counters(args, cache)
            finally:
                if own:
                    import shutil
                    shutil.rmtree(cache, ignore_errors=True)
        passes = regions[kinds[-1]].get('estep_passes', {})
        out['roofline'] = roofline(ab, e_ms, N, G, K, live, coarse_share=passes.get('coarse', 0) / max(1, passes.get('of', 1)), passes=passes, ms_per_step=head['ms_per_step'])
        if world == 1 and not args.no_e2e and not args.flat_genotypes and N > 200_000_000:
            out['e2e'] = {'skipped': 'the object form of an experiment of this size (1.3 M dictionary entries, 1 M barcode strings, 4e8 container records) is '
                                     'minutes of Python before any call is made'}
        elif world == 1 and not args.no_e2e and not args.flat_genotypes:
            ctx.close()  # the end-to-end calls bring their own contexts; free this one's 4 GB first
            out['e2e'] = e2e_timing(whole, dp)
        if world == 1 and not args.no_cpu_baseline:
            base, ref_logits, ref_post, n_s = cpu_baseline(whole, betas, dp)
            out['cpu_baseline'] = base
            out['cpu_baseline']['speedup_vs_gpu_value'] = out['value'] / base['value']
            # sanity: the GPU rows of the sampled barcodes against the oracle's (first pass = importers' table: bit-exact in every mode)
            n_c = min(n_s, len(probs0))
            out['parity_on_sample'] = {
                'rows_compared': n_c,
                'argmax_identical': bool(np.array_equal(ref_post[:n_c].argmax(1), probs0[:n_c].argmax(1))),
                'max_abs_posterior_diff': float(np.abs(ref_post[:n_c] - probs0[:n_c]).max()),
                'logits_bitwise_equal': bool(np.array_equal(ref_logits[:n_c].view(np.uint32), logits0[:n_c].view(np.uint32))),
            }
        else:
            out['cpu_baseline'] = None
        if isinstance(out.get('e2e'), dict) and 'split_s' in out['e2e']:
            out['default_call_5it_ms_per_iteration'] = 1e3 * out['e2e']['split_s']['em_5_iterations_fused'] / 5
        details = write_details(out, args.workload, world)
        print(json.dumps(out), file=sys.stderr)   # (the whole result for a reader of the log; stdout carries the compact line only)
        sys.stdout.flush()
        sys.stderr.flush()
        os.write(json_fd, (compact_line(out, details) + '\n').encode())
    phase('final barrier')
    if plane is not None:
        plane.barrier()
        if rank == 0 and world > 1 and not os.environ.get('DEMUXALOT_BENCH_PROBLEM'):
            import shutil
            shutil.rmtree(shared_directory(), ignore_errors=True)
        plane.close()
    phase('done')


if __name__ == '__main__':
    main()
