"""World sizes 2, 3 and 4 on ONE GPU: every rank is a thread with its own DeviceContext, the exchange of the library
runs over caller-provided collectives (dmx_comm_init_host; tests/thread_plane.py).  Everything on the data path
except the RCCL calls themselves is what an N-GPU run executes: the padded variant slices, the partial sums written
straight into the exchange buffer (prow), reduce-scatter -> rounded slice -> sliced P-step -> all-gather of
genotype_prob, the re-based E-step records, the all-reduce fallback for scattered SNP groups, and the Python
sharding (distributed.py) on top.  (RCCL itself is exercised with one rank in tests/test_gpu_parity.py.)"""
import os

import numpy as np
import pytest

from tests import fixture_io as fio
from tests.thread_plane import ThreadWorld

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('world', [2, 3])
@pytest.mark.parametrize('name', ['f2_synthetic_g4.npz', 'f3_small_3.npz', 'f1_synthetic_default.npz'])
def test_sharded_entry_points_match_the_reference(name, world):
    from demuxalot_amd import distributed
    fx = fio.load(name)
    calls, genotypes, handler = fio.product_inputs(fx)
    kwargs = dict(n_iterations=int(fx['em0_n_iterations']), p_genotype_clip=float(fx['em0_clip']),
                  doublet_prior=float(fx['em0_dp']))
    prior = fx.get('em0_prior_logits')
    shared = ThreadWorld(world)

    def rank_body(plane):
        learnt, probs_df = distributed.learn_genotypes(calls, genotypes, handler, plane, barcode_prior_logits=prior, **kwargs)
        logits_df, p_df = distributed.predict_posteriors(calls, genotypes, handler, plane, p_genotype_clip=float(fx['predict0_clip']),
                                                         doublet_prior=float(fx['predict0_dp']))
        return learnt.variant_betas, probs_df, logits_df, p_df

    results = shared.run(rank_body)
    want_probs = fx[f'em0_it{kwargs["n_iterations"] - 1}_probs']
    ops = {op for op, _dtype, _shape in shared.collectives}
    _cuts, _rows, contiguous = distributed.exchange_slices(fx['pack_v2snp'], world)
    # contiguous SNP groups: the M-step sharded on variants (these experiments have fewer posterior bytes than sum bytes)
    assert ops == ({'all_gather'} if contiguous else {'all_reduce'}), ops
    for betas, probs_df, logits_df, p_df in results:
        assert list(probs_df.index) == [str(b) for b in fx['barcodes']]
        if contiguous:  # variant-sharded M-step: every sum is formed by ONE rank in the reference's order
            fio.assert_bitwise(betas, fx['em0_learnt_betas'], f'{name} learnt betas, world {world}')
        # (all-reduce fallback: float64 re-association over ranks can move a rounding tie of a beta by one float32 ulp)
        assert np.allclose(betas, fx['em0_learnt_betas'], rtol=3e-7, atol=0) and (betas != fx['em0_learnt_betas']).sum() <= 3
        assert np.array_equal(probs_df.values.argmax(1), want_probs.argmax(1))
        assert np.abs(probs_df.values - want_probs).max() <= 1e-5
        # predict needs no exchange: the gathered rows are the reference's, bit for bit
        fio.assert_bitwise(logits_df.values, fx['predict0_logits'], f'{name} predict logits, world {world}')
        fio.assert_bitwise(p_df.values, fx['predict0_probs'], f'{name} predict posteriors, world {world}')
    for other in results[1:]:  # every rank holds the same answers
        assert np.array_equal(other[0], results[0][0]) and np.array_equal(other[1].values, results[0][1].values)


@pytest.mark.parametrize('reduce_dtype', ['f64', 'f32'])
@pytest.mark.parametrize('world', [2, 4])
def test_sharded_em_equals_the_single_context_run(world, reduce_dtype, monkeypatch):
    """6000 barcodes x 3000 SNPs x 24 genotypes with doublets, 4 EM iterations: the posterior rows and the addition
    of the sharded run against one context holding everything - bit for bit, with the M-step sharded on variants
    (forced: with float32 sums on the wire the size rule would pick the exchange of the sums for this shape)."""
    monkeypatch.setenv('DEMUXALOT_AMD_EXCHANGE', 'variant')
    from demuxalot_amd import distributed, synth
    from demuxalot_amd.device import DeviceContext
    p = synth.generate(6000, 3000, 24, calls_per_barcode=80, seed=12)
    betas = p.prior_betas()
    G = 24
    K = G * (G + 1) // 2
    pen = np.zeros(K, dtype=np.float32)
    pen[G:] = np.float32(-1.5)
    with DeviceContext(0) as ctx:
        ctx.set_problem(p.n_barcodes, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
        ctx.set_betas(betas)
        _l, want_probs, want_add = ctx.em(4, 0.01, pen, True, fetch_logits=False)
    shared = ThreadWorld(world)

    def rank_body(plane):
        em = distributed.ShardedEM(plane, p.n_barcodes, p.v2snp, betas, p.variant_id, p.compressed_cb, p.p_base_wrong,
                                   reduce_dtype=reduce_dtype)
        try:
            probs, addition = em.learn(4, 0.01, pen, True)
            return em.lo, em.hi, probs, addition
        finally:
            em.ctx.close()

    results = shared.run(rank_body)
    assert [r[0] for r in results][0] == 0 and results[-1][1] == p.n_barcodes
    # variant-sharded M-step: nothing is added across ranks (reduce_dtype plays no role), so the sharded run IS the single-context one
    for lo, hi, probs, addition in results:
        fio.assert_bitwise(probs, want_probs[lo:hi], f'posterior rows [{lo}, {hi})')
        fio.assert_bitwise(addition, want_add, 'addition')
    kinds = {(op, dtype) for op, dtype, _shape in shared.collectives}
    assert kinds == {('all_gather', 'float32')}, kinds


@pytest.mark.parametrize('exchange', ['variant', 'reduce_scatter'])
@pytest.mark.parametrize('n_barcodes,n_snps,world', [(3, 40, 4), (50, 2, 4), (7, 3, 3)])
def test_sharded_em_with_empty_ranks_and_empty_slices(n_barcodes, n_snps, world, exchange, monkeypatch):
    """Fewer barcodes than ranks (ranks without a barcode or a call) and fewer SNPs than ranks (variant slices without a
    variant): the sharded run still equals the single context."""
    monkeypatch.setenv('DEMUXALOT_AMD_EXCHANGE', exchange)
    from demuxalot_amd import distributed, synth
    from demuxalot_amd.device import DeviceContext
    G = 5
    p = synth.generate(n_barcodes, n_snps, G, calls_per_barcode=12, seed=31 + n_barcodes)
    betas = p.prior_betas()
    pen = np.zeros(G, dtype=np.float32)
    with DeviceContext(0) as ctx:
        ctx.set_problem(p.n_barcodes, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
        ctx.set_betas(betas)
        _l, want_probs, want_add = ctx.em(3, 0.01, pen, False, fetch_logits=False)
    shared = ThreadWorld(world)

    def rank_body(plane):
        em = distributed.ShardedEM(plane, p.n_barcodes, p.v2snp, betas, p.variant_id, p.compressed_cb, p.p_base_wrong, reduce_dtype='f64')
        try:
            probs, addition = em.learn(3, 0.01, pen, False)
            return em.lo, em.hi, probs, addition
        finally:
            em.ctx.close()

    results = shared.run(rank_body)
    assert results[0][0] == 0 and results[-1][1] == p.n_barcodes
    for lo, hi, probs, addition in results:
        assert probs.shape == (hi - lo, G)
        if exchange == 'variant':
            fio.assert_bitwise(probs, want_probs[lo:hi], f'posterior rows [{lo}, {hi})')
            fio.assert_bitwise(addition, want_add, 'addition')
        else:
            assert np.allclose(addition, want_add, rtol=3e-7, atol=0) and np.allclose(probs, want_probs[lo:hi], rtol=0, atol=1e-5)


@pytest.mark.parametrize('n_genotypes,doublets', [(100, False), (130, False), (70, True)])
def test_variant_sharded_mstep_beyond_64_genotypes(n_genotypes, doublets, monkeypatch):
    """Bitmaps of two and three words per barcode (and the wide M-step kernel) through the variant-sharded exchange: three
    ranks against one context, bit for bit."""
    monkeypatch.setenv('DEMUXALOT_AMD_EXCHANGE', 'variant')
    from demuxalot_amd import Demultiplexer, distributed, synth
    from demuxalot_amd.device import DeviceContext
    G = n_genotypes
    p = synth.generate(900, 400, G, calls_per_barcode=60, doublets=doublets, seed=77 + G)
    betas = p.prior_betas()
    pen = Demultiplexer._doublet_penalties(G, 0.2 if doublets else 0.0)
    with DeviceContext(0) as ctx:
        ctx.set_problem(p.n_barcodes, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
        ctx.set_betas(betas)
        _l, want_probs, want_add = ctx.em(3, 0.01, pen, doublets, fetch_logits=False)
    shared = ThreadWorld(3)

    def rank_body(plane):
        em = distributed.ShardedEM(plane, p.n_barcodes, p.v2snp, betas, p.variant_id, p.compressed_cb, p.p_base_wrong, reduce_dtype='f64')
        try:
            assert em.ctx.exchange_mode() == 'variant'
            probs, addition = em.learn(3, 0.01, pen, doublets)
            return em.lo, em.hi, probs, addition
        finally:
            em.ctx.close()

    for lo, hi, probs, addition in shared.run(rank_body):
        fio.assert_bitwise(probs, want_probs[lo:hi], f'posterior rows [{lo}, {hi})')
        fio.assert_bitwise(addition, want_add, 'addition')


def _socket_rank(rank, world, port, out):
    from demuxalot_amd import _lib, distributed
    from demuxalot_amd.plane import SocketControlPlane
    plane = SocketControlPlane(rank, world, '127.0.0.1', port=port, host_collectives=True)  # the exchange over sockets, no RCCL
    try:
        report = {}
        for name in ('f2_synthetic_g4.npz', 'f3_small_3.npz'):
            fx = fio.load(name)
            calls, genotypes, handler = fio.product_inputs(fx)
            kwargs = dict(n_iterations=int(fx['em0_n_iterations']), p_genotype_clip=float(fx['em0_clip']),
                          doublet_prior=float(fx['em0_dp']))
            learnt, probs_df = distributed.learn_genotypes(calls, genotypes, handler, plane, device=0,
                                                           barcode_prior_logits=fx.get('em0_prior_logits'), **kwargs)
            want = fx[f'em0_it{kwargs["n_iterations"] - 1}_probs']
            report[name] = (bool(np.allclose(learnt.variant_betas, fx['em0_learnt_betas'], rtol=3e-7, atol=0)),
                            bool(np.array_equal(probs_df.values.argmax(1), want.argmax(1))),
                            float(np.abs(probs_df.values - want).max()))
        report['hip_runtimes'] = _lib.runtime_info()['hip']
        out.put((rank, report))
    finally:
        plane.close()


def test_two_processes_one_gpu_exchange_over_the_socket_plane():
    """Two PROCESSES on the one GPU, control plane AND per-iteration exchange carried by the torch-free socket plane
    (demuxalot_amd/plane.py, through the caller-provided-collectives entry): the sharded learn_genotypes against the
    reference's outputs, and exactly one HIP runtime mapped in each worker."""
    import multiprocessing as mp
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    procs = [ctx.Process(target=_socket_rank, args=(r, 2, port, out)) for r in range(2)]
    for pr in procs:
        pr.start()
    reports = [out.get(timeout=600) for _ in range(2)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    for _rank, report in reports:
        assert len(report.pop('hip_runtimes')) == 1
        for name, (betas_close, argmax_same, max_dev) in report.items():
            assert betas_close and argmax_same and max_dev <= 1e-5, (name, betas_close, argmax_same, max_dev)


@pytest.mark.parametrize('world', [3])
def test_device_resident_sharded_results_against_pandas(world):
    """results='device' (world 3, F1 = the reference's tests/test_synthetic.py inputs): every rank keeps its rows on the
    GPU; assignments / best / option sums over ALL barcodes equal what pandas makes of the reference's matrix, with only
    O(B) / O(K) numbers crossing ranks; results='root' hands the frame to rank 0 only."""
    import pandas as pd
    from demuxalot_amd import distributed
    fx = fio.load('f1_synthetic_default.npz')
    calls, genotypes, handler = fio.product_inputs(fx)
    i = 1  # a doublet run of the fixture
    dp, clip = float(fx[f'predict{i}_dp']), float(fx[f'predict{i}_clip'])
    want = pd.DataFrame(fx[f'predict{i}_probs'], index=[str(b) for b in fx['barcodes']], columns=[str(c) for c in fx[f'predict{i}_columns']])
    shared = ThreadWorld(world)

    def rank_body(plane):
        with distributed.predict_posteriors(calls, genotypes, handler, plane, p_genotype_clip=clip, doublet_prior=dp,
                                            results='device') as sharded:
            assert sharded.shape == want.shape and sharded.local.shape[0] == sharded.hi - sharded.lo < want.shape[0]
            got = dict(assign=sharded.assignments(0.9), best=sharded.best(), sums=sharded.option_sums(),
                       frame=sharded.to_dataframe(root_only=True))
        logits_root, probs_root = distributed.predict_posteriors(calls, genotypes, handler, plane, p_genotype_clip=clip,
                                                                 doublet_prior=dp, results='root')
        assert (probs_root is None) == (plane.rank != 0) and (logits_root is None) == (plane.rank != 0)
        if plane.rank == 0:
            fio.assert_bitwise(probs_root.values, want.values, 'results=root')
        return got

    for rank, got in enumerate(shared.run(rank_body)):
        expect = want[want.max(axis=1).gt(0.9)].idxmax(axis=1)
        assert got['assign'].index.tolist() == expect.index.tolist() and got['assign'].tolist() == expect.tolist()
        assert got['best']['option'].tolist() == want.idxmax(axis=1).tolist()
        assert np.array_equal(got['best']['probability'].values, want.max(axis=1).values)
        assert np.allclose(got['sums'].values, want.values.astype(np.float64).sum(axis=0), rtol=1e-12, atol=1e-9)
        assert (got['frame'] is None) == (rank != 0)
        if rank == 0:
            fio.assert_bitwise(got['frame'].values, want.values, 'to_dataframe at the root')


@pytest.mark.parametrize('aggregate', [False, True])
def test_sharded_staged_learning(aggregate):
    """distributed.staged_genotype_learning at world 2 against the reference's per-iteration captures: the default
    E-step (float32, the exchange inside the library) and aggregate_on_snps (float64 posteriors; the ranks' float64
    M-step sums added on the host)."""
    from demuxalot_amd import Demultiplexer, distributed
    fx = fio.load('f7_aggregate_synthetic_g4.npz' if aggregate else 'f2_synthetic_g4.npz')
    inputs = fio.load(str(fx['inputs_of'])) if aggregate else fx
    calls, genotypes, handler = fio.product_inputs(inputs)
    n_it = int(fx['em0_n_iterations'])
    kwargs = dict(n_iterations=n_it, doublet_prior=float(fx['em0_dp']))
    if not aggregate:
        kwargs['p_genotype_clip'] = float(fx['em0_clip'])
    if fx.get('em0_prior_logits') is not None:
        kwargs['barcode_prior_logits'] = fx['em0_prior_logits']
    shared = ThreadWorld(2)

    def rank_body(plane):
        return [(frame.values, dbg['barcode_logits'], dbg['genotype_addition'])
                for frame, dbg in distributed.staged_genotype_learning(calls, genotypes, handler, plane, **kwargs)]

    Demultiplexer.aggregate_on_snps = aggregate
    try:
        per_rank = shared.run(rank_body)
    finally:
        Demultiplexer.aggregate_on_snps = False
    for stages in per_rank:
        assert len(stages) == n_it
        for it, (probs, logits, addition) in enumerate(stages):
            want = fx[f'em0_it{it}_probs']
            assert probs.dtype == want.dtype and probs.shape == want.shape
            assert np.array_equal(probs.argmax(1), want.argmax(1)) and np.abs(probs - want).max() <= 1e-5, it
            assert np.allclose(addition, fx[f'em0_it{it}_addition'], rtol=3e-7, atol=1e-12), it
    assert all(np.array_equal(a[2], b[2]) for a, b in zip(per_rank[0], per_rank[1]))  # the same additions on every rank


def _rccl_rank(rank, world, port, out):
    from demuxalot_amd import _lib, distributed
    from demuxalot_amd.plane import SocketControlPlane
    plane = SocketControlPlane(rank, world, '127.0.0.1', port=port)  # control plane only: the exchange is RCCL
    try:
        report = {}
        for name in ('f2_synthetic_g4.npz', 'f3_small_3.npz', 'f1_synthetic_default.npz'):  # F3: the all-reduce fallback
            fx = fio.load(name)
            calls, genotypes, handler = fio.product_inputs(fx)
            kwargs = dict(n_iterations=int(fx['em0_n_iterations']), p_genotype_clip=float(fx['em0_clip']), doublet_prior=float(fx['em0_dp']))
            learnt, probs_df = distributed.learn_genotypes(calls, genotypes, handler, plane, device=rank,
                                                           barcode_prior_logits=fx.get('em0_prior_logits'), **kwargs)
            want = fx[f'em0_it{kwargs["n_iterations"] - 1}_probs']
            report[name] = (bool(np.allclose(learnt.variant_betas, fx['em0_learnt_betas'], rtol=3e-7, atol=0)),
                            bool(np.array_equal(probs_df.values.argmax(1), want.argmax(1))), float(np.abs(probs_df.values - want).max()))
        info = _lib.runtime_info()
        report['runtime'] = (len(info['hip']), info['rccl_loaded'])
        out.put((rank, report))
    finally:
        plane.close()


def test_two_ranks_over_rccl():
    """The default multi-GPU exchange (ncclReduceScatter into the owned slice, in-place ncclAllGather of the padded
    genotype table, the all-reduce fallback) with two ranks on two GPUs, against the reference's outputs.  Skipped on a
    one-GPU box (the driver's scaling node has eight)."""
    import multiprocessing as mp
    import socket
    from demuxalot_amd import _lib
    if _lib.device_count() < 2:
        pytest.skip('needs two GPUs')
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    procs = [ctx.Process(target=_rccl_rank, args=(r, 2, port, out)) for r in range(2)]
    for pr in procs:
        pr.start()
    reports = [out.get(timeout=900) for _ in range(2)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    for _rank, report in reports:
        n_hip, rccl = report.pop('runtime')
        assert n_hip == 1 and rccl, (n_hip, rccl)
        for name, (betas_close, argmax_same, max_dev) in report.items():
            assert betas_close and argmax_same and max_dev <= 1e-5, (name, betas_close, argmax_same, max_dev)


def _bench_rank(rank, world, port, out, scaling, exchange=None, broken_rccl=False):
    """bench.py as one rank of `world` on GPU 0, launched the way torch.distributed.run does (environment only)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
               TORCHELASTIC_RUN_ID=f'benchtest{port}')
    env.pop('DEMUXALOT_AMD_ESTEP', None)  # the bench runs the library's default mode, not the suite's pin (tests/conftest.py)
    if exchange:
        env['DEMUXALOT_AMD_EXCHANGE'] = exchange
    if broken_rccl:
        env['DEMUXALOT_AMD_RCCL'] = '/nonexistent/librccl.so'
    done = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', str(world), '--steps', '2', '--warmup', '1',
                           '--workload', 'em_20k_10k_64', '--scaling', scaling, '--no-cpu-baseline', '--no-fast-mode',
                           '--no-live-traffic', '--no-e2e'] + ([] if broken_rccl else ['--host-plane']),
                          env=env, cwd=root, capture_output=True, text=True, timeout=900)
    out.put((rank, done.returncode, done.stdout.strip(), done.stderr[-2000:]))


def _bench_details(line):
    """The sidecar of a bench line (bench.py: write_details): everything the compact line leaves out."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = line['details'] if os.path.isabs(line['details']) else os.path.join(root, line['details'])
    full = json.load(open(path))
    assert full['value'] == pytest.approx(line['value'], rel=1e-5) and full['n_gpus'] == line['n_gpus']   # the same run's
    return full


@pytest.mark.parametrize('scaling,exchange', [('strong', None), ('weak', None), ('strong', 'reduce_scatter')])
def test_bench_with_two_ranks_on_one_gpu(scaling, exchange):
    """`bench.py --gpus 2 --scaling strong|weak` end to end (socket control plane, exchange staged over the plane): rank 0
    prints ONE JSON line for the whole job, the other rank nothing."""
    import json
    import multiprocessing as mp
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    procs = [ctx.Process(target=_bench_rank, args=(r, 2, port, out, scaling, exchange)) for r in range(2)]
    for pr in procs:
        pr.start()
    results = sorted(out.get(timeout=900) for _ in range(2))
    for pr in procs:
        pr.join(timeout=60)
    for rank, code, stdout, stderr in results:
        assert code == 0, (rank, stderr)
        if rank:
            assert stdout == ''
    line = json.loads(results[0][2])
    assert line['n_gpus'] == 2 and line['scaling'] == scaling and line['value'] > 0
    assert line['config']['barcodes_total'] == (20_000 if scaling == 'strong' else 40_000)
    assert len(results[0][2]) < 4000   # (bench.LINE_LIMIT: the driver's reader)
    assert len(_bench_details(line)['config']['runtimes']['hip']) == 1 and line['exchange_ms_per_step'] > 0


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher and a scrubbed environment: the parent starts the two rank processes
    itself (both on this box's one GPU, exchange staged over the plane), relays rank 0's one JSON line - the headline is
    BASELINE.json configs[3] as written (one experiment, barcodes sharded: strong), the per-GPU workload rides along as
    `weak` - and the experiment is generated once."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'TORCHELASTIC_RUN_ID', 'DEMUXALOT_AMD_ESTEP')}
    done = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                           '--workload', 'em_20k_10k_64', '--host-plane'], env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert done.returncode == 0, done.stderr[-2000:]
    lines = [ln for ln in done.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, done.stdout[-2000:]
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['scaling'] == 'strong' and line['value'] > 0
    assert line['config']['barcodes_total'] == 20_000 and line['config']['estep_mode'].startswith('guarded')
    assert line['weak']['barcodes_total'] == 40_000 and line['weak']['value'] > 0
    full = _bench_details(line)
    assert 'shared through' in full['setup_s']['problem_source'] and line['exchange_ms_per_step'] > 0
    assert full['exact_mode']['value'] > 0 and line['exact_mode_ms_per_step'] > 0


def test_bench_falls_back_to_the_host_plane_without_rccl():
    """No RCCL communicator (here: the library path points nowhere) on a multi-rank run: every rank learns so over the
    control plane and the exchange is staged through host memory; the line says so."""
    import json
    import multiprocessing as mp
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    procs = [ctx.Process(target=_bench_rank, args=(r, 2, port, out, 'weak', None, True)) for r in range(2)]
    for pr in procs:
        pr.start()
    results = sorted(out.get(timeout=900) for _ in range(2))
    for pr in procs:
        pr.join(timeout=60)
    for rank, code, _stdout, stderr in results:
        assert code == 0, (rank, stderr)
    line = json.loads(results[0][2])
    assert line['n_gpus'] == 2 and line['value'] > 0 and line['exchange_ms_per_step'] > 0
    assert 'rccl_fallback' in _bench_details(line)['config'] and 'rccl_fallback' in line['config'] and '(host-staged)' in line['config']['parallelism'], line['config']


@pytest.mark.parametrize('name', ['f2_synthetic_g4.npz', 'f1_synthetic_default.npz'])
def test_dictionary_form_on_the_padded_multi_rank_table(name, monkeypatch):
    """With a communicator attached genotype_prob lives in the padded slice layout and the E-step records / call rows
    are re-based to it; the dictionary form (forced on: these problems are too small for it by default) must walk that
    layout too - sharded predict bitwise, sharded EM within the contract, at world 3."""
    from demuxalot_amd import distributed
    monkeypatch.setenv('DEMUXALOT_AMD_ESTEP_DICT', 'always')  # read by the contexts the ranks create
    fx = fio.load(name)
    calls, genotypes, handler = fio.product_inputs(fx)
    kwargs = dict(n_iterations=int(fx['em0_n_iterations']), p_genotype_clip=float(fx['em0_clip']), doublet_prior=float(fx['em0_dp']))
    shared = ThreadWorld(3)

    def rank_body(plane):
        learnt, probs_df = distributed.learn_genotypes(calls, genotypes, handler, plane, barcode_prior_logits=fx.get('em0_prior_logits'),
                                                       force_comm=True, **kwargs)
        logits_df, p_df = distributed.predict_posteriors(calls, genotypes, handler, plane, p_genotype_clip=float(fx['predict0_clip']),
                                                         doublet_prior=float(fx['predict0_dp']))
        # one explicit look at the form on a padded table: this rank's shard, communicator attached
        from demuxalot_amd.device import DeviceContext
        ctx = DeviceContext(0)
        try:
            distributed.attach_communicator(ctx, plane)
            from demuxalot_amd.demux import _pack_on_device
            lo, hi = [int(x) for x in distributed.partition_barcodes(distributed.calls_per_barcode(calls, handler.n_barcodes), plane.world)[plane.rank:plane.rank + 2]]
            _pack_on_device(distributed.shard_containers(calls, lo, hi), genotypes, hi - lo, False, fetch_betas=False, ctx=ctx)
            ctx.set_addition(None)
            ctx.probs_from_betas(0.01, fetch=False)
            ctx.estep(np.zeros(genotypes.n_genotypes, dtype=np.float32), with_doublets=False, fetch_logits=False, fetch_probs=False)
            form = ctx.estep_form()
        finally:
            ctx.close()
        return learnt.variant_betas, probs_df.values, logits_df.values, p_df.values, form

    want = fx[f'em0_it{kwargs["n_iterations"] - 1}_probs']
    for betas, probs, logits, predicted, form in shared.run(rank_body):
        assert form[0] == 'dict', form
        assert np.allclose(betas, fx['em0_learnt_betas'], rtol=3e-7, atol=0)
        assert np.array_equal(probs.argmax(1), want.argmax(1)) and np.abs(probs - want).max() <= 1e-5
        fio.assert_bitwise(logits, fx['predict0_logits'], 'sharded predict logits through the dictionary form')
        fio.assert_bitwise(predicted, fx['predict0_probs'], 'sharded predict posteriors through the dictionary form')


@pytest.mark.parametrize('world', [2, 3, 4])
def test_variant_sharded_mstep_is_bit_identical_to_the_reference(world, monkeypatch):
    """The default exchange (csrc/dmx_api.cpp: shard_mstep_by_variant): the ranks all-gather what the M-step reads of a
    barcode and every rank sums ITS variant slice over the barcodes of all ranks, in the reference's order - nothing is
    added across ranks.  So in the exact mode (the suite's pin) a sharded run reproduces the reference's captured
    outputs BIT FOR BIT at any number of ranks - learnt betas, posteriors -, with no reduce-scatter on the wire; round
    3's exchange (DEMUXALOT_AMD_EXCHANGE=reduce_scatter), which adds per-rank partial sums, agrees within a float32 ulp."""
    from demuxalot_amd import distributed
    fx = fio.load('f1_synthetic_default.npz')
    calls, genotypes, handler = fio.product_inputs(fx)
    n_it = int(fx['em0_n_iterations'])
    kwargs = dict(n_iterations=n_it, p_genotype_clip=float(fx['em0_clip']), doublet_prior=float(fx['em0_dp']))

    def run(mode):
        if mode:
            monkeypatch.setenv('DEMUXALOT_AMD_EXCHANGE', mode)
        else:
            monkeypatch.delenv('DEMUXALOT_AMD_EXCHANGE', raising=False)
        shared = ThreadWorld(world)

        def rank_body(plane):
            learnt, probs_df = distributed.learn_genotypes(calls, genotypes, handler, plane, **kwargs)
            return learnt.variant_betas, probs_df.values
        return shared.run(rank_body), shared.collectives

    results, ops = run(None)
    assert sum(op == 'reduce_scatter' for op, _d, _s in ops) == 0 and sum(op == 'all_gather' for op, _d, _s in ops) > 0
    for betas, probs in results:
        fio.assert_bitwise(betas, fx['em0_learnt_betas'], f'learnt betas, {world} ranks vs the reference')
        fio.assert_bitwise(probs, fx[f'em0_it{n_it - 1}_probs'], f'posteriors, {world} ranks vs the reference')
    by_sums, ops = run('reduce_scatter')
    assert sum(op == 'reduce_scatter' for op, _d, _s in ops) == n_it - 1
    for betas, probs in by_sums:
        assert np.allclose(betas, fx['em0_learnt_betas'], rtol=3e-7, atol=0)
        assert np.abs(probs - fx[f'em0_it{n_it - 1}_probs']).max() <= 1e-5


@pytest.mark.parametrize('mode', ['exact', 'guarded'])
def test_compact_exchange_of_the_posterior_rows(mode, monkeypatch):
    """Variant-sharded M-step, G <= 64 (include/demux_hip_debug.h: dmx_get_exchange_compact): a barcode with ONE live posterior is
    described by its 8-byte code, which travels anyway; only the rows of the others travel, in a list of bounded capacity, and the
    receivers rebuild the table.  10 000 barcodes x 3 000 SNPs x 48 genotypes on 3 ranks, 6 iterations: with the compact form
    (default), with a capacity of 16 rows (every list overflows: the whole table travels, decided alike on every rank) and with the
    compact form off - the same bits every time, equal to ONE context holding everything; and the compact form did carry the
    exchange where it could."""
    monkeypatch.setenv('DEMUXALOT_AMD_EXCHANGE', 'variant')
    monkeypatch.setenv('DEMUXALOT_AMD_ESTEP', mode)
    from demuxalot_amd import distributed, synth
    from demuxalot_amd.device import DeviceContext
    G, world, n_it = 48, 3, 6
    p = synth.generate(10_000, 3000, G, calls_per_barcode=400, seed=61)
    betas = p.prior_betas()
    pen = np.zeros(G, dtype=np.float32)
    with DeviceContext(0) as ctx:
        ctx.set_problem(p.n_barcodes, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
        ctx.set_betas(betas)
        ctx.set_mstep_tiles('never')           # (a rank's M-step is the work-item form: the same float64 order)
        ctx.set_mstep_incremental(False)
        _l, want_probs, want_add = ctx.em(n_it, 0.01, pen, False, fetch_logits=False)
    outcomes = {}
    for setting in (None, '16', '0'):
        if setting is None:
            monkeypatch.delenv('DEMUXALOT_AMD_EXCHANGE_COMPACT', raising=False)
        else:
            monkeypatch.setenv('DEMUXALOT_AMD_EXCHANGE_COMPACT', setting)
        shared = ThreadWorld(world)

        def rank_body(plane):
            em = distributed.ShardedEM(plane, p.n_barcodes, p.v2snp, betas, p.variant_id, p.compressed_cb, p.p_base_wrong)
            try:
                probs, addition = em.learn(n_it, 0.01, pen, False)
                return em.lo, em.hi, probs, addition, em.ctx.exchange_compact(), em.ctx.exchange_compact_table()
            finally:
                em.ctx.close()

        outcomes[setting] = shared.run(rank_body)
    for setting, results in outcomes.items():
        for lo, hi, probs, addition, _stats, _table_stats in results:
            if mode == 'exact':
                fio.assert_bitwise(probs, want_probs[lo:hi], f'compact={setting}: posterior rows [{lo}, {hi})')
                fio.assert_bitwise(addition, want_add, f'compact={setting}: addition')
            fio.assert_bitwise(probs, outcomes['0'][[r[0] for r in outcomes['0']].index(lo)][2], f'compact={setting} vs off: posterior rows')
            fio.assert_bitwise(addition, outcomes['0'][0][3], f'compact={setting} vs off: addition')
    taken, overflows, cap = outcomes[None][0][4]
    assert cap >= 64 and taken >= 2 and taken + overflows == n_it - 1, outcomes[None][0][4]   # (one exchange per M-step; the first ones may overflow)
    taken16, overflows16, cap16 = outcomes['16'][0][4]
    assert cap16 == 16 and overflows16 == n_it - 1 and taken16 == 0, outcomes['16'][0][4]
    assert outcomes['0'][0][4] == (0, 0, 0)
    # ... and so for the table behind the sliced P-step: the rows of a slice that changed since they were sent (the first table of a
    # layout travels whole: nobody holds anything yet)
    t_taken, t_overflows, t_cap = outcomes[None][0][5]
    assert t_cap >= 64 and t_taken >= 2 and t_taken + t_overflows == n_it - 1, outcomes[None][0][5]
    assert outcomes['16'][0][5][0] == 0 and outcomes['16'][0][5][1] == n_it - 1 and outcomes['0'][0][5] == (0, 0, 0), (outcomes['16'][0][5], outcomes['0'][0][5])
    print(f'{mode}: compact exchanges of the posteriors {taken} of {n_it - 1} (capacity {cap} rows per rank), of the table {t_taken} of {n_it - 1} '
          f'(capacity {t_cap}); capacity 16: {overflows16} / {outcomes["16"][0][5][1]} fallbacks')


@pytest.mark.parametrize('row_index', [True, False])
@pytest.mark.parametrize('world', [2, 3])
def test_incremental_mstep_of_a_variant_sharded_rank(world, row_index, monkeypatch):
    """A rank of a variant-sharded run sums ITS variant slice over the barcodes of all ranks.  With the tile-major records of the slice
    its sums are integers, so they can be kept and updated (kernels.h: MIncrArgs::changed_map): the changed barcodes are found in the
    gathered tables, the delta pass visits the changed barcodes' calls through the slice's records sorted by barcode row (row_index;
    build_slice_row_index) or is a masked walk of the slice's variant-major records (DEMUXALOT_AMD_SLICE_INDEX=0).  8 iterations on 2 and 3
    ranks with it and with every M-step the full tile pass: posteriors and additions bit for bit, and the delta pass did run."""
    monkeypatch.setenv('DEMUXALOT_AMD_EXCHANGE', 'variant')
    monkeypatch.setenv('DEMUXALOT_AMD_SLICE_INDEX', '1' if row_index else '0')
    monkeypatch.setenv('DEMUXALOT_AMD_ESTEP', 'guarded')
    from demuxalot_amd import distributed, synth
    G, n_it = 40, 8
    p = synth.generate(9_000, 2500, G, calls_per_barcode=400, seed=77)
    betas = p.prior_betas()
    pen = np.zeros(G, dtype=np.float32)
    outcomes = {}
    for incremental in (True, False):
        shared = ThreadWorld(world)

        def rank_body(plane):
            em = distributed.ShardedEM(plane, p.n_barcodes, p.v2snp, betas, p.variant_id, p.compressed_cb, p.p_base_wrong)
            try:
                em.ctx.set_mstep_tiles('always')
                em.ctx.set_mstep_incremental(incremental)
                em.ctx.reset_timings()
                probs, addition = em.learn(n_it, 0.01, pen, False)
                return em.lo, em.hi, probs, addition, em.ctx.mstep_incremental(), em.ctx.mstep_form()
            finally:
                em.ctx.close()

        outcomes[incremental] = shared.run(rank_body)
    for got, want in zip(outcomes[True], outcomes[False]):
        assert got[:2] == want[:2]
        fio.assert_bitwise(got[2], want[2], f'posterior rows [{got[0]}, {got[1]}) with the incremental M-step')
        fio.assert_bitwise(got[3], want[3], 'addition with the incremental M-step')
        assert got[5] == 'tiles' and want[5] == 'tiles'
        full, delta, last = got[4]
        assert full >= 1 and delta >= 3 and full + delta == n_it - 1, got[4]
        assert want[4] == (0, 0, 0)
    print(f'{world} ranks: (full, delta, barcodes changed in the last M-step) per rank', [r[4] for r in outcomes[True]])


@pytest.mark.parametrize('wire', ['f32', 'f64'])
def test_incremental_mstep_of_a_rank_that_exchanges_sums(wire, monkeypatch):
    """A rank of a run that reduce-scatters the SUMS (many more barcodes than variants: n x 200k-barcode weak scaling) holds all calls of
    its barcodes, like one context: its partial sums are the integers of the tile-major / fixed-point work-item form and stay in the padded
    exchange buffer between two M-steps, so the incremental M-step applies - the delta pass rewrites the rows it touched (float32 or
    float64, as the wire; the records' padded table rows brought back to variants: MIncrArgs::row_variant), the reduce-scatter reads the
    buffer as before.  8 iterations on 2 ranks: the full tile pass per M-step against the incremental M-step on the tile-major records
    and on the work items (no records) - posteriors and additions bit for bit, and the delta pass did run."""
    monkeypatch.setenv('DEMUXALOT_AMD_EXCHANGE', 'reduce_scatter')
    monkeypatch.setenv('DEMUXALOT_AMD_ESTEP', 'guarded')
    from demuxalot_amd import distributed, synth
    G, n_it, world = 40, 8, 2
    p = synth.generate(9_000, 2500, G, calls_per_barcode=400, seed=78)
    betas = p.prior_betas()
    pen = np.zeros(G, dtype=np.float32)
    outcomes = {}
    for tiles, incremental in (('always', False), ('always', True), ('auto', True)):
        shared = ThreadWorld(world)

        def rank_body(plane):
            em = distributed.ShardedEM(plane, p.n_barcodes, p.v2snp, betas, p.variant_id, p.compressed_cb, p.p_base_wrong, reduce_dtype=wire)
            try:
                assert em.ctx.exchange_mode() == 'reduce_scatter'
                em.ctx.set_mstep_tiles(tiles)
                em.ctx.set_mstep_incremental(incremental)
                em.ctx.reset_timings()
                probs, addition = em.learn(n_it, 0.01, pen, False)
                return em.lo, em.hi, probs, addition, em.ctx.mstep_incremental(), em.ctx.mstep_form()
            finally:
                em.ctx.close()

        outcomes[tiles, incremental] = shared.run(rank_body)
    for key, form in ((('always', True), 'tiles'), (('auto', True), 'items_fixed')):
        for got, want in zip(outcomes[key], outcomes['always', False]):
            assert got[:2] == want[:2]
            fio.assert_bitwise(got[2], want[2], f'posterior rows [{got[0]}, {got[1]}) with the incremental M-step, {key}')
            fio.assert_bitwise(got[3], want[3], f'addition with the incremental M-step, {key}')
            full, delta, last = got[4]
            assert full >= 1 and delta >= 3 and full + delta == n_it - 1, got[4]
            assert got[5] == form and want[5] == 'tiles' and want[4] == (0, 0, 0), (got[5], want[5], want[4])
        print(f'{key} / {wire}: (full, delta, changed) per rank', [r[4] for r in outcomes[key]])
