"""Multi-rank host logic on CPU (no GPU here): the nnz-balanced barcode partition, the container sharding, the
variant slices of the exchange (C ABI: dmx_exchange_slices), and - with two and three gloo processes - the sharded
entry points of demuxalot_amd/distributed.py themselves (ShardedEM, learn_genotypes, predict_posteriors) driven end
to end against the reference's captured outputs.  The device context is replaced by tests/cpu_context.py (oracle
arithmetic; the exchange runs the library's collective sequence - padded slices, reduce-scatter, sliced P-step,
all-gather - over torch.distributed), so everything else under test is product code: partitioning, filtering of the
containers, re-basing of barcodes, the control-plane exchanges (unique id, molecule counts, posterior rows)."""
import os
import socket

import numpy as np
import pytest

from demuxalot_amd import synth
from demuxalot_amd.distributed import partition_barcodes, shard_calls


def test_partition_balances_calls():
    rng = np.random.default_rng(0)
    counts = rng.lognormal(5, 1.0, size=5000).astype(np.int64)
    for world in (1, 2, 3, 8):
        bounds = partition_barcodes(counts, world)
        assert bounds[0] == 0 and bounds[-1] == len(counts) and (np.diff(bounds) >= 0).all() and len(bounds) == world + 1
        loads = np.add.reduceat(counts, bounds[:-1])[:world] if world > 1 else [counts.sum()]
        assert max(loads) - min(loads) <= 2 * counts.max()
    assert list(partition_barcodes([0, 0, 0], 2)) in ([0, 0, 3], [0, 3, 3])
    assert list(partition_barcodes([], 4)) == [0, 0, 0, 0, 0]
    v, cb, e = shard_calls(np.array([5, 6, 7, 8]), np.array([0, 3, 1, 3]), np.array([.1, .2, .3, .4], dtype='f4'), 1, 4)
    assert list(v) == [6, 7, 8] and list(cb) == [2, 0, 2] and e.dtype == np.float32


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_exchange_slices_cut_at_snp_boundaries():
    from demuxalot_amd.distributed import exchange_slices
    v2snp = np.array([0, 0, 1, 1, 1, 2, 3, 3, 4, 4, 4, 4, 5], dtype=np.int32)
    for world in (1, 2, 3, 4, 8, 20):
        cuts, rows, contiguous = exchange_slices(v2snp, world)
        assert contiguous and cuts[0] == 0 and cuts[-1] == len(v2snp) and (np.diff(cuts) >= 0).all()
        assert all(c == len(v2snp) or c == 0 or v2snp[c] != v2snp[c - 1] for c in cuts)  # never inside a SNP
        assert rows == max(1, np.diff(cuts).max())
    cuts, rows, contiguous = exchange_slices(np.array([0, 1, 0, 2], dtype=np.int32), 2)  # SNP 0 is scattered
    assert not contiguous
    cuts, rows, contiguous = exchange_slices(np.zeros(0, dtype=np.int32), 3)
    assert list(cuts) == [0, 0, 0, 0] and contiguous


def test_shard_containers_keep_the_shard_rows_of_the_global_pack():
    """Packing a barcode range of the containers gives exactly that range's rows of the global pack."""
    from demuxalot_amd import Demultiplexer
    from demuxalot_amd.distributed import calls_per_barcode, shard_containers
    from tests import fixture_io as fio
    fx = fio.load('f2_synthetic_g4.npz')
    calls, genotypes, handler = fio.product_inputs(fx)
    counts = calls_per_barcode(calls, handler.n_barcodes)
    assert counts.sum() == sum(c.n_snp_calls for c in calls.values())
    _v2snp, _betas, _mol, whole = Demultiplexer.pack_calls(calls, genotypes, add_data_prior=False)
    bounds = partition_barcodes(counts, 3)
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        _a, _b, _c, part = Demultiplexer.pack_calls(shard_containers(calls, lo, hi), genotypes, add_data_prior=False)
        keep = (whole['compressed_cb'] >= lo) & (whole['compressed_cb'] < hi)
        assert np.array_equal(part['variant_id'], whole['variant_id'][keep])
        assert np.array_equal(part['compressed_cb'], whole['compressed_cb'][keep] - lo)
        assert np.array_equal(part['p_base_wrong'].view(np.uint32), whole['p_base_wrong'][keep].view(np.uint32))


def _worker(rank, world, port, out):
    import torch.distributed as dist
    from demuxalot_amd import distributed
    from tests import fixture_io as fio
    from tests.cpu_context import OracleContext
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        plane = distributed.TorchControlPlane()
        report = {}
        # the gloo data plane (dmx_host_collective semantics: include/demux_hip.h)
        host = distributed.TorchControlPlane(host_collectives=True).host_collective
        buf = np.arange(world * 5, dtype=np.float64).reshape(world, 5) * (rank + 1)
        host('reduce_scatter', buf)
        assert np.array_equal(buf[rank], np.arange(world * 5).reshape(world, 5)[rank] * sum(range(1, world + 1)))
        buf = np.zeros((world, 3), dtype=np.float32)
        buf[rank] = rank + 1
        host('all_gather', buf)
        assert np.array_equal(buf, np.repeat(np.arange(1, world + 1, dtype=np.float32)[:, None], 3, axis=1))
        buf = np.full(4, rank + 1, dtype=np.float32)
        host('all_reduce', buf)
        assert np.array_equal(buf, np.full(4, sum(range(1, world + 1)), dtype=np.float32))
        # F2 / F1: SNP groups contiguous -> reduce-scatter / sliced P-step / all-gather; F3: scattered -> all-reduce
        for name in ('f2_synthetic_g4.npz', 'f3_small_3.npz', 'f1_synthetic_default.npz'):
            fx = fio.load(name)
            calls, genotypes, handler = fio.product_inputs(fx)
            kwargs = dict(n_iterations=int(fx['em0_n_iterations']), p_genotype_clip=float(fx['em0_clip']),
                          doublet_prior=float(fx['em0_dp']))
            prior = fx.get('em0_prior_logits')
            learnt, probs_df = distributed.learn_genotypes(calls, genotypes, handler, plane, context_factory=OracleContext,
                                                           barcode_prior_logits=prior, **kwargs)
            want_probs = fx[f'em0_it{kwargs["n_iterations"] - 1}_probs']
            assert list(probs_df.index) == [str(b) for b in fx['barcodes']] and probs_df.values.shape == want_probs.shape
            report[name] = dict(
                betas_mismatch=int((learnt.variant_betas != fx['em0_learnt_betas']).sum()),
                betas_close=bool(np.allclose(learnt.variant_betas, fx['em0_learnt_betas'], rtol=3e-7, atol=0)),
                argmax_same=bool(np.array_equal(probs_df.values.argmax(1), want_probs.argmax(1))),
                max_dev=float(np.abs(probs_df.values - want_probs).max()))
            # predict needs no exchange at all: the gathered rows are the reference's rows, bit for bit
            dp, clip = float(fx['predict0_dp']), float(fx['predict0_clip'])
            logits_df, p_df = distributed.predict_posteriors(calls, genotypes, handler, plane, p_genotype_clip=clip,
                                                             doublet_prior=dp, context_factory=OracleContext)
            assert logits_df.index.name == 'BARCODE'
            report[name]['predict_bitwise'] = bool(
                np.array_equal(logits_df.values.view(np.uint32), fx['predict0_logits'].view(np.uint32)) and
                np.array_equal(p_df.values.view(np.uint32), fx['predict0_probs'].view(np.uint32)))
        # ShardedEM on packed calls
        p = synth.generate(600, 400, 6, calls_per_barcode=50, seed=3)
        betas = p.prior_betas()
        em = distributed.ShardedEM(plane, p.n_barcodes, p.v2snp, betas, p.variant_id, p.compressed_cb, p.p_base_wrong,
                                   context_factory=OracleContext)
        probs_local, addition = em.learn(3, 0.01, np.zeros(6, dtype=np.float32), False)
        from oracle import demux_oracle as oracle
        hist = oracle.em(dict(variant_id=p.variant_id, compressed_cb=p.compressed_cb, p_base_wrong=p.p_base_wrong, betas=betas,
                              v2snp=p.v2snp), p.n_barcodes, 3, 0.01, 0.)
        report['sharded_em'] = dict(rows=(em.lo, em.hi),
                                    max_dev=float(np.abs(probs_local - hist[-1]['probs'][em.lo:em.hi]).max()),
                                    add_close=bool(np.allclose(addition, hist[-1]['addition'], rtol=3e-7, atol=1e-12)))
        out.put((rank, report))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_gloo_ranks_drive_the_sharded_entry_points(oracle, world):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(timeout=600)
        assert pr.exitcode == 0
    results = sorted((out.get(timeout=10) for _ in range(world)), key=lambda t: t[0])
    for _rank, report in results:
        for name, r in report.items():
            if name == 'sharded_em':
                assert r['max_dev'] <= 1e-5 and r['add_close'], (name, r)
                continue
            # float64 re-association over ranks can move a rounding tie of a beta by one float32 ulp
            assert r['betas_close'] and r['betas_mismatch'] <= 3, (name, r)
            assert r['argmax_same'] and r['max_dev'] <= 1e-5 and r['predict_bitwise'], (name, r)
    # every rank holds the same answers
    assert all(rep == results[0][1] or {k: v for k, v in rep.items() if k != 'sharded_em'} ==
               {k: v for k, v in results[0][1].items() if k != 'sharded_em'} for _r, rep in results)
