"""Multi-rank host logic on CPU: nnz-balanced barcode partition, and - with two gloo processes -
that per-shard M-step partial sums all-reduced over the process group equal the unsharded M-step,
and that per-shard E-step rows equal the rows of the full problem (what the RCCL path relies on).
The per-shard math is done by the oracle (the checker); the product pieces under test are
demuxalot_amd.distributed.partition_barcodes / shard_calls and the unique-id exchange plumbing."""
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


def _worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    from oracle import demux_oracle as oracle
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        p = synth.generate(600, 400, 6, calls_per_barcode=50, seed=3)
        betas = p.prior_betas()
        prob = oracle.probs_from_betas(p.v2snp, betas, 0.01)
        full_logits = oracle.barcode_logits(p.variant_id, p.compressed_cb, p.p_base_wrong, prob, p.n_barcodes, 0.)
        full_post = oracle.softmax_rows(full_logits)
        full_add = oracle.beta_addition(p.variant_id, p.compressed_cb, p.p_base_wrong, full_post, p.n_variants, 6)
        # the unique-id exchange used by bench.py / ShardedEM (bytes made on rank 0 reach every rank)
        box = [b'id-from-rank-0' * 9 + b'xx' if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        assert box[0] == b'id-from-rank-0' * 9 + b'xx' and len(box[0]) == 128
        bounds = partition_barcodes(np.bincount(p.compressed_cb, minlength=p.n_barcodes), world)
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        v, cb, e = shard_calls(p.variant_id, p.compressed_cb, p.p_base_wrong, lo, hi)
        logits = oracle.barcode_logits(v, cb, e, prob, hi - lo, 0.)
        post = oracle.softmax_rows(logits)
        assert np.array_equal(logits, full_logits[lo:hi]) and np.array_equal(post, full_post[lo:hi])
        # float64 partial sums of the shard -> all-reduce -> one float32 rounding (the DMX_F64 mode)
        part = np.zeros((p.n_variants, 6))
        keep = 1 - e
        for g in range(6):
            w = post[cb, g] * keep
            w **= 2.
            part[:, g] = np.bincount(v, weights=w, minlength=p.n_variants)
        t = torch.from_numpy(part)
        dist.all_reduce(t)
        reduced = t.numpy().astype(np.float32)
        out.put((rank, float(np.abs(reduced - full_add).max()), int((reduced != full_add).sum())))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_shards_reproduce_the_unsharded_iteration(oracle):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(timeout=300)
        assert pr.exitcode == 0
    results = sorted(out.get(timeout=10) for _ in range(2))
    for _rank, max_diff, n_diff in results:
        assert max_diff <= 1e-6 and n_diff <= 2  # float64 re-association can move a tie by one float32 ulp
