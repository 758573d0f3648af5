"""Barcode sharding for multi-GPU runs.

The EM path shards on barcodes: E-step rows are independent; the M-step is a sum over calls, hence
over barcode shards; the P-step is a pure function of [V, G] tables and is replicated.  One process
per GPU holds one contiguous barcode range (balanced by number of calls, not by number of
barcodes), its calls with barcode indices re-based to the range, and a full copy of the genotype
tables.  The single exchange per EM iteration - the all-reduce of the beta addition - happens inside
libdemux_hip.so over RCCL (dmx_comm_init / dmx_mstep); this module only cuts the ranges and carries
the RCCL unique id through whatever control plane the launcher provides.
"""
import numpy as np


def partition_barcodes(calls_per_barcode, n_ranks):
    """Contiguous ranges [lo, hi) per rank with (nearly) equal numbers of calls.
    Returns an int64 array of n_ranks + 1 boundaries (first 0, last n_barcodes)."""
    calls_per_barcode = np.asarray(calls_per_barcode, dtype=np.int64)
    n_barcodes = len(calls_per_barcode)
    assert n_ranks >= 1
    prefix = np.concatenate([[0], np.cumsum(calls_per_barcode)])
    targets = prefix[-1] * np.arange(1, n_ranks) / n_ranks
    cuts = np.searchsorted(prefix, targets, side='left')
    bounds = np.concatenate([[0], cuts, [n_barcodes]]).astype(np.int64)
    return np.maximum.accumulate(np.clip(bounds, 0, n_barcodes))


def shard_calls(variant_id, compressed_cb, p_base_wrong, lo, hi):
    """Calls of barcodes [lo, hi), order preserved, barcode indices re-based to 0."""
    compressed_cb = np.asarray(compressed_cb)
    keep = (compressed_cb >= lo) & (compressed_cb < hi)
    return (np.ascontiguousarray(variant_id[keep], dtype=np.int32),
            np.ascontiguousarray(compressed_cb[keep] - lo, dtype=np.int32),
            np.ascontiguousarray(p_base_wrong[keep], dtype=np.float32))


class ShardedEM:
    """One rank's view of a barcode-sharded EM run.

        em = ShardedEM(rank, world, n_barcodes, v2snp, prior_betas, variant_id, compressed_cb, p_base_wrong,
                       exchange_id=lambda make: broadcast(make() if rank == 0 else None))
        probs_local, addition = em.learn(n_iterations, p_clip, penalties, with_doublets)

    `exchange_id(make)` must return, on every rank, the bytes produced by `make()` on rank 0
    (e.g. via torch.distributed.broadcast_object_list over gloo, MPI, or a file)."""

    def __init__(self, rank, world, n_barcodes, v2snp, prior_betas, variant_id, compressed_cb, p_base_wrong,
                 exchange_id=None, device=None, reduce_dtype='f64'):
        from .device import DeviceContext, default_device
        self.rank, self.world = int(rank), int(world)
        counts = np.bincount(compressed_cb, minlength=n_barcodes)
        self.bounds = partition_barcodes(counts, self.world)
        self.lo, self.hi = int(self.bounds[self.rank]), int(self.bounds[self.rank + 1])
        v, cb, e = shard_calls(variant_id, compressed_cb, p_base_wrong, self.lo, self.hi)
        self.ctx = DeviceContext(default_device() if device is None else device)
        self.ctx.set_problem(self.hi - self.lo, len(v2snp), prior_betas.shape[1], v, cb, e, v2snp)
        self.ctx.set_betas(prior_betas)
        if self.world > 1:
            assert exchange_id is not None, 'multi-rank runs need a way to share the RCCL unique id'
            unique_id = exchange_id(DeviceContext.new_unique_id)
            self.ctx.comm_init(self.rank, self.world, unique_id, reduce_dtype=reduce_dtype)

    def learn(self, n_iterations, p_genotype_clip, penalties, with_doublets, prior_logits_local=None,
              contribution_power=2.):
        """Runs the EM loop; returns this rank's posterior rows [hi-lo, K] and the (global, identical on
        every rank) beta addition used by the last E-step."""
        _logits, probs, addition = self.ctx.em(
            n_iterations, p_genotype_clip, penalties, with_doublets, prior_logits=prior_logits_local,
            contribution_power=contribution_power, fetch_logits=False)
        return probs, addition
