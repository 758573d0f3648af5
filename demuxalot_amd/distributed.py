"""Barcode-sharded multi-GPU runs of the Demultiplexer entry points.

The EM path shards on barcodes: E-step rows are independent; the M-step is a sum over calls, hence over barcode
shards; the P-step is a pure function of [V, G] tables.  One process per GPU holds one contiguous barcode range
(balanced by number of calls, not by number of barcodes), the calls of those barcodes with barcode indices re-based
to the range, and a full copy of the beta tables.  The per-iteration exchange -- reduce-scatter of the partial beta
additions over variant slices, P-step on the owned slice, all-gather of genotype_prob -- happens inside
libdemux_hip.so over RCCL (include/demux_hip.h: "Multi-GPU").  What is left for Python:

  * cutting the barcode ranges and filtering the reference's call containers to a range (every rank packs only
    its own barcodes on its GPU),
  * three small control-plane exchanges: the RCCL unique id (broadcast), the molecule counts per variant that the
    regularised prior needs (sum over ranks, demux.py:372-388), and the posterior rows (gather),
  * `learn_genotypes` / `predict_posteriors` with the reference's signatures plus a `plane` argument.

The control plane is any object with the four methods of `SingleProcess`; `TorchControlPlane` runs them over a
torch.distributed process group (gloo on the host; the data plane never touches it).  A plane that also has a
`host_collective(op, array)` method carries the per-iteration exchange itself (staged through host memory) in place
of RCCL: `TorchControlPlane(host_collectives=True)`, or the thread plane of tests/test_gpu_ranks_on_one_gpu.py.
"""
import numpy as np
import pandas as pd


# ---------------------------------------------------------------------------------------------------------
# control plane
# ---------------------------------------------------------------------------------------------------------
class SingleProcess:
    """The degenerate control plane of a one-rank run."""
    rank, world = 0, 1

    def broadcast_bytes(self, payload):
        return payload

    def sum_int64(self, array):
        return np.asarray(array, dtype=np.int64)

    def gather_rows(self, rows):
        return rows

    def barrier(self):
        pass


class TorchControlPlane:
    """Control plane over the default torch.distributed process group (CPU tensors: use a gloo group).
    With host_collectives=True the per-iteration exchange of the library runs over this group as well (staged through
    host memory, include/demux_hip.h: dmx_comm_init_host) instead of RCCL - for hosts without a usable RCCL fabric."""

    def __init__(self, host_collectives=False):
        import torch.distributed as dist
        assert dist.is_initialized(), 'torch.distributed.init_process_group first'
        self._dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        if host_collectives:
            self.host_collective = self._host_collective

    def _host_collective(self, op, array):
        import torch
        t = torch.from_numpy(array)  # shares the library's staging buffer
        if op in ('all_reduce', 'reduce_scatter'):  # gloo has no reduce_scatter: row `rank` of the full sum is it
            self._dist.all_reduce(t)
        else:
            parts = [torch.empty_like(t[0]) for _ in range(self.world)]
            self._dist.all_gather(parts, t[self.rank].clone())
            for r, part in enumerate(parts):
                t[r] = part

    def broadcast_bytes(self, payload):
        box = [payload if self.rank == 0 else None]
        self._dist.broadcast_object_list(box, src=0)
        return box[0]

    def sum_int64(self, array):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(array, dtype=np.int64).copy())
        self._dist.all_reduce(t)
        return t.numpy()

    def gather_rows(self, rows):
        """Rows of every rank, concatenated in rank order (= barcode order), on every rank."""
        parts = [None] * self.world
        self._dist.all_gather_object(parts, np.ascontiguousarray(rows))
        return np.concatenate(parts, axis=0)

    def barrier(self):
        self._dist.barrier()


# ---------------------------------------------------------------------------------------------------------
# sharding of the inputs
# ---------------------------------------------------------------------------------------------------------
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


def calls_per_barcode(chromosome2compressed_snp_calls, n_barcodes):
    """Molecule calls per barcode over all chromosomes (what the shards are balanced by)."""
    counts = np.zeros(n_barcodes, dtype=np.int64)
    for container in chromosome2compressed_snp_calls.values():
        calls = container.snp_calls[:container.n_snp_calls]
        cb = container.molecules['compressed_cb'][:container.n_molecules][calls['molecule_index']]
        counts += np.bincount(cb, minlength=n_barcodes)
    return counts


class _ShardContainer:
    """A CompressedSNPCalls-shaped view of the calls of one barcode range (duck-typed: what the pack reads)."""

    def __init__(self, snp_calls, molecules):
        self.snp_calls, self.molecules = snp_calls, molecules
        self.n_snp_calls, self.n_molecules = len(snp_calls), len(molecules)


def shard_containers(chromosome2compressed_snp_calls, lo, hi):
    """The containers restricted to the molecules of barcodes [lo, hi), barcode indices re-based to the range.
    Chromosome order, call order and molecule numbering are kept, so the packed calls of the shard are the
    shard's rows of the packed calls of the whole experiment."""
    out = {}
    for chrom, container in chromosome2compressed_snp_calls.items():
        calls = container.snp_calls[:container.n_snp_calls]
        molecules = np.array(container.molecules[:container.n_molecules])  # copy: re-based below
        inside = (molecules['compressed_cb'] >= lo) & (molecules['compressed_cb'] < hi)
        molecules['compressed_cb'] = np.where(inside, molecules['compressed_cb'] - lo, 0)
        out[chrom] = _ShardContainer(np.ascontiguousarray(calls[inside[calls['molecule_index']]]), molecules)
    return out


def exchange_slices(v2snp, n_ranks):
    """The variant slices of the multi-GPU exchange as libdemux_hip.so cuts them (dmx_exchange_slices):
    (cuts int64[n_ranks + 1], rows per padded slice, SNP groups contiguous?)."""
    import ctypes
    from . import _lib
    v2snp = np.ascontiguousarray(v2snp, dtype=np.int32)
    cuts = np.zeros(n_ranks + 1, dtype=np.int64)
    rows, contiguous = ctypes.c_int64(0), ctypes.c_int(0)
    _lib.check(_lib.load().dmx_exchange_slices(len(v2snp), _lib.ptr(v2snp), int(n_ranks), _lib.ptr(cuts),
                                               ctypes.byref(rows), ctypes.byref(contiguous)))
    return cuts, rows.value, bool(contiguous.value)


# ---------------------------------------------------------------------------------------------------------
# one rank's EM on already packed calls
# ---------------------------------------------------------------------------------------------------------
class ShardedEM:
    """One rank's view of a barcode-sharded EM run on packed calls (the reference's `barcode_calls` columns).

        em = ShardedEM(plane, n_barcodes, v2snp, prior_betas, variant_id, compressed_cb, p_base_wrong)
        probs_local, addition = em.learn(n_iterations, p_clip, penalties, with_doublets)

    `context_factory(device)` builds the device context (tests substitute a CPU stand-in); `force_comm` attaches a
    communicator even with one rank (exercises the collective path on a one-GPU box)."""

    def __init__(self, plane, n_barcodes, v2snp, prior_betas, variant_id, compressed_cb, p_base_wrong,
                 device=None, reduce_dtype='f64', context_factory=None, force_comm=False):
        from .device import DeviceContext, default_device
        self.plane = plane
        counts = np.bincount(compressed_cb, minlength=n_barcodes)
        self.bounds = partition_barcodes(counts, plane.world)
        self.lo, self.hi = int(self.bounds[plane.rank]), int(self.bounds[plane.rank + 1])
        v, cb, e = shard_calls(variant_id, compressed_cb, p_base_wrong, self.lo, self.hi)
        make = context_factory or DeviceContext
        self.ctx = make(default_device() if device is None else device)
        attach_communicator(self.ctx, plane, reduce_dtype, force_comm)
        self.ctx.set_problem(self.hi - self.lo, len(v2snp), prior_betas.shape[1], v, cb, e, v2snp)
        self.ctx.set_betas(prior_betas)

    def learn(self, n_iterations, p_genotype_clip, penalties, with_doublets, prior_logits_local=None,
              contribution_power=2.):
        """Runs the EM loop; returns this rank's posterior rows [hi-lo, K] and the (global, identical on
        every rank) beta addition used by the last E-step."""
        _logits, probs, addition = self.ctx.em(
            n_iterations, p_genotype_clip, penalties, with_doublets, prior_logits=prior_logits_local,
            contribution_power=contribution_power, fetch_logits=False)
        return probs, addition


def attach_communicator(ctx, plane, reduce_dtype='f64', force=False):
    if plane.world == 1 and not force:
        return
    if getattr(plane, 'host_collective', None) is not None:  # the plane brings its own collectives
        ctx.comm_init_host(plane.rank, plane.world, plane.host_collective, reduce_dtype=reduce_dtype)
        return
    make_id = type(ctx).new_unique_id
    unique_id = plane.broadcast_bytes(make_id() if plane.rank == 0 else None)
    ctx.comm_init(plane.rank, plane.world, unique_id, reduce_dtype=reduce_dtype)


# ---------------------------------------------------------------------------------------------------------
# the Demultiplexer entry points, sharded
# ---------------------------------------------------------------------------------------------------------
def _install_shard(chromosome2compressed_snp_calls, genotypes, barcode_handler, plane, add_data_prior, device,
                   reduce_dtype, context_factory, force_comm):
    """Packs this rank's barcodes on its GPU and installs the regularised prior (whose data term needs the
    molecule counts of ALL ranks, demux.py:381-384).  Returns (ctx, lo, hi)."""
    from .demux import _pack_on_device
    from .device import DeviceContext, default_device
    n_barcodes = barcode_handler.n_barcodes
    bounds = partition_barcodes(calls_per_barcode(chromosome2compressed_snp_calls, n_barcodes), plane.world)
    lo, hi = int(bounds[plane.rank]), int(bounds[plane.rank + 1])
    shard = shard_containers(chromosome2compressed_snp_calls, lo, hi) if plane.world > 1 else chromosome2compressed_snp_calls
    ctx = (context_factory or DeviceContext)(default_device() if device is None else device)
    try:
        attach_communicator(ctx, plane, reduce_dtype, force_comm)
        _pack_on_device(shard, genotypes, hi - lo, add_data_prior, fetch_betas=False, ctx=ctx,
                        reduce_molecule_counts=plane.sum_int64 if plane.world > 1 else None)
    except BaseException:
        ctx.close()
        raise
    return ctx, lo, hi


def learn_genotypes(chromosome2compressed_snp_calls, genotypes, barcode_handler, plane, n_iterations=5,
                    p_genotype_clip=0.01, doublet_prior=0., barcode_prior_logits=None, device=None,
                    reduce_dtype='f64', context_factory=None, force_comm=False):
    """Demultiplexer.learn_genotypes (demux.py:35-66) over the ranks of `plane`: every rank passes the SAME
    inputs (whole experiment), works on its barcode range, and gets back the same learnt genotypes and the
    posterior DataFrame of ALL barcodes."""
    from .demux import Demultiplexer, _option_names
    assert 0 <= doublet_prior < 1
    penalties = Demultiplexer._doublet_penalties(genotypes.n_genotypes, doublet_prior)
    if barcode_prior_logits is not None:
        assert barcode_prior_logits.shape == (barcode_handler.n_barcodes, len(penalties)), 'wrong shape of priors'
    assert n_iterations >= 1, 'n_iterations should be positive'
    ctx, lo, hi = _install_shard(chromosome2compressed_snp_calls, genotypes, barcode_handler, plane, True, device,
                                 reduce_dtype, context_factory, force_comm)
    try:
        prior = None if barcode_prior_logits is None else np.ascontiguousarray(barcode_prior_logits[lo:hi])
        _logits, probs, addition = ctx.em(
            n_iterations, p_genotype_clip, penalties, with_doublets=doublet_prior != 0, prior_logits=prior,
            contribution_power=Demultiplexer.contribution_power, fetch_logits=False)
    finally:
        ctx.close()
    probs = plane.gather_rows(probs)
    probs_df = pd.DataFrame(data=probs, index=barcode_handler.ordered_barcodes,
                            columns=_option_names(genotypes.genotype_names, doublet_prior))
    return genotypes._with_betas(genotypes.get_betas() + addition), probs_df


def predict_posteriors(chromosome2compressed_snp_calls, genotypes, barcode_handler, plane, p_genotype_clip=0.01,
                       doublet_prior=0.35, device=None, context_factory=None):
    """Demultiplexer.predict_posteriors (demux.py:120-156) over the ranks of `plane`; needs no data-plane
    collective at all (rows are independent, the P-step is replicated)."""
    from .demux import Demultiplexer, _option_names
    from .device import DeviceContext, default_device
    from .demux import _pack_on_device
    penalties = Demultiplexer._doublet_penalties(genotypes.n_genotypes, doublet_prior)
    n_barcodes = barcode_handler.n_barcodes
    bounds = partition_barcodes(calls_per_barcode(chromosome2compressed_snp_calls, n_barcodes), plane.world)
    lo, hi = int(bounds[plane.rank]), int(bounds[plane.rank + 1])
    shard = shard_containers(chromosome2compressed_snp_calls, lo, hi) if plane.world > 1 else chromosome2compressed_snp_calls
    ctx = (context_factory or DeviceContext)(default_device() if device is None else device)
    try:
        _pack_on_device(shard, genotypes, hi - lo, False, fetch_betas=False, ctx=ctx)
        ctx.set_addition(None)
        genotype_prob = ctx.probs_from_betas(p_genotype_clip)
        assert np.isfinite(genotype_prob).all()
        logits, probs = ctx.estep(penalties, with_doublets=doublet_prior != 0)
    finally:
        ctx.close()
    columns = _option_names(genotypes.genotype_names, doublet_prior)
    frames = []
    for block in (plane.gather_rows(logits), plane.gather_rows(probs)):
        frame = pd.DataFrame(data=block, index=list(barcode_handler.ordered_barcodes), columns=columns)
        frame.index.name = 'BARCODE'
        frames.append(frame)
    return tuple(frames)
