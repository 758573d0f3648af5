"""Barcode-sharded multi-GPU runs of the Demultiplexer entry points.

The EM path shards on barcodes: E-step rows are independent; the M-step is a sum over calls, hence over barcode
shards; the P-step is a pure function of [V, G] tables.  One process per GPU holds one contiguous barcode range
(balanced by number of calls, not by number of barcodes), the calls of those barcodes with barcode indices re-based
to the range, and a full copy of the beta tables.  The per-iteration exchange happens inside libdemux_hip.so over RCCL
(include/demux_hip.h: "Multi-GPU"): the M-step sharded on variants - all-gather of what it reads of every barcode, every
rank summing its variant slice over all barcodes (additions bit-identical to one GPU) - or, where that would move more
bytes, the reduce-scatter of the per-rank sums; then the P-step on the owned slice and the all-gather of genotype_prob.
What is left for Python:

  * cutting the barcode ranges and filtering the reference's call containers to a range (every rank packs only
    its own barcodes on its GPU; the communicator is attached afterwards, once all ranks agree that their packs worked),
  * three small control-plane exchanges: the RCCL unique id (broadcast), the molecule counts per variant that the
    regularised prior needs (sum over ranks, demux.py:372-388), and the posterior rows (gather),
  * `learn_genotypes` / `predict_posteriors` with the reference's signatures plus a `plane` argument.

The control plane is any object with the four methods of `SingleProcess`.  The one to use is
`demuxalot_amd.plane.SocketControlPlane` (plain TCP, no torch: a worker that imports torch maps torch's own HIP
runtime and RCCL next to the ROCm ones, and the library refuses to attach an RCCL communicator in such a process).
`TorchControlPlane` runs the same methods over an existing torch.distributed (gloo) group for hosts that live in torch
anyway; there the per-iteration exchange must be host-staged (`host_collectives=True`).  A plane that has a
`host_collective(op, array)` method carries the per-iteration exchange itself (staged through host memory) in place
of RCCL: `SocketControlPlane(host_collectives=True)`, or the thread plane of tests/test_gpu_ranks_on_one_gpu.py.

Where the results go (`results=`): 'all' (default, the reference's contract: every rank gets the DataFrame of ALL
barcodes), 'root' (rank 0 gets it - raw buffers over the plane, no pickles - the others None), 'device' (every rank
keeps its rows on its GPU behind a ShardedPosteriors whose assignments / best / option_sums reduce O(B) / O(K)
numbers across ranks: what users take from the matrix, snp_detection.py:166, notebook cells 14 / 19).
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

    def gather_to_root(self, rows):
        return rows

    def all_ok(self, ok, message=''):
        return bool(ok), message

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
        if op == 'all_reduce':
            self._dist.all_reduce(t)
        elif op == 'reduce_scatter':  # gloo has no reduce_scatter: one reduce per destination, block by block
            for r in range(self.world):
                self._dist.reduce(t[r], dst=r)
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

    def _gather(self, rows, everywhere):
        """Rows of every rank concatenated in rank order (= barcode order) as tensors, not pickles; on every rank or
        on rank 0 only."""
        import torch
        rows = np.ascontiguousarray(rows)
        counts = torch.zeros(self.world, dtype=torch.int64)
        counts[self.rank] = rows.shape[0]
        self._dist.all_reduce(counts)
        width = int(np.prod(rows.shape[1:], dtype=np.int64))
        most = int(counts.max())
        mine = torch.zeros((most, width), dtype=torch.from_numpy(rows.reshape(rows.shape[0], width)).dtype)
        mine[:rows.shape[0]] = torch.from_numpy(rows.reshape(rows.shape[0], width))
        if everywhere:
            parts = [torch.empty_like(mine) for _ in range(self.world)]
            self._dist.all_gather(parts, mine)
        else:
            parts = [torch.empty_like(mine) for _ in range(self.world)] if self.rank == 0 else None
            self._dist.gather(mine, parts, dst=0)
            if self.rank != 0:
                return None
        whole = np.concatenate([p.numpy()[:int(n)] for p, n in zip(parts, counts)], axis=0)
        return whole.reshape((whole.shape[0],) + rows.shape[1:])

    def gather_rows(self, rows):
        return self._gather(rows, True)

    def gather_to_root(self, rows):
        return self._gather(rows, False)

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
    """reduce_dtype: 'f64' (default: float64 partial sums on the wire, results independent of the number of ranks up to
    float32 rounding ties), 'f32' (half the bytes), 'auto' (f64 up to 2 ranks, f32 from 4 on)."""
    if plane.world == 1 and not force:
        return
    if reduce_dtype == 'auto':
        reduce_dtype = 'f32' if plane.world >= 4 else 'f64'
    if getattr(plane, 'host_collective', None) is not None:  # the plane brings its own collectives
        ctx.comm_init_host(plane.rank, plane.world, plane.host_collective, reduce_dtype=reduce_dtype)
        return
    make_id = type(ctx).new_unique_id
    unique_id = plane.broadcast_bytes(make_id() if plane.rank == 0 else None)
    ctx.comm_init(plane.rank, plane.world, unique_id, reduce_dtype=reduce_dtype)


# ---------------------------------------------------------------------------------------------------------
# the Demultiplexer entry points, sharded
# ---------------------------------------------------------------------------------------------------------
def _agree(plane, error):
    """Every rank reports whether its local step worked; a failure anywhere raises on EVERY rank (instead of one rank
    raising and the others waiting in the next collective for ever)."""
    all_ok = getattr(plane, 'all_ok', None)
    if all_ok is None:  # a plane without the status exchange (older duck-typed planes): local behaviour
        if error is not None:
            raise error
        return
    ok, message = all_ok(error is None, '' if error is None else f'{type(error).__name__}: {error}')
    if error is not None:
        raise error
    if not ok:
        raise RuntimeError(f'a rank of the sharded run failed: {message}')


def _install_shard(chromosome2compressed_snp_calls, genotypes, barcode_handler, plane, add_data_prior, device,
                   reduce_dtype, context_factory, force_comm, with_communicator=True, keep_molecule_calls=False):
    """Packs this rank's barcodes on its GPU and installs the regularised prior (whose data term needs the
    molecule counts of ALL ranks, demux.py:381-384).  Returns (ctx, lo, hi).
    The ranks agree twice before anybody enters a collective that a failed rank would never join: after the set-up,
    and after the device pack (where e.g. calls on a chromosome without variants assert, on the one shard that holds
    them)."""
    from .demux import _pack_on_device
    from .device import DeviceContext, default_device
    n_barcodes = barcode_handler.n_barcodes
    bounds = partition_barcodes(calls_per_barcode(chromosome2compressed_snp_calls, n_barcodes), plane.world)
    lo, hi = int(bounds[plane.rank]), int(bounds[plane.rank + 1])
    ctx, shard, error = None, None, None
    try:
        shard = shard_containers(chromosome2compressed_snp_calls, lo, hi) if plane.world > 1 else chromosome2compressed_snp_calls
        ctx = (context_factory or DeviceContext)(default_device() if device is None else device)
        if keep_molecule_calls:
            ctx.set_keep_molecule_calls(True)
    except Exception as exc:  # noqa: BLE001 - reported to every rank
        error = exc
    try:
        _agree(plane, error)
        agreed = []

        def reduce_counts(molecules):  # called by the pack between the device pack and the prior
            agreed.append(True)
            _agree(plane, None)
            return plane.sum_int64(molecules)
        share_counts = plane.world > 1 and add_data_prior
        try:
            _pack_on_device(shard, genotypes, hi - lo, add_data_prior, fetch_betas=False, ctx=ctx,
                            reduce_molecule_counts=reduce_counts if share_counts else None)
        except Exception as exc:  # noqa: BLE001
            if not agreed:
                _agree(plane, exc)  # tells the others, then raises exc
            raise
        if not share_counts:
            _agree(plane, None)
        # The communicator is attached to the RESIDENT problem, after every rank has agreed that its pack worked: attaching
        # lays the problem out for the exchange, which is itself collective (the ranks' call records are all-gathered for
        # the variant-sharded M-step) - a rank that failed in its pack would never join it.
        if with_communicator:
            attach_communicator(ctx, plane, reduce_dtype, force_comm)
    except BaseException:
        if ctx is not None:
            ctx.close()
        raise
    return ctx, lo, hi


class ShardedPosteriors:
    """This rank's rows of a sharded run's posteriors, on its GPU, plus the plane: the reductions users apply to the
    [B, K] matrix come back for ALL barcodes while only O(B) / O(K) numbers travel.  Every method is a collective
    (all ranks call it).  `local` is the rank's own DevicePosteriors (rows [lo, hi) of the experiment)."""

    def __init__(self, local, plane, lo, hi, barcodes):
        self.local, self.plane, self.lo, self.hi = local, plane, lo, hi
        self.barcodes = list(barcodes)
        self.columns = local.columns

    @property
    def shape(self):
        return len(self.barcodes), len(self.columns)

    def _index(self, rows=None):
        index = pd.Index(self.barcodes if rows is None else [self.barcodes[i] for i in rows])
        index.name = self.local.index_name
        return index

    def best(self) -> pd.DataFrame:
        best, prob = self.local._ctx.get_assignments()
        best = self.plane.gather_rows(best.astype(np.int32))
        prob = self.plane.gather_rows(prob.astype(np.float32))
        return pd.DataFrame({'option': np.asarray(self.columns, dtype=object)[best], 'probability': prob}, index=self._index())

    def assignments(self, threshold=0.9) -> pd.Series:
        """probs[probs.max(axis=1).gt(threshold)].idxmax(axis=1) over all barcodes (snp_detection.py:166)."""
        best, _prob, _n = self.local._ctx.get_assignments_above(threshold)
        best = self.plane.gather_rows(best.astype(np.int32))
        rows = np.flatnonzero(best >= 0)
        return pd.Series(np.asarray(self.columns, dtype=object)[best[rows]], index=self._index(rows))

    def option_sums(self) -> pd.Series:
        """probs.sum(axis=0) over all barcodes (float64; rank partial sums added in rank order)."""
        sums = self.plane.gather_rows(np.asarray(self.local._ctx.get_option_sums(), dtype=np.float64)[None, :])
        total = sums[0].copy()
        for part in sums[1:]:
            total += part
        return pd.Series(total, index=self.columns)

    def to_dataframe(self, what='probs', root_only=True):
        """The whole matrix after all: on rank 0 (others None), or on every rank."""
        block = self.local._ctx.get_probs() if what == 'probs' else self.local._ctx.get_logits()
        gather = getattr(self.plane, 'gather_to_root', None) if root_only else None
        whole = gather(block) if gather is not None else self.plane.gather_rows(block)
        if whole is None:
            return None
        return pd.DataFrame(whole, index=self._index(), columns=self.columns)

    def close(self):
        self.local.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def _collect(plane, block, results):
    """`block` (this rank's rows) -> all rows on every rank ('all') or on rank 0 ('root': others None)."""
    assert results in ('all', 'root')
    if results == 'root' and getattr(plane, 'gather_to_root', None) is not None:
        return plane.gather_to_root(block)
    return plane.gather_rows(block)


def learn_genotypes(chromosome2compressed_snp_calls, genotypes, barcode_handler, plane, n_iterations=5,
                    p_genotype_clip=0.01, doublet_prior=0., barcode_prior_logits=None, device=None,
                    reduce_dtype='f64', context_factory=None, force_comm=False, results='all'):
    """Demultiplexer.learn_genotypes (demux.py:35-66) over the ranks of `plane`: every rank passes the SAME
    inputs (whole experiment), works on its barcode range, and gets back the same learnt genotypes and - see
    `results` in the module docstring - the posterior DataFrame of ALL barcodes / None / a ShardedPosteriors.
    Every rank must pass the same arguments (the calls inside are collectives)."""
    from .demux import Demultiplexer, DevicePosteriors, _option_names
    assert 0 <= doublet_prior < 1
    assert results in ('all', 'root', 'device')
    penalties = Demultiplexer._doublet_penalties(genotypes.n_genotypes, doublet_prior)
    if barcode_prior_logits is not None:
        assert barcode_prior_logits.shape == (barcode_handler.n_barcodes, len(penalties)), 'wrong shape of priors'
    assert n_iterations >= 1, 'n_iterations should be positive'
    columns = _option_names(genotypes.genotype_names, doublet_prior)
    if Demultiplexer.aggregate_on_snps:  # the staged loop is the implementation (float64 posteriors)
        assert results != 'device', 'device-resident results are float32; aggregate_on_snps yields float64 posteriors'
        *_, (probs_df, last) = staged_genotype_learning(
            chromosome2compressed_snp_calls, genotypes, barcode_handler, plane, n_iterations=n_iterations,
            p_genotype_clip=p_genotype_clip, doublet_prior=doublet_prior, barcode_prior_logits=barcode_prior_logits,
            device=device, reduce_dtype=reduce_dtype, context_factory=context_factory, force_comm=force_comm, results=results)
        return genotypes._with_betas(genotypes.get_betas() + last['genotype_addition']), probs_df
    ctx, lo, hi = _install_shard(chromosome2compressed_snp_calls, genotypes, barcode_handler, plane, True, device,
                                 reduce_dtype, context_factory, force_comm)
    keep = results == 'device'
    try:
        prior = None if barcode_prior_logits is None else np.ascontiguousarray(barcode_prior_logits[lo:hi])
        _logits, probs, addition = ctx.em(
            n_iterations, p_genotype_clip, penalties, with_doublets=doublet_prior != 0, prior_logits=prior,
            contribution_power=Demultiplexer.contribution_power, fetch_logits=False, fetch_probs=not keep)
    except BaseException:
        keep = False
        raise
    finally:
        if not keep:
            ctx.close()
    learnt = genotypes._with_betas(genotypes.get_betas() + addition)
    if results == 'device':
        local = DevicePosteriors(ctx, barcode_handler.ordered_barcodes[lo:hi], columns, pooled=False)
        return learnt, ShardedPosteriors(local, plane, lo, hi, barcode_handler.ordered_barcodes)
    probs = _collect(plane, probs, results)
    probs_df = None if probs is None else pd.DataFrame(data=probs, index=barcode_handler.ordered_barcodes, columns=columns)
    return learnt, probs_df


def staged_genotype_learning(chromosome2compressed_snp_calls, genotypes, barcode_handler, plane, n_iterations=5,
                             p_genotype_clip=0.01, doublet_prior=0., barcode_prior_logits=None, device=None,
                             reduce_dtype='f64', context_factory=None, force_comm=False, results='all'):
    """Demultiplexer.staged_genotype_learning (demux.py:69-118) over the ranks of `plane`: a generator yielding, per
    EM iteration, (posterior DataFrame of all barcodes, {'barcode_logits', 'genotype_prior', 'genotype_addition'}),
    the addition being the one the iteration's E-step used.  Every rank iterates the generator in step (each iteration
    runs the exchange).  results = 'root': the frames / logits only on rank 0 (None elsewhere).
    Honours Demultiplexer.aggregate_on_snps (demux.py:204-244): the float64 partial sums of the ranks are added in rank
    order through the plane (they are [V, G] float64: the wire format of the exchange)."""
    from .demux import Demultiplexer, _option_names
    assert 0 <= doublet_prior < 1
    assert results in ('all', 'root')
    penalties = Demultiplexer._doublet_penalties(genotypes.n_genotypes, doublet_prior)
    if barcode_prior_logits is not None:
        assert barcode_prior_logits.shape == (barcode_handler.n_barcodes, len(penalties)), 'wrong shape of priors'
    aggregate = bool(Demultiplexer.aggregate_on_snps)
    columns = _option_names(genotypes.genotype_names, doublet_prior)
    # aggregate mode: the float64 M-step is single-context; its [V, G] result is summed over ranks on the host
    ctx, lo, hi = _install_shard(chromosome2compressed_snp_calls, genotypes, barcode_handler, plane, True, device,
                                 reduce_dtype, context_factory, force_comm, with_communicator=not aggregate,
                                 keep_molecule_calls=aggregate)
    try:
        prior_betas = ctx.get_prior_betas() if hasattr(ctx, 'get_prior_betas') else None
        addition = None if prior_betas is None else np.zeros_like(prior_betas)
        ctx.set_addition(None)
        local_prior = None if barcode_prior_logits is None else np.ascontiguousarray(barcode_prior_logits[lo:hi])
        for iteration in range(n_iterations):
            ctx.probs_from_betas(p_genotype_clip, fetch=False)
            prior = local_prior if iteration == 0 else None
            if aggregate:
                logits, probs = ctx.estep_snp(doublet_prior != 0, Demultiplexer.compensation_during_computing_barcode_logits,
                                              prior_logits=prior)
            else:
                logits, probs = ctx.estep(penalties, with_doublets=doublet_prior != 0, prior_logits=prior)
            all_probs, all_logits = _collect(plane, probs, results), _collect(plane, logits, results)
            frame = None if all_probs is None else pd.DataFrame(data=all_probs, index=barcode_handler.ordered_barcodes, columns=columns)
            yield frame, {'barcode_logits': all_logits, 'genotype_prior': prior_betas, 'genotype_addition': addition}
            if aggregate:
                partial = ctx.mstep_f64(Demultiplexer.contribution_power, as_float64=True)
                total = sum_float64(plane, partial)
                addition = total.astype(np.float32)
                ctx.set_addition(addition)
            else:
                addition = ctx.mstep(Demultiplexer.contribution_power)  # the exchange + the full addition on every rank
    finally:
        ctx.close()


def sum_float64(plane, array):
    """Sum over ranks of a float64 array, added in rank order on every rank (deterministic)."""
    if plane.world == 1:
        return np.asarray(array, dtype=np.float64)
    parts = plane.gather_rows(np.ascontiguousarray(array, dtype=np.float64).reshape(1, -1))
    total = parts[0].copy()
    for part in parts[1:]:
        total += part
    return total.reshape(np.shape(array))


def predict_posteriors(chromosome2compressed_snp_calls, genotypes, barcode_handler, plane, p_genotype_clip=0.01,
                       doublet_prior=0.35, device=None, context_factory=None, results='all'):
    """Demultiplexer.predict_posteriors (demux.py:120-156) over the ranks of `plane`; needs no data-plane
    collective at all (rows are independent, the P-step is replicated).  `results`: module docstring; 'device'
    returns ONE ShardedPosteriors."""
    from .demux import Demultiplexer, DevicePosteriors, _option_names
    assert results in ('all', 'root', 'device')
    penalties = Demultiplexer._doublet_penalties(genotypes.n_genotypes, doublet_prior)
    aggregate = bool(Demultiplexer.aggregate_on_snps)
    assert not (aggregate and results == 'device'), 'device-resident results are float32; aggregate_on_snps yields float64'
    ctx, lo, hi = _install_shard(chromosome2compressed_snp_calls, genotypes, barcode_handler, plane, False, device,
                                 'f64', context_factory, False, with_communicator=False, keep_molecule_calls=aggregate)
    keep = results == 'device'
    try:
        ctx.set_addition(None)
        genotype_prob = ctx.probs_from_betas(p_genotype_clip)
        assert np.isfinite(genotype_prob).all()
        if aggregate:
            logits, probs = ctx.estep_snp(doublet_prior != 0, Demultiplexer.compensation_during_computing_barcode_logits)
        else:
            logits, probs = ctx.estep(penalties, with_doublets=doublet_prior != 0, fetch_logits=not keep, fetch_probs=not keep)
    except BaseException:
        keep = False
        raise
    finally:
        if not keep:
            ctx.close()
    columns = _option_names(genotypes.genotype_names, doublet_prior)
    if results == 'device':
        local = DevicePosteriors(ctx, barcode_handler.ordered_barcodes[lo:hi], columns, index_name='BARCODE', pooled=False)
        return ShardedPosteriors(local, plane, lo, hi, barcode_handler.ordered_barcodes)
    frames = []
    for block in (_collect(plane, logits, results), _collect(plane, probs, results)):
        if block is None:
            frames.append(None)
            continue
        frame = pd.DataFrame(data=block, index=list(barcode_handler.ordered_barcodes), columns=columns)
        frame.index.name = 'BARCODE'
        frames.append(frame)
    return tuple(frames)
