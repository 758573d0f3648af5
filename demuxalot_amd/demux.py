"""Demultiplexer: the reference's EM front-end (demuxalot/demux.py) over the MI355X kernels.

Same static-method surface, argument meaning, output conventions and assertion behaviour as the
reference class, so `from demuxalot_amd import Demultiplexer` can replace
`from demuxalot import Demultiplexer` behind an unchanged front-end (BAM scanning, genotype
import).  Inputs are duck-typed: the reference's own CompressedSNPCalls / ProbabilisticGenotypes /
BarcodeHandler objects work as well as this package's mirrors.

What runs where
  GPU (rocPRIM sorts/scans + HIP kernels, dmx_pack_and_set_problem): variant matching, de-duplication
             with float32 products in call order, CSR/CSC derivation -> demux.py:276-300, 332-365
             (the public pack_calls(), whose results are host arrays, uses the C++ host twin
             dmx_pack_calls_host, which also works without a GPU)
  GPU:        regularised prior betas (dmx_set_prior_betas)     -> demux.py:367-388
              (numpy twin `_prior_betas` only inside the host-side pack_calls())
  GPU (HIP):  beta -> probability normalisation                -> demux.py:267-274
              per-barcode log-likelihood accumulation + softmax -> demux.py:246-265, :101, :152
              squared-posterior beta update (+ RCCL all-reduce) -> demux.py:113-118
  GPU (HIP):  Demultiplexer.aggregate_on_snps = True: per-(barcode, SNP) regularised E-step over the molecule
              calls and the float64 M-step that follows it                   -> demux.py:204-244
There is no CPU fallback for the GPU steps.
"""
import os
from typing import Dict, Tuple

import numpy as np
import pandas as pd

from . import _lib
from .device import (DeviceContext, acquire_private_context, default_device, get_context, release_private_context,
                     shared_context_lock)

_MOLECULE_CALL_DTYPE = [('variant_id', 'int32'), ('snp_id', 'int32'), ('compressed_cb', 'int32'),
                        ('molecule_id', 'int32'), ('p_base_wrong', 'float32'), ('p_molecule_aligned_wrong', 'float32')]
_BASES = {'A': 0, 'C': 1, 'G': 2, 'T': 3, 'N': 4}
_BASE_TABLE = np.full(256, 255, dtype=np.uint8)
for _letter, _code in _BASES.items():
    _BASE_TABLE[ord(_letter)] = _code


class _Packed:
    """Result of the host-side repack: what the device needs plus what pack_calls returns."""
    __slots__ = ('v2snp', 'betas', 'variant_id', 'compressed_cb', 'p_base_wrong', 'variant_count',
                 'molecule_calls', 'n_molecule_calls')


_UNSET = object()


def _variant_keys(genotypes, columns=_UNSET):
    """var2varid -> per-row key arrays (chromosome index, position, base code) and the chromosome numbering.
    `columns`: variant_columns(genotypes.var2varid) when the caller has them already."""
    from .genotypes import variant_columns
    n_variants = genotypes.n_variants
    if columns is _UNSET:
        columns = variant_columns(genotypes.var2varid)
    base_codes = None
    if columns is not None:
        rows, chrom_codes, chrom_names, pos, bases = columns
        try:  # single-letter bases: one join + a 256-entry table instead of a dict lookup per variant
            letters = np.frombuffer(''.join(bases).encode('ascii'), dtype=np.uint8)
            if len(letters) == len(rows):
                base_codes = _BASE_TABLE[letters]
        except (TypeError, UnicodeEncodeError):
            base_codes = None
    if base_codes is not None and (base_codes < 255).all() and (len(pos) == 0 or pos.max() < 2 ** 31):
        assert len(rows) == n_variants and (len(rows) == 0 or (rows.min() >= 0 and rows.max() < n_variants)), \
            'var2varid rows must enumerate 0..n_variants-1'
        seen = np.zeros(n_variants, dtype=bool)
        seen[rows] = True
        assert seen.all(), 'var2varid rows must enumerate 0..n_variants-1'  # demux.py:317 (a repeated row leaves a gap)
        var_chrom = np.zeros(n_variants, dtype=np.int32)
        var_pos = np.zeros(n_variants, dtype=np.int32)
        var_base = np.zeros(n_variants, dtype=np.uint8)
        var_chrom[rows] = chrom_codes
        var_pos[rows] = pos
        var_base[rows] = base_codes
        return (var_chrom, var_pos, var_base), {name: i for i, name in enumerate(chrom_names)}
    chrom_index = {}
    var_chrom = np.zeros(n_variants, dtype=np.int32)
    var_pos = np.zeros(n_variants, dtype=np.int32)
    var_base = np.zeros(n_variants, dtype=np.uint8)
    seen = np.zeros(n_variants, dtype=bool)
    for (chrom, pos, base), row in genotypes.var2varid.items():
        assert 0 <= row < n_variants and not seen[row], 'var2varid rows must enumerate 0..n_variants-1'
        seen[row] = True
        var_chrom[row] = chrom_index.setdefault(chrom, len(chrom_index))
        var_pos[row] = pos
        var_base[row] = _BASES[base]
    assert seen.all(), 'var2varid rows must enumerate 0..n_variants-1'  # demux.py:317
    return (var_chrom, var_pos, var_base), chrom_index


def _var2varid_fingerprint(var2varid):
    """Identity of a (chrom, pos, base) -> row dict cheap enough to take on every call: the object, its length, its first
    and last entries and ~61 entries at a stride (dicts keep insertion order and the importers only ever append:
    genotypes.py extend_variants; the mutators of demuxalot_amd.ProbabilisticGenotypes drop the kept arrays themselves)."""
    n = len(var2varid)
    if n == 0:
        return id(var2varid), 0
    import itertools
    sample = tuple(itertools.islice(var2varid.items(), 0, None, max(1, n // 61)))  # (a walk at C speed: ~1 ms for 200k entries)
    return id(var2varid), n, next(reversed(var2varid.items())), hash(sample)


def _cached_variant_keys(genotypes):
    """(fingerprint, v2snp, (var_chrom, var_pos, var_base), chromosome -> index) of a genotypes object.  Walking var2varid is
    30 ms for 200k variants - a third of a predict_posteriors call on 78 M calls - and predict -> learn -> predict on the
    same genotypes (examples/2-with-detection-of-new-SNPs.ipynb cells 12 / 16-18, tests/test_synthetic.py:184-190) walked it
    every time: the arrays are kept on the object, keyed by the fingerprint of its var2varid (assigning another dict, or
    adding variants, invalidates them).  A mapping edited in place at unchanged length, first and last entry is not
    detected: set `genotypes._amd_variant_keys = None` after such surgery."""
    from .genotypes import ProbabilisticGenotypes, snp_ids_from_columns, variant_columns
    own_numbering = getattr(type(genotypes), 'get_snp_ids_for_variants', None) is ProbabilisticGenotypes.get_snp_ids_for_variants
    fingerprint = (_var2varid_fingerprint(genotypes.var2varid), genotypes.n_variants, own_numbering)
    cached = getattr(genotypes, '_amd_variant_keys', None)
    if cached is not None and cached[0] == fingerprint:
        return cached
    columns = variant_columns(genotypes.var2varid)  # one walk of var2varid for the SNP numbering and the row keys
    if own_numbering and columns is not None:
        v2snp = snp_ids_from_columns(columns)
    else:
        v2snp = genotypes.get_snp_ids_for_variants()
    assert np.all(v2snp >= 0)
    keys, chrom_index = _variant_keys(genotypes, columns)
    cached = (fingerprint, v2snp, keys, chrom_index)
    try:
        genotypes._amd_variant_keys = cached
    except AttributeError:  # an object that takes no attributes: nothing is kept
        pass
    return cached


def _sampled_checksum(records):
    """crc32 of the head, the tail and ~512 strided records of a record array.  NOT what the resident problem is keyed by any more
    (a sparse in-place edit slips through): kept for DEMUXALOT_AMD_RESIDENT=sampled, the opt-in of callers who never edit their
    containers in place and want the last 15 ms of a repeated call."""
    import zlib
    n = len(records)
    if n == 0:
        return 0
    flat = np.ascontiguousarray(records).view(np.uint8).reshape(-1)
    crc = zlib.crc32(flat[:4096].tobytes())
    crc = zlib.crc32(flat[-4096:].tobytes(), crc)
    return zlib.crc32(np.ascontiguousarray(records[::max(1, n // 512)]).view(np.uint8).tobytes(), crc)


def _content_hash(records):
    """64-bit hash of EVERY byte of a record array (dmx_hash_host: several host threads, memory bandwidth - 2 GB in ~15 ms).
    The reference packs its inputs anew on every call (demux.py:303); a packed problem of an earlier call stands in for that
    only when every record of every container hashes as it did then, whatever object the records live in."""
    import ctypes
    records = np.ascontiguousarray(records)
    out = ctypes.c_uint64(0)
    _lib.check(_lib.load().dmx_hash_host(records.ctypes.data if records.size else None, int(records.nbytes), 0, ctypes.byref(out)))
    return int(out.value)


def resident_policy():
    """DEMUXALOT_AMD_RESIDENT = full (default: reuse keyed by the content hash of every record) | sampled (identity + a
    sampled checksum, round 5's key: in-place edits of single records are NOT seen) | 0 (never reuse: every call packs)."""
    policy = os.environ.get('DEMUXALOT_AMD_RESIDENT', 'full').lower()
    policy = {'1': 'full', 'on': 'full', 'off': '0', 'never': '0', 'none': '0'}.get(policy, policy)
    assert policy in ('full', 'sampled', '0'), f'DEMUXALOT_AMD_RESIDENT={policy!r}: full, sampled or 0'
    return policy


def invalidate_resident(genotypes=None, barcode_handler=None):
    """Forget what the front-end keeps between calls: the packed problem resident on the shared device context(s) and, when
    given, the key arrays kept on a genotypes object and the Index kept on a barcode handler.  The next call packs anew, as
    every call of the reference does."""
    from . import device
    with device.shared_context_lock:
        for ctx in device.shared_contexts():
            ctx._resident_key = None
    if genotypes is not None:
        try:
            genotypes._amd_variant_keys = None
        except AttributeError:
            pass
    if barcode_handler is not None:
        try:
            barcode_handler._amd_index = None
        except AttributeError:
            pass


def _flatten_inputs(chromosome2compressed_snp_calls, genotypes, want_molecule_table):
    """var2varid -> per-row key arrays; per-chromosome call containers -> flat call arrays."""
    (var_chrom, var_pos, var_base), chrom_index = _variant_keys(genotypes)

    parts = []
    n_expected = n_taken = 0
    for chrom, container in chromosome2compressed_snp_calls.items():
        n = container.n_snp_calls
        n_expected += n
        if chrom not in chrom_index:
            continue  # demux.py:339-341: no SNPs on this chromosome (the cursor is not advanced)
        calls = container.snp_calls[:n]
        molecules = container.molecules[:container.n_molecules]
        mol = calls['molecule_index']
        part = dict(chrom=np.full(n, chrom_index[chrom], dtype=np.int32), pos=calls['snp_position'],
                    base=calls['base_index'], cb=molecules['compressed_cb'][mol], p=calls['p_base_wrong'])
        if want_molecule_table:
            part['mol'] = mol
            part['pmis'] = molecules['p_group_misaligned'][mol]
        parts.append(part)
        n_taken += n
    assert n_taken == n_expected  # demux.py:359 (fires when a chromosome with calls has no variants)

    def cat(name, dtype):
        if not parts:
            return np.zeros(0, dtype=dtype)
        return np.ascontiguousarray(np.concatenate([p[name] for p in parts]), dtype=dtype)

    flat = dict(chrom=cat('chrom', np.int32), pos=cat('pos', np.int32), base=cat('base', np.uint8),
                cb=cat('cb', np.int32), p=cat('p', np.float32))
    if want_molecule_table:
        flat['mol'] = cat('mol', np.int32)
        flat['pmis'] = cat('pmis', np.float32)
    return (var_chrom, var_pos, var_base), flat


def _prior_betas(genotypes, v2snp, mol_per_variant, add_data_prior):
    """demux.py:367-388. beta' = beta + f32((1 + [data] n_mol/(n_mol_snp + 100)
    + betasum/(betasum_snp + 100)) * default_prior)[:, None]; float64 per-SNP sums."""
    betas = genotypes.get_betas()
    assert np.all(betas >= 0), 'bad genotypes provided, negative betas appeared'

    def share_within_snp(per_variant, regularization):
        assert len(per_variant) == len(v2snp)
        per_snp = np.bincount(v2snp, weights=per_variant)[v2snp] if len(v2snp) else np.zeros(0)
        return per_variant / (per_snp + regularization)

    scale = 1.
    if add_data_prior:
        scale = scale + share_within_snp(mol_per_variant, 100.)
    scale = scale + share_within_snp(betas.sum(axis=1), 100.)
    addition = scale[:, np.newaxis] * genotypes.default_prior
    out = betas + addition.astype(betas.dtype)
    out.flags.writeable = False
    return out


def _pack(chromosome2compressed_snp_calls, genotypes, add_data_prior, want_molecule_table=False) -> _Packed:
    lib = _lib.load()
    (var_chrom, var_pos, var_base), flat = _flatten_inputs(
        chromosome2compressed_snp_calls, genotypes, want_molecule_table)
    v2snp = genotypes.get_snp_ids_for_variants()
    assert np.all(v2snp >= 0)
    n_variants, n_calls = len(var_pos), len(flat['pos'])

    call_variant = np.empty(n_calls, dtype=np.int32) if want_molecule_table else None
    out_variant = np.empty(n_calls, dtype=np.int32)
    out_cb = np.empty(n_calls, dtype=np.int32)
    out_p = np.empty(n_calls, dtype=np.float32)
    out_count = np.empty(n_calls, dtype=np.int64)
    mol_per_variant = np.zeros(n_variants, dtype=np.int64)
    import ctypes
    n_matched, n_unique = ctypes.c_int64(0), ctypes.c_int64(0)
    _lib.check(lib.dmx_pack_calls_host(
        n_variants, _lib.ptr(var_chrom), _lib.ptr(var_pos), _lib.ptr(var_base),
        n_calls, _lib.ptr(flat['chrom']), _lib.ptr(flat['pos']), _lib.ptr(flat['base']), _lib.ptr(flat['cb']),
        _lib.ptr(flat['p']), _lib.ptr(call_variant), ctypes.byref(n_matched), ctypes.byref(n_unique),
        _lib.ptr(out_variant), _lib.ptr(out_cb), _lib.ptr(out_p), _lib.ptr(out_count), _lib.ptr(mol_per_variant)))
    n = n_unique.value

    packed = _Packed()
    packed.v2snp = v2snp
    packed.variant_id = out_variant[:n].copy()
    packed.compressed_cb = out_cb[:n].copy()
    packed.p_base_wrong = out_p[:n].copy()
    packed.variant_count = out_count[:n].copy()
    packed.n_molecule_calls = n_matched.value
    packed.betas = _prior_betas(genotypes, v2snp, mol_per_variant, add_data_prior)
    packed.molecule_calls = None
    if want_molecule_table:
        keep = call_variant != -1
        table = np.zeros(int(keep.sum()), dtype=_MOLECULE_CALL_DTYPE)
        table['variant_id'] = call_variant[keep]
        table['snp_id'] = v2snp[call_variant[keep]]
        table['compressed_cb'] = flat['cb'][keep]
        table['molecule_id'] = flat['mol'][keep]
        table['p_base_wrong'] = flat['p'][keep]
        table['p_molecule_aligned_wrong'] = flat['pmis'][keep]
        packed.molecule_calls = table
    return packed


def _pack_on_device(chromosome2compressed_snp_calls, genotypes, n_barcodes, add_data_prior, fetch_betas=True, ctx=None,
                    reduce_molecule_counts=None):
    """The repack of predict / learn, on the GPU: flattening of the containers' records, matching +
    de-duplication + layout derivation (dmx_pack_containers_and_set_problem) and the regularised prior betas
    (dmx_set_prior_betas).
    Returns (ctx with the problem and betas resident, regularised prior betas).  `ctx` None = the shared cached
    context (the caller holds shared_context_lock).  `reduce_molecule_counts` (barcode-sharded runs): maps this
    shard's molecule counts per variant to the counts of the whole experiment, which the data term of the
    prior is made of (demux.py:381-384)."""
    from .snp_counter import MOLECULE_DTYPE, SNP_CALL_DTYPE
    if ctx is None:
        ctx = get_context()
    shared = bool(getattr(ctx, '_is_shared', False))  # the process-wide context (its users hold shared_context_lock)
    containers = list(chromosome2compressed_snp_calls.values())
    raw = all(c.snp_calls.dtype == SNP_CALL_DTYPE and c.molecules.dtype == MOLECULE_DTYPE for c in containers)
    # The packed problem stays resident on the shared context: predict_posteriors followed by learn_genotypes on the same
    # containers and genotypes (the reference's own usage pattern, see _cached_variant_keys) packs once.  Key: the content
    # hash of EVERY record of every container (no object identities: an array edited in place hashes differently, a freed
    # id() that comes back means nothing), the fingerprint of var2varid, the shape.  resident_policy(): the switch.
    # The full hashes cost 16 ms for 2 GB of records: on a call whose inputs CANNOT be the resident ones (other shapes, or a sampled
    # checksum that already differs) they are taken in a second thread BEHIND the upload - next to it they competed with the runtime's
    # own host copies for memory bandwidth: 91 instead of 77 ms - while the device packs (13 ms) and computes the prior, and are joined
    # before this function returns (the caller may edit its arrays after that); only a call that looks like a repeat pays for them
    # up front - and saves the 45 ms upload and the 13 ms device pack when they confirm it.
    key = None
    hashes_later = None   # thread computing the full hashes of a call that packs (policy 'full')
    policy = resident_policy() if (shared and raw and reduce_molecule_counts is None) else '0'
    resident = getattr(ctx, '_resident_key', None)
    named = list(chromosome2compressed_snp_calls.items())

    def full_hashes():
        return tuple((_content_hash(c.snp_calls[:c.n_snp_calls]), _content_hash(c.molecules[:c.n_molecules])) for _chrom, c in named)

    if policy == 'full':
        meta = (tuple((chrom, int(c.n_snp_calls), int(c.n_molecules)) for chrom, c in named),
                _var2varid_fingerprint(genotypes.var2varid), genotypes.n_variants, genotypes.n_genotypes, int(n_barcodes),
                bool(getattr(ctx, '_keep_molecule_calls', False)))
        sampled = tuple((_sampled_checksum(c.snp_calls[:c.n_snp_calls]), _sampled_checksum(c.molecules[:c.n_molecules])) for _chrom, c in named)
        if resident is not None and resident[:3] == ('full', meta, sampled):
            key = ('full', meta, sampled, full_hashes())   # looks like a repeat: every record decides
        else:
            import threading
            box = []
            hashes_later = (threading.Thread(target=lambda: box.append(full_hashes())), box, meta, sampled)
    elif policy == 'sampled':
        key = ('sampled', tuple((chrom, id(c.snp_calls), id(c.molecules), int(c.n_snp_calls), int(c.n_molecules),
                                 _sampled_checksum(c.snp_calls[:c.n_snp_calls]), _sampled_checksum(c.molecules[:c.n_molecules]))
                                for chrom, c in named),
               _var2varid_fingerprint(genotypes.var2varid), genotypes.n_variants, genotypes.n_genotypes, int(n_barcodes),
               bool(getattr(ctx, '_keep_molecule_calls', False)))
    if key is not None and resident == key:
        molecules = None  # (the counts per variant are on the device: dmx_set_prior_betas takes them from there)
    elif raw:
        # The containers' packed records go to the GPU as they are and are taken apart there - and they go FIRST, in a
        # second thread (the upload is one foreign call, which releases the interpreter), while this one walks
        # var2varid: 45 ms of upload next to 30 ms of walk for 78.65 M calls and 200 k variants.
        import threading
        items = list(chromosome2compressed_snp_calls.items())
        staged = [(k, container.snp_calls[:container.n_snp_calls], container.molecules[:container.n_molecules])
                  for k, (_chrom, container) in enumerate(items)]
        failure = []

        def stage():
            try:
                ctx.stage_containers(staged)
            except BaseException as exc:  # noqa: BLE001 - re-raised below, on the calling thread
                failure.append(exc)

        worker = threading.Thread(target=stage)
        worker.start()
        try:
            _fp, v2snp, (var_chrom, var_pos, var_base), chrom_index = _cached_variant_keys(genotypes)
        except BaseException:
            worker.join()
            if not failure:  # the worker did stage them (~17 bytes per call on the GPU): give them back before unwinding
                try:
                    ctx.release_problem()
                except Exception:  # noqa: BLE001
                    pass
            raise
        worker.join()
        if failure:
            raise failure[0]
        if hashes_later is not None:
            hashes_later[0].start()   # (the records are on the device: the host's memory system is free for the hashes)
        chrom_of_container = []
        for chrom, container in items:
            if chrom not in chrom_index:  # demux.py:339-341, 359: calls on a chromosome without variants trip the reference's final assert
                assert container.n_snp_calls == 0
            chrom_of_container.append(chrom_index.get(chrom, -1))
        try:
            _m, _u, molecules = ctx.pack_staged_and_set_problem(n_barcodes, genotypes.n_genotypes, var_chrom, var_pos, var_base, v2snp,
                                                                chrom_of_container)
        except BaseException:
            if hashes_later is not None:
                hashes_later[0].join()
            raise
        ctx._resident_key = key   # (policy 'full': set below, once the hashes are in)
    else:
        v2snp = genotypes.get_snp_ids_for_variants()
        assert np.all(v2snp >= 0)
        (var_chrom, var_pos, var_base), flat = _flatten_inputs(chromosome2compressed_snp_calls, genotypes, False)
        _m, _u, molecules = ctx.pack_and_set_problem(n_barcodes, genotypes.n_genotypes, var_chrom, var_pos, var_base, v2snp,
                                                     flat['chrom'], flat['pos'], flat['base'], flat['cb'], flat['p'])
    # regularised prior on the GPU too (a single rank's molecule counts per variant stay on the device)
    if reduce_molecule_counts is not None and add_data_prior:
        molecules = reduce_molecule_counts(molecules)
    else:
        molecules = None
    try:
        betas = ctx.set_prior_betas(genotypes.get_betas(), genotypes.default_prior, add_data_prior,
                                    mol_per_variant=molecules, fetch=fetch_betas)
    finally:
        if hashes_later is not None and hashes_later[0].ident is not None:   # (started: the records went to the device)
            thread, box, meta, sampled = hashes_later
            thread.join()
            ctx._resident_key = ('full', meta, sampled, box[0]) if box else None   # (a hash that failed: nothing is kept)
    return ctx, betas


def _barcode_index(barcode_handler, name=None):
    """pd.Index of barcode_handler.ordered_barcodes (200k strings: ~4 ms to build, per frame), kept on the handler and
    shared by the frames of later calls (an Index is immutable; every frame gets its own shallow copy with its name)."""
    barcodes = barcode_handler.ordered_barcodes
    fingerprint = (id(barcodes), len(barcodes), barcodes[0] if len(barcodes) else None, barcodes[-1] if len(barcodes) else None)
    cached = getattr(barcode_handler, '_amd_index', None)
    if cached is None or cached[0] != fingerprint:
        cached = (fingerprint, pd.Index(list(barcodes)))
        try:
            barcode_handler._amd_index = cached
        except AttributeError:
            pass
    index = cached[1].copy()
    index.name = name
    return index


def _option_names(genotype_names, doublet_prior):
    """Column names: singlets, then 'A+B' for A before B (demux.py:175-191)."""
    names = list(genotype_names)
    if doublet_prior != 0:
        assert doublet_prior > 0
        names = names + [f'{a}+{b}' for i, a in enumerate(genotype_names) for b in genotype_names[i + 1:]]
    return names


class DevicePosteriors:
    """Posteriors (and logits) of one predict_posteriors / learn_genotypes call, kept on the GPU
    (`on_device=True`), with the reductions users of the reference apply to the DataFrame done there:
    nothing of size [B, K] crosses PCIe unless to_dataframes() / rows() ask for it.
    Owns a private device context; close() (or garbage collection) hands it back to a small pool with the problem
    released into the context's block cache - parked, re-used by the next call, and returned to the driver by
    demuxalot_amd.device.trim_device_caches() or by any allocation that would otherwise run out of device memory."""

    def __init__(self, ctx, barcodes, column_names, index_name=None, pooled=True):
        self._ctx = ctx
        self._pooled = pooled  # False: a context set up elsewhere (e.g. with a communicator attached): destroyed on close
        self.barcodes = list(barcodes)
        self.columns = list(column_names)
        self.index_name = index_name

    @property
    def shape(self):
        return len(self.barcodes), len(self.columns)

    def _index(self, barcodes=None):
        index = pd.Index(self.barcodes if barcodes is None else barcodes)
        index.name = self.index_name
        return index

    def assignments(self, threshold=0.9) -> pd.Series:
        """probs[probs.max(axis=1).gt(threshold)].idxmax(axis=1)
        (examples/2-with-detection-of-new-SNPs.ipynb cell 14; snp_detection.py:166)."""
        best, _prob, _n = self._ctx.get_assignments_above(threshold)
        rows = np.flatnonzero(best >= 0)
        names = np.asarray(self.columns, dtype=object)[best[rows]]
        return pd.Series(names, index=self._index([self.barcodes[i] for i in rows]))

    def best(self) -> pd.DataFrame:
        """Per barcode: the most probable option and its posterior (idxmax / max of every row)."""
        best, prob = self._ctx.get_assignments()
        return pd.DataFrame({'option': np.asarray(self.columns, dtype=object)[best], 'probability': prob},
                            index=self._index())

    def top_options(self, k=2) -> pd.DataFrame:
        """The k <= 4 best options per barcode, best first: columns option_1, probability_1, option_2, ..."""
        options, probs = self._ctx.get_top_options(k)
        names = np.asarray(self.columns + [None], dtype=object)  # -1 (row shorter than k) -> None
        data = {}
        for j in range(options.shape[1]):
            data[f'option_{j + 1}'] = names[options[:, j]]
            data[f'probability_{j + 1}'] = probs[:, j]
        return pd.DataFrame(data, index=self._index())

    def option_sums(self) -> pd.Series:
        """probs.sum(axis=0) (`probs[genotype_names].sum()`, same notebook cells 19 / 21), float64."""
        return pd.Series(self._ctx.get_option_sums(), index=self.columns)

    def rows(self, lo, hi, what='probs') -> pd.DataFrame:
        """Rows [lo, hi) of the 'probs' or 'logits' matrix as a DataFrame."""
        block = self._ctx.get_block(what, lo, hi)
        return pd.DataFrame(block, index=self._index(self.barcodes[lo:hi]), columns=self.columns)

    def to_dataframes(self):
        """(logits_df, probs_df) as the host-side entry points return them."""
        index = self._index()
        return (pd.DataFrame(self._ctx.get_logits(), index=index, columns=self.columns),
                pd.DataFrame(self._ctx.get_probs(), index=index.copy(), columns=self.columns))

    def close(self):
        if self._ctx is not None:
            if self._pooled:
                release_private_context(self._ctx)
            else:
                self._ctx.close()
            self._ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class Demultiplexer:
    """
    Demultiplexer that can infer (learn) additional information about genotypes to achieve better quality.
    GPU-backed; see the module docstring for what runs where.
    """
    # same knobs as the reference class (demux.py:30-32)
    contribution_power = 2.
    aggregate_on_snps = False
    compensation_during_computing_barcode_logits = 0.5

    # ------------------------------------------------------------------------------------
    @staticmethod
    def learn_genotypes(chromosome2compressed_snp_calls,
                        genotypes,
                        barcode_handler,
                        n_iterations=5,
                        p_genotype_clip=0.01,
                        doublet_prior=0.,
                        barcode_prior_logits: np.ndarray = None,
                        on_device: bool = False,
                        ) -> Tuple[object, pd.DataFrame]:
        """
        Learn genotypes starting from an initial guess (demux.py:35-66).
        :return: learnt genotypes (a copy of `genotypes` with betas = raw betas + the addition used by the
            last E-step) and the barcode-to-donor posteriors of the last iteration.
        The whole loop runs on the GPU (dmx_em); only the last iteration's results come back.
        :param on_device: (not in the reference) leave the posteriors on the GPU and return a DevicePosteriors
            in place of the DataFrame.
        """
        assert 0 <= doublet_prior < 1
        n_genotypes = genotypes.n_genotypes
        penalties = Demultiplexer._doublet_penalties(n_genotypes, doublet_prior)
        if barcode_prior_logits is not None:
            assert barcode_prior_logits.shape == (barcode_handler.n_barcodes, len(penalties)), 'wrong shape of priors'
        assert n_iterations >= 1, 'n_iterations should be positive'  # the reference fails to unpack an empty run

        column_names = _option_names(genotypes.genotype_names, doublet_prior)
        if Demultiplexer.aggregate_on_snps:  # the staged loop is the implementation (float64 posteriors, no fused driver)
            assert not on_device, 'on_device results are float32; aggregate_on_snps yields float64 posteriors'
            *_, (probs_df, last) = Demultiplexer.staged_genotype_learning(
                chromosome2compressed_snp_calls, genotypes, barcode_handler, n_iterations=n_iterations,
                p_genotype_clip=p_genotype_clip, doublet_prior=doublet_prior, barcode_prior_logits=barcode_prior_logits)
            return genotypes._with_betas(genotypes.get_betas() + last['genotype_addition']), probs_df
        if on_device:
            ctx = acquire_private_context()
            try:
                _pack_on_device(chromosome2compressed_snp_calls, genotypes, barcode_handler.n_barcodes, True,
                                fetch_betas=False, ctx=ctx)
                ctx.em(n_iterations, p_genotype_clip, penalties, with_doublets=doublet_prior != 0,
                       prior_logits=barcode_prior_logits, contribution_power=Demultiplexer.contribution_power,
                       fetch_logits=False, fetch_probs=False, fetch_addition=False)
                learnt_betas = ctx.get_learnt_betas()  # genotypes.get_betas() + addition (demux.py:65), added on the device
            except BaseException:
                ctx.close()
                raise
            learnt_genotypes = genotypes._with_betas(learnt_betas, _take=True)
            return learnt_genotypes, DevicePosteriors(ctx, barcode_handler.ordered_barcodes, column_names)
        with shared_context_lock:
            ctx, _betas = _pack_on_device(chromosome2compressed_snp_calls, genotypes, barcode_handler.n_barcodes, True,
                                          fetch_betas=False)
            # only the last iteration's posteriors go back to the caller (demux.py:65-66): nobody reads its logits
            ctx.set_logits_needed(False)
            try:
                _logits, probs, _addition = ctx.em(
                    n_iterations, p_genotype_clip, penalties, with_doublets=doublet_prior != 0,
                    prior_logits=barcode_prior_logits, contribution_power=Demultiplexer.contribution_power,
                    fetch_logits=False, fetch_addition=False)
                learnt_betas = ctx.get_learnt_betas()  # genotypes.get_betas() + addition (demux.py:65), added on the device
            finally:
                ctx.set_logits_needed(True)
        probs_df = pd.DataFrame(data=probs, index=_barcode_index(barcode_handler), columns=column_names)
        learnt_genotypes = genotypes._with_betas(learnt_betas, _take=True)
        return learnt_genotypes, probs_df

    @staticmethod
    def staged_genotype_learning(chromosome2compressed_snp_calls,
                                 genotypes,
                                 barcode_handler,
                                 n_iterations=5,
                                 p_genotype_clip=0.01,
                                 doublet_prior=0.,
                                 barcode_prior_logits: np.ndarray = None):
        """Generator over EM iterations (demux.py:69-118): yields (posterior DataFrame, debug dict with
        'barcode_logits', 'genotype_prior', 'genotype_addition'), aligned as in the reference: the yielded
        addition is the one the iteration's E-step used.
        The EM state lives on the GPU between yields, in a device context private to this generator: like the
        reference's (pure) generator it is unaffected by other Demultiplexer calls made between iterations."""
        assert 0 <= doublet_prior < 1
        n_genotypes = genotypes.n_genotypes
        penalties = Demultiplexer._doublet_penalties(n_genotypes, doublet_prior)
        if barcode_prior_logits is not None:
            assert barcode_prior_logits.shape == (barcode_handler.n_barcodes, len(penalties)), 'wrong shape of priors'
        aggregate = bool(Demultiplexer.aggregate_on_snps)  # read once, like the other class-level knobs of a run

        ctx = acquire_private_context()
        try:
            ctx.set_keep_molecule_calls(aggregate)
            _ctx, prior_betas = _pack_on_device(chromosome2compressed_snp_calls, genotypes, barcode_handler.n_barcodes,
                                                True, ctx=ctx)
            column_names = _option_names(genotypes.genotype_names, doublet_prior)
            genotype_addition = np.zeros_like(prior_betas)
            ctx.set_addition(None)

            for iteration in range(n_iterations):
                ctx.probs_from_betas(p_genotype_clip, fetch=False)
                prior = barcode_prior_logits if iteration == 0 else None
                if aggregate:  # demux.py:204-244: float64 logits / posteriors, float64 M-step
                    logits, probs = ctx.estep_snp(doublet_prior != 0, Demultiplexer.compensation_during_computing_barcode_logits,
                                                  prior_logits=prior)
                    probs_df = pd.DataFrame(data=probs, index=barcode_handler.ordered_barcodes, columns=column_names)
                    yield probs_df, {
                        'barcode_logits': logits,
                        'genotype_prior': prior_betas,
                        'genotype_addition': genotype_addition,
                    }
                    genotype_addition = ctx.mstep_f64(Demultiplexer.contribution_power)
                    continue
                logits, probs = ctx.estep(penalties, with_doublets=doublet_prior != 0, prior_logits=prior)
                probs_df = pd.DataFrame(data=probs, index=barcode_handler.ordered_barcodes, columns=column_names)
                yield probs_df, {
                    'barcode_logits': logits,
                    'genotype_prior': prior_betas,
                    'genotype_addition': genotype_addition,
                }
                genotype_addition = ctx.mstep(Demultiplexer.contribution_power)
        except GeneratorExit:  # the consumer stopped early: nothing wrong with the context
            release_private_context(ctx)
            raise
        except BaseException:
            release_private_context(ctx, failed=True)
            raise
        else:
            release_private_context(ctx)

    @staticmethod
    def predict_posteriors(chromosome2compressed_snp_calls,
                           genotypes,
                           barcode_handler,
                           p_genotype_clip=0.01,
                           doublet_prior=0.35,
                           on_device: bool = False):
        """One P + E pass (demux.py:120-156). Returns (logits_df, probs_df), rows in
        barcode_handler.ordered_barcodes order, index named 'BARCODE'.
        :param on_device: (not in the reference) keep logits and posteriors on the GPU and return ONE
            DevicePosteriors object (assignments / top options / column sums are then computed there)."""
        penalties = Demultiplexer._doublet_penalties(genotypes.n_genotypes, doublet_prior)
        column_names = _option_names(genotypes.genotype_names, doublet_prior)
        aggregate = bool(Demultiplexer.aggregate_on_snps)
        assert not (aggregate and on_device), 'on_device results are float32; aggregate_on_snps yields float64 posteriors'

        def run(ctx, fetch):
            ctx.set_keep_molecule_calls(aggregate)
            try:
                _pack_on_device(chromosome2compressed_snp_calls, genotypes, barcode_handler.n_barcodes, False,
                                fetch_betas=False, ctx=ctx)
            finally:
                ctx.set_keep_molecule_calls(False)
            ctx.set_addition(None)
            genotype_prob = ctx.probs_from_betas(p_genotype_clip)
            assert np.isfinite(genotype_prob).all()
            if aggregate:  # demux.py:204-244
                return ctx.estep_snp(doublet_prior != 0, Demultiplexer.compensation_during_computing_barcode_logits)
            return ctx.estep(penalties, with_doublets=doublet_prior != 0, fetch_logits=fetch, fetch_probs=fetch)

        if on_device:
            ctx = acquire_private_context()
            try:
                run(ctx, False)
            except BaseException:
                ctx.close()
                raise
            return DevicePosteriors(ctx, barcode_handler.ordered_barcodes, column_names, index_name='BARCODE')
        with shared_context_lock:
            logits, probs = run(get_context(), True)

        logits_df = pd.DataFrame(data=logits, index=_barcode_index(barcode_handler, 'BARCODE'), columns=column_names)
        probs_df = pd.DataFrame(data=probs, index=_barcode_index(barcode_handler, 'BARCODE'), columns=column_names)  # own Index object, shared labels
        return logits_df, probs_df

    # ------------------------------------------------------------------------------------
    @staticmethod
    def _doublet_penalties(n_genotypes: int, doublet_prior: float) -> np.ndarray:
        """Logit offsets of the K options (demux.py:158-173): zero for singlets; for pairs the log-odds
        that makes the prior doublet mass equal `doublet_prior` whatever the number of genotypes."""
        assert 0 <= doublet_prior < 1
        if doublet_prior == 0:
            return np.zeros(n_genotypes, dtype='float32')
        n_pair_slots = n_genotypes * max(n_genotypes - 1, 1) / 2
        bonus = np.log(n_genotypes * doublet_prior) - np.log(n_pair_slots * (1 - doublet_prior))
        penalties = np.zeros(n_genotypes * (n_genotypes + 1) // 2, dtype='float32')
        penalties[n_genotypes:] = bonus
        return penalties

    @staticmethod
    def _iterate_genotypes_options(genotype_names, genotype_prob: np.ndarray, doublet_prior: float):
        """Enumerates the K options as the reference does (demux.py:175-191): (index, column name,
        per-variant probability vector), singlets first, then pairs g1 < g2 row-major with
        (p1 + p2) * 0.5.  Kept for callers that inspect the options; the GPU kernels enumerate the same
        order internally and never call this."""
        k = 0
        for g, name in enumerate(genotype_names):
            yield k, name, genotype_prob[:, g]
            k += 1
        if doublet_prior != 0:
            assert doublet_prior > 0
            for g1, name1 in enumerate(genotype_names):
                for g2 in range(g1 + 1, len(genotype_names)):
                    yield k, f'{name1}+{genotype_names[g2]}', (genotype_prob[:, g1] + genotype_prob[:, g2]) * 0.5
                    k += 1

    @staticmethod
    def compute_barcode_logits(genotype_names, barcode_calls, molecule_calls, doublet_prior: float,
                               genotype_prob: np.ndarray, n_barcodes: int, n_genotypes: int):
        """Dispatcher of demux.py:193-202; with Demultiplexer.aggregate_on_snps the per-(barcode, SNP)
        regularised form of demux.py:204-244 over `molecule_calls` (float64 logits)."""
        if not Demultiplexer.aggregate_on_snps:
            return Demultiplexer.compute_barcode_logits_using_barcode_calls(
                genotype_names, barcode_calls=barcode_calls, doublet_prior=doublet_prior,
                genotype_prob=genotype_prob, n_barcodes=n_barcodes, n_genotypes=n_genotypes)
        genotype_prob = np.asarray(genotype_prob)
        assert genotype_prob.shape[1] == n_genotypes == len(genotype_names)
        n_variants = genotype_prob.shape[0]
        # SNP of every variant, from the molecule calls themselves (snp_id is a function of variant_id)
        v2snp = np.arange(n_variants, dtype=np.int32) + (int(np.max(molecule_calls['snp_id'])) + 1 if len(molecule_calls) else 0)
        v2snp[np.asarray(molecule_calls['variant_id'])] = molecule_calls['snp_id']
        with shared_context_lock:
            ctx = get_context()
            empty_i = np.zeros(0, dtype=np.int32)
            ctx.set_problem(n_barcodes, n_variants, n_genotypes, empty_i, empty_i, np.zeros(0, dtype=np.float32), v2snp)
            ctx.set_probs(genotype_prob)
            ctx.set_molecule_calls(molecule_calls['variant_id'], molecule_calls['compressed_cb'], molecule_calls['p_base_wrong'])
            logits, _ = ctx.estep_snp(doublet_prior != 0, Demultiplexer.compensation_during_computing_barcode_logits)
        return logits, _option_names(genotype_names, doublet_prior)

    @staticmethod
    def compute_barcode_logits_using_barcode_calls(genotype_names, barcode_calls, doublet_prior,
                                                   genotype_prob: np.ndarray, n_barcodes: int, n_genotypes: int):
        """E-step on caller-supplied tables (demux.py:246-265): barcode_calls carries the columns
        'variant_id', 'compressed_cb', 'p_base_wrong'; genotype_prob is float32[V, G]."""
        genotype_prob = np.asarray(genotype_prob)
        assert genotype_prob.shape[1] == n_genotypes == len(genotype_names)
        penalties = Demultiplexer._doublet_penalties(n_genotypes, doublet_prior=doublet_prior)
        with shared_context_lock:
            ctx = get_context()
            ctx.set_problem(n_barcodes, genotype_prob.shape[0], n_genotypes, barcode_calls['variant_id'],
                            barcode_calls['compressed_cb'], barcode_calls['p_base_wrong'],
                            np.zeros(genotype_prob.shape[0], dtype=np.int32))
            ctx.set_probs(genotype_prob)
            logits, _ = ctx.estep(penalties, with_doublets=doublet_prior != 0, fetch_probs=False)
        return logits, _option_names(genotype_names, doublet_prior)

    @staticmethod
    def _compute_probs_from_betas(variant_index2snp_index, variant_index2betas, p_genotype_clip):
        """P-step on caller-supplied tables (demux.py:267-274).  float32 betas take the EM path's kernel; any
        other dtype is carried as float64, as numpy's own promotion does (float64 division, one rounding)."""
        betas = np.asarray(variant_index2betas)
        as_f32 = betas.dtype == np.float32
        betas = np.ascontiguousarray(betas, dtype=np.float32 if as_f32 else np.float64)
        empty_i = np.zeros(0, dtype=np.int32)
        with shared_context_lock:
            ctx = get_context()
            ctx.set_problem(0, betas.shape[0], betas.shape[1], empty_i, empty_i, np.zeros(0, dtype=np.float32),
                            variant_index2snp_index)
            if not as_f32:
                return ctx.probs_from_betas_f64(betas, p_genotype_clip)
            ctx.set_betas(betas)
            ctx.set_addition(None)
            return ctx.probs_from_betas(p_genotype_clip)

    @staticmethod
    def molecule_calls2barcode_calls(molecule_calls, _prepacked=None):
        """Unique (variant, barcode) calls from matched molecule calls (demux.py:276-300), as the
        reference's record array (variant-major order)."""
        variant_id = np.ascontiguousarray(molecule_calls['variant_id'], dtype=np.int32)
        snp_id = np.ascontiguousarray(molecule_calls['snp_id'], dtype=np.int32)
        cb = np.ascontiguousarray(molecule_calls['compressed_cb'], dtype=np.int32)
        p = np.ascontiguousarray(molecule_calls['p_base_wrong'], dtype=np.float32)
        if _prepacked is None:
            # route the already-matched calls through the same host packer: key = variant row
            import ctypes
            lib = _lib.load()
            n = len(variant_id)
            n_variants = int(variant_id.max()) + 1 if n else 0
            rows = np.arange(n_variants, dtype=np.int32)
            zeros_v = np.zeros(n_variants, dtype=np.int32)
            zeros_b = np.zeros(n_variants, dtype=np.uint8)
            out_v, out_cb = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.int32)
            out_p, out_count = np.empty(n, dtype=np.float32), np.empty(n, dtype=np.int64)
            n_matched, n_unique = ctypes.c_int64(0), ctypes.c_int64(0)
            _lib.check(lib.dmx_pack_calls_host(
                n_variants, _lib.ptr(zeros_v), _lib.ptr(rows), _lib.ptr(zeros_b),
                n, _lib.ptr(np.zeros(n, dtype=np.int32)), _lib.ptr(variant_id), _lib.ptr(np.zeros(n, dtype=np.uint8)),
                _lib.ptr(cb), _lib.ptr(p), None, ctypes.byref(n_matched), ctypes.byref(n_unique),
                _lib.ptr(out_v), _lib.ptr(out_cb), _lib.ptr(out_p), _lib.ptr(out_count), None))
            k = n_unique.value
            u_variant, u_cb, u_p, u_count = out_v[:k].copy(), out_cb[:k].copy(), out_p[:k].copy(), out_count[:k].copy()
        else:
            u_variant, u_cb, u_p, u_count = _prepacked
        # snp id of a unique call: snp_id is a function of variant_id
        variant2snp = np.zeros(int(variant_id.max()) + 1 if len(variant_id) else 0, dtype=np.int32)
        variant2snp[variant_id] = snp_id
        u_snp = variant2snp[u_variant] if len(u_variant) else np.zeros(0, dtype=np.int32)
        # how many molecules of the barcode hit the SNP (any variant of it); nothing downstream reads it
        if len(u_variant):
            key = u_snp.astype(np.int64) * (int(u_cb.max()) + 1) + u_cb
            _, inverse = np.unique(key, return_inverse=True)
            snp_count = np.bincount(inverse, u_count)[inverse]
        else:
            snp_count = np.zeros(0, dtype=np.float64)
        return np.rec.fromarrays(
            [u_variant, u_snp, u_cb, u_p, u_count, snp_count],
            names=['variant_id', 'snp_id', 'compressed_cb', 'p_base_wrong', 'barcode_variant_count',
                   'barcode_snp_count'])

    @staticmethod
    def pack_calls(chromosome2compressed_snp_calls, genotypes, add_data_prior: bool):
        """demux.py:303-392: returns (variant_index2snp_index, regularised betas (read-only),
        matched molecule calls, unique barcode calls)."""
        packed = _pack(chromosome2compressed_snp_calls, genotypes, add_data_prior, want_molecule_table=True)
        barcode_calls = Demultiplexer.molecule_calls2barcode_calls(
            packed.molecule_calls,
            _prepacked=(packed.variant_id, packed.compressed_cb, packed.p_base_wrong, packed.variant_count))
        return packed.v2snp, packed.betas, packed.molecule_calls, barcode_calls
