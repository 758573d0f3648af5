"""TEST INFRASTRUCTURE: a CPU stand-in for demuxalot_amd.device.DeviceContext, used to drive the multi-rank host
code (demuxalot_amd/distributed.py) over a gloo process group on a box without GPUs.

The arithmetic is the oracle's.  The multi-rank exchange repeats, step for step and over torch.distributed (or the
caller's collectives), the sequences libdemux_hip.so runs over RCCL (csrc/dmx_api.cpp "Multi-GPU"):
  * default, the M-step sharded on variants (shard_mstep_by_variant): at set-up every rank's calls are all-gathered and
    each rank keeps those of its variant slice (slices from the library's own dmx_exchange_slices) with global barcode
    rows; per iteration the singlet posteriors are all-gathered and every rank sums its slice over all barcodes - one
    float64 sum per entry in the reference's order, so the results equal a single-rank run's bit for bit;
  * `exchange='reduce_scatter'`: per-rank float64 partial sums in the PADDED slice layout -> reduce-scatter (gloo has
    none; it is spelled as one dist.reduce per slice) -> float32 slice;
then in both the P-step on the owned slice -> all-gather of genotype_prob."""
import numpy as np

from demuxalot_amd import _lib
from demuxalot_amd.distributed import exchange_slices
from oracle import demux_oracle as oracle


class _HostCollectives:
    """torch.distributed-shaped adapter over a caller-provided collective (include/demux_hip.h: dmx_host_collective
    semantics, as demuxalot_amd.device hands them to a plane): what OracleContext needs of `dist`."""

    def __init__(self, rank, world, collective):
        self.rank, self.world, self.collective = rank, world, collective

    def all_reduce(self, t):
        self.collective('all_reduce', t.numpy().reshape(-1))

    def reduce(self, t, dst):
        # one slice of a reduce-scatter: stage it as row `dst` of a [world, block] buffer
        buf = np.zeros((self.world, t.numel()), dtype=t.numpy().dtype)
        buf[dst] = t.numpy().reshape(-1)
        self.collective('reduce_scatter', buf)
        if dst == self.rank:
            t.numpy().reshape(-1)[...] = buf[dst]

    def all_gather(self, parts, mine):
        buf = np.zeros((self.world, mine.numel()), dtype=mine.numpy().dtype)
        buf[self.rank] = mine.numpy().reshape(-1)
        self.collective('all_gather', buf)
        for r, part in enumerate(parts):
            part.numpy().reshape(-1)[...] = buf[r]


class _Array:
    """The two tensor methods the exchange code below uses, on a numpy array (keeps the module torch-free when the
    collectives are the caller's)."""

    def __init__(self, array):
        self.array = array

    def numpy(self):
        return self.array

    def numel(self):
        return self.array.size


class OracleContext:
    exchange = 'variant'  # or 'reduce_scatter' (module docstring)

    def __init__(self, device=0):
        self.rank, self.world, self.dist = 0, 1, None
        self.addition = None
        self.torch_free = False
        self.mcalls = None  # variant-sharded M-step: (variant, global barcode row, p_base_wrong) of this rank's slice

    def _gather_padded(self, array, rows):
        """Every rank's `array` (first axis padded to `rows`), as float64 (exact for the integers that travel here)."""
        mine = np.zeros((rows,) + array.shape[1:], dtype=np.float64)
        mine[:len(array)] = array
        parts = [self._tensor(np.zeros_like(mine)) for _ in range(self.world)]
        self.dist.all_gather(parts, self._tensor(mine))
        return [part.numpy() for part in parts]

    def comm_init_host(self, rank, nranks, collective, reduce_dtype='f64'):
        """The exchange over the caller's collectives (a plane with host_collective): no torch anywhere."""
        self.rank, self.world, self.dist = rank, nranks, _HostCollectives(rank, nranks, collective)
        self.torch_free = True
        if getattr(self, 'calls', None) is not None:
            self._layout_exchange()

    def _tensor(self, array):
        if self.torch_free:
            return _Array(array)
        import torch
        return torch.from_numpy(array)

    def _zeros(self, shape):
        return self._tensor(np.zeros(shape, dtype=np.float32))

    def get_prior_betas(self):
        return self.prior

    def set_keep_molecule_calls(self, keep):
        assert not keep, 'the CPU stand-in has no aggregate_on_snps mode'

    @staticmethod
    def new_unique_id():
        return b'cpu-stand-in'.ljust(_lib.UNIQUE_ID_BYTES, b'.')

    def comm_init(self, rank, nranks, unique_id, reduce_dtype='f64'):
        import torch.distributed as dist
        assert len(unique_id) == _lib.UNIQUE_ID_BYTES and unique_id.startswith(b'cpu-stand-in')  # rank 0's bytes arrived
        self.rank, self.world, self.dist = rank, nranks, dist
        if getattr(self, 'calls', None) is not None:
            self._layout_exchange()

    def close(self):
        pass

    # ---- problem ------------------------------------------------------------------------------------------
    def set_problem(self, n_barcodes, n_variants, n_genotypes, variant_id, compressed_cb, p_base_wrong, v2snp):
        self.B, self.V, self.G = n_barcodes, n_variants, n_genotypes
        order = np.lexsort((compressed_cb, variant_id))  # the reference's barcode_calls order
        self.calls = (np.asarray(variant_id)[order], np.asarray(compressed_cb)[order], np.asarray(p_base_wrong, dtype=np.float32)[order])
        self.v2snp = np.asarray(v2snp, dtype=np.int32)
        self._layout_exchange()

    def _layout_exchange(self):
        """Slices and (variant-sharded M-step) the calls of this rank's slice: when a problem is installed with a communicator
        attached, or a communicator is attached to the resident problem (include/demux_hip.h: collective either way)."""
        self.cuts, self.slice_rows, self.sliced = exchange_slices(self.v2snp, self.world)
        self.sliced = self.sliced and self.dist is not None
        self.mcalls = None
        if self.sliced and self.world > 1 and self.exchange == 'variant':
            # set-up of the variant-sharded M-step: sizes, then everybody's calls; this rank keeps its variant slice
            sizes = self._gather_padded(np.array([[self.B, len(self.calls[0])]], dtype=np.float64), 1)
            self.rows_pad = max(1, int(max(part[0, 0] for part in sizes)))
            n_pad = max(1, int(max(part[0, 1] for part in sizes)))
            v, cb, e = self.calls
            wire = np.stack([v.astype(np.float64), cb.astype(np.float64) + self.rank * self.rows_pad, e.astype(np.float64),
                             np.ones(len(v))], axis=1) if len(v) else np.zeros((0, 4))
            lo, hi = self._slice(self.rank)
            keep_v, keep_row, keep_e = [], [], []
            for part in self._gather_padded(wire, n_pad):  # rank-major: ascending global barcode rows inside a variant
                sel = (part[:, 3] == 1) & (part[:, 0] >= lo) & (part[:, 0] < hi)
                keep_v.append(part[sel, 0].astype(np.int64))
                keep_row.append(part[sel, 1].astype(np.int64))
                keep_e.append(part[sel, 2].astype(np.float32))
            mv, mrow, me = np.concatenate(keep_v), np.concatenate(keep_row), np.concatenate(keep_e)
            order = np.argsort(mv, kind='stable')
            self.mcalls = (mv[order], mrow[order], me[order])

    def stage_containers(self, containers):
        """Two-step form of the device pack (DeviceContext.stage_containers): the chromosome numbers are provisional."""
        self._staged = list(containers)

    def pack_staged_and_set_problem(self, n_barcodes, n_genotypes, var_chrom, var_pos, var_base, v2snp, chrom_of_container):
        staged, self._staged = self._staged, None
        final = []
        for k, calls, molecules in staged:
            chrom = int(chrom_of_container[k])
            assert chrom >= 0 or len(calls) == 0, 'calls on a chromosome without variants'
            if chrom >= 0:
                final.append((chrom, calls, molecules))
        return self.pack_containers_and_set_problem(n_barcodes, n_genotypes, var_chrom, var_pos, var_base, v2snp, final)

    def pack_containers_and_set_problem(self, n_barcodes, n_genotypes, var_chrom, var_pos, var_base, v2snp, containers):
        """The host twin of the device pack (dmx_pack_calls_host: product code that needs no GPU)."""
        import ctypes
        chrom = np.concatenate([np.full(len(c), k, dtype=np.int32) for k, c, _m in containers] or [np.zeros(0, np.int32)])
        pos = np.concatenate([c['snp_position'] for _k, c, _m in containers] or [np.zeros(0, np.int32)]).astype(np.int32)
        base = np.concatenate([c['base_index'] for _k, c, _m in containers] or [np.zeros(0, np.uint8)]).astype(np.uint8)
        cb = np.concatenate([m['compressed_cb'][c['molecule_index']] for _k, c, m in containers] or [np.zeros(0, np.int32)]).astype(np.int32)
        p = np.concatenate([c['p_base_wrong'] for _k, c, _m in containers] or [np.zeros(0, np.float32)]).astype(np.float32)
        n, V = len(pos), len(var_pos)
        out_v, out_cb = np.empty(n, np.int32), np.empty(n, np.int32)
        out_p, out_count, mol = np.empty(n, np.float32), np.empty(n, np.int64), np.zeros(V, np.int64)
        n_matched, n_unique = ctypes.c_int64(0), ctypes.c_int64(0)
        as_c = np.ascontiguousarray
        _lib.check(_lib.load().dmx_pack_calls_host(
            V, _lib.ptr(as_c(var_chrom, np.int32)), _lib.ptr(as_c(var_pos, np.int32)), _lib.ptr(as_c(var_base, np.uint8)), n,
            _lib.ptr(as_c(chrom)), _lib.ptr(as_c(pos)), _lib.ptr(as_c(base)), _lib.ptr(as_c(cb)), _lib.ptr(as_c(p)), None,
            ctypes.byref(n_matched), ctypes.byref(n_unique), _lib.ptr(out_v), _lib.ptr(out_cb), _lib.ptr(out_p),
            _lib.ptr(out_count), _lib.ptr(mol)))
        k = n_unique.value
        self.local_mol = mol
        self.set_problem(n_barcodes, V, n_genotypes, out_v[:k], out_cb[:k], out_p[:k], v2snp)
        return n_matched.value, k, mol

    def set_betas(self, betas):
        self.prior = np.asarray(betas, dtype=np.float32)

    def set_prior_betas(self, raw_betas, default_prior, add_data_prior, mol_per_variant=None, fetch=True):
        mol = self.local_mol if mol_per_variant is None else mol_per_variant
        self.prior = oracle.prior_betas(np.asarray(raw_betas, dtype=np.float32), self.v2snp,
                                        np.repeat(np.arange(self.V), mol), default_prior, add_data_prior)
        return self.prior if fetch else None

    def set_addition(self, addition=None):
        self.addition = np.zeros_like(self.prior) if addition is None else np.asarray(addition, dtype=np.float32)

    # ---- steps --------------------------------------------------------------------------------------------
    def _slice(self, r):
        return int(self.cuts[r]), int(self.cuts[r + 1])

    def probs_from_betas(self, p_genotype_clip, fetch=True):
        betas = self.prior + self.addition
        if not self.sliced:
            self.prob = oracle.probs_from_betas(self.v2snp, betas, p_genotype_clip)
            return self.prob
        lo, hi = self._slice(self.rank)  # the P-step of the owned slice only (whole SNP groups)
        mine = np.zeros((self.slice_rows, self.G), dtype=np.float32)
        if hi > lo:
            mine[:hi - lo] = oracle.probs_from_betas(self.v2snp[lo:hi] - self.v2snp[lo], betas[lo:hi], p_genotype_clip)
        parts = [self._zeros(mine.shape) for _ in range(self.world)]
        self.dist.all_gather(parts, self._tensor(mine))
        self.prob = np.concatenate([parts[r].numpy()[:self._slice(r)[1] - self._slice(r)[0]] for r in range(self.world)])
        return self.prob

    def estep(self, penalties, with_doublets, prior_logits=None, fetch_logits=True, fetch_probs=True):
        v, cb, e = self.calls
        # oracle.barcode_logits with the caller's penalties (demux.py:252-263)
        g1, g2 = oracle.option_pairs(self.G, 0.5 if with_doublets else 0.)
        logits = np.zeros([self.B, 1], dtype='float32') + np.asarray(penalties, dtype=np.float32)
        keep, floor = 1 - e, e.clip(1e-4)
        for k, (a, b) in enumerate(zip(g1, g2)):
            col = self.prob[:, a] if a == b else (self.prob[:, a] + self.prob[:, b]) * 0.5
            logits[:, k] = logits[:, k] + np.bincount(cb, weights=np.log(col[v] * keep + floor), minlength=self.B)
        if prior_logits is not None:
            logits += prior_logits
        self.logits, self.post = logits, oracle.softmax_rows(logits)
        return self.logits, self.post

    def mstep(self, contribution_power=2., fetch=True):
        v, cb, e = self.calls
        keep = 1 - e
        if self.mcalls is None:
            part = np.zeros((self.V, self.G))
            for g in range(self.G):
                w = self.post[cb, g] * keep
                w **= contribution_power
                part[:, g] = np.bincount(v, weights=w, minlength=self.V)
        if self.mcalls is not None:
            # variant-sharded: everybody's singlet posteriors, then this rank's slice summed over all barcodes
            posts = self._gather_padded(self.post[:, :self.G].astype(np.float64), self.rows_pad)
            post_all = np.concatenate(posts).astype(np.float32)
            mv, mrow, me = self.mcalls
            lo, hi = self._slice(self.rank)
            self.addition = np.full((self.V, self.G), np.nan, dtype=np.float32)  # foreign slices are NOT current
            keep_m = 1 - me
            for g in range(self.G):
                w = post_all[mrow, g] * keep_m
                w **= contribution_power
                self.addition[lo:hi, g] = np.bincount(mv - lo, weights=w, minlength=hi - lo)
            self.partial = True
            return self._full_addition() if fetch else self.addition
        if self.dist is None:
            self.addition = part.astype(np.float32)
        elif not self.sliced:
            t = self._tensor(part)
            self.dist.all_reduce(t)
            self.addition = t.numpy().astype(np.float32)
        else:
            self.addition = np.full((self.V, self.G), np.nan, dtype=np.float32)  # foreign slices are NOT current
            for r in range(self.world):  # reduce-scatter of the padded slices
                lo, hi = self._slice(r)
                padded = np.zeros((self.slice_rows, self.G))
                padded[:hi - lo] = part[lo:hi]
                t = self._tensor(padded)
                self.dist.reduce(t, dst=r)
                if r == self.rank:
                    self.addition[lo:hi] = t.numpy()[:hi - lo].astype(np.float32)
            self.partial = True
        return self._full_addition() if fetch else self.addition

    def _full_addition(self):
        if self.dist is not None and self.sliced and getattr(self, 'partial', False):
            lo, hi = self._slice(self.rank)
            mine = np.zeros((self.slice_rows, self.G), dtype=np.float32)
            mine[:hi - lo] = self.addition[lo:hi]
            parts = [self._zeros(mine.shape) for _ in range(self.world)]
            self.dist.all_gather(parts, self._tensor(mine))
            self.addition = np.concatenate([parts[r].numpy()[:self._slice(r)[1] - self._slice(r)[0]] for r in range(self.world)])
            self.partial = False
        return self.addition

    def em(self, n_iterations, p_genotype_clip, penalties, with_doublets, prior_logits=None, contribution_power=2.,
           fetch_logits=True, fetch_probs=True, fetch_addition=True):
        self.set_addition(None)
        self.partial = False
        for it in range(n_iterations):
            self.probs_from_betas(p_genotype_clip)
            self.estep(penalties, with_doublets, prior_logits if it == 0 else None)
            if it + 1 < n_iterations:
                self.mstep(contribution_power, fetch=False)
        return self.logits, self.post, self._full_addition()
