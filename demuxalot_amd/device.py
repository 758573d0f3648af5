"""Thin object wrapper over the dmx_ctx of libdemux_hip.so: one context = one GPU + one HIP
stream + one resident problem (calls CSR/CSC, beta tables, logits/posteriors)."""
import ctypes
import os
import threading

import numpy as np

from . import _lib
from ._lib import DMX_F32, DMX_F64, as_c, check, ptr


# E-step arithmetic of a new context (include/demux_hip.h: dmx_set_estep_mode); DEMUXALOT_AMD_ESTEP overrides.
# 'guarded': the contract of the path - assignments identical to the reference, posteriors within 1e-5 - PROVEN per barcode and
# E-step, every barcode that cannot be proven redone with the bit-exact kernel; E-steps that take the dictionary form (large
# singlet runs on the importers' genotype tables: predict_posteriors, EM iteration 0) stay bit-identical to the reference.
# 'exact': every logit, posterior and beta addition bit-identical to the reference, at 1.6x the time per EM iteration.
DEFAULT_ESTEP_MODE = 'guarded'


class DeviceContext:
    def __init__(self, device=0):
        self._lib = _lib.load()
        handle = ctypes.c_void_p()
        check(self._lib.dmx_create(int(device), ctypes.byref(handle)))
        self._h = handle
        self.device = int(device)
        self.B = self.V = self.G = self.N = 0
        self.K = 0
        self._resident_key = None
        self._keep_molecule_calls = False
        self.apply_environment()

    def apply_environment(self):
        """The modes the environment selects (also re-applied when a pooled context is handed out again):
        DEMUXALOT_AMD_ESTEP = exact | guarded | fast (dmx_set_estep_mode); DEMUXALOT_AMD_EXACT_ADDITIONS = 1 | 0: the
        bit-identical genotype additions of the hottest variants (several work items) cost ~4 % per EM iteration
        (include/demux_hip.h: dmx_set_exact_additions; default: with the exact E-step); DEMUXALOT_AMD_ESTEP_SCHEDULE = direct | auto | tiled
        (dmx_set_estep_schedule); DEMUXALOT_AMD_ESTEP_DICT = never | auto | always (dmx_set_estep_dictionary);
        DEMUXALOT_AMD_ESTEP_PACKED = never | auto | always (dmx_set_estep_packing); DEMUXALOT_AMD_COARSE_PASS = 1 | 0: the guarded
        mode's binary16 pass for E-steps whose logits nobody reads (dmx_set_coarse_pass; default on); DEMUXALOT_AMD_MSTEP_INCREMENTAL
        = 1 | 0: the tile-major M-step keeps its integer sums and adds differences (dmx_set_mstep_incremental; default on);
        DEMUXALOT_AMD_LEAN = 0 | 1: release the fine pass's copy of the E-step records once the coarse pass's are built (dmx_set_lean_memory)."""
        mode = os.environ.get('DEMUXALOT_AMD_ESTEP', '') or DEFAULT_ESTEP_MODE
        assert mode in ('exact', 'fast', 'guarded'), f'DEMUXALOT_AMD_ESTEP={mode!r}: exact, fast or guarded'
        self.set_estep_mode(mode)
        # the M-step's exact summation keeps the additions bit-identical to the reference GIVEN bit-identical posteriors:
        # it goes with the exact E-step unless asked for
        exact_additions = os.environ.get('DEMUXALOT_AMD_EXACT_ADDITIONS', '')
        self.set_exact_additions(mode == 'exact' if exact_additions == '' else exact_additions != '0')
        self.set_estep_schedule(os.environ.get('DEMUXALOT_AMD_ESTEP_SCHEDULE', 'auto'))
        self.set_estep_dictionary(os.environ.get('DEMUXALOT_AMD_ESTEP_DICT', 'auto'))
        self.set_estep_packing(os.environ.get('DEMUXALOT_AMD_ESTEP_PACKED', 'auto'))
        self.set_coarse_pass(os.environ.get('DEMUXALOT_AMD_COARSE_PASS', '1') != '0')
        self.set_mstep_incremental(os.environ.get('DEMUXALOT_AMD_MSTEP_INCREMENTAL', '1') != '0')
        self.set_lean_memory(os.environ.get('DEMUXALOT_AMD_LEAN', '0') not in ('', '0'))
        self.set_phase_timers(False)
        self.set_logits_needed(True)

    def __enter__(self):
        return self

    def __exit__(self, *_exc):
        self.close()

    def close(self):
        if getattr(self, '_h', None):
            self._lib.dmx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- problem ----------------------------------------------------------------------
    def set_problem(self, n_barcodes, n_variants, n_genotypes, variant_id, compressed_cb, p_base_wrong, v2snp):
        self._resident_key = None  # (demux.py: _pack_on_device keeps the packed problem of the shared context across calls)
        variant_id = as_c(variant_id, np.int32)
        compressed_cb = as_c(compressed_cb, np.int32)
        p_base_wrong = as_c(p_base_wrong, np.float32)
        v2snp = as_c(v2snp, np.int32)
        assert len(variant_id) == len(compressed_cb) == len(p_base_wrong)
        assert len(v2snp) == n_variants
        check(self._lib.dmx_set_problem(self._h, n_barcodes, n_variants, n_genotypes, len(variant_id),
                                        ptr(variant_id), ptr(compressed_cb), ptr(p_base_wrong), ptr(v2snp)))
        self.B, self.V, self.G, self.N = int(n_barcodes), int(n_variants), int(n_genotypes), len(variant_id)

    def pack_and_set_problem(self, n_barcodes, n_genotypes, var_chrom, var_pos, var_base, v2snp,
                             call_chrom, call_pos, call_base, call_cb, call_p):
        """Device pack (variant matching + de-duplication, demux.py:276-300, 332-365) installed directly as
        the resident problem. Returns (n_matched, n_unique, molecules per variant)."""
        self._resident_key = None  # (demux.py: _pack_on_device keeps the packed problem of the shared context across calls)
        var_chrom, var_pos, var_base = as_c(var_chrom, np.int32), as_c(var_pos, np.int32), as_c(var_base, np.uint8)
        v2snp = as_c(v2snp, np.int32)
        call_chrom, call_pos = as_c(call_chrom, np.int32), as_c(call_pos, np.int32)
        call_base, call_cb, call_p = as_c(call_base, np.uint8), as_c(call_cb, np.int32), as_c(call_p, np.float32)
        n_variants, n_calls = len(var_pos), len(call_pos)
        assert len(v2snp) == n_variants
        mol_per_variant = np.zeros(n_variants, dtype=np.int64)
        n_matched, n_unique = ctypes.c_int64(0), ctypes.c_int64(0)
        check(self._lib.dmx_pack_and_set_problem(
            self._h, n_barcodes, n_variants, n_genotypes, ptr(var_chrom), ptr(var_pos), ptr(var_base), ptr(v2snp),
            n_calls, ptr(call_chrom), ptr(call_pos), ptr(call_base), ptr(call_cb), ptr(call_p),
            ctypes.byref(n_matched), ctypes.byref(n_unique), ptr(mol_per_variant)))
        self.B, self.V, self.G, self.N = int(n_barcodes), int(n_variants), int(n_genotypes), n_unique.value
        return n_matched.value, n_unique.value, mol_per_variant

    def _container_list(self, containers):
        from .snp_counter import MOLECULE_DTYPE, SNP_CALL_DTYPE

        class Container(ctypes.Structure):
            _fields_ = [('snp_calls', ctypes.c_void_p), ('n_snp_calls', ctypes.c_int64),
                        ('molecules', ctypes.c_void_p), ('n_molecules', ctypes.c_int64), ('chrom', ctypes.c_int32)]

        keep_alive, parts = [], (Container * max(1, len(containers)))()
        for k, (chrom, snp_calls, molecules) in enumerate(containers):
            assert snp_calls.dtype == SNP_CALL_DTYPE and molecules.dtype == MOLECULE_DTYPE
            snp_calls, molecules = np.ascontiguousarray(snp_calls), np.ascontiguousarray(molecules)
            keep_alive += [snp_calls, molecules]
            parts[k] = Container(snp_calls.ctypes.data, len(snp_calls), molecules.ctypes.data, len(molecules), int(chrom))
        return parts, keep_alive

    def pack_containers_and_set_problem(self, n_barcodes, n_genotypes, var_chrom, var_pos, var_base, v2snp, containers):
        """pack_and_set_problem fed with the raw record arrays of the call containers:
        `containers` = [(chromosome index, snp_calls[:n] (SNP_CALL_DTYPE), molecules[:m] (MOLECULE_DTYPE))].
        The field extraction and the molecule -> barcode lookup happen on the GPU."""
        self._resident_key = None  # (demux.py: _pack_on_device keeps the packed problem of the shared context across calls)
        var_chrom, var_pos, var_base = as_c(var_chrom, np.int32), as_c(var_pos, np.int32), as_c(var_base, np.uint8)
        v2snp = as_c(v2snp, np.int32)
        n_variants = len(var_pos)
        assert len(v2snp) == n_variants
        parts, _keep_alive = self._container_list(containers)
        mol_per_variant = np.zeros(n_variants, dtype=np.int64)
        n_matched, n_unique = ctypes.c_int64(0), ctypes.c_int64(0)
        check(self._lib.dmx_pack_containers_and_set_problem(
            self._h, n_barcodes, n_variants, n_genotypes, ptr(var_chrom), ptr(var_pos), ptr(var_base), ptr(v2snp),
            ctypes.cast(parts, ctypes.c_void_p), len(containers), ctypes.byref(n_matched), ctypes.byref(n_unique),
            ptr(mol_per_variant)))
        self.B, self.V, self.G, self.N = int(n_barcodes), int(n_variants), int(n_genotypes), n_unique.value
        return n_matched.value, n_unique.value, mol_per_variant

    def stage_containers(self, containers):
        """First half of pack_containers_and_set_problem: upload + field extraction of the containers' records, which
        needs nothing of the genotypes (`containers` as there, the chromosome index provisional: the position in the
        list).  The resident problem stays as it is.  include/demux_hip.h: dmx_stage_containers."""
        self._resident_key = None  # (demux.py: _pack_on_device keeps the packed problem of the shared context across calls)
        parts, _keep_alive = self._container_list(containers)
        check(self._lib.dmx_stage_containers(self._h, ctypes.cast(parts, ctypes.c_void_p), len(containers)))

    def pack_staged_and_set_problem(self, n_barcodes, n_genotypes, var_chrom, var_pos, var_base, v2snp, chrom_of_container):
        """Second half: variant matching, de-duplication and layouts on the staged calls.  chrom_of_container[k] = the
        chromosome index (numbering of var_chrom) of the k-th staged container, -1 when no variant lies on it."""
        self._resident_key = None  # (demux.py: _pack_on_device keeps the packed problem of the shared context across calls)
        var_chrom, var_pos, var_base = as_c(var_chrom, np.int32), as_c(var_pos, np.int32), as_c(var_base, np.uint8)
        v2snp = as_c(v2snp, np.int32)
        table = as_c(chrom_of_container, np.int32)
        n_variants = len(var_pos)
        assert len(v2snp) == n_variants
        mol_per_variant = np.zeros(n_variants, dtype=np.int64)
        n_matched, n_unique = ctypes.c_int64(0), ctypes.c_int64(0)
        check(self._lib.dmx_pack_staged_and_set_problem(
            self._h, n_barcodes, n_variants, n_genotypes, ptr(var_chrom), ptr(var_pos), ptr(var_base), ptr(v2snp),
            ptr(table), len(table), ctypes.byref(n_matched), ctypes.byref(n_unique), ptr(mol_per_variant)))
        self.B, self.V, self.G, self.N = int(n_barcodes), int(n_variants), int(n_genotypes), n_unique.value
        return n_matched.value, n_unique.value, mol_per_variant

    def get_packed_calls(self):
        """Unique (variant, barcode) calls left on the device by pack_and_set_problem, variant-major."""
        n = self.N
        variant, cb = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.int32)
        p, count = np.empty(n, dtype=np.float32), np.empty(n, dtype=np.int64)
        check(self._lib.dmx_get_packed_calls(self._h, ptr(variant), ptr(cb), ptr(p), ptr(count)))
        return variant, cb, p, count

    def set_betas(self, betas):
        betas = as_c(betas, np.float32)
        assert betas.shape == (self.V, self.G)
        check(self._lib.dmx_set_betas(self._h, ptr(betas)))

    def set_prior_betas(self, raw_betas, default_prior, add_data_prior, mol_per_variant=None, fetch=True):
        """Regularised prior betas (demux.py:372-388) computed on the GPU from the raw betas."""
        raw_betas = as_c(raw_betas, np.float32)
        assert raw_betas.shape == (self.V, self.G)
        assert np.all(raw_betas >= 0), 'bad genotypes provided, negative betas appeared'
        if mol_per_variant is not None:
            mol_per_variant = as_c(mol_per_variant, np.int64)
            assert mol_per_variant.shape == (self.V,)
        out = np.empty((self.V, self.G), dtype=np.float32) if fetch else None
        check(self._lib.dmx_set_prior_betas(self._h, ptr(raw_betas), float(default_prior), int(bool(add_data_prior)),
                                            ptr(mol_per_variant), ptr(out)))
        if out is not None:
            out.flags.writeable = False
        return out

    def set_addition(self, addition=None):
        if addition is not None:
            addition = as_c(addition, np.float32)
            assert addition.shape == (self.V, self.G)
        check(self._lib.dmx_set_addition(self._h, ptr(addition)))

    def set_probs(self, prob):
        prob = as_c(prob, np.float32)
        assert prob.shape == (self.V, self.G)
        check(self._lib.dmx_set_probs(self._h, ptr(prob)))

    # ---- steps ------------------------------------------------------------------------
    @staticmethod
    def clip_bounds(p_genotype_clip):
        # ndarray.clip(p, 1 - p) on a float32 array: both Python floats become float32 scalars
        return float(np.float32(p_genotype_clip)), float(np.float32(1 - p_genotype_clip))

    def probs_from_betas(self, p_genotype_clip, fetch=True):
        lo, hi = self.clip_bounds(p_genotype_clip)
        out = np.empty((self.V, self.G), dtype=np.float32) if fetch else None
        check(self._lib.dmx_probs_from_betas(self._h, lo, hi, ptr(out)))
        return out

    def probs_from_betas_f64(self, betas, p_genotype_clip):
        """P-step on caller-supplied float64 betas (dmx_probs_from_betas_f64)."""
        betas = as_c(betas, np.float64)
        assert betas.shape == (self.V, self.G)
        lo, hi = self.clip_bounds(p_genotype_clip)
        out = np.empty((self.V, self.G), dtype=np.float32)
        check(self._lib.dmx_probs_from_betas_f64(self._h, ptr(betas), lo, hi, ptr(out)))
        return out

    def _n_options(self, with_doublets):
        return self.G * (self.G + 1) // 2 if with_doublets else self.G

    @staticmethod
    def _prior_arg(prior_logits, shape):
        if prior_logits is None:
            return None, DMX_F32
        prior_logits = np.asarray(prior_logits)
        assert prior_logits.shape == shape, 'mismatching priors passed'
        if prior_logits.dtype == np.float32:
            return as_c(prior_logits, np.float32), DMX_F32
        # any other dtype takes numpy's float64 path of `logits += prior`
        return as_c(prior_logits, np.float64), DMX_F64

    def estep(self, penalties, with_doublets, prior_logits=None, fetch_logits=True, fetch_probs=True):
        K = self._n_options(with_doublets)
        penalties = as_c(penalties, np.float32)
        assert penalties.shape == (K,)
        prior, prior_dtype = self._prior_arg(prior_logits, (self.B, K))
        logits = np.empty((self.B, K), dtype=np.float32) if fetch_logits else None
        probs = np.empty((self.B, K), dtype=np.float32) if fetch_probs else None
        check(self._lib.dmx_estep(self._h, int(with_doublets), ptr(penalties), ptr(prior), prior_dtype,
                                  ptr(logits), ptr(probs)))
        self.K = K
        return logits, probs

    def mstep(self, contribution_power=2., fetch=True):
        out = np.empty((self.V, self.G), dtype=np.float32) if fetch else None
        check(self._lib.dmx_mstep(self._h, float(contribution_power), ptr(out)))
        return out

    def em(self, n_iterations, p_genotype_clip, penalties, with_doublets, prior_logits=None,
           contribution_power=2., fetch_logits=True, fetch_probs=True, fetch_addition=True):
        K = self._n_options(with_doublets)
        lo, hi = self.clip_bounds(p_genotype_clip)
        penalties = as_c(penalties, np.float32)
        assert penalties.shape == (K,)
        prior, prior_dtype = self._prior_arg(prior_logits, (self.B, K))
        logits = np.empty((self.B, K), dtype=np.float32) if fetch_logits else None
        probs = np.empty((self.B, K), dtype=np.float32) if fetch_probs else None
        addition = np.empty((self.V, self.G), dtype=np.float32) if fetch_addition else None
        check(self._lib.dmx_em(self._h, int(n_iterations), lo, hi, int(with_doublets), ptr(penalties), ptr(prior),
                               prior_dtype, float(contribution_power), ptr(logits), ptr(probs), ptr(addition)))
        self.K = K
        return logits, probs, addition

    def run_iterations(self, n_iterations, p_genotype_clip, contribution_power=2.):
        """Enqueue n full EM iterations (P, E, M) without synchronising; see dmx_run_iterations."""
        lo, hi = self.clip_bounds(p_genotype_clip)
        check(self._lib.dmx_run_iterations(self._h, int(n_iterations), lo, hi, float(contribution_power)))

    def get_logits(self):
        out = np.empty((self.B, self.K), dtype=np.float32)
        check(self._lib.dmx_get_logits(self._h, ptr(out)))
        return out

    def get_probs(self):
        out = np.empty((self.B, self.K), dtype=np.float32)
        check(self._lib.dmx_get_probs(self._h, ptr(out)))
        return out

    def get_addition(self):
        out = np.empty((self.V, self.G), dtype=np.float32)
        check(self._lib.dmx_get_addition(self._h, ptr(out)))
        return out

    def get_block(self, what, b0=0, b1=None, k0=0, k1=None):
        """Rows [b0, b1) x columns [k0, k1) of the resident 'logits' or 'probs' (dmx_get_block)."""
        b1 = self.B if b1 is None else b1
        k1 = self.K if k1 is None else k1
        out = np.empty((b1 - b0, k1 - k0), dtype=np.float32)
        check(self._lib.dmx_get_block(self._h, {'logits': 0, 'probs': 1}[what], b0, b1, k0, k1, ptr(out)))
        return out

    def get_assignments(self):
        best = np.empty(self.B, dtype=np.int32)
        prob = np.empty(self.B, dtype=np.float32)
        check(self._lib.dmx_get_assignments(self._h, ptr(best), ptr(prob)))
        return best, prob

    # ---- aggregate_on_snps (demux.py:204-244) --------------------------------------------------------
    def set_keep_molecule_calls(self, keep):
        self._keep_molecule_calls = bool(keep)
        check(self._lib.dmx_set_keep_molecule_calls(self._h, int(bool(keep))))

    def set_molecule_calls(self, variant_id, compressed_cb, p_base_wrong):
        self._resident_key = None  # (demux.py: _pack_on_device keeps the packed problem of the shared context across calls)
        variant_id, compressed_cb = as_c(variant_id, np.int32), as_c(compressed_cb, np.int32)
        p_base_wrong = as_c(p_base_wrong, np.float32)
        assert len(variant_id) == len(compressed_cb) == len(p_base_wrong)
        check(self._lib.dmx_set_molecule_calls(self._h, len(variant_id), ptr(variant_id), ptr(compressed_cb), ptr(p_base_wrong)))

    def estep_snp(self, with_doublets, compensation, prior_logits=None):
        """E-step of aggregate_on_snps: (logits float64[B, K], posteriors float64[B, K])."""
        K = self._n_options(with_doublets)
        n = ctypes.c_int64(0)
        check(self._lib.dmx_get_max_pair_count(self._h, ctypes.byref(n)))
        # `counts[:, None] ** compensation` of demux.py:232 for every count that occurs, evaluated by numpy itself
        # on an int64 array like the reference's, so that its pow() is repeated to the last bit
        count_pow = np.ascontiguousarray(np.arange(n.value + 1, dtype=np.int64) ** compensation, dtype=np.float64)
        prior, prior_dtype = self._prior_arg(prior_logits, (self.B, K))
        logits = np.empty((self.B, K), dtype=np.float64)
        probs = np.empty((self.B, K), dtype=np.float64)
        check(self._lib.dmx_estep_snp(self._h, int(with_doublets), ptr(count_pow), len(count_pow), ptr(prior), prior_dtype,
                                      ptr(logits), ptr(probs)))
        self.K = K
        return logits, probs

    def mstep_f64(self, contribution_power=2., as_float64=False):
        """M-step on the float64 posteriors of estep_snp.  as_float64: the unrounded float64 sums (a barcode-sharded run
        adds them over ranks before the one float32 rounding)."""
        if as_float64:
            out = np.empty((self.V, self.G), dtype=np.float64)
            check(self._lib.dmx_mstep_f64_sums(self._h, float(contribution_power), ptr(out)))
            return out
        out = np.empty((self.V, self.G), dtype=np.float32)
        check(self._lib.dmx_mstep_f64(self._h, float(contribution_power), ptr(out)))
        return out

    def get_prior_betas(self):
        out = np.empty((self.V, self.G), dtype=np.float32)
        check(self._lib.dmx_get_prior_betas(self._h, ptr(out)))
        out.flags.writeable = False
        return out

    def get_learnt_betas(self):
        """raw betas (as given to set_prior_betas) + the last M-step's addition, float32 [V, G], added on the device
        (include/demux_hip.h: dmx_get_learnt_betas; demux.py:65)."""
        out = np.empty((self.V, self.G), dtype=np.float32)
        check(self._lib.dmx_get_learnt_betas(self._h, ptr(out)))
        return out

    def get_assignments_above(self, threshold):
        """(best option or -1 where the row maximum is not > threshold, row maximum, number assigned)."""
        best = np.empty(self.B, dtype=np.int32)
        prob = np.empty(self.B, dtype=np.float32)
        n = ctypes.c_int64(0)
        check(self._lib.dmx_get_assignments_above(self._h, float(threshold), ptr(best), ptr(prob), ctypes.byref(n)))
        return best, prob, n.value

    def get_top_options(self, k):
        """The k (<= 4) best options per barcode, best first: (int32[B, k], float32[B, k])."""
        options = np.empty((self.B, int(k)), dtype=np.int32)
        probs = np.empty((self.B, int(k)), dtype=np.float32)
        check(self._lib.dmx_get_top_options(self._h, int(k), ptr(options), ptr(probs)))
        return options, probs

    def get_option_sums(self):
        """probs.sum(axis=0) as float64[K]."""
        out = np.empty(self.K, dtype=np.float64)
        check(self._lib.dmx_get_option_sums(self._h, ptr(out)))
        return out

    def synchronize(self):
        check(self._lib.dmx_synchronize(self._h))

    # ---- multi-GPU ----------------------------------------------------------------------
    @staticmethod
    def new_unique_id() -> bytes:
        buf = ctypes.create_string_buffer(_lib.UNIQUE_ID_BYTES)
        check(_lib.load().dmx_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, rank, nranks, unique_id: bytes, reduce_dtype='f64'):
        assert len(unique_id) == _lib.UNIQUE_ID_BYTES
        buf = ctypes.create_string_buffer(unique_id, _lib.UNIQUE_ID_BYTES)
        check(self._lib.dmx_comm_init(self._h, int(rank), int(nranks), buf, DMX_F64 if reduce_dtype == 'f64' else DMX_F32))

    def comm_init_host(self, rank, nranks, collective, reduce_dtype='f64'):
        """The exchange over the caller's collectives instead of RCCL (include/demux_hip.h: dmx_comm_init_host).
        collective(op, array): op in ('all_reduce', 'reduce_scatter', 'all_gather'); array is the host buffer as a
        numpy view, float32 or float64 - [count] for all_reduce (sum in place), [nranks, count] otherwise
        (reduce_scatter: row `rank` must hold the sum over ranks of their row `rank`; all_gather: row `rank` is
        filled, every row must be on return).  An exception fails the calling step."""
        names = {_lib.COLL_ALL_REDUCE: 'all_reduce', _lib.COLL_REDUCE_SCATTER: 'reduce_scatter', _lib.COLL_ALL_GATHER: 'all_gather'}

        def trampoline(_user, op, buf, count, dtype):
            try:
                kind = np.float64 if dtype == DMX_F64 else np.float32
                n = count if op == _lib.COLL_ALL_REDUCE else count * nranks
                raw = (ctypes.c_char * (n * np.dtype(kind).itemsize)).from_address(buf)
                view = np.frombuffer(raw, dtype=kind)
                collective(names[op], view if op == _lib.COLL_ALL_REDUCE else view.reshape(nranks, count))
                return 0
            except Exception:  # noqa: BLE001 - the C side turns the code into DemuxHipError
                import traceback
                traceback.print_exc()
                return 1
        self._host_collective = _lib.HOST_COLLECTIVE(trampoline)  # kept alive with the context
        check(self._lib.dmx_comm_init_host(self._h, int(rank), int(nranks), self._host_collective, None,
                                           DMX_F64 if reduce_dtype == 'f64' else DMX_F32))

    def comm_init_emulated(self, rank, nranks, link_gbytes_per_s=50.0, latency_us=10.0, reduce_dtype='f64'):
        """This context as rank `rank` of `nranks` over an EMULATED wire (include/demux_hip.h: dmx_comm_init_emulated): for
        timing the exchange schedule on one GPU; the results of such a run are not an EM of any experiment."""
        check(self._lib.dmx_comm_init_emulated(self._h, int(rank), int(nranks), float(link_gbytes_per_s), float(latency_us),
                                               DMX_F64 if reduce_dtype == 'f64' else DMX_F32))

    def exchange_compact(self):
        """(E-steps whose posterior rows travelled as a list, E-steps that fell back to the whole table, rows a list can hold; 0: the
        compact form is off) - include/demux_hip_debug.h: dmx_get_exchange_compact."""
        taken, overflows, cap = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
        check(self._lib.dmx_get_exchange_compact(self._h, ctypes.byref(taken), ctypes.byref(overflows), ctypes.byref(cap)))
        return int(taken.value), int(overflows.value), int(cap.value)

    def exchange_compact_table(self):
        """The same for the all-gather of genotype_prob behind the sliced P-step (dmx_get_exchange_compact_table)."""
        taken, overflows, cap = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
        check(self._lib.dmx_get_exchange_compact_table(self._h, ctypes.byref(taken), ctypes.byref(overflows), ctypes.byref(cap)))
        return int(taken.value), int(overflows.value), int(cap.value)

    def exchange_mode(self):
        """None (no communicator) | 'variant' (M-step sharded on variants, posteriors all-gathered) | 'reduce_scatter' |
        'allreduce' (exchanges of the per-rank sums); include/demux_hip.h: dmx_get_exchange_mode."""
        mode = ctypes.c_int32(0)
        check(self._lib.dmx_get_exchange_mode(self._h, ctypes.byref(mode)))
        return {0: None, 1: 'variant', 2: 'reduce_scatter', 3: 'allreduce'}[mode.value]

    def set_estep_mode(self, mode):
        """'exact' (logits / posteriors bit-identical to the reference), 'guarded' (tolerance-mode arithmetic, every
        barcode whose posteriors are not provably within the contract's 1e-5 of the reference's - or whose argmax could
        differ - redone exactly) or 'fast' (tolerance mode, unguarded); include/demux_hip.h: dmx_set_estep_mode."""
        check(self._lib.dmx_set_estep_mode(self._h, {'exact': 0, 'fast': 1, 'guarded': 2}[mode]))

    def guard_stats(self):
        """(barcodes the last guarded E-step redid exactly, the same over all E-steps since reset_timings, barcode rows
        those E-steps walked); include/demux_hip.h: dmx_get_guard_stats."""
        last, total, rows = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
        check(self._lib.dmx_get_guard_stats(self._h, ctypes.byref(last), ctypes.byref(total), ctypes.byref(rows)))
        return last.value, total.value, rows.value

    def set_guard_adaptive(self, adaptive):
        """Guarded mode: the passes are timed on the device and every E-step takes the cheapest of coarse pass + redo, fine pass + redo
        and the exact kernel on every barcode (default on; off: never the last one; include/demux_hip.h: dmx_set_guard_adaptive)."""
        check(self._lib.dmx_set_guard_adaptive(self._h, int(bool(adaptive))))

    def set_coarse_pass(self, coarse):
        """Guarded mode: E-steps whose logits nobody reads may take the coarse pass (binary16 genotype table; default on;
        'always': every E-step, single ones included - their logits then carry the coarse bound; include/demux_hip.h: dmx_set_coarse_pass)."""
        check(self._lib.dmx_set_coarse_pass(self._h, 2 if coarse == 'always' else int(bool(coarse))))

    def set_lean_memory(self, lean):
        """Memory before speed for the E-steps whose logits are kept: the tile-major copy of the E-step records (16 of 67 bytes per call) is
        released once the coarse pass's records are built from it (include/demux_hip.h: dmx_set_lean_memory; default off)."""
        check(self._lib.dmx_set_lean_memory(self._h, int(bool(lean))))

    def set_mstep_incremental(self, incremental):
        """Default mode, tile-major M-step: update the kept integer sums for the barcodes whose posteriors changed instead of summing every
        call again (same bits; default on; include/demux_hip.h: dmx_set_mstep_incremental)."""
        check(self._lib.dmx_set_mstep_incremental(self._h, 2 if incremental == 'bootstrap' else int(bool(incremental))))

    def mstep_incremental(self):
        """(full passes, delta passes since reset_timings, barcodes the last delta pass visited)"""
        full, delta, last = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        check(self._lib.dmx_get_mstep_incremental(self._h, ctypes.byref(full), ctypes.byref(delta), ctypes.byref(last)))
        return full.value, delta.value, last.value

    def guard_levels(self):
        """dict: level of the last guarded E-step (0 coarse, 1 fine, 2 direct, -1 none), coarse E-steps since reset_timings, barcodes
        flagged by the fine / coarse guard in the last one, the device's timings of the passes in ms (dmx_get_guard_levels)."""
        level = ctypes.c_int32()
        steps, fine, coarse = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        c_ms, f_ms, e_ms = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        check(self._lib.dmx_get_guard_levels(self._h, ctypes.byref(level), ctypes.byref(steps), ctypes.byref(fine), ctypes.byref(coarse),
                                             ctypes.byref(c_ms), ctypes.byref(f_ms), ctypes.byref(e_ms)))
        return {'level': level.value, 'coarse_steps': steps.value, 'flagged_fine': fine.value, 'flagged_coarse': coarse.value,
                'coarse_pass_ms': c_ms.value, 'fine_pass_ms': f_ms.value, 'exact_pass_ms': e_ms.value}

    def set_logits_needed(self, needed):
        """False: nobody will read the logits of the last E-step of the em / run_iterations calls that follow (learn_genotypes returns
        posteriors only), so that E-step too may take the coarse pass; get_logits / get_block('logits') then raise until an E-step
        keeps its logits again (include/demux_hip.h: dmx_set_logits_needed).  Default True."""
        check(self._lib.dmx_set_logits_needed(self._h, int(bool(needed))))

    def guard_probes(self):
        """(E-steps that ran another level than the cheapest to have it timed again since reset_timings, E-steps in a row on the
        current level); include/demux_hip.h: dmx_get_guard_probes."""
        probes, streak = ctypes.c_int64(), ctypes.c_int64()
        check(self._lib.dmx_get_guard_probes(self._h, ctypes.byref(probes), ctypes.byref(streak)))
        return probes.value, streak.value

    def debug_set_pass_ms(self, coarse=-1.0, fine=-1.0, exact=-1.0):
        """Testing aid: overwrite the device's own times of the passes (include/demux_hip.h: dmx_debug_set_pass_ms)."""
        check(self._lib.dmx_debug_set_pass_ms(self._h, float(coarse), float(fine), float(exact)))

    def guard_direct(self):
        """(the last guarded E-step ran direct, E-steps run direct since reset_timings, barcodes the last one queued or -
        direct - would have queued); include/demux_hip.h: dmx_get_guard_direct."""
        return self.guard_state()[:3]

    def guard_state(self):
        """guard_direct() + the device's own timings of the two passes: (..., fast pass over all barcodes in ms, exact kernel
        over all barcodes in ms; 0 = not known yet, negative = estimated from the redo's share, not yet measured)."""
        last, steps, would = ctypes.c_int32(0), ctypes.c_int64(0), ctypes.c_int64(0)
        fast_ms, exact_ms = ctypes.c_double(0), ctypes.c_double(0)
        check(self._lib.dmx_get_guard_direct(self._h, ctypes.byref(last), ctypes.byref(steps), ctypes.byref(would),
                                             ctypes.byref(fast_ms), ctypes.byref(exact_ms)))
        return bool(last.value), steps.value, would.value, fast_ms.value, exact_ms.value

    def set_estep_dictionary(self, mode):
        """'never' | 'auto' (default: tried for genotype tables computed without a beta addition) | 'always' (tried for
        every E-step).  Bit-identical results in every mode; include/demux_hip.h: dmx_set_estep_dictionary."""
        check(self._lib.dmx_set_estep_dictionary(self._h, {'never': 0, 'auto': 1, 'always': 2}[mode]))

    def estep_form(self):
        """(form, distinct) of the last E-step: form 'direct' | 'packed' | 'dict' | 'dict_block' | None, distinct = most distinct
        values in a row of genotype_prob as counted by the last dictionary build (0: none was tried, 9: too many)."""
        import ctypes
        form, distinct = ctypes.c_int32(0), ctypes.c_int32(0)
        check(self._lib.dmx_get_estep_form(self._h, ctypes.byref(form), ctypes.byref(distinct)))
        return {0: None, 1: 'direct', 2: 'dict', 3: 'dict_block', 4: 'packed'}[form.value], distinct.value

    def set_estep_packing(self, mode):
        """'never' | 'auto' (default: where it pays) | 'always' (every barcode on packed lane groups) | 'split' (the
        longest barcodes on 64 lanes, the rest packed, whatever their number): several option slots per lane for narrow
        doublet tables (include/demux_hip.h: dmx_set_estep_packing)."""
        check(self._lib.dmx_set_estep_packing(self._h, {'never': 0, 'auto': 1, 'always': 2, 'split': 3}[mode]))

    def set_estep_schedule(self, schedule):
        """'auto' (default: tile-major schedule where it pays), 'tiled' (whenever built), 'direct' (never);
        include/demux_hip.h: dmx_set_estep_schedule."""
        check(self._lib.dmx_set_estep_schedule(self._h, {'direct': 0, 'auto': 1, 'tiled': 2}[schedule]))

    def set_mstep_wide_addresses(self, wide):
        """include/demux_hip.h: dmx_set_mstep_wide_addresses (the M-step form of the largest problems, at any size)."""
        check(self._lib.dmx_set_mstep_wide_addresses(self._h, int(bool(wide))))

    def set_mstep_tiles(self, mode):
        """Tile-major form of the M-step: 'never' | 'auto' (default: when building its records pays, i.e. from 8 M-steps on) |
        'always' (at the first M-step); needs the exact additions off and G <= 64.  include/demux_hip.h: dmx_set_mstep_tiles."""
        if isinstance(mode, str):
            mode = {'never': 0, 'auto': 1, 'always': 2}[mode]
        check(self._lib.dmx_set_mstep_tiles(self._h, int(mode) if not isinstance(mode, bool) else (2 if mode else 0)))

    def set_msteps_expected(self, n):
        """Hint: about n more M-steps will run on the resident problem (include/demux_hip.h: dmx_set_msteps_expected)."""
        check(self._lib.dmx_set_msteps_expected(self._h, int(n)))

    def mstep_tiles_info(self):
        """(the tile-major M-step records exist, host wall time of their build in ms)."""
        built, ms = ctypes.c_int32(0), ctypes.c_double(0)
        check(self._lib.dmx_get_mstep_tiles_info(self._h, ctypes.byref(built), ctypes.byref(ms)))
        return bool(built.value), ms.value

    def mstep_form(self):
        """None | 'items' | 'tiles' | 'items_fixed' (the work items adding the tile-major form's integers, under the incremental
        M-step: calls too short to pay for the tile-major records): the form of the last M-step launch (dmx_get_mstep_form)."""
        form = ctypes.c_int32(0)
        check(self._lib.dmx_get_mstep_form(self._h, ctypes.byref(form)))
        return {0: None, 1: 'items', 2: 'tiles', 3: 'items_fixed'}[form.value]

    def redo_count(self):
        """Sums the last exact-mode M-step redid in the reference's order (include/demux_hip.h: dmx_get_redo_count)."""
        n = ctypes.c_int64(0)
        check(self._lib.dmx_get_redo_count(self._h, ctypes.byref(n)))
        return n.value

    def set_exact_additions(self, exact):
        """M-step summation mode (include/demux_hip.h: dmx_set_exact_additions). Default: exact."""
        check(self._lib.dmx_set_exact_additions(self._h, int(bool(exact))))

    # ---- instrumentation ----------------------------------------------------------------
    def set_phase_timers(self, on):
        """HIP events around the phases of the calls that follow, read by timings() (default off - an event record is a barrier
        packet of its own, 6 us per phase boundary; the launch counts are kept either way: include/demux_hip.h dmx_set_phase_timers)."""
        check(self._lib.dmx_set_phase_timers(self._h, int(bool(on))))

    def timings(self):
        """{phase: {ms, launches}} since reset_timings(); ms only of the phases that ran with set_phase_timers(True), -1 when
        the phase ran but never with the timers on (dmx_get_timings)."""
        ms = (ctypes.c_double * _lib.T_COUNT)()
        n = (ctypes.c_int64 * _lib.T_COUNT)()
        check(self._lib.dmx_get_timings(self._h, ms, n))
        return {name: dict(ms=ms[i], launches=int(n[i])) for i, name in enumerate(_lib.TIMER_NAMES)}

    def reset_timings(self):
        check(self._lib.dmx_reset_timings(self._h))

    def trim_cache(self):
        """Give the device blocks this context keeps for re-use back to the driver; returns the bytes released
        (include/demux_hip.h: dmx_trim_cache)."""
        n = ctypes.c_int64(0)
        check(self._lib.dmx_trim_cache(self._h, ctypes.byref(n)))
        return n.value

    def release_problem(self):
        """The resident problem (and any staged containers) back into this context's block cache
        (include/demux_hip.h: dmx_release_problem)."""
        self._resident_key = None  # (demux.py: _pack_on_device keeps the packed problem of the shared context across calls)
        check(self._lib.dmx_release_problem(self._h))
        self.B = self.V = self.G = self.N = self.K = 0

    def device_bytes(self):
        n = ctypes.c_int64(0)
        check(self._lib.dmx_device_bytes(self._h, ctypes.byref(n)))
        return n.value

    # ---- device self-tests (numpy-exact float32 building blocks) -------------------------
    def test_log(self, x):
        x = as_c(x, np.float32)
        out = np.empty_like(x)
        check(self._lib.dmx_test_logf(self._h, ptr(x), ptr(out), x.size))
        return out

    def test_log_hot(self, x):
        x = as_c(x, np.float32)
        out = np.empty_like(x)
        check(self._lib.dmx_test_logf_hot(self._h, ptr(x), ptr(out), x.size))
        return out

    def test_log2_hw(self, x):
        x = as_c(x, np.float32)
        out = np.empty_like(x)
        check(self._lib.dmx_test_log2_hw(self._h, ptr(x), ptr(out), x.size))
        return out

    def test_exp(self, x):
        x = as_c(x, np.float32)
        out = np.empty_like(x)
        check(self._lib.dmx_test_expf(self._h, ptr(x), ptr(out), x.size))
        return out

    def test_softmax(self, x):
        x = as_c(x, np.float32)
        out = np.empty_like(x)
        check(self._lib.dmx_test_softmax(self._h, ptr(x), ptr(out), x.shape[0], x.shape[1]))
        return out


_contexts = {}
_contexts_lock = threading.RLock()


def default_device():
    """GPU used by the Demultiplexer front-end: DEMUXALOT_AMD_DEVICE (taken as is), else LOCAL_RANK, else 0.
    Launchers often set HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES per rank as well, so that every rank sees
    ONE GPU as ordinal 0 while LOCAL_RANK still counts up: LOCAL_RANK is therefore folded into the visible
    device count."""
    if os.environ.get('DEMUXALOT_AMD_DEVICE', '') != '':
        return int(os.environ['DEMUXALOT_AMD_DEVICE'])
    if os.environ.get('LOCAL_RANK', '') != '':
        n = _lib.device_count()
        return int(os.environ['LOCAL_RANK']) % n if n > 0 else 0
    return 0


def get_context(device=None) -> DeviceContext:
    """Process-wide cached context per device. Raises when no GPU is visible (no CPU fallback).
    The cached context holds ONE resident problem: callers that keep state on the GPU across calls of other
    entry points (the staged_genotype_learning generator, DevicePosteriors) own a private DeviceContext
    instead, and the Demultiplexer entry points that use this one serialise on `shared_context_lock`."""
    device = default_device() if device is None else int(device)
    with _contexts_lock:
        if device not in _contexts:
            _contexts[device] = DeviceContext(device)
            _contexts[device]._is_shared = True  # (demux.py: _pack_on_device keeps its packed problem across calls)
        return _contexts[device]


shared_context_lock = threading.RLock()


def shared_contexts():
    """The process-wide contexts created so far, one per device (demux.py: invalidate_resident)."""
    with _contexts_lock:
        return list(_contexts.values())

# Private contexts (a DevicePosteriors, a staged_genotype_learning generator) come from a small pool: a context that
# has been used keeps its stream and, through the runtime's allocator, the address ranges of its buffers, and
# installing the next problem on it costs what it costs on the shared context; a brand-new context pays for fresh
# multi-gigabyte allocations first (predict_posteriors(on_device=True) at 200k x 100k x 64: 0.73 s against 0.31 s).
_idle_private = {}
_PRIVATE_POOL = 2


def acquire_private_context(device=None) -> DeviceContext:
    device = default_device() if device is None else int(device)
    with _contexts_lock:
        idle = _idle_private.get(device)
        ctx = idle.pop() if idle else None
    if ctx is None:
        return DeviceContext(device)
    ctx.apply_environment()
    ctx.set_keep_molecule_calls(False)
    return ctx


def release_private_context(ctx, failed=False):
    """Back to the pool, its resident problem released into its block cache (the next problem installed on it re-uses the
    blocks; an allocation that runs out of device memory anywhere in the process gives the parked blocks of every context
    back first: csrc/dmx_api.cpp ctx_malloc).  `failed`: the holder is unwinding from an exception - the context, which
    may carry a sticky HIP error, is destroyed instead.  The pool holds two contexts per device; drain_private_contexts()
    empties it, trim_device_caches() returns every parked block of a device to the driver."""
    if ctx is None or getattr(ctx, '_h', None) is None:
        return
    if not failed:
        try:
            ctx.release_problem()
        except Exception:  # noqa: BLE001 - a context that cannot even release is not worth keeping
            failed = True
    if not failed:
        with _contexts_lock:
            idle = _idle_private.setdefault(ctx.device, [])
            if len(idle) < _PRIVATE_POOL:
                idle.append(ctx)
                return
    ctx.close()


def drain_private_contexts():
    """Destroys the pooled private contexts (their blocks go to the device's retired list; trim_device_caches() frees those)."""
    with _contexts_lock:
        pooled = [ctx for idle in _idle_private.values() for ctx in idle]
        _idle_private.clear()
    for ctx in pooled:
        ctx.close()


def trim_device_caches(device=None):
    """Every device block this process keeps parked on `device` - idle blocks of all live contexts, blocks of destroyed
    ones - back to the driver; returns the bytes (include/demux_hip.h: dmx_trim_device_caches)."""
    device = default_device() if device is None else int(device)
    n = ctypes.c_int64(0)
    check(_lib.load().dmx_trim_device_caches(device, ctypes.byref(n)))
    return n.value
