"""
tests/golden/make_fixtures.py -- regenerates the golden vectors under tests/golden/.

Runs ONLY in the build container, where the reference lives at /root/reference:
it imports the reference's own `demuxalot` package (pure Python) and captures the
inputs and outputs of `Demultiplexer.pack_calls`, `predict_posteriors` and
`staged_genotype_learning` / `learn_genotypes` as plain arrays.  Nothing of the
reference's source travels: the fixtures are data (inputs + expected outputs).

`pysam` (htslib bindings) is not installed here and is only needed by the
reference's BAM front-end, so an in-memory stand-in module is registered before
the import.  It implements just the surface the reference's synthetic test
(tests/test_synthetic.py:106-145) and `count_snps` touch, so that the F1/F2
inputs are produced by the reference's own generator and BAM scanner.

    python tests/golden/make_fixtures.py            # writes tests/golden/*.npz
"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = '/root/reference'


# --------------------------------------------------------------------------- #
# in-memory pysam stand-in
# --------------------------------------------------------------------------- #

def _install_pysam_standin():
    mod = types.ModuleType('pysam')
    store = {}  # filename -> {'header': dict, 'reads': list}

    class AlignedSegment:
        def __init__(self):
            self.query_name = None
            self.query_sequence = None
            self.flag = 0
            self.reference_id = -1
            self.reference_start = 0
            self.mapping_quality = 0
            self.cigar = ()
            self.template_length = 0
            self.query_qualities = None
            self.tags = ()

        @property
        def seq(self):
            return self.query_sequence

        @property
        def pos(self):
            return self.reference_start

        @property
        def mapq(self):
            return self.mapping_quality

        @property
        def cigartuples(self):
            return list(self.cigar)

        @property
        def reference_end(self):
            return self.reference_start + sum(n for op, n in self.cigar if op in (0, 2, 3, 7, 8))

        def has_tag(self, name):
            return any(t[0] == name for t in self.tags)

        def get_tag(self, name):
            for t in self.tags:
                if t[0] == name:
                    return t[1]
            raise KeyError(name)

    class _Stat:
        def __init__(self, contig, mapped):
            self.contig, self.mapped = contig, mapped

    def decode_bam(path):
        """Minimal BAM reader (BGZF = concatenated gzip members; record layout of the SAM/BAM spec)
        for the reference's shipped example data."""
        import gzip
        import struct
        data = gzip.open(path, 'rb').read()
        assert data[:4] == b'BAM\x01'
        off = 4
        l_text, = struct.unpack_from('<i', data, off)
        off += 4 + l_text
        n_ref, = struct.unpack_from('<i', data, off)
        off += 4
        sq = []
        for _ in range(n_ref):
            l_name, = struct.unpack_from('<i', data, off)
            name = data[off + 4:off + 4 + l_name - 1].decode()
            l_ref, = struct.unpack_from('<i', data, off + 4 + l_name)
            off += 8 + l_name
            sq.append(dict(SN=name, LN=l_ref))
        reads = []
        tag_fmt = {'c': '<b', 'C': '<B', 's': '<h', 'S': '<H', 'i': '<i', 'I': '<I', 'f': '<f'}
        while off < len(data):
            size, = struct.unpack_from('<i', data, off)
            rec = data[off + 4:off + 4 + size]
            off += 4 + size
            ref_id, pos, l_name, mapq, _bin, n_cigar, flag, l_seq, _nr, _np, tlen = struct.unpack_from('<iiBBHHHiiii', rec, 0)
            p = 32
            seg = AlignedSegment()
            seg.query_name = rec[p:p + l_name - 1].decode()
            p += l_name
            seg.cigar = tuple((c & 0xF, c >> 4) for c in struct.unpack_from(f'<{n_cigar}I', rec, p))
            p += 4 * n_cigar
            packed = rec[p:p + (l_seq + 1) // 2]
            p += (l_seq + 1) // 2
            seg.query_sequence = ''.join('=ACMGRSVTWYHKDBN'[(packed[i >> 1] >> (4 if i % 2 == 0 else 0)) & 0xF] for i in range(l_seq))
            seg.query_qualities = np.frombuffer(rec[p:p + l_seq], dtype=np.uint8).astype(np.int64)
            p += l_seq
            tags = []
            while p < len(rec):
                key, typ = rec[p:p + 2].decode(), chr(rec[p + 2])
                p += 3
                if typ == 'A':
                    val = chr(rec[p]); p += 1
                elif typ in tag_fmt:
                    val, = struct.unpack_from(tag_fmt[typ], rec, p); p += struct.calcsize(tag_fmt[typ])
                elif typ in 'ZH':
                    end = rec.index(b'\x00', p)
                    val = rec[p:end].decode(); p = end + 1
                else:
                    raise NotImplementedError(f'BAM tag type {typ}')
                tags.append((key, val))
            seg.tags = tuple(tags)
            seg.flag, seg.reference_id, seg.reference_start, seg.mapping_quality, seg.template_length = flag, ref_id, pos, mapq, tlen
            reads.append(seg)
        return {'header': {'SQ': sq}, 'reads': reads}

    class AlignmentFile:
        def __init__(self, filename, mode='rb', header=None):
            self.filename = str(filename)
            self.mode = mode
            if 'w' in mode:
                store[self.filename] = {'header': header, 'reads': []}
            elif self.filename not in store:
                store[self.filename] = decode_bam(self.filename)
            self._data = store[self.filename]

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def close(self):
            pass

        def write(self, read):
            self._data['reads'].append(read)

        def _names(self):
            return [sq['SN'] for sq in self._data['header']['SQ']]

        def get_reference_length(self, chrom):
            return {sq['SN']: sq['LN'] for sq in self._data['header']['SQ']}[chrom]

        def get_index_statistics(self):
            names = self._names()
            counts = {n: 0 for n in names}
            for r in self._data['reads']:
                counts[names[r.reference_id]] += 1
            return [_Stat(n, counts[n]) for n in names]

        def fetch(self, contig=None, start=None, stop=None):
            rid = self._names().index(contig)
            for r in self._data['reads']:
                if r.reference_id != rid:
                    continue
                if start is not None and r.reference_end <= start:
                    continue
                if stop is not None and r.reference_start >= stop:
                    continue
                yield r

    def sort(*args):
        assert args[0] == '-o'
        out, src = args[1], args[2]
        reads = sorted(store[src]['reads'], key=lambda r: (r.reference_id, r.reference_start))  # stable
        store[out] = {'header': store[src]['header'], 'reads': reads}

    def index(filename):
        return None

    def qualitystring_to_array(s):
        return np.frombuffer(s.encode(), dtype=np.uint8).astype(np.int64) - 33

    mod.AlignedSegment = AlignedSegment
    mod.AlignedRead = AlignedSegment
    mod.AlignmentFile = AlignmentFile
    class _VcfRecord:
        def __init__(self, chrom, pos, alleles, samples):
            self.chrom, self.pos, self.alleles, self.samples = chrom, pos, alleles, samples

    class VariantFile:
        """Text VCF -> the record surface the reference's add_vcf reads (genotypes.py:123-154):
        .chrom, .pos (1-based), .alleles, .samples[name]['GT'] (allele indices, None when missing)."""

        def __init__(self, path):
            self.path = path

        def fetch(self):
            names = None
            for line in open(self.path):
                if line.startswith('##') or not line.strip():
                    continue
                f = line.rstrip('\n').split('\t')
                if line.startswith('#'):
                    names = f[9:]
                    continue
                alleles = tuple([f[3]] + ([] if f[4] == '.' else f[4].split(',')))
                slot = f[8].split(':').index('GT')
                samples = {}
                for name, entry in zip(names, f[9:]):
                    gt = entry.split(':')[slot].replace('|', '/').split('/')
                    samples[name] = {'GT': tuple(None if t == '.' else int(t) for t in gt)}
                yield _VcfRecord(f[0], int(f[1]), alleles, samples)

    mod.VariantFile = VariantFile
    mod.sort = sort
    mod.index = index
    mod.qualitystring_to_array = qualitystring_to_array
    sys.modules['pysam'] = mod
    return mod


def import_reference():
    """Imports the reference package and its synthetic-test module (container only)."""
    if 'pysam' not in sys.modules:
        _install_pysam_standin()
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
    import demuxalot  # noqa: the reference
    assert demuxalot.__file__.startswith(REFERENCE), demuxalot.__file__
    spec = importlib.util.spec_from_file_location('ref_test_synthetic', f'{REFERENCE}/tests/test_synthetic.py')
    ref_tests = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_tests)
    return demuxalot, ref_tests


# --------------------------------------------------------------------------- #
# plain-array views of the reference's objects
# --------------------------------------------------------------------------- #
BASES = {'A': 0, 'C': 1, 'G': 2, 'T': 3, 'N': 4}


def inputs_as_arrays(calls, genotypes, barcode_handler):
    out = {}
    chroms = list(calls)
    out['chroms'] = np.asarray(chroms, dtype=str)
    for i, chrom in enumerate(chroms):
        c = calls[chrom]
        mol = c.molecules[:c.n_molecules]
        sc = c.snp_calls[:c.n_snp_calls]
        out[f'c{i}_mol_cb'] = mol['compressed_cb'].copy()
        out[f'c{i}_mol_ub'] = mol['compressed_ub'].copy()
        out[f'c{i}_mol_pmis'] = mol['p_group_misaligned'].copy()
        out[f'c{i}_call_mol'] = sc['molecule_index'].copy()
        out[f'c{i}_call_pos'] = sc['snp_position'].copy()
        out[f'c{i}_call_base'] = sc['base_index'].copy()
        out[f'c{i}_call_p'] = sc['p_base_wrong'].copy()
    keys = list(genotypes.var2varid.items())
    out['var_chrom'] = np.asarray([k[0] for k, _ in keys], dtype=str)
    out['var_pos'] = np.asarray([k[1] for k, _ in keys], dtype=np.int64)
    out['var_base'] = np.asarray([BASES[k[2]] for k, _ in keys], dtype=np.uint8)
    out['var_row'] = np.asarray([row for _, row in keys], dtype=np.int32)
    out['betas'] = np.array(genotypes.variant_betas[:genotypes.n_variants], dtype=np.float32)
    out['genotype_names'] = np.asarray(genotypes.genotype_names, dtype=str)
    out['default_prior'] = np.float64(genotypes.default_prior)
    out['barcodes'] = np.asarray(barcode_handler.ordered_barcodes, dtype=str)
    return out


def capture(ref, calls, genotypes, barcode_handler, predict_dps=(0., 0.35), em_runs=()):
    """Runs the reference and records everything the parity tests compare."""
    D = ref.Demultiplexer
    out = inputs_as_arrays(calls, genotypes, barcode_handler)
    for flag in (False, True):
        v2snp, betas, mol_calls, bc = D.pack_calls(calls, genotypes, add_data_prior=flag)
        tag = f'pack{int(flag)}'
        out[f'{tag}_betas'] = np.array(betas)
        if not flag:
            out['pack_v2snp'] = v2snp
            out['pack_n_molecule_calls'] = np.int64(len(mol_calls))
            out['pack_bc_variant_id'] = np.array(bc['variant_id'])
            out['pack_bc_snp_id'] = np.array(bc['snp_id'])
            out['pack_bc_cb'] = np.array(bc['compressed_cb'])
            out['pack_bc_p'] = np.array(bc['p_base_wrong'])
            out['pack_bc_variant_count'] = np.array(bc['barcode_variant_count'])
            out['pack_bc_snp_count'] = np.array(bc['barcode_snp_count'])
    for i, spec in enumerate(predict_dps):
        dp, clip = spec if isinstance(spec, tuple) else (spec, 0.01)
        logits, probs = D.predict_posteriors(calls, genotypes, barcode_handler,
                                             p_genotype_clip=clip, doublet_prior=dp)
        assert logits.index.name == 'BARCODE' and list(logits.index) == list(barcode_handler.ordered_barcodes)
        out[f'predict{i}_dp'] = np.float64(dp)
        out[f'predict{i}_clip'] = np.float64(clip)
        out[f'predict{i}_logits'] = logits.values
        out[f'predict{i}_probs'] = probs.values
        out[f'predict{i}_columns'] = np.asarray(list(logits.columns), dtype=str)
    out['n_predict'] = np.int64(len(predict_dps))
    for i, run in enumerate(em_runs):
        kwargs = dict(n_iterations=run.get('n_iterations', 5), p_genotype_clip=run.get('clip', 0.01),
                      doublet_prior=run.get('dp', 0.))
        prior = run.get('prior_logits')
        stages = list(D.staged_genotype_learning(calls, genotypes, barcode_handler,
                                                 barcode_prior_logits=None if prior is None else prior.copy(),
                                                 **kwargs))
        for it, (probs_df, dbg) in enumerate(stages):
            out[f'em{i}_it{it}_logits'] = np.array(dbg['barcode_logits'])
            out[f'em{i}_it{it}_probs'] = probs_df.values.copy()
            out[f'em{i}_it{it}_addition'] = np.array(dbg['genotype_addition'])
        learnt, last_probs = D.learn_genotypes(calls, genotypes, barcode_handler,
                                               barcode_prior_logits=None if prior is None else prior.copy(),
                                               **kwargs)
        assert last_probs.index.name is None
        assert np.array_equal(last_probs.values, stages[-1][0].values)
        out[f'em{i}_learnt_betas'] = np.array(learnt.variant_betas)
        out[f'em{i}_columns'] = np.asarray(list(last_probs.columns), dtype=str)
        out[f'em{i}_n_iterations'] = np.int64(kwargs['n_iterations'])
        out[f'em{i}_clip'] = np.float64(kwargs['p_genotype_clip'])
        out[f'em{i}_dp'] = np.float64(kwargs['doublet_prior'])
        if prior is not None:
            out[f'em{i}_prior_logits'] = prior
    out['n_em'] = np.int64(len(em_runs))
    return out


def save(name, arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f'{name}: {os.path.getsize(path) / 1e6:.2f} MB, {len(arrays)} arrays')


# --------------------------------------------------------------------------- #
# fixture families
# --------------------------------------------------------------------------- #

def synthetic_bam_case(ref, ref_tests, **gen_kwargs):
    """The reference's own test inputs: generate_bam_file (tests/test_synthetic.py:106-145,
    seed 42 as in :160) scanned by the reference's count_snps (single process)."""
    np.random.seed(42)
    filename, genotypes, _ids, bc2names = ref_tests.generate_bam_file(filename='/tmp/golden_fixture.bam', **gen_kwargs)
    handler = ref.BarcodeHandler(list(bc2names))
    calls = ref.count_snps(filename, chromosome2positions=genotypes.get_chromosome2positions(),
                           barcode_handler=handler, joblib_n_jobs=1, joblib_verbosity=0)
    return calls, genotypes, handler, bc2names


def build_calls(ref, per_chrom):
    """per_chrom: {chrom: (mol_cb list, [(mol_index, pos, base_index, p), ...])}"""
    from demuxalot.snp_counter import CompressedSNPCalls
    calls = {}
    for chrom, (mol_cb, rows) in per_chrom.items():
        cc = CompressedSNPCalls()
        cc.molecules = np.zeros(len(mol_cb), dtype=cc.molecules.dtype)
        cc.molecules['compressed_cb'] = mol_cb
        cc.molecules['compressed_ub'] = np.arange(len(mol_cb))
        cc.molecules['p_group_misaligned'] = 0.01
        cc.snp_calls = np.zeros(len(rows), dtype=cc.snp_calls.dtype)
        for j, (m, pos, base, p) in enumerate(rows):
            cc.snp_calls[j] = (m, pos, base, p)
        cc.n_molecules, cc.n_snp_calls = len(mol_cb), len(rows)
        calls[chrom] = cc
    return calls


def random_small_case(ref, rng, n_genotypes, n_barcodes, n_snps, n_molecules, calls_per_mol=3,
                      zero_beta_genotype=False, multiallelic=True, unmatched=True):
    """Hand-sized random problem exercising the edge cases listed in SURVEY 8c/F3: duplicated
    (variant, barcode) molecules, a tri-allelic SNP, unmatched calls (wrong base / unknown
    position), a barcode without calls, tiny p_base_wrong products, an all-zero genotype column."""
    names = [f'D{i:02d}' for i in range(n_genotypes)]
    g = ref.ProbabilisticGenotypes(names)
    chroms = ['chrA', 'chrB']
    var2varid = {}
    snps = []
    for s in range(n_snps):
        chrom = chroms[s % 2]
        pos = 10 + 7 * s
        n_alleles = 3 if (multiallelic and s % 5 == 0) else 2
        bases = list(rng.choice(4, size=n_alleles, replace=False))
        snps.append((chrom, pos, bases))
    order = rng.permutation(sum(len(b) for _, _, b in snps))  # scrambled variant rows
    flat = [(chrom, pos, b) for chrom, pos, bases in snps for b in bases]
    for slot in order:
        chrom, pos, b = flat[slot]
        var2varid[(chrom, pos, 'ACGT'[b])] = len(var2varid)
    g.var2varid = var2varid
    betas = rng.choice([0.5, 50., 100.], size=(len(var2varid), n_genotypes)).astype('float32')
    betas *= rng.uniform(0.5, 1.5, size=betas.shape).astype('float32')
    if zero_beta_genotype:
        betas[:, -1] = 0
    g.variant_betas = betas
    barcodes = [f'BC{i:04d}-1' for i in range(n_barcodes)]
    handler = ref.BarcodeHandler(barcodes)
    per_chrom = {}
    for chrom in chroms:
        chrom_snps = [s for s in snps if s[0] == chrom]
        mol_cb = rng.integers(0, n_barcodes - 1, size=n_molecules)  # last barcode never observed
        rows = []
        for m in range(n_molecules):
            for _ in range(calls_per_mol):
                _c, pos, bases = chrom_snps[rng.integers(len(chrom_snps))]
                roll = rng.random()
                if unmatched and roll < 0.05:
                    base = [b for b in range(4) if b not in bases][0]  # base not among the variants
                elif unmatched and roll < 0.08:
                    pos, base = pos + 1, 0  # position that is not a SNP
                else:
                    base = bases[rng.integers(len(bases))]
                q = rng.integers(5, 41)
                p = np.float32(0.1 ** (0.1 * q))
                if rng.random() < 0.1:
                    p = np.float32(p * 1e-30)  # products that run towards FLT_MIN
                rows.append((m, pos, base, p))
        # molecules of one barcode hitting the same variant repeatedly
        for rep in range(6):
            _c, pos, bases = chrom_snps[0]
            rows.append((0, pos, bases[0], np.float32(0.05 + 0.01 * rep)))
        per_chrom[chrom] = (mol_cb, rows)
    return build_calls(ref, per_chrom), g, handler


def main():
    ref, ref_tests = import_reference()
    rng = np.random.default_rng(20240607)

    # F4: doublet penalties (reference demux.py:158-173; pinned by tests/test_utils.py:34-40)
    f4 = {}
    for G in (2, 3, 10, 20, 64, 128):
        for dp in (0., 0.25, 0.35, 0.5):
            f4[f'G{G}_dp{dp}'] = ref.Demultiplexer._doublet_penalties(G, dp)
    save('f4_doublet_penalties.npz', f4)

    # F3: hand-sized edge cases
    cases = [
        dict(n_genotypes=2, n_barcodes=6, n_snps=5, n_molecules=12),
        dict(n_genotypes=3, n_barcodes=17, n_snps=11, n_molecules=40, zero_beta_genotype=True),
        dict(n_genotypes=5, n_barcodes=40, n_snps=23, n_molecules=150),
        dict(n_genotypes=9, n_barcodes=33, n_snps=30, n_molecules=200, calls_per_mol=5),
        dict(n_genotypes=70, n_barcodes=24, n_snps=40, n_molecules=300, calls_per_mol=4),
    ]
    for i, kw in enumerate(cases):
        calls, g, handler = random_small_case(ref, rng, **kw)
        G, B = g.n_genotypes, handler.n_barcodes
        K2 = G * (G + 1) // 2
        prior_s = (rng.normal(size=(B, G)) * 3).astype('float32')
        prior_d = (rng.normal(size=(B, K2)) * 3).astype('float32')
        arrays = capture(
            ref, calls, g, handler,
            predict_dps=(0., 0.35, (0.2, 0.05), (0., 0.)),
            em_runs=(dict(n_iterations=4, dp=0.), dict(n_iterations=3, dp=0.25, clip=0.02),
                     dict(n_iterations=3, dp=0., prior_logits=prior_s),
                     dict(n_iterations=2, dp=0.4, prior_logits=prior_d)),
        )
        save(f'f3_small_{i}.npz', arrays)

    # F2: reduced reference synthetic test (closest to BASELINE.json configs[0] wording)
    calls, g, handler, _ = synthetic_bam_case(ref, ref_tests, n_genotypes=4, n_barcodes=200, mutation_prob=0.12)
    arrays = capture(ref, calls, g, handler, predict_dps=(0., 0.35),
                     em_runs=(dict(n_iterations=5, dp=0.), dict(n_iterations=3, dp=0.25)))
    print('F2: G', g.n_genotypes, 'V', g.n_variants, 'B', handler.n_barcodes,
          'M', int(arrays['pack_n_molecule_calls']), 'N', len(arrays['pack_bc_cb']))
    save('f2_synthetic_g4.npz', arrays)

    # F1: the reference's test_synthetic defaults (tests/test_synthetic.py:106-115,160)
    calls, g, handler, bc2names = synthetic_bam_case(ref, ref_tests)
    B, G = handler.n_barcodes, g.n_genotypes
    # the labelled-barcodes scenario of tests/test_synthetic.py:200-239 (+100 on known singlets)
    prior = np.zeros((B, G), dtype='float32')
    label_p = rng.random(B)
    for (bc, donors), p in zip(bc2names.items(), label_p):
        if len(donors) == 1 and p < 0.2:
            prior[handler.barcode2index[bc], g.genotype_names.index(donors[0])] += 100.
    empty = g.clone()
    empty.variant_betas[:] = 0
    arrays = capture(ref, calls, g, handler, predict_dps=(0., 0.25, 0.35),
                     em_runs=(dict(n_iterations=5, dp=0.), dict(n_iterations=2, dp=0.25)))
    print('F1: G', G, 'V', g.n_variants, 'S', int(arrays['pack_v2snp'].max()) + 1, 'B', B,
          'M', int(arrays['pack_n_molecule_calls']), 'N', len(arrays['pack_bc_cb']))
    save('f1_synthetic_default.npz', arrays)
    # same calls, empty genotypes + prior logits: only outputs are stored (inputs = F1's)
    arr2 = capture(ref, calls, empty, handler, predict_dps=(),
                   em_runs=(dict(n_iterations=5, dp=0., prior_logits=prior),))
    keep = {k: v for k, v in arr2.items() if k.startswith(('em', 'n_em', 'pack1', 'pack0'))}
    save('f1_from_assignment.npz', keep)


def reference_inputs(ref, fx):
    """The reference's own objects rebuilt from the stored inputs of an existing fixture (inputs_as_arrays)."""
    calls = {}
    for i, chrom in enumerate(fx['chroms']):
        c = ref.snp_counter.CompressedSNPCalls()
        c.molecules = np.zeros(len(fx[f'c{i}_mol_cb']), dtype=c.molecules.dtype)
        c.molecules['compressed_cb'] = fx[f'c{i}_mol_cb']
        c.molecules['compressed_ub'] = fx[f'c{i}_mol_ub']
        c.molecules['p_group_misaligned'] = fx[f'c{i}_mol_pmis']
        c.snp_calls = np.zeros(len(fx[f'c{i}_call_mol']), dtype=c.snp_calls.dtype)
        c.snp_calls['molecule_index'] = fx[f'c{i}_call_mol']
        c.snp_calls['snp_position'] = fx[f'c{i}_call_pos']
        c.snp_calls['base_index'] = fx[f'c{i}_call_base']
        c.snp_calls['p_base_wrong'] = fx[f'c{i}_call_p']
        c.n_molecules, c.n_snp_calls = len(c.molecules), len(c.snp_calls)
        calls[str(chrom)] = c
    g = ref.ProbabilisticGenotypes([str(n) for n in fx['genotype_names']], default_prior=float(fx['default_prior']))
    g.var2varid = {(str(c), int(p), 'ACGTN'[int(b)]): int(r)
                   for c, p, b, r in zip(fx['var_chrom'], fx['var_pos'], fx['var_base'], fx['var_row'])}
    g.variant_betas = np.array(fx['betas'], dtype=np.float32)
    handler = ref.BarcodeHandler([str(b) for b in fx['barcodes']])
    assert handler.ordered_barcodes == [str(b) for b in fx['barcodes']]
    return calls, g, handler


def aggregate_on_snps_cases():
    """F7: Demultiplexer.aggregate_on_snps = True (demux.py:204-244) on the inputs of existing fixtures; only the
    outputs are stored (float64 logits / posteriors, float32 additions and learnt betas), the inputs are those of
    the named fixture."""
    ref, _ = import_reference()
    D = ref.Demultiplexer
    rng = np.random.default_rng(77)
    for name, predict_dps, em_runs in (
            ('f3_small_2.npz', (0., 0.35), (dict(n_iterations=3, dp=0.), dict(n_iterations=2, dp=0.3, prior=True))),
            ('f3_small_4.npz', (0., 0.2), (dict(n_iterations=2, dp=0.),)),
            ('f2_synthetic_g4.npz', (0., 0.35), (dict(n_iterations=3, dp=0.), dict(n_iterations=2, dp=0.25))),
            ('f1_synthetic_default.npz', (0., 0.25), (dict(n_iterations=3, dp=0., prior=True),)),
    ):
        with np.load(os.path.join(HERE, name), allow_pickle=False) as z:
            fx = {k: z[k] for k in z.files}
        calls, g, handler = reference_inputs(ref, fx)
        out = {'inputs_of': np.asarray(name)}
        D.aggregate_on_snps = True
        try:
            for i, dp in enumerate(predict_dps):
                logits, probs = D.predict_posteriors(calls, g, handler, doublet_prior=dp)
                assert logits.values.dtype == np.float64 and probs.values.dtype == np.float64
                out[f'predict{i}_dp'] = np.float64(dp)
                out[f'predict{i}_logits'] = logits.values
                out[f'predict{i}_probs'] = probs.values
                out[f'predict{i}_columns'] = np.asarray(list(logits.columns), dtype=str)
            out['n_predict'] = np.int64(len(predict_dps))
            for i, run in enumerate(em_runs):
                K = len(D._doublet_penalties(g.n_genotypes, run['dp']))
                prior = (rng.normal(size=(handler.n_barcodes, K)) * 3).astype('float32') if run.get('prior') else None
                kwargs = dict(n_iterations=run['n_iterations'], doublet_prior=run['dp'])
                stages = list(D.staged_genotype_learning(calls, g, handler,
                                                         barcode_prior_logits=None if prior is None else prior.copy(), **kwargs))
                for it, (probs_df, dbg) in enumerate(stages):
                    out[f'em{i}_it{it}_logits'] = np.array(dbg['barcode_logits'])
                    out[f'em{i}_it{it}_probs'] = probs_df.values.copy()
                    out[f'em{i}_it{it}_addition'] = np.array(dbg['genotype_addition'])
                learnt, last = D.learn_genotypes(calls, g, handler, barcode_prior_logits=None if prior is None else prior.copy(),
                                                 **kwargs)
                assert np.array_equal(last.values, stages[-1][0].values)
                out[f'em{i}_learnt_betas'] = np.array(learnt.variant_betas)
                out[f'em{i}_n_iterations'] = np.int64(run['n_iterations'])
                out[f'em{i}_dp'] = np.float64(run['dp'])
                if prior is not None:
                    out[f'em{i}_prior_logits'] = prior
            out['n_em'] = np.int64(len(em_runs))
        finally:
            D.aggregate_on_snps = False
        save('f7_aggregate_' + name.split('_', 1)[1], out)


def synthetic_generator_case():
    """F5: a mid-size problem from the benchmark generator (demuxalot_amd/synth.py) pushed
    through the REFERENCE's full entry points, so that the generator's object form, the host repack and
    the kernels are pinned at a size with thousands of calls per variant. Inputs are regenerated by the
    deterministic generator at test time (a hash of them is stored); only outputs are saved."""
    import hashlib
    ref, _ = import_reference()
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from demuxalot_amd import synth
    problem = synth.generate(2000, 5000, 16, calls_per_barcode=300, doublets=True, seed=4242)
    calls_amd, genotypes_amd, handler_amd = synth.as_objects(problem)
    # rebuild the same inputs as reference objects
    from demuxalot.snp_counter import CompressedSNPCalls
    calls = {}
    for chrom, c in calls_amd.items():
        cc = CompressedSNPCalls()
        cc.molecules, cc.snp_calls = c.molecules.copy(), c.snp_calls.copy()
        cc.n_molecules, cc.n_snp_calls = c.n_molecules, c.n_snp_calls
        calls[chrom] = cc
    g = ref.ProbabilisticGenotypes(genotypes_amd.genotype_names)
    g.var2varid = dict(genotypes_amd.var2varid)
    g.variant_betas = genotypes_amd.variant_betas.copy()
    handler = ref.BarcodeHandler(handler_amd.ordered_barcodes)
    h = hashlib.sha256()
    for arr in (problem.variant_id, problem.compressed_cb, problem.p_base_wrong, problem.raw_betas):
        h.update(np.ascontiguousarray(arr).tobytes())
    out = {'input_sha256': np.asarray(h.hexdigest())}
    D = ref.Demultiplexer
    v2snp, betas, _mol, bc = D.pack_calls(calls, g, add_data_prior=True)
    assert np.array_equal(bc['variant_id'], problem.variant_id) and np.array_equal(bc['compressed_cb'], problem.compressed_cb)
    assert np.array_equal(bc['p_base_wrong'], problem.p_base_wrong)
    out['pack1_betas'] = np.array(betas)
    for i, dp in enumerate((0., 0.3)):
        logits, probs = D.predict_posteriors(calls, g, handler, doublet_prior=dp)
        out[f'predict{i}_dp'] = np.float64(dp)
        out[f'predict{i}_logits'] = logits.values
        out[f'predict{i}_probs'] = probs.values
    learnt, probs = D.learn_genotypes(calls, g, handler, n_iterations=3)
    out['em_learnt_betas'] = np.array(learnt.variant_betas)
    out['em_probs'] = probs.values
    print('F5: N', problem.n_calls, 'max calls per variant', np.bincount(problem.variant_id).max())
    save('f5_generator_2k_5k_16.npz', out)


def shipped_example_case():
    """F6: the reference's CI example (examples/1-plain_demultiplexing.py on examples/example_data):
    the reference's own add_vcf / BarcodeHandler.from_file / count_snps (through the stand-in pysam that
    decodes the shipped BAM and VCF) and learn_genotypes(doublet_prior=0.25)."""
    import shutil
    ref, _ = import_reference()
    data = f'{REFERENCE}/examples/example_data'
    genotypes = ref.ProbabilisticGenotypes(genotype_names=['Donor01', 'Donor02', 'Donor03', 'Donor04'])
    genotypes.add_vcf(f'{data}/test_genotypes.vcf')
    handler = ref.BarcodeHandler.from_file(f'{data}/test_barcodes.csv')
    calls = ref.count_snps(bamfile_location=f'{data}/test_bamfile.bam',
                           chromosome2positions=genotypes.get_chromosome2positions(), barcode_handler=handler,
                           joblib_n_jobs=1, joblib_verbosity=0)
    arrays = capture(ref, calls, genotypes, handler, predict_dps=(0.35, 0.),
                     em_runs=(dict(n_iterations=5, dp=0.25), dict(n_iterations=5, dp=0.)))
    print('F6: V', genotypes.n_variants, 'B', handler.n_barcodes, 'M', int(arrays['pack_n_molecule_calls']),
          'N', len(arrays['pack_bc_cb']))
    save('f6_shipped_example.npz', arrays)
    # the two small text inputs of the example are data files of the reference's CI run: kept as fixtures
    # so that add_vcf / BarcodeHandler.from_file of this package can be checked against the captured genotypes
    shutil.copy(f'{data}/test_genotypes.vcf', os.path.join(HERE, 'example_genotypes.vcf'))
    shutil.copy(f'{data}/test_barcodes.csv', os.path.join(HERE, 'example_barcodes.csv'))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'f6':
        shipped_example_case()
    elif len(sys.argv) > 1 and sys.argv[1] == 'f5':
        synthetic_generator_case()
    elif len(sys.argv) > 1 and sys.argv[1] == 'f7':
        aggregate_on_snps_cases()
    else:
        main()
        synthetic_generator_case()
        shipped_example_case()
        aggregate_on_snps_cases()  # reads the inputs of the fixtures written above
