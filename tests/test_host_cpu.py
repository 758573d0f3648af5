"""CPU-side checks of the product: the C-ABI library loads and exports every declared symbol,
the host repack (C++ dmx_pack_calls_host + numpy prior betas) reproduces the reference's
pack_calls bit for bit on the golden fixtures, and the mirrored containers behave like the
reference's. No GPU compute is attempted here."""
import os
import re

import numpy as np
import pytest

from tests import fixture_io as fio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    """Per header: demux_hip.h is what a front-end binds, demux_hip_debug.h the test / tuning surface.  Every
    declared symbol is exported and bound with a signature, nothing is bound that no header declares, no test or debug name sits
    in the public header, and the library exports no dmx_* symbol that neither header declares."""
    import subprocess
    from demuxalot_amd import _lib
    lib = _lib.load()
    both = set()
    for header_name, signatures, at_least in (('demux_hip.h', _lib.SIGNATURES, 25), ('demux_hip_debug.h', _lib.DEBUG_SIGNATURES, 10)):
        header = open(os.path.join(ROOT, 'include', header_name)).read()
        declared = set(re.findall(r'^(?:int|const char \*)\s*(dmx_[a-z0-9_]+)\s*\(', header, flags=re.M))
        assert len(declared) >= at_least
        for name in sorted(declared):
            assert hasattr(lib, name), f'{name} is declared in include/{header_name} but not exported'
        assert declared == set(signatures), (header_name, declared ^ set(signatures))
        assert not (declared & both), declared & both
        both |= declared
    public = set(_lib.SIGNATURES)
    assert not [n for n in public if n.startswith(('dmx_test_', 'dmx_debug_')) or 'emulated' in n], 'test / debug names in the public header'
    assert len(public) <= 64, len(public)
    nm = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r'\bT (dmx_[a-z0-9_]+)$', nm, flags=re.M))
    assert exported == both, exported ^ both
    assert lib.dmx_version().decode().startswith('demux_hip')


def test_no_gpu_means_loud_failure():
    from demuxalot_amd import _lib
    from demuxalot_amd.device import DeviceContext
    if _lib.device_count() > 0:
        pytest.skip('a GPU is visible')
    with pytest.raises(_lib.DemuxHipError, match='no HIP device'):
        DeviceContext(0)


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under demuxalot_amd/ may reference it."""
    pkg = os.path.join(ROOT, 'demuxalot_amd')
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.cpp', '.hip', '.h')):
                text = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in text and 'from oracle' not in text and 'demux_oracle' not in text, f


@pytest.mark.parametrize('name', fio.SMALL + fio.SYNTH)
def test_pack_calls_matches_reference(name):
    from demuxalot_amd import Demultiplexer
    fx = fio.load(name)
    calls, genotypes, _handler = fio.product_inputs(fx)
    for flag in (False, True):
        v2snp, betas, molecule_calls, bc = Demultiplexer.pack_calls(calls, genotypes, add_data_prior=flag)
        assert v2snp.dtype == np.int32 and np.array_equal(v2snp, fx['pack_v2snp'])
        fio.assert_bitwise(betas, fx[f'pack{int(flag)}_betas'], 'prior betas')
        assert not betas.flags.writeable
        assert len(molecule_calls) == int(fx['pack_n_molecule_calls'])
        assert molecule_calls.dtype.names == ('variant_id', 'snp_id', 'compressed_cb', 'molecule_id',
                                              'p_base_wrong', 'p_molecule_aligned_wrong')
        assert np.array_equal(bc['variant_id'], fx['pack_bc_variant_id'])
        assert np.array_equal(bc['snp_id'], fx['pack_bc_snp_id'])
        assert np.array_equal(bc['compressed_cb'], fx['pack_bc_cb'])
        fio.assert_bitwise(bc['p_base_wrong'], fx['pack_bc_p'], 'p_base_wrong')
        assert np.array_equal(bc['barcode_variant_count'], fx['pack_bc_variant_count'])
        assert np.array_equal(bc['barcode_snp_count'], fx['pack_bc_snp_count'])
        assert bc['barcode_snp_count'].dtype == np.float64 and bc['barcode_variant_count'].dtype == np.int64


def test_pack_asserts_like_reference():
    from demuxalot_amd import CompressedSNPCalls, Demultiplexer, ProbabilisticGenotypes
    g = ProbabilisticGenotypes(['A', 'B'])
    g.var2varid = {('chr1', 5, 'A'): 0, ('chr1', 5, 'C'): 1}
    g.variant_betas = np.ones((2, 2), dtype=np.float32)
    ok = CompressedSNPCalls.from_arrays([0], [0], [5], [0], [0.01])
    stray = CompressedSNPCalls.from_arrays([0], [0], [7], [1], [0.01])
    # a chromosome that carries calls but has no variant in the genotypes trips the reference's assert (demux.py:359)
    with pytest.raises(AssertionError):
        Demultiplexer.pack_calls({'chr1': ok, 'chrX': stray}, g, add_data_prior=False)
    # ... an empty container on such a chromosome does not
    Demultiplexer.pack_calls({'chr1': ok, 'chrX': CompressedSNPCalls.from_arrays([], [], [], [], [])}, g, False)
    g.variant_betas = -np.ones((2, 2), dtype=np.float32)
    with pytest.raises(AssertionError, match='negative betas'):
        Demultiplexer.pack_calls({'chr1': ok}, g, add_data_prior=False)


def test_doublet_penalties_identity_and_golden():
    """tests/test_utils.py:34-40 of the reference + the captured table."""
    from scipy.special import softmax
    from demuxalot_amd import Demultiplexer
    for n_genotypes in [2, 3, 10]:
        for doublet_prob in [0., 0.25, 0.5]:
            pen = Demultiplexer._doublet_penalties(n_genotypes=n_genotypes, doublet_prior=doublet_prob)
            assert np.allclose(softmax(pen)[:n_genotypes].sum(), 1 - doublet_prob)
    for key, ref in fio.load('f4_doublet_penalties.npz').items():
        G, dp = key.split('_dp')
        fio.assert_bitwise(Demultiplexer._doublet_penalties(int(G[1:]), float(dp)), ref, key)
    with pytest.raises(AssertionError):
        Demultiplexer._doublet_penalties(4, 1.0)


def test_genotypes_store_and_roundtrip(tmp_path):
    """tests/test_synthetic.py:241-260 of the reference (save_betas -> add_prior_betas)."""
    from demuxalot_amd import ProbabilisticGenotypes
    fx = fio.load('f2_synthetic_g4.npz')
    _calls, genotypes, _h = fio.product_inputs(fx)
    assert genotypes.n_variants == len(fx['var_row']) and genotypes.n_genotypes == 4
    assert not genotypes.get_betas().flags.writeable
    path = tmp_path / 'genotypes.parquet'
    genotypes.save_betas(path)
    again = ProbabilisticGenotypes(genotype_names=genotypes.genotype_names, default_prior=genotypes.default_prior)
    again.add_prior_betas(path)
    assert set(again.var2varid) == set(genotypes.var2varid)
    for variant, row in genotypes.var2varid.items():
        assert np.allclose(genotypes.variant_betas[row], again.variant_betas[again.var2varid[variant]])
    learnt = genotypes._with_betas(genotypes.get_betas() + np.float32(1))
    assert learnt is not genotypes and learnt.variant_betas.shape == (genotypes.n_variants, 4)
    with pytest.raises(AssertionError):
        ProbabilisticGenotypes(['b', 'a'])
    with pytest.raises(AssertionError):
        genotypes._with_betas(np.zeros((3, 4), dtype=np.float32))


def test_add_vcf_plain_text(tmp_path):
    """Import rules of the reference's add_vcf (genotypes.py:112-168) on a hand-checked file."""
    from demuxalot_amd import ProbabilisticGenotypes
    vcf = tmp_path / 'toy.vcf'
    vcf.write_text('\n'.join([
        '##fileformat=VCFv4.2',
        '#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tD1\tD2\tD3\tExtra',
        'chr1\t10\ts0\tG\tT\t.\t.\t.\tGT\t0/0\t1/1\t0/1\t1/1',           # plain biallelic
        'chr1\t20\ts1\tA\tC,G\t.\t.\t.\tGT:DP\t0|2:5\t./.:0\t1/1:7\t0/0:1',  # tri-allelic, phased, missing donor
        'chr1\t30\ts2\tAT\tA\t.\t.\t.\tGT\t0/0\t1/1\t0/0\t0/0',           # indel: skipped as non-SNP
        'chr2\t5\ts3\tC\tT\t.\t.\t.\tGT\t0/1\t./.\t./.\t1/1',             # one genotyped donor of ours: skipped, rows stay
        'chr2\t7\ts4\tC\tT\t.\t.\t.\tGT\t./1\t0/0\t./.\t0/0',             # half-missing call
    ]) + '\n')
    g = ProbabilisticGenotypes(['D1', 'D2', 'D3'])
    with pytest.warns(UserWarning, match='will not be imported'):
        g.add_vcf(str(vcf))
    assert list(g.var2varid) == [('chr1', 9, 'G'), ('chr1', 9, 'T'), ('chr1', 19, 'A'), ('chr1', 19, 'C'), ('chr1', 19, 'G'),
                                 ('chr2', 4, 'C'), ('chr2', 4, 'T'), ('chr2', 6, 'C'), ('chr2', 6, 'T')]
    b = g.get_betas()
    assert np.array_equal(b[0], [100, 0, 50]) and np.array_equal(b[1], [0, 100, 50])
    # tri-allelic: D1 = A/G, D3 = C/C, D2 not provided -> 0.1 x mean of the provided donors per allele
    assert np.allclose(b[2], [50, 2.5, 0]) and np.allclose(b[3], [0, 5.0, 100]) and np.allclose(b[4], [50, 2.5, 0])
    assert np.array_equal(b[5:7], np.zeros((2, 3)))                      # skipped record keeps zero rows
    assert np.allclose(b[7], [0, 100, 5.0]) and np.allclose(b[8], [50, 0, 2.5])   # './1' assigns only half
    assert list(g.get_snp_ids_for_variants()) == [0, 0, 1, 1, 1, 2, 2, 3, 3]
    assert {k: list(v) for k, v in g.get_chromosome2positions().items()} == {'chr1': [9, 19], 'chr2': [4, 6]}


def test_shipped_example_inputs_match_reference_import():
    """The reference's CI example data: add_vcf (own text parser) and BarcodeHandler.from_file of this
    package against the genotypes / barcode order the reference produced from the same two files (F6)."""
    from demuxalot_amd import BarcodeHandler, ProbabilisticGenotypes
    fx = fio.load('f6_shipped_example.npz')
    g = ProbabilisticGenotypes(genotype_names=['Donor01', 'Donor02', 'Donor03', 'Donor04'])
    g.add_vcf(os.path.join(fio.GOLDEN, 'example_genotypes.vcf'))
    want_keys = [(str(c), int(p), 'ACGTN'[int(b)]) for c, p, b in zip(fx['var_chrom'], fx['var_pos'], fx['var_base'])]
    assert list(g.var2varid) == want_keys and list(g.var2varid.values()) == list(fx['var_row'])
    fio.assert_bitwise(np.array(g.get_betas()), fx['betas'], 'betas imported from the VCF')
    handler = BarcodeHandler.from_file(os.path.join(fio.GOLDEN, 'example_barcodes.csv'))
    assert handler.ordered_barcodes == [str(b) for b in fx['barcodes']]


def test_barcode_handler_and_container():
    from demuxalot_amd import BarcodeHandler, CompressedSNPCalls
    h = BarcodeHandler(['T-1', 'A-1', 'G-1'])
    assert h.ordered_barcodes == ['A-1', 'G-1', 'T-1'] and h.n_barcodes == 3 and h.barcode2index['G-1'] == 1
    with pytest.raises(AssertionError):
        BarcodeHandler(['A', 'A'])
    c = CompressedSNPCalls.from_arrays([3, 1], [0, 0, 0, 1], [10, 11, 15, 10], [0, 3, 2, 1], [0.1, 0.2, 0.3, 0.1])
    assert (c.n_molecules, c.n_snp_calls) == (2, 4) and c.snp_calls.dtype.itemsize == 13 and c.molecules.dtype.itemsize == 12
    assert list(c.snp_calls['base_index']) == [0, 3, 2, 1] and list(c.molecules['compressed_cb']) == [3, 1]
    empty = CompressedSNPCalls()
    assert (empty.n_molecules, empty.n_snp_calls) == (0, 0)


def test_genotype_clone_is_independent():
    """clone / _with_betas (genotypes.py:327-334, 360-361): nothing is shared that a caller could mutate."""
    from demuxalot_amd import ProbabilisticGenotypes
    g = ProbabilisticGenotypes(['A', 'B'])
    g.var2varid = {('chr1', 5, 'A'): 0, ('chr1', 5, 'C'): 1}
    g.variant_betas = np.array([[1, 2], [3, 4]], dtype=np.float32)
    g.extra = {'note': [1, 2]}
    c = g.clone()
    c.var2varid[('chr2', 1, 'G')] = 2
    c.variant_betas[0, 0] = 9
    c.genotype_names.append('C')
    c.extra['note'].append(3)
    assert len(g.var2varid) == 2 and g.variant_betas[0, 0] == 1 and g.genotype_names == ['A', 'B'] and g.extra == {'note': [1, 2]}
    w = g._with_betas(np.full((2, 2), 7, dtype=np.float32))
    assert (w.variant_betas == 7).all() and (g.variant_betas != 7).all() and w.var2varid == g.var2varid
    assert w.var2varid is not g.var2varid and type(w) is type(g)


def test_variant_key_columns_follow_the_dict_walk():
    """The vectorised readers of var2varid (genotypes.variant_columns: _variant_keys, get_snp_ids_for_variants) against
    a plain walk of the dict: chromosomes and SNPs numbered in order of first appearance (genotypes.py:56-66), rows
    in any order, equal positions on different chromosomes."""
    from demuxalot_amd import ProbabilisticGenotypes
    from demuxalot_amd.demux import _variant_keys
    rng = np.random.default_rng(5)
    n = 3000
    chroms = rng.choice(['chr2', 'chr10', 'chrX', '7', 'MT'], size=n)
    positions = rng.integers(1, 400, size=n)  # many (chrom, pos) shared by several bases, positions shared across chromosomes
    bases = rng.choice(list('ACGTN'), size=n)
    keys = list(dict.fromkeys(zip(chroms.tolist(), positions.tolist(), bases.tolist())))
    rows = rng.permutation(len(keys))
    g = ProbabilisticGenotypes(['a', 'b'])
    g.var2varid = {k: int(r) for k, r in zip(keys, rows)}
    g.variant_betas = np.zeros((len(keys), 2), dtype=np.float32)
    chrom_index, snp_index = {}, {}
    want_chrom, want_pos, want_base, want_snp = (np.zeros(len(keys), dtype=np.int64) for _ in range(4))
    for (chrom, pos, base), row in g.var2varid.items():
        want_chrom[row] = chrom_index.setdefault(chrom, len(chrom_index))
        want_pos[row] = pos
        want_base[row] = 'ACGTN'.index(base)
        want_snp[row] = snp_index.setdefault((chrom, pos), len(snp_index))
    (var_chrom, var_pos, var_base), got_index = _variant_keys(g)
    assert got_index == chrom_index and list(got_index) == list(chrom_index)
    assert np.array_equal(var_chrom, want_chrom) and np.array_equal(var_pos, want_pos) and np.array_equal(var_base, want_base)
    assert var_chrom.dtype == np.int32 and var_pos.dtype == np.int32 and var_base.dtype == np.uint8
    assert np.array_equal(g.get_snp_ids_for_variants(), want_snp)
    # what the fast path cannot represent goes through the dict walk: same answers or the same errors
    g.var2varid[('chr2', 2 ** 40, 'A')] = len(keys)
    g.variant_betas = np.zeros((len(keys) + 1, 2), dtype=np.float32)
    snp = g.get_snp_ids_for_variants()
    assert snp[len(keys)] == len(snp_index) and np.array_equal(snp[:len(keys)], want_snp)
    del g.var2varid[('chr2', 2 ** 40, 'A')]
    g.variant_betas = np.zeros((len(keys), 2), dtype=np.float32)
    g.var2varid[keys[0]] = int(rows[1])  # a row twice, another one missing
    with pytest.raises(AssertionError):
        _variant_keys(g)
    g.var2varid[keys[0]] = int(rows[0])
    g.var2varid[('chr2', 5, 'AC')] = len(keys)  # not a single letter
    g.variant_betas = np.zeros((len(keys) + 1, 2), dtype=np.float32)
    with pytest.raises(KeyError):
        _variant_keys(g)


def test_content_hash_sees_every_byte_and_does_not_depend_on_the_thread_count():
    """dmx_hash_host keys the resident packed problem (demuxalot_amd/demux.py: _pack_on_device): any single edited byte must
    change it, the same bytes in another array must not, and the number of hashing threads must not matter."""
    import ctypes
    from demuxalot_amd import _lib
    from demuxalot_amd.demux import _content_hash
    lib = _lib.load()
    rng = np.random.default_rng(5)
    data = rng.integers(0, 256, size=9_000_001, dtype=np.uint8)   # several 1 MiB chunks and a ragged tail
    want = _content_hash(data)
    assert _content_hash(data.copy()) == want
    for threads in (1, 2, 3, 7, 16):
        out = ctypes.c_uint64(0)
        _lib.check(lib.dmx_hash_host(data.ctypes.data, data.nbytes, threads, ctypes.byref(out)))
        assert out.value == want, threads
    for position in (0, 1, 31, 32, 33, 4_500_000, (1 << 20) - 1, 1 << 20, len(data) - 2, len(data) - 1):
        edited = data.copy()
        edited[position] ^= 1
        assert _content_hash(edited) != want, position
    assert _content_hash(data[:-1]) != want and _content_hash(np.concatenate([data, np.zeros(1, np.uint8)])) != want
    assert _content_hash(data[:0]) == _content_hash(np.zeros(0, np.float32))
    records = np.zeros(1000, dtype=[('a', 'i4'), ('b', 'f4'), ('c', 'u1')])   # a record array with padding-free odd item size
    h0 = _content_hash(records)
    records['c'][517] = 1
    assert _content_hash(records) != h0
    out = ctypes.c_uint64(0)
    assert lib.dmx_hash_host(None, 8, 0, ctypes.byref(out)) != 0 and lib.dmx_hash_host(data.ctypes.data, -1, 0, ctypes.byref(out)) != 0


def test_variant_key_cache_is_dropped_by_the_mutators_and_sees_an_edit_in_the_middle():
    from demuxalot_amd import ProbabilisticGenotypes
    from demuxalot_amd.demux import _var2varid_fingerprint
    genotypes = ProbabilisticGenotypes(['a', 'b', 'c'])
    for pos in range(5000):
        genotypes.get_variant_id('chr1', pos, 'ACGT'[pos % 4])
    genotypes._amd_variant_keys = ('stale',)
    genotypes.get_variant_id('chr1', 17, 'A' if 17 % 4 else 'C')      # a new variant: the kept arrays go
    assert getattr(genotypes, '_amd_variant_keys', None) is None
    genotypes._amd_variant_keys = ('stale',)
    genotypes.get_variant_id('chr1', 0, 'A')                          # an existing one: nothing changed, nothing dropped
    assert genotypes._amd_variant_keys == ('stale',)
    genotypes.invalidate()
    assert getattr(genotypes, '_amd_variant_keys', None) is None
    # entries re-pointed in the middle of the mapping at unchanged length and ends: a strided sample of the items is in the fingerprint
    before = _var2varid_fingerprint(genotypes.var2varid)
    keys = list(genotypes.var2varid)
    stride = max(1, len(keys) // 61)
    a, b = keys[stride * 30], keys[stride * 31]
    genotypes.var2varid[a], genotypes.var2varid[b] = genotypes.var2varid[b], genotypes.var2varid[a]
    assert _var2varid_fingerprint(genotypes.var2varid) != before
