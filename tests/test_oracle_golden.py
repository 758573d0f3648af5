"""Pins the CPU oracle (oracle/demux_oracle.py) against the golden vectors captured from
the imported reference (tests/golden/make_fixtures.py). CPU only."""
import numpy as np
import pytest

from tests import fixture_io as fio


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32 if a.dtype == np.float32 else np.uint64)


def assert_bitwise(a, b, what):
    assert a.shape == b.shape and a.dtype == b.dtype, (what, a.shape, b.shape, a.dtype, b.dtype)
    assert np.array_equal(_bits(a), _bits(b)), f'{what}: {(a != b).sum()} of {a.size} differ, max |d|={np.abs(a - b).max()}'


def test_doublet_penalties(oracle):
    fx = fio.load('f4_doublet_penalties.npz')
    for key, ref in fx.items():
        G, dp = key.split('_dp')
        got = oracle.doublet_penalties(int(G[1:]), float(dp))
        assert_bitwise(got, ref, key)
    # the identity the reference tests (tests/test_utils.py:34-40)
    for G in (2, 3, 10):
        for dp in (0., 0.25, 0.5):
            pen = oracle.doublet_penalties(G, dp).astype(np.float64)
            w = np.exp(pen) / np.exp(pen).sum()
            assert np.allclose(w[:G].sum(), 1 - dp)


@pytest.mark.parametrize('name', fio.SMALL + fio.SYNTH)
def test_pack_matches_reference(oracle, name):
    fx = fio.load(name)
    for flag in (False, True):
        packed = oracle.pack(fio.oracle_calls(fx), fio.oracle_geno(fx), add_data_prior=flag)
        assert np.array_equal(packed['v2snp'], fx['pack_v2snp'])
        assert_bitwise(packed['betas'], fx[f'pack{int(flag)}_betas'], 'prior betas')
        assert len(packed['mol_variant']) == int(fx['pack_n_molecule_calls'])
        assert np.array_equal(packed['variant_id'], fx['pack_bc_variant_id'])
        assert np.array_equal(packed['compressed_cb'], fx['pack_bc_cb'])
        assert_bitwise(packed['p_base_wrong'], fx['pack_bc_p'], 'p_base_wrong products')
        assert np.array_equal(packed['barcode_variant_count'], fx['pack_bc_variant_count'])


@pytest.mark.parametrize('impl', ['numpy', 'npsimd'])
@pytest.mark.parametrize('name', fio.SMALL + fio.SYNTH)
def test_predict_matches_reference(oracle, name, impl):
    fx = fio.load(name)
    packed = oracle.pack(fio.oracle_calls(fx), fio.oracle_geno(fx), add_data_prior=False)
    B = len(fx['barcodes'])
    names = [str(s) for s in fx['genotype_names']]
    for i in range(int(fx['n_predict'])):
        dp, clip = float(fx[f'predict{i}_dp']), float(fx[f'predict{i}_clip'])
        if name.startswith('f1') and impl == 'npsimd' and dp == 0.25:
            continue  # keep the CPU suite short; dp=.35 covers the doublet path
        logits, probs, _ = oracle.predict(packed, B, clip, dp, impl=impl)
        assert_bitwise(logits, fx[f'predict{i}_logits'], f'logits dp={dp}')
        assert_bitwise(probs, fx[f'predict{i}_probs'], f'probs dp={dp}')
        assert oracle.option_names(names, dp) == [str(s) for s in fx[f'predict{i}_columns']]


@pytest.mark.parametrize('impl', ['numpy', 'npsimd'])
@pytest.mark.parametrize('name', fio.SMALL + fio.SYNTH)
def test_em_matches_reference(oracle, name, impl):
    fx = fio.load(name)
    packed = oracle.pack(fio.oracle_calls(fx), fio.oracle_geno(fx), add_data_prior=True)
    B = len(fx['barcodes'])
    for i in range(int(fx['n_em'])):
        n_it = int(fx[f'em{i}_n_iterations'])
        prior = fx.get(f'em{i}_prior_logits')
        hist = oracle.em(packed, B, n_it, float(fx[f'em{i}_clip']), float(fx[f'em{i}_dp']),
                         prior_logits=prior, impl=impl)
        for it, rec in enumerate(hist):
            assert_bitwise(rec['logits'], fx[f'em{i}_it{it}_logits'], f'run {i} it {it} logits')
            assert_bitwise(rec['probs'], fx[f'em{i}_it{it}_probs'], f'run {i} it {it} probs')
            assert_bitwise(rec['addition'], fx[f'em{i}_it{it}_addition'], f'run {i} it {it} addition')
        learnt = fx['betas'] + hist[-1]['addition']
        assert_bitwise(learnt, fx[f'em{i}_learnt_betas'], 'learnt betas')


def test_em_from_assignment(oracle):
    """tests/test_synthetic.py:200-239 scenario: empty genotypes + prior logits."""
    fx = fio.load('f1_synthetic_default.npz')
    out = fio.load('f1_from_assignment.npz')
    geno = fio.oracle_geno(fx, betas=np.zeros_like(fx['betas']))
    packed = oracle.pack(fio.oracle_calls(fx), geno, add_data_prior=True)
    assert_bitwise(packed['betas'], out['pack1_betas'], 'prior betas (empty genotypes)')
    hist = oracle.em(packed, len(fx['barcodes']), int(out['em0_n_iterations']), float(out['em0_clip']),
                     float(out['em0_dp']), prior_logits=out['em0_prior_logits'])
    for it, rec in enumerate(hist):
        assert_bitwise(rec['logits'], out[f'em0_it{it}_logits'], f'it {it} logits')
        assert_bitwise(rec['probs'], out[f'em0_it{it}_probs'], f'it {it} probs')
        assert_bitwise(rec['addition'], out[f'em0_it{it}_addition'], f'it {it} addition')


@pytest.mark.parametrize('name', ['f7_aggregate_small_2.npz', 'f7_aggregate_small_4.npz', 'f7_aggregate_synthetic_g4.npz'])
def test_oracle_aggregate_on_snps_matches_reference(oracle, name):
    """Demultiplexer.aggregate_on_snps = True (demux.py:204-244): the restatement against the reference's captured
    float64 logits / posteriors and float32 additions, bit for bit (same numpy, same machine class)."""
    out = fio.load(name)
    fx = fio.load(str(out['inputs_of']))
    n_barcodes = len(fx['barcodes'])
    for i in range(int(out['n_predict'])):
        packed = oracle.pack(fio.oracle_calls(fx), fio.oracle_geno(fx), add_data_prior=False)
        prob = oracle.probs_from_betas(packed['v2snp'], packed['betas'], 0.01)
        logits = oracle.barcode_logits_aggregated(packed['mol_variant'], packed['mol_cb'], packed['mol_p'], packed['v2snp'],
                                                  prob, n_barcodes, float(out[f'predict{i}_dp']))
        fio.assert_bitwise(logits, out[f'predict{i}_logits'], f'{name} predict {i} logits')
    packed = oracle.pack(fio.oracle_calls(fx), fio.oracle_geno(fx), add_data_prior=True)
    for i in range(int(out['n_em'])):
        n_it = int(out[f'em{i}_n_iterations'])
        prior = out.get(f'em{i}_prior_logits')
        hist = oracle.em_aggregated(packed, n_barcodes, n_it, 0.01, float(out[f'em{i}_dp']), prior_logits=prior)
        for it in range(n_it):
            fio.assert_bitwise(hist[it]['logits'], out[f'em{i}_it{it}_logits'], f'{name} run {i} it {it} logits')
            fio.assert_bitwise(hist[it]['probs'], out[f'em{i}_it{it}_probs'], f'{name} run {i} it {it} probs')
            fio.assert_bitwise(hist[it]['addition'], out[f'em{i}_it{it}_addition'], f'{name} run {i} it {it} addition')
