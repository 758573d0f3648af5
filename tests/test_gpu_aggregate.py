"""Demultiplexer.aggregate_on_snps = True (demux.py:204-244) on the GPU against the reference's captured outputs
(F7 fixtures: tests/golden/make_fixtures.py: aggregate_on_snps_cases).

Everything up to the first log_softmax is float32 arithmetic reproduced bit for bit (numpy's float32 log / exp /
pairwise sums, float64 accumulation in molecule order); np.logaddexp and the second log_softmax are float64, where
numpy's exp / log / log1p are its own SIMD kernels or libm depending on the host CPU and the device has the ROCm
device library's.  The gate is therefore: assignments identical, posteriors within 1e-12 (the contract says 1e-5),
logits within 1e-11 relative; the achieved agreement in float64 ulps is printed."""
import numpy as np
import pytest

from tests import fixture_io as fio

pytestmark = pytest.mark.gpu

NAMES = ['f7_aggregate_small_2.npz', 'f7_aggregate_small_4.npz', 'f7_aggregate_synthetic_g4.npz',
         'f7_aggregate_synthetic_default.npz']


@pytest.fixture()
def aggregating():
    from demuxalot_amd import Demultiplexer
    Demultiplexer.aggregate_on_snps = True
    yield Demultiplexer
    Demultiplexer.aggregate_on_snps = False


def check64(got_logits, got_probs, ref_logits, ref_probs, what):
    assert got_logits.dtype == np.float64 and got_probs.dtype == np.float64, what
    assert np.array_equal(got_probs.argmax(axis=1), ref_probs.argmax(axis=1)), f'{what}: assignments differ'
    assert np.abs(got_probs - ref_probs).max() <= 1e-12, f'{what}: posteriors {np.abs(got_probs - ref_probs).max():.3g}'
    assert np.allclose(got_logits, ref_logits, rtol=1e-11, atol=1e-11), f'{what}: logits {np.abs(got_logits - ref_logits).max():.3g}'
    return float((np.abs(got_logits - ref_logits) / np.spacing(np.abs(ref_logits))).max())


@pytest.mark.parametrize('name', NAMES)
def test_aggregate_on_snps_matches_reference(aggregating, name):
    D = aggregating
    out = fio.load(name)
    fx = fio.load(str(out['inputs_of']))
    calls, genotypes, handler = fio.product_inputs(fx)
    worst = 0.0
    for i in range(int(out['n_predict'])):
        if out[f'predict{i}_logits'].shape[1] > 1024:  # stated limit of this mode (include/demux_hip.h)
            from demuxalot_amd._lib import DemuxHipError
            with pytest.raises(DemuxHipError, match='up to 1024 options'):
                D.predict_posteriors(calls, genotypes, handler, doublet_prior=float(out[f'predict{i}_dp']))
            continue
        logits_df, probs_df = D.predict_posteriors(calls, genotypes, handler, doublet_prior=float(out[f'predict{i}_dp']))
        assert logits_df.index.name == 'BARCODE' and list(logits_df.columns) == [str(c) for c in out[f'predict{i}_columns']]
        worst = max(worst, check64(logits_df.values, probs_df.values, out[f'predict{i}_logits'], out[f'predict{i}_probs'],
                                   f'{name} predict {i}'))
    for i in range(int(out['n_em'])):
        kwargs = dict(n_iterations=int(out[f'em{i}_n_iterations']), doublet_prior=float(out[f'em{i}_dp']))
        prior = out.get(f'em{i}_prior_logits')
        stages = list(D.staged_genotype_learning(calls, genotypes, handler,
                                                 barcode_prior_logits=None if prior is None else prior.copy(), **kwargs))
        for it, (probs_df, dbg) in enumerate(stages):
            worst = max(worst, check64(dbg['barcode_logits'], probs_df.values, out[f'em{i}_it{it}_logits'],
                                       out[f'em{i}_it{it}_probs'], f'{name} run {i} it {it}'))
            want = out[f'em{i}_it{it}_addition']
            assert dbg['genotype_addition'].dtype == np.float32
            assert np.allclose(dbg['genotype_addition'], want, rtol=3e-7, atol=0), f'{name} run {i} it {it} addition'
            assert (dbg['genotype_addition'] != want).mean() <= 1e-3  # a float64 ulp upstream rarely moves a float32 rounding
        learnt, last = D.learn_genotypes(calls, genotypes, handler,
                                         barcode_prior_logits=None if prior is None else prior.copy(), **kwargs)
        assert np.allclose(learnt.variant_betas, out[f'em{i}_learnt_betas'], rtol=3e-7, atol=0)
        assert np.array_equal(last.values, stages[-1][0].values)
    print(f'{name}: float64 logits within {worst:.1f} ulp of the reference')


def test_aggregate_dispatcher_on_caller_supplied_tables(aggregating, oracle):
    """compute_barcode_logits with molecule_calls (the dispatcher of demux.py:193-202) against the oracle."""
    D = aggregating
    fx = fio.load('f3_small_3.npz')
    calls, genotypes, handler = fio.product_inputs(fx)
    v2snp, betas, molecule_calls, barcode_calls = D.pack_calls(calls, genotypes, add_data_prior=False)
    prob = oracle.probs_from_betas(v2snp, betas, 0.01)
    for dp in (0., 0.3):
        want = oracle.barcode_logits_aggregated(molecule_calls['variant_id'], molecule_calls['compressed_cb'],
                                                molecule_calls['p_base_wrong'], v2snp, prob, handler.n_barcodes, dp)
        got, names = D.compute_barcode_logits(genotypes.genotype_names, barcode_calls, molecule_calls, dp, prob,
                                              handler.n_barcodes, genotypes.n_genotypes)
        assert got.dtype == np.float64 and len(names) == want.shape[1]
        assert np.allclose(got, want, rtol=1e-11, atol=1e-11)


def test_aggregate_mode_refuses_what_it_cannot_do(aggregating):
    fx = fio.load('f3_small_0.npz')
    calls, genotypes, handler = fio.product_inputs(fx)
    with pytest.raises(AssertionError, match='float64'):
        aggregating.predict_posteriors(calls, genotypes, handler, on_device=True)
