"""The tolerance-mode E-step (dmx_set_estep_mode(ctx, DMX_ESTEP_FAST) / DEMUXALOT_AMD_ESTEP=fast) on the GPU.

Contract (BASELINE.json north_star): barcode -> donor assignments identical to the reference, posteriors within
1e-5 on the reference's test inputs.  The default mode is bit-exact; the fast mode multiplies the terms of 8 calls
and takes one hardware log2 per product, so its logits differ from the reference's by about one float32 rounding
of the logit -- the size of the rounding noise the reference's own float32 terms carry.  A posterior moves by at
most p (1 - p) x (logit deviation): for |logit| < 512 (every barcode of the reference's tests) a one-ulp flip
(3.05e-5) is at most 7.6e-6.  On longer rows (|logit| up to 2^k) the bound scales with the logit's ulp, which is
what `posterior_bound` states."""
import numpy as np
import pytest

from tests import fixture_io as fio

pytestmark = pytest.mark.gpu

TOL_POSTERIOR = 1e-5  # BASELINE.json north_star, on the reference's test inputs


@pytest.fixture()
def fast(monkeypatch):
    """Fast mode for the shared context and (through the environment) for the private contexts the front-end
    creates; restored afterwards."""
    from demuxalot_amd.device import get_context
    monkeypatch.setenv('DEMUXALOT_AMD_ESTEP', 'fast')
    ctx = get_context()
    ctx.apply_environment()
    yield ctx
    monkeypatch.setenv('DEMUXALOT_AMD_ESTEP', 'exact')  # the suite's pin (tests/conftest.py)
    ctx.apply_environment()


def posterior_bound(ref_logits):
    """max(1e-5, 0.3 x ulp of the largest |logit| of the row): one rounding flip of the best two logits."""
    top = np.abs(ref_logits).max(axis=1, keepdims=True)
    ulp = np.spacing(top.astype(np.float32))
    return np.maximum(TOL_POSTERIOR, 0.3 * ulp)


def check_contract(got_logits, got_probs, ref_logits, ref_probs, what, strict):
    """strict: the contract's 1e-5 on every posterior (reference test inputs); otherwise the ulp-scaled bound."""
    dev = np.abs(got_probs.astype(np.float64) - ref_probs)
    bound = TOL_POSTERIOR if strict else posterior_bound(ref_logits)
    assert (dev <= bound).all(), f'{what}: posterior deviation {dev.max():.3g}'
    # assignments: identical wherever the reference's best two posteriors are not a rounding apart
    best, ref_best = got_probs.argmax(axis=1), ref_probs.argmax(axis=1)
    if not np.array_equal(best, ref_best):
        rows = np.flatnonzero(best != ref_best)
        gap = ref_probs[rows, ref_best[rows]] - ref_probs[rows, best[rows]]
        assert strict is False and (gap <= 2 * posterior_bound(ref_logits)[rows, 0]).all(), f'{what}: assignments differ on rows {rows[:5]}'
    with np.errstate(invalid='ignore'):
        rel = np.abs(got_logits.astype(np.float64) - ref_logits) / np.maximum(np.spacing(np.abs(ref_logits)), 1e-30)
    return float(np.nanmax(rel)), float(dev.max())


@pytest.mark.parametrize('name', fio.SMALL + fio.SYNTH)
def test_fast_mode_meets_the_contract_on_reference_outputs(fast, name):
    """predict_posteriors and every EM iteration of the golden fixtures (the reference's own outputs)."""
    from demuxalot_amd import Demultiplexer
    fx = fio.load(name)
    calls, genotypes, handler = fio.product_inputs(fx)
    worst = (0.0, 0.0)
    for i in range(int(fx['n_predict'])):
        dp, clip = float(fx[f'predict{i}_dp']), float(fx[f'predict{i}_clip'])
        logits_df, probs_df = Demultiplexer.predict_posteriors(calls, genotypes, handler, p_genotype_clip=clip, doublet_prior=dp)
        worst = max(worst, check_contract(logits_df.values, probs_df.values, fx[f'predict{i}_logits'], fx[f'predict{i}_probs'],
                                          f'{name} predict {i}', strict=True))
    for i in range(int(fx['n_em'])):
        kwargs = dict(n_iterations=int(fx[f'em{i}_n_iterations']), p_genotype_clip=float(fx[f'em{i}_clip']),
                      doublet_prior=float(fx[f'em{i}_dp']))
        prior = fx.get(f'em{i}_prior_logits')
        stages = list(Demultiplexer.staged_genotype_learning(
            calls, genotypes, handler, barcode_prior_logits=None if prior is None else prior.copy(), **kwargs))
        for it, (probs_df, dbg) in enumerate(stages):
            worst = max(worst, check_contract(dbg['barcode_logits'], probs_df.values, fx[f'em{i}_it{it}_logits'],
                                              fx[f'em{i}_it{it}_probs'], f'{name} run {i} it {it}', strict=True))
            # the genotype additions follow the posteriors: close, not identical
            assert np.allclose(dbg['genotype_addition'], fx[f'em{i}_it{it}_addition'], rtol=1e-3, atol=1e-4)
        learnt, last = Demultiplexer.learn_genotypes(
            calls, genotypes, handler, barcode_prior_logits=None if prior is None else prior.copy(), **kwargs)
        assert np.allclose(learnt.variant_betas, fx[f'em{i}_learnt_betas'], rtol=1e-3, atol=1e-4)
    print(f'{name}: worst logit deviation {worst[0]:.2f} ulp, worst posterior deviation {worst[1]:.3g}')


@pytest.mark.parametrize('G,dp', [(2, 0.), (3, 0.3), (8, 0.35), (16, 0.), (20, 0.25), (32, 0.), (33, 0.), (64, 0.), (64, 0.1),
                                  (100, 0.), (128, 0.), (200, 0.), (24, 0.3), (32, 0.25), (45, 0.1), (130, 0.05), (300, 0.), (600, 0.)])
def test_fast_mode_every_kernel_shape_against_oracle(fast, oracle, G, dp):
    """All lane-group widths, slot counts and the workgroup-per-barcode form: posteriors within the contract, logits
    within a few float32 ulps of the oracle's.  (The deviation is mostly the REFERENCE's: numpy's float32 log is off
    by up to 3.8 ulp, identically for identical arguments, and an option that mismatches at 50 calls adds the same
    log(p_clip (1 - e) + e) term -- and its error -- 50 times.)"""
    from demuxalot_amd import Demultiplexer, synth
    p = synth.generate(n_barcodes=150, n_snps=300, n_genotypes=G, calls_per_barcode=60, doublets=dp > 0, seed=G)
    prob = oracle.probs_from_betas(p.v2snp, p.prior_betas(), 0.01)
    bc = dict(variant_id=p.variant_id, compressed_cb=p.compressed_cb, p_base_wrong=p.p_base_wrong)
    want = oracle.barcode_logits(p.variant_id, p.compressed_cb, p.p_base_wrong, prob, 150, dp, log_impl='npsimd')
    names = [f'g{i:03d}' for i in range(G)]
    got, _ = Demultiplexer.compute_barcode_logits_using_barcode_calls(
        names, bc, doublet_prior=dp, genotype_prob=prob, n_barcodes=150, n_genotypes=G)
    probs = fast.get_probs()
    ulps, dev = check_contract(got, probs, want, oracle.softmax_rows(want, impl='npsimd'), f'G={G} dp={dp}', strict=True)
    assert ulps <= 16, ulps  # float32 ulps of the logit


def test_fast_mode_midsize_em_against_oracle(fast, oracle):
    """20k x 10k x 64, three EM iterations (the deviations of one iteration feed the next through the M-step)."""
    from demuxalot_amd import synth
    p = synth.generate(20000, 10000, 64, calls_per_barcode=200, seed=77)
    betas = p.prior_betas()
    fast.set_problem(p.n_barcodes, p.n_variants, 64, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
    fast.set_betas(betas)
    logits, probs, addition = fast.em(3, 0.01, np.zeros(64, dtype=np.float32), with_doublets=False)
    packed = dict(variant_id=p.variant_id, compressed_cb=p.compressed_cb, p_base_wrong=p.p_base_wrong, betas=betas, v2snp=p.v2snp)
    hist = oracle.em(packed, p.n_barcodes, 3, 0.01, 0., impl='npsimd')
    ulps, dev = check_contract(logits, probs, hist[-1]['logits'], hist[-1]['probs'], 'midsize EM', strict=False)
    assert np.allclose(addition, hist[-1]['addition'], rtol=1e-3, atol=1e-4)
    print(f'midsize EM, fast mode: logits within {ulps:.2f} ulp, posteriors within {dev:.3g}')


def test_fast_mode_nan_and_degenerate_calls(fast):
    """p_base_wrong of exactly 0 and 1, a 1e-38 product and an empty barcode behave as in the exact mode."""
    from demuxalot_amd.device import DeviceContext
    variant = np.array([0, 1, 2, 0, 1], dtype=np.int32)
    cb = np.array([0, 0, 0, 1, 1], dtype=np.int32)
    e = np.array([0.0, 1.0, 1e-38, 0.5, 0.25], dtype=np.float32)
    table = np.array([[.9, .1], [.2, .8], [.5, .5]], dtype=np.float32)
    out = {}
    for mode in ('exact', 'fast'):
        ctx = DeviceContext(0)
        try:
            ctx.set_estep_mode(mode)
            ctx.set_problem(3, 3, 2, variant, cb, e, np.zeros(3, dtype=np.int32))
            ctx.set_probs(table)
            out[mode] = ctx.estep(np.zeros(2, dtype=np.float32), with_doublets=False)
        finally:
            ctx.close()
    assert np.allclose(out['fast'][0], out['exact'][0], rtol=3e-7, atol=1e-7)
    assert np.allclose(out['fast'][1], out['exact'][1], rtol=0, atol=1e-6)
    assert np.array_equal(out['fast'][0][2], [0, 0]) and np.array_equal(out['fast'][1][2], [.5, .5])  # the empty barcode
