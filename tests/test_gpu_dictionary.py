"""Dictionary form of the exact E-step (csrc/estep_dict.hip) against the direct form and the oracle (-m gpu).

The form evaluates numpy's float32 log once per (call, distinct value of the call's genotype_prob row) instead of once
per (call, option) (demuxalot/demux.py:246-265; why rows have few distinct values: genotypes.py:147-164).  Every option
still receives the same addends in the same order, so everything here is BITWISE equality."""
import numpy as np
import pytest

from tests import fixture_io as fio

pytestmark = pytest.mark.gpu


def low_cardinality_table(rng, n_rows, n_genotypes, n_values, clip=0.01):
    """[V, G] float32 probabilities with at most n_values distinct values per row (a fifth of the rows: one value)."""
    values = rng.uniform(clip, 1 - clip, size=(n_rows, n_values)).astype(np.float32)
    pick = rng.integers(0, n_values, size=(n_rows, n_genotypes))
    pick[rng.random(n_rows) < 0.2] = 0
    return np.take_along_axis(values, pick, axis=1)


def random_calls(rng, n_barcodes, n_rows, mean_calls):
    """Packed calls (variant-major like the reference's barcode_calls); some barcodes have no call at all."""
    n_b = rng.poisson(mean_calls, size=n_barcodes)
    n_b[rng.random(n_barcodes) < 0.05] = 0
    n_b = np.minimum(n_b, n_rows)
    cb = np.repeat(np.arange(n_barcodes, dtype=np.int32), n_b)
    variant = np.concatenate([rng.choice(n_rows, size=k, replace=False) for k in n_b]).astype(np.int32) if len(cb) else np.zeros(0, np.int32)
    e = (10.0 ** (-rng.integers(10, 41, size=len(cb)) / 10.0)).astype(np.float32)
    e[rng.random(len(cb)) < 0.02] = 0.0
    order = np.lexsort((cb, variant))
    return variant[order], cb[order], e[order]


def run_estep(ctx, table, pen, doublets, mode):
    # 'auto' would pass these small problems to the direct form (the dictionary form pays from a few rounds of
    # wavefronts on): 'always' tries it for every table
    ctx.set_estep_dictionary('always' if mode == 'auto' else mode)
    ctx.set_probs(table)
    logits, probs = ctx.estep(pen, with_doublets=doublets)
    return logits, probs, ctx.estep_form()


@pytest.mark.parametrize('n_genotypes,doublet_prior,n_values', [
    (2, 0., 2), (5, 0., 3), (8, 0.35, 3), (8, 0.35, 4), (20, 0., 4), (20, 0.25, 4), (22, 0.25, 2), (33, 0., 8),
    (64, 0., 4), (64, 0., 8), (64, 0., 1), (70, 0., 5), (130, 0., 4), (130, 0., 7), (256, 0., 3), (3, 0., 3), (4, 0.3, 2),
    (16, 0., 8), (17, 0., 2), (32, 0., 4), (129, 0., 8),
    (23, 0.2, 4), (30, 0.2, 3), (64, 0.25, 4), (100, 0.1, 1), (130, 0.25, 2)])  # the last five: doublet tables of more than 256 options
def test_dictionary_form_equals_direct_form(oracle, n_genotypes, doublet_prior, n_values):
    from demuxalot_amd import Demultiplexer
    from demuxalot_amd.device import DeviceContext
    rng = np.random.default_rng(1000 * n_genotypes + n_values)
    n_barcodes, n_rows = 700, 900
    variant, cb, e = random_calls(rng, n_barcodes, n_rows, 45)
    table = low_cardinality_table(rng, n_rows, n_genotypes, n_values)
    pen = Demultiplexer._doublet_penalties(n_genotypes, doublet_prior)
    ctx = DeviceContext(0)
    try:
        ctx.set_problem(n_barcodes, n_rows, n_genotypes, variant, cb, e, np.arange(n_rows, dtype=np.int32))
        l_dir, p_dir, form_dir = run_estep(ctx, table, pen, doublet_prior != 0, 'never')
        l_dic, p_dic, form_dic = run_estep(ctx, table, pen, doublet_prior != 0, 'auto')
        assert form_dir in (('direct', 0), ('packed', 0))  # 8 genotypes with doublets: several option slots per lane
        n_options = len(pen)
        assert form_dic[0] == ('dict_block' if doublet_prior and n_options > 256 else 'dict') and 1 <= form_dic[1] <= n_values, form_dic
        fio.assert_bitwise(l_dic, l_dir, 'logits: dictionary vs direct form')
        fio.assert_bitwise(p_dic, p_dir, 'posteriors: dictionary vs direct form')
        want = oracle.barcode_logits(variant, cb, e, table, n_barcodes, doublet_prior, log_impl='npsimd')
        fio.assert_bitwise(l_dic, want, 'logits: dictionary form vs oracle')
        # the M-step reads what the E-step epilogue left (bitmaps, barcode codes): same additions either way
        add_dic = ctx.mstep(2.)
        run_estep(ctx, table, pen, doublet_prior != 0, 'never')
        fio.assert_bitwise(add_dic, ctx.mstep(2.), 'M-step after either form')
    finally:
        ctx.close()


@pytest.mark.parametrize('n_genotypes,doublet_prior,n_values,why', [
    (64, 0., 9, 'nine values in a row'), (12, 0.3, 5, 'five values with doublets'), (64, 0., 64, 'all distinct'),
    (300, 0., 3, 'more than 256 options'), (30, 0.2, 5, 'five values with doublets, more than 256 options')])
def test_rows_with_many_values_take_the_direct_form(n_genotypes, doublet_prior, n_values, why):
    from demuxalot_amd import Demultiplexer
    from demuxalot_amd.device import DeviceContext
    rng = np.random.default_rng(5)
    variant, cb, e = random_calls(rng, 300, 400, 30)
    table = low_cardinality_table(rng, 400, n_genotypes, min(n_values, 8))
    if 'options' not in why:
        table[7, :n_values] = np.linspace(0.1, 0.9, n_values, dtype=np.float32)  # ONE row beyond the capacity
    pen = Demultiplexer._doublet_penalties(n_genotypes, doublet_prior)
    ctx = DeviceContext(0)
    try:
        ctx.set_problem(300, 400, n_genotypes, variant, cb, e, np.arange(400, dtype=np.int32))
        l_dir, p_dir, _ = run_estep(ctx, table, pen, doublet_prior != 0, 'never')
        l_try, p_try, form = run_estep(ctx, table, pen, doublet_prior != 0, 'always')
        assert form[0] in ('direct', 'packed') and ('options' in why or form[1] >= min(n_values, 9)), (why, form)
        fio.assert_bitwise(l_try, l_dir, why)
        fio.assert_bitwise(p_try, p_dir, why)
    finally:
        ctx.close()


@pytest.mark.parametrize('name,expect', [('f1_synthetic_default.npz', 'dict'), ('f2_synthetic_g4.npz', 'dict'),
                                         ('f6_shipped_example.npz', 'dict'), ('f3_small_4.npz', 'direct')])
def test_reference_fixtures_through_the_dictionary_form(name, expect, monkeypatch):
    """predict_posteriors and EM on the reference's own test inputs with the dictionary form tried for every E-step
    (DEMUXALOT_AMD_ESTEP_DICT=always; by default problems this small take the direct form): their rows hold 2-3 distinct
    values, so predict and EM iteration 0 run the form; the outputs are the reference's, bit for bit."""
    from demuxalot_amd import Demultiplexer
    from demuxalot_amd.device import get_context
    monkeypatch.setenv('DEMUXALOT_AMD_ESTEP_DICT', 'always')  # read by every new context (the generator's private one)
    ctx = get_context()
    ctx.set_estep_dictionary('always')
    try:
        fx = fio.load(name)
        calls, genotypes, handler = fio.product_inputs(fx)
        for i in range(int(fx['n_predict'])):
            dp, clip = float(fx[f'predict{i}_dp']), float(fx[f'predict{i}_clip'])
            logits_df, probs_df = Demultiplexer.predict_posteriors(calls, genotypes, handler, p_genotype_clip=clip, doublet_prior=dp)
            form, distinct = ctx.estep_form()
            wide = logits_df.shape[1] > 256
            assert form == ('direct' if wide else expect), (name, dp, form, distinct)
            fio.assert_bitwise(logits_df.values, fx[f'predict{i}_logits'], f'{name} predict {i} logits')
            fio.assert_bitwise(probs_df.values, fx[f'predict{i}_probs'], f'{name} predict {i} probs')
        for i in range(int(fx['n_em'])):
            kwargs = dict(n_iterations=int(fx[f'em{i}_n_iterations']), p_genotype_clip=float(fx[f'em{i}_clip']),
                          doublet_prior=float(fx[f'em{i}_dp']))
            prior = fx.get(f'em{i}_prior_logits')
            stages = Demultiplexer.staged_genotype_learning(
                calls, genotypes, handler, barcode_prior_logits=None if prior is None else prior.copy(), **kwargs)
            for it, (probs_df, dbg) in enumerate(stages):
                fio.assert_bitwise(dbg['barcode_logits'], fx[f'em{i}_it{it}_logits'], f'{name} EM {i} it {it} logits')
                fio.assert_bitwise(probs_df.values, fx[f'em{i}_it{it}_probs'], f'{name} EM {i} it {it} probs')
                fio.assert_bitwise(dbg['genotype_addition'], fx[f'em{i}_it{it}_addition'], f'{name} EM {i} it {it} addition')
    finally:
        ctx.set_estep_dictionary('auto')


def test_em_driver_switches_forms_between_iterations(oracle):
    """dmx_em: iteration 0 (no addition yet) runs the dictionary form, iteration 1 (all-distinct rows) the direct one,
    and the result equals a run with the dictionary form switched off."""
    from demuxalot_amd import synth
    from demuxalot_amd.device import DeviceContext
    p = synth.generate(3000, 2000, 16, seed=5)
    pen = np.zeros(16, dtype=np.float32)
    out = {}
    for mode in ('never', 'auto'):
        ctx = DeviceContext(0)
        try:
            ctx.set_estep_dictionary('always' if mode == 'auto' else mode)
            ctx.set_problem(p.n_barcodes, p.n_variants, 16, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
            ctx.set_betas(p.prior_betas())
            _, probs1, add1 = ctx.em(1, 0.01, pen, with_doublets=False, fetch_logits=False)
            form1 = ctx.estep_form()
            _, probs3, add3 = ctx.em(3, 0.01, pen, with_doublets=False, fetch_logits=False)
            out[mode] = (probs1, probs3, add3, form1, ctx.estep_form())
        finally:
            ctx.close()
    assert out['never'][3][0] == 'direct' and out['never'][4][0] == 'direct'
    assert out['auto'][3][0] == 'dict' and out['auto'][3][1] <= 4, out['auto'][3]
    assert out['auto'][4][0] == 'direct'
    for k, what in enumerate(('posteriors after 1 iteration', 'posteriors after 3 iterations', 'addition')):
        fio.assert_bitwise(out['auto'][k], out['never'][k], what)


def test_headline_shape_predict_dictionary_vs_direct():
    """200k x 100k x 64 (N ~ 78.7 M): the whole [B, K] logits / posteriors of the predict pass, dictionary form against
    direct form, bitwise; the rows' cardinality is what the generator (add_vcf-shaped betas) gives: <= 4."""
    from demuxalot_amd import synth
    from demuxalot_amd.device import DeviceContext
    p = synth.generate(200_000, 100_000, 64, seed=1237)
    pen = np.zeros(64, dtype=np.float32)
    ctx = DeviceContext(0)
    try:
        ctx.set_problem(p.n_barcodes, p.n_variants, 64, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
        ctx.set_betas(p.prior_betas(add_data_prior=False))
        got = {}
        for mode in ('never', 'auto'):  # 200k barcodes: the default mode takes the form by itself
            ctx.set_estep_dictionary(mode)
            ctx.set_addition(None)
            ctx.probs_from_betas(0.01, fetch=False)
            got[mode] = ctx.estep(pen, with_doublets=False) + (ctx.estep_form(),)
        assert got['never'][2][0] == 'direct' and got['auto'][2] == ('dict', 4), (got['never'][2], got['auto'][2])
        fio.assert_bitwise(got['auto'][0], got['never'][0], 'logits')
        fio.assert_bitwise(got['auto'][1], got['never'][1], 'posteriors')
    finally:
        ctx.close()


@pytest.mark.parametrize('doublets', [False, True])
def test_more_than_1024_genotypes_is_refused_with_a_message(doublets):
    """1025 genotypes (singlets, and doublets whose dictionary gate once let them through to a kernel that does not exist):
    the E-step refuses loudly instead of failing inside a launch; 1024 genotypes run."""
    from demuxalot_amd import Demultiplexer, synth
    from demuxalot_amd._lib import DemuxHipError
    from demuxalot_amd.device import DeviceContext
    for G, fails in ((1024, False), (1025, True)):
        if doublets and not fails:
            continue  # K = 524 800 options: covered at smaller G by the wide-table tests
        p = synth.generate(40, 30, G, calls_per_barcode=10, doublets=doublets, seed=5 + G)
        pen = Demultiplexer._doublet_penalties(G, 0.3 if doublets else 0.0)
        ctx = DeviceContext(0)
        try:
            ctx.set_problem(p.n_barcodes, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
            ctx.set_betas(p.prior_betas(add_data_prior=False))
            ctx.set_addition(None)
            ctx.probs_from_betas(0.01, fetch=False)
            if fails:
                with pytest.raises(DemuxHipError, match='more than 1024 genotypes'):
                    ctx.estep(pen, with_doublets=doublets, fetch_logits=False, fetch_probs=True)
            else:
                _, probs = ctx.estep(pen, with_doublets=doublets, fetch_logits=False, fetch_probs=True)
                assert np.isfinite(probs).all() and np.allclose(probs.sum(1), 1.0, atol=1e-4)
        finally:
            ctx.close()
