"""Device-side result reductions and the context ownership of the front-end, on the GPU.

The reductions are what users of the reference apply to the posterior DataFrame
(examples/2-with-detection-of-new-SNPs.ipynb cells 14 / 19, demuxalot/snp_detection.py:166); the checker is
pandas / numpy applied to the reference's own captured posteriors (golden fixtures)."""
import numpy as np
import pandas as pd
import pytest

from tests import fixture_io as fio

pytestmark = pytest.mark.gpu


def reference_probs_df(fx, i):
    return pd.DataFrame(fx[f'predict{i}_probs'], index=[str(b) for b in fx['barcodes']],
                        columns=[str(c) for c in fx[f'predict{i}_columns']])


@pytest.mark.parametrize('name', ['f1_synthetic_default.npz', 'f6_shipped_example.npz', 'f3_small_2.npz'])
def test_device_reductions_match_pandas_on_reference_posteriors(name):
    from demuxalot_amd import Demultiplexer
    fx = fio.load(name)
    calls, genotypes, handler = fio.product_inputs(fx)
    for i in range(int(fx['n_predict'])):
        dp, clip = float(fx[f'predict{i}_dp']), float(fx[f'predict{i}_clip'])
        want = reference_probs_df(fx, i)
        with Demultiplexer.predict_posteriors(calls, genotypes, handler, p_genotype_clip=clip, doublet_prior=dp,
                                              on_device=True) as dev:
            assert dev.shape == want.shape and dev.columns == list(want.columns)
            for thr in (0.9, 0.8, 0.5, 0.0, 0.999999):
                got = dev.assignments(thr)
                ref = want[want.max(axis=1).gt(thr)].idxmax(axis=1)
                assert list(got.index) == list(ref.index) and list(got.values) == list(ref.values), (name, i, thr)
                assert got.index.name == 'BARCODE'
            best = dev.best()
            assert list(best['option']) == list(want.idxmax(axis=1))
            fio.assert_bitwise(best['probability'].values, want.values.max(axis=1), 'row maxima')
            k = min(3, want.shape[1])
            top = dev.top_options(k)
            order = np.argsort(-want.values, axis=1, kind='stable')[:, :k]  # ties: lower column first
            for j in range(k):
                assert list(top[f'option_{j + 1}']) == [want.columns[c] for c in order[:, j]], (name, i, j)
                fio.assert_bitwise(top[f'probability_{j + 1}'].values,
                                   np.take_along_axis(want.values, order[:, j:j + 1], axis=1)[:, 0], 'top probabilities')
            sums = dev.option_sums()
            assert list(sums.index) == list(want.columns)
            assert np.allclose(sums.values, want.values.astype(np.float64).sum(axis=0), rtol=1e-12, atol=0)
            assert np.allclose(sums.values, want.sum().values, rtol=2e-5)  # pandas adds in float32
            logits_df, probs_df = dev.to_dataframes()
            fio.assert_bitwise(probs_df.values, fx[f'predict{i}_probs'], 'to_dataframes probs')
            fio.assert_bitwise(logits_df.values, fx[f'predict{i}_logits'], 'to_dataframes logits')
            assert probs_df.index.name == 'BARCODE' and list(probs_df.index) == list(want.index)
            lo, hi = len(want) // 3, len(want) // 3 + 5
            fio.assert_bitwise(dev.rows(lo, hi).values, fx[f'predict{i}_probs'][lo:hi], 'rows')


def test_top_options_with_ties_short_rows_and_threshold_edges():
    """Hand-made posteriors pushed through the raw entry points: ties go to the lower column, rows shorter than
    k are padded with -1 / NaN, `gt` is strict."""
    from demuxalot_amd.device import DeviceContext
    ctx = DeviceContext(0)
    try:
        # 3 barcodes x 2 genotypes: equal evidence everywhere -> posteriors exactly (.5, .5)
        ctx.set_problem(3, 2, 2, np.array([0, 1, 0]), np.array([0, 1, 2]), np.full(3, .1, dtype='f4'), np.zeros(2, dtype='i4'))
        ctx.set_probs(np.full((2, 2), 0.5, dtype=np.float32))
        _, probs = ctx.estep(np.zeros(2, dtype=np.float32), with_doublets=False)
        assert np.array_equal(probs, np.full((3, 2), 0.5, dtype=np.float32))
        options, p = ctx.get_top_options(4)
        assert options.tolist() == [[0, 1, -1, -1]] * 3
        assert np.array_equal(p[:, :2], probs) and np.isnan(p[:, 2:]).all()
        best, prob, n = ctx.get_assignments_above(0.5)  # .5 is not > .5
        assert best.tolist() == [-1, -1, -1] and n == 0 and np.array_equal(prob, np.full(3, .5, dtype='f4'))
        best, prob, n = ctx.get_assignments_above(np.nextafter(np.float32(0.5), np.float32(0)))
        assert best.tolist() == [0, 0, 0] and n == 3
        assert np.array_equal(ctx.get_option_sums(), [1.5, 1.5])
    finally:
        ctx.close()


def test_learn_genotypes_on_device_matches_reference():
    from demuxalot_amd import Demultiplexer
    fx = fio.load('f1_synthetic_default.npz')
    calls, genotypes, handler = fio.product_inputs(fx)
    kwargs = dict(n_iterations=int(fx['em0_n_iterations']), p_genotype_clip=float(fx['em0_clip']),
                  doublet_prior=float(fx['em0_dp']))
    learnt, dev = Demultiplexer.learn_genotypes(calls, genotypes, handler, on_device=True, **kwargs)
    try:
        fio.assert_bitwise(learnt.variant_betas, fx['em0_learnt_betas'], 'learnt betas')
        want = fx[f'em0_it{kwargs["n_iterations"] - 1}_probs']
        _logits_df, probs_df = dev.to_dataframes()
        fio.assert_bitwise(probs_df.values, want, 'posteriors left on the device')
        assert probs_df.index.name is None
        ref = pd.DataFrame(want, index=probs_df.index, columns=probs_df.columns)
        got = dev.assignments(0.9)
        ref_assign = ref[ref.max(axis=1).gt(0.9)].idxmax(axis=1)
        assert list(got.index) == list(ref_assign.index) and list(got.values) == list(ref_assign.values)
    finally:
        dev.close()


def test_generator_survives_other_calls_between_iterations():
    """The reference's generator is pure; here its EM state lives on the GPU, in a context of its own: running
    other Demultiplexer entry points (which install other problems on the shared context) between two
    iterations must not disturb it."""
    from demuxalot_amd import Demultiplexer
    fx = fio.load('f2_synthetic_g4.npz')
    other = fio.load('f3_small_3.npz')
    calls, genotypes, handler = fio.product_inputs(fx)
    o_calls, o_genotypes, o_handler = fio.product_inputs(other)
    kwargs = dict(n_iterations=int(fx['em0_n_iterations']), p_genotype_clip=float(fx['em0_clip']),
                  doublet_prior=float(fx['em0_dp']))
    prior = fx.get('em0_prior_logits')
    gen = Demultiplexer.staged_genotype_learning(calls, genotypes, handler,
                                                 barcode_prior_logits=None if prior is None else prior.copy(), **kwargs)
    for it, (probs_df, dbg) in enumerate(gen):
        fio.assert_bitwise(probs_df.values, fx[f'em0_it{it}_probs'], f'it {it} probs')
        fio.assert_bitwise(dbg['genotype_addition'], fx[f'em0_it{it}_addition'], f'it {it} addition')
        # same B/V/G problem AND a different one on the shared context, then a P-step helper
        Demultiplexer.predict_posteriors(calls, genotypes, handler, doublet_prior=0.35)
        logits_df, _ = Demultiplexer.predict_posteriors(o_calls, o_genotypes, o_handler,
                                                        doublet_prior=float(other['predict0_dp']),
                                                        p_genotype_clip=float(other['predict0_clip']))
        fio.assert_bitwise(logits_df.values, other['predict0_logits'], 'interleaved predict')
        Demultiplexer.learn_genotypes(o_calls, o_genotypes, o_handler, n_iterations=2)
    assert it == kwargs['n_iterations'] - 1


# ---- input validation and the public helpers on caller-supplied tables --------------------------------------
def test_inputs_outside_their_domain_are_refused():
    """The E-step's log is the hot-path form (finite argument >= 1e-4): p_base_wrong outside [0, 1] and caller-supplied
    probability tables with entries outside [0, 1] (or NaN) are refused instead of answered with meaningless numbers."""
    from demuxalot_amd import _lib
    from demuxalot_amd.device import DeviceContext
    ctx = DeviceContext(0)
    try:
        args = (2, 3, 2, np.array([0, 1]), np.array([0, 1]))
        for bad in (1.5, -0.1, np.nan, np.inf):
            with pytest.raises(_lib.DemuxHipError, match=r'p_base_wrong\[1\]'):
                ctx.set_problem(*args, np.array([.1, bad], dtype='f4'), np.zeros(3, dtype='i4'))
        ctx.set_problem(*args, np.array([0., 1.], dtype='f4'), np.zeros(3, dtype='i4'))  # the closed interval is fine
        for bad in (1.0000001, -1e-9, np.nan):
            table = np.full((3, 2), 0.5, dtype=np.float32)
            table[2, 1] = bad
            with pytest.raises(_lib.DemuxHipError, match='outside'):
                ctx.set_probs(table)
        ctx.set_probs(np.array([[0, 1], [.5, .5], [1, 0]], dtype=np.float32))
        logits, _ = ctx.estep(np.zeros(2, dtype=np.float32), with_doublets=False)
        assert np.isfinite(logits).all()
    finally:
        ctx.close()


def test_compute_probs_from_betas_float64_input(oracle):
    """Demultiplexer._compute_probs_from_betas with float64 betas: numpy divides float64 by float64 and rounds
    once (demux.py:267-274), which differs from casting the betas to float32 first."""
    from demuxalot_amd import Demultiplexer
    rng = np.random.default_rng(8)
    v2snp = np.repeat(np.arange(400, dtype=np.int32), rng.integers(1, 4, size=400))
    betas64 = rng.gamma(2.0, 30.0, size=(len(v2snp), 7)) * (rng.random((len(v2snp), 7)) > 0.1)
    want = oracle.probs_from_betas(v2snp, betas64, 0.02)
    got = Demultiplexer._compute_probs_from_betas(v2snp, betas64, 0.02)
    fio.assert_bitwise(got, want, 'float64 betas')
    assert not np.array_equal(want, oracle.probs_from_betas(v2snp, betas64.astype(np.float32), 0.02))
    got32 = Demultiplexer._compute_probs_from_betas(v2snp, betas64.astype(np.float32), 0.02)
    fio.assert_bitwise(got32, oracle.probs_from_betas(v2snp, betas64.astype(np.float32), 0.02), 'float32 betas')


# ---- the sharded front-end on one GPU -----------------------------------------------------------------------
@pytest.mark.parametrize('name', ['f1_synthetic_default.npz', 'f3_small_3.npz', 'f6_shipped_example.npz'])
@pytest.mark.parametrize('reduce_dtype', ['f64', 'f32'])
def test_sharded_entry_points_with_one_rank_communicator(name, reduce_dtype):
    """distributed.learn_genotypes / predict_posteriors with a one-rank RCCL communicator attached (force_comm):
    the whole multi-GPU code path -- sharding of the containers, padded exchange layout, reduce-scatter, sliced
    P-step, all-gather (F1 / F6: SNP groups contiguous) or the all-reduce fallback (F3: scattered) -- with the
    reference's captured outputs as the checker.  One rank's collectives are copies: results stay bit-exact."""
    from demuxalot_amd import distributed
    fx = fio.load(name)
    calls, genotypes, handler = fio.product_inputs(fx)
    kwargs = dict(n_iterations=int(fx['em0_n_iterations']), p_genotype_clip=float(fx['em0_clip']),
                  doublet_prior=float(fx['em0_dp']))
    prior = fx.get('em0_prior_logits')
    learnt, probs_df = distributed.learn_genotypes(calls, genotypes, handler, distributed.SingleProcess(), device=0,
                                                   barcode_prior_logits=prior, reduce_dtype=reduce_dtype, force_comm=True,
                                                   **kwargs)
    fio.assert_bitwise(learnt.variant_betas, fx['em0_learnt_betas'], 'learnt betas')
    fio.assert_bitwise(probs_df.values, fx[f'em0_it{kwargs["n_iterations"] - 1}_probs'], 'posteriors')
    logits_df, p_df = distributed.predict_posteriors(calls, genotypes, handler, distributed.SingleProcess(), device=0,
                                                     p_genotype_clip=float(fx['predict0_clip']),
                                                     doublet_prior=float(fx['predict0_dp']))
    fio.assert_bitwise(logits_df.values, fx['predict0_logits'], 'predict logits')
    fio.assert_bitwise(p_df.values, fx['predict0_probs'], 'predict probs')


def test_learnt_betas_formed_on_the_device():
    """dmx_get_learnt_betas: raw betas (as dmx_set_prior_betas was given them) + the last M-step's addition, one float32 addition per
    element - numpy's `genotypes.get_betas() + genotype_addition` (demux.py:65), bit for bit; refused when the prior table was set as
    such (no raw betas behind it); the front-end hands the downloaded table itself to the learnt genotypes, writeable, no copy."""
    from demuxalot_amd import _lib, synth
    from demuxalot_amd.device import DeviceContext
    p = synth.generate(3000, 2500, 12, seed=77)
    rng = np.random.default_rng(5)
    raw = (rng.gamma(2.0, 3.0, size=(p.n_variants, p.n_genotypes)) * (rng.random((p.n_variants, p.n_genotypes)) > 0.1)).astype(np.float32)
    pen = np.zeros(p.n_genotypes, dtype=np.float32)
    ctx = DeviceContext(0)
    try:
        ctx.set_problem(p.n_barcodes, p.n_variants, p.n_genotypes, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
        mol = np.bincount(p.variant_id, minlength=p.n_variants).astype(np.int64)
        ctx.set_prior_betas(raw, 1.0, True, mol_per_variant=mol, fetch=False)
        _l, _p, addition = ctx.em(3, 0.01, pen, with_doublets=False)
        learnt = ctx.get_learnt_betas()
        fio.assert_bitwise(learnt, raw + addition, 'raw + addition')
        assert learnt.flags.writeable and learnt.flags.owndata
        ctx.set_addition(None)
        fio.assert_bitwise(ctx.get_learnt_betas(), raw, 'addition reset')
        ctx.set_betas(raw)
        with pytest.raises(_lib.DemuxHipError, match='raw betas'):
            ctx.get_learnt_betas()
    finally:
        ctx.close()
