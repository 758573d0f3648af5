"""GPU parity tests (run with -m gpu on an MI355X). Everything goes through the C ABI of
libdemux_hip.so via the Python front-end; the oracle and the golden fixtures are the checkers.

Acceptance from BASELINE.json: barcode->donor assignments bit-identical, posteriors within 1e-5.
The kernels repeat numpy's float32 operation order, so the tests first demand the stronger
property -- bitwise equality of logits, posteriors and beta additions with the reference's own
outputs -- and only report the tolerance numbers alongside."""
import numpy as np
import pytest

from tests import fixture_io as fio

pytestmark = pytest.mark.gpu

TOL_POSTERIOR = 1e-5  # BASELINE.json north_star


@pytest.fixture(scope='module')
def ctx():
    from demuxalot_amd.device import get_context
    return get_context()


def check_posteriors(got_logits, got_probs, ref_logits, ref_probs, what):
    # the contractual gate
    assert np.array_equal(got_probs.argmax(axis=1), ref_probs.argmax(axis=1)), f'{what}: assignments differ'
    assert np.abs(got_probs - ref_probs).max() <= TOL_POSTERIOR, f'{what}: posterior tolerance'
    # the stronger property this implementation is built for
    fio.assert_bitwise(got_logits, ref_logits, f'{what} logits')
    fio.assert_bitwise(got_probs, ref_probs, f'{what} probs')


# ---- numpy-exact float32 building blocks on the device ------------------------------------
def test_device_log_matches_numpy_kernel(ctx, oracle):
    lo, hi = np.float32(1e-5).view(np.int32), np.float32(4.0).view(np.int32)
    bits = np.arange(int(lo), int(hi), 7, dtype=np.int32)
    x = bits.view(np.float32)
    fio.assert_bitwise(ctx.test_log(x), oracle.log_f32(x, impl='npsimd'), 'log')
    special = np.array([0.0, -1.0, np.inf, np.nan, 1e-45, 1e-38, 3e38], dtype=np.float32)
    with np.errstate(all='ignore'):
        want = oracle.log_f32(special, impl='npsimd')
    got = ctx.test_log(special)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(got[~np.isnan(want)], want[~np.isnan(want)])


def test_device_log_hot_path_exhaustive(ctx, oracle):
    """The E-step inlines log with a range-restricted division instead of the generic IEEE sequence.
    Numerator and denominator depend only on the reduced argument, so every float32 in [0.5, 2)
    (all 2^24 mantissa/branch combinations) proves the equality for every positive normal input;
    a strided sweep over [1e-5, 4) adds the exponent handling."""
    bits = np.arange(np.float32(0.5).view(np.int32), np.float32(2.0).view(np.int32), dtype=np.int32)
    x = bits.view(np.float32)
    assert len(x) == 2 ** 24
    fio.assert_bitwise(ctx.test_log_hot(x), oracle.log_f32(x, impl='npsimd'), 'hot log, all mantissas')
    lo, hi = np.float32(1e-5).view(np.int32), np.float32(4.0).view(np.int32)
    x = np.arange(int(lo), int(hi), 11, dtype=np.int32).view(np.float32)
    fio.assert_bitwise(ctx.test_log_hot(x), oracle.log_f32(x, impl='npsimd'), 'hot log, sweep')


def test_device_exp_and_softmax_match_numpy_kernel(ctx, oracle):
    lo, hi = np.float32(-0.0).view(np.uint32), np.float32(-104.5).view(np.uint32)
    bits = np.arange(int(lo), int(hi), 41, dtype=np.uint32)
    x = bits.view(np.float32)
    fio.assert_bitwise(ctx.test_exp(x), oracle._c_unary('npsimd_exp_array', x), 'exp')
    rng = np.random.default_rng(3)
    for K in (1, 2, 7, 8, 9, 20, 36, 64, 127, 128, 129, 135, 210, 255, 256, 257, 528, 2080, 8191, 8192, 8193, 8256):
        logits = (rng.normal(size=(5, K)) * rng.choice([1., 30., 200.])).astype(np.float32)
        fio.assert_bitwise(ctx.test_softmax(logits), oracle.softmax_rows(logits, impl='npsimd'), f'softmax K={K}')


# ---- golden fixtures: the reference's own outputs -----------------------------------------
@pytest.mark.parametrize('name', fio.SMALL + fio.SYNTH)
def test_predict_posteriors_matches_reference(name):
    from demuxalot_amd import Demultiplexer
    fx = fio.load(name)
    calls, genotypes, handler = fio.product_inputs(fx)
    for i in range(int(fx['n_predict'])):
        dp, clip = float(fx[f'predict{i}_dp']), float(fx[f'predict{i}_clip'])
        logits_df, probs_df = Demultiplexer.predict_posteriors(
            calls, genotypes, handler, p_genotype_clip=clip, doublet_prior=dp)
        assert logits_df.index.name == 'BARCODE' and probs_df.index.name == 'BARCODE'
        assert list(logits_df.index) == [str(b) for b in fx['barcodes']]
        assert list(probs_df.columns) == [str(c) for c in fx[f'predict{i}_columns']]
        assert logits_df.values.dtype == np.float32 and probs_df.values.dtype == np.float32
        check_posteriors(logits_df.values, probs_df.values, fx[f'predict{i}_logits'], fx[f'predict{i}_probs'],
                         f'{name} predict dp={dp} clip={clip}')


@pytest.mark.parametrize('name', fio.SMALL + fio.SYNTH)
def test_staged_learning_matches_reference(name):
    from demuxalot_amd import Demultiplexer
    fx = fio.load(name)
    calls, genotypes, handler = fio.product_inputs(fx)
    for i in range(int(fx['n_em'])):
        kwargs = dict(n_iterations=int(fx[f'em{i}_n_iterations']), p_genotype_clip=float(fx[f'em{i}_clip']),
                      doublet_prior=float(fx[f'em{i}_dp']))
        prior = fx.get(f'em{i}_prior_logits')
        stages = list(Demultiplexer.staged_genotype_learning(
            calls, genotypes, handler, barcode_prior_logits=None if prior is None else prior.copy(), **kwargs))
        assert len(stages) == kwargs['n_iterations']
        for it, (probs_df, dbg) in enumerate(stages):
            what = f'{name} run {i} it {it}'
            assert probs_df.index.name is None and list(probs_df.columns) == [str(c) for c in fx[f'em{i}_columns']]
            check_posteriors(dbg['barcode_logits'], probs_df.values, fx[f'em{i}_it{it}_logits'],
                             fx[f'em{i}_it{it}_probs'], what)
            fio.assert_bitwise(dbg['genotype_addition'], fx[f'em{i}_it{it}_addition'], what + ' addition')
            fio.assert_bitwise(dbg['genotype_prior'], fx['pack1_betas'], what + ' prior')
        learnt, last_probs = Demultiplexer.learn_genotypes(
            calls, genotypes, handler, barcode_prior_logits=None if prior is None else prior.copy(), **kwargs)
        assert learnt is not genotypes and type(learnt) is type(genotypes)
        fio.assert_bitwise(learnt.variant_betas, fx[f'em{i}_learnt_betas'], f'{name} run {i} learnt betas')
        fio.assert_bitwise(last_probs.values, fx[f'em{i}_it{kwargs["n_iterations"] - 1}_probs'], 'last probs')
        assert last_probs.index.name is None


def test_learning_from_assignment_matches_reference():
    """The reference's test_demultiplex_start_from_assignment scenario (tests/test_synthetic.py:200-239)."""
    from demuxalot_amd import Demultiplexer
    fx, out = fio.load('f1_synthetic_default.npz'), fio.load('f1_from_assignment.npz')
    calls, genotypes, handler = fio.product_inputs(fx, betas=np.zeros_like(fx['betas']))
    prior = out['em0_prior_logits']
    learnt, probs = Demultiplexer.learn_genotypes(calls, genotypes, handler, barcode_prior_logits=prior)
    n_it = int(out['em0_n_iterations'])
    fio.assert_bitwise(probs.values, out[f'em0_it{n_it - 1}_probs'], 'probs')
    fio.assert_bitwise(learnt.variant_betas, out['em0_learnt_betas'], 'learnt betas')
    # float64 prior logits take numpy's float64 in-place add path: same values here, must still agree
    learnt64, probs64 = Demultiplexer.learn_genotypes(calls, genotypes, handler,
                                                      barcode_prior_logits=prior.astype(np.float64))
    assert np.abs(probs64.values - probs.values).max() <= TOL_POSTERIOR


def test_generator_problem_matches_reference():
    """F5: a benchmark-generator problem (2k barcodes x 5k SNPs x 16 donors, 504k calls, up to 1768 calls
    per variant) run through the reference's full entry points."""
    import hashlib
    from demuxalot_amd import Demultiplexer, synth
    fx = fio.load('f5_generator_2k_5k_16.npz')
    problem = synth.generate(2000, 5000, 16, calls_per_barcode=300, doublets=True, seed=4242)
    h = hashlib.sha256()
    for arr in (problem.variant_id, problem.compressed_cb, problem.p_base_wrong, problem.raw_betas):
        h.update(np.ascontiguousarray(arr).tobytes())
    if h.hexdigest() != str(fx['input_sha256']):
        pytest.skip('numpy on this host generates a different random stream than the fixture was made with')
    calls, genotypes, handler = synth.as_objects(problem)
    v2snp, betas, _mol, bc = Demultiplexer.pack_calls(calls, genotypes, add_data_prior=True)
    fio.assert_bitwise(betas, fx['pack1_betas'], 'prior betas')
    assert np.array_equal(bc['variant_id'], problem.variant_id) and np.array_equal(bc['compressed_cb'], problem.compressed_cb)
    for i in range(2):
        logits, probs = Demultiplexer.predict_posteriors(calls, genotypes, handler, doublet_prior=float(fx[f'predict{i}_dp']))
        check_posteriors(logits.values, probs.values, fx[f'predict{i}_logits'], fx[f'predict{i}_probs'], f'F5 predict {i}')
    learnt, probs = Demultiplexer.learn_genotypes(calls, genotypes, handler, n_iterations=3)
    assert np.array_equal(probs.values.argmax(1), fx['em_probs'].argmax(1))
    assert np.abs(probs.values - fx['em_probs']).max() <= TOL_POSTERIOR
    fio.assert_bitwise(learnt.variant_betas, fx['em_learnt_betas'], 'F5 learnt betas')
    fio.assert_bitwise(probs.values, fx['em_probs'], 'F5 EM posteriors')


@pytest.mark.parametrize('name', fio.SMALL + fio.SYNTH)
def test_device_pack_matches_reference(name):
    """dmx_pack_and_set_problem (matching + de-duplication on the GPU) against the reference's pack_calls."""
    from demuxalot_amd.demux import _pack_on_device
    fx = fio.load(name)
    calls, genotypes, handler = fio.product_inputs(fx)
    for flag in (False, True):
        ctx, betas = _pack_on_device(calls, genotypes, handler.n_barcodes, flag)
        fio.assert_bitwise(betas, fx[f'pack{int(flag)}_betas'], 'prior betas')
        variant, cb, p, count = ctx.get_packed_calls()
        assert np.array_equal(variant, fx['pack_bc_variant_id']) and np.array_equal(cb, fx['pack_bc_cb'])
        fio.assert_bitwise(p, fx['pack_bc_p'], 'p_base_wrong products')
        assert np.array_equal(count, fx['pack_bc_variant_count'])
    # the flat-array entry point (containers of a foreign dtype take it) gives the same problem
    from demuxalot_amd.demux import _flatten_inputs
    (var_chrom, var_pos, var_base), flat = _flatten_inputs(calls, genotypes, False)
    n_matched, n_unique, mol = ctx.pack_and_set_problem(
        handler.n_barcodes, genotypes.n_genotypes, var_chrom, var_pos, var_base, genotypes.get_snp_ids_for_variants(),
        flat['chrom'], flat['pos'], flat['base'], flat['cb'], flat['p'])
    variant2, cb2, p2, count2 = ctx.get_packed_calls()
    assert np.array_equal(variant2, variant) and np.array_equal(cb2, cb) and np.array_equal(count2, count)
    fio.assert_bitwise(p2, p, 'flat entry point')


def test_container_pack_validates_molecule_index():
    from demuxalot_amd import CompressedSNPCalls
    from demuxalot_amd._lib import DemuxHipError
    from demuxalot_amd.device import get_context
    c = CompressedSNPCalls.from_arrays([0, 1], [0, 1, 5], [10, 10, 11], [0, 1, 2], [0.1, 0.1, 0.1])  # molecule 5 of 2
    ctx = get_context()
    keys = (np.zeros(2, np.int32), np.array([10, 11], np.int32), np.array([0, 2], np.uint8))
    with pytest.raises(DemuxHipError, match='molecule_index'):
        ctx.pack_containers_and_set_problem(2, 3, *keys, np.array([0, 1], np.int32),
                                            [(0, c.snp_calls[:c.n_snp_calls], c.molecules[:c.n_molecules])])
    # no containers at all: an empty problem
    assert ctx.pack_containers_and_set_problem(2, 3, *keys, np.array([0, 1], np.int32), [])[:2] == (0, 0)


def test_front_end_asserts_like_reference():
    from demuxalot_amd import Demultiplexer
    fx = fio.load('f3_small_0.npz')
    calls, genotypes, handler = fio.product_inputs(fx)
    with pytest.raises(AssertionError):
        Demultiplexer.predict_posteriors(calls, genotypes, handler, doublet_prior=1.0)
    with pytest.raises(AssertionError, match='wrong shape of priors'):
        Demultiplexer.learn_genotypes(calls, genotypes, handler, barcode_prior_logits=np.zeros((3, 3), dtype='float32'))


# ---- unit entry points on caller-supplied tables vs the oracle -------------------------------
@pytest.mark.parametrize('G,dp', [(2, 0.), (3, 0.3), (8, 0.35), (16, 0.), (20, 0.25), (22, 0.5), (23, 0.2), (32, 0.),
                                  (33, 0.), (64, 0.), (64, 0.1), (100, 0.), (128, 0.), (130, 0.05), (200, 0.), (24, 0.3), (32, 0.25),
                                  (40, 0.2), (45, 0.1), (300, 0.), (600, 0.)])
def test_unit_steps_match_oracle(oracle, G, dp):
    from demuxalot_amd import Demultiplexer, synth
    rng = np.random.default_rng(G * 1000 + int(dp * 100))
    prob_size = synth.generate(n_barcodes=150, n_snps=300, n_genotypes=G, calls_per_barcode=60, doublets=dp > 0, seed=G)
    betas = prob_size.prior_betas(add_data_prior=True)
    betas = (betas * rng.uniform(0.3, 2.0, size=betas.shape)).astype(np.float32)
    names = [f'g{i:03d}' for i in range(G)]
    # P-step
    for clip in (0.01, 0.0, 0.2):
        want = oracle.probs_from_betas(prob_size.v2snp, betas, clip)
        got = Demultiplexer._compute_probs_from_betas(prob_size.v2snp, betas, clip)
        fio.assert_bitwise(got, want, f'probs_from_betas clip={clip}')
    prob = oracle.probs_from_betas(prob_size.v2snp, betas, 0.01)
    # E-step on a shuffled COO (bincount order = given order; float64 re-association only)
    order = rng.permutation(prob_size.n_calls)
    bc = dict(variant_id=prob_size.variant_id[order], compressed_cb=prob_size.compressed_cb[order],
              p_base_wrong=prob_size.p_base_wrong[order])
    want = oracle.barcode_logits(bc['variant_id'], bc['compressed_cb'], bc['p_base_wrong'], prob, 150, dp, log_impl='npsimd')
    got, columns = Demultiplexer.compute_barcode_logits_using_barcode_calls(
        names, bc, doublet_prior=dp, genotype_prob=prob, n_barcodes=150, n_genotypes=G)
    assert columns == oracle.option_names(names, dp)
    fio.assert_bitwise(got, want, f'logits G={G} dp={dp}')


@pytest.mark.parametrize('seed', range(10))
def test_randomised_em_against_oracle(oracle, seed):
    """Random shapes: ragged rows (some barcodes and variants without calls), multi-allelic SNPs,
    G from 1 to 70, with / without doublets and prior logits, 1-4 EM iterations; everything bitwise."""
    from demuxalot_amd import Demultiplexer
    from demuxalot_amd.device import get_context
    rng = np.random.default_rng(1000 + seed)
    G = int(rng.choice([1, 2, 3, 5, 7, 12, 17, 31, 33, 64, 65, 70, 129, 270]))
    B, V = int(rng.integers(5, 400)), int(rng.integers(4, 300))
    dp = float(rng.choice([0., 0., 0.2, 0.45])) if G > 1 else 0.
    K = G if dp == 0 else G * (G + 1) // 2
    n_it = int(rng.integers(1, 5))
    N = int(rng.integers(1, 6000))
    # unique (variant, barcode) pairs, variant-major like the reference's barcode_calls; leave gaps
    pairs = np.unique(np.stack([rng.integers(0, max(1, V - 2), N), rng.integers(0, max(1, B - 3), N)], axis=1), axis=0)
    variant_id, cb = pairs[:, 0].astype(np.int32), pairs[:, 1].astype(np.int32)
    e = (10.0 ** (-rng.integers(3, 45, len(cb)) / 10.0)).astype(np.float32)
    e[rng.random(len(e)) < 0.05] = np.float32(1e-38)
    e[rng.random(len(e)) < 0.02] = np.float32(0.0)
    v2snp = np.sort(rng.integers(0, max(1, V // 2), V)).astype(np.int32)  # SNPs with 1..several variants
    betas = (rng.gamma(0.5, 30.0, size=(V, G)) + 0.01).astype(np.float32)
    betas[rng.random((V, G)) < 0.1] = 0
    prior = (rng.normal(size=(B, K)) * 5).astype(np.float32) if rng.random() < 0.5 else None
    clip = float(rng.choice([0.01, 0.0, 0.1]))
    ctx = get_context()
    ctx.set_problem(B, V, G, variant_id, cb, e, v2snp)
    ctx.set_betas(betas)
    pen = Demultiplexer._doublet_penalties(G, dp)
    logits, probs, addition = ctx.em(n_it, clip, pen, with_doublets=dp != 0, prior_logits=prior)
    packed = dict(variant_id=variant_id, compressed_cb=cb, p_base_wrong=e, betas=betas, v2snp=v2snp)
    hist = oracle.em(packed, B, n_it, clip, dp, prior_logits=None if prior is None else prior.copy(), impl='npsimd')
    what = f'seed {seed}: G={G} B={B} V={V} N={len(cb)} dp={dp} it={n_it} prior={prior is not None} clip={clip}'
    check_posteriors(logits, probs, hist[-1]['logits'], hist[-1]['probs'], what)
    fio.assert_bitwise(addition, hist[-1]['addition'], what + ' addition')


def test_contribution_power_is_honoured(ctx, oracle):
    from demuxalot_amd import Demultiplexer
    fx = fio.load('f3_small_2.npz')
    calls, genotypes, handler = fio.product_inputs(fx)
    packed = oracle.pack(fio.oracle_calls(fx), fio.oracle_geno(fx), add_data_prior=True)
    Demultiplexer.contribution_power = 1.5
    try:
        stages = list(Demultiplexer.staged_genotype_learning(calls, genotypes, handler, n_iterations=2))
    finally:
        Demultiplexer.contribution_power = 2.
    hist = oracle.em(packed, handler.n_barcodes, 2, 0.01, 0., power=1.5)
    assert np.allclose(stages[1][1]['genotype_addition'], hist[1]['addition'], rtol=2e-6, atol=1e-7)


def test_mstep_underflow_floor_is_exact(oracle):
    """With contribution_power == 2 the M-step skips posteriors <= 2^-80 (their squared contribution is +0).
    The problem is built so that posteriors land on both sides of that floor, in (0, 2^-80] included;
    the additions must stay bitwise those of the oracle, which skips nothing.  Small G with full 64-call
    chunks also drives the per-genotype queues of the call-parallel kernel into their overflow path."""
    from demuxalot_amd import synth
    from demuxalot_amd.device import get_context
    for G, cpb in ((64, 160), (6, 400), (33, 250)):
        p = synth.generate(3000, 1500, G, calls_per_barcode=cpb, seed=G + 5)
        betas = p.prior_betas()
        ctx = get_context()
        ctx.set_problem(p.n_barcodes, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
        ctx.set_betas(betas)
        pen = np.zeros(G, dtype=np.float32)
        logits, probs, addition = ctx.em(2, 0.01, pen, with_doublets=False)
        packed = dict(variant_id=p.variant_id, compressed_cb=p.compressed_cb, p_base_wrong=p.p_base_wrong,
                      betas=betas, v2snp=p.v2snp)
        hist = oracle.em(packed, p.n_barcodes, 2, 0.01, 0., impl='npsimd')
        first_pass = hist[0]['probs']
        floor = np.float32(2.0 ** -80)
        assert ((first_pass > 0) & (first_pass <= floor)).sum() > 100, 'no posterior under the floor: test is vacuous'
        assert ((first_pass > floor) & (first_pass < 1e-12)).sum() > 100
        fio.assert_bitwise(addition, hist[-1]['addition'], f'addition G={G}')
        # any other power: the bitmap is rebuilt with the exact `!= 0` rule when the steps are called separately
        ctx.set_addition(None)
        ctx.probs_from_betas(0.01, fetch=False)
        ctx.estep(pen, with_doublets=False)
        got = ctx.mstep(contribution_power=1.0)
        want = oracle.beta_addition(p.variant_id, p.compressed_cb, p.p_base_wrong, first_pass, p.n_variants, G, power=1.0)
        assert np.allclose(got, want, rtol=1e-5, atol=0), f'power 1 G={G}'
        assert (got > 0).sum() == (want > 0).sum()


@pytest.mark.parametrize('G', [63, 70])
def test_summation_modes(oracle, G):
    """Variants with more calls than the work-item length (1024 .. 16384, chosen from the problem size; 1024
    here) are summed from several float64 partial sums.  Exact mode
    (default) redoes, in the reference's order, every sum whose float32 rounding could depend on that: additions
    bit-identical.  Fast mode (dmx_set_exact_additions(0)) accepts the combined sums: an addition may move by one
    float32 ulp when the total sits on a rounding boundary; posteriors stay far inside the 1e-5 tolerance."""
    from demuxalot_amd import synth
    from demuxalot_amd.device import get_context
    B, S = 40000, 20  # ~20 000 calls per variant: twenty work items each (G = 70: the genotype-per-lane M-step)
    p = synth.generate(B, S, G, calls_per_barcode=S, seed=1203)
    betas = p.prior_betas()
    packed = dict(variant_id=p.variant_id, compressed_cb=p.compressed_cb, p_base_wrong=p.p_base_wrong, betas=betas, v2snp=p.v2snp)
    hist = oracle.em(packed, B, 2, 0.0, 0., impl='npsimd')
    ctx = get_context()
    pen = np.zeros(G, dtype=np.float32)
    try:
        for exact in (True, False):
            ctx.set_exact_additions(exact)
            ctx.set_problem(B, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
            ctx.set_betas(betas)
            logits, probs, addition = ctx.em(2, 0.0, pen, with_doublets=False)
            want = hist[-1]['addition']
            if exact:
                fio.assert_bitwise(addition, want, 'exact mode')
                check_posteriors(logits, probs, hist[-1]['logits'], hist[-1]['probs'], 'exact mode')
            else:
                ulps = np.abs(addition.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
                assert ulps.max() <= 1 and (ulps != 0).mean() < 1e-3, (ulps.max(), (ulps != 0).mean())
                assert np.array_equal(probs.argmax(axis=1), hist[-1]['probs'].argmax(axis=1))
                assert np.abs(probs - hist[-1]['probs']).max() <= 1e-5
        # another contribution_power through the same redo path (powf: float32 accuracy, not bits)
        ctx.set_exact_additions(True)
        _, _, add15 = ctx.em(2, 0.0, pen, with_doublets=False, contribution_power=1.5)
        want15 = oracle.em(packed, B, 2, 0.0, 0., power=1.5, impl='npsimd')[-1]['addition']
        assert np.allclose(add15, want15, rtol=2e-6, atol=1e-30)
    finally:
        ctx.set_exact_additions(True)


# ---- edge cases --------------------------------------------------------------------------------------
def test_empty_and_degenerate_inputs(ctx, oracle):
    """No calls at all, a single genotype, barcodes without calls, variants without calls."""
    from demuxalot_amd import BarcodeHandler, CompressedSNPCalls, Demultiplexer, ProbabilisticGenotypes
    from scipy.special import softmax
    g = ProbabilisticGenotypes(['A', 'B', 'C'])
    g.var2varid = {('chr1', 5, 'A'): 0, ('chr1', 5, 'C'): 1, ('chr1', 9, 'G'): 2, ('chr1', 9, 'T'): 3}
    g.variant_betas = np.array([[5, 1, 1], [1, 5, 5], [9, 0, 2], [0, 9, 7]], dtype=np.float32)
    handler = BarcodeHandler(['b0', 'b1', 'b2', 'b3'])
    empty = {'chr1': CompressedSNPCalls.from_arrays([], [], [], [], [])}
    for dp in (0., 0.35):
        logits, probs = Demultiplexer.predict_posteriors(empty, g, handler, doublet_prior=dp)
        pen = Demultiplexer._doublet_penalties(3, dp)
        assert np.array_equal(logits.values, np.tile(pen, (4, 1)))
        fio.assert_bitwise(probs.values, softmax(np.tile(pen, (4, 1)), axis=1), 'softmax of the penalties')
    learnt, probs = Demultiplexer.learn_genotypes(empty, g, handler, n_iterations=3)
    assert np.array_equal(learnt.variant_betas, g.variant_betas) and np.allclose(probs.values, 1 / 3)
    # one call only, on barcode b2; G = 1
    one = {'chr1': CompressedSNPCalls.from_arrays([2], [0], [9], [3], [0.05])}
    g1 = ProbabilisticGenotypes(['only'])
    g1.var2varid = dict(g.var2varid)
    g1.variant_betas = g.variant_betas[:, :1].copy()
    logits, probs = Demultiplexer.predict_posteriors(one, g1, handler, doublet_prior=0.)
    assert np.array_equal(probs.values, np.ones((4, 1), dtype=np.float32)) and logits.values[2, 0] < 0
    assert (logits.values[[0, 1, 3], 0] == 0).all()
    learnt, _ = Demultiplexer.learn_genotypes(one, g1, handler, n_iterations=2)
    keep = np.float32(1) - np.float32(0.05)
    expect = g1.variant_betas.copy()
    expect[3, 0] += np.float32(keep * keep)
    fio.assert_bitwise(learnt.variant_betas, expect, 'single-call addition')


def test_c_abi_rejects_bad_calls():
    from demuxalot_amd import _lib
    from demuxalot_amd.device import DeviceContext
    ctx = DeviceContext(0)
    try:
        with pytest.raises(_lib.DemuxHipError, match='outside'):
            ctx.set_problem(2, 3, 2, np.array([0, 5]), np.array([0, 1]), np.array([.1, .1], dtype='f4'), np.zeros(3, dtype='i4'))
        with pytest.raises(_lib.DemuxHipError, match='outside'):
            ctx.set_problem(2, 3, 2, np.array([0, 1]), np.array([0, 2]), np.array([.1, .1], dtype='f4'), np.zeros(3, dtype='i4'))
        with pytest.raises(_lib.DemuxHipError, match='call order'):
            _lib.check(_lib.load().dmx_set_betas(ctx._h, None))  # no problem resident after the failed uploads
        ctx.set_problem(2, 3, 2, np.array([0, 1]), np.array([0, 1]), np.array([.1, .1], dtype='f4'), np.zeros(3, dtype='i4'))
        with pytest.raises(_lib.DemuxHipError, match='call order'):
            ctx.estep(np.zeros(2, dtype=np.float32), with_doublets=False)  # no probabilities yet
        with pytest.raises(_lib.DemuxHipError, match='call order'):
            ctx.mstep()
        with pytest.raises(_lib.DemuxHipError, match='call order'):
            ctx.probs_from_betas(0.01)  # no betas yet
        ctx.set_betas(np.ones((3, 2), dtype=np.float32))
        ctx.probs_from_betas(0.01)
        logits, probs = ctx.estep(np.zeros(2, dtype=np.float32), with_doublets=False)
        assert np.isfinite(logits).all() and np.allclose(probs.sum(axis=1), 1)
        assert ctx.device_bytes() > 0
    finally:
        ctx.close()


def test_repeated_problems_do_not_leak_device_memory():
    from demuxalot_amd import Demultiplexer, synth
    from demuxalot_amd.device import DeviceContext
    ctx = DeviceContext(0)
    try:
        sizes = []
        for rep in range(6):
            p = synth.generate(300 + 50 * (rep % 2), 200, 8 + 8 * (rep % 3), calls_per_barcode=40, seed=rep)
            ctx.set_problem(p.n_barcodes, p.n_variants, p.n_genotypes, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
            ctx.set_prior_betas(p.raw_betas, 1.0, True, mol_per_variant=np.bincount(p.variant_id, minlength=p.n_variants))
            ctx.em(2, 0.01, Demultiplexer._doublet_penalties(p.n_genotypes, 0.2), with_doublets=True)
            sizes.append((p.n_barcodes, p.n_genotypes, ctx.device_bytes()))
        # identical shapes must account for identical bytes (nothing accumulates across uploads)
        by_shape = {}
        for b, g, n in sizes:
            by_shape.setdefault((b, g), set()).add(n)
        assert all(len(v) == 1 for v in by_shape.values()), sizes
        # released blocks stay with the context for its next allocations; trim_cache hands them back, and the context
        # goes on working (results of the same problem unchanged)
        pen = Demultiplexer._doublet_penalties(p.n_genotypes, 0.2)
        before = ctx.em(2, 0.01, pen, with_doublets=True)
        assert ctx.trim_cache() > 0 and ctx.trim_cache() == 0
        ctx.set_problem(p.n_barcodes, p.n_variants, p.n_genotypes, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
        ctx.set_prior_betas(p.raw_betas, 1.0, True, mol_per_variant=np.bincount(p.variant_id, minlength=p.n_variants))
        after = ctx.em(2, 0.01, pen, with_doublets=True)
        for x, y in zip(before, after):
            assert np.array_equal(x, y)
    finally:
        ctx.close()


# ---- multi-GPU plumbing on one GPU -----------------------------------------------------------------
@pytest.mark.parametrize('reduce_dtype', ['f64', 'f32'])
def test_rccl_communicator_single_rank(reduce_dtype):
    """A one-rank RCCL communicator exercises the whole collective path (dlopen of librccl, ncclGetUniqueId /
    ncclCommInitRank, padded exchange layout, partial sums -> ncclReduceScatter -> float32 slice -> sliced P-step ->
    ncclAllGather) on the one GPU a test box has; with one rank the collectives are copies, so results stay
    bit-exact.  The communicator is attached AFTER the problem here (the E-step records are re-laid in place) and
    before it in ShardedEM below."""
    from demuxalot_amd.device import DeviceContext
    from demuxalot_amd.distributed import ShardedEM
    from demuxalot_amd import Demultiplexer
    fx = fio.load('f3_small_3.npz')
    calls, genotypes, handler = fio.product_inputs(fx)
    v2snp, betas, _mol, bc = Demultiplexer.pack_calls(calls, genotypes, add_data_prior=True)
    ctx = DeviceContext(0)
    try:
        ctx.set_problem(handler.n_barcodes, len(v2snp), genotypes.n_genotypes, bc['variant_id'], bc['compressed_cb'],
                        bc['p_base_wrong'], v2snp)
        ctx.set_betas(betas)
        ctx.comm_init(0, 1, DeviceContext.new_unique_id(), reduce_dtype=reduce_dtype)
        n_it = int(fx['em0_n_iterations'])
        logits, probs, addition = ctx.em(n_it, float(fx['em0_clip']), np.zeros(genotypes.n_genotypes, dtype=np.float32),
                                         with_doublets=False)
        fio.assert_bitwise(probs, fx[f'em0_it{n_it - 1}_probs'], 'probs through the RCCL path')
        fio.assert_bitwise(addition, fx[f'em0_it{n_it - 1}_addition'], 'addition through the RCCL path')
        # F3's SNPs have scattered variants: the exchange is the all-reduce fallback (one per M-step); with contiguous
        # SNP groups it would be an all-gather per P-step plus a reduce-scatter per M-step
        from demuxalot_amd.distributed import exchange_slices
        sliced = exchange_slices(v2snp, 1)[2]
        assert not sliced and ctx.timings()['allreduce']['launches'] == n_it - 1
    finally:
        ctx.close()
    # the sharded front-end with world size 1, without and with a (one-rank) communicator, gives the same rows
    from demuxalot_amd.distributed import SingleProcess
    for force in (False, True):
        em = ShardedEM(SingleProcess(), handler.n_barcodes, v2snp, betas, bc['variant_id'], bc['compressed_cb'],
                       bc['p_base_wrong'], device=0, reduce_dtype=reduce_dtype, force_comm=force)
        probs1, addition1 = em.learn(n_it, float(fx['em0_clip']), np.zeros(genotypes.n_genotypes, dtype=np.float32), False)
        em.ctx.close()
        fio.assert_bitwise(probs1, probs, f'ShardedEM world=1 comm={force}')
        fio.assert_bitwise(addition1, addition, f'ShardedEM world=1 comm={force} addition')
    # variants cut into several work items (order-sensitive sums redone exactly): the collective path
    # must give what the single-stream path gives; with float64 on the wire that is bit for bit
    from demuxalot_amd import synth
    from demuxalot_amd.device import get_context
    p = synth.generate(40000, 20, 63, calls_per_barcode=20, seed=1203)
    prior = p.prior_betas()
    pen = np.zeros(63, dtype=np.float32)
    plain = get_context()
    plain.set_problem(p.n_barcodes, p.n_variants, 63, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
    plain.set_betas(prior)
    _, probs_plain, add_plain = plain.em(2, 0.0, pen, with_doublets=False)
    ctx = DeviceContext(0)
    try:
        ctx.set_problem(p.n_barcodes, p.n_variants, 63, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
        ctx.set_betas(prior)
        ctx.comm_init(0, 1, DeviceContext.new_unique_id(), reduce_dtype=reduce_dtype)
        _, probs_coll, add_coll = ctx.em(2, 0.0, pen, with_doublets=False)
    finally:
        ctx.close()
    if reduce_dtype == 'f64':
        fio.assert_bitwise(add_coll, add_plain, 'multi-item variants through the RCCL path')
        fio.assert_bitwise(probs_coll, probs_plain, 'multi-item variants through the RCCL path: posteriors')
    else:
        assert np.allclose(add_coll, add_plain, rtol=3e-7, atol=0)


def test_variant_sharded_exchange_through_a_one_rank_rccl_communicator(monkeypatch):
    """The variant-sharded M-step forced onto ONE rank: its set-up all-gather of the call records, the per-iteration grouped
    all-gathers of the posterior tables (ncclGroupStart / ncclGroupEnd) and the genotype_prob all-gather run through a real
    RCCL communicator on the one GPU a test box has.  With one rank the collectives are copies: everything stays
    bit-identical to the plain context, in the exact and (same kernels, same order) the guarded mode."""
    from demuxalot_amd import synth
    from demuxalot_amd.device import DeviceContext, get_context
    p = synth.generate(30000, 3000, 48, calls_per_barcode=40, seed=1717)
    prior = p.prior_betas()
    pen = np.zeros(48, dtype=np.float32)
    for mode in ('exact', 'guarded'):
        plain = get_context()
        plain.set_estep_mode(mode)
        plain.set_exact_additions(mode == 'exact')
        try:
            plain.set_problem(p.n_barcodes, p.n_variants, 48, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
            plain.set_betas(prior)
            _, probs_plain, add_plain = plain.em(3, 0.01, pen, with_doublets=False)
        finally:
            plain.apply_environment()
        monkeypatch.setenv('DEMUXALOT_AMD_EXCHANGE', 'variant')
        ctx = DeviceContext(0)
        try:
            ctx.set_estep_mode(mode)
            ctx.set_exact_additions(mode == 'exact')
            ctx.comm_init(0, 1, DeviceContext.new_unique_id(), reduce_dtype='f64')
            ctx.set_problem(p.n_barcodes, p.n_variants, 48, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
            assert ctx.exchange_mode() == 'variant'
            ctx.set_betas(prior)
            _, probs_coll, add_coll = ctx.em(3, 0.01, pen, with_doublets=False)
            assert ctx.timings()['allreduce']['launches'] >= 2
        finally:
            ctx.close()
            monkeypatch.delenv('DEMUXALOT_AMD_EXCHANGE')
        if mode == 'exact':
            fio.assert_bitwise(add_coll, add_plain, 'variant-sharded M-step through RCCL: additions')
            fio.assert_bitwise(probs_coll, probs_plain, 'variant-sharded M-step through RCCL: posteriors')
        else:  # sums of float64 in another order: a float32 rounding tie at most
            assert np.allclose(add_coll, add_plain, rtol=3e-7, atol=0)
            assert np.array_equal(probs_coll.argmax(1), probs_plain.argmax(1)) and np.abs(probs_coll - probs_plain).max() <= 1e-5


# ---- mid/large sizes ------------------------------------------------------------------------------
def test_midsize_em_matches_oracle(oracle):
    """20k barcodes x 10k SNPs x 64 genotypes (N ~ 3.6M): three EM iterations against the numpy oracle."""
    from demuxalot_amd import synth
    from demuxalot_amd.device import get_context
    p = synth.generate(20000, 10000, 64, calls_per_barcode=200, seed=77)
    betas = p.prior_betas()
    ctx = get_context()
    ctx.set_problem(p.n_barcodes, p.n_variants, 64, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
    ctx.set_betas(betas)
    pen = np.zeros(64, dtype=np.float32)
    logits, probs, addition = ctx.em(3, 0.01, pen, with_doublets=False)
    packed = dict(variant_id=p.variant_id, compressed_cb=p.compressed_cb, p_base_wrong=p.p_base_wrong,
                  betas=betas, v2snp=p.v2snp)
    hist = oracle.em(packed, p.n_barcodes, 3, 0.01, 0., impl='npsimd')
    check_posteriors(logits, probs, hist[-1]['logits'], hist[-1]['probs'], 'midsize EM')
    fio.assert_bitwise(addition, hist[-1]['addition'], 'midsize addition')
    best, best_p = ctx.get_assignments()
    assert np.array_equal(best, probs.argmax(axis=1)) and np.array_equal(best_p, probs.max(axis=1))


def test_doublets_block_kernel_matches_oracle(oracle):
    """Wide option tables (workgroup-per-barcode E-step: option tiles of up to 17 per thread, then k_softmax_rows):
    40 genotypes with doublets (K = 820, one tile), 128 (K = 8256, two tiles), 140 / 190 / 270 (K = 9870 / 18 145 /
    36 585, several tiles; rows staged in LDS for the softmax) and 285 (K = 40 755: the row does not fit LDS, the
    softmax works through global memory)."""
    from demuxalot_amd import synth
    from demuxalot_amd.device import get_context
    from demuxalot_amd import Demultiplexer
    for G, B in ((40, 300), (128, 24), (140, 16), (190, 12), (270, 9), (285, 6)):
        p = synth.generate(B, 500, G, calls_per_barcode=80 if G < 140 else 30, doublets=True, seed=G)
        betas = p.prior_betas()
        ctx = get_context()
        ctx.set_problem(B, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
        ctx.set_betas(betas)
        pen = Demultiplexer._doublet_penalties(G, 0.3)
        logits, probs, addition = ctx.em(2, 0.01, pen, with_doublets=True)
        packed = dict(variant_id=p.variant_id, compressed_cb=p.compressed_cb, p_base_wrong=p.p_base_wrong,
                      betas=betas, v2snp=p.v2snp)
        hist = oracle.em(packed, B, 2, 0.01, 0.3, impl='npsimd')
        check_posteriors(logits, probs, hist[-1]['logits'], hist[-1]['probs'], f'doublets G={G}')
        fio.assert_bitwise(addition, hist[-1]['addition'], f'doublets G={G} addition')


@pytest.mark.parametrize('G,cpb,power', [(64, 160, 2.), (6, 400, 2.), (33, 250, 2.), (20, 120, 1.5)])
def test_mstep_wide_addresses_are_the_same_sums(G, cpb, power):
    """The call-parallel M-step has two forms of its loads: 32-bit buffer offsets (tables below 4 GiB) and 64-bit
    addresses (the largest problems).  dmx_set_mstep_wide_addresses selects the second at any size: both must give
    the same bits - on problems with sparse, multi-posterior and dense calls and queue overflows (small G)."""
    from demuxalot_amd import synth
    from demuxalot_amd.device import get_context
    p = synth.generate(3000, 1500, G, calls_per_barcode=cpb, seed=G + 5)
    ctx = get_context()
    ctx.set_problem(p.n_barcodes, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
    ctx.set_betas(p.prior_betas())
    ctx.set_addition(None)
    ctx.probs_from_betas(0.01, fetch=False)
    _logits, probs = ctx.estep(np.zeros(G, dtype=np.float32), with_doublets=False)
    nnz = (probs > np.float32(2.0 ** -80)).sum(1) if power == 2. else (probs != 0).sum(1)
    assert (nnz == 1).any() and ((nnz > 1) & (nnz <= 4)).any() and (nnz > 4).any(), 'the problem must mix the three kinds of calls'
    narrow = ctx.mstep(power)
    try:
        ctx.set_mstep_wide_addresses(True)
        wide = ctx.mstep(power)
    finally:
        ctx.set_mstep_wide_addresses(False)
    fio.assert_bitwise(wide, narrow, f'wide vs 32-bit M-step loads, G={G}')
    assert np.array_equal(narrow, ctx.mstep(power))


def test_containers_on_chromosomes_without_variants():
    """The front-end stages the containers before it knows which chromosomes carry variants (the upload runs beside the
    walk of var2varid): an empty container on such a chromosome changes nothing, one with calls trips the same assertion
    as the reference's (demux.py:339-341, 359), and the raw ABI refuses staged calls without a chromosome."""
    from demuxalot_amd import CompressedSNPCalls, Demultiplexer, _lib
    from demuxalot_amd.device import DeviceContext
    fx = fio.load('f2_synthetic_g4.npz')
    calls, genotypes, handler = fio.product_inputs(fx)
    want = Demultiplexer.predict_posteriors(calls, genotypes, handler, doublet_prior=0.)
    empty = CompressedSNPCalls.from_arrays(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.uint8),
                                           np.zeros(0, np.float32))
    with_empty = dict([('chrNone', empty)] + list(calls.items()) + [('chrZ', empty)])
    got = Demultiplexer.predict_posteriors(with_empty, genotypes, handler, doublet_prior=0.)
    for a, b in zip(got, want):
        fio.assert_bitwise(a.values, b.values, 'an empty container on a chromosome without variants')
    stray = CompressedSNPCalls.from_arrays(np.zeros(1, np.int32), np.zeros(1, np.int32), np.full(1, 77, np.int32), np.zeros(1, np.uint8),
                                           np.full(1, 0.01, np.float32))
    with pytest.raises(AssertionError):
        Demultiplexer.predict_posteriors(dict(list(calls.items()) + [('chrZ', stray)]), genotypes, handler, doublet_prior=0.)
    # and the next call on the shared context works as if nothing had been staged
    again = Demultiplexer.predict_posteriors(calls, genotypes, handler, doublet_prior=0.)
    fio.assert_bitwise(again[1].values, want[1].values, 'after the refused call')
    with DeviceContext(0) as ctx:
        ctx.stage_containers([(0, stray.snp_calls[:1], stray.molecules[:1])])
        with pytest.raises(_lib.DemuxHipError, match='chromosome without variants'):
            ctx.pack_staged_and_set_problem(handler.n_barcodes, 2, np.zeros(1, np.int32), np.full(1, 77, np.int32), np.zeros(1, np.uint8),
                                            np.zeros(1, np.int32), [-1])
        with pytest.raises(_lib.DemuxHipError, match='call order'):  # nothing staged any more
            ctx.pack_staged_and_set_problem(handler.n_barcodes, 2, np.zeros(1, np.int32), np.full(1, 77, np.int32), np.zeros(1, np.uint8),
                                            np.zeros(1, np.int32), [0])
