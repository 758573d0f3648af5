"""The guarded E-step (dmx_set_estep_mode(ctx, DMX_ESTEP_GUARDED) / DEMUXALOT_AMD_ESTEP=guarded) on the GPU.

Contract (BASELINE.json north_star): barcode -> donor assignments identical to the reference, posteriors within 1e-5.
The guarded mode runs the tolerance-mode arithmetic (products of 8 float32 terms, one hardware log2 per product) and
PROVES the contract per barcode from a bound on its deviation from the reference (csrc/estep_epilogue.h: estep_guard);
barcodes it cannot prove are redone by the exact kernel inside the same E-step.  So here, unlike in
tests/test_gpu_fast_mode.py, the contract is asserted without exceptions: every posterior, every argmax."""
import numpy as np
import pytest

from tests import fixture_io as fio

pytestmark = pytest.mark.gpu

TOL_POSTERIOR = 1e-5  # BASELINE.json north_star


@pytest.fixture()
def guarded(monkeypatch):
    """Guarded mode for the shared context and (through the environment) for the private contexts the front-end
    creates; restored afterwards."""
    from demuxalot_amd import device
    monkeypatch.setenv('DEMUXALOT_AMD_ESTEP', 'guarded')
    ctx = device.get_context()
    ctx.apply_environment()
    yield ctx
    monkeypatch.setenv('DEMUXALOT_AMD_ESTEP', 'exact')  # the suite's pin (tests/conftest.py)
    ctx.apply_environment()


def addition_bound(fx, reference):
    """What posteriors within 1e-5 of the reference's allow a beta addition to differ by (demux.py:113-118): an entry is the
    sum over the variant's n calls of (p k)^2 with k = 1 - p_base_wrong <= 1, and |(p' k)^2 - (p k)^2| = k^2 |p' - p| (p' + p)
    <= 1e-5 (2 + 1e-5) per call; plus a float32 ulp of the sum for its one rounding (and of the learnt beta = beta + addition).
    Two hundred times tighter than the rtol 1e-3 / atol 1e-4 this file used until round 5 - and a theorem, given the
    posteriors the tests assert first."""
    n_calls = np.bincount(fx['pack_bc_variant_id'], minlength=len(reference)).astype(np.float64)[:, None]
    return n_calls * (TOL_POSTERIOR * (2 + TOL_POSTERIOR)) + 2.0 ** -22 * np.abs(reference.astype(np.float64))


def assert_addition_within_bound(got, reference, fx, what):
    dev = np.abs(got.astype(np.float64) - reference)
    bound = addition_bound(fx, reference)
    assert (dev <= bound).all(), f'{what}: beta addition off by {dev.max():.3g} where {bound.flat[dev.argmax()]:.3g} is allowed'
    return float((dev / np.maximum(bound, 1e-300)).max())


def check_contract(got_probs, ref_probs, what):
    dev = np.abs(got_probs.astype(np.float64) - ref_probs)
    assert (dev <= TOL_POSTERIOR).all(), f'{what}: posterior deviation {dev.max():.3g}'
    assert np.array_equal(got_probs.argmax(axis=1), ref_probs.argmax(axis=1)), f'{what}: assignments differ'
    return float(dev.max())


def test_hardware_log2_of_a_mantissa_is_within_two_ulp():
    """v_log_f32 on EVERY float32 in [0.5, 1) - the only arguments the tolerance / guarded modes give it - against the
    float64 log2: the guard prices it at 2 ulp of a value below 1 (2 x 2^-24); the ISA documents 1 ulp."""
    from demuxalot_amd.device import get_context
    ctx = get_context()
    lo, hi = np.float32(0.5).view(np.uint32), np.float32(1.0).view(np.uint32)
    m = np.arange(int(lo), int(hi), dtype=np.uint32).view(np.float32)
    got = ctx.test_log2_hw(m).astype(np.float64)
    err = np.abs(got - np.log2(m.astype(np.float64)))
    assert err.max() <= 2 * 2.0 ** -24, err.max()
    # and exact on the exponent's side: frexp leaves a mantissa in [0.5, 1), never 1.0
    assert got.max() < 0.0 and got.min() == -1.0
    print(f'v_log_f32 on [0.5, 1): max abs error {err.max():.3e} = {err.max() / 2.0 ** -24:.2f} ulp(<1)')


def test_hardware_log2_of_a_whole_product_is_within_two_ulp_of_its_result():
    """The coarse pass (kernels.hip: coarse_walk) takes v_log_f32 of a 4-term product itself - anything in [1e-16, 2^100) - and prices
    it at 2 ulp of a result below 128 in magnitude (1.53e-5 log2 units; kernels.h: GUARD_PER_CALL_COARSE).  Every binade of that range:
    all 2^23 mantissas of 12 of them, 2^16 mantissas of every other one; the error against float64 log2 stays within 2 ulp OF THE
    RESULT (and therefore below the priced 1.53e-5 everywhere)."""
    from demuxalot_amd.device import get_context
    ctx = get_context()
    worst_abs, worst_ulp = 0.0, 0.0
    dense = {-56, -40, -24, -8, -2, -1, 0, 1, 7, 33, 64, 99}
    for e in range(-56, 100):
        step = 1 if e in dense else 128
        bits = (np.uint32(e + 127) << np.uint32(23)) + np.arange(0, 1 << 23, step, dtype=np.uint32)
        x = bits.view(np.float32)
        got = ctx.test_log2_hw(x).astype(np.float64)
        want = np.log2(x.astype(np.float64))
        err = np.abs(got - want)
        ulp = np.spacing(np.abs(want).astype(np.float32)).astype(np.float64)
        worst_abs = max(worst_abs, float(err.max()))
        worst_ulp = max(worst_ulp, float((err / np.maximum(ulp, 2.0 ** -149)).max()))
    assert worst_abs <= 1.53e-5 and worst_ulp <= 2.0, (worst_abs, worst_ulp)
    print(f'v_log_f32 on [2^-56, 2^100): max abs error {worst_abs:.3e}, {worst_ulp:.2f} ulp of the result')


@pytest.mark.parametrize('name', fio.SMALL + fio.SYNTH)
def test_guarded_mode_meets_the_contract_on_reference_outputs(guarded, name):
    """predict_posteriors and every EM iteration of the golden fixtures (the reference's own outputs).  The first
    E-step of every run sees the importers' table, i.e. the dictionary form (exact); iterations after the first M-step
    run the guarded kernels on tables that follow the reference's within the posteriors' tolerance."""
    from demuxalot_amd import Demultiplexer
    fx = fio.load(name)
    calls, genotypes, handler = fio.product_inputs(fx)
    worst = 0.0
    for i in range(int(fx['n_predict'])):
        dp, clip = float(fx[f'predict{i}_dp']), float(fx[f'predict{i}_clip'])
        _, probs_df = Demultiplexer.predict_posteriors(calls, genotypes, handler, p_genotype_clip=clip, doublet_prior=dp)
        worst = max(worst, check_contract(probs_df.values, fx[f'predict{i}_probs'], f'{name} predict {i}'))
    for i in range(int(fx['n_em'])):
        kwargs = dict(n_iterations=int(fx[f'em{i}_n_iterations']), p_genotype_clip=float(fx[f'em{i}_clip']),
                      doublet_prior=float(fx[f'em{i}_dp']))
        prior = fx.get(f'em{i}_prior_logits')
        stages = list(Demultiplexer.staged_genotype_learning(
            calls, genotypes, handler, barcode_prior_logits=None if prior is None else prior.copy(), **kwargs))
        for it, (probs_df, dbg) in enumerate(stages):
            worst = max(worst, check_contract(probs_df.values, fx[f'em{i}_it{it}_probs'], f'{name} run {i} it {it}'))
            assert_addition_within_bound(dbg['genotype_addition'], fx[f'em{i}_it{it}_addition'], fx, f'{name} run {i} it {it}')
        learnt, last = Demultiplexer.learn_genotypes(
            calls, genotypes, handler, barcode_prior_logits=None if prior is None else prior.copy(), **kwargs)
        check_contract(last.values, fx[f'em{i}_it{kwargs["n_iterations"] - 1}_probs'], f'{name} learn {i}')
        assert_addition_within_bound(learnt.variant_betas, fx[f'em{i}_learnt_betas'], fx, f'{name} learnt betas {i}')
    print(f'{name}: worst posterior deviation {worst:.3g}')


def _estep_in_modes(table, problem, dp, modes=('exact', 'guarded'), prior=None):
    """One E-step on `table` in each mode on a private context: {mode: (logits, probs, guard_stats, form)}."""
    from demuxalot_amd import Demultiplexer
    from demuxalot_amd.device import DeviceContext
    G = problem.n_genotypes
    pen = Demultiplexer._doublet_penalties(G, dp)
    out = {}
    ctx = DeviceContext(0)
    try:
        ctx.set_estep_dictionary('never')  # the kernels under test, not the dictionary form
        ctx.set_problem(problem.n_barcodes, problem.n_variants, G, problem.variant_id, problem.compressed_cb, problem.p_base_wrong, problem.v2snp)
        ctx.set_probs(table)
        for mode in modes:
            ctx.set_estep_mode(mode)
            ctx.reset_timings()
            logits, probs = ctx.estep(pen, with_doublets=dp > 0, prior_logits=prior)
            out[mode] = (logits, probs, ctx.guard_stats(), ctx.estep_form()[0])
    finally:
        ctx.close()
    return out


@pytest.mark.parametrize('G,dp', [(2, 0.), (3, 0.3), (8, 0.35), (16, 0.), (20, 0.25), (32, 0.), (33, 0.), (64, 0.), (64, 0.1),
                                  (100, 0.), (128, 0.), (200, 0.), (22, 0.3), (24, 0.3), (32, 0.25), (45, 0.1), (130, 0.05), (300, 0.), (600, 0.)])
def test_guarded_mode_every_kernel_shape_against_the_exact_mode_and_the_oracle(oracle, G, dp):
    """All lane-group widths and slot counts, and the workgroup-per-barcode forms (K > 1024, doublet tables > 256): posteriors within the
    contract of the exact mode's = the oracle's, argmax identical, on EVERY barcode; the barcodes the guard queued carry
    the exact mode's bits."""
    from demuxalot_amd import synth
    B = 400
    p = synth.generate(n_barcodes=B, n_snps=300, n_genotypes=G, calls_per_barcode=60, doublets=dp > 0, seed=G)
    prob = oracle.probs_from_betas(p.v2snp, p.prior_betas(), 0.01)
    want = oracle.barcode_logits(p.variant_id, p.compressed_cb, p.p_base_wrong, prob, B, dp, log_impl='npsimd')
    out = _estep_in_modes(prob, p, dp)
    exact_logits, exact_probs = out['exact'][:2]
    fio.assert_bitwise(exact_logits, want, 'exact mode vs oracle')
    logits, probs, (redone, total, rows), _ = out['guarded']
    check_contract(probs, exact_probs, f'G={G} dp={dp}')
    assert rows == B and redone == total  # every shape has a guarded kernel (lane-per-option forms and workgroup-per-barcode forms)
    same = (logits.view(np.uint32) == exact_logits.view(np.uint32)).all(axis=1) & (probs.view(np.uint32) == exact_probs.view(np.uint32)).all(axis=1)
    assert same.sum() >= redone  # every queued barcode was rewritten by the exact kernel
    print(f'G={G} dp={dp}: {redone} of {B} barcodes redone exactly, {int(same.sum())} rows bit-identical')


@pytest.mark.parametrize('G,dp', [(64, 0.), (40, 0.), (100, 0.), (200, 0.), (8, 0.35), (20, 0.2), (32, 0.)])
def test_guarded_mode_with_split_rows(oracle, G, dp):
    """Few barcodes with long rows: the 64-lane tolerance kernels cut rows beyond 256 calls into segments walked by separate
    wavefronts (csrc/dmx_api.cpp: build_row_segments, k_estep_join adds the segment sums in a fixed order).  Same
    contract against the exact mode on every barcode; the same bits run after run."""
    from demuxalot_amd import synth
    B = 300
    p = synth.generate(n_barcodes=B, n_snps=3000, n_genotypes=G, calls_per_barcode=700, doublets=dp > 0, seed=50 + G)
    assert np.bincount(p.compressed_cb).max() > 600  # rows beyond 256 calls are cut here (128 CallPairs: the floor of the segment length)
    prob = oracle.probs_from_betas(p.v2snp, p.prior_betas(), 0.01)
    out = _estep_in_modes(prob, p, dp, modes=('exact', 'guarded', 'fast'))
    again = _estep_in_modes(prob, p, dp, modes=('guarded',))
    fio.assert_bitwise(again['guarded'][0], out['guarded'][0], 'guarded logits, second run')
    fio.assert_bitwise(again['guarded'][1], out['guarded'][1], 'guarded posteriors, second run')
    check_contract(out['guarded'][1], out['exact'][1], f'split rows, G={G} dp={dp}')
    redone = out['guarded'][2][0]
    same = (out['guarded'][0].view(np.uint32) == out['exact'][0].view(np.uint32)).all(axis=1)
    assert same.sum() >= redone
    # the unguarded tolerance mode runs the same split walk: its logits are the guarded ones wherever nothing was redone
    differs = (out['fast'][0].view(np.uint32) != out['guarded'][0].view(np.uint32)).any(axis=1)
    assert differs.sum() <= redone


def test_ambiguous_barcodes_are_redone_exactly(oracle):
    """Genotypes in identical pairs: the best two logits of every barcode tie, nothing can be proven about the argmax,
    so every barcode must come back with the exact mode's bits; with half of the genotypes duplicated, many do."""
    from demuxalot_amd import synth
    for G, dup in ((8, 8), (64, 64), (64, 16), (128, 128)):
        B = 600
        p = synth.generate(n_barcodes=B, n_snps=400, n_genotypes=G, calls_per_barcode=80, seed=100 + G + dup)
        prob = oracle.probs_from_betas(p.v2snp, p.prior_betas(), 0.01).copy()
        prob[:, 1:dup:2] = prob[:, 0:dup - 1:2]  # genotype 2j+1 = genotype 2j
        out = _estep_in_modes(prob, p, 0.)
        logits, probs, (redone, _t, rows), _ = out['guarded']
        if dup == G:
            assert redone == B == rows
            fio.assert_bitwise(logits, out['exact'][0], 'all-ambiguous logits')
            fio.assert_bitwise(probs, out['exact'][1], 'all-ambiguous posteriors')
        else:
            check_contract(probs, out['exact'][1], f'G={G} dup={dup}')
            best = out['exact'][1].argmax(axis=1)
            tied = best < dup  # the barcode's best genotype has a twin
            assert redone >= tied.sum()
            fio.assert_bitwise(logits[tied], out['exact'][0][tied], 'tied rows')


def test_guarded_mode_with_prior_logits_and_degenerate_calls(oracle):
    """barcode_prior_logits (float32 and float64), p_base_wrong of exactly 0 and 1, an empty barcode."""
    from demuxalot_amd import synth
    B, G = 300, 20
    p = synth.generate(n_barcodes=B, n_snps=200, n_genotypes=G, calls_per_barcode=40, seed=9)
    p.p_base_wrong[:50] = 0.0
    p.p_base_wrong[50:100] = 1.0
    keep = p.compressed_cb != 7  # barcode 7 has no calls
    p = synth.SyntheticProblem(B, p.n_snps, G, p.v2snp, p.raw_betas, p.variant_id[keep], p.compressed_cb[keep], p.p_base_wrong[keep], p.truth)
    prob = oracle.probs_from_betas(p.v2snp, p.prior_betas(), 0.01)
    rng = np.random.default_rng(4)
    for dtype in (np.float32, np.float64):
        prior = (rng.normal(size=(B, G)) * 3).astype(dtype)
        prior[::5, 3] += 100
        out = _estep_in_modes(prob, p, 0., prior=prior)
        check_contract(out['guarded'][1], out['exact'][1], f'prior {dtype.__name__}')
    out = _estep_in_modes(prob, p, 0.)
    check_contract(out['guarded'][1], out['exact'][1], 'degenerate calls')
    assert np.array_equal(out['guarded'][1][7], np.full(G, 1 / G, dtype=np.float32))  # the empty barcode: ties -> exact redo
    # workgroup-per-barcode form (K = 300 doublet options) with prior logits: no guard for that combination, the exact mode runs
    p24 = synth.generate(n_barcodes=100, n_snps=200, n_genotypes=24, calls_per_barcode=40, doublets=True, seed=10)
    prob24 = oracle.probs_from_betas(p24.v2snp, p24.prior_betas(), 0.01)
    prior24 = (rng.normal(size=(100, 300)) * 3).astype(np.float32)
    out = _estep_in_modes(prob24, p24, 0.3, prior=prior24)
    fio.assert_bitwise(out['guarded'][0], out['exact'][0], 'block form with prior logits: exact')
    assert out['guarded'][2][2] == 0


def test_guarded_em_follows_the_exact_em(oracle):
    """20k x 10k x 64, four EM iterations in either mode.  Per E-step the contract is proven against the exact mode ON
    THE SAME TABLE; across iterations the tables differ by what a 1e-5 change of the posteriors does to the M-step,
    which stays far below the contract here: asserted with the contract's own tolerance."""
    from demuxalot_amd import synth
    from demuxalot_amd.device import DeviceContext
    p = synth.generate(20000, 10000, 64, calls_per_barcode=200, seed=77)
    betas = p.prior_betas()
    pen = np.zeros(64, dtype=np.float32)
    res = {}
    for mode in ('exact', 'guarded'):
        ctx = DeviceContext(0)
        try:
            ctx.set_estep_mode(mode)
            ctx.set_estep_dictionary('never')
            ctx.set_problem(p.n_barcodes, p.n_variants, 64, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
            ctx.set_betas(betas)
            ctx.reset_timings()
            res[mode] = ctx.em(4, 0.01, pen, with_doublets=False) + (ctx.guard_stats(),)
        finally:
            ctx.close()
    dev = check_contract(res['guarded'][1], res['exact'][1], 'guarded EM vs exact EM')
    n_calls = np.bincount(p.variant_id, minlength=p.n_variants).astype(np.float64)[:, None]
    assert (np.abs(res['guarded'][2].astype(np.float64) - res['exact'][2]) <= n_calls * 2.00001e-5 + 2.0 ** -22 * res['exact'][2]).all()
    _last, total, rows = res['guarded'][3]
    assert rows == 4 * p.n_barcodes
    print(f'guarded EM: posteriors within {dev:.3g} of the exact run after 4 iterations, {total} of {rows} barcode rows redone exactly')


@pytest.mark.parametrize('G,dp,B,cpb', [(64, 0., 30000, 300), (12, 0.3, 20000, 200), (24, 0.2, 4000, 60), (128, 0.25, 600, 40)])
def test_guarded_mode_adapts_to_a_workload_it_cannot_prove(G, dp, B, cpb):
    """Worst case of the default mode (include/demux_hip.h: dmx_set_guard_adaptive).  Genotypes in identical pairs: the two
    best logits of every barcode tie, the guard can prove nothing, and the fast pass is wasted on every barcode.  The kernels
    time their two passes on the device; after the first such E-step the next ones run DIRECT - the exact kernel on every
    barcode, bit-identical to the exact mode - and keep counting what the guard would have queued.  Prior logits that settle
    every barcode empty the queue again, and the fast pass returns when - by the device's own timings - it is the cheaper
    way: every decision is checked against the rule F + f E > E (3 % of hysteresis) on the numbers the device reports.
    The lane-per-option kernels (K = 64, and K = 78 with doublets) and the workgroup-per-barcode forms (K = 300, K = 8256)
    all carry the switch.  Without the adaptation nothing ever runs direct."""
    from demuxalot_amd import Demultiplexer, synth
    from demuxalot_amd.device import DeviceContext
    p = synth.generate(B, 2000, G, calls_per_barcode=cpb, doublets=dp > 0, seed=300 + G)
    pen = Demultiplexer._doublet_penalties(G, dp)
    K = len(pen)
    block_shape = K > 1024 or (dp > 0 and K > 256)  # no guard with prior logits there: the exact mode runs
    ctx = DeviceContext(0)
    try:
        ctx.set_estep_dictionary('never')
        ctx.set_estep_packing('never')  # (the packed form is exact and runs unguarded)
        ctx.set_problem(p.n_barcodes, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
        ctx.set_betas(p.prior_betas())
        ctx.set_addition(None)
        table = ctx.probs_from_betas(0.01).copy()
        table[:, 1::2] = table[:, 0:2 * (G // 2):2]  # genotype 2j + 1 = genotype 2j
        ctx.set_probs(table)
        ctx.set_estep_mode('exact')
        logits_e, probs_e = ctx.estep(pen, with_doublets=dp > 0)
        rng = np.random.default_rng(5)
        sharp = np.zeros((B, K), dtype=np.float32)
        sharp[np.arange(B), rng.integers(0, G, size=B)] = 200.0  # prior logits that settle every barcode
        if not block_shape:
            logits_es, probs_es = ctx.estep(pen, with_doublets=dp > 0, prior_logits=sharp)
        ctx.set_estep_mode('guarded')
        ctx.reset_timings()
        history = []
        with_prior = (False, False, False) + ((True,) * 4 + (False, False) if not block_shape else (False,))
        for step, prior in enumerate(with_prior):
            logits_g, probs_g = ctx.estep(pen, with_doublets=dp > 0, prior_logits=sharp if prior else None)
            direct, _steps, would, fast_ms, exact_ms = ctx.guard_state()
            history.append((direct, would, fast_ms, exact_ms))
            want_l, want_p = (logits_es, probs_es) if prior else (logits_e, probs_e)
            check_contract(probs_g, want_p, f'step {step}')
            if direct or not prior:  # (every barcode of the tied table is redone by the exact kernel either way)
                fio.assert_bitwise(logits_g, want_l, f'step {step}: logits')
                fio.assert_bitwise(probs_g, want_p, f'step {step}: posteriors')
        assert history[0][0] is False and history[0][1] == B, history  # nothing could be proven
        assert history[1][0] and history[2][0] and history[1][1] == B, history  # so the next E-steps ran direct, still counting
        assert history[2][3] > 0, history  # ... and measured the exact kernel over all barcodes
        for step in range(1, len(history)):  # every decision against the rule, on the device's own numbers
            was_direct, queued = history[step - 1][0], history[step - 1][1]
            fast_ms, exact_ms = history[step][2], abs(history[step][3])
            if fast_ms > 0 and exact_ms > 0:
                guarded_cost = fast_ms + queued / B * exact_ms
                margin = 0.97 if was_direct else 1.03
                if abs(guarded_cost / (margin * exact_ms) - 1) > 1e-3:
                    assert history[step][0] == (guarded_cost > margin * exact_ms), (step, history)
        if not block_shape:
            assert history[3][1] < 0.05 * B, history  # the sharp prior leaves (almost) nothing to queue
        last, total, rows = ctx.guard_stats()
        assert rows == len(with_prior) * B and total == sum(B if d else w for d, w, _f, _e in history)
        # adaptation off: the fast pass + redo every time
        ctx.set_guard_adaptive(False)
        for _ in range(3):
            _l, probs_g = ctx.estep(pen, with_doublets=dp > 0)
            assert ctx.guard_direct()[0] is False
            check_contract(probs_g, probs_e, 'adaptation off')
    finally:
        ctx.close()
    print(f'G={G} dp={dp}: (direct, queued or would-queue, fast pass ms, exact pass ms) per E-step {history}')


@pytest.mark.parametrize('name', ['f1_synthetic_default.npz', 'f2_synthetic_g4.npz'])
def test_default_mode_through_the_tile_major_mstep_on_reference_outputs_and_twice_the_same_bits(guarded, name):
    """learn_genotypes in the library's default mode with the M-step form of long runs (tile-major records, fixed-point sums)
    on the reference's own fixtures: every yielded iteration within the contract (posteriors 1e-5, assignments identical),
    the learnt betas within what such posteriors allow - and the whole call twice: the same bits (the reference is
    bit-reproducible run to run; so is the default mode since its M-step sums in fixed point)."""
    from demuxalot_amd import Demultiplexer
    fx = fio.load(name)
    calls, genotypes, handler = fio.product_inputs(fx)
    guarded.set_mstep_tiles('always')
    try:
        for i in range(int(fx['n_em'])):
            kwargs = dict(n_iterations=int(fx[f'em{i}_n_iterations']), p_genotype_clip=float(fx[f'em{i}_clip']), doublet_prior=float(fx[f'em{i}_dp']))
            prior = fx.get(f'em{i}_prior_logits')
            runs = []
            for _ in range(2):
                learnt, last = Demultiplexer.learn_genotypes(calls, genotypes, handler,
                                                             barcode_prior_logits=None if prior is None else prior.copy(), **kwargs)
                assert guarded.mstep_form() == ('tiles' if len(genotypes.genotype_names) <= 64 and kwargs['n_iterations'] > 1 else guarded.mstep_form())
                check_contract(last.values, fx[f'em{i}_it{kwargs["n_iterations"] - 1}_probs'], f'{name} learn {i} through the tile-major M-step')
                assert_addition_within_bound(learnt.variant_betas, fx[f'em{i}_learnt_betas'], fx, f'{name} learnt betas {i} (tiles)')
                runs.append((last.values.copy(), learnt.variant_betas.copy()))
            fio.assert_bitwise(runs[1][0], runs[0][0], f'{name} run {i}: posteriors of a second call')
            fio.assert_bitwise(runs[1][1], runs[0][1], f'{name} run {i}: learnt betas of a second call')
    finally:
        guarded.set_mstep_tiles('auto')
