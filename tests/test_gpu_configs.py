"""GPU parity at the sizes BASELINE.json names (run with -m gpu on an MI355X).

configs[1]  20k x 20k x 8, predict_posteriors, doublet_prior .35          -> whole result vs the oracle, bitwise
configs[2]  200k x 100k x 32, learn_genotypes                             -> every step of three EM iterations
configs[3]  200k x 100k x 64 (the headline shape)                          -> same, plus size-independent properties
configs[4]  1M x 650k x 128 with doublets: one rank's share (130k barcodes) -> sampled rows + M-step rows

At the full sizes the oracle cannot redo the whole problem in seconds, so each step is checked against the
oracle GIVEN THE GPU'S OWN PREVIOUS STATE (the steps are pure functions of it):
  P-step   the whole [V, G] table from prior + the GPU's addition                       (bitwise)
  E-step   logits / posteriors of sampled barcode ranges from that table (rows are independent) (bitwise)
  M-step   addition rows of EVERY variant that is cut into several work items (the 16 384-call items and the
           exact in-order redo), the hottest single-item variants and 1 000 random ones, from the
           GPU's posteriors with np.bincount                                             (bitwise)
"""
import numpy as np
import pytest

from tests import fixture_io as fio
from tests.test_gpu_parity import check_posteriors

pytestmark = pytest.mark.gpu


def item_calls_for(n_calls):
    """Work-item length of the M-step for a problem of n_calls calls (csrc/kernels.h: item_calls_for)."""
    length = 1024
    while length < 16384 and length * 6000 < n_calls:
        length *= 2
    return length


def variants_to_check(p, rng, n_random=1000, n_hot=20):
    """(sorted variant ids to recompute, number of them that span several work items)."""
    counts = np.bincount(p.variant_id, minlength=p.n_variants)
    multi = np.flatnonzero(counts > item_calls_for(p.n_calls))
    hot = np.argsort(counts)[-n_hot:]
    some = rng.choice(p.n_variants, size=min(n_random, p.n_variants), replace=False)
    return np.unique(np.concatenate([multi, hot, some])), len(multi)


def addition_rows(p, post_singlets, variants, power=2.):
    """demux.py:113-118 restricted to `variants` (sorted): calls keep the reference's order (variant-major,
    barcodes ascending), so np.bincount adds them exactly as the reference's full pass does."""
    sel = np.flatnonzero(np.isin(p.variant_id, variants))
    idx = np.searchsorted(variants, p.variant_id[sel])
    cb = p.compressed_cb[sel]
    keep = 1 - p.p_base_wrong[sel]
    G = post_singlets.shape[1]
    out = np.zeros((len(variants), G), dtype=np.float32)
    for g in range(G):
        w = post_singlets[cb, g] * keep
        w **= power
        out[:, g] = out[:, g] + np.bincount(idx, weights=w, minlength=len(variants))
    return out


def check_e_rows(ctx, oracle, p, prob, lo, hi, doublet_prior, what):
    v, cb, e = p.subset_barcodes(lo, hi)
    # the oracle builds one full-length column per option: hand it only the rows these barcodes touch
    rows, v = np.unique(v, return_inverse=True)
    want_logits = oracle.barcode_logits(v, cb, e, prob[rows], hi - lo, doublet_prior, log_impl='npsimd')
    want_post = oracle.softmax_rows(want_logits, impl='npsimd')
    check_posteriors(ctx.get_block('logits', lo, hi), ctx.get_block('probs', lo, hi), want_logits, want_post, what)


def staged_em_against_oracle(oracle, p, n_iterations, doublet_prior, samples, seed, p_rows=None):
    """Drives P / E / M by hand through the C ABI and checks every step as described in the module docstring.
    Returns (ctx, last addition, posteriors of the singlet columns)."""
    from demuxalot_amd import Demultiplexer
    from demuxalot_amd.device import get_context
    rng = np.random.default_rng(seed)
    G = p.n_genotypes
    prior = p.prior_betas()
    pen = Demultiplexer._doublet_penalties(G, doublet_prior)
    variants, n_multi = variants_to_check(p, rng)
    ctx = get_context()
    ctx.set_problem(p.n_barcodes, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
    ctx.set_betas(prior)
    ctx.set_addition(None)
    addition = np.zeros_like(prior)
    singlets = None
    for it in range(n_iterations):
        prob = ctx.probs_from_betas(0.01)
        for v0, v1 in (p_rows or [(0, p.n_variants)]):  # whole SNP groups (variants 2s, 2s+1)
            want = oracle.probs_from_betas(p.v2snp[v0:v1] - p.v2snp[v0], (prior + addition)[v0:v1], 0.01)
            fio.assert_bitwise(prob[v0:v1], want, f'it {it} P-step rows [{v0},{v1})')
        ctx.estep(pen, with_doublets=doublet_prior != 0, fetch_logits=False, fetch_probs=False)
        for lo, hi in samples[it % len(samples)]:
            check_e_rows(ctx, oracle, p, prob, lo, hi, doublet_prior, f'it {it} E rows [{lo},{hi})')
        singlets = ctx.get_block('probs', 0, p.n_barcodes, 0, G)
        addition = ctx.mstep(2.)
        fio.assert_bitwise(addition[variants], addition_rows(p, singlets, variants), f'it {it} M-step rows')
    return ctx, addition, singlets, n_multi


# ---- configs[1] ------------------------------------------------------------------------------------------
def test_config1_predict_20k_20k_8_doublets(oracle):
    """20k barcodes x 20k SNPs x 8 donors (N ~ 7.4 M), doublet_prior .35 (K = 36): the whole predict_posteriors
    result through the Python front-end (objects in, DataFrames out) against the oracle, bitwise."""
    from demuxalot_amd import Demultiplexer, synth
    p = synth.generate(20_000, 20_000, 8, doublets=True, seed=1235)
    calls, genotypes, handler = synth.as_objects(p)
    logits_df, probs_df = Demultiplexer.predict_posteriors(calls, genotypes, handler, doublet_prior=0.35)
    packed = dict(variant_id=p.variant_id, compressed_cb=p.compressed_cb, p_base_wrong=p.p_base_wrong,
                  betas=p.prior_betas(add_data_prior=False), v2snp=p.v2snp)
    want_logits, want_probs, _ = oracle.predict(packed, p.n_barcodes, 0.01, 0.35, impl='npsimd')
    assert logits_df.shape == (20_000, 36) and list(probs_df.columns[:2]) == ['Donor001', 'Donor002']
    assert probs_df.columns[8] == 'Donor001+Donor002'
    check_posteriors(logits_df.values, probs_df.values, want_logits, want_probs, 'configs[1] predict')
    # the device-side reductions users apply to this matrix agree with pandas on it
    from demuxalot_amd.device import get_context
    best, best_p = get_context().get_assignments()
    assert np.array_equal(best, probs_df.values.argmax(axis=1)) and np.array_equal(best_p, probs_df.values.max(axis=1))


# ---- configs[2] ------------------------------------------------------------------------------------------
def test_config2_em_200k_100k_32(oracle):
    """200k x 100k x 32 (N ~ 78 M): three EM iterations step by step, then the fused driver (dmx_em) must
    reproduce the staged run bit for bit."""
    from demuxalot_amd import synth
    p = synth.generate(200_000, 100_000, 32, seed=1236)
    samples = [[(0, 3000), (150_000, 152_000)], [(60_000, 62_500)], [(197_500, 200_000)]]
    ctx, addition, singlets, n_multi = staged_em_against_oracle(oracle, p, 3, 0., samples, seed=11)
    assert n_multi >= 20, n_multi  # the workload does have variants of several 16 384-call items
    pen = np.zeros(32, dtype=np.float32)
    _, probs_fused, addition_fused = ctx.em(4, 0.01, pen, with_doublets=False, fetch_logits=False)
    # dmx_em skips the dead last M-step: after 4 iterations its addition is the one of the staged 3rd M-step
    fio.assert_bitwise(addition_fused, addition, 'fused driver: addition')
    assert np.abs(probs_fused.sum(axis=1) - 1).max() < 1e-5
    # configs[2] as written: learn_genotypes with 10 EM iterations.  The staged run goes on for six more iterations
    # (steps driven one by one through the C ABI, each a pure function of the previous state that the three checked
    # iterations above pin against the oracle); the fused driver's 10 iterations must land on the same bits.
    ctx.set_addition(addition)
    for _ in range(6):
        ctx.probs_from_betas(0.01, fetch=False)
        ctx.estep(pen, with_doublets=False, fetch_logits=False, fetch_probs=False)
        addition = ctx.mstep(2.)
    ctx.probs_from_betas(0.01, fetch=False)
    _, probs_staged10 = ctx.estep(pen, with_doublets=False, fetch_logits=False)
    _, probs_fused10, addition_fused10 = ctx.em(10, 0.01, pen, with_doublets=False, fetch_logits=False)
    fio.assert_bitwise(addition_fused10, addition, 'fused driver, 10 iterations: addition')
    fio.assert_bitwise(probs_fused10, probs_staged10, 'fused driver, 10 iterations: posteriors')


# ---- configs[3] shape on one GPU -------------------------------------------------------------------------
@pytest.fixture(scope='module')
def problem64():
    from demuxalot_amd import synth
    return synth.generate(200_000, 100_000, 64, seed=1237)


@pytest.mark.parametrize('schedule', ['auto', 'tiled'])
def test_config3_em_200k_100k_64_steps(oracle, problem64, schedule):
    """The headline shape (200k x 100k x 64, N ~ 78.6 M, hottest variant 72 863 calls = five work items):
    two EM iterations step by step; once with the default work distribution (one barcode per wavefront in the
    exact mode) and once with the tile-major schedule forced (bins of barcodes walked variant tile by tile,
    accumulators parked in LDS): the sums must be bit-identical under either."""
    from demuxalot_amd.device import get_context
    samples = [[(0, 2500), (100_000, 101_500)], [(198_000, 200_000)]]
    get_context().set_estep_schedule(schedule)
    try:
        _ctx, _addition, _singlets, n_multi = staged_em_against_oracle(oracle, problem64, 2, 0., samples, seed=12)
    finally:
        get_context().set_estep_schedule('auto')
    assert n_multi >= 20, n_multi


def test_config3_fast_mode_within_contract(oracle, problem64):
    """The tolerance-mode E-step at the headline size (tile-major schedule, pipelined record stream): sampled
    barcode rows against the oracle -- assignments identical, posteriors within the ulp-scaled bound of
    tests/test_gpu_fast_mode.py -- and against the exact mode on all 200k barcodes."""
    from demuxalot_amd.device import get_context
    from tests.test_gpu_fast_mode import check_contract
    p = problem64
    ctx = get_context()
    ctx.set_problem(p.n_barcodes, p.n_variants, 64, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
    ctx.set_betas(p.prior_betas())
    ctx.set_addition(None)
    prob = ctx.probs_from_betas(0.01)
    pen = np.zeros(64, dtype=np.float32)
    logits_exact, probs_exact = ctx.estep(pen, with_doublets=False)
    ctx.set_estep_mode('fast')
    try:
        logits_fast, probs_fast = ctx.estep(pen, with_doublets=False)
    finally:
        ctx.set_estep_mode('exact')
    ulps, dev = check_contract(logits_fast, probs_fast, logits_exact, probs_exact, 'fast vs exact, all barcodes', strict=False)
    for lo, hi in ((0, 1500), (120_000, 121_500)):
        v, cb, e = p.subset_barcodes(lo, hi)
        rows, v = np.unique(v, return_inverse=True)
        want = oracle.barcode_logits(v, cb, e, prob[rows], hi - lo, 0., log_impl='npsimd')
        check_contract(logits_fast[lo:hi], probs_fast[lo:hi], want, oracle.softmax_rows(want, impl='npsimd'),
                       f'fast rows [{lo},{hi})', strict=False)
    print(f'fast mode at 200k x 100k x 64: logits within {ulps:.1f} ulp of the exact mode, posteriors within {dev:.3g}')


@pytest.mark.parametrize('G', [64, 32])
def test_config3_and_config2_guarded_mode_proves_the_contract(oracle, request, G):
    """The guarded E-step at the headline size (tile-major schedule) and at configs[2]'s (two barcodes per wavefront), on
    the table of EM iteration 1 (all-distinct rows: the case every iteration after the first runs): against the exact
    mode on all 200k barcodes - every posterior within 1e-5, every argmax identical, no exceptions -, sampled rows
    against the oracle, and two guarded EM iterations against two exact ones."""
    from demuxalot_amd.device import get_context
    from tests.test_gpu_guarded import check_contract
    from demuxalot_amd import synth
    p = request.getfixturevalue('problem64') if G == 64 else synth.generate(200_000, 100_000, 32, seed=1236)
    ctx = get_context()
    ctx.set_problem(p.n_barcodes, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
    ctx.set_betas(p.prior_betas())
    pen = np.zeros(G, dtype=np.float32)
    try:
        ctx.set_estep_mode('exact')
        ctx.em(2, 0.01, pen, with_doublets=False, fetch_logits=False, fetch_probs=False, fetch_addition=False)  # P E M P E
        ctx.mstep(2., fetch=False)
        prob = ctx.probs_from_betas(0.01)
        logits_exact, probs_exact = ctx.estep(pen, with_doublets=False)
        addition_exact = ctx.mstep(2.)
        ctx.set_estep_mode('guarded')
        ctx.reset_timings()
        logits_g, probs_g = ctx.estep(pen, with_doublets=False)
        redone, _total, rows = ctx.guard_stats()
        addition_g = ctx.mstep(2.)
        # the COARSE pass (binary16 table; include/demux_hip.h: dmx_set_coarse_pass) on the same table: forced for a single E-step
        ctx.set_coarse_pass('always')
        logits_c, probs_c = ctx.estep(pen, with_doublets=False)
        levels = ctx.guard_levels()
        redone_c = ctx.guard_stats()[0]
        addition_c = ctx.mstep(2.)
    finally:
        ctx.apply_environment()
    assert rows == p.n_barcodes
    n_calls_b = 8 * ((np.bincount(p.compressed_cb, minlength=p.n_barcodes) + 7) // 8).astype(np.float64)[:, None]  # rows are padded to 8 calls
    assert levels['level'] == 0 and levels['flagged_coarse'] == redone_c and 0 <= levels['flagged_fine'] <= 2 * redone_c + 100, levels
    dev_c = check_contract(probs_c, probs_exact, f'coarse pass vs exact, all {p.n_barcodes} barcodes')
    assert redone_c <= 0.05 * p.n_barcodes, redone_c
    # its logits: within the bound the guard itself priced them with (estep_epilogue.h) - barcodes it queued are the exact kernel's
    mag = np.abs(logits_exact.astype(np.float64))
    bound = 5.1e-4 * (n_calls_b + 8) + 6.0e-8 * (0.125 * n_calls_b + 2) * (mag + 3e-4 * n_calls_b) + 3.0e-7 * (mag + 2.1e-4 * n_calls_b) + 2.4e-7 * mag
    worst = np.abs(logits_c.astype(np.float64) - logits_exact) / bound
    assert worst.max() <= 1.0, (worst.max(), np.unravel_index(worst.argmax(), worst.shape))
    n_calls_v = np.bincount(p.variant_id, minlength=p.n_variants).astype(np.float64)[:, None]
    assert (np.abs(addition_c.astype(np.float64) - addition_exact) <= n_calls_v * 2.00001e-5 + 2.0 ** -22 * addition_exact).all()
    print(f'coarse pass at 200k x 100k x {G}: posteriors within {dev_c:.3g} of the exact mode, {redone_c} barcodes redone exactly, '
          f'logits at most {worst.max():.3f} of their bound (largest deviation {np.abs(logits_c - logits_exact).max():.3g})')
    dev = check_contract(probs_g, probs_exact, f'guarded vs exact, all {p.n_barcodes} barcodes')
    assert redone <= 0.05 * p.n_barcodes, redone
    for lo, hi in ((0, 1500), (120_000, 121_500)):
        v, cb, e = p.subset_barcodes(lo, hi)
        table_rows, v = np.unique(v, return_inverse=True)
        want = oracle.barcode_logits(v, cb, e, prob[table_rows], hi - lo, 0., log_impl='npsimd')
        check_contract(probs_g[lo:hi], oracle.softmax_rows(want, impl='npsimd'), f'guarded rows [{lo},{hi}) vs oracle')
    # same table, posteriors within 1e-5: an addition entry may differ by 1e-5 (2 + 1e-5) per call of the variant (+ float32 roundings)
    n_calls = np.bincount(p.variant_id, minlength=p.n_variants).astype(np.float64)[:, None]
    assert (np.abs(addition_g.astype(np.float64) - addition_exact) <= n_calls * 2.00001e-5 + 2.0 ** -22 * addition_exact).all()
    print(f'guarded mode at 200k x 100k x {G}: posteriors within {dev:.3g} of the exact mode, {redone} of {p.n_barcodes} barcodes redone exactly')


def test_config3_twenty_iterations_in_the_default_mode_follow_the_exact_mode(problem64):
    """The headline configuration as the bench runs it: ONE call of 20 EM iterations in the default mode - E-step 0 in the
    dictionary form, 1 .. 18 on the coarse pass (binary16 table), the last one on the fine pass, the tile-major fixed-point
    M-step from the first M-step on - against the same call in the exact mode: after 20 iterations every posterior of all
    200 000 barcodes within 1e-5, every arg-max identical, the returned logits as close as the fine pass leaves them, the
    additions within what such posteriors allow."""
    from demuxalot_amd.device import DeviceContext
    from tests.test_gpu_guarded import check_contract
    p = problem64
    pen = np.zeros(64, dtype=np.float32)
    out = {}
    for mode in ('exact', 'guarded'):
        ctx = DeviceContext(0)
        try:
            ctx.set_estep_mode(mode)
            ctx.set_exact_additions(mode == 'exact')
            ctx.set_problem(p.n_barcodes, p.n_variants, 64, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
            ctx.set_betas(p.prior_betas())
            ctx.reset_timings()
            logits, probs, addition = ctx.em(20, 0.01, pen, with_doublets=False)
            out[mode] = (logits, probs, addition, ctx.guard_levels(), ctx.mstep_form())
        finally:
            ctx.close()
    levels, form = out['guarded'][3], out['guarded'][4]
    assert levels['level'] == 1 and levels['coarse_steps'] >= 17 and form in ('tiles', 'items_fixed'), (levels, form)   # (round 6: a converging call never builds the tile records)
    dev = check_contract(out['guarded'][1], out['exact'][1], 'default vs exact after 20 iterations, all 200 000 barcodes')
    d_logit = float(np.abs(out['guarded'][0] - out['exact'][0]).max())
    assert d_logit <= 5e-3, d_logit
    n_calls_v = np.bincount(p.variant_id, minlength=p.n_variants).astype(np.float64)[:, None]
    assert (np.abs(out['guarded'][2].astype(np.float64) - out['exact'][2]) <= n_calls_v * 2.00001e-5 + 2.0 ** -22 * out['exact'][2]).all()
    print(f'20 iterations at 200k x 100k x 64: {levels["coarse_steps"]} coarse E-steps; posteriors within {dev:.3g} of the exact run, logits within {d_logit:.3g}')


def test_config3_uninformative_posteriors_mstep(oracle, problem64):
    """M-step worst case: posteriors from a flat genotype table (every donor equally likely at every variant,
    the start-from-assignment scenario of tests/test_synthetic.py:200-239 before any label is used): every call
    has all 64 posteriors alive, so every call takes the dense path of the call-parallel M-step."""
    from demuxalot_amd.device import get_context
    p = problem64
    rng = np.random.default_rng(5)
    ctx = get_context()
    ctx.set_problem(p.n_barcodes, p.n_variants, 64, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
    flat = np.ones((p.n_variants, 64), dtype=np.float32)
    ctx.set_betas(flat)
    ctx.set_addition(None)
    ctx.probs_from_betas(0.01, fetch=False)
    pen = np.zeros(64, dtype=np.float32)
    ctx.estep(pen, with_doublets=False, fetch_logits=False, fetch_probs=False)
    singlets = ctx.get_block('probs', 0, p.n_barcodes, 0, 64)
    assert np.array_equal(singlets, np.full_like(singlets, 1 / 64))
    addition = ctx.mstep(2.)
    variants, _ = variants_to_check(p, rng, n_random=300)
    fio.assert_bitwise(addition[variants], addition_rows(p, singlets, variants), 'dense M-step rows')


def test_config3_full_size_properties(problem64):
    """Size-independent properties at the headline size: posteriors are distributions; a barcode shard computed
    alone gives the same rows (E-step rows are independent); the beta addition is additive over barcode shards
    (what the multi-GPU exchange relies on); results are reproducible run to run."""
    from demuxalot_amd.device import get_context
    p = problem64
    betas = p.prior_betas()
    ctx = get_context()
    pen = np.zeros(64, dtype=np.float32)
    ctx.set_problem(p.n_barcodes, p.n_variants, 64, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
    ctx.set_betas(betas)
    ctx.set_addition(None)
    ctx.probs_from_betas(0.01, fetch=False)
    logits, probs = ctx.estep(pen, with_doublets=False)
    addition = ctx.mstep(2.)
    assert np.isfinite(logits).all() and np.abs(probs.sum(axis=1) - 1).max() < 1e-5
    assert (probs >= 0).all() and (addition >= 0).all()
    truth_hit = (probs.argmax(axis=1) == p.truth[:, 0]).mean()
    assert truth_hit > 0.95, truth_hit
    # run-to-run determinism (no atomics anywhere)
    logits2, probs2 = ctx.estep(pen, with_doublets=False)
    addition2 = ctx.mstep(2.)
    fio.assert_bitwise(logits2, logits, 'E determinism')
    fio.assert_bitwise(addition2, addition, 'M determinism')
    # shard independence / additivity on two halves of the barcodes
    cut = 100_000
    total = np.zeros_like(addition, dtype=np.float64)
    for lo, hi in ((0, cut), (cut, p.n_barcodes)):
        v, cb, e = p.subset_barcodes(lo, hi)
        ctx.set_problem(hi - lo, p.n_variants, 64, v, cb, e, p.v2snp)
        ctx.set_betas(betas)
        ctx.set_addition(None)
        ctx.probs_from_betas(0.01, fetch=False)
        l_s, p_s = ctx.estep(pen, with_doublets=False)
        fio.assert_bitwise(l_s, logits[lo:hi], f'shard [{lo},{hi}) logits')
        fio.assert_bitwise(p_s, probs[lo:hi], f'shard [{lo},{hi}) probs')
        total += ctx.mstep(2.)
    assert np.allclose(total, addition, rtol=3e-7, atol=1e-6)


# ---- configs[4]: one rank's share --------------------------------------------------------------------------
def test_config4_rank_share_130k_650k_128_doublets(oracle):
    """One rank's share of 1M x 650k x 128 with doublets on 8 GPUs: 130k barcodes, 650k SNPs (V = 1.3 M, the
    666 MB genotype table lives in HBM, not in the Infinity Cache), K = 8256 options, 4.3 GB of posteriors.
    Sampled barcode rows against the oracle (8256 column passes each), M-step rows (genotype-per-lane kernel,
    G > 64) from the GPU's singlet posteriors."""
    from demuxalot_amd import synth
    p = synth.generate(130_000, 650_000, 128, doublets=True, seed=1242)
    samples = [[(0, 40), (129_950, 130_000)]]
    p_rows = [(0, 200_000), (1_200_000, 1_300_000)]
    _ctx, addition, singlets, _n_multi = staged_em_against_oracle(oracle, p, 1, 0.25, samples, seed=13, p_rows=p_rows)
    assert singlets.shape == (130_000, 128) and (addition >= 0).all()


# ---- configs[4] as written: the whole experiment on one GPU -----------------------------------------------
def test_config4_as_written_1M_650k_128_doublets_on_one_gpu(oracle):
    """BASELINE.json configs[4] at its full size on ONE MI355X: 1M barcodes x 650k SNPs (V = 1.3 M) x 128 genotypes,
    doublet_prior .25 (K = 8256), ~4e8 calls.  B x K = 8.26e9 ELEMENTS (> 2^32: 33 GB of logits, 33 GB of posteriors)
    and N > 2^28 are exactly where a 32-bit offset would hide, so the checks sit on both sides of those boundaries:
      P-step   slices of the [V, G] table from both ends, bitwise (exact mode and default mode: the P-step is the same)
      E-step   exact mode: sampled barcode rows from the first rows, the last rows and the rows whose element offset
               crosses 2^32 (row 520 223), bitwise against the oracle (8256 column passes each);
               default (guarded) mode: the same rows within the contract, the singlet posteriors of ALL 1M barcodes
               within 1e-5 of the exact mode's and the arg-max of all 1M rows (device reduction) identical
      M-step   rows of multi-item variants (the 16 384-call work items and the in-order redo), the hottest variants
               and random ones against np.bincount on the GPU's own singlet posteriors, bitwise
      results  dmx_get_block / dmx_get_assignments / dmx_get_top_options at row indices beyond the 2^32-element offset
    then two fused EM iterations (dmx_em) in the default mode."""
    import time
    import psutil
    from demuxalot_amd import Demultiplexer, synth
    from demuxalot_amd.device import get_context
    from tests.test_gpu_guarded import check_contract
    if psutil.virtual_memory().available < 56e9:
        pytest.skip(f'needs ~50 GB of host memory for the 4e8-call experiment ({psutil.virtual_memory().available / 1e9:.0f} GB free)')
    B, S, G, dp = 1_000_000, 650_000, 128, 0.25
    t0 = time.perf_counter()
    p = synth.generate_sharded(B, S, G, n_shards=16, doublets=True, seed=1242)
    t_gen = time.perf_counter() - t0
    K = G * (G + 1) // 2
    assert B * K > 2 ** 32 and p.n_calls > 2 ** 28, (B * K, p.n_calls)
    rng = np.random.default_rng(41)
    prior = p.prior_betas()
    pen = Demultiplexer._doublet_penalties(G, dp)
    cross = 2 ** 32 // K  # the row whose elements straddle the 2^32-element offset
    samples = [(0, 24), (cross - 10, cross + 14), (B - 24, B)]
    ctx = get_context()
    try:
        t0 = time.perf_counter()
        ctx.set_problem(B, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
        ctx.set_betas(prior)
        ctx.set_addition(None)
        t_install = time.perf_counter() - t0
        prob = ctx.probs_from_betas(0.01)
        for v0, v1 in ((0, 100_000), (p.n_variants - 100_000, p.n_variants)):  # whole SNP groups (variants 2s, 2s + 1)
            want = oracle.probs_from_betas(p.v2snp[v0:v1] - p.v2snp[v0], prior[v0:v1], 0.01)
            fio.assert_bitwise(prob[v0:v1], want, f'P-step rows [{v0},{v1})')
        # ---- iteration 0, exact mode: the importers' table (a handful of distinct values per row: the dictionary form) ----
        ctx.set_estep_mode('exact')
        ctx.set_exact_additions(True)
        ctx.estep(pen, with_doublets=True, fetch_logits=False, fetch_probs=False)
        form0 = ctx.estep_form()[0]
        for lo, hi in samples[1:]:
            check_e_rows(ctx, oracle, p, prob, lo, hi, dp, f'it 0 exact E rows [{lo},{hi})')
        singlets = ctx.get_block('probs', 0, B, 0, G)
        addition = ctx.mstep(2.)
        counts = np.bincount(p.variant_id, minlength=p.n_variants)
        multi = np.flatnonzero(counts > item_calls_for(p.n_calls))
        assert len(multi) >= 20, len(multi)
        variants = np.unique(np.concatenate([rng.choice(multi, size=40, replace=False), np.argsort(counts)[-8:],
                                             rng.choice(p.n_variants, size=300, replace=False)]))
        fio.assert_bitwise(addition[variants], addition_rows(p, singlets, variants), 'it 0 M-step rows (exact mode)')
        assert (addition >= 0).all() and np.isfinite(addition).all()
        # ---- iteration 1, exact mode: the table of prior + addition (all-distinct rows: the one-log-per-term kernels) ----
        prob = ctx.probs_from_betas(0.01)
        for v0, v1 in ((0, 100_000), (p.n_variants - 100_000, p.n_variants)):
            want = oracle.probs_from_betas(p.v2snp[v0:v1] - p.v2snp[v0], (prior + addition)[v0:v1], 0.01)
            fio.assert_bitwise(prob[v0:v1], want, f'it 1 P-step rows [{v0},{v1})')
        t0 = time.perf_counter()
        ctx.estep(pen, with_doublets=True, fetch_logits=False, fetch_probs=False)
        ctx.synchronize()
        t_exact = time.perf_counter() - t0
        form1 = ctx.estep_form()[0]
        assert form1 == 'direct', form1
        device_bytes = ctx.device_bytes()
        exact_rows = {}
        for lo, hi in samples:
            check_e_rows(ctx, oracle, p, prob, lo, hi, dp, f'it 1 exact E rows [{lo},{hi})')
            exact_rows[lo] = ctx.get_block('probs', lo, hi)
        singlets = ctx.get_block('probs', 0, B, 0, G)
        best_exact, best_p_exact = ctx.get_assignments()
        for lo, hi in samples:  # device reductions at rows beyond the 2^32-element offset against numpy on the fetched block
            block = exact_rows[lo]
            assert np.array_equal(best_exact[lo:hi], block.argmax(axis=1)) and np.array_equal(best_p_exact[lo:hi], block.max(axis=1))
        top, top_p = ctx.get_top_options(2)
        block = exact_rows[samples[-1][0]]
        order = np.argsort(-block, axis=1, kind='stable')[:, :2]
        assert np.array_equal(top[B - 24:], order) and np.array_equal(top_p[B - 24:], np.take_along_axis(block, order, axis=1))
        assert abs(float(ctx.get_option_sums().sum()) - B) < 1e-3 * B
        addition1 = ctx.mstep(2.)
        fio.assert_bitwise(addition1[variants], addition_rows(p, singlets, variants), 'it 1 M-step rows (exact mode)')
        # ---- the library's default mode on the same table ----
        ctx.set_estep_mode('guarded')
        ctx.set_exact_additions(False)
        ctx.reset_timings()
        t0 = time.perf_counter()
        ctx.estep(pen, with_doublets=True, fetch_logits=False, fetch_probs=False)
        ctx.synchronize()
        t_guarded = time.perf_counter() - t0
        redone, _total, rows = ctx.guard_stats()
        assert rows == B and ctx.estep_form()[0] == 'direct'
        for lo, hi in samples:
            check_contract(ctx.get_block('probs', lo, hi), exact_rows[lo], f'guarded E rows [{lo},{hi})')
        singlets_g = ctx.get_block('probs', 0, B, 0, G)
        assert np.abs(singlets_g.astype(np.float64) - singlets).max() <= 1e-5
        best_g, _ = ctx.get_assignments()
        assert np.array_equal(best_g, best_exact), 'default mode: assignments of all 1M barcodes'
        singlet = p.truth[:, 0] == p.truth[:, 1]
        hit = (best_g[singlet] == p.truth[singlet, 0]).mean()
        assert hit > 0.9, hit
        addition_g = ctx.mstep(2.)
        assert np.abs(addition_g[variants].astype(np.float64) - addition1[variants]).max() <= 2e-5 * counts[variants].max()
        # ---- two fused EM iterations, default mode ----
        t0 = time.perf_counter()
        ctx.em(2, 0.01, pen, with_doublets=True, fetch_logits=False, fetch_probs=False, fetch_addition=False)
        t_em = time.perf_counter() - t0
        best_em, best_p_em = ctx.get_assignments()
        assert np.isfinite(best_p_em).all() and (best_p_em > 0).all() and (best_em == best_g).mean() > 0.98
    finally:
        ctx.apply_environment()
        ctx.release_problem()
        ctx.trim_cache()
    print(f'configs[4] as written on one GPU: {p.n_calls} calls generated in {t_gen:.0f} s, installed in {t_install:.1f} s, '
          f'{device_bytes / 1e9:.1f} GB on the device ({device_bytes / p.n_calls:.0f} B per call incl. the [B, K] results); iteration 0 took the {form0} form; '
          f'E-step of iteration 1: exact {t_exact:.2f} s, default {t_guarded:.2f} s ({redone} barcodes redone exactly); 2 fused EM iterations {t_em:.2f} s')


# ---- the tile-major schedule with two accumulators per lane (65..128 genotypes, singlets) ------------------
def test_tiled_schedule_two_slots_per_lane(oracle):
    """70k barcodes x 20k SNPs x 128 genotypes (K = 128 singlets: two options per lane; 20 MB genotype table, so the
    repack builds the tile-major schedule): the exact mode under the forced tile-major schedule must equal the direct
    schedule bit for bit and the oracle on sampled rows; the tolerance mode (which uses the schedule by default) must
    stay within its contract against the exact mode."""
    from demuxalot_amd import synth
    from demuxalot_amd.device import get_context
    from tests.test_gpu_fast_mode import check_contract
    p = synth.generate(70_000, 20_000, 128, calls_per_barcode=100, seed=31)
    ctx = get_context()
    ctx.set_problem(p.n_barcodes, p.n_variants, 128, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
    ctx.set_betas(p.prior_betas())
    ctx.set_addition(None)
    prob = ctx.probs_from_betas(0.01)
    pen = np.zeros(128, dtype=np.float32)
    try:
        ctx.set_estep_schedule('direct')
        logits_d, probs_d = ctx.estep(pen, with_doublets=False)
        ctx.set_estep_schedule('tiled')
        logits_t, probs_t = ctx.estep(pen, with_doublets=False)
        fio.assert_bitwise(logits_t, logits_d, 'tile-major vs direct schedule: logits')
        fio.assert_bitwise(probs_t, probs_d, 'tile-major vs direct schedule: posteriors')
        check_e_rows(ctx, oracle, p, prob, 1000, 1400, 0., 'tile-major rows [1000,1400)')
        addition_tiled = ctx.mstep(2.)
        ctx.set_estep_schedule('auto')
        ctx.set_estep_mode('fast')
        logits_f, probs_f = ctx.estep(pen, with_doublets=False)
    finally:
        ctx.set_estep_mode('exact')
        ctx.set_estep_schedule('auto')
    check_contract(logits_f, probs_f, logits_d, probs_d, 'fast (tile-major, two slots) vs exact', strict=False)
    rng = np.random.default_rng(3)
    variants, _ = variants_to_check(p, rng, n_random=300)
    fio.assert_bitwise(addition_tiled[variants], addition_rows(p, probs_t, variants), 'M-step rows (G = 128)')


def test_posterior_table_beyond_4_gib(oracle):
    """132k barcodes x 128 genotypes with doublets: 4.36 GB of posteriors, i.e. byte offsets into the [B, K] tables
    that no longer fit 32 bits (the configs[4] share above stays 2 MB below 4 GiB): the wide-row E-step, its softmax
    and the genotype-per-lane M-step with 64-bit addressing, rows from both ends of the table against the oracle."""
    from demuxalot_amd import synth
    p = synth.generate(132_000, 6_000, 128, calls_per_barcode=120, doublets=True, seed=1250)
    assert p.n_barcodes * 8256 * 4 > 2 ** 32
    samples = [[(0, 30), (131_960, 132_000)]]
    _ctx, addition, singlets, _n_multi = staged_em_against_oracle(oracle, p, 1, 0.25, samples, seed=17, p_rows=[(0, 12_000)])
    assert singlets.shape == (132_000, 128) and (addition >= 0).all()
