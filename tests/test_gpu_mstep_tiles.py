"""Tile-major M-step (kernels.hip: k_mstep_tiles; taken when the exact additions are off and G <= 64) against the
work-item form and against the exact additions.  Its sums are 64-bit fixed-point (every contribution, a float32 in [0, 1], is
added as the integer rint(c 2^s), s = 50 for all but the hottest tiles): order-independent, hence bit-reproducible run to run
like the reference's np.bincount, exact for every contribution of 2^-27 and more, and within n 2^-(s + 1) of the real sum in
general - against the reference's float64 sum: one float32 ulp + n 2^-51."""
import numpy as np
import pytest

from tests.thread_plane import ThreadWorld

pytestmark = pytest.mark.gpu


def assert_within_tile_bound(got, want, p, what=''):
    """|got - want| <= one float32 ulp of want + n(v) 2^-50 per entry (n(v) = calls of the variant; twice the grid's bound)."""
    counts = np.bincount(p.variant_id, minlength=p.n_variants).astype(np.float64)[:, None]
    dev = np.abs(got.astype(np.float64) - want)
    bound = 2.0 ** -23 * np.abs(want.astype(np.float64)) + counts * 2.0 ** -50
    assert np.isfinite(got).all() and (dev <= bound).all(), (what, float((dev - bound).max()), float(dev.max()))


def _additions(ctx, pen, doublets, power, n_iterations=2):
    out = []
    for _ in range(n_iterations):
        ctx.probs_from_betas(0.01, fetch=False)
        ctx.estep(pen, with_doublets=doublets, fetch_logits=False, fetch_probs=False)
        out.append(ctx.mstep(power))
    return out


def _context(p, G, exact, tiles):
    from demuxalot_amd.device import DeviceContext
    ctx = DeviceContext(0)
    ctx.set_estep_mode('exact')  # same posteriors for every M-step form
    ctx.set_exact_additions(exact)
    ctx.set_mstep_tiles(tiles)
    ctx.set_problem(p.n_barcodes, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
    ctx.set_betas(p.prior_betas())
    ctx.set_addition(None)
    return ctx


@pytest.mark.parametrize('power', [2.0, 1.5])
@pytest.mark.parametrize('G,doublets,B,S,cpb', [(2, False, 3000, 500, 30), (5, True, 2000, 800, 60), (8, True, 20000, 3000, 120),
                                               (16, False, 5000, 40, 200), (33, False, 4000, 3000, 80), (64, False, 30000, 6000, 150),
                                               (64, True, 300, 200, 50)])
def test_tile_major_mstep_against_the_other_forms(G, doublets, B, S, cpb, power):
    from demuxalot_amd import Demultiplexer, synth
    p = synth.generate(B, S, G, calls_per_barcode=cpb, doublets=doublets, seed=900 + G + B)
    pen = Demultiplexer._doublet_penalties(G, 0.2 if doublets else 0.0)
    results = {}
    for name, exact, tiles in (('exact', True, True), ('items', False, False), ('tiles', False, True)):
        ctx = _context(p, G, exact, tiles)
        try:
            results[name] = _additions(ctx, pen, doublets, power)
            assert ctx.mstep_form() == ('tiles' if name == 'tiles' else 'items')
        finally:
            ctx.close()
    for it in range(2):
        want = results['exact'][it]
        assert np.isfinite(results['tiles'][it]).all()
        assert np.allclose(results['items'][it], want, rtol=3e-7, atol=0)
        assert_within_tile_bound(results['tiles'][it], want, p, f'G={G} it {it}')
        # exact integer sums, one rounding: all but a handful of the entries that are not made of dead posteriors alone are the
        # reference's bits
        big = want >= 1e-3
        assert (results['tiles'][it][big] != want[big]).mean() < 2e-3


def test_flat_genotypes_take_the_dense_kernel():
    """All-equal betas: every posterior is 1 / G, every call has 64 live posteriors; the dense regime's kernel takes the launch
    (decided on the device), the tile kernel and - in its turn - the combining pass stand back."""
    from demuxalot_amd import synth
    G = 64
    p = synth.generate(6000, 1500, G, calls_per_barcode=100, seed=4242)
    pen = np.zeros(G, dtype=np.float32)
    out = {}
    for name, exact, tiles in (('exact', True, True), ('tiles', False, True)):
        ctx = _context(p, G, exact, tiles)
        try:
            ctx.set_betas(np.ones_like(p.prior_betas()))
            out[name] = _additions(ctx, pen, False, 2.0, n_iterations=3)
        finally:
            ctx.close()
    for got, want in zip(out['tiles'], out['exact']):
        assert np.allclose(got, want, rtol=3e-7, atol=0)  # (the dense regime's kernel: float64 sums over work items)
    # and back to informative posteriors on the same context: the tile kernel again
    ctx = _context(p, G, False, True)
    try:
        ctx.set_betas(np.ones_like(p.prior_betas()))
        flat = _additions(ctx, pen, False, 2.0, n_iterations=1)[0]
        ctx.set_betas(p.prior_betas())
        ctx.set_addition(None)
        sharp = _additions(ctx, pen, False, 2.0, n_iterations=1)[0]
    finally:
        ctx.close()
    ref = _context(p, G, True, True)
    try:
        want = _additions(ref, pen, False, 2.0, n_iterations=1)[0]
    finally:
        ref.close()
    assert np.allclose(flat, out['exact'][0], rtol=3e-7, atol=0)
    assert_within_tile_bound(sharp, want, p, 'sharp posteriors after flat ones')


@pytest.mark.parametrize('exchange', ['variant', 'reduce_scatter', 'allreduce'])
def test_tile_major_mstep_on_three_ranks(exchange, monkeypatch):
    """The default mode (guarded E-step, additions in any order) sharded over three ranks, in every exchange: the learnt
    posteriors and additions against one context in the same mode."""
    monkeypatch.setenv('DEMUXALOT_AMD_EXCHANGE', exchange)
    from demuxalot_amd import distributed, synth
    from demuxalot_amd.device import DeviceContext
    G = 24
    p = synth.generate(5000, 2500, G, calls_per_barcode=90, seed=515)
    betas = p.prior_betas()
    pen = np.zeros(G, dtype=np.float32)
    with DeviceContext(0) as ctx:
        ctx.set_estep_mode('exact')
        ctx.set_exact_additions(True)
        ctx.set_problem(p.n_barcodes, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
        ctx.set_betas(betas)
        _l, want_probs, want_add = ctx.em(4, 0.01, pen, False, fetch_logits=False)
    shared = ThreadWorld(3)

    def rank_body(plane):
        em = distributed.ShardedEM(plane, p.n_barcodes, p.v2snp, betas, p.variant_id, p.compressed_cb, p.p_base_wrong, reduce_dtype='f64')
        try:
            em.ctx.set_estep_mode('exact')
            em.ctx.set_exact_additions(False)
            em.ctx.set_mstep_tiles(True)
            probs, addition = em.learn(4, 0.01, pen, False)
            return em.lo, em.hi, probs, addition
        finally:
            em.ctx.close()

    for lo, hi, probs, addition in shared.run(rank_body):
        if exchange == 'variant':  # every sum formed whole on one rank
            assert_within_tile_bound(addition, want_add, p, exchange)
        else:  # per-rank sums added across ranks in float64
            assert np.allclose(addition, want_add, rtol=3e-7, atol=1e-12 * np.bincount(p.variant_id, minlength=p.n_variants).max())
        assert np.array_equal(probs.argmax(1), want_probs[lo:hi].argmax(1)) and np.allclose(probs, want_probs[lo:hi], rtol=0, atol=1e-5)


def test_reinstalling_a_problem_rebuilds_the_tiles():
    from demuxalot_amd import synth
    from demuxalot_amd.device import DeviceContext
    ctx = DeviceContext(0)
    ctx.set_estep_mode('exact')
    ctx.set_exact_additions(False)
    try:
        for seed, (B, S, G) in enumerate([(3000, 700, 12), (800, 2000, 40), (3000, 700, 12)]):
            p = synth.generate(B, S, G, calls_per_barcode=50, seed=70 + seed)
            ctx.set_problem(p.n_barcodes, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
            ctx.set_betas(p.prior_betas())
            ctx.set_addition(None)
            got = _additions(ctx, np.zeros(G, dtype=np.float32), False, 2.0)
            ref = _context(p, G, True, True)
            try:
                want = _additions(ref, np.zeros(G, dtype=np.float32), False, 2.0)
            finally:
                ref.close()
            for x, y in zip(got, want):
                assert_within_tile_bound(x, y, p, f'problem {seed}')
    finally:
        ctx.close()


@pytest.mark.parametrize('G,B,S,cpb', [(64, 30000, 6000, 150), (24, 20000, 300, 200)])
def test_tile_major_mstep_is_bit_reproducible(G, B, S, cpb):
    """The reference's np.bincount (utils.py:35-36) is a fixed sequential sum: run twice it gives the same bits.  So does the
    tile-major form: its fixed-point sums do not depend on the order in which the wavefronts' LDS atomics arrive - the same
    additions from run to run on one context, on a second context, and with the tile records built at another moment."""
    from demuxalot_amd import synth
    p = synth.generate(B, S, G, calls_per_barcode=cpb, seed=1000 + G)
    pen = np.zeros(G, dtype=np.float32)
    runs = []
    for _ in range(2):
        ctx = _context(p, G, False, True)
        try:
            for _rep in range(3):
                ctx.set_addition(None)
                runs.append(_additions(ctx, pen, False, 2.0, n_iterations=3))
                assert ctx.mstep_form() == 'tiles'
        finally:
            ctx.close()
    for other in runs[1:]:
        for x, y in zip(runs[0], other):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32))


@pytest.mark.parametrize('power', [2.0, 1.5])
@pytest.mark.parametrize('G,doublets,B,S,cpb,hard', [(8, True, 6000, 3000, 400, False), (33, False, 6000, 3000, 500, False),
                                                    (64, False, 12000, 6000, 400, False), (64, False, 12000, 4000, 400, True),
                                                    (24, True, 5000, 2000, 60, True)])
def test_incremental_mstep_is_the_full_pass_bit_for_bit(G, doublets, B, S, cpb, hard, power):
    """The incremental M-step (include/demux_hip.h: dmx_set_mstep_incremental; kernels.h: MIncrArgs): after one full pass of the
    tile-major kernel the integer sums stay on the device, and an M-step only adds, for the barcodes whose posteriors changed where
    it matters, the differences of their new and old integer contributions.  Eight EM iterations with it and without it (every M-step
    the full pass): the additions equal BIT FOR BIT after every one of them, and the delta pass did run (sibling donors / few calls per barcode: the
    device falls back to the full pass by itself wherever the changed barcodes hold an eighth of the calls or the posteriors are dense)."""
    from demuxalot_amd import Demultiplexer, synth
    p = synth.generate(B, S, G, calls_per_barcode=cpb, doublets=doublets, seed=1700 + G + B, sibling_pairs=hard)
    pen = Demultiplexer._doublet_penalties(G, 0.2 if doublets else 0.0)
    runs = {}
    for incremental in (True, False):
        ctx = _context(p, G, False, True)
        try:
            ctx.set_mstep_incremental(incremental)
            ctx.reset_timings()
            runs[incremental] = (_additions(ctx, pen, doublets, power, n_iterations=8), ctx.mstep_incremental())
            assert ctx.mstep_form() == 'tiles'
        finally:
            ctx.close()
    for it, (got, want) in enumerate(zip(runs[True][0], runs[False][0])):
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (it, int((got != want).sum()), float(np.abs(got - want).max()))
    full, delta, last = runs[True][1]
    assert full + delta == 8 and full >= 1 and runs[False][1] == (0, 0, 0), (runs[True][1], runs[False][1])
    if not hard:
        assert delta >= 5 and 0 <= last < 0.2 * B, (full, delta, last)   # separable donors with 400 calls per barcode: the later iterations change little
    print(f'G={G} doublets={doublets} hard={hard} power={power}: {full} full + {delta} delta passes, the last delta pass visited {last} of {B} barcodes')


def test_incremental_mstep_starts_over_when_the_addition_is_replaced():
    """dmx_set_addition, a dmx_em call (which zeroes the addition first) and a switch of the M-step form make the kept sums useless:
    the next M-step is a full pass, and the results stay those of a context that never kept anything."""
    from demuxalot_amd import synth
    p = synth.generate(8000, 2000, 16, calls_per_barcode=400, seed=1790)
    pen = np.zeros(16, dtype=np.float32)
    out = {}
    for incremental in (True, False):
        ctx = _context(p, 16, False, True)
        try:
            ctx.set_mstep_incremental(incremental)
            ctx.reset_timings()
            a = _additions(ctx, pen, False, 2.0, n_iterations=3)
            ctx.set_addition(a[0] * np.float32(0.5))                       # somebody else's addition
            b = _additions(ctx, pen, False, 2.0, n_iterations=2)
            _l, _p, c_add = ctx.em(4, 0.01, pen, with_doublets=False, fetch_logits=False, fetch_probs=False)
            ctx.set_mstep_tiles(False)                                     # the work-item form in between
            d = _additions(ctx, pen, False, 2.0, n_iterations=1)
            ctx.set_mstep_tiles(True)
            e = _additions(ctx, pen, False, 2.0, n_iterations=2)
            out[incremental] = (a + b + [c_add] + e, ctx.mstep_incremental())
        finally:
            ctx.close()
    for i, (got, want) in enumerate(zip(out[True][0], out[False][0])):
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), i
    full, delta, _last = out[True][1]
    assert full >= 4 and delta >= 4, (full, delta)   # a full pass after each of the four breaks, delta passes behind them


def test_the_delta_pass_builds_the_full_pass_sums_from_nothing():
    """dmx_set_mstep_incremental(ctx, 2): the first M-step's sums are built by the DELTA pass - every barcode against an all-zero
    posterior row, every contribution a difference from 0 - instead of the tile-major full pass.  Two different kernels, two different
    walks of the calls (barcode-major with device-scope atomics / tile-major in LDS): the same integers, hence the same bits."""
    from demuxalot_amd import synth
    p = synth.generate(9000, 3000, 32, calls_per_barcode=400, seed=1811)
    pen = np.zeros(32, dtype=np.float32)
    out = {}
    for mode in (True, 'bootstrap'):
        ctx = _context(p, 32, False, True)
        try:
            ctx.set_mstep_incremental(mode)
            ctx.reset_timings()
            out[mode] = (_additions(ctx, pen, False, 2.0, n_iterations=3), ctx.mstep_incremental())
        finally:
            ctx.close()
    for it in range(3):
        assert np.array_equal(out[True][0][it].view(np.uint32), out['bootstrap'][0][it].view(np.uint32)), it
    assert out[True][1][0] == 1 and out['bootstrap'][1][:2] == (0, 3), (out[True][1], out['bootstrap'][1])


@pytest.mark.parametrize('power', [2.0, 1.5])
@pytest.mark.parametrize('G,doublets,B,S,cpb,hard', [(8, True, 6000, 3000, 400, False), (33, False, 6000, 3000, 500, False),
                                                    (64, False, 30000, 6000, 150, False), (64, False, 12000, 4000, 400, True),
                                                    (2, False, 3000, 500, 30, False), (16, False, 5000, 40, 200, False)])
def test_fixed_point_work_items_are_the_tile_form_bit_for_bit(G, doublets, B, S, cpb, hard, power):
    """A call too short to pay for the tile-major records (learn_genotypes' default of 5 iterations) runs the WORK-ITEM kernel with the
    tile-major form's arithmetic - every contribution added as the integer rint(c 2^shift), shift from the same tile cut - and the
    incremental M-step on top of its sums (kernels.h: MstepArgs::fixed_shift_v).  Integer sums do not depend on who adds in which
    order: the additions equal the tile-major full pass's BIT FOR BIT in every iteration, no tile-major records are ever built, the
    delta pass runs, and two runs give the same bits."""
    from demuxalot_amd import Demultiplexer, synth
    p = synth.generate(B, S, G, calls_per_barcode=cpb, doublets=doublets, seed=2100 + G + B, sibling_pairs=hard)
    pen = Demultiplexer._doublet_penalties(G, 0.2 if doublets else 0.0)
    n_it = 5
    want_ctx = _context(p, G, False, True)          # tile-major records at the first M-step, every M-step the full pass
    try:
        want_ctx.set_mstep_incremental(False)
        want = _additions(want_ctx, pen, doublets, power, n_iterations=n_it)
        assert want_ctx.mstep_form() == 'tiles'
    finally:
        want_ctx.close()
    runs = []
    for _ in range(2):
        ctx = _context(p, G, False, 'auto')         # the library's defaults: records only when 8 M-steps are to come
        try:
            ctx.set_mstep_incremental(True)
            ctx.reset_timings()
            got = _additions(ctx, pen, doublets, power, n_iterations=n_it)
            assert ctx.mstep_form() == 'items_fixed' and ctx.mstep_tiles_info()[0] is False
            full, delta, last = ctx.mstep_incremental()
            runs.append(got)
        finally:
            ctx.close()
        for it in range(n_it):
            assert np.array_equal(got[it].view(np.uint32), want[it].view(np.uint32)), (it, int((got[it] != want[it]).sum()), float(np.abs(got[it] - want[it]).max()))
        assert full + delta == n_it and full >= 1, (full, delta, last)
        if not hard and cpb >= 400:   # (few calls per barcode: dense posteriors in the first iterations - the dense regime's kernel, no kept sums)
            assert delta >= 2, (full, delta, last)
    for x, y in zip(*runs):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    print(f'G={G} doublets={doublets} hard={hard} power={power}: {full} full + {delta} delta passes on the work items\' sums')


def test_fixed_point_work_items_in_the_dense_regime_and_through_the_switches():
    """All-equal betas (every posterior 1 / G): the dense regime's kernel takes the launches - float64 sums, as in every other form -
    and the combining pass must read its partial sums as float64, not as the fixed-point form's integers.  dmx_set_mstep_tiles(0) and
    dmx_set_mstep_incremental(0) both mean the float64 work-item form of before."""
    from demuxalot_amd import synth
    G = 64
    p = synth.generate(6000, 1500, G, calls_per_barcode=100, seed=4243)
    pen = np.zeros(G, dtype=np.float32)
    flat = np.ones_like(p.prior_betas())
    out = {}
    for name, tiles, incremental in (('items', False, True), ('auto', 'auto', True), ('auto_full', 'auto', False)):
        ctx = _context(p, G, False, tiles)
        try:
            ctx.set_mstep_incremental(incremental)
            ctx.set_betas(flat)
            ctx.set_addition(None)
            out[name] = (_additions(ctx, pen, False, 2.0, n_iterations=3), ctx.mstep_form())
        finally:
            ctx.close()
    assert out['items'][1] == 'items' and out['auto_full'][1] == 'items' and out['auto'][1] == 'items_fixed'
    for it in range(3):
        assert np.array_equal(out['auto'][0][it].view(np.uint32), out['items'][0][it].view(np.uint32)), it
        assert np.array_equal(out['auto_full'][0][it].view(np.uint32), out['items'][0][it].view(np.uint32)), it
        assert np.isfinite(out['auto'][0][it]).all() and out['auto'][0][it].max() > 0


def test_tile_records_are_built_only_where_full_passes_keep_coming():
    """Under the incremental M-step only full passes cost anything, and the work items make them without the tile-major records (0.71
    against 0.33 ms at 200k x 100k x 64, no 2.6 ms sort): a context that can go incremental starts on the work items and looks at the
    device's count of (sparse-regime) full passes at its 4th, 16th and 64th M-step.  Separable donors - the delta pass takes over -: a
    12-iteration call never builds the records.  A run whose every M-step is a full pass (here: the addition is replaced behind each,
    which makes the kept sums useless) with 8 or more M-steps still announced at the 5th: the records are built there.  Either way the additions
    are those of a context that had the records from the start, bit for bit."""
    from demuxalot_amd import synth
    G = 32
    pen = np.zeros(G, dtype=np.float32)
    p = synth.generate(12000, 4000, G, calls_per_barcode=400, seed=2700)
    out = {}
    for tiles in ('auto', True):
        ctx = _context(p, G, False, tiles)
        try:
            ctx.set_mstep_incremental(True)
            ctx.reset_timings()
            _l, _p, addition = ctx.em(12, 0.01, pen, with_doublets=False, fetch_logits=False, fetch_probs=False)
            converging = (addition, ctx.mstep_form(), ctx.mstep_tiles_info()[0], ctx.mstep_incremental())
        finally:
            ctx.close()
        ctx = _context(p, G, False, tiles)   # the same problem on a fresh context, every M-step a full pass
        try:
            ctx.set_mstep_incremental(True)
            ctx.set_msteps_expected(16)   # (8 or more still to come when the device's count is read at the 5th)
            forms, additions = [], []
            for _ in range(10):
                ctx.probs_from_betas(0.01, fetch=False)
                ctx.estep(pen, with_doublets=False, fetch_logits=False, fetch_probs=False)
                additions.append(ctx.mstep(2.0))
                forms.append((ctx.mstep_form(), ctx.mstep_tiles_info()[0]))
                ctx.set_addition(additions[-1])   # (somebody else's addition, as far as the kept sums are concerned)
            out[tiles] = (converging, forms, additions)
        finally:
            ctx.close()
    (add_auto, form_auto, built_auto, (full, delta, _last)), forms_auto, additions_auto = out['auto']
    assert np.array_equal(add_auto.view(np.uint32), out[True][0][0].view(np.uint32))
    assert built_auto is False and form_auto == 'items_fixed' and full == 1 and delta == 10, out['auto'][0][1:]
    for it, (x, y) in enumerate(zip(additions_auto, out[True][2])):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), it
    # the first four M-steps run on the work items (full passes each); the fifth looks at the device's count and builds the records
    assert [f for f, _b in forms_auto[:4]] == ['items_fixed'] * 4 and forms_auto[3][1] is False, forms_auto
    assert forms_auto[-1] == ('tiles', True) and ('tiles', True) in forms_auto[4:6], forms_auto
    print('forms of the ten full-pass M-steps:', forms_auto)
