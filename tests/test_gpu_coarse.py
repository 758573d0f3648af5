"""The coarse pass of the default (guarded) mode - k_estep_tiled_coarse, include/demux_hip.h: dmx_set_coarse_pass - inside the calls
that take it: dmx_em / dmx_run_iterations on problems large enough for the tile-major schedule (>= 8 192 barcodes, a genotype
table of >= 1 MB, 33 .. 64 genotypes, no doublets).  Same-table checks at the headline size: tests/test_gpu_configs.py."""
import numpy as np
import pytest

from tests.test_gpu_guarded import check_contract

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def separable():
    from demuxalot_amd import synth
    return synth.generate(70_000, 40_000, 64, calls_per_barcode=200, seed=4100)


@pytest.fixture(scope='module')
def siblings():
    from demuxalot_amd import synth
    return synth.generate(70_000, 40_000, 64, calls_per_barcode=50, seed=4101, sibling_pairs=True)


def _install(ctx, p):
    ctx.set_problem(p.n_barcodes, p.n_variants, p.n_genotypes, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
    ctx.set_betas(p.prior_betas())


def test_em_call_takes_the_coarse_pass_for_all_but_its_last_estep(separable):
    """dmx_em with 6 iterations: E-steps 0 .. 4 may take the coarse pass - on separable donors they do (E-step 0 took the exact
    dictionary form until round 6: faster than the fine pass, slower than the coarse one) -, the last one, whose logits the call
    returns, the fine pass.  Against the exact mode's call: posteriors of every barcode
    within 1e-5 with the same arg-max, additions within what such posteriors allow, returned logits as close as the fine pass
    leaves them; and the guarded mode of round 4 (coarse pass off) for comparison."""
    from demuxalot_amd.device import DeviceContext
    p = separable
    pen = np.zeros(p.n_genotypes, dtype=np.float32)
    out = {}
    for name, mode, coarse in (('exact', 'exact', True), ('default', 'guarded', True), ('fine only', 'guarded', False)):
        ctx = DeviceContext(0)
        try:
            ctx.set_estep_mode(mode)
            ctx.set_exact_additions(mode == 'exact')
            ctx.set_coarse_pass(coarse)
            _install(ctx, p)
            ctx.reset_timings()
            logits, probs, addition = ctx.em(6, 0.01, pen, with_doublets=False)
            out[name] = (logits, probs, addition, ctx.guard_levels(), ctx.guard_stats())
        finally:
            ctx.close()
    lv = out['default'][3]
    assert lv['level'] == 1, lv                      # the last E-step: the fine pass
    assert lv['coarse_steps'] == 5, lv               # E-steps 0 .. 4
    assert lv['coarse_pass_ms'] > 0, lv              # (the last E-step's own time is folded in when the next one begins)
    assert out['fine only'][3]['coarse_steps'] == 0 and out['fine only'][3]['flagged_coarse'] == -1
    n_calls_v = np.bincount(p.variant_id, minlength=p.n_variants).astype(np.float64)[:, None]
    for name in ('default', 'fine only'):
        dev = check_contract(out[name][1], out['exact'][1], f'{name} vs exact after 6 iterations')
        d_add = np.abs(out[name][2].astype(np.float64) - out['exact'][2])
        assert (d_add <= n_calls_v * 2.00001e-5 + 2.0 ** -22 * out['exact'][2]).all()
        d_logit = np.abs(out[name][0] - out['exact'][0]).max()
        assert d_logit <= 2e-3, (name, d_logit)   # the fine pass's arithmetic + what 1e-5 posteriors do to the tables over 5 M-steps
        print(f'{name}: posteriors within {dev:.3g}, logits within {d_logit:.3g} of the exact call; levels {out[name][3]}; guard {out[name][4]}')


def test_lean_memory_releases_the_fine_pass_records(separable):
    """dmx_set_lean_memory(1): the tile-major copy of the E-step records - 16 bytes per (padded) call, what the fine pass reads - and the
    dictionary form's row array (4) are released once the coarse pass's records are built.  The coarse E-steps are untouched; the E-steps that keep their logits (the last one of
    the call, a dmx_estep behind it) run the tolerance kernel on the barcode-major records: same contract against the exact mode, and the
    context holds that much less."""
    from demuxalot_amd.device import DeviceContext
    p = separable
    pen = np.zeros(p.n_genotypes, dtype=np.float32)
    out = {}
    for name, mode, lean in (('exact', 'exact', False), ('default', 'guarded', False), ('lean', 'guarded', True)):
        ctx = DeviceContext(0)
        try:
            ctx.set_estep_mode(mode)
            ctx.set_exact_additions(mode == 'exact')
            ctx.set_lean_memory(lean)
            _install(ctx, p)
            installed = ctx.device_bytes()
            ctx.reset_timings()
            logits, probs, addition = ctx.em(6, 0.01, pen, with_doublets=False)
            levels = ctx.guard_levels()
            logits_1, probs_1 = ctx.estep(pen, with_doublets=False)   # (the resident problem again: an E-step that keeps its logits)
            out[name] = (logits, probs, addition, levels, installed, ctx.device_bytes(), logits_1, probs_1, ctx.guard_levels())
        finally:
            ctx.close()
    assert out['default'][4] == out['lean'][4]  # (the installs are the same size: the release comes with the first coarse E-step)
    saved = out['default'][5] - out['lean'][5]   # the tile-major stream (16 bytes per padded call) + the dictionary form's row array (4)
    assert 20 * len(p.variant_id) <= saved <= 20 * (len(p.variant_id) + 8 * p.n_barcodes) + 8192, (saved, len(p.variant_id))
    # the call's M-steps follow coarse E-steps in both runs - the incremental M-step's delta passes read the table rows from the records
    # where the row array is gone: the same additions, bit for bit
    from tests import fixture_io as fio
    fio.assert_bitwise(out['lean'][2], out['default'][2], 'additions of the lean run')
    lv = out['lean'][3]
    assert lv['coarse_steps'] == 5 and lv['level'] in (1, 2), lv      # E-steps 0 .. 4 coarse as ever; the last one: the fine level or direct
    assert out['lean'][8]['level'] in (1, 2), out['lean'][8]
    n_calls_v = np.bincount(p.variant_id, minlength=p.n_variants).astype(np.float64)[:, None]
    for name in ('default', 'lean'):
        dev = check_contract(out[name][1], out['exact'][1], f'{name} vs exact after 6 iterations')
        dev_1 = check_contract(out[name][7], out['exact'][7], f'{name} vs exact, the E-step behind the call')
        d_add = np.abs(out[name][2].astype(np.float64) - out['exact'][2])
        assert (d_add <= n_calls_v * 2.00001e-5 + 2.0 ** -22 * out['exact'][2]).all()
        d_logit = max(np.abs(out[name][0] - out['exact'][0]).max(), np.abs(out[name][6] - out['exact'][6]).max())
        assert d_logit <= 2e-3, (name, d_logit)
        print(f'{name}: posteriors within {dev:.3g} / {dev_1:.3g}, logits within {d_logit:.3g}; device bytes {out[name][4]} -> {out[name][5]}; levels {out[name][3]}')
    print(f'released: {saved} bytes = {saved / len(p.variant_id):.2f} per call')


def test_coarse_pass_gives_way_where_it_proves_too_little(siblings):
    """Sibling donors, 50 calls per barcode: the coarse guard (D ~ 0.03) flags most barcodes, the fine one a fifth.  The first
    admissible E-step takes the coarse pass and finds that out; the device then prices  C + f_coarse E  against  F + f_fine E  and E
    on its own timings and leaves the coarse pass.  Results stay within the contract of the exact mode's whichever level ran."""
    from demuxalot_amd.device import DeviceContext
    p = siblings
    pen = np.zeros(p.n_genotypes, dtype=np.float32)
    ref = DeviceContext(0)
    try:
        ref.set_estep_mode('exact')
        _install(ref, p)
        _l, probs_exact, _a = ref.em(2, 0.01, pen, with_doublets=False, fetch_logits=False)
    finally:
        ref.close()
    ctx = DeviceContext(0)
    try:
        ctx.set_estep_mode('guarded')
        _install(ctx, p)
        ctx.set_coarse_pass('always')   # (single E-steps below: admissible every time, so that every decision can be read back)
        ctx.reset_timings()
        ctx.em(2, 0.01, pen, with_doublets=False, fetch_logits=False, fetch_probs=False, fetch_addition=False)  # P E M P E: both E-steps guarded, the first one's logits nobody reads
        history = [ctx.guard_levels()]
        for step in range(6):
            _l, probs = ctx.estep(pen, with_doublets=False)
            history.append(ctx.guard_levels())
            check_contract(probs, probs_exact, f'E-step {step} at level {history[-1]["level"]}')
    finally:
        ctx.close()
    B = p.n_barcodes
    assert history[0]['coarse_steps'] >= 1, history   # it had its turn (nothing was known before the call's first E-step)
    assert history[-1]['coarse_steps'] <= 3 and history[-1]['level'] != 0 and history[-2]['level'] != 0, history  # ... and was left
    assert max(h['flagged_coarse'] for h in history) > 0.4 * B, history
    assert all(0 < h['flagged_fine'] < 0.4 * B for h in history), history
    for prev, cur in zip(history[:-1], history[1:]):   # every decision against the rule, on the numbers the device reports
        C, F, E = cur['coarse_pass_ms'], cur['fine_pass_ms'], abs(cur['exact_pass_ms'])
        if not (C > 0 or F > 0):
            continue
        F_ = F if F > 0 else C / 0.63
        C_ = C if C > 0 else 0.63 * F_
        e_redo = E if E > 0 else 1.8 * F_
        cost = [C_ + prev['flagged_coarse'] / B * e_redo, F_ + prev['flagged_fine'] / B * e_redo, E if E > 0 else float('inf')]
        cost[prev['level']] *= 0.97
        best = min(range(3), key=lambda i: cost[i])
        ranked = sorted(cost)
        if ranked[1] / ranked[0] > 1.002:   # (not a tie within the clock's resolution)
            assert cur['level'] == best, (prev, cur, cost)
    print('levels', [(h['level'], h['flagged_coarse'], h['flagged_fine'], round(h['coarse_pass_ms'], 3), round(h['fine_pass_ms'], 3), round(h['exact_pass_ms'], 3)) for h in history])


def test_coarse_pass_on_the_padded_multi_rank_table(monkeypatch):
    """Two ranks on ONE GPU (threads over caller-provided collectives: tests/thread_plane.py), 140 000 barcodes: each rank's shard of
    70 000 runs the tile-major schedule, so its E-steps take the coarse pass - on the padded slice layout of the multi-rank genotype
    table (re-based row offsets, the all-zero row behind the padded table, the table's conversion behind the all-gather of the
    P-step's slices) - with the M-step sharded on variants.  Against ONE context in the exact mode: every posterior row within the
    contract, the additions within what such posteriors allow."""
    monkeypatch.setenv('DEMUXALOT_AMD_ESTEP', 'guarded')
    monkeypatch.setenv('DEMUXALOT_AMD_EXCHANGE', 'variant')
    from demuxalot_amd import distributed, synth
    from demuxalot_amd.device import DeviceContext
    from tests.thread_plane import ThreadWorld
    p = synth.generate(140_000, 40_000, 64, calls_per_barcode=200, seed=4102)
    betas = p.prior_betas()
    pen = np.zeros(64, dtype=np.float32)
    with DeviceContext(0) as ctx:
        ctx.set_estep_mode('exact')
        ctx.set_exact_additions(True)
        ctx.set_problem(p.n_barcodes, p.n_variants, 64, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
        ctx.set_betas(betas)
        _l, want_probs, want_add = ctx.em(5, 0.01, pen, False, fetch_logits=False)
    shared = ThreadWorld(2)

    def rank_body(plane):
        em = distributed.ShardedEM(plane, p.n_barcodes, p.v2snp, betas, p.variant_id, p.compressed_cb, p.p_base_wrong)
        try:
            em.ctx.reset_timings()
            probs, addition = em.learn(5, 0.01, pen, False)
            return em.lo, em.hi, probs, addition, em.ctx.guard_levels()
        finally:
            em.ctx.close()

    results = shared.run(rank_body)
    n_calls_v = np.bincount(p.variant_id, minlength=p.n_variants).astype(np.float64)[:, None]
    for lo, hi, probs, addition, levels in results:
        # (the last E-step keeps its logits: not the coarse pass - the fine pass, or the direct level where the device's controller, whose
        # timings two ranks sharing ONE GPU disturb, prices that lower: seen once in four runs)
        assert hi - lo >= 65_536 and levels['coarse_steps'] >= 2 and levels['level'] in (1, 2), (lo, hi, levels)
        check_contract(probs, want_probs[lo:hi], f'posterior rows [{lo}, {hi}) of the sharded run')
        assert (np.abs(addition.astype(np.float64) - want_add) <= n_calls_v * 2.00001e-5 + 2.0 ** -22 * want_add).all()
    print('levels per rank', [r[4] for r in results])


@pytest.mark.parametrize('G', [17, 32, 33, 64, 65, 100, 128])
def test_coarse_pass_shapes_on_one_table(G):
    """Every shape of the coarse pass - four calls per gather (17 .. 32 genotypes), two (33 .. 64), one (65 .. 128), at both ends of
    each range - forced for a single E-step on the table of EM iteration 1: against the exact mode every posterior within 1e-5, every
    arg-max identical, every logit within the bound the guard priced it with (kernels.h: GUARD_PER_CALL_COARSE, GUARD_ACCUM_F32)."""
    from demuxalot_amd import synth
    from demuxalot_amd.device import DeviceContext
    S = max(30_000, (9 << 20) // (8 * G) + 1000)   # a genotype table of at least 8 MB: the tile-major schedule
    p = synth.generate(66_000, S, G, calls_per_barcode=240, seed=4200 + G)
    pen = np.zeros(G, dtype=np.float32)
    ctx = DeviceContext(0)
    try:
        ctx.set_estep_mode('exact')
        ctx.set_problem(p.n_barcodes, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
        ctx.set_betas(p.prior_betas())
        ctx.em(2, 0.01, pen, with_doublets=False, fetch_logits=False, fetch_probs=False, fetch_addition=False)
        ctx.mstep(2., fetch=False)
        ctx.probs_from_betas(0.01, fetch=False)
        logits_e, probs_e = ctx.estep(pen, with_doublets=False)
        ctx.set_estep_mode('guarded')
        ctx.set_coarse_pass('always')
        logits_c, probs_c = ctx.estep(pen, with_doublets=False)
        levels, redone = ctx.guard_levels(), ctx.guard_stats()[0]
    finally:
        ctx.close()
    assert levels['level'] == 0 and levels['flagged_coarse'] == redone, levels
    dev = check_contract(probs_c, probs_e, f'coarse pass, {G} genotypes')
    n = 8 * ((np.bincount(p.compressed_cb, minlength=p.n_barcodes) + 7) // 8).astype(np.float64)[:, None]
    mag = np.abs(logits_e.astype(np.float64))
    lk = np.bincount(p.compressed_cb, weights=-np.log(1.0 - p.p_base_wrong.astype(np.float64)), minlength=p.n_barcodes)[:, None]
    bound = 5.1e-4 * (n + 8) + 6.0e-8 * (0.125 * n + 2) * (mag + 3e-4 * n + 3 * lk) + 3.0e-7 * (mag + 2.1e-4 * n) + 2.4e-7 * mag
    worst = float((np.abs(logits_c.astype(np.float64) - logits_e) / bound).max())
    assert worst <= 1.0, worst
    print(f'G={G}: {redone} of {p.n_barcodes} barcodes redone, posteriors within {dev:.3g}, logits at most {worst:.3f} of their bound')


@pytest.mark.parametrize('G', [17, 32, 33, 64, 65, 100, 128])
def test_fine_pass_on_the_coarse_records_shapes(G):
    """dmx_set_lean_memory: once the tile-major stream is released the guard's fine level walks the coarse pass's 8-byte records with the
    float32 table and float64 sums (k_estep_tiled_fine8), every shape of them: an E-step that keeps its logits against the exact mode -
    every posterior within 1e-5, every arg-max identical, every logit within the bound the guard prices the pass with
    (kernels.h: guard_per_call_fine8) plus the common term's (the sum of log2 keep: v_log_f32 results)."""
    from demuxalot_amd import synth
    from demuxalot_amd.device import DeviceContext
    S = max(30_000, (9 << 20) // (8 * G) + 1000)
    p = synth.generate(66_000, S, G, calls_per_barcode=240, seed=4300 + G)
    pen = np.zeros(G, dtype=np.float32)
    ctx = DeviceContext(0)
    try:
        ctx.set_estep_mode('exact')
        ctx.set_problem(p.n_barcodes, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
        ctx.set_betas(p.prior_betas())
        ctx.em(2, 0.01, pen, with_doublets=False, fetch_logits=False, fetch_probs=False, fetch_addition=False)
        ctx.mstep(2., fetch=False)
        ctx.probs_from_betas(0.01, fetch=False)
        logits_e, probs_e = ctx.estep(pen, with_doublets=False)
        ctx.set_estep_mode('guarded')
        ctx.set_lean_memory(True)
        ctx.set_coarse_pass('always')
        held = ctx.device_bytes()
        ctx.estep(pen, with_doublets=False, fetch_logits=False, fetch_probs=False)   # the coarse pass: its records are built, the stream goes
        assert ctx.guard_levels()['level'] == 0 and ctx.device_bytes() < held + 9 * len(p.variant_id)   # (+ 8 bytes of records - 16 of stream - 4 of rows per call)
        ctx.set_coarse_pass(True)
        ctx.set_guard_adaptive(False)   # (the fast pass + redo whatever the device's timings say: the level under test)
        logits_f, probs_f = ctx.estep(pen, with_doublets=False)
        levels, redone = ctx.guard_levels(), ctx.guard_stats()[0]
    finally:
        ctx.close()
    assert levels['level'] == 1 and levels['flagged_fine'] == redone, levels
    dev = check_contract(probs_f, probs_e, f'fine pass on the coarse records, {G} genotypes')
    n = 8 * ((np.bincount(p.compressed_cb, minlength=p.n_barcodes) + 7) // 8).astype(np.float64)[:, None]
    mag = np.abs(logits_e.astype(np.float64))
    cpg = 1 if G > 64 else 2 if G > 32 else 4
    per_call = 1.91e-6 / (8 // cpg) + 4 * 6.0e-8 + 7.0e-8
    bound = per_call * (n + 8) + 2.0e-7 * n + 3.0e-7 * (mag + 2.1e-4 * n) + 2.4e-7 * mag
    worst = float((np.abs(logits_f.astype(np.float64) - logits_e) / bound).max())
    assert worst <= 1.0, worst
    print(f'G={G}: {redone} of {p.n_barcodes} barcodes redone, posteriors within {dev:.3g}, logits at most {worst:.3f} of their bound')


def test_a_stale_time_of_a_pass_that_does_not_run_is_taken_again(separable):
    """The device times a pass only when it runs.  A coarse pass timed once at 10 ms (planted here; in the field: the first E-step on
    a device that had idled) loses against the fine pass - and would lose for ever, since it never runs again to be timed.  After 64
    E-steps in a row on the fine pass the coarse pass runs once (its standing price, planted at 1.5 x the fine pass's, is below twice
    the running level's), is timed, and takes over; the posteriors stay those of a run that never left the coarse pass, within the
    contract of either pass."""
    from demuxalot_amd.device import DeviceContext
    p = separable
    pen = np.zeros(p.n_genotypes, dtype=np.float32)
    ctx = DeviceContext(0)
    try:
        ctx.set_estep_mode('guarded')
        ctx.set_exact_additions(False)
        _install(ctx, p)
        ctx.em(6, 0.01, pen, with_doublets=False, fetch_logits=False, fetch_probs=False, fetch_addition=False)
        ctx.reset_timings()
        ctx.run_iterations(10, 0.01)
        lv = ctx.guard_levels()   # (an E-step's own time is folded in when the next one begins: the fine pass's is that of dmx_em's last)
        assert lv['coarse_steps'] == 9 and ctx.guard_probes()[0] == 0, lv
        assert lv['coarse_pass_ms'] > 0 and lv['fine_pass_ms'] > lv['coarse_pass_ms'], lv
        ctx.run_iterations(10, 0.01)                             # (the device at its clocks: the times the factors below multiply are the warm ones)
        lv = ctx.guard_levels()
        ctx.debug_set_pass_ms(coarse=1.5 * lv['fine_pass_ms'])   # stale and wrong
        ctx.reset_timings()
        ctx.run_iterations(60, 0.01)
        assert ctx.guard_levels()['coarse_steps'] == 0 and ctx.guard_probes()[0] == 0, (ctx.guard_levels(), ctx.guard_probes())
        ctx.run_iterations(40, 0.01)     # the 64th E-step on the fine pass falls into this call
        lv2, (probes, _streak) = ctx.guard_levels(), ctx.guard_probes()
        assert probes == 1, (lv2, probes)
        assert lv2['coarse_steps'] >= 30, lv2            # the probe and every admissible E-step behind it
        assert lv2['coarse_pass_ms'] < lv2['fine_pass_ms'], lv2
        # a time too far off to be worth an E-step (beyond twice the running level's price) stays: no probe
        ctx.debug_set_pass_ms(coarse=5.0 * lv['fine_pass_ms'])
        ctx.reset_timings()
        ctx.run_iterations(150, 0.01)
        # (the exact kernel's time is an estimate here since round 6 - the call's first E-step is a guarded one on the prior table and
        # queues enough barcodes to price it -, and an estimate below twice the fine pass's price earns the DIRECT level a probe)
        assert ctx.guard_levels()['coarse_steps'] == 0 and ctx.guard_probes()[0] <= 3, (ctx.guard_levels(), ctx.guard_probes())
    finally:
        ctx.close()


def test_a_call_that_returns_no_logits_may_take_the_coarse_pass_for_its_last_estep(separable):
    """learn_genotypes returns the learnt genotypes and the last iteration's posteriors, no logits (demux.py:55-66): its dmx_em call says
    so (dmx_set_logits_needed(0)) and the last E-step becomes one whose logits nobody reads, like the ones before it.  Posteriors of
    every barcode within the contract of the exact mode's call; the logits are then not served (loudly), everything else is; a logits
    output pointer, or an E-step of its own, keeps them as before."""
    from demuxalot_amd._lib import DemuxHipError
    from demuxalot_amd.device import DeviceContext
    p = separable
    pen = np.zeros(p.n_genotypes, dtype=np.float32)
    ctx = DeviceContext(0)
    try:
        ctx.set_estep_mode('exact')
        ctx.set_exact_additions(True)
        _install(ctx, p)
        _l, probs_exact, add_exact = ctx.em(6, 0.01, pen, with_doublets=False, fetch_logits=False)
        ctx.set_estep_mode('guarded')
        ctx.set_exact_additions(False)
        ctx.set_logits_needed(False)
        ctx.reset_timings()
        _l, probs, addition = ctx.em(6, 0.01, pen, with_doublets=False, fetch_logits=False)
        lv = ctx.guard_levels()
        assert lv['level'] == 0 and lv['coarse_steps'] == 6, lv      # E-steps 0 .. 5: coarse, the last one too
        check_contract(probs, probs_exact, 'no logits needed: 6 iterations vs exact')
        assert np.array_equal(ctx.get_block('probs', 0, 100), probs[:100])
        assert np.array_equal(ctx.get_assignments()[0], probs.argmax(axis=1))
        with pytest.raises(DemuxHipError, match='logits were not kept'):
            ctx.get_logits()
        with pytest.raises(DemuxHipError, match='logits were not kept'):
            ctx.get_block('logits', 0, 10)
        # the same call asked for its logits: kept, the last E-step on the fine pass
        ctx.reset_timings()
        logits, probs2, _a = ctx.em(6, 0.01, pen, with_doublets=False)
        lv = ctx.guard_levels()
        assert lv['level'] == 1 and lv['coarse_steps'] == 5, lv
        assert np.array_equal(ctx.get_logits(), logits)
        check_contract(probs2, probs_exact, 'logits asked for: 6 iterations vs exact')
        # dmx_run_iterations has no output pointers: the setting alone decides
        ctx.reset_timings()
        ctx.run_iterations(5, 0.01)
        lv = ctx.guard_levels()
        assert lv['level'] == 0 and lv['coarse_steps'] == 5, lv
        with pytest.raises(DemuxHipError, match='logits were not kept'):
            ctx.get_logits()
        ctx.estep(pen, with_doublets=False, fetch_logits=False, fetch_probs=False)   # an E-step of its own keeps them
        assert np.isfinite(ctx.get_block('logits', 0, 10)).all()
        ctx.set_logits_needed(True)
        ctx.reset_timings()
        ctx.run_iterations(5, 0.01)
        lv = ctx.guard_levels()
        assert lv['level'] == 1 and lv['coarse_steps'] == 4, lv
        assert np.isfinite(ctx.get_logits()).all()
    finally:
        ctx.close()


def test_coarse_pass_on_adversarial_rows_against_the_reference_arithmetic():
    """(Advisor, round 5.)  The coarse pass's guard constant rests on a derivation (kernels.h: GUARD_PER_CALL_COARSE) - binary16 table,
    r = floor / keep, one v_log_f32 per 4-term product - whose worst cases a generator problem never visits.  Here they are planted:
    barcodes of 1 500 calls next to ordinary ones, keep factors of 2^-24, 2^-12 and 0, error probabilities of 1e-7 and 0 (floor 1e-4),
    and a genotype table pushed to the clip (0.01 / 0.99) in a third of its entries.  One E-step forced onto the coarse pass against
    the exact mode - the reference's bits (tests/test_gpu_parity.py) - and, on a sample of rows, against the oracle itself: every
    posterior within 1e-5, every arg-max identical, every served logit within the bound the guard priced it with."""
    from demuxalot_amd import synth
    from demuxalot_amd.device import DeviceContext
    from oracle import demux_oracle
    G, S = 64, 5000
    light = synth.generate(20_000, S, G, calls_per_barcode=100, seed=4300)
    heavy = synth.generate(300, S, G, calls_per_barcode=1500, seed=4300, seed_calls=4301)   # the same genotype table, other barcodes
    B = light.n_barcodes + heavy.n_barcodes
    variant = np.concatenate([light.variant_id, heavy.variant_id])
    cb = np.concatenate([light.compressed_cb, heavy.compressed_cb + np.int32(light.n_barcodes)])
    e = np.concatenate([light.p_base_wrong, heavy.p_base_wrong]).copy()
    rng = np.random.default_rng(4302)
    pick = rng.permutation(len(e))
    n = len(e) // 50
    for j, value in enumerate((1.0 - 2.0 ** -24, 1.0 - 2.0 ** -12, 1.0, 1e-7, 0.0, 0.5)):
        e[pick[j * n:(j + 1) * n]] = np.float32(value)
    betas = light.prior_betas().copy()
    scale = rng.choice(np.array([1e-4, 1.0, 1e4], dtype=np.float32), size=betas.shape, p=[0.2, 0.6, 0.2])
    betas = (betas * scale).astype(np.float32)
    pen = np.zeros(G, dtype=np.float32)
    ctx = DeviceContext(0)
    try:
        ctx.set_estep_mode('exact')
        ctx.set_problem(B, light.n_variants, G, variant, cb, e, light.v2snp)
        ctx.set_betas(betas)
        ctx.set_addition(None)
        prob = ctx.probs_from_betas(0.01)
        assert ((prob == np.float32(0.01)) | (prob == np.float32(0.99))).mean() > 0.2   # a table at the clip
        logits_e, probs_e = ctx.estep(pen, with_doublets=False)
        ctx.set_estep_mode('guarded')
        ctx.set_coarse_pass('always')
        ctx.set_estep_dictionary('never')   # (a prior table: the dictionary form would take the E-step)
        ctx.reset_timings()
        logits_c, probs_c = ctx.estep(pen, with_doublets=False)
        levels, redone = ctx.guard_levels(), ctx.guard_stats()[0]
    finally:
        ctx.close()
    assert levels['level'] == 0, levels
    assert np.isfinite(probs_c).all() and np.isfinite(logits_c).all()
    dev = check_contract(probs_c, probs_e, 'coarse pass on adversarial rows vs the exact mode')
    # the oracle on the heavy barcodes and a sample of the light ones (float32 terms, float64 sums: the reference's arithmetic)
    rows = np.concatenate([np.arange(light.n_barcodes, B), rng.choice(light.n_barcodes, 200, replace=False)])
    take = np.isin(cb, rows)
    renumber = np.full(B, -1, dtype=np.int64)
    renumber[rows] = np.arange(len(rows))
    want_logits = demux_oracle.barcode_logits(variant[take], renumber[cb[take]], e[take], prob, len(rows), 0.)
    want_probs = demux_oracle.softmax_rows(want_logits)
    assert np.array_equal(logits_e[rows].view(np.uint32), want_logits.view(np.uint32)), 'exact mode vs the oracle on the sampled rows'
    check_contract(probs_c[rows], want_probs, 'coarse pass on adversarial rows vs the oracle')
    calls = 8 * ((np.bincount(cb, minlength=B) + 7) // 8).astype(np.float64)[:, None]
    mag = np.abs(logits_e.astype(np.float64))
    keep = 1.0 - e.astype(np.float64)
    lk = np.bincount(cb[keep > 0], weights=-np.log(keep[keep > 0]), minlength=B)[:, None]
    bound = 5.1e-4 * (calls + 8) + 6.0e-8 * (0.125 * calls + 2) * (mag + 3e-4 * calls + 3 * lk) + 3.0e-7 * (mag + 2.1e-4 * calls) + 2.4e-7 * mag
    kept = np.ones(B, dtype=bool)   # (a barcode the guard queued carries the exact kernel's logits: inside any bound)
    worst = float((np.abs(logits_c.astype(np.float64) - logits_e) / bound)[kept].max())
    assert worst <= 1.0, worst
    print(f'adversarial rows: {redone} of {B} barcodes redone exactly, posteriors within {dev:.3g}, logits at most {worst:.3f} of their bound; '
          f'heaviest barcode {int(np.bincount(cb).max())} calls')
