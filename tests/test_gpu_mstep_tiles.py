"""Tile-major M-step (kernels.hip: k_mstep_tiles; taken when the exact additions are off and G <= 64) against the
work-item form and against the exact additions: the sums are float64 in another order, so a float32 rounding tie at most."""
import numpy as np
import pytest

from tests.thread_plane import ThreadWorld

pytestmark = pytest.mark.gpu


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
        assert np.allclose(results['tiles'][it], want, rtol=3e-7, atol=0), (G, it, np.abs(results['tiles'][it] - want).max())
        # float64 sums, one rounding: all but a handful of entries are the reference's bits
        assert (results['tiles'][it] != want).mean() < 1e-3


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
        assert np.allclose(got, want, rtol=3e-7, atol=0)
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
    assert np.allclose(flat, out['exact'][0], rtol=3e-7, atol=0) and np.allclose(sharp, want, rtol=3e-7, atol=0)


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
        assert np.allclose(addition, want_add, rtol=3e-7, atol=0)
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
                assert np.allclose(x, y, rtol=3e-7, atol=0)
    finally:
        ctx.close()
