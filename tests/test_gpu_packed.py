"""Narrow doublet tables: several option slots per lane (csrc/estep_packed.hip) against the direct form and the oracle
(-m gpu).  Same float32 terms in the same order (demuxalot/demux.py:246-265 over the options of :175-191), so
everything here is BITWISE equality."""
import numpy as np
import pytest

from tests import fixture_io as fio
from tests.test_gpu_dictionary import random_calls

pytestmark = pytest.mark.gpu

# genotypes -> (options, form the exact mode picks): 8 lanes x 3 / 5 slots, 16 x 3 / 5, 32 x 3 / 5 where that wastes fewer
# slots than the next power of two (up to 64 lanes) or multiple of 64
SHAPES = {4: (10, 'direct'), 5: (15, 'direct'), 6: (21, 'packed'), 7: (28, 'direct'), 8: (36, 'packed'), 9: (45, 'packed'),
          10: (55, 'direct'), 11: (66, 'packed'), 12: (78, 'packed'), 13: (91, 'packed'), 14: (105, 'direct'),
          16: (136, 'packed'), 17: (153, 'packed'), 18: (171, 'direct')}


@pytest.mark.parametrize('n_genotypes', sorted(SHAPES))
@pytest.mark.parametrize('with_prior', [False, True])
def test_packed_form_equals_direct_form(oracle, n_genotypes, with_prior):
    from demuxalot_amd import Demultiplexer
    from demuxalot_amd.device import DeviceContext
    n_options, form = SHAPES[n_genotypes]
    rng = np.random.default_rng(77 * n_genotypes + with_prior)
    n_barcodes, n_rows = 701, 600   # 701: the last wavefront has lane groups without a barcode
    variant, cb, e = random_calls(rng, n_barcodes, n_rows, 70)
    table = rng.uniform(0.01, 0.99, size=(n_rows, n_genotypes)).astype(np.float32)
    pen = Demultiplexer._doublet_penalties(n_genotypes, 0.3)
    assert len(pen) == n_options
    prior = rng.normal(size=(n_barcodes, n_options)).astype(np.float32) if with_prior else None
    ctx = DeviceContext(0)
    try:
        ctx.set_estep_dictionary('never')
        ctx.set_problem(n_barcodes, n_rows, n_genotypes, variant, cb, e, np.arange(n_rows, dtype=np.int32))
        ctx.set_probs(table)
        ctx.set_estep_packing('never')
        l_dir, p_dir = ctx.estep(pen, with_doublets=True, prior_logits=prior)
        assert ctx.estep_form()[0] == 'direct'
        add_dir = ctx.mstep(2.)
        ctx.set_estep_packing('always')
        l_pk, p_pk = ctx.estep(pen, with_doublets=True, prior_logits=prior)
        assert ctx.estep_form()[0] == form
        fio.assert_bitwise(l_pk, l_dir, 'logits: packed vs direct form')
        fio.assert_bitwise(p_pk, p_dir, 'posteriors: packed vs direct form')
        if not with_prior:
            want = oracle.barcode_logits(variant, cb, e, table, n_barcodes, 0.3, log_impl='npsimd')
            fio.assert_bitwise(l_pk, want, 'logits: packed form vs oracle')
        # the M-step reads what the E-step epilogue left (bitmaps, barcode codes): same additions either way
        fio.assert_bitwise(ctx.mstep(2.), add_dir, 'M-step after either form')
        # the longest barcodes on 64 lanes inside the packed launch (here: nearly all of them - few calls per SIMD)
        ctx.set_estep_packing('split')
        l_sp, p_sp = ctx.estep(pen, with_doublets=True, prior_logits=prior)
        assert ctx.estep_form()[0] == form
        fio.assert_bitwise(l_sp, l_dir, 'logits: split launch vs direct form')
        fio.assert_bitwise(p_sp, p_dir, 'posteriors: split launch vs direct form')
        fio.assert_bitwise(ctx.mstep(2.), add_dir, 'M-step after the split launch')
    finally:
        ctx.close()


def test_packed_form_rows_of_very_different_lengths(oracle):
    """Lane groups of one wavefront run out of calls at different steps (and some have none): the exhausted ones read
    the neutral record behind the last row."""
    from demuxalot_amd import Demultiplexer
    from demuxalot_amd.device import DeviceContext
    rng = np.random.default_rng(3)
    n_barcodes, n_rows, n_genotypes = 37, 500, 8
    lengths = rng.integers(0, 400, size=n_barcodes)
    lengths[[0, 5, 36]] = 0
    lengths[7] = 500
    cb = np.repeat(np.arange(n_barcodes, dtype=np.int32), lengths)
    variant = np.concatenate([rng.choice(n_rows, size=k, replace=False) for k in lengths]).astype(np.int32)
    e = rng.uniform(0, 0.2, size=len(cb)).astype(np.float32)
    order = np.lexsort((cb, variant))
    variant, cb, e = variant[order], cb[order], e[order]
    table = rng.uniform(0.01, 0.99, size=(n_rows, n_genotypes)).astype(np.float32)
    pen = Demultiplexer._doublet_penalties(n_genotypes, 0.2)
    with DeviceContext(0) as ctx:
        ctx.set_estep_dictionary('never')
        ctx.set_estep_packing('always')
        ctx.set_problem(n_barcodes, n_rows, n_genotypes, variant, cb, e, np.arange(n_rows, dtype=np.int32))
        ctx.set_probs(table)
        logits, _ = ctx.estep(pen, with_doublets=True)
        assert ctx.estep_form()[0] == 'packed'
        want = oracle.barcode_logits(variant, cb, e, table, n_barcodes, 0.2, log_impl='npsimd')
        fio.assert_bitwise(logits, want, 'logits: packed form vs oracle, ragged rows')


@pytest.mark.parametrize('n_genotypes', [8, 12, 16])
def test_split_launch_long_rows_on_64_lanes(oracle, n_genotypes):
    """19 000 short rows and 1 000 long ones: 'auto' lets the long ones (more calls than a SIMD gets on average) walk on
    64 lanes and packs the rest, in one launch; bit-identical to the direct form and the oracle."""
    from demuxalot_amd import Demultiplexer
    from demuxalot_amd.device import DeviceContext
    rng = np.random.default_rng(21 + n_genotypes)
    n_barcodes, n_rows = 20000, 400
    lengths = np.full(n_barcodes, 20)
    lengths[rng.choice(n_barcodes, size=1000, replace=False)] = 200
    cb = np.repeat(np.arange(n_barcodes, dtype=np.int32), lengths)
    variant = np.concatenate([rng.choice(n_rows, size=k, replace=False) for k in lengths]).astype(np.int32)
    e = rng.uniform(0, 0.2, size=len(cb)).astype(np.float32)
    order = np.lexsort((cb, variant))
    variant, cb, e = variant[order], cb[order], e[order]
    table = rng.uniform(0.01, 0.99, size=(n_rows, n_genotypes)).astype(np.float32)
    pen = Demultiplexer._doublet_penalties(n_genotypes, 0.2)
    with DeviceContext(0) as ctx:
        ctx.set_estep_dictionary('never')
        ctx.set_problem(n_barcodes, n_rows, n_genotypes, variant, cb, e, np.arange(n_rows, dtype=np.int32))
        ctx.set_probs(table)
        logits, probs = ctx.estep(pen, with_doublets=True)
        assert ctx.estep_form()[0] == 'packed'   # auto: 1 000 of 20 000 rows are long
        added = ctx.mstep(2.)
        ctx.set_estep_packing('never')
        l_dir, p_dir = ctx.estep(pen, with_doublets=True)
        assert ctx.estep_form()[0] == 'direct'
        fio.assert_bitwise(logits, l_dir, 'logits: split launch vs direct form')
        fio.assert_bitwise(probs, p_dir, 'posteriors: split launch vs direct form')
        fio.assert_bitwise(added, ctx.mstep(2.), 'M-step after either')
        want = oracle.barcode_logits(variant, cb, e, table, n_barcodes, 0.2, log_impl='npsimd')
        fio.assert_bitwise(logits, want, 'logits: split launch vs oracle')


def test_packed_form_through_the_front_end():
    """configs[1]'s shape (8 genotypes with doublets, K = 36) through Demultiplexer.predict_posteriors and learn_genotypes,
    with and without packing (the reference's captured doublet problems, F3, have 5 genotypes: K = 15, direct form)."""
    import os
    from demuxalot_amd import Demultiplexer, synth
    calls, genotypes, handler = synth.as_objects(synth.generate(900, 700, 8, doublets=True, seed=11, seed_calls=12))
    out = {}
    for packed in ('always', 'never'):
        os.environ['DEMUXALOT_AMD_ESTEP_PACKED'] = packed
        try:
            logits, probs = Demultiplexer.predict_posteriors(calls, genotypes, handler, doublet_prior=0.35)
            learnt, last = Demultiplexer.learn_genotypes(calls, genotypes, handler, doublet_prior=0.35, n_iterations=3)
        finally:
            os.environ.pop('DEMUXALOT_AMD_ESTEP_PACKED')
        out[packed] = (logits.values, probs.values, learnt.variant_betas, last.values)
    for i, what in enumerate(('logits', 'posteriors', 'learnt betas', 'posteriors after 3 iterations')):
        fio.assert_bitwise(out['always'][i], out['never'][i], 'front-end ' + what)


def test_auto_mode_weighs_the_longest_row():
    """'auto' packs when the calls per SIMD are a multiple of the longest barcode's (slots times longer) serial walk."""
    from demuxalot_amd import Demultiplexer
    from demuxalot_amd.device import DeviceContext
    rng = np.random.default_rng(9)
    pen = Demultiplexer._doublet_penalties(8, 0.2)
    table = rng.uniform(0.01, 0.99, size=(64, 8)).astype(np.float32)
    for n_barcodes, per_barcode, expect in ((2000, 40, 'direct'), (300000, 16, 'packed')):
        cb = np.repeat(np.arange(n_barcodes, dtype=np.int32), per_barcode)
        variant = np.tile(np.arange(per_barcode, dtype=np.int32), n_barcodes)
        e = np.full(len(cb), 0.01, dtype=np.float32)
        order = np.lexsort((cb, variant))
        with DeviceContext(0) as ctx:
            ctx.set_estep_dictionary('never')
            ctx.set_problem(n_barcodes, 64, 8, variant[order], cb[order], e[order], np.arange(64, dtype=np.int32))
            ctx.set_probs(table)
            ctx.estep(pen, with_doublets=True, fetch_logits=False, fetch_probs=False)
            assert ctx.estep_form()[0] == expect, (n_barcodes, ctx.estep_form())
