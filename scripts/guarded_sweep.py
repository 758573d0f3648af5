"""Randomised check of the GUARDED E-step against the exact one (GPU box): python scripts/guarded_sweep.py [n_trials] [first_seed]
Every trial: a random problem (genotypes, doublets or not, barcodes, heavy-tailed / empty / very long rows so that rows get
split), its importer-style table and the tables after one and two M-steps; on each the E-step runs in the exact and in the
guarded mode (dictionary form off: the kernels under test) and the contract must hold on EVERY barcode - argmax identical,
|posterior - exact posterior| <= 1e-5 -, the barcodes the guard queued must carry the exact bits, and a second guarded run
must reproduce the first bit for bit.  Prints the redo fractions."""
import os
import sys
import time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from demuxalot_amd import Demultiplexer, synth
from demuxalot_amd.device import DeviceContext

n_trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t0 = time.time()
rows = redone_total = 0
worst = 0.0
for trial in range(first, first + n_trials):
    rng = np.random.default_rng(120000 + trial)
    doublets = bool(rng.random() < 0.5)
    G = int(rng.integers(2, 60)) if doublets else int(rng.choice([2, 3, 5, 8, 9, 16, 17, 31, 32, 33, 48, 64, 65, 100, 128, 200, 256, 300, 600]))
    B = int(rng.choice([37, 200, 701, 2500]))
    S = int(rng.integers(60, 3000))
    cpb = int(rng.choice([8, 40, 120, 300, 900]))
    dp = float(rng.choice([0.1, 0.35])) if doublets else 0.0
    p = synth.generate(B, S, G, calls_per_barcode=min(cpb, S), doublets=doublets, seed=9000 + trial)
    variant, cb, e = p.variant_id, p.compressed_cb, p.p_base_wrong
    if rng.random() < 0.5:
        drop = np.isin(cb, rng.choice(B, size=max(1, B // 20), replace=False))
        variant, cb, e = variant[~drop], cb[~drop], e[~drop]
    pen = Demultiplexer._doublet_penalties(G, dp)
    what = f'trial {trial}: G={G} K={len(pen)} B={B} S={S} cpb={cpb} dp={dp} N={len(cb)}'
    ctx = DeviceContext(0)
    try:
        ctx.set_estep_dictionary('never')
        ctx.set_guard_adaptive(False)  # (two E-steps of one table must run the same form for the bit comparison below: fast pass + redo both times)
        ctx.set_problem(B, p.n_variants, G, variant, cb, e, p.v2snp)
        ctx.set_betas(p.prior_betas(add_data_prior=False))
        ctx.set_addition(None)
        for stage in range(3):
            ctx.probs_from_betas(0.01, fetch=False)
            ctx.set_estep_mode('exact')
            ctx.set_exact_additions(True)
            le, pe = ctx.estep(pen, with_doublets=doublets)
            addition = ctx.mstep(2.)
            ctx.set_estep_mode('guarded')
            ctx.set_phase_timers(True); ctx.reset_timings()
            lg, pg = ctx.estep(pen, with_doublets=doublets)
            redone, _t, n_rows = ctx.guard_stats()
            lg2, pg2 = ctx.estep(pen, with_doublets=doublets)
            assert np.array_equal(lg.view(np.uint32), lg2.view(np.uint32)) and np.array_equal(pg.view(np.uint32), pg2.view(np.uint32)), what + ': not reproducible'
            dev = np.abs(pg.astype(np.float64) - pe)
            assert dev.max() <= 1e-5, f'{what} stage {stage}: posterior deviation {dev.max():.3g}'
            assert np.array_equal(pg.argmax(1), pe.argmax(1)), f'{what} stage {stage}: assignments differ'
            same = (lg.view(np.uint32) == le.view(np.uint32)).all(axis=1) & (pg.view(np.uint32) == pe.view(np.uint32)).all(axis=1)
            assert n_rows == B and same.sum() >= redone, (what, n_rows, int(same.sum()), redone)
            rows += B
            redone_total += redone
            worst = max(worst, float(dev.max()))
            ctx.set_addition(addition)
        print('ok', what, f'redone {redone}/{B}', flush=True)
    finally:
        ctx.close()
print(f'{n_trials} trials: contract holds on every barcode; {redone_total} of {rows} barcode rows redone exactly ({100 * redone_total / max(1, rows):.1f} %), '
      f'worst posterior deviation {worst:.3g}; {time.time() - t0:.0f} s')
