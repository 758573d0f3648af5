"""Drift of the default mode over EM iterations.  Per E-step the guarded mode PROVES its posteriors within
1e-5 of the reference's ON THE SAME TABLE; across iterations the tables differ by what such posteriors do to the M-step.
This script runs 10 EM iterations on the reference's fixtures (F1, F2, F5, the shipped example) and on three larger generator
problems in the default mode (guarded E-step; M-step: work items, and the tile-major form forced) next to the exact mode -
which is the reference bit for bit (tests/test_gpu_parity.py) - and records, per iteration, the largest posterior deviation,
the assignments that differ, and the largest deviation of the beta addition in units of its bound n(v) 2e-5.
GPU box: python scripts/guarded_drift.py > profiles/r5_guarded_drift.txt"""
import sys

import numpy as np

sys.path.insert(0, '.')
from demuxalot_amd import Demultiplexer, synth  # noqa: E402
from demuxalot_amd.device import DeviceContext  # noqa: E402
from tests import fixture_io as fio  # noqa: E402

N_IT = 10


def staged(ctx, pen, doublets, prior_logits=None):
    out = []
    ctx.set_addition(None)
    for it in range(N_IT):
        ctx.probs_from_betas(0.01, fetch=False)
        _l, probs = ctx.estep(pen, with_doublets=doublets, prior_logits=prior_logits if it == 0 else None, fetch_logits=False)
        out.append((probs, ctx.mstep(2.)))
    return out


def run(name, install, G, dp, n_calls_per_variant):
    pen = Demultiplexer._doublet_penalties(G, dp)
    results = {}
    for mode, tiles in (('exact', 'never'), ('guarded', 'never'), ('guarded', 'always')):
        ctx = DeviceContext(0)
        try:
            ctx.set_estep_mode(mode)
            ctx.set_exact_additions(mode == 'exact')
            ctx.set_mstep_tiles(tiles)
            install(ctx)
            ctx.set_phase_timers(True); ctx.reset_timings()
            results[(mode, tiles)] = (staged(ctx, pen, dp > 0), ctx.guard_stats())
        finally:
            ctx.close()
    ref = results[('exact', 'never')][0]
    bound = n_calls_per_variant[:, None] * 2.00001e-5
    for key in (('guarded', 'never'), ('guarded', 'always')):
        got, (_last, redone, rows) = results[key]
        line = []
        for it in range(N_IT):
            dev = float(np.abs(got[it][0].astype(np.float64) - ref[it][0]).max())
            flips = int((got[it][0].argmax(1) != ref[it][0].argmax(1)).sum())
            add = float((np.abs(got[it][1].astype(np.float64) - ref[it][1]) / np.maximum(bound + 2.0 ** -22 * ref[it][1], 1e-300)).max())
            line.append(f'{dev:.2e}/{flips}/{add:.2e}')
        print(f'{name:28s} M-step {"tiles" if key[1] == "always" else "items"}: redone {redone}/{rows}; per iteration max|dposterior| / assignments that differ / '
              f'max addition deviation in units of its bound: ' + '  '.join(line), flush=True)


for fixture in ('f1_synthetic_default.npz', 'f2_synthetic_g4.npz', 'f5_generator_2k_5k_16.npz', 'f6_shipped_example.npz'):
    try:
        fx = fio.load(fixture)
    except FileNotFoundError:
        continue
    if 'pack_bc_variant_id' not in fx:  # (F5 keeps the generator's arrays: it is the first generator problem below)
        continue
    G = len(fx['genotype_names'])
    n_variants = len(fx['pack1_betas'])

    def install(ctx, fx=fx, G=G, n_variants=n_variants):
        ctx.set_problem(len(fx['barcodes']), n_variants, G, fx['pack_bc_variant_id'], fx['pack_bc_cb'], fx['pack_bc_p'], fx['pack_v2snp'])
        ctx.set_betas(fx['pack1_betas'])

    for dp in (0.0, 0.25):
        run(f'{fixture[:-4]} dp={dp}', install, G, dp, np.bincount(fx['pack_bc_variant_id'], minlength=n_variants).astype(np.float64))

for B, S, G, cpb, dp, seed in ((2000, 5000, 16, 300, 0.25, 4242),  # F5's problem (tests/golden/make_fixtures.py)
                              (20000, 10000, 64, 200, 0.0, 4064), (60000, 3000, 24, 100, 0.0, 4024), (5000, 20000, 8, 300, 0.35, 4008),
                              (200000, 100000, 64, 400, 0.0, 1237)):
    p = synth.generate(B, S, G, calls_per_barcode=cpb, doublets=dp > 0, seed=seed)

    def install(ctx, p=p, G=G):
        ctx.set_problem(p.n_barcodes, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
        ctx.set_betas(p.prior_betas())

    run(f'generator {B}x{S}x{G} dp={dp}', install, G, dp, np.bincount(p.variant_id, minlength=p.n_variants).astype(np.float64))
