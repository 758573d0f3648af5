"""Randomised check of the coarse pass (k_estep_tiled_coarse<1|2|4>) on problems large enough for the tile-major
schedule: random genotype counts 17 .. 128 (odd ones included: one, two or four calls per gather), calls per barcode, P-step clips down to binary16's normal range,
sibling donors, degenerate error probabilities.  Per problem: one EM iteration in the exact mode, then on the same table the exact
E-step and the coarse pass forced for a single E-step (dmx_set_coarse_pass 2): every posterior within 1e-5, every arg-max identical,
every logit within the bound the guard priced it with.  GPU box: [SWEEP_SMALL=1] python3 scripts/coarse_sweep.py [n_problems] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from demuxalot_amd import synth
from demuxalot_amd.device import DeviceContext

n_problems = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
worst_ratio, worst_dev, t0 = 0.0, 0.0, time.time()
for trial in range(n_problems):
    G = int(rng.integers(17, 129))
    cpb = int(rng.choice([24, 60, 150, 400]))
    clip = float(rng.choice([0.01, 0.01, 0.002, 1e-4]))
    siblings = bool(rng.random() < 0.3)
    if os.environ.get('SWEEP_SMALL'):                       # the schedule's smallest sizes: 8 192 barcodes, a table of 1 MB
        B = int(rng.integers(8_200, 40_000))
        S = int(rng.integers((1 << 20) // (8 * G) + 50, (6 << 20) // (8 * G) + 100))
    else:
        B = int(rng.integers(66_000, 90_000))
        S = int(max(40_000, (9 << 20) // (8 * G) + 1000))   # a genotype table of 9 MB and more
    p = synth.generate(B, S, G, calls_per_barcode=cpb, seed=int(rng.integers(1, 1 << 30)), sibling_pairs=siblings)
    e = p.p_base_wrong.copy()
    odd = rng.random(len(e)) < 0.002                         # degenerate error probabilities: 0, ~1, exactly 1
    e[odd] = rng.choice(np.array([0.0, 0.999999, 1.0, 0.5], dtype=np.float32), size=int(odd.sum()))
    pen = np.zeros(G, dtype=np.float32)
    ctx = DeviceContext(0)
    try:
        ctx.set_estep_mode('exact')
        ctx.set_problem(B, p.n_variants, G, p.variant_id, p.compressed_cb, e, p.v2snp)
        ctx.set_betas(p.prior_betas())
        ctx.em(2, clip, pen, with_doublets=False, fetch_logits=False, fetch_probs=False, fetch_addition=False)
        ctx.mstep(2., fetch=False)
        ctx.probs_from_betas(clip, fetch=False)
        logits_e, probs_e = ctx.estep(pen, with_doublets=False)
        ctx.set_estep_mode('guarded')
        ctx.set_coarse_pass('always')
        logits_c, probs_c = ctx.estep(pen, with_doublets=False)
        levels = ctx.guard_levels()
        redone = ctx.guard_stats()[0]
    finally:
        ctx.close()
    assert levels['level'] == 0, levels
    n = 8 * ((np.bincount(p.compressed_cb, minlength=B) + 7) // 8).astype(np.float64)[:, None]
    dev = np.abs(probs_c.astype(np.float64) - probs_e).max()
    same = bool((probs_c.argmax(axis=1) == probs_e.argmax(axis=1)).all())
    mag = np.abs(logits_e.astype(np.float64))
    keep = 1.0 - e.astype(np.float32)
    lk = np.bincount(p.compressed_cb, weights=np.where(keep > 0, -np.log(np.maximum(keep, 1e-30)), 0.0), minlength=B)[:, None]
    bound = 5.1e-4 * (n + 8) + 6.0e-8 * (0.125 * n + 2) * (mag + 3e-4 * n + 3 * lk) + 3.0e-7 * (mag + 2.1e-4 * n) + 2.4e-7 * mag
    ratio = (np.abs(logits_c.astype(np.float64) - logits_e) / bound).max()
    worst_ratio, worst_dev = max(worst_ratio, ratio), max(worst_dev, dev)
    print(f'{trial:3d} G={G:2d} B={B} S={S} calls/barcode={cpb:3d} clip={clip:g} siblings={int(siblings)}: redone {redone} ({100 * redone / B:.2f} %), '
          f'max |dp| {dev:.3g}, arg-max identical {same}, worst logit deviation {ratio:.3f} of its bound', flush=True)
    assert dev <= 1e-5 and same and ratio <= 1.0, 'CONTRACT VIOLATED'
print(f'{n_problems} problems in {time.time() - t0:.0f} s: worst posterior deviation {worst_dev:.3g}, worst logit deviation {worst_ratio:.3f} of its bound')
