"""Randomised check of the call learn_genotypes makes (dmx_em, no logits back: dmx_set_logits_needed(0), the library's default modes -
coarse pass where the device takes it, its last E-step included, incremental M-step where the tile form runs) against the same call in
the exact mode: random genotype counts 17 .. 128, calls per barcode, clips, sibling donors, degenerate error probabilities, 3 .. 12
iterations.  Per problem: the posteriors of the LAST iteration on every barcode within 1e-5 and the same arg-max; the additions within
what such posteriors allow (n(v) 2e-5 + float32 roundings).  GPU box: [SWEEP_SMALL=1] python3 scripts/em_call_sweep.py [n_problems] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from demuxalot_amd import synth
from demuxalot_amd.device import DeviceContext

n_problems = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2027)
worst_dev, worst_add, t0 = 0.0, 0.0, time.time()
for trial in range(n_problems):
    G = int(rng.integers(17, 129))
    cpb = int(rng.choice([24, 60, 150, 400]))
    clip = float(rng.choice([0.01, 0.01, 0.002, 1e-4]))
    siblings = bool(rng.random() < 0.3)
    n_it = int(rng.choice([3, 5, 5, 8, 12]))
    if os.environ.get('SWEEP_SMALL'):
        B = int(rng.integers(8_200, 40_000))
        S = int(rng.integers((1 << 20) // (8 * G) + 50, (6 << 20) // (8 * G) + 100))
    else:
        B = int(rng.integers(66_000, 90_000))
        S = int(max(40_000, (9 << 20) // (8 * G) + 1000))
    p = synth.generate(B, S, G, calls_per_barcode=cpb, seed=int(rng.integers(1, 1 << 30)), sibling_pairs=siblings)
    e = p.p_base_wrong.copy()
    odd = rng.random(len(e)) < 0.002
    e[odd] = rng.choice(np.array([0.0, 0.999999, 1.0, 0.5], dtype=np.float32), size=int(odd.sum()))
    pen = np.zeros(G, dtype=np.float32)
    out = {}
    for mode in ('exact', 'guarded'):
        ctx = DeviceContext(0)
        try:
            ctx.set_estep_mode(mode)
            ctx.set_exact_additions(mode == 'exact')
            ctx.set_logits_needed(False)
            ctx.set_problem(B, p.n_variants, G, p.variant_id, p.compressed_cb, e, p.v2snp)
            ctx.set_betas(p.prior_betas())
            ctx.reset_timings()
            _l, probs, addition = ctx.em(n_it, clip, pen, with_doublets=False, fetch_logits=False)
            out[mode] = (probs, addition, ctx.guard_levels(), ctx.guard_stats(), ctx.mstep_incremental(), ctx.mstep_form())
        finally:
            ctx.close()
    probs_e, add_e = out['exact'][:2]
    probs_g, add_g, levels, stats, incr, form = out['guarded']
    dev = np.abs(probs_g.astype(np.float64) - probs_e).max()
    same = bool((probs_g.argmax(axis=1) == probs_e.argmax(axis=1)).all())
    n_calls_v = np.bincount(p.variant_id, minlength=p.n_variants).astype(np.float64)[:, None]
    d_add = np.abs(add_g.astype(np.float64) - add_e)
    add_ratio = float((d_add / (n_calls_v * 2.00001e-5 + 2.0 ** -22 * np.abs(add_e) + 1e-30)).max())
    worst_dev, worst_add = max(worst_dev, dev), max(worst_add, add_ratio)
    print(f'{trial:3d} G={G:3d} B={B} S={S} calls/barcode={cpb:3d} clip={clip:g} siblings={int(siblings)} iterations={n_it:2d}: coarse E-steps {levels["coarse_steps"]} '
          f'(last level {levels["level"]}), redone {stats[1]} of {stats[2]} barcode rows, M-step {form} full / delta {incr[0]} / {incr[1]}: '
          f'max |dp| {dev:.3g}, arg-max identical {same}, additions at most {add_ratio:.3f} of their bound', flush=True)
    assert dev <= 1e-5 and same and add_ratio <= 1.0, 'CONTRACT VIOLATED'
print(f'{n_problems} problems in {time.time() - t0:.0f} s: worst posterior deviation {worst_dev:.3g}, additions at most {worst_add:.3f} of their bound')
