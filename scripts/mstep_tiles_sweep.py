"""Randomised check of the tile-major M-step against the exact additions (GPU box): python scripts/mstep_tiles_sweep.py [n_trials] [first_seed]
Random shapes (1..64 genotypes, with and without doublets, 1..60k barcodes, hot and cold variants, contribution powers 2 and others,
informative and flat genotypes): every addition within a float32 ulp of the exact one, all but a sliver identical."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from demuxalot_amd import Demultiplexer, synth
from demuxalot_amd.device import DeviceContext

n_trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t0 = time.time()
worst_frac, entries = 0.0, 0
for trial in range(first, first + n_trials):
    rng = np.random.default_rng(trial)
    G = int(rng.choice([1, 2, 3, 5, 8, 13, 16, 24, 32, 33, 48, 63, 64]))
    doublets = bool(rng.random() < 0.3) and G <= 24 and G > 1
    B = int(rng.choice([1, 7, 300, 3000, 20000, 60000]))
    S = int(rng.choice([1, 3, 40, 700, 5000]))
    cpb = int(rng.choice([1, 5, 40, 200]))
    power = float(rng.choice([2.0, 2.0, 1.0, 1.5, 3.0]))
    flat = rng.random() < 0.15
    p = synth.generate(B, S, G, calls_per_barcode=cpb, doublets=doublets, seed=10_000 + trial)
    pen = Demultiplexer._doublet_penalties(G, 0.25 if doublets else 0.0)
    betas = np.ones_like(p.prior_betas()) if flat else p.prior_betas()
    got = {}
    for name, exact in (('exact', True), ('tiles', False)):
        ctx = DeviceContext(0)
        try:
            ctx.set_estep_mode('exact')
            ctx.set_exact_additions(exact)
            ctx.set_mstep_tiles('always')
            ctx.set_problem(p.n_barcodes, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
            ctx.set_betas(betas)
            ctx.set_addition(None)
            outs = []
            for _ in range(2):
                ctx.probs_from_betas(0.01, fetch=False)
                ctx.estep(pen, with_doublets=doublets, fetch_logits=False, fetch_probs=False)
                outs.append(ctx.mstep(power))
            got[name] = (outs, ctx.mstep_form())
        finally:
            ctx.close()
    assert got['exact'][1] == 'items'
    for a, b in zip(got['tiles'][0], got['exact'][0]):
        assert np.isfinite(a).all() and np.allclose(a, b, rtol=3e-7, atol=0), (trial, G, B, S, cpb, power, flat, float(np.abs(a - b).max()))
        frac = float((a != b).mean())
        worst_frac = max(worst_frac, frac if a.size > 2000 else 0.0)
        entries += a.size
    print(f'ok trial {trial}: G={G} doublets={doublets} B={B} S={S} cpb={cpb} power={power} flat={flat} N={p.n_calls} form={got["tiles"][1]}', flush=True)
print(f'{n_trials} trials: every addition within a float32 ulp of the exact one; {entries} entries, largest fraction differing in a table of more '
      f'than 2000 entries {worst_frac:.2e}; {time.time() - t0:.0f} s')
