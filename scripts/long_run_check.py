"""GPU box: a long EM run (the tile-major M-step takes over by itself) in the default mode against the exact mode."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from demuxalot_amd import synth
from demuxalot_amd.device import DeviceContext
n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for (B, S, G, cpb, seed) in ((20000, 10000, 64, 200, 77), (60000, 3000, 24, 60, 78), (5000, 20000, 8, 400, 79)):
    p = synth.generate(B, S, G, calls_per_barcode=cpb, seed=seed)
    pen = np.zeros(G, dtype=np.float32)
    out = {}
    for mode in ('exact', 'guarded'):
        ctx = DeviceContext(0)
        ctx.set_estep_mode(mode); ctx.set_exact_additions(mode == 'exact')
        ctx.set_problem(B, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
        ctx.set_betas(p.prior_betas())
        _l, probs, add = ctx.em(n_it, 0.01, pen, False, fetch_logits=False)
        out[mode] = (probs, add, ctx.mstep_form(), ctx.guard_stats())
        ctx.close()
    pe, pg = out['exact'][0], out['guarded'][0]
    print(f'{B}x{S}x{G}, {n_it} iterations: M-step forms {out["exact"][2]} / {out["guarded"][2]}; assignments identical '
          f'{bool(np.array_equal(pe.argmax(1), pg.argmax(1)))}; max |posterior difference| {float(np.abs(pe - pg).max()):.2e}; '
          f'additions max relative difference {float((np.abs(out["exact"][1] - out["guarded"][1]) / np.maximum(np.abs(out["exact"][1]), 1e-30)).max()):.2e}; guard {out["guarded"][3]}', flush=True)
