"""E-step of the three arithmetic / schedule combinations on short and long barcode rows (200k barcodes x 100k SNPs x 64
genotypes at 25 .. 400 calls per barcode): where does the tile-major schedule stop paying, and what is fast : exact there."""
import sys
import numpy as np

sys.path.insert(0, '.')
from demuxalot_amd import synth  # noqa: E402
from demuxalot_amd.device import DeviceContext  # noqa: E402

G = 64
pen = np.zeros(G, dtype=np.float32)
for cpb in (25, 50, 100, 200, 400):
    p = synth.generate(200_000, 100_000, G, calls_per_barcode=cpb, seed=1237 + cpb)
    ctx = DeviceContext(0)
    ctx.set_estep_dictionary('never')
    ctx.set_problem(p.n_barcodes, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
    ctx.set_betas(p.prior_betas())
    ctx.set_addition(None)
    ctx.probs_from_betas(0.01, fetch=False)
    out = {}
    for mode, schedule in (('fast', 'auto'), ('fast', 'direct'), ('exact', 'auto')):
        ctx.set_estep_mode(mode)
        ctx.set_estep_schedule(schedule)
        ctx.estep(pen, with_doublets=False, fetch_logits=False, fetch_probs=False)
        ctx.set_phase_timers(True); ctx.reset_timings()
        for _ in range(10):
            ctx.estep(pen, with_doublets=False, fetch_logits=False, fetch_probs=False)
        t = ctx.timings()['estep']
        out[f'{mode}/{schedule}'] = t['ms'] / t['launches']
    print(cpb, p.n_calls, {k: round(v, 4) for k, v in out.items()}, flush=True)
    ctx.close()
