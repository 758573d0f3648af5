"""Does the adaptive guarded mode choose well on small problems?  E-step time and the guard's own state per EM iteration,
adaptation on and off, for shards of the 200k x 100k x 64 experiment (what a rank of a 4- / 8-GPU run holds)."""
import sys
import numpy as np
sys.path.insert(0, '.')
from demuxalot_amd import synth
from demuxalot_amd.device import DeviceContext
G = 64
pen = np.zeros(G, dtype=np.float32)
whole = synth.generate(200_000, 100_000, G, seed=1237)
for B in (25_000, 50_000, 100_000):
    v, cb, e = whole.subset_barcodes(0, B)
    for adaptive in (True, False):
        ctx = DeviceContext(0)
        ctx.set_guard_adaptive(adaptive)
        ctx.set_problem(B, whole.n_variants, G, v, cb, e, whole.v2snp)
        ctx.set_betas(whole.prior_betas())
        ctx.set_addition(None)
        ctx.probs_from_betas(0.01, fetch=False)
        ctx.estep(pen, with_doublets=False, fetch_logits=False, fetch_probs=False)
        hist = []
        for it in range(8):
            ctx.set_phase_timers(True); ctx.reset_timings()
            ctx.run_iterations(1, 0.01)
            ctx.synchronize()
            t = ctx.timings()
            hist.append((round(t['estep']['ms'], 3),) + tuple(round(x, 3) if isinstance(x, float) else x for x in ctx.guard_state()))
        print(B, 'adaptive' if adaptive else 'fixed', hist, flush=True)
        ctx.close()
