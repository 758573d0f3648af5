"""The incremental M-step's sums built from nothing by the delta pass (dmx_set_mstep_incremental 2) against the full pass: same bits? how long?"""
import sys, time, numpy as np
sys.path.insert(0, '.')
from demuxalot_amd import synth
from demuxalot_amd.device import DeviceContext
B, S, G = 200000, 100000, 64
p = synth.generate(B, S, G, seed=1237)
pen = np.zeros(G, dtype=np.float32)
res = {}
for mode in (True, 'bootstrap'):
    ctx = DeviceContext(0); ctx.set_mstep_tiles(True); ctx.set_mstep_incremental(mode)
    ctx.set_problem(B, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp); ctx.set_betas(p.prior_betas()); ctx.set_addition(None)
    out = []
    for it in range(3):
        ctx.probs_from_betas(0.01, fetch=False); ctx.estep(pen, with_doublets=False, fetch_logits=False, fetch_probs=False)
        ctx.synchronize(); ctx.set_phase_timers(True); ctx.reset_timings(); a = ctx.mstep(2.); t = ctx.timings()
        out.append((a, t['mstep']['ms'], ctx.mstep_incremental()))
    res[mode] = out; ctx.close()
for it in range(3):
    a, b = res[True][it], res['bootstrap'][it]
    print(it, 'full-pass run', round(a[1], 3), a[2], '| bootstrap run', round(b[1], 3), b[2], '| same bits', np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)))
