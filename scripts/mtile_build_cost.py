"""GPU box: what the first tile-major M-step costs (it builds the tile records) against the later ones and the work-item form."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from demuxalot_amd import Demultiplexer, synth
from demuxalot_amd.device import DeviceContext
wl = sys.argv[1] if len(sys.argv) > 1 else 'em_200k_100k_64'
B, S, G, dp, seed = bench.WORKLOADS[wl]
p = synth.generate(B, S, G, doublets=dp > 0, seed=seed)
pen = Demultiplexer._doublet_penalties(G, dp)
for tiles in (True, False):
    ctx = DeviceContext(0)
    ctx.set_mstep_tiles(tiles)
    ctx.set_problem(B, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
    ctx.set_betas(p.prior_betas(add_data_prior=False)); ctx.set_addition(None)
    ctx.probs_from_betas(0.01, fetch=False)
    ctx.estep(pen, with_doublets=dp > 0, fetch_logits=False, fetch_probs=False)
    ctx.synchronize()
    times = []
    for _ in range(4):
        t = time.perf_counter(); ctx.mstep(2.0, fetch=False); ctx.synchronize(); times.append(1e3 * (time.perf_counter() - t))
    print(wl, 'tiles' if tiles else 'items', ctx.mstep_form(), 'M-step wall ms:', [round(x, 3) for x in times], flush=True)
    ctx.close()
