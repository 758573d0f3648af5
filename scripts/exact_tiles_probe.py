"""GPU box: would the EXACT additions pay on the tile-major form?  Exact E-step; the M-step as exact additions (work items + in-order redo:
the reference's bits) and as the tile-major fixed-point sums; how many [V, G] entries differ, and how many an ambiguity rule
(no float32 rounding boundary within n 2^-(s+1) + 4 n u S of the fixed-point sum; zero sums of variants with calls: always ambiguous)
would send to the in-order redo.  python3 scripts/exact_tiles_probe.py [workload] [iterations]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from demuxalot_amd import Demultiplexer, synth
from demuxalot_amd.device import DeviceContext
wl = sys.argv[1] if len(sys.argv) > 1 else 'em_200k_100k_64'
n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B, S, G, dp, seed = bench.WORKLOADS[wl]
p = synth.generate(B, S, G, doublets=dp > 0, seed=seed)
pen = Demultiplexer._doublet_penalties(G, dp)
n_v = np.bincount(p.variant_id, minlength=p.n_variants).astype(np.float64)[:, None]
shift = np.minimum(50, 62 - np.ceil(np.log2(n_v + 1)))   # per variant (the tile's is the minimum over its variants: at most this)
ctx = DeviceContext(0)
ctx.set_estep_mode('exact')
ctx.set_problem(B, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
ctx.set_betas(p.prior_betas()); ctx.set_addition(None)
for it in range(n_it):
    ctx.probs_from_betas(0.01, fetch=False)
    ctx.estep(pen, with_doublets=dp > 0, fetch_logits=False, fetch_probs=False)
    ctx.set_exact_additions(False); ctx.set_mstep_tiles('always'); ctx.set_mstep_incremental(False)
    tiles = ctx.mstep(2.0).astype(np.float64)
    ctx.set_exact_additions(True); ctx.set_mstep_tiles('never')
    exact = ctx.mstep(2.0)          # (the last M-step of the iteration: the EM goes on with the reference's addition)
    e64 = exact.astype(np.float64)
    differ = tiles != e64
    f = tiles.astype(np.float32)
    lo = 0.5 * (tiles + np.nextafter(f, np.float32(0)).astype(np.float64))
    hi = 0.5 * (tiles + np.nextafter(f, np.float32(np.inf)).astype(np.float64))
    bound = n_v * 2.0 ** -(shift + 1) + 4 * n_v * 2.0 ** -53 * tiles
    ambiguous = ~((tiles - bound > lo) & (tiles + bound < hi)) & (n_v > 0)
    zero_amb = ambiguous & (tiles == 0)
    missed = differ & ~ambiguous
    print(f'iteration {it}: entries {tiles.size}; differ from the exact additions {int(differ.sum())} ({differ.mean():.2e}); rule flags {int(ambiguous.sum())} '
          f'({ambiguous.mean():.2e}), of them zero sums {int(zero_amb.sum())}; differing entries the rule misses {int(missed.sum())}; '
          f'flagged with exact >= 1e-3: {int((ambiguous & (e64 >= 1e-3)).sum())}; exact zeros among zero sums: {int(((e64 == 0) & (tiles == 0) & (n_v > 0)).sum())}', flush=True)
ctx.close()
