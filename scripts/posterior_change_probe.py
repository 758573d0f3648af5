"""How many barcodes change their posterior row (bitwise) from one EM iteration to the next, and how many calls they hold:
what an incremental M-step would have to touch.  GPU box: python3 scripts/posterior_change_probe.py [workload] [iterations]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from demuxalot_amd import synth
from demuxalot_amd.device import DeviceContext

workload = sys.argv[1] if len(sys.argv) > 1 else 'em_200k_100k_64'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 8
B, S, G, dp, seed = bench.WORKLOADS[workload]
p = synth.generate(B, S, G, doublets=dp > 0, seed=seed)
pen = np.zeros(G, dtype=np.float32)
n_b = np.bincount(p.compressed_cb, minlength=B)
ctx = DeviceContext(0)
ctx.set_problem(B, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
ctx.set_betas(p.prior_betas(add_data_prior=False))
ctx.set_addition(None)
prev = None
for it in range(iters):
    ctx.probs_from_betas(0.01, fetch=False)
    _l, post = ctx.estep(pen, with_doublets=False, fetch_logits=False)
    if prev is not None:
        changed = (post.view(np.uint32) != prev.view(np.uint32)).any(axis=1)
        live = (post > 1e-24).sum(axis=1)
        grid = 2.0 ** -51  # contributions below the fixed-point grid of the tile-major M-step add nothing
        sq_new, sq_old = (post.astype(np.float64)) ** 2, (prev.astype(np.float64)) ** 2
        matters = (np.abs(sq_new - sq_old) > grid).any(axis=1)
        print(f'iteration {it}: {changed.mean() * 100:.2f} % of the barcodes changed a posterior bit ({n_b[changed].sum() / n_b.sum() * 100:.2f} % of the calls); '
              f'{matters.mean() * 100:.2f} % by more than the M-step\'s grid ({n_b[matters].sum() / n_b.sum() * 100:.2f} % of the calls); '
              f'barcodes with one live posterior {np.mean(live == 1) * 100:.1f} %, exactly 1.0: {np.mean(post.max(axis=1) == 1.0) * 100:.1f} %')
    prev = post
    ctx.mstep(2., fetch=False)
ctx.close()
