"""How much of the genotype table changes from one EM iteration to the next - as float32 and as the binary16 the coarse pass reads:
what an incremental E-step would have to touch.  GPU box: python3 scripts/table_change_probe.py [iterations]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from demuxalot_amd import synth
from demuxalot_amd.device import DeviceContext

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 12
B, S, G = 200_000, 100_000, 64
p = synth.generate(B, S, G, seed=1237)
pen = np.zeros(G, dtype=np.float32)
ctx = DeviceContext(0)
ctx.set_mstep_tiles(True)
ctx.set_problem(B, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
ctx.set_betas(p.prior_betas(add_data_prior=False))
ctx.set_addition(None)
calls_per_variant = np.bincount(p.variant_id, minlength=p.n_variants)
prev32 = prev16 = None
for it in range(iters):
    table = ctx.probs_from_betas(0.01).copy()
    half = table.astype(np.float16)
    if prev32 is not None:
        rows32 = (table.view(np.uint32) != prev32.view(np.uint32)).any(axis=1)
        rows16 = (half.view(np.uint16) != prev16.view(np.uint16)).any(axis=1)
        ent16 = (half.view(np.uint16) != prev16.view(np.uint16)).mean()
        print(f'iteration {it}: rows changed as float32 {rows32.mean() * 100:.2f} % ({calls_per_variant[rows32].sum() / calls_per_variant.sum() * 100:.2f} % of the calls), '
              f'as binary16 {rows16.mean() * 100:.3f} % ({calls_per_variant[rows16].sum() / calls_per_variant.sum() * 100:.3f} % of the calls; {ent16 * 100:.4f} % of the entries)')
    prev32, prev16 = table, half
    ctx.estep(pen, with_doublets=False, fetch_logits=False, fetch_probs=False)
    ctx.mstep(2., fetch=False)
ctx.close()
