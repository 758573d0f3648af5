import sys, numpy as np
sys.path.insert(0, '.')
from demuxalot_amd import synth
from demuxalot_amd.device import DeviceContext
p = synth.generate(200_000, 100_000, 64, seed=1237)
ctx = DeviceContext(0)
ctx.set_problem(p.n_barcodes, p.n_variants, 64, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
ctx.set_betas(p.prior_betas(add_data_prior=False)); ctx.set_addition(None)
pen = np.zeros(64, dtype=np.float32)
for it in range(4):
    ctx.probs_from_betas(0.01, fetch=False)
    ctx.estep(pen, with_doublets=False, fetch_logits=False, fetch_probs=False)
    ctx.mstep(2., fetch=False)
    print('iteration', it, 'sums redone', ctx.redo_count())
