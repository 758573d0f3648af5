"""How many distinct genotype_prob values does a variant row hold (weighted by the calls that read it)?"""
import sys
import numpy as np
sys.path.insert(0, '.')
from demuxalot_amd import Demultiplexer, synth
from demuxalot_amd.device import DeviceContext
import bench

B, S, G, dp, seed = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else 'em_200k_100k_64']
problem = synth.generate(B, S, G, doublets=dp > 0, seed=seed, seed_calls=seed * 1000)
betas = problem.prior_betas(add_data_prior=False)
pen = Demultiplexer._doublet_penalties(G, dp)
ctx = DeviceContext(0)
ctx.set_problem(B, problem.n_variants, G, problem.variant_id, problem.compressed_cb, problem.p_base_wrong, problem.v2snp)
ctx.set_betas(betas)
ctx.set_addition(None)
per_variant = np.bincount(problem.variant_id, minlength=problem.n_variants).astype(np.float64)
rng = np.random.default_rng(0)
rows = rng.choice(problem.n_variants, size=20000, replace=False, p=per_variant / per_variant.sum())
for it in range(4):
    prob = ctx.probs_from_betas(0.01, fetch=True)
    d = np.array([len(np.unique(prob[r])) for r in rows])
    clipped = ((prob[rows] == np.float32(0.01)) | (prob[rows] == np.float32(0.99))).mean()
    print('iter', it, 'distinct per row (call-weighted sample): mean %.1f' % d.mean(), 'pct', np.percentile(d, [10, 50, 90]).tolist(),
          'clipped share %.3f' % clipped)
    ctx.estep(pen, with_doublets=dp > 0)
    ctx.mstep()
