"""Distribution of non-zero float32 posteriors per barcode in the bench workload (M-step sparsity)."""
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
for it in range(3):
    ctx.probs_from_betas(0.01, fetch=False)
    logits, post = ctx.estep(pen, with_doublets=dp > 0)
    nnz = (post[:, :G] != 0).sum(1)
    print('iter', it, 'calls', problem.n_calls, 'nnz per barcode: mean %.2f' % nnz.mean(),
          'pct', np.percentile(nnz, [1, 10, 50, 90, 99]).tolist(),
          'share<=4: %.3f  <=8: %.3f' % ((nnz <= 4).mean(), (nnz <= 8).mean()))
    ctx.mstep()
    if it == 1:
        pb = np.bincount(problem.compressed_cb, minlength=B)
        for thr in (0.0, 2.0 ** -80):
            nn = (post[:, :G] > thr).sum(1)
            hist = np.bincount(np.minimum(nn, 65), weights=pb, minlength=66) / pb.sum()
            print('thr', thr, 'call-weighted: nnz==1 %.3f, 2..4 %.3f, 5..16 %.3f, 17..63 %.3f, 64 %.3f, mean %.2f' % (
                hist[1], hist[2:5].sum(), hist[5:17].sum(), hist[17:64].sum(), hist[64], (pb * nn).sum() / pb.sum()))
        per_bc = np.bincount(problem.compressed_cb, minlength=B)
        print('share of CALLS from barcodes with nnz>4: %.3f' % (per_bc[nnz > 4].sum() / per_bc.sum()),
              ' nnz==64: %.3f' % (per_bc[nnz == G].sum() / per_bc.sum()),
              ' mean nnz weighted by calls: %.2f' % ((per_bc * nnz).sum() / per_bc.sum()))
