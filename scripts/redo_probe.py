import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
import bench
from demuxalot_amd import Demultiplexer, synth
from demuxalot_amd.device import DeviceContext
from demuxalot_amd.distributed import partition_barcodes
B, S, G, dp, seed = bench.WORKLOADS['em_200k_100k_64']
whole = synth.generate(B, S, G, doublets=False, seed=seed)
betas = whole.prior_betas(add_data_prior=False)
pen = Demultiplexer._doublet_penalties(G, dp)
counts = np.bincount(whole.compressed_cb, minlength=B)
for n, exch in ((1, ''), (8, ''), (8, 'reduce_scatter'), (2, '')):
    os.environ['DEMUXALOT_AMD_EXCHANGE'] = exch
    if n > 1:
        b = partition_barcodes(counts, n); lo, hi = int(b[0]), int(b[1])
        v, cb, e = whole.subset_barcodes(lo, hi)
        p = synth.SyntheticProblem(hi - lo, S, G, whole.v2snp, whole.raw_betas, v, cb, e, whole.truth[lo:hi])
    else:
        p = whole
    ctx = DeviceContext(0)
    ctx.set_estep_mode('exact'); ctx.set_exact_additions(True)
    if n > 1: ctx.comm_init_emulated(0, n, 50.0, 10.0, reduce_dtype='f32')
    ctx.set_problem(p.n_barcodes, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
    ctx.set_betas(betas); ctx.set_addition(None); ctx.probs_from_betas(0.01, fetch=False)
    ctx.estep(pen, with_doublets=False, fetch_logits=False, fetch_probs=False)
    counts_it = []
    for _ in range(4):
        ctx.run_iterations(1, 0.01); ctx.synchronize()
        counts_it.append(ctx.redo_count())
    print('n', n, exch or 'auto', ctx.exchange_mode(), 'redo_count per iteration', counts_it, flush=True)
    ctx.close()
