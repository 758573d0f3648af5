"""Coarse pass of the guarded E-step (k_estep_tiled_coarse) against the fine pass and the exact mode on one workload:
posteriors of EVERY barcode (contract: within 1e-5, same arg-max), queued fractions, E-step times.
GPU box: python3 scripts/coarse_probe.py [workload] [iterations]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from demuxalot_amd import synth
from demuxalot_amd.device import DeviceContext
from oracle import demux_oracle  # noqa: F401  (penalties only)

workload = sys.argv[1] if len(sys.argv) > 1 else 'em_200k_100k_64'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
B, S, G, dp, seed = bench.WORKLOADS[workload]
extra = {}
if os.environ.get('PROBE_CALLS'):
    extra['calls_per_barcode'] = int(os.environ['PROBE_CALLS'])
if os.environ.get('PROBE_SIBLINGS'):
    extra['sibling_pairs'] = True
pdir = os.environ.get('DEMUXALOT_BENCH_PROBLEM')
if pdir and os.path.exists(pdir) and not extra:
    problem = bench.load_problem(pdir)
else:
    problem = synth.generate(B, S, G, doublets=dp > 0, seed=seed, **extra)
betas = problem.prior_betas(add_data_prior=False)
pen = demux_oracle.doublet_penalties(G, dp)
V = problem.n_variants


def run(mode, coarse):
    ctx = DeviceContext(0)
    ctx.set_estep_mode(mode)
    ctx.set_coarse_pass(coarse)
    ctx.set_problem(problem.n_barcodes, V, G, problem.variant_id, problem.compressed_cb, problem.p_base_wrong, problem.v2snp)
    ctx.set_betas(betas)
    ctx.set_addition(None)
    ctx.probs_from_betas(0.01, fetch=False)
    ctx.estep(pen, with_doublets=dp > 0, fetch_logits=False, fetch_probs=False)
    ctx.set_msteps_expected(iters + 12)
    ctx.run_iterations(iters, 0.01)
    ctx.synchronize()
    ctx.set_phase_timers(True); ctx.reset_timings()
    ctx.run_iterations(10, 0.01)
    ctx.synchronize()
    t = ctx.timings()
    ms = {k: v['ms'] / max(1, v['launches']) for k, v in t.items()}
    post = ctx.get_block('probs', 0, problem.n_barcodes)
    logits = ctx.get_block('logits', 0, problem.n_barcodes)
    stats = ctx.guard_stats()
    state = ctx.guard_state()
    ctx.close() if hasattr(ctx, 'close') else None
    return post, logits, ms, stats, state


ref_post, ref_logits, ms_exact, _, _ = run('exact', False)
print('exact    estep %.3f ms  mstep %.3f' % (ms_exact['estep'], ms_exact['mstep']))
for name, coarse in (('fine', False), ('coarse', True)):
    post, logits, ms, stats, state = run('guarded', coarse)
    d = np.abs(post - ref_post).max(axis=1)
    same = (post.argmax(axis=1) == ref_post.argmax(axis=1))
    print('%-8s estep %.3f ms  mstep %.3f  max |dp| %.3g  barcodes beyond 1e-5: %d  arg-max differs: %d  max |dlogit| %.3g  guard %s state %s' % (
        name, ms['estep'], ms['mstep'], d.max(), int((d > 1e-5).sum()), int((~same).sum()), np.abs(logits - ref_logits).max(), stats, state))
