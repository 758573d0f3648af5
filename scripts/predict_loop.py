"""GPU box helper: n predict passes (P-step + E-step on the table without beta addition) of a bench workload, for
rocprofv3 runs of the dictionary-form E-step.  usage: python3 scripts/predict_loop.py [workload] [passes] [never|auto]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from demuxalot_amd import Demultiplexer, synth
from demuxalot_amd.device import DeviceContext

workload = sys.argv[1] if len(sys.argv) > 1 else 'em_200k_100k_64'
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 5
mode = sys.argv[3] if len(sys.argv) > 3 else 'auto'
B, S, G, dp, seed = bench.WORKLOADS[workload]
p = synth.generate(B, S, G, doublets=dp > 0, seed=seed, seed_calls=seed * 1000)
pen = Demultiplexer._doublet_penalties(G, dp)
ctx = DeviceContext(0)
ctx.set_estep_dictionary(mode)
ctx.set_problem(B, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
ctx.set_betas(p.prior_betas(add_data_prior=False))
ctx.set_addition(None)
ctx.probs_from_betas(0.01, fetch=False)
ctx.estep(pen, with_doublets=dp > 0, fetch_logits=False, fetch_probs=False)
ctx.synchronize()
ctx.set_phase_timers(True); ctx.reset_timings()
t0 = time.perf_counter()
for _ in range(passes):
    ctx.probs_from_betas(0.01, fetch=False)
    ctx.estep(pen, with_doublets=dp > 0, fetch_logits=False, fetch_probs=False)
ctx.synchronize()
dt = (time.perf_counter() - t0) / passes
t = ctx.timings()
print(workload, mode, 'ablate', os.environ.get('DEMUXALOT_AMD_DICT_ABLATE', '0'), 'form', ctx.estep_form(), 'calls', p.n_calls,
      'pass ms %.3f' % (1e3 * dt), 'estep ms %.3f' % (t['estep']['ms'] / t['estep']['launches']), flush=True)
