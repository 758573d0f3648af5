"""GPU box: ONE learn_genotypes-shaped call from the prior - dmx_em(n iterations) on a freshly installed problem, no logits and no
addition fetched (what demuxalot_amd/demux.py: learn_genotypes makes) - timed on the host, several times (a fresh install each).
Under `rocprofv3 --kernel-trace` with COLD_TRACE=1 one call only: scripts/cold_call_trace.py prints its kernels and gaps.
    python3 scripts/cold_call.py [workload] [n_iterations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from demuxalot_amd import Demultiplexer, synth
from demuxalot_amd.device import DeviceContext
wl = sys.argv[1] if len(sys.argv) > 1 else 'em_200k_100k_64'
n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B, S, G, dp, seed = bench.WORKLOADS[wl]
cache = os.environ.get('DEMUXALOT_BENCH_PROBLEM')
p = bench.load_problem(cache) if cache and os.path.exists(os.path.join(cache, 'meta.json')) else synth.generate(B, S, G, doublets=dp > 0, seed=seed)
pen = Demultiplexer._doublet_penalties(G, dp)
betas = p.prior_betas(add_data_prior=False)
ctx = DeviceContext(0)
ctx.set_logits_needed(False)
rounds = 1 if os.environ.get('COLD_TRACE') else 4
for r in range(rounds):
    ctx.set_problem(B, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
    ctx.set_betas(betas); ctx.set_addition(None)
    ctx.synchronize()
    t = time.perf_counter()
    ctx.em(n_it, 0.01, pen, with_doublets=dp > 0, fetch_logits=False, fetch_probs=False, fetch_addition=False)
    ctx.synchronize()
    dt = 1e3 * (time.perf_counter() - t)
    lv = ctx.guard_levels()
    print(f'{wl} dmx_em({n_it}) from the prior: {dt:.3f} ms = {dt / n_it:.3f} ms per iteration; M-step form {ctx.mstep_form()}, tiles built {ctx.mstep_tiles_info()}, '
          f'coarse E-steps {lv["coarse_steps"]}, incremental {ctx.mstep_incremental()}', flush=True)
ctx.close()
