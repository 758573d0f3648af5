"""Consecutive 20-iteration calls of one EM run: per call the wall time per iteration, the E-steps that took the coarse pass, what the guards flagged."""
import sys
import time

sys.path.insert(0, '.')
import numpy as np

import bench
from demuxalot_amd import synth
from demuxalot_amd.device import DeviceContext

B, S, G, dp, seed = bench.WORKLOADS['em_200k_100k_64']
problem = synth.generate(B, S, G, doublets=dp > 0, seed=seed)
ctx = DeviceContext(0)
ctx.set_problem(problem.n_barcodes, problem.n_variants, G, problem.variant_id, problem.compressed_cb, problem.p_base_wrong, problem.v2snp)
ctx.set_betas(problem.prior_betas(add_data_prior=False))
ctx.set_addition(None)
ctx.probs_from_betas(0.01, fetch=False)
pen = np.zeros(G, dtype=np.float32)
ctx.estep(pen, with_doublets=False, fetch_logits=False, fetch_probs=False)
ctx.set_mstep_incremental(len(sys.argv) > 1 and sys.argv[1] == 'incremental')
ctx.set_msteps_expected(400)
lengths = [x if x.startswith('e') else int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else [20] * (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
done = 0
for call, n_it in enumerate(lengths):
    if isinstance(n_it, str):  # e<n>: n explicit E-steps (logits kept: the fine pass)
        for _ in range(int(n_it[1:])):
            ctx.estep(pen, with_doublets=False, fetch_logits=False, fetch_probs=False)
        print(n_it[1:], 'explicit E-steps', flush=True)
        continue
    ctx.synchronize()
    ctx.reset_timings()
    t0 = time.perf_counter()
    ctx.run_iterations(n_it, 0.01)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    lv = ctx.guard_levels()
    done += n_it
    print(f'iterations {done - n_it + 1:3d}-{done:3d}: {dt / n_it * 1e3:.4f} ms per iteration; coarse E-steps {lv["coarse_steps"]:2d}; last level {lv["level"]}; '
          f'passes coarse {lv["coarse_pass_ms"]:.3f} fine {lv["fine_pass_ms"]:.3f} exact {lv["exact_pass_ms"]:.3f} ms; flagged by the fine / coarse guard '
          f'{lv["flagged_fine"]} / {lv["flagged_coarse"]}; redone {ctx.guard_stats()[1]}', flush=True)
