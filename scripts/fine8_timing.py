"""E-steps that keep their logits at 200k x 100k x 64: the fine pass on the tile-major stream (default) against the fine pass on the coarse
pass's records (dmx_set_lean_memory: k_estep_tiled_fine8) and the coarse pass itself; ms per E-step from the phase timers."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from demuxalot_amd import synth  # noqa: E402
from demuxalot_amd.device import DeviceContext  # noqa: E402

B, S, G, dp, seed = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else 'em_200k_100k_64']
p = synth.generate(B, S, G, doublets=False, seed=seed)
pen = np.zeros(G, dtype=np.float32)
for lean in (False, True):
    ctx = DeviceContext(0)
    try:
        ctx.set_estep_mode('guarded')
        ctx.set_lean_memory(lean)
        ctx.set_problem(p.n_barcodes, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
        ctx.set_betas(p.prior_betas(add_data_prior=False))
        ctx.set_logits_needed(False)
        ctx.em(6, 0.01, pen, with_doublets=False, fetch_logits=False, fetch_probs=False, fetch_addition=False)
        ctx.set_logits_needed(True)
        for kind, coarse in (('kept logits', True), ('coarse (always)', 'always')):
            ctx.set_coarse_pass(coarse)
            for _ in range(5):
                ctx.estep(pen, with_doublets=False, fetch_logits=False, fetch_probs=False)
            ctx.synchronize()
            ctx.set_phase_timers(True); ctx.reset_timings()
            t0 = time.perf_counter()
            for _ in range(20):
                ctx.estep(pen, with_doublets=False, fetch_logits=False, fetch_probs=False)
            ctx.synchronize()
            wall = (time.perf_counter() - t0) / 20
            t = ctx.timings()['estep']
            ctx.set_phase_timers(False)
            print(f"lean={lean} {kind}: estep {t['ms'] / max(1, t['launches']):.3f} ms (wall {1e3 * wall:.3f}), levels {ctx.guard_levels()}, redone {ctx.guard_stats()[0]}, device bytes per call {ctx.device_bytes() / len(p.variant_id):.1f}")
    finally:
        ctx.close()
