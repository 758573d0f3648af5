"""Randomised check of the tolerance-mode E-step against the exact mode on the GPU (no oracle needed):
python scripts/fast_mode_sweep.py [n_trials] [first_seed].  Every trial: random shape (singlets / doublets, all
kernel forms), one E-step in both modes on the same resident problem, tests/test_gpu_fast_mode.check_contract."""
import os
import sys
import time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from demuxalot_amd import Demultiplexer, synth
from demuxalot_amd.device import get_context
from tests.test_gpu_fast_mode import check_contract

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(99 + first)
worst = (0.0, 0.0)
t0 = time.time()
for trial in range(n):
    doublets = rng.random() < 0.5
    G = int(rng.choice([2, 3, 5, 8, 12, 16, 23, 24, 31, 32, 33, 45, 46, 64, 70, 91, 128] if doublets else
                       [2, 4, 7, 16, 31, 32, 33, 64, 65, 100, 128, 200, 300, 600]))
    K = G * (G + 1) // 2 if doublets else G
    B = int(rng.integers(100, 3000 if K < 600 else 400))
    S = int(rng.integers(100, 2000))
    cpb = int(rng.choice([20, 60, 150, 400]))
    dp = float(rng.choice([0.1, 0.3, 0.5])) if doublets else 0.
    p = synth.generate(B, S, G, calls_per_barcode=min(cpb, S), doublets=doublets, seed=5000 + first + trial)
    ctx = get_context()
    ctx.set_problem(p.n_barcodes, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
    ctx.set_betas(p.prior_betas())
    ctx.set_addition(None)
    ctx.probs_from_betas(float(rng.choice([0.01, 0.0, 0.05])), fetch=False)
    pen = Demultiplexer._doublet_penalties(G, dp)
    try:
        logits_e, probs_e = ctx.estep(pen, with_doublets=doublets)
        ctx.set_estep_mode('fast')
        logits_f, probs_f = ctx.estep(pen, with_doublets=doublets)
    finally:
        ctx.set_estep_mode('exact')
    what = f'trial {trial}: G={G} K={K} B={B} S={S} cpb={cpb} dp={dp}'
    ulps, dev = check_contract(logits_f, probs_f, logits_e.astype(np.float64), probs_e.astype(np.float64), what, strict=False)
    worst = max(worst, (dev, ulps))
    print('ok', what, f'posterior dev {dev:.2e}, logits {ulps:.1f} ulp', flush=True)
print(f'{n} trials ok in {time.time() - t0:.0f} s; worst posterior deviation {worst[0]:.3g}')
