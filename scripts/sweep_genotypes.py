"""E/M kernel time across genotype counts (all lane-group paths), 50k barcodes x 20k SNPs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from demuxalot_amd import Demultiplexer, synth
from demuxalot_amd.device import get_context

ctx = get_context()
for G, dp in [(4, 0.), (8, 0.), (8, .35), (16, 0.), (16, .3), (24, 0.), (32, 0.), (48, 0.), (64, 0.), (100, 0.), (128, 0.), (200, 0.), (22, .3), (23, .3)]:
    p = synth.generate(50000, 20000, G, doublets=dp > 0, seed=3)
    betas = p.prior_betas(False)
    ctx.set_problem(p.n_barcodes, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
    ctx.set_betas(betas)
    pen = Demultiplexer._doublet_penalties(G, dp)
    ctx.set_addition(None); ctx.probs_from_betas(0.01, fetch=False)
    ctx.estep(pen, with_doublets=dp > 0, fetch_logits=False, fetch_probs=False)
    ctx.run_iterations(2, 0.01); ctx.synchronize(); ctx.set_phase_timers(True); ctx.reset_timings()
    ctx.run_iterations(5, 0.01); ctx.synchronize()
    t = ctx.timings()
    e = t['estep']['ms'] / t['estep']['launches']; m = t['mstep']['ms'] / t['mstep']['launches']
    K = len(pen)
    print(f'G={G:4d} K={K:5d} N={p.n_calls:9d}  E {e:7.3f} ms ({p.n_calls * K / e / 1e9:7.1f} G terms/s)   M {m:7.3f} ms ({p.n_calls * G / m / 1e9:7.1f} G lane-terms/s)')
