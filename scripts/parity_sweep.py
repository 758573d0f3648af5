"""Wider randomised parity sweep than the test suite runs (GPU box): python scripts/parity_sweep.py [n_seeds] [n_trials] [first_seed] [n_doublet_trials]
Reuses the suite's own randomised EM check with more seeds, then mid-size EM problems with random G, density and
iteration counts against the numpy oracle (everything bitwise)."""
import os
import sys
import time
import numpy as np
os.environ.setdefault('DEMUXALOT_AMD_ESTEP', 'exact')  # bitwise comparisons: the exact mode (the library's default is the guarded one)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, 'tests'))
import subprocess
subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle")])
import fixture_io as fio
import test_gpu_parity as T
from oracle import demux_oracle as oracle

oracle.load_npsimd()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
t0 = time.time()
first = int(sys.argv[3]) if len(sys.argv) > 3 else 10
for seed in range(first, first + n):
    print('seed', seed, flush=True)
    T.test_randomised_em_against_oracle(oracle, seed)
print(f'{n} randomised EM problems ok ({time.time() - t0:.0f} s)', flush=True)

from demuxalot_amd import synth
from demuxalot_amd.device import get_context
rng = np.random.default_rng(5 + first)
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    G = int(rng.choice([2, 3, 4, 6, 8, 12, 16, 24, 32, 33, 48, 63, 64, 65, 96, 128]))
    B = int(rng.integers(500, 6000))
    S = int(rng.integers(200, 3000))
    cpb = int(rng.choice([20, 60, 150, 400, 900]))
    n_it = int(rng.integers(1, 4))
    clip = float(rng.choice([0.01, 0.0, 0.05]))
    p = synth.generate(B, S, G, calls_per_barcode=min(cpb, S), seed=1000 + first + trial)
    betas = p.prior_betas()
    ctx = get_context()
    ctx.set_problem(p.n_barcodes, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
    ctx.set_betas(betas)
    pen = np.zeros(G, dtype=np.float32)
    logits, probs, addition = ctx.em(n_it, clip, pen, with_doublets=False)
    packed = dict(variant_id=p.variant_id, compressed_cb=p.compressed_cb, p_base_wrong=p.p_base_wrong, betas=betas, v2snp=p.v2snp)
    hist = oracle.em(packed, p.n_barcodes, n_it, clip, 0., impl='npsimd')
    what = f'trial {trial}: G={G} B={B} S={S} cpb={cpb} it={n_it} clip={clip} N={p.n_calls}'
    T.check_posteriors(logits, probs, hist[-1]['logits'], hist[-1]['probs'], what)
    fio.assert_bitwise(addition, hist[-1]['addition'], what + ' addition')
    print('ok', what, flush=True)
print(f'sweep done in {time.time() - t0:.0f} s')

# doublet tables across the kernel forms (lane-per-option up to 512 options, workgroup-per-barcode tiles beyond)
from demuxalot_amd import Demultiplexer
n_doublet = int(sys.argv[4]) if len(sys.argv) > 4 else 16
for trial in range(n_doublet):
    G = int(rng.choice([5, 9, 16, 22, 23, 31, 32, 33, 40, 45, 46, 52, 64, 70, 90, 91]))
    B = int(rng.integers(60, 400))
    S = int(rng.integers(100, 600))
    cpb = int(rng.choice([20, 50, 90]))
    dp = float(rng.choice([0.1, 0.3, 0.5]))
    n_it = int(rng.integers(1, 3))
    p = synth.generate(B, S, G, calls_per_barcode=min(cpb, S), doublets=True, seed=3000 + first + trial)
    betas = p.prior_betas()
    ctx = get_context()
    ctx.set_problem(p.n_barcodes, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
    ctx.set_betas(betas)
    pen = Demultiplexer._doublet_penalties(G, dp)
    logits, probs, addition = ctx.em(n_it, 0.01, pen, with_doublets=True)
    packed = dict(variant_id=p.variant_id, compressed_cb=p.compressed_cb, p_base_wrong=p.p_base_wrong, betas=betas, v2snp=p.v2snp)
    hist = oracle.em(packed, p.n_barcodes, n_it, 0.01, dp, impl='npsimd')
    what = f'doublets {trial}: G={G} K={G * (G + 1) // 2} B={B} S={S} cpb={cpb} it={n_it} dp={dp} N={p.n_calls}'
    T.check_posteriors(logits, probs, hist[-1]['logits'], hist[-1]['probs'], what)
    fio.assert_bitwise(addition, hist[-1]['addition'], what + ' addition')
    print('ok', what, flush=True)
print(f'all done in {time.time() - t0:.0f} s')
