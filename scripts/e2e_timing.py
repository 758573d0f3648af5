"""End-to-end wall time of the Python entry points (host flattening + device pack + EM + DataFrames)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from demuxalot_amd import Demultiplexer, synth
from demuxalot_amd.demux import _flatten_inputs, _pack_on_device, _prior_betas

sizes = ((20000, 20000, 8, 0.35), (50000, 50000, 32, 0.0))
if len(sys.argv) > 1 and sys.argv[1] == 'big':
    sizes = ((200000, 100000, 64, 0.0),)
for (B, S, G, dp) in sizes:
    p = synth.generate(B, S, G, doublets=dp > 0, seed=7)
    calls, genotypes, handler = synth.as_objects(p)
    Demultiplexer.predict_posteriors(calls, genotypes, handler, doublet_prior=dp)  # warm-up (context, code objects)
    t = time.perf_counter(); _flatten_inputs(calls, genotypes, False); t_flat = time.perf_counter() - t
    t = time.perf_counter(); ctx, betas = _pack_on_device(calls, genotypes, handler.n_barcodes, True); t_pack = time.perf_counter() - t
    t = time.perf_counter(); logits, probs = Demultiplexer.predict_posteriors(calls, genotypes, handler, doublet_prior=dp); t_pred = time.perf_counter() - t
    t = time.perf_counter(); learnt, probs2 = Demultiplexer.learn_genotypes(calls, genotypes, handler, n_iterations=5, doublet_prior=0.); t_learn = time.perf_counter() - t
    print(f'B={B} S={S} G={G} dp={dp} calls={p.n_calls}: flatten {t_flat:.3f}s, flatten+device pack+prior {t_pack:.3f}s, '
          f'predict_posteriors {t_pred:.3f}s, learn_genotypes(5 it) {t_learn:.3f}s')
