"""Where the wall time of learn_genotypes (5 iterations, packed problem resident after predict_posteriors) goes: cProfile, top entries."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demuxalot_amd import Demultiplexer, synth

p = synth.generate(200_000, 100_000, 64, seed=1237)
calls, genotypes, handler = synth.as_objects(p)
Demultiplexer.predict_posteriors(calls, genotypes, handler, doublet_prior=0.)
Demultiplexer.learn_genotypes(calls, genotypes, handler, n_iterations=5)
for _ in range(2):
    t = time.perf_counter()
    Demultiplexer.learn_genotypes(calls, genotypes, handler, n_iterations=5)
    print('learn_genotypes(5) on resident inputs:', round((time.perf_counter() - t) * 1e3, 2), 'ms')
pr = cProfile.Profile()
pr.enable()
Demultiplexer.learn_genotypes(calls, genotypes, handler, n_iterations=5)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
