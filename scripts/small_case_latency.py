import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np
from tests import fixture_io as fio
from demuxalot_amd import Demultiplexer
fx = fio.load('f1_synthetic_default.npz'); calls, g, h = fio.product_inputs(fx)
Demultiplexer.predict_posteriors(calls, g, h)
for name, fn in (('predict dp=.35', lambda: Demultiplexer.predict_posteriors(calls, g, h)), ('predict dp=0', lambda: Demultiplexer.predict_posteriors(calls, g, h, doublet_prior=0.)), ('learn 5 it', lambda: Demultiplexer.learn_genotypes(calls, g, h))):
    t=time.perf_counter(); [fn() for _ in range(5)]; print(name, (time.perf_counter()-t)/5*1e3, 'ms')
