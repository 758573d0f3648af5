"""GPU box helper: how long the device pack of the bench workload takes on re-used, fresh and coexisting contexts.
usage: [DEMUXALOT_AMD_CACHE_GB=0] python3 scripts/alloc_probe.py"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from demuxalot_amd import synth, demux
from demuxalot_amd.device import DeviceContext

B, S, G, dp, seed = bench.WORKLOADS['em_200k_100k_64']
problem = synth.generate(B, S, G, doublets=False, seed=seed, seed_calls=seed * 1000)
calls, genotypes, handler = synth.as_objects(problem)


def pack(ctx):
    t = time.perf_counter()
    demux._pack_on_device(calls, genotypes, handler.n_barcodes, False, fetch_betas=False, ctx=ctx)
    ctx.synchronize()
    return round(1e3 * (time.perf_counter() - t), 1)


print('cache GB', os.environ.get('DEMUXALOT_AMD_CACHE_GB', 'default'))
a = DeviceContext(0)
print('A: same context x4', [pack(a) for _ in range(4)], flush=True)
b = DeviceContext(0)
print('B: second context while A is open x3', [pack(b) for _ in range(3)], flush=True)
a.close()
print('   after closing A: B again x2', [pack(b) for _ in range(2)], flush=True)
b.close()
times = []
for _ in range(3):
    c = DeviceContext(0)
    times.append(pack(c))
    c.close()
print('C: fresh context each time (previous closed) x3', times, flush=True)
