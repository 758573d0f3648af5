"""GPU box helper: bench.e2e_timing on its own (fresh process), twice.  usage: python3 scripts/e2e_only.py [workload]"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from demuxalot_amd import synth
from demuxalot_amd.device import DeviceContext

workload = sys.argv[1] if len(sys.argv) > 1 else 'em_200k_100k_64'
B, S, G, dp, seed = bench.WORKLOADS[workload]
problem = synth.generate(B, S, G, doublets=dp > 0, seed=seed, seed_calls=seed * 1000)
t = time.perf_counter(); c = DeviceContext(0); print('first DeviceContext', round(time.perf_counter() - t, 3)); 
t = time.perf_counter(); c2 = DeviceContext(0); print('second DeviceContext', round(time.perf_counter() - t, 3)); c2.close(); c.close()
for rep in range(2):
    out = bench.e2e_timing(problem, dp)
    print(rep, json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in out.items() if k != 'note'}), flush=True)
