"""GPU box: the end-to-end part of bench.py alone (the drop-in calls on the containers of the whole workload)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from demuxalot_amd import synth
wl = sys.argv[1] if len(sys.argv) > 1 else 'em_200k_100k_64'
B, S, G, dp, seed = bench.WORKLOADS[wl]
p = synth.generate(B, S, G, doublets=dp > 0, seed=seed)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2):
    out = bench.e2e_timing(p, dp)
    print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in out.items() if k != 'note'}), flush=True)
