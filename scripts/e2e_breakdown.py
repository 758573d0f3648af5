"""GPU box helper: where the wall time of the drop-in predict_posteriors goes before the kernels start (bench workload).
usage: python3 scripts/e2e_breakdown.py [workload]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from demuxalot_amd import synth, demux
from demuxalot_amd.device import DeviceContext

workload = sys.argv[1] if len(sys.argv) > 1 else 'em_200k_100k_64'
B, S, G, dp, seed = bench.WORKLOADS[workload]
problem = synth.generate(B, S, G, doublets=dp > 0, seed=seed, seed_calls=seed * 1000)
calls, genotypes, handler = synth.as_objects(problem)
ctx = DeviceContext(0)
for rep in range(3):
    t = [time.perf_counter()]
    v2snp = genotypes.get_snp_ids_for_variants(); t.append(time.perf_counter())
    (var_chrom, var_pos, var_base), chrom_index = demux._variant_keys(genotypes); t.append(time.perf_counter())
    parts = [(chrom_index[ch], c.snp_calls[:c.n_snp_calls], c.molecules[:c.n_molecules]) for ch, c in calls.items()]
    t.append(time.perf_counter())
    ctx.pack_containers_and_set_problem(handler.n_barcodes, genotypes.n_genotypes, var_chrom, var_pos, var_base, v2snp, parts)
    ctx.synchronize(); t.append(time.perf_counter())
    betas = genotypes.get_betas(); t.append(time.perf_counter())
    ctx.set_prior_betas(betas, genotypes.default_prior, True, fetch=False)
    ctx.synchronize(); t.append(time.perf_counter())
    names = ['get_snp_ids_for_variants', '_variant_keys', 'slices', 'pack_containers_and_set_problem', 'get_betas', 'set_prior_betas']
    print(rep, {n: round(1e3 * (b - a), 2) for n, a, b in zip(names, t, t[1:])}, 'ms', flush=True)
nbytes = sum(c.snp_calls[:c.n_snp_calls].nbytes + c.molecules[:c.n_molecules].nbytes for c in calls.values())
print('record bytes', nbytes, 'timings', {k: v for k, v in ctx.timings().items() if v['launches']})
