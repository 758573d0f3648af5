"""Summarises the rocprofv3 outputs of scripts/pmc_r1.sh into profiles/ (per-kernel means per launch)."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r1'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(os.path.join(root, 'gpurun_out', 'profiles_out'), exist_ok=True)
out_dir = os.path.join(root, 'gpurun_out', 'profiles_out')


def short(name):
    name = name.split('(')[0].replace('void ', '')
    return name.replace('dmx::', '')


lines = []
stats = glob.glob(os.path.join(root, 'gpurun_out', f'prof_{tag}_trace', '*', '*kernel_stats.csv')) or \
    glob.glob(os.path.join(root, 'gpurun_out', f'prof_{tag}_em_200k_100k_64', '*', '*kernel_stats.csv'))
if stats:
    lines.append('== rocprofv3 --kernel-trace --stats (python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline) ==')
    lines.append(open(stats[0]).read())
counters = collections.defaultdict(dict)
for d in sorted(glob.glob(os.path.join(root, 'gpurun_out', f'pmc_{tag}_*/'))):
    f = glob.glob(d + '/*/*counter_collection.csv')
    if not f:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f[0])):
        acc[short(row['Kernel_Name'])][row['Counter_Name']].append(float(row['Counter_Value']))
    for k, cs in acc.items():
        for c, v in cs.items():
            counters[k][c] = sum(v) / len(v)
lines.append('== rocprofv3 --pmc passes (separate runs; mean per launch) ==')
for k in sorted(counters):
    if k.startswith('__amd'):
        continue
    lines.append(k)
    for c in sorted(counters[k]):
        lines.append(f'    {c:32s} {counters[k][c]:.6g}')
traffic = {}
for k, cs in counters.items():
    if 'FETCH_SIZE' in cs and 'WRITE_SIZE' in cs and not k.startswith('__amd'):
        # rocprofv3 reports both in KiB. The reads here are 4-byte-per-lane row gathers, not the 16-byte-per-lane
        # streams for which the guide documents the 2x under-count, so FETCH_SIZE is taken as reported.
        key = k.split('<')[0]
        traffic[key] = dict(bytes_per_launch=(cs['FETCH_SIZE'] + cs['WRITE_SIZE']) * 1024.0,
                            fetch_bytes=cs['FETCH_SIZE'] * 1024.0, write_bytes=cs['WRITE_SIZE'] * 1024.0,
                            l2_hit_rate=cs.get('TCC_HIT_sum', 0) / max(1.0, cs.get('TCC_HIT_sum', 0) + cs.get('TCC_MISS_sum', 0)))
open(os.path.join(out_dir, f'{tag}_pmc_em_200k_100k_64.txt'), 'w').write('\n'.join(lines) + '\n')
json.dump({'em_200k_100k_64': traffic}, open(os.path.join(out_dir, 'pmc_traffic.json'), 'w'), indent=1)
print('\n'.join(lines[-60:]))
print(json.dumps(traffic, indent=1))
