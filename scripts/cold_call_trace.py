"""Kernels and gaps of the dmx_em call of scripts/cold_call.py out of a rocprofv3 --kernel-trace csv: everything from the first
P-step kernel after the install to the end, one line per launch, and totals per kernel name."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
first = max(i for i, r in enumerate(rows) if 'k_probs_from_betas' in r['Kernel_Name'] and (i == 0 or 'rocprim' in rows[i - 1]['Kernel_Name'] or True))
# the call starts at the first k_probs_from_betas that follows the install's last kernel: take the earliest P-step launch of the last burst
starts = [i for i, r in enumerate(rows) if 'k_probs_from_betas' in r['Kernel_Name']]
n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 5
a = starts[-n_it]
t0, prev_end, totals = int(rows[a]['Start_Timestamp']), None, {}
for r in rows[a:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('void dmx::', '').replace('(anonymous namespace)::', '').split('(')[0][:70]
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(s - t0) / 1e3:9.1f} gap {gap:7.1f} dur {(e - s) / 1e3:8.1f} us  {name}")
    totals.setdefault(name, [0, 0.0])
    totals[name][0] += 1; totals[name][1] += (e - s) / 1e3
    totals.setdefault('(gaps)', [0, 0.0])[1] += max(0.0, gap)
    prev_end = e
print('# totals, us')
for name, (n, us) in sorted(totals.items(), key=lambda kv: -kv[1][1]):
    print(f'{us:10.1f} {n:4d}  {name}')
print(f'# call: {(prev_end - t0) / 1e3:.1f} us')
