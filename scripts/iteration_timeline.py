"""One EM iteration of a rocprofv3 --kernel-trace of `bench.py --timed-only`: start, gap to the kernel before, duration (us)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
coarse = [i for i, r in enumerate(rows) if 'tiled_coarse' in r['Kernel_Name']]
# bench.py's timed call runs without the phase timers, the call behind it with them (dmx_set_phase_timers)
for title, (a, b) in (('an iteration of the timed call (phase timers off)', (coarse[len(coarse) // 2 - 8], coarse[len(coarse) // 2 - 7])),
                      ('an iteration of the call behind it (phase timers on)', (coarse[-6], coarse[-5]))):
    print('#', title)
    t0 = int(rows[a]['Start_Timestamp'])
    prev_end = None
    for r in rows[a:b + 1]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        print(f"{(s - t0) / 1e3:9.1f} gap {gap:6.1f} dur {(e - s) / 1e3:8.1f} us  {r['Kernel_Name'][:60]}  grid {r['Grid_Size_X']} wg {r['Workgroup_Size_X']}")
        prev_end = e
