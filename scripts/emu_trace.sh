cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/emu_trace; mkdir -p gpurun_out/emu_trace
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/emu_trace -o run -- python3 scripts/emulated_scaling.py --kinds strong --ranks 8 --steps 12 --warmup 8 > /dev/null 2> gpurun_out/emu_trace/err.txt
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/emu_trace/run_kernel_trace.csv')))
try:
    rows += [dict(Kernel_Name='MEMCPY ' + r.get('Direction', ''), Start_Timestamp=r['Start_Timestamp'], End_Timestamp=r['End_Timestamp']) for r in csv.DictReader(open('gpurun_out/emu_trace/run_memory_copy_trace.csv'))]
except Exception as e:
    print('no memcpy trace', e)
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ms = [i for i, r in enumerate(rows) if 'k_probs_from_betas' in r['Kernel_Name']]
a, b = ms[-3], ms[-2]
t0 = int(rows[a]['Start_Timestamp']); prev = None
for r in rows[a:b + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0) / 1e3:8.1f} gap {((s - prev) / 1e3 if prev else 0):6.1f} dur {(e - s) / 1e3:7.1f}  {r['Kernel_Name'].replace('void dmx::', '').split('(')[0][:60]}")
    prev = e
PY
