#!/bin/bash
# Round-6 profiles (GPU box, through gpurun; the summaries are copied into profiles/ afterwards):
#   1. rocprofv3 --kernel-trace --stats of the bench command (default run, no child profiler runs) -> kernel_stats_<workload>.csv + the line
#   2. PMC passes (one group per run, as MI355X_MICROARCH.md prescribes) of the timed call only (bench.py --timed-only):
#      default mode -> k_estep_tiled_coarse + k_mstep_tiles;  DEMUXALOT_AMD_ESTEP=exact -> k_estep_direct + k_mstep_calls
#   usage: bash scripts/profile_r6.sh [workload]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
WL=${1:-em_200k_100k_64}
OUT=gpurun_out/r6_prof
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$WL -- python3 bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-live-traffic --no-hard-workload > $OUT/bench_line_${WL}_under_tracer.json 2> $OUT/bench_$WL.err
f=$(find $OUT/stats_$WL -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && python3 - "$f" > $OUT/kernel_stats_$WL.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
w = csv.writer(sys.stdout)
w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
for r in rows:
    name = r['Name'].replace('(anonymous namespace)::', '').split('(')[0]
    if 'rocprim' in name:
        name = 'rocprim::' + name.split('rocprim::')[-1][:60] + ' (device repack)'
    w.writerow([name, r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'], r['MaxNs']])
PY
rm -rf $OUT/stats_$WL
pmc() {  # pmc <tag> <counters...>: one pass
  tag=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$tag -- python3 bench.py --workload $WL --steps 12 --warmup 2 --timed-only > $OUT/pmc_$tag.log 2>&1
}
for MODE in default exact; do
  if [ $MODE = exact ]; then export DEMUXALOT_AMD_ESTEP=exact; else unset DEMUXALOT_AMD_ESTEP; fi
  pmc ${MODE}_fetch FETCH_SIZE
  pmc ${MODE}_write WRITE_SIZE
  pmc ${MODE}_grbm GRBM_GUI_ACTIVE
  pmc ${MODE}_sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
  pmc ${MODE}_sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS
  pmc ${MODE}_ta TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
  pmc ${MODE}_tcc TCC_HIT_sum TCC_REQ_sum TCC_MISS_sum
done
unset DEMUXALOT_AMD_ESTEP
python3 - $OUT > $OUT/pmc_summary.txt <<'PY'
import csv, glob, collections, sys, os
out = sys.argv[1]
print('# rocprofv3 --pmc passes of `bench.py --timed-only --steps 12 --warmup 2` (one counter group per run); per kernel: MEAN PER LAUNCH over the')
print('# launches that did work (stand-back launches, < 20 us, are left out), FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them, duration from the same pass')
for mode in ('default', 'exact'):
    print(f'\n## {mode} mode')
    for d in sorted(glob.glob(f'{out}/pmc_{mode}_*/')):
        tag = os.path.basename(d.rstrip('/'))
        trace = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)
        coll = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
        if not trace or not coll:
            print(tag, 'no output'); continue
        dur = {}
        for r in csv.DictReader(open(trace[0])):
            dur[r['Dispatch_Id']] = (r['Kernel_Name'], float(r['End_Timestamp']) - float(r['Start_Timestamp']))
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
        for r in csv.DictReader(open(coll[0])):
            name, ns = dur.get(r['Dispatch_Id'], (r['Kernel_Name'], 0.0))
            if ns < 20e3:
                continue
            k = name.split('(')[0].replace('void dmx::', '').replace('dmx::', '')[:60]
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); disp[k].add(r['Dispatch_Id'])
        for k, c in agg.items():
            if not any(s in k for s in ('k_estep', 'k_mstep', 'k_probs')):
                continue
            n = len(disp[k])
            ns = sum(dur[i][1] for i in disp[k]) / n
            print(f'{tag:18s} {k:44s} launches {n:3d}  avg {ns / 1e3:8.1f} us  ' + '  '.join(f'{name} {v / n:.6g}' for name, v in sorted(c.items())))
PY
cat $OUT/pmc_summary.txt
