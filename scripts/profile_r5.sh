#!/bin/bash
# Round-5 profiles (run on the GPU box through gpurun; summaries are copied into profiles/ afterwards):
#   rocprofv3 --kernel-trace --stats of the bench command per workload + the bench line of the same run
#   usage: STEPS=20 WARMUP=5 bash scripts/profile_r5.sh <workload> ...
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5_prof
mkdir -p $OUT
for WL in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$WL -- python3 $R/bench.py --workload $WL --steps ${STEPS:-10} --warmup ${WARMUP:-2} --no-cpu-baseline --no-e2e --no-live-traffic --no-hard-workload > $OUT/bench_line_$WL.json 2> $OUT/bench_$WL.err
  f=$(find $OUT/$WL -name '*kernel_stats.csv' | head -1)
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
  rm -rf $OUT/$WL
done
