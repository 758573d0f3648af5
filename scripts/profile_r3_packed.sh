# round 3, packed form: rocprofv3 --kernel-trace --stats of the bench command on the narrow doublet tables; GPU box: bash scripts/profile_r3_packed.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/profiles_r3
mkdir -p $OUT
for wl in predict_200k_20k_8 predict_200k_20k_12 predict_20k_20k_8; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3_$wl -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-live-traffic --no-fast-mode --workload $wl > $OUT/r3_bench_line_$wl.json 2> $OUT/bench_$wl.err
  python3 - gpurun_out/prof_r3_$wl $OUT/r3_kernel_stats_$wl.csv <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/*/*kernel_stats.csv'):
    rows = list(csv.DictReader(open(f)))
with open(sys.argv[2], 'w') as out:
    w = csv.writer(out)
    w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
    for r in rows:
        name = r['Name'].replace('(anonymous namespace)::', '').split('(')[0]
        if 'rocprim' in name: name = 'rocprim::' + name.split('rocprim::')[-1][:60] + ' (device repack)'
        w.writerow([name, r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'], r['MaxNs']])
PY
  head -4 $OUT/r3_kernel_stats_$wl.csv
done
