# kernel times of the predict pass (dictionary form); GPU box: bash scripts/trace_dict.sh workload...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for WL in "$@"; do
  OUT=gpurun_out/trace_$WL; mkdir -p $OUT
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 scripts/predict_loop.py $WL 6 auto > $OUT/log.txt 2>&1
  tail -1 $OUT/log.txt
  python3 - $OUT <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/t/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Name'].split('(')[0][-60:]
        if 'rocprim' in n: continue
        print('%-62s calls %4s avg_us %10.1f' % (n, r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
