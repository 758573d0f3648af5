# round 3 profiles; run on the GPU box:  bash scripts/profile_r3.sh
# Per workload: rocprofv3 --kernel-trace --stats of the bench command (summary -> profiles/r3_kernel_stats_*.csv) and the bench line;
# for the headline workload also the PMC passes of the dictionary-form E-step (separate runs, one counter set per pass).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/profiles_r3
mkdir -p $OUT
for wl in em_200k_100k_64 em_200k_100k_32 predict_20k_20k_8 em_130k_650k_128_doublets; do
  steps=10; [ $wl = em_130k_650k_128_doublets ] && steps=3
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3_$wl -- python3 bench.py --steps $steps --warmup 2 --no-cpu-baseline --no-e2e --no-live-traffic --workload $wl > $OUT/r3_bench_line_$wl.json 2> $OUT/bench_$wl.err
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
done
bash scripts/pmc_dict.sh em_200k_100k_64 > $OUT/pmc_dict.log 2>&1
cp gpurun_out/dict_em_200k_100k_64/pmc_summary.txt $OUT/r3_pmc_dictq_em_200k_100k_64.txt 2>/dev/null
grep "ablate" gpurun_out/dict_em_200k_100k_64/ablate.log >> $OUT/r3_pmc_dictq_em_200k_100k_64.txt 2>/dev/null
bash scripts/pmc_block.sh predict_20k_20k_64_doublets > $OUT/r3_pmc_dict_block_predict_20k_20k_64_doublets.txt 2>&1
ls -la $OUT
