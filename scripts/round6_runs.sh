#!/bin/bash
# Round 6: the bench lines of the other configurations, the emulated scaling, the certificate-bound experiment (GPU box)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6_runs; mkdir -p $OUT
for WL in em_200k_100k_32 predict_20k_20k_8; do
  python3 bench.py --workload $WL --steps 20 --warmup 5 > $OUT/line_$WL.json 2> $OUT/err_$WL.txt
  cp gpurun_out/bench_details_${WL}_n1.json $OUT/ 2>/dev/null
done
python3 scripts/emulated_scaling.py > $OUT/emulated_scaling_em_200k_100k_64.json 2> $OUT/emulated_err.txt
python3 scripts/certificate_bound.py > $OUT/certificate_bound.txt 2>&1
python3 bench.py --workload em_1M_650k_128_doublets --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-live-traffic > $OUT/line_em_1M_650k_128_doublets.json 2> $OUT/err_1M.txt
cp gpurun_out/bench_details_em_1M_650k_128_doublets_n1.json $OUT/ 2>/dev/null
tail -c 1500 $OUT/line_em_200k_100k_32.json; echo; tail -c 1200 $OUT/line_predict_20k_20k_8.json; echo; tail -c 1500 $OUT/line_em_1M_650k_128_doublets.json; echo; tail -20 $OUT/certificate_bound.txt
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/r6_runs/emulated_scaling_em_200k_100k_64.json'))
for r in d['runs']:
    print(r['scaling'], r['n'], round(r['ms_per_step'], 3), r['kernel_ms'], round(r['speedup_vs_1'], 2))
PY
