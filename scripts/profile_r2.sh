# round 2 profiles; run on the GPU box:  bash scripts/profile_r2.sh
# Per workload: rocprofv3 --kernel-trace --stats of the bench command; for the headline workload also the PMC passes
# (separate runs, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/profiles_r2
mkdir -p $OUT
for wl in em_200k_100k_64 em_200k_100k_32 predict_20k_20k_8 em_130k_650k_128_doublets; do
  steps=10; [ $wl = em_130k_650k_128_doublets ] && steps=3
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2_$wl -- python3 bench.py --steps $steps --warmup 2 --no-cpu-baseline --workload $wl > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
  cp gpurun_out/prof_r2_$wl/*/*kernel_stats.csv $OUT/r2_kernel_stats_$wl.csv 2>/dev/null
done
# M-step worst case (uniform posteriors)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2_flat -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-fast-mode --flat-genotypes > $OUT/bench_em_200k_100k_64_flat.json 2> $OUT/bench_flat.err
cp gpurun_out/prof_r2_flat/*/*kernel_stats.csv $OUT/r2_kernel_stats_em_200k_100k_64_flat.csv 2>/dev/null
run() { name=$1; shift; timeout 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_r2_$name -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_r2_$name.log 2>&1; }
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run tcc1 FETCH_SIZE TCC_HIT_sum
run tcc2 WRITE_SIZE TCC_MISS_sum
run grbm GRBM_GUI_ACTIVE
python3 scripts/summarize_pmc.py r2 > $OUT/summarize.log 2>&1
cp gpurun_out/profiles_out/r2_pmc_em_200k_100k_64.txt gpurun_out/profiles_out/pmc_traffic.json $OUT/ 2>/dev/null
ls -la $OUT
