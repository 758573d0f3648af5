# round 2: tile-major E-step -- parity at the headline size, timings of the four E-step variants, L2 counters
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "config3" 2>&1 | tail -5
for sched in tiled direct; do
  DEMUXALOT_AMD_ESTEP_SCHEDULE=$sched python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_$sched.json 2> gpurun_out/bench_$sched.err
  python - <<PY
import json
d = json.load(open('gpurun_out/bench_$sched.json'))
print('$sched', 'exact', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['kernel_ms'].items()})
print('$sched', 'fast ', round(d['fast_mode']['ms_per_step'], 3), {k: round(v, 3) for k, v in d['fast_mode']['kernel_ms'].items()}, d['fast_mode']['vs_exact_first_pass'])
PY
done
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_r2t_$name -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_r2t_$name.log 2>&1; }
run tcc1 FETCH_SIZE TCC_HIT_sum
run tcc2 WRITE_SIZE TCC_MISS_sum
python3 scripts/summarize_pmc.py r2t | tail -40
