# round 2: fast-mode E-step variants -- tolerance tests, then timings of tiled / direct schedules
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_fast_mode.py -x -q -m gpu 2>&1 | tail -4
for sched in tiled direct; do
  DEMUXALOT_AMD_ESTEP_SCHEDULE=$sched python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_$sched.json 2> gpurun_out/bench_$sched.err
  python - <<PY
import json
d = json.load(open('gpurun_out/bench_$sched.json'))
print('$sched', 'exact', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['kernel_ms'].items()})
print('$sched', 'fast ', round(d['fast_mode']['ms_per_step'], 3), {k: round(v, 3) for k, v in d['fast_mode']['kernel_ms'].items()}, d['fast_mode']['vs_exact_first_pass'])
PY
done
