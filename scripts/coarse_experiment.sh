# Coarse pass / single-exit pipeline loops against the last commit's build (build/base_tree: git archive HEAD + its library).
# GPU box: bash scripts/coarse_experiment.sh  ->  gpurun_out/coarse_experiment.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/coarse_experiment.txt
mkdir -p gpurun_out; : > $OUT
python3 - <<'PY'
import sys
sys.path.insert(0, '.')
from demuxalot_amd import synth
import bench
p = synth.generate(200_000, 100_000, 64, seed=1237)
bench.save_problem('/tmp/probe_problem', p)
PY
export DEMUXALOT_BENCH_PROBLEM=/tmp/probe_problem
line() { python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'estep_ms', round(d['kernel_ms']['estep'],4), 'mstep', round(d['kernel_ms']['mstep'],4), 'ms_per_step', round(d['ms_per_step'],4), d.get('guard'))"; }
for rep in 1 2; do
  (cd build/base_tree && timeout 300 python3 bench.py --timed-only --steps 30 --warmup 5 2>/dev/null | line base) >> $OUT
  timeout 300 python3 bench.py --timed-only --steps 30 --warmup 5 2>/dev/null | line new_fine >> $OUT
done
timeout 900 python3 scripts/coarse_probe.py em_200k_100k_64 3 >> $OUT 2>&1
cat $OUT
