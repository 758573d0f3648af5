# Variants of the coarse pass against the shipped build (build/variants/libdemux_hip_<v>.so: kernels.hip compiled with the variant's define).
# GPU box: bash scripts/coarse_variants.sh <variant> ...  ->  gpurun_out/coarse_variants.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/coarse_variants.txt
mkdir -p gpurun_out; : > $OUT
python3 - <<'PY'
import sys
sys.path.insert(0, '.')
from demuxalot_amd import synth
import bench
bench.save_problem('/tmp/probe_problem', synth.generate(200_000, 100_000, 64, seed=1237))
PY
export DEMUXALOT_BENCH_PROBLEM=/tmp/probe_problem
for rep in 1 2; do
  for v in shipped "$@"; do
    lib=$GRAFT_REPO_ROOT/build/variants/libdemux_hip_$v.so
    [ $v = shipped ] && lib=$GRAFT_REPO_ROOT/demuxalot_amd/libdemux_hip.so
    DEMUXALOT_AMD_LIB=$lib timeout 300 python3 bench.py --timed-only --steps 30 --warmup 5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'ms_per_step', round(d['ms_per_step'],4), 'estep', round(d['kernel_ms']['estep'],4), 'mstep', round(d['kernel_ms']['mstep'],4), 'coarse pass (device)', d['estep_passes']['device_timed_ms']['coarse_pass'], 'guard', d['guard']['fraction'])" >> $OUT
  done
done
cat $OUT
