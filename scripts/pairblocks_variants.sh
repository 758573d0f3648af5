# Timed iterations of configs[4]'s rank share (em_130k_650k_128_doublets) with variant builds of k_estep_pairblocks
# (build/variants/libdemux_hip_<name>.so; csrc/kernels.hip: DMX_PAIRBLOCK_WAVES, DMX_PAIRBLOCK_SERIAL_PRODUCT).
# GPU box: bash scripts/pairblocks_variants.sh base unroll8 ...  ->  gpurun_out/pairblocks_variants.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pairblocks_variants.txt
: > $OUT
if [ ! -f /tmp/pairblocks_problem/shape.json ]; then
python3 - <<'PY'
import sys
sys.path.insert(0, '.')
from demuxalot_amd import synth
import bench
bench.save_problem('/tmp/pairblocks_problem', synth.generate(130_000, 650_000, 128, doublets=True, seed=1242))
PY
fi
for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/build/variants/libdemux_hip_$v.so
  [ $v = base ] && lib=$GRAFT_REPO_ROOT/demuxalot_amd/libdemux_hip.so
  DEMUXALOT_BENCH_PROBLEM=/tmp/pairblocks_problem DEMUXALOT_AMD_LIB=$lib timeout 600 python3 bench.py --workload em_130k_650k_128_doublets --timed-only --steps 4 --warmup 1 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'ms_per_step', round(d['ms_per_step'],2), 'estep_ms', round(d['kernel_ms']['estep'],2), 'guard', d['guard']['fraction'])" >> $OUT
done
cat $OUT
