# Tile size of the tile-major E-step schedule under the coarse pass (build/variants/libdemux_hip_exp.so = repack_device.hip with -DDMX_EXPERIMENTS: DEMUXALOT_AMD_TILE_KB).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/tile_kb_sweep.txt
: > $OUT
python3 - <<'PY'
import sys
sys.path.insert(0, '.')
from demuxalot_amd import synth
import bench
bench.save_problem('/tmp/probe_problem', synth.generate(200_000, 100_000, 64, seed=1237))
PY
export DEMUXALOT_BENCH_PROBLEM=/tmp/probe_problem
for kb in 512 1024 2048 4096 8192; do
  DEMUXALOT_AMD_TILE_KB=$kb DEMUXALOT_AMD_LIB=$GRAFT_REPO_ROOT/build/variants/libdemux_hip_exp.so timeout 300 python3 bench.py --timed-only --steps 30 --warmup 5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tile KB $kb', 'ms_per_step', round(d['ms_per_step'],4), 'estep', round(d['kernel_ms']['estep'],4), 'coarse', d['estep_passes']['device_timed_ms']['coarse_pass'], 'fine', d['estep_passes']['device_timed_ms']['fine_pass'])" >> $OUT
done
cat $OUT
