# Row-gather experiments of k_estep_tiled<1,true,false>: cache-policy bits on the gathers' buffer loads and
# deeper / shallower gather pipelines.  Variant libraries are built by hand (csrc/kernels.hip: DMX_ROW_AUX, DMX_GATHER_DEPTH):
#   hipcc ... -DDMX_ROW_AUX=2 -c kernels.hip -o kernels_aux2.o; hipcc -shared ... -o build/variants/libdemux_hip_aux2.so
# GPU box: bash scripts/gather_experiments.sh   ->  gpurun_out/gather_experiments.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/gather_experiments.txt
: > $OUT
python3 - <<'PY'
import os, sys
sys.path.insert(0, '.')
import numpy as np
from demuxalot_amd import synth
import bench
p = synth.generate(200_000, 100_000, 64, seed=1237)
bench.save_problem('/tmp/gather_problem', p)
PY
for v in base aux2 aux16 depth3 depth5 depth6; do
  lib=$GRAFT_REPO_ROOT/build/variants/libdemux_hip_$v.so
  [ $v = base ] && lib=$GRAFT_REPO_ROOT/demuxalot_amd/libdemux_hip.so
  for rep in 1 2; do
    DEMUXALOT_BENCH_PROBLEM=/tmp/gather_problem DEMUXALOT_AMD_LIB=$lib timeout 300 python3 bench.py --timed-only --steps 30 --warmup 5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'estep_ms', round(d['kernel_ms']['estep'],4), 'ms_per_step', round(d['ms_per_step'],4))" >> $OUT
  done
done
cat $OUT
