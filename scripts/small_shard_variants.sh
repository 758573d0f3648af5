# The tile-major schedule (and with it the coarse pass) below 65 536 barcodes: build/variants/libdemux_hip_t8k.so = -DDMX_TILE_MIN_BARCODES=8192.
# GPU box: bash scripts/small_shard_variants.sh  ->  gpurun_out/small_shard_variants.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/small_shard_variants.txt
: > $OUT
for wl in ${WORKLOADS:-em_25k_100k_64 em_20k_10k_64}; do
  for v in shipped t8k; do
    lib=$GRAFT_REPO_ROOT/build/variants/libdemux_hip_$v.so
    [ $v = shipped ] && lib=$GRAFT_REPO_ROOT/demuxalot_amd/libdemux_hip.so
    for rep in 1 2; do
      DEMUXALOT_AMD_LIB=$lib timeout 300 python3 bench.py --workload $wl --timed-only --steps 30 --warmup 5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$wl', '$v', 'ms_per_step', round(d['ms_per_step'],4), {k:round(x,4) for k,x in d['kernel_ms'].items()}, d['estep_passes']['coarse'], d['guard']['fraction'])" >> $OUT
    done
  done
done
cat $OUT
