# P-step lanes per variant row (round 5; variant libraries built by hand with another PSTEP(L) in launch_pstep): 32-genotype tables 32 / 16 lanes: 0.032 / 0.028 ms (16 shipped since);
# 64-genotype tables: 64 / 32 / 16 lanes measured 0.054 / 0.045 / 0.048 ms on 200 000 variants (32 shipped since).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/pstep_lanes.txt
: > $OUT
for v in shipped pstepN32; do
  lib=$GRAFT_REPO_ROOT/build/variants/libdemux_hip_$v.so
  [ $v = shipped ] && lib=$GRAFT_REPO_ROOT/demuxalot_amd/libdemux_hip.so
  for wl in em_200k_100k_32 em_200k_100k_64; do
    DEMUXALOT_AMD_LIB=$lib timeout 300 python3 bench.py --workload $wl --timed-only --steps 30 --warmup 5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$wl', '$v', 'ms_per_step', round(d['ms_per_step'],4), {k:round(x,4) for k,x in d['kernel_ms'].items()})" >> $OUT
  done
done
cat $OUT
