# bench.py --timed-only of one workload under variant builds: bash scripts/variant_bench.sh <workload> <steps> base <variant> ...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
WL=$1; STEPS=$2; shift; shift
for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/build/variants/libdemux_hip_$v.so
  [ $v = base ] && lib=$GRAFT_REPO_ROOT/demuxalot_amd/libdemux_hip.so
  DEMUXALOT_AMD_LIB=$lib timeout 600 python3 bench.py --workload $WL --timed-only --steps $STEPS --warmup 2 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$WL', '$v', 'ms_per_step', round(d['ms_per_step'],3), 'estep_ms', round(d['kernel_ms']['estep'],3), 'guard', round(d['guard']['fraction'],4))"
done
