# Tile-major tolerance E-step (and tile-major M-step): PMC passes of the timed iterations; GPU box: bash scripts/pmc_estep_tiled.sh [workload] [variant]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
WL=${1:-em_200k_100k_64}
VARIANT=${2:-base}   # base | aux16 | depth6 ...: build/variants/libdemux_hip_<variant>.so (scripts/gather_experiments.sh)
[ $VARIANT != base ] && export DEMUXALOT_AMD_LIB=$GRAFT_REPO_ROOT/build/variants/libdemux_hip_$VARIANT.so
OUT=gpurun_out/estep_tiled_${WL}_$VARIANT
mkdir -p $OUT
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -- python3 bench.py --workload $WL --steps 10 --warmup 2 --timed-only > $OUT/pmc_$name.log 2>&1; }
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS
run ta TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
run grbm GRBM_GUI_ACTIVE
run tcc1 FETCH_SIZE TCC_HIT_sum TCC_REQ_sum TCC_MISS_sum
run tcp TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum
python3 - $OUT <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
for f in sorted(glob.glob(out + '/pmc_*/*/*counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:70]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); disp[k].add(r['Dispatch_Id'])
    for k, d in agg.items():
        if 'estep_tiled' in k or 'mstep_tiles' in k:
            print(f.split('/')[-3], k, 'launches', len(disp[k]), {c: f'{v/len(disp[k]):.5g}' for c, v in d.items()})
PY
