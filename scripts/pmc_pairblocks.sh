# Wide doublet tables in the tolerance arithmetic (k_estep_pairblocks<2,3,512>, 94 % of an EM iteration of configs[4]'s shard):
# PMC passes of the timed iterations.  GPU box: bash scripts/pmc_pairblocks.sh [variant]  ->  gpurun_out/pmc_pairblocks_<variant>.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
WL=em_130k_650k_128_doublets
VARIANT=${1:-base}
[ $VARIANT != base ] && export DEMUXALOT_AMD_LIB=$GRAFT_REPO_ROOT/build/variants/libdemux_hip_$VARIANT.so
OUT=gpurun_out/pairblocks_$VARIANT
mkdir -p $OUT
if [ ! -f /tmp/pairblocks_problem/shape.json ]; then
python3 - <<'PY'
import sys
sys.path.insert(0, '.')
from demuxalot_amd import synth
import bench
bench.save_problem('/tmp/pairblocks_problem', synth.generate(130_000, 650_000, 128, doublets=True, seed=1242))
PY
fi
export DEMUXALOT_BENCH_PROBLEM=/tmp/pairblocks_problem
run() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -- python3 bench.py --workload $WL --steps 3 --warmup 1 --timed-only > $OUT/pmc_$name.log 2>&1; }
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS
run grbm GRBM_GUI_ACTIVE
run mem FETCH_SIZE WRITE_SIZE TCP_TCC_READ_REQ_sum TA_BUSY_avr
python3 - $OUT <<'PY' | tee gpurun_out/pmc_pairblocks_$VARIANT.txt
import csv, glob, collections, sys
out = sys.argv[1]
for f in sorted(glob.glob(out + '/pmc_*/*/*kernel_trace.csv'))[:1]:
    spans = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        spans[r['Kernel_Name'].split('(')[0][:60]].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
    for k, v in sorted(spans.items(), key=lambda kv: -sum(kv[1]))[:6]:
        print('trace', k, 'launches', len(v), 'avg_us', round(sum(v) / len(v) / 1e3, 1), 'total_ms', round(sum(v) / 1e6, 2))
for f in sorted(glob.glob(out + '/pmc_*/*/*counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:70]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); disp[k].add(r['Dispatch_Id'])
    for k, d in agg.items():
        if 'pairblocks' in k or 'softmax_rows' in k:
            print(f.split('/')[-3], k, 'launches', len(disp[k]), {c: f'{v/len(disp[k]):.5g}' for c, v in d.items()})
PY
