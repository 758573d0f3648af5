# Direct-form E-step (exact mode, no dictionary): PMC passes; GPU box: bash scripts/pmc_direct.sh [workload ...]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for WL in "$@"; do
OUT=gpurun_out/direct_$WL
mkdir -p $OUT
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -- python3 scripts/predict_loop.py $WL 3 never > $OUT/pmc_$name.log 2>&1; }
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM
run grbm GRBM_GUI_ACTIVE
python3 - $OUT <<'PY' > $OUT/pmc_summary.txt 2>&1
import csv, glob, collections, sys
out = sys.argv[1]
for f in sorted(glob.glob(out + '/pmc_*/*/*counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:70]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); disp[k].add(r['Dispatch_Id'])
    for k, d in agg.items():
        if 'estep' in k:
            print(f.split('/')[-3], k, 'launches', len(disp[k]), {c: f'{v/len(disp[k]):.5g}' for c, v in d.items()})
PY
tail -1 $OUT/pmc_sq1.log; cat $OUT/pmc_summary.txt
done
