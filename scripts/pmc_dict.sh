# Dictionary-form E-step: kernel times, ablations and PMC passes; GPU box: bash scripts/pmc_dict.sh [workload]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
WL=${1:-em_200k_100k_64}
OUT=gpurun_out/dict_$WL
mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 scripts/predict_loop.py $WL 6 auto > $OUT/trace.log 2>&1
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
# the ablations need an experiment build of the library: make -C demuxalot_amd/csrc clean all EXPERIMENTS=1
for abl in 1 2 3 4 5; do DEMUXALOT_AMD_DICT_ABLATE=$abl timeout 200 python3 scripts/predict_loop.py $WL 6 auto >> $OUT/ablate.log 2>&1; done
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -- python3 scripts/predict_loop.py $WL 3 auto > $OUT/pmc_$name.log 2>&1; }
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS
run tcc1 FETCH_SIZE TCC_HIT_sum TCC_REQ_sum
run tcc2 WRITE_SIZE TCC_MISS_sum
run ta TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
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
        if 'estep' in k or 'build' in k:
            print(f.split('/')[-3], k, 'launches', len(disp[k]), {c: f'{v/len(disp[k]):.5g}' for c, v in d.items()})
PY
cat $OUT/trace.log | tail -2; cat $OUT/ablate.log; head -12 $OUT/kernel_stats.csv; cat $OUT/pmc_summary.txt
