# PMC passes for the M-step kernels of the headline workload; GPU box: bash scripts/pmc_mstep.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_m_$name -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-mode > gpurun_out/pmc_m_$name.log 2>&1; }
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum
run grbm GRBM_GUI_ACTIVE
python3 - <<'PY'
import csv, glob, collections
for name in ('sq1', 'tcc1', 'tcp', 'grbm'):
    for f in glob.glob(f'gpurun_out/pmc_m_{name}/*/*counter_collection.csv'):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0][:60]
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); disp[k].add(r['Dispatch_Id'])
        for k, d in agg.items():
            if 'mstep_calls' in k:
                print(name, k, len(disp[k]), {c: f'{v/len(disp[k]):.4g}' for c, v in d.items()})
PY
