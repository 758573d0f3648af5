# M-step counters for one kernel variant: bash scripts/pmc_mstep.sh <DMX_MSTEP value>   (GPU box)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export DMX_MSTEP=$1
TAG=ms_$1
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_${TAG}_$name -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_${TAG}_$name.log 2>&1; }
run a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run b SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM
run c FETCH_SIZE TCC_HIT_sum
run d WRITE_SIZE TCC_MISS_sum
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_${TAG}_*/*/*counter_collection.csv'):
    for row in csv.DictReader(open(f)):
        if 'mstep' in row['Kernel_Name'] and 'combine' not in row['Kernel_Name']:
            acc[row['Counter_Name']].append(float(row['Counter_Value']))
print('$1', {k: round(sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
