# PMC passes of the M-step worst case (uniform posteriors: k_mstep_dense); run on the GPU box: bash scripts/pmc_flat.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_flat_$name -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-mode --flat-genotypes > gpurun_out/pmc_flat_$name.log 2>&1; }
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run tcc1 FETCH_SIZE TCC_HIT_sum
run tcc2 WRITE_SIZE TCC_MISS_sum
python3 - <<'PY'
import csv, glob, collections
for name in ('sq1', 'tcc1', 'tcc2'):
    for f in glob.glob(f'gpurun_out/pmc_flat_{name}/*/*counter_collection.csv'):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0][:60]
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        for k, d in agg.items():
            if 'mstep' in k or 'estep' in k:
                print(name, k, {c: f'{v:.4g}' for c, v in d.items()})
PY
