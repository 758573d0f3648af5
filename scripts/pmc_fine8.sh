#!/bin/bash
# PMC passes (one group per run) of scripts/fine8_timing.py: k_estep_tiled<1,true,false> against k_estep_tiled_fine8<2> against k_estep_tiled_coarse<2>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_fine8; rm -rf $OUT; mkdir -p $OUT
pmc() { tag=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$tag -- python3 scripts/fine8_timing.py > $OUT/$tag.log 2>&1; }
pmc sq SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS
pmc grbm GRBM_GUI_ACTIVE
pmc ta TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
pmc fetch FETCH_SIZE
python3 - <<'PY'
import collections, csv, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_fine8/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void dmx::', '')
        if not k.startswith('k_estep_tiled'):
            continue
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    # launches that did work: keep those above the median / 4 for SQ_WAVES-independent counters
    print(k)
    for c in sorted(acc[k]):
        v = sorted(acc[k][c]); big = [x for x in v if x > 0.25 * v[-1]] or v
        print(f'    {c:32s} launches {len(v):4d} (working {len(big):3d})  mean of the working ones {sum(big) / len(big):16.1f}')
PY
