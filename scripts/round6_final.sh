#!/bin/bash
# Round 6, final build: the GPU suite, the profiles of scripts/profile_r6.sh, the timeline of a cold 5-iteration call, the default bench run
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6_final; mkdir -p $OUT
python -m pytest tests -m gpu -q 2>&1 | tail -8 > $OUT/pytest_gpu.txt
cat $OUT/pytest_gpu.txt
bash scripts/profile_r6.sh > $OUT/profile_r6.log 2>&1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 scripts/cold_call.py > $OUT/cold_call.txt 2>&1
COLD_TRACE=1 rocprofv3 --kernel-trace --output-format csv -d $OUT/cold -o run -- python3 scripts/cold_call.py > $OUT/cold_traced.txt 2>&1
python3 scripts/cold_call_trace.py $OUT/cold/run_kernel_trace.csv > $OUT/cold_call_timeline.txt
rm -rf $OUT/cold
python3 bench.py --steps 20 --warmup 5 > $OUT/line.json 2> $OUT/bench.err
cp gpurun_out/bench_details_em_200k_100k_64_n1.json $OUT/
cat $OUT/cold_call.txt; tail -25 $OUT/cold_call_timeline.txt; cat $OUT/line.json
