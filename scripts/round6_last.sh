#!/bin/bash
# Round 6, the last build: GPU suite, smoke, the default bench run (the committed line), the kernel statistics of the same command
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6_last; mkdir -p $OUT
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|Error" | head -8 > $OUT/pytest_gpu.txt; cat $OUT/pytest_gpu.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python3 bench.py > $OUT/line.json 2> $OUT/bench.err
cp gpurun_out/bench_details_em_200k_100k_64_n1.json $OUT/
cat $OUT/line.json
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-live-traffic --no-hard-workload > $OUT/line_under_tracer.json 2> $OUT/bench_tracer.err
f=$(find $OUT/stats -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $OUT/kernel_stats_raw.csv && head -12 "$f" | cut -c1-160
rm -rf $OUT/stats
