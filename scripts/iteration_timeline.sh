# Kernel trace of the timed region only (bench.py --timed-only): the timeline of one EM iteration, kernels and the gaps between them.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/timeline && mkdir -p gpurun_out/timeline
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/timeline -o run -- python3 bench.py --timed-only --steps 20 --warmup 5 > gpurun_out/timeline/bench.json 2> gpurun_out/timeline/bench.err
python3 bench.py --timed-only --steps 20 --warmup 5 > gpurun_out/timeline/bench_plain.json 2> gpurun_out/timeline/bench_plain.err
python3 scripts/iteration_timeline.py gpurun_out/timeline/run_kernel_trace.csv > gpurun_out/timeline/timeline.txt
cat gpurun_out/timeline/timeline.txt
