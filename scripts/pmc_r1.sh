# rocprofv3 passes for the bench workload; run on the GPU box: bash scripts/pmc_r1.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r1}
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof_${TAG}_trace.log 2>&1
run() { name=$1; shift; timeout 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_${TAG}_$name -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_${TAG}_$name.log 2>&1; }
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_CVT SQ_INSTS_SMEM
run tcc1 FETCH_SIZE TCC_HIT_sum
run tcc2 WRITE_SIZE TCC_MISS_sum
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE
python3 scripts/summarize_pmc.py $TAG
