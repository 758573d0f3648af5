#!/bin/bash
# Round 6: the GPU suite and the emulated scaling tables on the build with the compact exchanges
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6_multi; mkdir -p $OUT
python -m pytest tests -m gpu -q 2>&1 | tail -6 > $OUT/pytest_gpu.txt; cat $OUT/pytest_gpu.txt
python3 scripts/emulated_scaling.py > $OUT/emulated_scaling_em_200k_100k_64.json 2> $OUT/emu_64.txt; cat $OUT/emu_64.txt
DEMUXALOT_AMD_EXCHANGE_COMPACT=0 python3 scripts/emulated_scaling.py --kinds strong > $OUT/emulated_scaling_em_200k_100k_64_whole_tables.json 2> $OUT/emu_64_whole.txt; cat $OUT/emu_64_whole.txt
python3 scripts/emulated_scaling.py --modes exact --kinds strong > $OUT/emulated_scaling_em_200k_100k_64_exact.json 2> $OUT/emu_64_exact.txt; cat $OUT/emu_64_exact.txt
python3 scripts/emulated_scaling.py --workload em_130k_650k_128_doublets --kinds weak --ranks 1,8 --steps 3 --warmup 1 > $OUT/emulated_scaling_em_130k_650k_128_doublets.json 2> $OUT/emu_130k.txt; cat $OUT/emu_130k.txt
