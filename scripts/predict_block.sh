# wide doublet tables: predict pass with the dictionary block kernel, tiles compared; GPU box
# DEMUXALOT_AMD_DICT_TILE needs an experiment build of the library: make -C demuxalot_amd/csrc clean all EXPERIMENTS=1
for t in 17 12 8; do DEMUXALOT_AMD_DICT_TILE=$t python3 scripts/predict_loop.py em_130k_650k_128_doublets 3 always 2>&1 | tail -1; done
python3 scripts/predict_loop.py predict_20k_20k_64_doublets 5 always 2>&1 | tail -1
python3 scripts/predict_loop.py predict_20k_20k_64_doublets 5 never 2>&1 | tail -1
python3 scripts/predict_loop.py predict_20k_20k_32_doublets 5 always 2>&1 | tail -1
python3 scripts/predict_loop.py predict_20k_20k_32_doublets 5 never 2>&1 | tail -1
