# M-step ablations (timing only; an ablated kernel gives wrong results by construction).  Put `#ifdef ABLATE_<NAME>` blocks
# at the points of interest in csrc/kernels.hip (e.g. skip the queue machinery, replace a gather by arithmetic: the
# variants behind the figures of DESIGN.md 4.2 were LOADS_ONLY, NO_GATHER, NO_EXTRAS, NO_DENSE), then here:
#   bash scripts/ablate_mstep.sh build NAME1 NAME2 ...      (builds build/ablate_NAME/libdemux_hip.so)
# and on the GPU box, timed on the posteriors of the first E-step of the headline workload:
#   bash scripts/ablate_mstep.sh run NAME1 NAME2 ...
set -e
cd "$(dirname "$0")/.."
SRC=demuxalot_amd/csrc
MODE=$1; shift
if [ "$MODE" = build ]; then
  for v in "$@"; do
    mkdir -p build/ablate_$v
    /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -Iinclude -I/opt/rocm/include --offload-arch=gfx950 \
        -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-rdc -DABLATE_$v -c $SRC/kernels.hip -o build/ablate_$v/kernels.o &
  done
  wait
  for v in "$@"; do
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build/ablate_$v/libdemux_hip.so build/ablate_$v/kernels.o \
        $SRC/dmx_api.o $SRC/repack_device.o $SRC/results.o $SRC/snp_aggregate.o $SRC/pack_host.o -ldl
  done
else
  cat > /tmp/ablate_run.py <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
from demuxalot_amd import synth
from demuxalot_amd.device import DeviceContext
G = int(sys.argv[2]) if len(sys.argv) > 2 else 64
p = synth.generate(200_000, 100_000, G, seed=0)
ctx = DeviceContext(0)
ctx.set_problem(p.n_barcodes, p.n_variants, G, p.variant_id, p.compressed_cb, p.p_base_wrong, p.v2snp)
ctx.set_betas(p.prior_betas(add_data_prior=False)); ctx.set_addition(None)
ctx.probs_from_betas(0.01, fetch=False)
ctx.estep(np.zeros(G, dtype=np.float32), with_doublets=False, fetch_logits=False, fetch_probs=False)
for _ in range(3): ctx.mstep(2., fetch=False)
ctx.reset_timings()
for _ in range(10): ctx.mstep(2., fetch=False)
t = ctx.timings()
print(sys.argv[1], 'G', G, {k: round(v['ms'] / max(1, v['launches']), 4) for k, v in t.items() if v['launches']})
PY
  python /tmp/ablate_run.py baseline 2>/dev/null | tail -1
  for v in "$@"; do
    DEMUXALOT_AMD_LIB=$PWD/build/ablate_$v/libdemux_hip.so python /tmp/ablate_run.py $v 2>/dev/null | tail -1
  done
fi
