"""Exhaustive: relative error of numpy's float32 log (and of its C restatement oracle/npsimd.c) against the float64
log on every float32 in [1e-4, 2.0002] - the constant GUARD_RHO of the guarded E-step (csrc/estep_epilogue.h) rests
on it.  CPU only, ~1 minute.  Measured in the build container (numpy 2.2.6, AVX512F kernels): 2.7283e-07 at 0.7464124;
per binade: below 8.6e-8 for arguments under 0.1."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import demux_oracle  # noqa: E402

lo, hi = np.float32(1e-4).view(np.uint32), np.float32(2.0002).view(np.uint32)
worst, where = 0.0, None
for start in range(int(lo), int(hi) + 1, 1 << 24):
    t = np.arange(start, min(start + (1 << 24), int(hi) + 1), dtype=np.uint32).view(np.float32)
    true = np.log(t.astype(np.float64))
    for name, got in (('numpy', np.log(t)), ('npsimd', demux_oracle.log_f32(t, impl='npsimd'))):
        with np.errstate(divide='ignore', invalid='ignore'):
            rel = np.where(true != 0, np.abs(got.astype(np.float64) - true) / np.abs(true), np.abs(got))
        i = int(rel.argmax())
        print(f'{name:7s} {t[0]:.6g} .. {t[-1]:.6g}: max relative error {rel[i]:.4e} at {t[i]!r}')
        if rel[i] > worst:
            worst, where = float(rel[i]), (name, float(t[i]))
print('worst', worst, where)
