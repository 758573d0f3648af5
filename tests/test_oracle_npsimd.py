"""oracle/npsimd.c (scalar restatement of numpy's float32 log / exp / sum) against numpy itself.
In the build container the same comparison was run exhaustively (every float32 in [1e-5,4) for
log, every float32 in [-104.5,0] for exp: 0 mismatches); here a dense strided sample keeps the
CPU suite short. If numpy on the running host dispatches to a different kernel (no AVX2/AVX512F),
the comparison is skipped: the fixtures, not the host's numpy, are what pin the oracle."""
import numpy as np
import pytest


def _numpy_uses_simd_kernels(oracle):
    probe = np.array([0.785213, 0.0123, 0.5, 0.99999994, 1e-4], dtype=np.float32)
    got = np.log(probe)
    want = np.array([oracle.load_npsimd().npsimd_logf(float(v)) for v in probe], dtype=np.float32)
    return np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_log_matches_numpy(oracle):
    if not _numpy_uses_simd_kernels(oracle):
        pytest.skip("host numpy does not use the AVX2/AVX512F float32 kernels")
    lo, hi = np.float32(1e-5).view(np.int32), np.float32(4.0).view(np.int32)
    bits = np.arange(int(lo), int(hi), 13, dtype=np.int32)
    x = bits.view(np.float32)
    assert np.array_equal(np.log(x).view(np.uint32), oracle.log_f32(x, impl='npsimd').view(np.uint32))


def test_exp_and_softmax_match_numpy(oracle):
    if not _numpy_uses_simd_kernels(oracle):
        pytest.skip("host numpy does not use the AVX2/AVX512F float32 kernels")
    from scipy.special import softmax
    lo, hi = np.float32(-0.0).view(np.uint32), np.float32(-104.5).view(np.uint32)
    bits = np.arange(int(lo), int(hi), 97, dtype=np.uint32)
    x = bits.view(np.float32)
    assert np.array_equal(np.exp(x).view(np.uint32), oracle._c_unary('npsimd_exp_array', x).view(np.uint32))
    rng = np.random.default_rng(5)
    for K in (1, 2, 7, 8, 9, 20, 36, 64, 127, 128, 129, 210, 256, 528, 2080, 8191, 8192, 8193, 8256, 17000):
        logits = (rng.normal(size=(6, K)) * rng.choice([1., 30., 200.])).astype(np.float32)
        want = softmax(logits, axis=-1)
        got = oracle.softmax_rows(logits, impl='npsimd')
        assert np.array_equal(want.view(np.uint32), got.view(np.uint32)), K


def test_log_relative_error_bound_of_the_guarded_mode(oracle):
    """The guarded E-step (include/demux_hip.h: DMX_ESTEP_GUARDED, csrc/estep_epilogue.h: GUARD_RHO = 3.0e-7) bounds the
    deviation of the reference's own logits from the true sums with: numpy's float32 log is within 2.73e-7 RELATIVE of
    the true log for every float32 argument a term can take, [1e-4, 2.0002].  scripts/log_relative_error.py runs the
    comparison on every float32 of the range (1.2e8 values, 2.7283e-07 at 0.7464124); here every 11th one, on the C
    restatement (which equals numpy bit for bit, see above) so that the host's numpy dispatch does not matter."""
    lo, hi = np.float32(1e-4).view(np.uint32), np.float32(2.0002).view(np.uint32)
    worst = 0.0
    for start in range(int(lo), int(hi) + 1, 11 << 21):
        bits = np.arange(start, min(start + (11 << 21), int(hi) + 1), 11, dtype=np.uint32)
        t = bits.view(np.float32)
        got = oracle.log_f32(t, impl='npsimd').astype(np.float64)
        true = np.log(t.astype(np.float64))
        with np.errstate(divide='ignore', invalid='ignore'):
            rel = np.where(true != 0, np.abs(got - true) / np.abs(true), np.abs(got))
        worst = max(worst, float(rel.max()))
    assert worst <= 2.8e-7, worst
