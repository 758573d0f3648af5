"""P-step quotients.  The reference divides in float64 and stores float32 (demux.py:267-274): prob = float32(float64(beta) / den), two
roundings.  The kernel does that division (k_probs_from_betas); checked here where a cheaper quotient would go wrong - quotients aimed AT
the midpoints of neighbouring float32 values, wide dynamic range, zeros, subnormal results.  (Round 5 tried the cheaper quotient - a SNP's
reciprocal refined once, a quotient = one multiplication + one correction step, the division itself only within a few float64 ulps of
such a midpoint: these tests passed, and the P-step took the same 0.055 ms - it waits on memory, not on the division.  Not kept.)"""
import numpy as np
import pytest

from tests import fixture_io as fio

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def oracle():
    from oracle import demux_oracle
    return demux_oracle


def _pstep(v2snp, betas, clip):
    from demuxalot_amd import Demultiplexer
    return Demultiplexer._compute_probs_from_betas(np.ascontiguousarray(v2snp, dtype=np.int32), betas, clip)


def test_quotients_aimed_at_float32_midpoints(oracle):
    """Two variants per SNP, float64 betas b1 = m / (1 - m) and b2 = 1 with m the midpoint of two neighbouring float32 values: b1 / (b1 + b2)
    lands within a few float64 ulps of m, on either side - where rounding an approximate quotient to float32 would go the wrong way half of
    the time.  100 000 such SNPs x 8 genotypes, a clip that clips nothing."""
    rng = np.random.default_rng(41)
    n_snps, G = 100_000, 8
    lo = rng.uniform(1e-6, 0.999, size=(n_snps, G)).astype(np.float32)
    hi = np.nextafter(lo, np.float32(2))
    m = (lo.astype(np.float64) + hi.astype(np.float64)) * 0.5                    # exactly representable in float64
    nudge = rng.integers(-3, 4, size=m.shape)                                    # a few float64 ulps to either side, and dead on
    m = (m.view(np.int64) + nudge).view(np.float64)
    betas = np.empty((2 * n_snps, G), dtype=np.float64)
    betas[0::2] = m / (1.0 - m)
    betas[1::2] = 1.0
    v2snp = np.repeat(np.arange(n_snps, dtype=np.int32), 2)
    want = oracle.probs_from_betas(v2snp, betas, 1e-30)
    got = _pstep(v2snp, betas, 1e-30)
    fio.assert_bitwise(got, want, 'quotients at float32 midpoints')
    # the aim was good: a plain float32 division of the rounded operands differs on many of them
    den = betas[0::2] + betas[1::2]
    q = betas[0::2] / den
    dist = np.abs((q.view(np.int64) & 0x1FFFFFFF) - 0x10000000)
    assert (dist <= 8).mean() > 0.5, (dist <= 8).mean()


@pytest.mark.parametrize('seed', [1, 2])
def test_float32_betas_wide_range_bitwise(oracle, seed):
    """The EM path's kernel (float32 betas + float32 addition): 1 .. 6 variants per SNP, betas over ten decades with exact zeros, whole-SNP
    zeros (denominator clipped to 1e-7), quotients down into the float32 subnormals: every bit against the numpy restatement."""
    from demuxalot_amd.device import DeviceContext
    rng = np.random.default_rng(seed)
    n_snps, G = 150_000, 64
    v2snp = np.repeat(np.arange(n_snps, dtype=np.int32), rng.integers(1, 7, size=n_snps))
    V = len(v2snp)
    betas = (10.0 ** rng.uniform(-6, 4, size=(V, G))).astype(np.float32)
    betas[rng.random((V, G)) < 0.05] = 0.0
    betas[v2snp % 97 == 0] = 0.0                                         # SNPs without any mass
    tiny = v2snp % 89 == 1
    betas[tiny] = (betas[tiny] * np.float32(1e-38)).astype(np.float32)  # subnormal / underflowing quotients next to normal ones
    addition = (10.0 ** rng.uniform(-8, 2, size=(V, G))).astype(np.float32) * (rng.random((V, G)) < 0.5)
    addition = addition.astype(np.float32)
    empty_i = np.zeros(0, dtype=np.int32)
    ctx = DeviceContext(0)
    try:
        ctx.set_problem(0, V, G, empty_i, empty_i, np.zeros(0, dtype=np.float32), v2snp)
        ctx.set_betas(betas)
        for add, clip in ((None, 1e-30), (addition, 1e-30), (addition, 0.01)):
            ctx.set_addition(add)
            got = ctx.probs_from_betas(clip)
            want = oracle.probs_from_betas(v2snp, betas if add is None else betas + add, clip)
            fio.assert_bitwise(got, want, f'float32 betas, addition {add is not None}, clip {clip}')
    finally:
        ctx.close()
