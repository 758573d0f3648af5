/*
 * oracle/npsimd.c -- TEST INFRASTRUCTURE ONLY (never imported by demuxalot_amd/).
 *
 * Scalar C restatement of the float32 arithmetic that the reference's hot path
 * delegates to numpy (the reference itself contains no native code):
 *
 *   np.log  on float32 arrays   -> reference demux.py:261
 *   np.exp  on float32 arrays   -> scipy.special.softmax, reference demux.py:101,152
 *   np.sum(axis=-1) on float32  -> scipy.special.softmax (same lines) and
 *                                  betas.sum(axis=1) in reference demux.py:383
 *
 * numpy 2.2.6 (the pinned oracle environment) evaluates
 * float32 log/exp on x86 (AVX2+FMA3 or AVX512F) with its own rational
 * minimax kernels (numpy/_core/src/umath/loops_exponent_log.dispatch.c.src,
 * published under the BSD licence; coefficients in npy_simd_data.h).  The
 * published algorithm is restated here operation by operation with fmaf() so
 * that every rounding matches:
 *
 *   log: x = m * 2^k, m in [0.5,1); if m <= 1/sqrt(2): m *= 2, k -= 1;
 *        r = m - 1; log(1+r) = P5(r)/Q5(r) (Horner, fused multiply-add);
 *        result = fma(k, ln2, P/Q).
 *   exp: k = rint(x * log2(e)) through the 1.5*2^23 magic constant;
 *        r = x - k*ln2 (Cody-Waite, two fused steps); exp(r) = P5(r)/Q2(r);
 *        result = scalef(P/Q, k); x <= xmin -> 0, x >= xmax -> inf.
 *   sum: chunks of 8192 elements accumulated left to right; inside a chunk
 *        numpy's pairwise scheme (8 interleaved partial sums for blocks of
 *        <= 128 elements, recursive halving above).
 *
 * Pinned: tests/test_oracle_npsimd.py checks these against numpy itself on the
 * machine running the tests (exhaustively for log on [1e-5,4) and exp on
 * [-104.5,0] in this container: 0 mismatches in 1.57e8 / 1.12e9 inputs).
 */
#include <math.h>
#include <stdint.h>

/* ---- log ---------------------------------------------------------------- */
static const float LP0 = 0.000000000000000000000e+00f;
static const float LP1 = 9.999999999999998702752e-01f;
static const float LP2 = 2.112677543073053063722e+00f;
static const float LP3 = 1.480000633576506585156e+00f;
static const float LP4 = 3.808837741388407920751e-01f;
static const float LP5 = 2.589979117907922693523e-02f;
static const float LQ0 = 1.000000000000000000000e+00f;
static const float LQ1 = 2.612677543073109236779e+00f;
static const float LQ2 = 2.453006071784736363091e+00f;
static const float LQ3 = 9.864942958519418960339e-01f;
static const float LQ4 = 1.546476374983906719538e-01f;
static const float LQ5 = 5.875095403124574342950e-03f;
static const float LN2F = 0.693147180559945309417232121458176568f;
static const float RSQRT2F = 0.707106781186547524400844362104849039f;

float npsimd_logf(float v)
{
    if (v != v) return v;
    if (v < 0.0f) return -NAN;
    if (v == 0.0f) return -INFINITY;
    if (isinf(v)) return v;
    int k;
    float m = frexpf(v, &k); /* m in [0.5, 1) */
    float kf = (float)k;
    if (m <= RSQRT2F) {
        m = m + m;
        kf = kf - 1.0f;
    }
    float r = m - 1.0f;
    float num = fmaf(LP5, r, LP4);
    num = fmaf(num, r, LP3);
    num = fmaf(num, r, LP2);
    num = fmaf(num, r, LP1);
    num = fmaf(num, r, LP0);
    float den = fmaf(LQ5, r, LQ4);
    den = fmaf(den, r, LQ3);
    den = fmaf(den, r, LQ2);
    den = fmaf(den, r, LQ1);
    den = fmaf(den, r, LQ0);
    float q = num / den;
    return fmaf(kf, LN2F, q);
}

/* ---- exp ---------------------------------------------------------------- */
static const float EP0 = 9.999999999980870924916e-01f;
static const float EP1 = 7.257664613233124478488e-01f;
static const float EP2 = 2.473615434895520810817e-01f;
static const float EP3 = 5.114512081637298353406e-02f;
static const float EP4 = 6.757896990527504603057e-03f;
static const float EP5 = 5.082762527590693718096e-04f;
static const float EQ0 = 1.000000000000000000000e+00f;
static const float EQ1 = -2.742335390411667452936e-01f;
static const float EQ2 = 2.159509375685829852307e-02f;
static const float CW_HI = -6.93145752e-1f;
static const float CW_LO = -1.42860677e-6f;
static const float LOG2EF = 1.442695040888963407359924681001892137f;
static const float RINT_MAGIC = 0x1.800000p+23f;
static const float EXP_XMAX = 88.72283935546875f;
static const float EXP_XMIN = -103.97208404541015625f;

float npsimd_expf(float v)
{
    if (v != v) return v;
    if (v >= EXP_XMAX) return INFINITY;
    if (v <= EXP_XMIN) return 0.0f;
    float k = v * LOG2EF;
    k = k + RINT_MAGIC;
    k = k - RINT_MAGIC;
    float r = fmaf(k, CW_HI, v);
    r = fmaf(k, CW_LO, r);
    float num = fmaf(EP5, r, EP4);
    num = fmaf(num, r, EP3);
    num = fmaf(num, r, EP2);
    num = fmaf(num, r, EP1);
    num = fmaf(num, r, EP0);
    float den = fmaf(EQ2, r, EQ1);
    den = fmaf(den, r, EQ0);
    float q = num / den;
    return ldexpf(q, (int)k);
}

/* ---- sum ---------------------------------------------------------------- */
static float pairwise_f32(const float *a, long n)
{
    if (n < 8) {
        float res = 0.0f;
        for (long i = 0; i < n; i++) res += a[i];
        return res;
    }
    if (n <= 128) {
        float r[8];
        for (int j = 0; j < 8; j++) r[j] = a[j];
        long nfull = n - (n % 8);
        long i;
        for (i = 8; i < nfull; i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    long half = n / 2;
    half -= half % 8;
    return pairwise_f32(a, half) + pairwise_f32(a + half, n - half);
}

float npsimd_sum_f32(const float *a, long n)
{
    const long chunk = 8192; /* numpy's reduction buffer, in elements */
    float res = 0.0f;
    for (long s = 0; s < n; s += chunk) {
        long m = n - s < chunk ? n - s : chunk;
        res = res + pairwise_f32(a + s, m);
    }
    return res;
}

/* ---- array forms (ctypes entry points) ----------------------------------- */
void npsimd_log_array(const float *in, float *out, long n)
{
    for (long i = 0; i < n; i++) out[i] = npsimd_logf(in[i]);
}

void npsimd_exp_array(const float *in, float *out, long n)
{
    for (long i = 0; i < n; i++) out[i] = npsimd_expf(in[i]);
}

/* row sums of a C-contiguous [rows, cols] float32 matrix */
void npsimd_rowsum_f32(const float *in, float *out, long rows, long cols)
{
    for (long r = 0; r < rows; r++) out[r] = npsimd_sum_f32(in + r * cols, cols);
}

/* softmax over the last axis, as scipy.special.softmax evaluates it in float32:
 * max, exp(x - max), sum, divide. */
void npsimd_softmax_rows(const float *in, float *out, long rows, long cols)
{
    for (long r = 0; r < rows; r++) {
        const float *x = in + r * cols;
        float *y = out + r * cols;
        float mx = x[0];
        for (long c = 1; c < cols; c++) mx = x[c] > mx ? x[c] : mx;
        for (long c = 0; c < cols; c++) y[c] = npsimd_expf(x[c] - mx);
        float s = npsimd_sum_f32(y, cols);
        for (long c = 0; c < cols; c++) y[c] = y[c] / s;
    }
}
