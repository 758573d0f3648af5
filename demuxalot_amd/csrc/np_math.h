// np_math.h -- device-side float32 building blocks with numpy's exact roundings.
//
// The reference evaluates log / exp / row sums with numpy on float32 arrays
// (demuxalot/demux.py:261 np.log; :101,:152 scipy softmax = np.exp + np.sum).  numpy's
// x86 float32 kernels are rational minimax approximations evaluated with fused
// multiply-adds (published in numpy's loops_exponent_log.dispatch.c.src); they are not
// correctly rounded (up to 3.83 ulp for log), so matching the reference bit for bit
// means repeating the same operation sequence.  gfx950 has everything needed in
// hardware: v_frexp_mant/exp, v_fma_f32, IEEE float32 division, v_ldexp_f32.
//
// Compile with -ffp-contract=off: every fused step below is an explicit fmaf and the
// compiler must not add or remove fusions.
#pragma once
#include <hip/hip_runtime.h>

namespace npm {

// Correctly rounded num / den for the operands that occur inside log_f32: den = Q(r) lies in
// [0.4, 2.7] and |num| < 1, so no range scaling / fix-up is needed: reciprocal estimate, one
// Newton step, quotient, exact remainder, one correction (Markstein's sequence).  Because num
// and den are functions of the 2^24 possible reduced arguments only, the equality with IEEE
// division is checked EXHAUSTIVELY on the device (tests/test_gpu_parity.py::test_device_log_*).
__device__ __forceinline__ float div_small_range(float num, float den)
{
    float rc = __builtin_amdgcn_rcpf(den);
    const float e0 = __builtin_fmaf(-den, rc, 1.0f);
    rc = __builtin_fmaf(e0, rc, rc);
    const float q0 = num * rc;
    const float rem = __builtin_fmaf(-den, q0, num);
    return __builtin_fmaf(rem, rc, q0);
}

// SPECIALS = false is the hot-path form: its argument is finite and >= 1e-4 by construction of
// the E-step term (a NaN input still comes out NaN through the arithmetic); FASTDIV selects
// div_small_range instead of the compiler's generic IEEE division sequence.
template <bool SPECIALS = true, bool FASTDIV = false>
__device__ __forceinline__ float log_f32(float v)
{
    const float P0 = 0.000000000000000000000e+00f, P1 = 9.999999999999998702752e-01f,
                P2 = 2.112677543073053063722e+00f, P3 = 1.480000633576506585156e+00f,
                P4 = 3.808837741388407920751e-01f, P5 = 2.589979117907922693523e-02f;
    const float Q0 = 1.000000000000000000000e+00f, Q1 = 2.612677543073109236779e+00f,
                Q2 = 2.453006071784736363091e+00f, Q3 = 9.864942958519418960339e-01f,
                Q4 = 1.546476374983906719538e-01f, Q5 = 5.875095403124574342950e-03f;
    const float LN2 = 0.693147180559945309417232121458176568f;
    const float RSQRT2 = 0.707106781186547524400844362104849039f;

    float m = __builtin_amdgcn_frexp_mantf(v);  // [0.5, 1)
    float kf = (float)__builtin_amdgcn_frexp_expf(v);
    const bool low = m <= RSQRT2;
    m = low ? m + m : m;
    kf = low ? kf - 1.0f : kf;
    const float r = m - 1.0f;
    float num = __builtin_fmaf(P5, r, P4);
    num = __builtin_fmaf(num, r, P3);
    num = __builtin_fmaf(num, r, P2);
    num = __builtin_fmaf(num, r, P1);
    num = __builtin_fmaf(num, r, P0);
    float den = __builtin_fmaf(Q5, r, Q4);
    den = __builtin_fmaf(den, r, Q3);
    den = __builtin_fmaf(den, r, Q2);
    den = __builtin_fmaf(den, r, Q1);
    den = __builtin_fmaf(den, r, Q0);
    // IEEE-754 division (-fhip-fp32-correctly-rounded-divide-sqrt) or its range-restricted equal
    const float q = FASTDIV ? div_small_range(num, den) : num / den;
    float res = __builtin_fmaf(kf, LN2, q);
    if (SPECIALS) {  // special values, as numpy returns them
        res = (v == 0.0f) ? -__builtin_inff() : res;
        res = (v < 0.0f) ? -__builtin_nanf("") : res;
        res = (v != v || v == __builtin_inff()) ? v : res;
    }
    return res;
}

// ---------------------------------------------------------------------------------------
// Hot-path form of log_f32 for the E-step: TWO independent positive finite arguments per lane,
// evaluated with packed float32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 issue
// two lanes' worth of work per instruction slot; the E-step is VALU-issue bound).
// Same roundings as log_f32, operation by operation:
//   * range reduction on the bit pattern: subtracting the bits of nextafter(1/sqrt(2)) splits
//     x = m' * 2^k with m' in (1/sqrt2, sqrt2] in 5 integer ops; m' and k are exactly the values
//     log_f32 derives from frexp + compare + select (both are exact operations, so equality of
//     the results is a matter of integer arithmetic; checked exhaustively on the device);
//   * P5(r), Q5(r): the two Horner chains of both arguments as packed FMAs;
//   * division: div_small_range, packed.
// ---------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));

static __device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

__device__ __forceinline__ f32x2 log_f32_hot2(f32x2 v)
{
    const float P0 = 0.000000000000000000000e+00f, P1 = 9.999999999999998702752e-01f,
                P2 = 2.112677543073053063722e+00f, P3 = 1.480000633576506585156e+00f,
                P4 = 3.808837741388407920751e-01f, P5 = 2.589979117907922693523e-02f;
    const float Q0 = 1.000000000000000000000e+00f, Q1 = 2.612677543073109236779e+00f,
                Q2 = 2.453006071784736363091e+00f, Q3 = 9.864942958519418960339e-01f,
                Q4 = 1.546476374983906719538e-01f, Q5 = 5.875095403124574342950e-03f;
    const float LN2 = 0.693147180559945309417232121458176568f;
    const int SPLIT = 0x3f3504f4;  // bits of the float32 just above 1/sqrt(2) (0x3f3504f3)

    // x = m' * 2^k with m' in (1/sqrt2, sqrt2]: d = bits - SPLIT = k * 2^23 + fraction, so the top 9 bits of d are
    // k * 2^23 itself (as an integer), m' = bits - k * 2^23, and float(k * 2^23) is exact (|k| < 2^8).  The factor 2^23
    // is taken out of LN2 in the final fma: fma(k * 2^23, LN2 * 2^-23, q) rounds the same real number as
    // fma(k, LN2, q) (both scalings are exact).  Four plain integer ops and one conversion per argument.
    const int xa = __float_as_int(v.x), xb = __float_as_int(v.y);
    const int ha = (xa - SPLIT) & (int)0xFF800000, hb = (xb - SPLIT) & (int)0xFF800000;
    f32x2 k, m;
    k.x = (float)ha;
    k.y = (float)hb;
    m.x = __int_as_float(xa - ha);
    m.y = __int_as_float(xb - hb);
    const f32x2 r = m - 1.0f;
    f32x2 num = pk_fma((f32x2)(P5), r, (f32x2)(P4));
    f32x2 den = pk_fma((f32x2)(Q5), r, (f32x2)(Q4));
    num = pk_fma(num, r, (f32x2)(P3));
    den = pk_fma(den, r, (f32x2)(Q3));
    num = pk_fma(num, r, (f32x2)(P2));
    den = pk_fma(den, r, (f32x2)(Q2));
    num = pk_fma(num, r, (f32x2)(P1));
    den = pk_fma(den, r, (f32x2)(Q1));
    num = pk_fma(num, r, (f32x2)(P0));
    den = pk_fma(den, r, (f32x2)(Q0));
    f32x2 rc;
    rc.x = __builtin_amdgcn_rcpf(den.x);
    rc.y = __builtin_amdgcn_rcpf(den.y);
    const f32x2 nden = -den;
#if defined(DMX_DIV_REFINE)
    const f32x2 e0 = pk_fma(nden, rc, (f32x2)(1.0f));
    rc = pk_fma(e0, rc, rc);
#endif
    // quotient estimate, exact remainder, one correction.  With the 1-ulp hardware reciprocal the
    // corrected quotient equals the IEEE quotient for every (num, den) that the 2^24 reduced
    // arguments can produce -- checked exhaustively on the device (test_device_log_hot_path_exhaustive)
    const f32x2 q0 = num * rc;
    const f32x2 rem = pk_fma(nden, q0, num);
    const f32x2 q = pk_fma(rem, rc, q0);
    return pk_fma(k, (f32x2)(LN2 * 1.1920928955078125e-07f), q);  // LN2 * 2^-23, exact
}

__device__ __forceinline__ float exp_f32(float v)
{
    const float P0 = 9.999999999980870924916e-01f, P1 = 7.257664613233124478488e-01f,
                P2 = 2.473615434895520810817e-01f, P3 = 5.114512081637298353406e-02f,
                P4 = 6.757896990527504603057e-03f, P5 = 5.082762527590693718096e-04f;
    const float Q0 = 1.000000000000000000000e+00f, Q1 = -2.742335390411667452936e-01f,
                Q2 = 2.159509375685829852307e-02f;
    const float CW_HI = -6.93145752e-1f, CW_LO = -1.42860677e-6f;
    const float LOG2E = 1.442695040888963407359924681001892137f;
    const float MAGIC = 0x1.800000p+23f;
    const float XMAX = 88.72283935546875f, XMIN = -103.97208404541015625f;

    const bool too_big = v >= XMAX, too_small = v <= XMIN, is_nan = v != v;
    const float x = (too_big || too_small || is_nan) ? 0.0f : v;
    float k = x * LOG2E;
    k = k + MAGIC;  // round to nearest integer through the 1.5*2^23 constant
    k = k - MAGIC;
    float r = __builtin_fmaf(k, CW_HI, x);
    r = __builtin_fmaf(k, CW_LO, r);
    float num = __builtin_fmaf(P5, r, P4);
    num = __builtin_fmaf(num, r, P3);
    num = __builtin_fmaf(num, r, P2);
    num = __builtin_fmaf(num, r, P1);
    num = __builtin_fmaf(num, r, P0);
    float den = __builtin_fmaf(Q2, r, Q1);
    den = __builtin_fmaf(den, r, Q0);
    const float q = num / den;
    float res = __builtin_amdgcn_ldexpf(q, (int)k);  // v_ldexp_f32: one rounding, subnormals kept
    res = too_small ? 0.0f : res;
    res = too_big ? __builtin_inff() : res;
    res = is_nan ? v : res;
    return res;
}

// ---------------------------------------------------------------------------------------
// np.sum over a contiguous float32 row, numpy's association:
//   chunks of 8192 elements added left to right; a chunk is summed pairwise: blocks of
//   <= 128 elements use 8 interleaved partial sums r[j] (j = i mod 8) combined as
//   ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) followed by the <8 leftover elements in order;
//   longer blocks split at n/2 rounded down to a multiple of 8.
// ---------------------------------------------------------------------------------------

// Leaf (8 <= n <= 128) summed by lanes 0..7 of the calling wave in parallel; a[] may be LDS
// or global.  All 64 lanes must call; the result is valid in lanes 0..7.
__device__ __forceinline__ float leaf_sum_wave8(const float *a, int n, int lane)
{
    const int j = lane & 7;
    const int nfull = n - (n & 7);
    float r = a[j];
    for (int i = 8 + j; i < nfull; i += 8) r += a[i];
    // IEEE addition is commutative, so xor-butterflies give exactly the bracketed tree
    r = r + __shfl_xor(r, 1);
    r = r + __shfl_xor(r, 2);
    r = r + __shfl_xor(r, 4);
    for (int i = nfull; i < n; i++) r += a[i];
    return r;
}

// Serial version for tiny rows (n < 8): starts from +0 like numpy.
__device__ __forceinline__ float short_sum(const float *a, int n)
{
    float res = 0.0f;
    for (int i = 0; i < n; i++) res += a[i];
    return res;
}

// Pairwise sum of a[0..n) for n <= 8192 executed by one whole wave (uniform control flow).
// Explicit stack instead of recursion: depth <= 7 for n <= 8192.
__device__ __forceinline__ float chunk_sum_wave(const float *a, int n, int lane)
{
    if (n < 8) return short_sum(a, n);
    if (n <= 128) return leaf_sum_wave8(a, n, lane);
    // post-order traversal of the split tree
    int st_start[16], st_len[16];
    unsigned char st_state[16];
    float st_left[16];
    int sp = 0;
    st_start[0] = 0;
    st_len[0] = n;
    st_state[0] = 0;
    float ret = 0.0f;
    while (sp >= 0) {
        const int s = st_start[sp], len = st_len[sp];
        if (len <= 128) {
            ret = (len < 8) ? short_sum(a + s, len) : leaf_sum_wave8(a + s, len, lane);
            sp--;
            continue;
        }
        int half = len / 2;
        half -= half % 8;
        if (st_state[sp] == 0) {
            st_state[sp] = 1;
            sp++;
            st_start[sp] = s;
            st_len[sp] = half;
            st_state[sp] = 0;
        } else if (st_state[sp] == 1) {
            st_left[sp] = ret;
            st_state[sp] = 2;
            sp++;
            st_start[sp] = s + half;
            st_len[sp] = len - half;
            st_state[sp] = 0;
        } else {
            ret = st_left[sp] + ret;
            sp--;
        }
    }
    return ret;
}

// Full np.sum(row) by one wave.
__device__ __forceinline__ float row_sum_wave(const float *a, int n, int lane)
{
    float res = 0.0f;
    for (int s = 0; s < n; s += 8192) {
        const int m = (n - s) < 8192 ? (n - s) : 8192;
        res = res + chunk_sum_wave(a + s, m, lane);
    }
    return res;
}


// ---------------------------------------------------------------------------------------
// np.sum over a row of n values in LDS by a whole 256-thread workgroup, along a PLAN the host spelled out for this n
// (dmx_api.cpp: ensure_sum_plan): {n_leaves, n_levels, n_roots, level offsets [n_levels + 1], leaves (start, length),
// inner nodes (left value, right value) level by level from the deepest, roots of the 8192-element chunks}.
// Blocks of <= 128 elements are summed by groups of 8 lanes (32 at a time), the inner nodes of numpy's pairwise tree
// level by level, the chunk roots left to right: the same additions in the same association as row_sum_wave /
// np.add.reduce, for float32 and float64 alike (numpy blocks both by 128 elements and 8 partial sums).
// val: LDS scratch of n_leaves + n_inner values; the result is returned to every thread.  Uniform control flow.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float plan_shfl_xor(float v, int m) { return __shfl_xor(v, m); }
__device__ __forceinline__ double plan_shfl_xor(double v, int m)
{
    const int lo = __shfl_xor(__double2loint(v), m), hi = __shfl_xor(__double2hiint(v), m);
    return __hiloint2double(hi, lo);
}

template <typename T>
__device__ __forceinline__ T plan_sum_block(const T *x, const int *__restrict__ plan, T *val, int tid)
{
    const int n_leaves = plan[0], n_levels = plan[1], n_roots = plan[2];
    const int *__restrict__ level_off = plan + 3;
    const int *__restrict__ leaves = level_off + n_levels + 1;
    const int *__restrict__ nodes = leaves + 2 * n_leaves;
    const int *__restrict__ roots = nodes + 2 * level_off[n_levels];
    const int j = tid & 7;
    for (int l = tid >> 3; l < n_leaves; l += 32) {
        const T *blk = x + leaves[2 * l];
        const int n = leaves[2 * l + 1];
        T r = (T)0;
        if (n < 8) {  // short_sum
            for (int i = 0; i < n; i++) r += blk[i];
        } else {      // leaf_sum_wave8
            const int nfull = n - (n & 7);
            r = blk[j];
            for (int i = 8 + j; i < nfull; i += 8) r += blk[i];
            r = r + plan_shfl_xor(r, 1);
            r = r + plan_shfl_xor(r, 2);
            r = r + plan_shfl_xor(r, 4);
            for (int i = nfull; i < n; i++) r += blk[i];
        }
        if (j == 0) val[l] = r;
    }
    __syncthreads();
    for (int h = 0; h < n_levels; h++) {
        for (int q = level_off[h] + tid; q < level_off[h + 1]; q += 256) val[n_leaves + q] = val[nodes[2 * q]] + val[nodes[2 * q + 1]];
        __syncthreads();
    }
    T res = (T)0;
    for (int r = 0; r < n_roots; r++) res = res + val[roots[r]];  // every thread: a handful of LDS broadcasts
    __syncthreads();                                              // val may be reused right away
    return res;
}

}  // namespace npm
