// valu_issue_bench.hip -- issue cost of the VALU instructions the E-step is made of, on gfx950.
//
//   hipcc -O2 --offload-arch=gfx950 scripts/valu_issue_bench.hip -o gpurun_out/valu_issue_bench
//   gpurun_out/valu_issue_bench > profiles/r2_valu_issue_bench.txt
//
// For every instruction: w workgroups of 256 threads per CU (= w waves per SIMD, w = 1, 2, 4, 8; the kernel is
// built for 8 waves per SIMD), each wave runs ITER x 64 independent instances of the instruction
// (8 register chains, so dependent-issue latency never binds) between two s_memtime stamps.
// Reported: shader cycles per wave-instruction PER SIMD = wave cycles / (w * instructions per wave),
// median over the waves of the launch.  2.0 means the SIMD retires one wave64 instruction every 2
// cycles (the 32-lane-per-cycle rate), 4.0 one every 4 cycles.  In parentheses: the same from the whole launch
// (first wave start to last wave end), an upper bound that does not depend on all waves running side by side.
// The last line of every block checks that premise: `conc` = median wave duration / span of the launch in
// s_memrealtime ticks, `clk` = the shader clock the waves saw.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));

enum Op {
    FMA_F32, MUL_F32, ADD_F32, PK_FMA_F32, PK_MUL_F32, PK_ADD_F32, ADD_F64, FMA_F64, CVT_F64_F32, CVT_F32_F64,
    RCP_F32, LOG_F32, EXP_F32, MAD_I24, ADD_U32, ASHR_I32, CVT_F32_I32, LDEXP_F32, FREXP_MANT, FREXP_EXP,
    CVT_I32_F32, ADD_CO_U32, MOV_B32, MIX_E_STEP, N_OPS
};
static const char *NAMES[N_OPS] = {
    "v_fma_f32", "v_mul_f32", "v_add_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_add_f64", "v_fma_f64",
    "v_cvt_f64_f32", "v_cvt_f32_f64", "v_rcp_f32", "v_log_f32", "v_exp_f32", "v_mad_i32_i24", "v_add_u32",
    "v_ashrrev_i32", "v_cvt_f32_i32", "v_ldexp_f32", "v_frexp_mant_f32", "v_frexp_exp_i32_f32", "v_cvt_i32_f32",
    "v_add_co_u32+v_addc_co_u32 (pair)", "v_mov_b32",
    "E-step mix per 2 terms: 17 pk + 8 int/cvt + 2 rcp + 2 cvt_f64 + 2 add_f64 (31 instr)"};

// eight instances of one instruction over eight independent register chains
#define R8(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)

template <int OP>
__device__ __forceinline__ void body(float (&a)[8], f32x2 (&p)[8], double (&d)[8], int (&i)[8], float s, double sd)
{
    // every case issues 64 instructions (MIX: 62)
#pragma unroll
    for (int rep = 0; rep < 8; rep++) {
        if constexpr (OP == FMA_F32) {
#define INS(k) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[k]) : "v"(s));
            R8(INS)
#undef INS
        } else if constexpr (OP == MUL_F32) {
#define INS(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(s));
            R8(INS)
#undef INS
        } else if constexpr (OP == ADD_F32) {
#define INS(k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(s));
            R8(INS)
#undef INS
        } else if constexpr (OP == PK_FMA_F32) {
#define INS(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[k]) : "v"(p[(k + 1) & 7]));
            R8(INS)
#undef INS
        } else if constexpr (OP == PK_MUL_F32) {
#define INS(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(p[(k + 1) & 7]));
            R8(INS)
#undef INS
        } else if constexpr (OP == PK_ADD_F32) {
#define INS(k) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k]) : "v"(p[(k + 1) & 7]));
            R8(INS)
#undef INS
        } else if constexpr (OP == ADD_F64) {
#define INS(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[k]) : "v"(sd));
            R8(INS)
#undef INS
        } else if constexpr (OP == FMA_F64) {
#define INS(k) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[k]) : "v"(sd));
            R8(INS)
#undef INS
        } else if constexpr (OP == CVT_F64_F32) {
#define INS(k) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[k]) : "v"(a[k]));
            R8(INS)
#undef INS
        } else if constexpr (OP == CVT_F32_F64) {
#define INS(k) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[k]) : "v"(d[k]));
            R8(INS)
#undef INS
        } else if constexpr (OP == RCP_F32) {
#define INS(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
            R8(INS)
#undef INS
        } else if constexpr (OP == LOG_F32) {
#define INS(k) asm volatile("v_log_f32 %0, %0" : "+v"(a[k]));
            R8(INS)
#undef INS
        } else if constexpr (OP == EXP_F32) {
#define INS(k) asm volatile("v_exp_f32 %0, %0" : "+v"(a[k]));
            R8(INS)
#undef INS
        } else if constexpr (OP == MAD_I24) {
#define INS(k) asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(i[k]) : "v"(i[(k + 1) & 7]));
            R8(INS)
#undef INS
        } else if constexpr (OP == ADD_U32) {
#define INS(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(i[k]) : "v"(i[(k + 1) & 7]));
            R8(INS)
#undef INS
        } else if constexpr (OP == ASHR_I32) {
#define INS(k) asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(i[k]));
            R8(INS)
#undef INS
        } else if constexpr (OP == CVT_F32_I32) {
#define INS(k) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a[k]) : "v"(i[k]));
            R8(INS)
#undef INS
        } else if constexpr (OP == LDEXP_F32) {
#define INS(k) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a[k]) : "v"(i[k]));
            R8(INS)
#undef INS
        } else if constexpr (OP == FREXP_MANT) {
#define INS(k) asm volatile("v_frexp_mant_f32 %0, %0" : "+v"(a[k]));
            R8(INS)
#undef INS
        } else if constexpr (OP == FREXP_EXP) {
#define INS(k) asm volatile("v_frexp_exp_i32_f32 %0, %1" : "=v"(i[k]) : "v"(a[k]));
            R8(INS)
#undef INS
        } else if constexpr (OP == CVT_I32_F32) {
#define INS(k) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(i[k]) : "v"(a[k]));
            R8(INS)
#undef INS
        } else if constexpr (OP == ADD_CO_U32) {
            // four 64-bit integer adds = 8 instructions
#define INS(k) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(i[2 * k]), "+v"(i[2 * k + 1]) : "v"(i[(2 * k + 3) & 7]) : "vcc");
            INS(0) INS(1) INS(2) INS(3)
#undef INS
        } else if constexpr (OP == MOV_B32) {
#define INS(k) asm volatile("v_mov_b32 %0, %1" : "=v"(i[k]) : "v"(i[(k + 1) & 7]));
            R8(INS)
#undef INS
        } else if constexpr (OP == MIX_E_STEP) {
            // the instruction mix of estep_terms + log_f32_hot2 for one pair of calls, twice (62 instructions):
            // 17 packed f32, 2 sub + 2 ashr + 2 cvt_f32_i32 + 2 mad_i24, 2 rcp, 2 cvt_f64_f32, 2 add_f64
            if (rep < 2) {
#pragma unroll
                for (int j = 0; j < 17; j++) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[j & 7]) : "v"(p[(j + 1) & 7]));
                asm volatile("v_sub_u32 %0, %0, %1" : "+v"(i[0]) : "v"(i[2]));
                asm volatile("v_sub_u32 %0, %0, %1" : "+v"(i[1]) : "v"(i[3]));
                asm volatile("v_ashrrev_i32 %0, 23, %1" : "=v"(i[4]) : "v"(i[0]));
                asm volatile("v_ashrrev_i32 %0, 23, %1" : "=v"(i[5]) : "v"(i[1]));
                asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a[0]) : "v"(i[4]));
                asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a[1]) : "v"(i[5]));
                asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(i[6]) : "v"(i[4]), "v"(i[2]));
                asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(i[7]) : "v"(i[5]), "v"(i[3]));
                asm volatile("v_rcp_f32 %0, %1" : "=v"(a[2]) : "v"(a[4]));
                asm volatile("v_rcp_f32 %0, %1" : "=v"(a[3]) : "v"(a[5]));
                asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[0]) : "v"(a[6]));
                asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[1]) : "v"(a[7]));
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[2]) : "v"(d[0]));
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[3]) : "v"(d[1]));
            }
        }
    }
}

template <int OP>
__global__ __launch_bounds__(256, 8) void k_issue(unsigned long long *cycles, float *sink, int iters, float s)
{
    float a[8];
    f32x2 p[8];
    double d[8];
    int i[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        a[k] = 1.0f + 0.001f * (float)(threadIdx.x + k);
        p[k] = f32x2{a[k], a[k] * 0.5f};
        d[k] = (double)a[k];
        i[k] = (int)threadIdx.x + k;
    }
    __syncthreads();
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0), "=s"(t0)::"memory");
    for (int it = 0; it < iters; it++) body<OP>(a, p, d, i, s, (double)s);
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; k++) acc += a[k] + p[k].x + p[k].y + (float)d[k] + (float)i[k];
    if (acc == 123.456f) sink[0] = acc;  // keeps everything live
    if ((threadIdx.x & 63) == 0) {
        const size_t wv = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        cycles[3 * wv] = t1 - t0;
        cycles[3 * wv + 1] = r0;
        cycles[3 * wv + 2] = r1;
    }
}

template <int OP>
static int run_one(unsigned long long *d_cycles, float *d_sink, int n_cu)
{
    const int iters = 3000;
    int per_iter = 64;
    if (OP == MIX_E_STEP) per_iter = 62;
    printf("%-44s", NAMES[OP]);
    char check[256];
    int pos = 0;
    for (int w : {1, 2, 4, 8}) {
        const int threads = 256;
        const int blocks = n_cu * w;
        const int waves = blocks * threads / 64;
        hipLaunchKernelGGL(k_issue<OP>, dim3(blocks), dim3(threads), 0, 0, d_cycles, d_sink, iters, 1.0000001f);  // warm
        hipLaunchKernelGGL(k_issue<OP>, dim3(blocks), dim3(threads), 0, 0, d_cycles, d_sink, iters, 1.0000001f);
        CHECK(hipDeviceSynchronize());
        std::vector<unsigned long long> raw(3 * (size_t)waves), h(waves), dur(waves);
        CHECK(hipMemcpy(raw.data(), d_cycles, sizeof(unsigned long long) * 3 * waves, hipMemcpyDeviceToHost));
        unsigned long long first = ~0ull, last = 0;
        for (int k = 0; k < waves; k++) {
            h[k] = raw[3 * k];
            dur[k] = raw[3 * k + 2] - raw[3 * k + 1];
            first = std::min(first, raw[3 * k + 1]);
            last = std::max(last, raw[3 * k + 2]);
        }
        std::sort(h.begin(), h.end());
        std::sort(dur.begin(), dur.end());
        const double med = (double)h[waves / 2];
        const double clk = med / (double)dur[waves / 2];  // shader cycles per 100 MHz tick
        // whole-launch figure: every SIMD retired w * iters * per_iter wave-instructions between the first start and the
        // last end of the launch (upper bound of the cost: includes the ramp-up and the tail)
        const double agg = (double)(last - first) * clk / ((double)w * iters * per_iter);
        printf("  w=%d: %5.2f (%5.2f)", w, med / ((double)w * iters * per_iter), agg);
        pos += snprintf(check + pos, sizeof check - pos, "  w=%d: conc %.2f clk %.2f GHz", w, (double)dur[waves / 2] / (double)(last - first),
                        clk * 0.1);
    }
    printf("   cycles per wave-instruction per SIMD\n%-44s%s\n", "", check);
    return 0;
}

template <int OP>
static int run_all(unsigned long long *d_cycles, float *d_sink, int n_cu)
{
    if (run_one<OP>(d_cycles, d_sink, n_cu)) return 1;
    if constexpr (OP + 1 < N_OPS) return run_all<OP + 1>(d_cycles, d_sink, n_cu);
    return 0;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("# %s, %d CUs, clock %d MHz; w workgroups of 256 threads per CU\n", prop.gcnArchName, n_cu, prop.clockRate / 1000);
    unsigned long long *d_cycles;
    float *d_sink;
    CHECK(hipMalloc(&d_cycles, sizeof(unsigned long long) * 3 * n_cu * 8 * 4));
    CHECK(hipMalloc(&d_sink, 64));
    return run_all<0>(d_cycles, d_sink, n_cu);
}
