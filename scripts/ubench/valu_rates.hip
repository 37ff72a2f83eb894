// Issue-rate microbenchmark for the VALU instructions of the HOG kernels (gfx950).  Each test runs 8 independent
// chains so that dependent-issue latency does not hide the rate; 1, 2, 4 and 8 waves per SIMD show whether a
// rate is a per-wave issue limit or a per-SIMD throughput limit.
//   cycles/instr (per SIMD) = t * f_clk / (instr per wave * waves per SIMD)
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/bin/valu_rates scripts/ubench/valu_rates.hip
// The output of a run on the round's MI355X box is kept in profiles/ (rNN_ubench_valu_rates.txt).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define ITER 2048
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum {
    T_ADD_F32, T_MUL_F32, T_FMA_F32, T_PK_MUL_F32, T_PK_ADD_F32, T_PK_FMA_F32, T_SQRT_F32, T_RSQ_F32, T_RCP_F32,
    T_CMP_F32, T_CNDMASK, T_CNDMASK_SGPR, T_ADD_U32, T_AND_B32, T_ADD3_U32, T_LSHRREV, T_LSHL_ADD, T_MAD_U24, T_MUL_U24,
    T_MUL_HI_U24, T_MUL_LO_U32, T_PERM, T_DOT2_U16, T_DPP_SHR, T_DPP_ADD, T_CVT_F32_U32, T_CVT_U32_F32, T_CVT_F64_F32, T_CVT_F32_F64,
    T_FMA_F64, T_MUL_F64, T_ADD_F64, T_MIN_F64, T_RSQ_F64, T_RCP_F64, T_SQRT_F64, T_READLANE, T_ADDC, T_BFE, T_XOR, T_ACC_WRITE,
    T_ACC_READ, T_CMP_CND_VCC, T_CMP_CND_SGPR, T_CND_VCC_DEFINED, T_MIX_F32_INT, T_MIX_ADD_SALU, T_COUNT
};

template <int T>
__global__ void k(float* out, float seed)
{
    float a[8]; double d[8]; int n[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; d[i] = a[i] * 0.5; n[i] = (int)a[i]; }
    const double two52 = 4503599627370496.0;
    unsigned long long msk = 0x5555555555555555ull;
    asm volatile("" : "+s"(msk));
    for (int it = 0; it < ITER; ++it) {
#define CASE(TT, ...) if (T == TT) { _Pragma("unroll") for (int r = 0; r < 1; ++r) { __VA_ARGS__ } }
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
        CASE(T_ADD_F32, REP8(X))
#undef X
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
        CASE(T_MUL_F32, REP8(X))
#undef X
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(seed));
        CASE(T_FMA_F32, REP8(X))
#undef X
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(d[i]));
        CASE(T_PK_MUL_F32, REP8(X))
#undef X
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(d[i]));
        CASE(T_PK_ADD_F32, REP8(X))
#undef X
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(d[i]));
        CASE(T_PK_FMA_F32, REP8(X))
#undef X
#define X(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
        CASE(T_SQRT_F32, REP8(X))
#undef X
#define X(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
        CASE(T_RSQ_F32, REP8(X))
#undef X
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        CASE(T_RCP_F32, REP8(X))
#undef X
#define X(i) asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(a[i]), "v"(seed) : "vcc");
        CASE(T_CMP_F32, REP8(X))
#undef X
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(n[i]) : "v"(n[(i + 1) & 7]) : "vcc");
        CASE(T_CNDMASK, REP8(X))
#undef X
// round 3 (VERDICT r02 "CNDMASK 22.5 cycles"): the VCC form with VCC defined -- by a compare right before it, as the kernels use it,
// or once per trip -- against the SGPR-pair form fed by the same compare
#define X(i) asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(n[i]) : "v"(a[i]), "v"(seed) : "vcc");
        CASE(T_CMP_CND_VCC, REP8(X))
#undef X
#define X(i) asm volatile("v_cmp_gt_f32 s[20:21], %1, %2\n\tv_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(n[i]) : "v"(a[i]), "v"(seed) : "s20", "s21");
        CASE(T_CMP_CND_SGPR, REP8(X))
#undef X
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
        if (T == T_CND_VCC_DEFINED) { asm volatile("s_mov_b64 vcc, %0" :: "s"(msk) : "vcc"); REP8(X) }
#undef X
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(n[i]) : "v"(n[(i + 1) & 7]), "s"(msk));
        CASE(T_CNDMASK_SGPR, REP8(X))
#undef X
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
        CASE(T_ADD_U32, REP8(X))
#undef X
#define X(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
        CASE(T_AND_B32, REP8(X))
#undef X
#define X(i) asm volatile("v_add3_u32 %0, %0, %1, 2" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
        CASE(T_ADD3_U32, REP8(X))
#undef X
#define X(i) asm volatile("v_lshrrev_b32 %0, 2, %0" : "+v"(n[i]));
        CASE(T_LSHRREV, REP8(X))
#undef X
#define X(i) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
        CASE(T_LSHL_ADD, REP8(X))
#undef X
#define X(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(n[i]) : "v"(n[(i + 1) & 7]), "v"(n[(i + 2) & 7]));
        CASE(T_MAD_U24, REP8(X))
#undef X
#define X(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
        CASE(T_MUL_U24, REP8(X))
#undef X
#define X(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
        CASE(T_MUL_HI_U24, REP8(X))
#undef X
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
        CASE(T_MUL_LO_U32, REP8(X))
#undef X
#define X(i) asm volatile("v_perm_b32 %0, 0, %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
        CASE(T_PERM, REP8(X))
#undef X
#define X(i) asm volatile("v_dot2_u32_u16 %0, %0, %1, 0" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
        CASE(T_DOT2_U16, REP8(X))
#undef X
#define X(i) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(n[i]));
        CASE(T_DPP_SHR, REP8(X))
#undef X
#define X(i) asm volatile("v_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(seed));
        CASE(T_DPP_ADD, REP8(X))
#undef X
#define X(i) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a[i]) : "v"(n[i]));
        CASE(T_CVT_F32_U32, REP8(X))
#undef X
#define X(i) asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(n[i]) : "v"(a[i]));
        CASE(T_CVT_U32_F32, REP8(X))
#undef X
#define X(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
        CASE(T_CVT_F64_F32, REP8(X))
#undef X
#define X(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
        CASE(T_CVT_F32_F64, REP8(X))
#undef X
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(two52), "v"(two52));
        CASE(T_FMA_F64, REP8(X))
#undef X
#define X(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(two52));
        CASE(T_MUL_F64, REP8(X))
#undef X
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(two52));
        CASE(T_ADD_F64, REP8(X))
#undef X
#define X(i) asm volatile("v_min_f64 %0, %0, %1" : "+v"(d[i]) : "v"(two52));
        CASE(T_MIN_F64, REP8(X))
#undef X
#define X(i) asm volatile("v_rsq_f64 %0, %0" : "+v"(d[i]));
        CASE(T_RSQ_F64, REP8(X))
#undef X
#define X(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
        CASE(T_RCP_F64, REP8(X))
#undef X
#define X(i) asm volatile("v_sqrt_f64 %0, %0" : "+v"(d[i]));
        CASE(T_SQRT_F64, REP8(X))
#undef X
#define X(i) asm volatile("v_readlane_b32 s20, %0, 3" :: "v"(n[i]) : "s20");
        CASE(T_READLANE, REP8(X))
#undef X
#define X(i) asm volatile("v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(n[i]) :: "vcc");
        CASE(T_ADDC, REP8(X))
#undef X
#define X(i) asm volatile("v_bfe_u32 %0, %0, 4, 19" : "+v"(n[i]));
        CASE(T_BFE, REP8(X))
#undef X
#define X(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
        CASE(T_XOR, REP8(X))
#undef X
#define X(i) asm volatile("v_accvgpr_write_b32 a" #i ", %0" :: "v"(n[i]) : "a" #i);
        CASE(T_ACC_WRITE, REP8(X))
#undef X
#define X(i) asm volatile("v_accvgpr_read_b32 %0, a" #i : "=v"(n[i]));
        CASE(T_ACC_READ, REP8(X))
#undef X
        // alternating f32 arithmetic and integer logic (does the pairing matter?)
#define X(i) asm volatile("v_add_f32 %0, %0, %2\n\tv_and_b32 %1, %1, %3" : "+v"(a[i]), "+v"(n[i]) : "v"(seed), "v"(n[(i + 1) & 7]));
        CASE(T_MIX_F32_INT, REP8(X))
#undef X
        // one SALU instruction per VALU instruction (co-issue from the same wave / other waves)
#define X(i) asm volatile("v_add_f32 %0, %0, %1\n\ts_add_u32 s20, s20, 1" : "+v"(a[i]) : "v"(seed) : "s20", "scc");
        CASE(T_MIX_ADD_SALU, REP8(X))
#undef X
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + (float)d[i] + n[i];
    if (s == 12345.678f) out[0] = s;
}

template <int T> void run(const char* name, float* d_out, int instr_per_x)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-22s", name);
    for (int w = 1; w <= 8; w *= 2) {
        const int blocks = 256 * 4 * w;
        k<T><<<blocks, 64>>>(d_out, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<T><<<blocks, 64>>>(d_out, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double cyc = ms * 1e-3 * 2.4e9 / ((double)ITER * 8 * instr_per_x * w);
        printf("  w%d %6.2f", w, cyc);
    }
    printf("   cycles per wave-instruction per SIMD @2.4 GHz (w = waves per SIMD)\n");
}

int main()
{
    float* d; hipMalloc(&d, 4);
#define R(T) run<T>(#T + 2, d, 1)
    R(T_ADD_F32); R(T_MUL_F32); R(T_FMA_F32); R(T_PK_MUL_F32); R(T_PK_ADD_F32); R(T_PK_FMA_F32); R(T_SQRT_F32); R(T_RSQ_F32); R(T_RCP_F32);
    R(T_CMP_F32); R(T_CNDMASK); R(T_CNDMASK_SGPR); R(T_ADD_U32); R(T_AND_B32); R(T_ADD3_U32); R(T_LSHRREV); R(T_LSHL_ADD); R(T_MAD_U24);
    R(T_MUL_U24); R(T_MUL_HI_U24); R(T_MUL_LO_U32); R(T_PERM); R(T_DOT2_U16); R(T_DPP_SHR); R(T_DPP_ADD); R(T_CVT_F32_U32); R(T_CVT_U32_F32);
    R(T_CVT_F64_F32); R(T_CVT_F32_F64); R(T_FMA_F64); R(T_MUL_F64); R(T_ADD_F64); R(T_MIN_F64); R(T_RSQ_F64); R(T_RCP_F64); R(T_SQRT_F64);
    R(T_READLANE); R(T_ADDC); R(T_BFE); R(T_XOR); R(T_ACC_WRITE); R(T_ACC_READ);
    run<T_CMP_CND_VCC>("CMP+CNDMASK via vcc (2 insts)", d, 2); run<T_CMP_CND_SGPR>("CMP+CNDMASK via s[20:21] (2 insts)", d, 2); R(T_CND_VCC_DEFINED);
    run<T_MIX_F32_INT>("MIX add_f32+and_b32", d, 2);
    run<T_MIX_ADD_SALU>("MIX add_f32+s_add", d, 1);
    return 0;
}
