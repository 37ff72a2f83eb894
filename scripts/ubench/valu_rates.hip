// Issue-rate microbenchmark for the VALU instructions of the HOG row loop (gfx950).  Each test runs 8 independent
// chains so that dependent-issue latency does not hide the rate.  cycles/instr = t * f_clk / (instr per wave * waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define ITER 4096
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int T>
__global__ void k(float* out, float seed)
{
    float a[8]; double d[8]; int n[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; d[i] = a[i] * 0.5; n[i] = (int)a[i]; }
    const double two52 = 4503599627370496.0;
    for (int it = 0; it < ITER; ++it) {
        if (T == 0) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
            REP8(X)
#undef X
        } else if (T == 1) {
#define X(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
            REP8(X)
#undef X
        } else if (T == 2) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(two52), "v"(two52));
            REP8(X)
#undef X
        } else if (T == 3) {
#define X(i) asm volatile("v_sqrt_f32 %0, %1" : "=v"(a[i]) : "v"(a[i]));
            REP8(X)
#undef X
        } else if (T == 4) {
#define X(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
            REP8(X)
#undef X
        } else if (T == 5) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(n[i]) : "v"(n[i]));
            REP8(X)
#undef X
        } else if (T == 6) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(n[i]) : "v"(n[(i + 1) & 7]) : "vcc");
            REP8(X)
#undef X
        } else if (T == 7) {
#define X(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
            REP8(X)
#undef X
        } else if (T == 8) {
#define X(i) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a[i]) : "v"(n[i]));
            REP8(X)
#undef X
        } else if (T == 9) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
            REP8(X)
#undef X
        } else if (T == 10) {
#define X(i) asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(a[i]), "v"(seed) : "vcc");
            REP8(X)
#undef X
        } else if (T == 11) {
#define X(i) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(n[i]) : "v"(n[(i + 1) & 7]), "v"(n[(i + 2) & 7]));
            REP8(X)
#undef X
        } else if (T == 12) {
#define X(i) asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(n[i]) : "v"(a[i]));
            REP8(X)
#undef X
        } else if (T == 13) {
#define X(i) asm volatile("v_readlane_b32 s20, %0, 3" :: "v"(n[i]) : "s20");
            REP8(X)
#undef X
        } else if (T == 14) {
#define X(i) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(d[i]));
            REP8(X)
#undef X
        } else if (T == 15) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(d[i]));
            REP8(X)
#undef X
        } else if (T == 16) {
#define X(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(two52));
            REP8(X)
#undef X
        } else if (T == 17) {
#define X(i) asm volatile("v_rsq_f32 %0, %1" : "=v"(a[i]) : "v"(a[i]));
            REP8(X)
#undef X
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + (float)d[i] + n[i];
    if (s == 12345.678f) out[0] = s;
}

template <int T> void run(const char* name, float* d_out, int waves_per_simd)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 4 * waves_per_simd;
    k<T><<<blocks, 64>>>(d_out, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<T><<<blocks, 64>>>(d_out, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * 2.4e9 / ((double)ITER * 8 * waves_per_simd);
    printf("%-18s waves/SIMD %d: %.3f ms -> %.2f cycles/instr @2.4GHz\n", name, waves_per_simd, ms, cyc);
}

int main()
{
    float* d; hipMalloc(&d, 4);
    for (int w = 1; w <= 4; w *= 4) {
        run<0>("v_add_f32", d, w); run<9>("v_mul_f32", d, w); run<1>("v_cvt_f64_f32", d, w); run<2>("v_fma_f64", d, w); run<16>("v_mul_f64", d, w);
        run<3>("v_sqrt_f32", d, w); run<17>("v_rsq_f32", d, w); run<4>("v_mul_u32_u24", d, w); run<11>("v_mad_i32_i24", d, w); run<5>("v_mov_dpp wave_shr", d, w);
        run<6>("v_cndmask", d, w); run<7>("v_lshl_add_u64", d, w); run<14>("v_lshlrev_b64", d, w); run<8>("v_cvt_f32_i32", d, w); run<12>("v_cvt_u32_f32", d, w);
        run<10>("v_cmp_gt_f32", d, w); run<13>("v_readlane", d, w); run<15>("v_pk_mul_f32", d, w);
    }
    return 0;
}
