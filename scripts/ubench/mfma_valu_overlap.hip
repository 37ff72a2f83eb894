// Does a SIMD's vector ALU keep its issue rate while ANOTHER wave of the same SIMD runs matrix instructions?  Workgroups of 8 waves
// (two per SIMD): waves 0-3 run 8 independent v_fma_f32 chains, waves 4-7 run nothing / v_mfma_f32_16x16x4_f32 (f32 in) /
// v_mfma_f32_16x16x32_f16.  Cycles per instruction of either role from s_memtime.  Question behind it (round 4): the HOG pixel
// kernel is 68 % vector-busy + 24 % f32-matrix-busy = 92 % -- do the band folds' f32 matrix instructions run on the vector
// multiply-adders (then float16 folds would give the time back), or beside them?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int MODE>      // 0: partner idle, 1: partner f32 MFMA, 2: partner f16 MFMA, 3: VALU role idle + f32 MFMA, 4: VALU role idle + f16 MFMA
__global__ void __launch_bounds__(512) k(float* out, unsigned long long* clk, int iters)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float r = 0.0f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (wave < 4) {
        if (MODE < 3) {
            float a[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = (float)(lane + i);
            const float c1 = 1.0001f, c2 = 0.5f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], c1, c2);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) r += a[i];
        }
    } else {
        if (MODE == 1 || MODE == 3) {
            f32x4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
            float av = (float)lane, bv = 1.0f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[1], 0, 0, 0);
                }
            }
            r = acc[0][0] + acc[1][1];
        } else if (MODE == 2 || MODE == 4) {
            f32x4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
            f16x8 av, bv;
#pragma unroll
            for (int i = 0; i < 8; ++i) { av[i] = (_Float16)(lane + i); bv[i] = (_Float16)1.0f; }
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc[1], 0, 0, 0);
                }
            }
            r = acc[0][0] + acc[1][1];
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4)) clk[wave >> 2] = t1 - t0;
}
template <int MODE> static void run(const char* name, int iters = 20000)
{
    float* out; unsigned long long* clk; hipMalloc(&out, (size_t)256 * 512 * 4); hipMalloc(&clk, 16); hipMemset(clk, 0, 16);
    k<MODE><<<256, 512>>>(out, clk, iters); hipDeviceSynchronize();
    k<MODE><<<256, 512>>>(out, clk, iters); hipDeviceSynchronize();
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("%-52s vector role: %6.2f cycles per v_fma_f32   matrix role: %6.2f cycles per MFMA\n", name,
           MODE < 3 ? (double)h[0] / (iters * 64.0) : 0.0, MODE ? (double)h[1] / (iters * 8.0) : 0.0);
    hipFree(out); hipFree(clk);
}
int main()
{
    run<0>("v_fma_f32 alone (partner wave idle)");
    run<1>("v_fma_f32 beside v_mfma_f32_16x16x4_f32");
    run<2>("v_fma_f32 beside v_mfma_f32_16x16x32_f16");
    run<3>("v_mfma_f32_16x16x4_f32 alone");
    run<4>("v_mfma_f32_16x16x32_f16 alone");
    return 0;
}
