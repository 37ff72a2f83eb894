// What does v_mfma_f32_16x16x4_f32 sustain on gfx950, alone and with the LDS fragment reads of the regressor-apply GEMM beside
// it?  Every wave runs N MFMAs on NACC independent accumulators; variants add 4 ds_read_b128 per 12 MFMAs (the apply's k-group)
// and an s_barrier per 48 MFMAs (its slab).  Core clock from s_memtime (shader clock) against s_memrealtime (100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC, int MODE>      // MODE 0: MFMA only, 1: + LDS reads, 2: + LDS reads + barrier per slab
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* clk, int iters)
{
    __shared__ __attribute__((aligned(16))) float lds[112 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
    for (int i = threadIdx.x; i < 112 * 64; i += 256) lds[i] = 1e-3f * (float)(i & 7);
    __syncthreads();
    f32x4 acc[NACC];
#pragma unroll
    for (int c = 0; c < NACC; ++c) acc[c] = (f32x4){0, 0, 0, 0};
    f32x4 av = {1.f, 2.f, 3.f, 4.f}, bv[3] = {{1, 1, 1, 1}, {2, 2, 2, 2}, {3, 3, 3, 3}};
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {          // one "slab": 4 k-groups x 12 MFMAs
        if (MODE == 2) __syncthreads();
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            if (MODE >= 1) {
                const int pos = 4 * ((lq + 4 * kg) ^ li);
                av = *(const f32x4*)(lds + (16 * wave + li) * 64 + pos);
#pragma unroll
                for (int c = 0; c < 3; ++c) bv[c] = *(const f32x4*)(lds + (64 + 16 * c + li) * 64 + pos);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    acc[(e * 3 + c) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[c][e], acc[(e * 3 + c) % NACC], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
#pragma unroll
    for (int c = 0; c < NACC; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
template <int NACC, int MODE> static void run(const char* name, int wgs, int iters = 2000)
{
    float* out; unsigned long long* clk; hipMalloc(&out, (size_t)wgs * 256 * 4); hipMalloc(&clk, 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<NACC, MODE><<<wgs, 256>>>(out, clk, iters); hipDeviceSynchronize();
    hipEventRecord(a); k<NACC, MODE><<<wgs, 256>>>(out, clk, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double mf = (double)wgs * 4 * iters * 48, fl = mf * 2 * 16 * 16 * 4;
    printf("%-46s %4d wgs: %7.1f us  %6.1f TF  %5.1f cycles/MFMA/SIMD  shader clock %.2f GHz\n", name, wgs, ms * 1e3, fl / (ms * 1e-3) * 1e-12,
           (double)h[0] / (iters * 48.0) / ((wgs + 255) / 256 > 1 ? (wgs / 256) : 1), (double)h[0] / ((double)h[1] / 100e6) * 1e-9);
    hipFree(out); hipFree(clk);
}
int main()
{
    run<3, 0>("MFMA only, 3 accumulators", 256);
    run<6, 0>("MFMA only, 6 accumulators", 256);
    run<12, 0>("MFMA only, 12 accumulators", 256);
    run<3, 0>("MFMA only, 3 accumulators", 512);
    run<3, 1>("+ 4 ds_read_b128 per 12 MFMAs", 256);
    run<3, 1>("+ 4 ds_read_b128 per 12 MFMAs", 512);
    run<6, 1>("+ reads, 6 accumulators", 512);
    run<3, 2>("+ reads + barrier per 48 MFMAs", 512);
    // sustained: ~70 ms launches, the length of the Gram launch of training (does the clock hold?)
    run<3, 0>("MFMA only, 3 accumulators, 55x longer", 512, 110000);
    run<3, 2>("+ reads + barrier, 55x longer", 512, 110000);
    return 0;
}
