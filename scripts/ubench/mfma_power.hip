// What does the matrix pipe sustain under the chip's power limit, per instruction type and operand content?  Every wave (one workgroup of
// four waves per compute unit x 2) issues N matrix instructions on eight independent accumulator tiles, cycling through eight operand
// register sets filled from a host buffer (random values / zeros).  ~0.1 s per launch (long enough for the clock to settle); shader
// clocks from s_memtime against the 100 MHz s_memrealtime.  f16 and bf16: v_mfma_f32_32x32x16 (16 K per instruction), i8:
// v_mfma_i32_32x32x32 (32 K per instruction: twice the multiply-adds).
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_power.hip -o scripts/ubench/bin/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int KIND>      // 0 f16, 1 bf16, 2 i8
__global__ void __launch_bounds__(256) k(const i32x4* __restrict__ src, float* out, unsigned long long* clk, int iters)
{
    i32x4 a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = src[(j * 2 + 0) * 64 + (threadIdx.x & 63)]; b[j] = src[(j * 2 + 1) * 64 + (threadIdx.x & 63)]; }
    f32x16 accf[8];
    i32x16 acci[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) { accf[t][e] = 0.f; acci[t][e] = 0; }
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (KIND == 0) accf[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[t]), __builtin_bit_cast(f16x8, b[(t + 3) & 7]), accf[t], 0, 0, 0);
            else if (KIND == 1) accf[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[t]), __builtin_bit_cast(bf16x8, b[(t + 3) & 7]), accf[t], 0, 0, 0);
            else acci[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[t], b[(t + 3) & 7], acci[t], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += accf[t][e] + (float)acci[t][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int KIND> static void run(const char* name, const std::vector<unsigned>& host, int wgs, int iters)
{
    i32x4* src; float* out; unsigned long long* clk;
    hipMalloc(&src, host.size() * 4); hipMalloc(&out, (size_t)wgs * 256 * 4); hipMalloc(&clk, 16);
    hipMemcpy(src, host.data(), host.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t ea, eb; hipEventCreate(&ea); hipEventCreate(&eb);
    k<KIND><<<wgs, 256>>>(src, out, clk, iters / 8); hipDeviceSynchronize();
    hipEventRecord(ea); k<KIND><<<wgs, 256>>>(src, out, clk, iters); hipEventRecord(eb); hipEventSynchronize(eb);
    float ms; hipEventElapsedTime(&ms, ea, eb);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double insts_per_simd = (double)(wgs / 256) * iters * 8;                  // (wgs / 256 waves share a SIMD)
    const double macs = (double)wgs * 4 * iters * 8 * 32.0 * 32.0 * (KIND == 2 ? 32 : 16);
    printf("%-34s %7.1f ms   %6.3f G matrix instructions / s / SIMD   %7.1f T multiply-adds / s   shader clock %.2f GHz   %.1f clocks per instruction\n", name, ms,
           insts_per_simd / (ms * 1e-3) * 1e-9, macs / (ms * 1e-3) * 1e-12, (double)h[0] / ((double)h[1] / 100e6) * 1e-9,
           (double)h[0] / (double)h[1] * 100e6 * (ms * 1e-3) / insts_per_simd);
    hipFree(src); hipFree(out); hipFree(clk);
}

int main()
{
    const int n = 16 * 64 * 4;
    std::vector<unsigned> zeros(n, 0u), rf16(n), rbf16(n), ri8(n), hog16(n);
    srand(7);
    auto r16 = [](int ebits_lo, int ebits_hi, int mant_bits, int ebshift) {      // a random half-word: sign, exponent in a range, mantissa
        const unsigned e = ebits_lo + rand() % (ebits_hi - ebits_lo + 1), m = rand() & ((1u << mant_bits) - 1), s = rand() & 1;
        return (s << 15) | (e << ebshift) | m;
    };
    for (int i = 0; i < n; ++i) {
        rf16[i] = r16(9, 15, 10, 10) | (r16(9, 15, 10, 10) << 16);              // float16: |x| in 2^-6 .. 2, full mantissas
        rbf16[i] = r16(121, 127, 7, 7) | (r16(121, 127, 7, 7) << 16);          // bfloat16: the same range
        ri8[i] = (unsigned)rand() ^ ((unsigned)rand() << 16);                    // int8: all 256 values
        // non-negative float16 with most entries small (what the high pieces of HOG features x 2^12 look like: 0 .. 1 600)
        const unsigned h0 = (rand() % 3 == 0) ? 0u : ((unsigned)(15 + rand() % 11) << 10 | (rand() & 1023)), h1 = (rand() % 3 == 0) ? 0u : ((unsigned)(15 + rand() % 11) << 10 | (rand() & 1023));
        hog16[i] = h0 | (h1 << 16);
    }
    const int it = 1 << 19;
    run<0>("f16 32x32x16, zeros", zeros, 512, it);
    run<0>("f16 32x32x16, random", rf16, 512, it);
    run<0>("f16 32x32x16, feature-like", hog16, 512, it);
    run<1>("bf16 32x32x16, zeros", zeros, 512, it);
    run<1>("bf16 32x32x16, random", rbf16, 512, it);
    run<2>("i8 32x32x32, zeros", zeros, 512, it);
    run<2>("i8 32x32x32, random", ri8, 512, it);
    run<0>("f16 32x32x16, random (again)", rf16, 512, it);
    return 0;
}
