// Cheaper ways to a correctly rounded sqrt on the HOG gradient's inputs (g2 = gx^2 + gy^2, integers <= 130050)?  The shipping
// form is v_sqrt_f32 + a 4-instruction one-sided residual test (sqrt_int_up); v_sqrt_f32 alone is exact or one ulp low.
// Each candidate is compared with (float)sqrt((double)g2) over all 511^2 gradients: mismatches, below, above.
#include <hip/hip_runtime.h>
#include <cstdio>
#define NC 8
__device__ inline float cand(int c, float x)
{
    switch (c) {
        case 0: return __builtin_amdgcn_sqrtf(x);
        case 1: return __builtin_amdgcn_sqrtf(x * 1.00000011920928955078125f);            // x (1 + 2^-23)
        case 2: return __builtin_amdgcn_sqrtf(__builtin_fmaf(x, 5.9604644775390625e-08f, x));   // x + x 2^-24 (rounded)
        case 3: return x * __builtin_amdgcn_rsqf(x);
        case 4: { const float r = __builtin_amdgcn_sqrtf(x); return __builtin_fmaf(r, 5.9604644775390625e-08f, r); }      // r (1 + 2^-24)
        case 5: { const float r = __builtin_amdgcn_sqrtf(x); return __builtin_fmaf(r, 2.98023223876953125e-08f, r); }     // r (1 + 2^-25)
        case 6: return (float)__builtin_sqrt((double)x);                                     // f64 path (cvt, sqrt_f64 sequence, cvt)
        default: { const float r = __builtin_amdgcn_sqrtf(x); const float rp = __builtin_bit_cast(float, __builtin_bit_cast(int, r) + 1);
                   return __builtin_fmaf(-rp, r, x) > 0.0f ? rp : r; }                      // the shipping form
    }
}
__global__ void k(int* stat)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 511 * 511) return;
    const float gx = (float)(t % 511 - 255), gy = (float)(t / 511 - 255);
    const float g2 = gx * gx + gy * gy;
    const float ref = (float)sqrt((double)g2);
    for (int c = 0; c < NC; ++c) {
        const float v = g2 == 0.0f && c == 3 ? 0.0f : cand(c, g2);
        const int d = __float_as_int(v) - __float_as_int(ref);
        if (d != 0) atomicAdd(stat + 3 * c, 1);
        if (d < 0) atomicAdd(stat + 3 * c + 1, 1);
        if (d > 0) atomicAdd(stat + 3 * c + 2, 1);
    }
}
int main()
{
    int *d, h[3 * NC];
    hipMalloc(&d, sizeof(h)); hipMemset(d, 0, sizeof(h));
    k<<<(511 * 511 + 255) / 256, 256>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[NC] = {"v_sqrt_f32(x)", "v_sqrt_f32(x (1 + 2^-23))", "v_sqrt_f32(fma(x, 2^-24, x))", "x * v_rsq_f32(x)", "fma(r, 2^-24, r), r = v_sqrt_f32(x)",
                             "fma(r, 2^-25, r)", "f64 sqrt", "v_sqrt_f32 + one-sided residual test (shipping)"};
    for (int c = 0; c < NC; ++c) printf("%-52s mismatches %6d  below %6d  above %6d\n", names[c], h[3 * c], h[3 * c + 1], h[3 * c + 2]);
    return 0;
}
