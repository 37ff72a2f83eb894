// Is the hardware v_sqrt_f32 already correctly rounded on the only inputs the HOG gradient can produce
// (g2 = gx^2 + gy^2, gx, gy integers in [-255, 255])?  Counts mismatches against (float)sqrt((double)g2).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* mism, int* first)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 511 * 511) return;
    const float gx = (float)(t % 511 - 255), gy = (float)(t / 511 - 255);
    const float g2 = gx * gx + gy * gy;
    const float raw = __builtin_amdgcn_sqrtf(g2);
    const float ref = (float)sqrt((double)g2);
    if (__float_as_int(raw) != __float_as_int(ref)) { if (atomicAdd(mism, 1) == 0) *first = t; }
    if (__float_as_int(raw) < __float_as_int(ref)) atomicAdd(mism + 2, 1);
    if (__float_as_int(raw) > __float_as_int(ref)) atomicAdd(mism + 3, 1);
    if (abs(__float_as_int(raw) - __float_as_int(ref)) > 1) atomicAdd(mism + 4, 1);
}
int main()
{
    int *d, h[5] = {0, 0, 0, 0, 0};
    hipMalloc(&d, 20); hipMemset(d, 0, 20);
    k<<<(511 * 511 + 255) / 256, 256>>>(d, d + 1);
    hipMemcpy(h, d, 20, hipMemcpyDeviceToHost);
    printf("raw v_sqrt_f32 mismatches over 511^2 gradients: %d (first at %d); below %d above %d, more than 1 ulp %d\n", h[0], h[1], h[2], h[3], h[4]);
    return 0;
}
