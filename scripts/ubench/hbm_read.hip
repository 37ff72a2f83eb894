// Practical HBM read bandwidth: every thread streams float4 loads over a large buffer (grid-stride), one dummy store.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void rd(const float4* __restrict__ p, size_t n, float* out)
{
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = p[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 123.456f) out[0] = s;
}
int main()
{
    for (size_t mb : {144, 1024, 4096}) {
        const size_t bytes = mb << 20;
        float4* d; float* o;
        hipMalloc(&d, bytes); hipMalloc(&o, 4); hipMemset(d, 1, bytes);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int blocks : {1024, 4096, 16384}) {
            rd<<<blocks, 256>>>(d, bytes / 16, o); hipDeviceSynchronize();
            hipEventRecord(a);
            const int reps = 10;
            for (int r = 0; r < reps; ++r) rd<<<blocks, 256>>>(d, bytes / 16, o);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("%5zu MB, %5d blocks: %.3f ms/pass -> %.2f TB/s\n", mb, blocks, ms / reps, bytes / (ms / reps * 1e-3) / 1e12);
        }
        hipFree(d); hipFree(o);
    }
    return 0;
}
