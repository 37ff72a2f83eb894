// micro-benchmark 2: integer LDS atomics (u32/u64) vs f32, and dependence on the active-lane count.
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ void k(unsigned long long* out, long long* cyc, int iters, int active)
{
    __shared__ unsigned long long h64[2048];
    unsigned int* h32 = (unsigned int*)h64;
    float* hf = (float*)h64;
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) h64[i] = 0;
    __syncthreads();
    int addr = ((lane / 11) * 8 + (lane * 5) % 8) + (threadIdx.x >> 6) * 256;   // hog-like
    long long t0 = clock64();
    if (lane < active) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (MODE == 0) atomicAdd(&hf[addr + u * 64], 1.0f + lane);
                if (MODE == 1) atomicAdd(&h32[addr + u * 64], 3u + lane);
                if (MODE == 2) atomicAdd(&h64[addr + u * 64], 3ull + lane);
                if (MODE == 3) atomicMax(&h32[addr + u * 64], 3u + lane + i);
            }
        }
    }
    __syncthreads();
    long long t1 = clock64();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    out[threadIdx.x] = h64[threadIdx.x];
}

template <int MODE>
void run(const char* name, int waves, int active)
{
    unsigned long long* d; long long* c;
    (void)hipMalloc(&d, 4096 * 8); (void)hipMalloc(&c, 8);
    const int iters = 1000;
    for (int r = 0; r < 2; ++r) { k<MODE><<<1, 64 * waves>>>(d, c, iters, active); (void)hipDeviceSynchronize(); }
    long long h;
    (void)hipMemcpy(&h, c, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-14s waves=%d active=%2d : %.1f cycles per wave-instruction (per CU-instr %.1f)\n", name, waves, active,
           (double)h / (iters * 8), (double)h / (iters * 8) / waves);
    (void)hipFree(d); (void)hipFree(c);
}

int main()
{
    for (int a : {64, 32, 16, 8}) run<0>("ds_add_f32", 1, a);
    for (int a : {64, 32, 16, 8}) run<1>("ds_add_u32", 1, a);
    for (int a : {64, 32, 16, 8}) run<2>("ds_add_u64", 1, a);
    run<3>("ds_max_u32", 1, 64);
    run<0>("ds_add_f32", 4, 64); run<1>("ds_add_u32", 4, 64); run<2>("ds_add_u64", 4, 64);
    return 0;
}
