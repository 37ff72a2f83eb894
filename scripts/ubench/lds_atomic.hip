// micro-benchmark: cost of ds_add_f32 (LDS float atomic add) per wave-instruction under different
// address patterns, vs plain ds_write/ds_read.  One wave per block, 1 block; cycles via s_memtime.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int MODE>
__global__ void k(float* out, long long* cyc, int iters)
{
    __shared__ float h[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) h[i] = 0.f;
    __syncthreads();
    int addr;
    switch (MODE) {
        case 0: addr = lane; break;                 // 64 distinct consecutive
        case 1: addr = lane >> 3; break;            // 8 addresses x 8 lanes, runs of equal neighbours
        case 2: addr = 0; break;                    // all the same
        case 3: addr = (lane * 37) & 255; break;    // scattered distinct
        case 4: addr = ((lane / 11) * 8 + (lane * 5) % 8) ; break; // hog-like: cell x bin
        default: addr = lane;
    }
    addr += (threadIdx.x >> 6) * 512;
    float v = 1.0f + lane;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE < 5) atomicAdd(&h[addr + u * 64 % 256], v);
            else if (MODE == 5) h[addr + (u * 64) % 256] = v;                     // plain store
            else { v += h[(addr + u * 64) % 256]; }                                // plain load
        }
    }
    __syncthreads();
    long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = h[threadIdx.x] + v;
}

template <int MODE>
void run(const char* name, int waves)
{
    float* d; long long* c;
    hipMalloc(&d, 1024 * 256 * sizeof(float)); hipMalloc(&c, 1024 * sizeof(long long));
    const int iters = 1000;
    k<MODE><<<1, 64 * waves>>>(d, c, iters);
    hipDeviceSynchronize();
    k<MODE><<<1, 64 * waves>>>(d, c, iters);
    hipDeviceSynchronize();
    long long h;
    hipMemcpy(&h, c, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-28s waves=%d  %.1f cycles per wave-instruction (x%d waves => %.1f per CU-instr)\n", name, waves,
           (double)h / (iters * 8), waves, (double)h / (iters * 8) / waves);
    hipFree(d); hipFree(c);
}

int main()
{
    for (int w : {1, 4}) {
        if (w == 1) {
            run<0>("atomic distinct consecutive", 1); run<1>("atomic 8 addr x 8 lanes", 1); run<2>("atomic all same", 1);
            run<3>("atomic scattered", 1); run<4>("atomic hog-like", 1); run<5>("plain ds_write_b32", 1); run<6>("plain ds_read_b32", 1);
        } else {
            run<0>("atomic distinct consecutive", 4); run<1>("atomic 8 addr x 8 lanes", 4); run<2>("atomic all same", 4);
            run<3>("atomic scattered", 4); run<4>("atomic hog-like", 4); run<5>("plain ds_write_b32", 4); run<6>("plain ds_read_b32", 4);
        }
    }
    return 0;
}
