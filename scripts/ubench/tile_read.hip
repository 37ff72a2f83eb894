// Does the feature matrix's ACCESS PATTERN cap the regressor-apply GEMM?  Streams a [4096][9216] f32 matrix (151 MB) the way
// apply_tiled_kernel does -- a workgroup owns 64 rows x one K-split and walks it in slabs of 64 floats: every wave instruction
// fetches 256 contiguous bytes of four rows -- and, for comparison, in slabs of 256 floats (1 KB contiguous per row and wave
// instruction) and as one linear stream.  No arithmetic beyond a dummy sum.  A second buffer is written between passes so that
// the matrix does not simply sit in the 256 MB memory-side cache (the HOG kernel writes it right before the apply reads it: the
// "warm" rows skip that flush).
#include <hip/hip_runtime.h>
#include <cstdio>
#define ROWS 4096
#define LD 9216
template <int BK, int UNROLL>
__global__ void __launch_bounds__(256) tile_rd(const float* __restrict__ f, int splits, float* out)
{
    constexpr int CH = BK / 4, RPP = 256 / CH, NP = 64 / RPP;      // chunks per row, rows per pass of the workgroup, passes
    const int t = threadIdx.x, r = t / CH, c = t % CH;
    const int row0 = blockIdx.x * 64, split = blockIdx.y;
    const int kslabs = LD / BK, s0 = kslabs * split / splits, s1 = kslabs * (split + 1) / splits;
    float4 acc = {0, 0, 0, 0};
    for (int s = s0; s < s1; s += UNROLL) {
        float4 v[UNROLL][NP];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int ss = s + u < s1 ? s + u : s1 - 1;
                v[u][p] = *(const float4*)(f + (size_t)(row0 + r + RPP * p) * LD + (size_t)ss * BK + 4 * c);
            }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int p = 0; p < NP; ++p) { acc.x += v[u][p].x; acc.y += v[u][p].y; acc.z += v[u][p].z; acc.w += v[u][p].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}
__global__ void linear_rd(const float4* __restrict__ p, size_t n, float* out)
{
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = p[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 123.456f) out[0] = s;
}
__global__ void fill(float4* p, size_t n, float v)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = {v, v, v, v};
}
template <class F> static void timeit(const char* name, F launch, float4* flush, size_t flush_n, bool cold)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float total = 0; const int reps = 20;
    for (int r = -2; r < reps; ++r) {
        if (cold) fill<<<2048, 256>>>(flush, flush_n, (float)r);
        hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (r >= 0) total += ms;
    }
    const double bytes = (double)ROWS * LD * 4;
    printf("%-44s %s: %7.2f us  %.2f TB/s\n", name, cold ? "cold" : "warm", total / reps * 1e3, bytes / (total / reps * 1e-3) / 1e12);
}
int main()
{
    float *f, *o; float4* flush; const size_t fb = (size_t)ROWS * LD * 4, flush_b = (size_t)1 << 30;
    hipMalloc(&f, fb); hipMalloc(&o, 4); hipMalloc(&flush, flush_b); hipMemset(f, 1, fb);
    for (int cold = 0; cold < 2; ++cold) {
        timeit("linear float4 stream, 4096 wgs", [&] { linear_rd<<<4096, 256>>>((const float4*)f, fb / 16, o); }, flush, flush_b / 16, cold);
        timeit("64 x 64 slabs, 8 splits (512 wgs), unroll 2", [&] { tile_rd<64, 2><<<dim3(64, 8), 256>>>(f, 8, o); }, flush, flush_b / 16, cold);
        timeit("64 x 64 slabs, 8 splits (512 wgs), unroll 4", [&] { tile_rd<64, 4><<<dim3(64, 8), 256>>>(f, 8, o); }, flush, flush_b / 16, cold);
        timeit("64 x 64 slabs, 16 splits (1024 wgs), unroll 4", [&] { tile_rd<64, 4><<<dim3(64, 16), 256>>>(f, 16, o); }, flush, flush_b / 16, cold);
        timeit("64 x 64 slabs, 32 splits (2048 wgs), unroll 2", [&] { tile_rd<64, 2><<<dim3(64, 32), 256>>>(f, 32, o); }, flush, flush_b / 16, cold);
        timeit("64 x 256 slabs, 8 splits (512 wgs), unroll 1", [&] { tile_rd<256, 1><<<dim3(64, 8), 256>>>(f, 8, o); }, flush, flush_b / 16, cold);
        timeit("64 x 256 slabs, 16 splits (1024 wgs), unroll 1", [&] { tile_rd<256, 1><<<dim3(64, 16), 256>>>(f, 16, o); }, flush, flush_b / 16, cold);
        timeit("64 x 128 slabs, 8 splits (512 wgs), unroll 2", [&] { tile_rd<128, 2><<<dim3(64, 8), 256>>>(f, 8, o); }, flush, flush_b / 16, cold);
    }
    return 0;
}
