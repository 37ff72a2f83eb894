// Where the microseconds of the factorisation's chain kernels go (round 5): potrf_tile_kernel and trsm_tile_kernel of
// superviseddescent_amd/csrc/sdm_solve.hip built with SDM_SOLVE_STAMPS -- thread 0 of workgroup 0 leaves the shader clock behind the
// kernels' barriers -- on a random SPD system of T tiles (default 70: the RCR-22 shape), plus HIP-event times of the launches.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o scripts/ubench/bin/chain_stamps scripts/ubench/chain_stamps.hip
#define SDM_SOLVE_STAMPS 1
#include "../../superviseddescent_amd/csrc/sdm_solve.hip"
#include <vector>
#include <cmath>
#include <cstdio>
#include <cstdlib>
// (the launcher of the translation unit refers to the float16 update kernels of sdm_gram_bf16.hip: not used here)
void sdm_launch_diag_absmax(const float*, long long, int, unsigned*, hipStream_t) {}
void sdm_launch_update_split_f16(const float*, long long, int, int, int, void*, unsigned*, int, int*, hipStream_t) {}
void sdm_launch_update_f16(const void*, int, int, int, float*, long long, const unsigned*, int, int, int, int, int, hipStream_t) {}
static void stamps(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_solve_stamps), 64 * sizeof(unsigned long long)); }
int main(int argc, char** argv)
{
    const int T = argc > 1 ? atoi(argv[1]) : 70, n = T * TILE;
    std::vector<float> G((size_t)TILE * n);
    srand(1);
    // tile row 0 of an SPD matrix: diagonal tile = B^T B + 5 I, the rest small random
    std::vector<float> A(TILE * TILE);
    for (auto& v : A) v = (float)rand() / RAND_MAX - 0.5f;
    for (int i = 0; i < TILE; ++i)
        for (int j = 0; j < n; ++j) {
            if (j < TILE) { double s = 0; for (int k = 0; k < TILE; ++k) s += (double)A[k * TILE + i] * A[k * TILE + j]; G[(size_t)i * n + j] = (float)s + (i == j ? 5.0f : 0.0f); }
            else G[(size_t)i * n + j] = (float)rand() / RAND_MAX - 0.5f;
        }
    float *d, *winv; int* st;
    (void)hipMalloc(&d, (size_t)TILE * n * 4); (void)hipMalloc(&winv, TILE * TILE * 4); (void)hipMalloc(&st, 4); (void)hipMemset(st, 0, 4);
    const size_t lds_potrf = ((size_t)IB * POTRF_PLD + IB * IB + 4 + 8 * IB * (IB + 1)) * sizeof(float);
    const size_t lds_trsm = (size_t)TRSM_LDS_FLOATS * sizeof(float);
    (void)hipFuncSetAttribute((const void*)potrf_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)trsm_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1, e2; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreate(&e2);
    float best_p = 1e9f, best_t = 1e9f;
    unsigned long long s[64];
    for (int rep = 0; rep < 6; ++rep) {
        (void)hipMemcpy(d, G.data(), (size_t)TILE * n * 4, hipMemcpyHostToDevice);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(potrf_tile_kernel, dim3(1), dim3(512), lds_potrf, 0, d, (long long)n, 0, st);
        (void)hipEventRecord(e1, 0);
        hipLaunchKernelGGL(trsm_tile_kernel, dim3(T - 1 + 1), dim3(512), lds_trsm, 0, d, (long long)n, 0, 1, T - 1, winv, 1, st);
        (void)hipEventRecord(e2, 0);
        (void)hipDeviceSynchronize();
        float a, b; (void)hipEventElapsedTime(&a, e0, e1); (void)hipEventElapsedTime(&b, e1, e2);
        if (a < best_p) best_p = a;
        if (b < best_t) best_t = b;
    }
    stamps(s);
    int status = 0; (void)hipMemcpy(&status, st, 4, hipMemcpyDeviceToHost);
    printf("T = %d tiles, status %d; HIP events: potrf %.1f us, trsm (%d tiles + inverse) %.1f us\n", T, status, best_p * 1e3f, T - 1, best_t * 1e3f);
    const double total = (double)(s[20] - s[0]);
    printf("potrf_tile_kernel, shader clocks (share of stamp 0 -> 20 = %.0f clocks):\n", total);
    printf("  load + first barrier        %6llu\n", s[1] - s[0]);
    for (int jb = 0; jb < 8; ++jb) {
        const unsigned long long a0 = jb == 0 ? s[1] : s[3 + 2 * (jb - 1)];
        printf("  step %d: (c of the previous step +) diagonal 16 x 16 factor %6llu", jb, s[2 + 2 * jb] - a0);
        if (jb < 7) printf("   panel solve %6llu", s[3 + 2 * jb] - s[2 + 2 * jb]);
        printf("\n");
    }
    printf("  store                       %6llu\n", s[20] - s[16]);
    printf("trsm_tile_kernel (workgroup 0), shader clocks:\n  load of the strip issued + U_kk -> LDS + barrier %6llu\n  diagonal-block inverses + barrier %6llu\n  eight steps %6llu\n  store %6llu\n",
           s[33] - s[32], s[34] - s[33], s[35] - s[34], s[36] - s[35]);
    return 0;
}
