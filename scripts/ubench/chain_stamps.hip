// Where the microseconds of the factorisation's chain kernels go (round 5), and a stand-alone check of both generations of them:
// potrf_tile_kernel / trsm_tile_kernel (rounds 2-3; scripts/ubench/chain_gen1.inc) and potrf_tile2_kernel / trsm_tile2_kernel (round 5: the diagonal blocks' inverses as a
// by-product of the factor) of superviseddescent_amd/csrc/sdm_solve.hip, built with SDM_SOLVE_STAMPS -- thread 0 of workgroup 0 leaves
// the shader clock behind the kernels' barriers -- on one tile row of a random SPD system of T tiles (default 70: the RCR-22 shape).
// Checked against float64: the factor U_kk, the solved tile row U_kk^-T B and the transposed inverse the back substitution uses.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -o scripts/ubench/bin/chain_stamps scripts/ubench/chain_stamps.hip
#define SDM_SOLVE_STAMPS 1
#include "../../superviseddescent_amd/csrc/sdm_solve.hip"
#include "chain_gen1.inc"
#include <vector>
#include <cmath>
#include <cstdio>
#include <cstdlib>
// (the launcher of the translation unit refers to the float16 update kernels of sdm_gram_bf16.hip: not used here)
void sdm_launch_diag_absmax(const float*, long long, int, unsigned*, hipStream_t) {}
void sdm_launch_update_split_f16(const float*, long long, int, int, int, void*, unsigned*, int, int*, hipStream_t, bool) {}
void sdm_launch_update_f16(const void*, int, int, int, float*, long long, const unsigned*, int, int, int, int, int, hipStream_t, int) {}
static void stamps(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_solve_stamps), 64 * sizeof(unsigned long long)); }
int main(int argc, char** argv)
{
    const int T = argc > 1 ? atoi(argv[1]) : 70, n = T * TILE;
    std::vector<float> G((size_t)TILE * n);
    srand(1);
    // tile row 0 of an SPD matrix: diagonal tile = B^T B + 5 I with column scales over two decades, the rest random
    std::vector<float> A(TILE * TILE);
    for (int i = 0; i < TILE * TILE; ++i) A[i] = ((float)rand() / (float)RAND_MAX - 0.5f) * (0.1f + 3.0f * (float)((i % TILE) % 7) / 6.0f);
    for (int i = 0; i < TILE; ++i)
        for (int j = 0; j < n; ++j) {
            if (j < TILE) { double s = 0; for (int k = 0; k < TILE; ++k) s += (double)A[k * TILE + i] * A[k * TILE + j]; G[(size_t)i * n + j] = (float)s + (i == j ? 0.5f : 0.0f); }
            else G[(size_t)i * n + j] = (float)rand() / (float)RAND_MAX - 0.5f;
        }
    // float64 reference: U (upper Cholesky of the diagonal tile), Y = U^-T B for the first other tile, W = U^-1
    std::vector<double> U(TILE * TILE, 0.0), Y((size_t)TILE * TILE), Winv(TILE * TILE, 0.0);
    for (int i = 0; i < TILE; ++i)
        for (int j = i; j < TILE; ++j) {
            double s = G[(size_t)i * n + j];
            for (int k = 0; k < i; ++k) s -= U[k * TILE + i] * U[k * TILE + j];
            U[i * TILE + j] = (i == j) ? sqrt(s) : s / U[i * TILE + i];
        }
    for (int c = 0; c < TILE; ++c)          // forward substitution with L = U^T, column by column
        for (int i = 0; i < TILE; ++i) {
            double s = G[(size_t)i * n + TILE + c];
            for (int k = 0; k < i; ++k) s -= U[k * TILE + i] * Y[(size_t)k * TILE + c];
            Y[(size_t)i * TILE + c] = s / U[i * TILE + i];
        }
    for (int c = 0; c < TILE; ++c)          // U W = I
        for (int i = TILE - 1; i >= 0; --i) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int k = i + 1; k < TILE; ++k) s -= U[i * TILE + k] * Winv[k * TILE + c];
            Winv[i * TILE + c] = s / U[i * TILE + i];
        }
    float *d, *winv; int* st;
    (void)hipMalloc(&d, (size_t)TILE * n * 4); (void)hipMalloc(&winv, TILE * TILE * 4); (void)hipMalloc(&st, 4); (void)hipMemset(st, 0, 4);
    const size_t lds_potrf = ((size_t)IB * POTRF_PLD + IB * IB + 4 + 8 * IB * (IB + 1)) * sizeof(float);
    const size_t lds_potrf2 = (size_t)POTRF2_LDS_FLOATS * sizeof(float);
    const size_t lds_trsm = (size_t)TRSM_LDS_FLOATS * sizeof(float), lds_trsm1 = ((size_t)TRSM_LDS_FLOATS + 8 * IB * (IB + 1)) * sizeof(float);
    (void)hipFuncSetAttribute((const void*)potrf_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)trsm_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)potrf_tile2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)trsm_tile2_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)trsm_tile2_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1, e2; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreate(&e2);
    for (int ver = 1; ver <= 3; ++ver) {      // 1: rounds 2-3; 2: round 5, eight-wave panel solve; 3: round 5, four-wave panel solve (two workgroups per tile)
        float best_p = 1e9f, best_t = 1e9f;
        unsigned long long s[64];
        for (int rep = 0; rep < 6; ++rep) {
            (void)hipMemcpy(d, G.data(), (size_t)TILE * n * 4, hipMemcpyHostToDevice);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0, 0);
            if (ver == 1) hipLaunchKernelGGL(potrf_tile_kernel, dim3(1), dim3(512), lds_potrf, 0, d, (long long)n, 0, st);
            else hipLaunchKernelGGL(potrf_tile2_kernel, dim3(1), dim3(512), lds_potrf2, 0, d, (long long)n, 0, st);      // (ver 2 and 3)
            (void)hipEventRecord(e1, 0);
            if (ver == 1) hipLaunchKernelGGL(trsm_tile_kernel, dim3(T - 1 + 1), dim3(512), lds_trsm1, 0, d, (long long)n, 0, 1, T - 1, winv, 1, st);
            else if (ver == 2) hipLaunchKernelGGL(trsm_tile2_kernel<8>, dim3(T - 1 + 1), dim3(512), lds_trsm, 0, d, (long long)n, 0, 1, T - 1, winv, 1, st, (unsigned*)nullptr, 1 << 30);
            else hipLaunchKernelGGL(trsm_tile2_kernel<4>, dim3(2 * (T - 1 + 1)), dim3(256), lds_trsm, 0, d, (long long)n, 0, 1, T - 1, winv, 1, st, (unsigned*)nullptr, 1 << 30);
            (void)hipEventRecord(e2, 0);
            (void)hipDeviceSynchronize();
            float a, b; (void)hipEventElapsedTime(&a, e0, e1); (void)hipEventElapsedTime(&b, e1, e2);
            if (a < best_p) best_p = a;
            if (b < best_t) best_t = b;
        }
        stamps(s);
        int status = 0; (void)hipMemcpy(&status, st, 4, hipMemcpyDeviceToHost);
        std::vector<float> R((size_t)TILE * n), Wd(TILE * TILE);
        (void)hipMemcpy(R.data(), d, (size_t)TILE * n * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(Wd.data(), winv, TILE * TILE * 4, hipMemcpyDeviceToHost);
        double eu = 0, nu = 0, ey = 0, ny = 0, ew = 0, nw = 0;
        for (int i = 0; i < TILE; ++i)
            for (int j = 0; j < TILE; ++j) {
                if (j >= i) { const double e = R[(size_t)i * n + j] - U[i * TILE + j]; eu += e * e; nu += U[i * TILE + j] * U[i * TILE + j]; }
                const double e2 = R[(size_t)i * n + TILE + j] - Y[(size_t)i * TILE + j]; ey += e2 * e2; ny += Y[(size_t)i * TILE + j] * Y[(size_t)i * TILE + j];
                const double e3 = Wd[j * TILE + i] - Winv[i * TILE + j]; ew += e3 * e3; nw += Winv[i * TILE + j] * Winv[i * TILE + j];      // (stored transposed)
            }
        printf("\n== generation %d ==  T = %d tiles, status %d; HIP events: potrf %.1f us, panel solve (%d tiles + inverse) %.1f us\n", ver, T, status, best_p * 1e3f, T - 1, best_t * 1e3f);
        printf("against float64 (relative Frobenius): factor %.2e, solved tile %.2e, inverse %.2e\n", sqrt(eu / nu), sqrt(ey / ny), sqrt(ew / nw));
        const double total = (double)(s[20] - s[0]);
        printf("potrf, shader clocks (stamp 0 -> 20 = %.0f clocks):\n", total);
        printf("  load + first barrier        %6llu\n", s[1] - s[0]);
        for (int jb = 0; jb < 8; ++jb) {
            const unsigned long long a0 = jb == 0 ? s[1] : s[3 + 2 * (jb - 1)];
            printf("  step %d: (update of the previous step +) diagonal 16 x 16 factor %6llu", jb, s[2 + 2 * jb] - a0);
            if (jb < 7) printf("   panel solve %6llu", s[3 + 2 * jb] - s[2 + 2 * jb]);
            printf("\n");
        }
        printf("  store                       %6llu\n", s[20] - s[16]);
        printf("panel solve (workgroup 0), shader clocks: strip load issued + U_kk -> LDS + barrier %llu", s[34] - s[32]);
        if (ver == 1) printf(" (of which the diagonal-block inverses + their barrier %llu)", s[34] - s[33]);
        printf(", eight steps %llu, store %llu\n", s[35] - s[34], s[36] - s[35]);
    }
    return 0;
}
