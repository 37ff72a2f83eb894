// Stand-alone check of potrf_tile_kernel (the kernels live in an anonymous namespace: the translation unit is included):
// factors a random SPD 128 x 128 tile, compares with an f64 Cholesky, prints the first differing elements and the rows that are
// wrong.  It located the missing wavefront-scope fences of round 2.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17
// -ffp-contract=off -I include -o scripts/ubench/bin/potrf_tile_test scripts/ubench/potrf_tile_test.hip
#include "../../superviseddescent_amd/csrc/sdm_solve.hip"
#include "chain_gen1.inc"      // (generation 1, the kernel this check was written for; scripts/ubench/chain_stamps.hip checks both generations)
#include <vector>
#include <cmath>
#include <cstdio>
int main()
{
    const int n = 128;
    std::vector<float> A(n * n), G(n * n);
    srand(1);
    for (auto& v : A) v = (float)rand() / RAND_MAX - 0.5f;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += (double)A[k * n + i] * A[k * n + j]; G[i * n + j] = (float)s + (i == j ? 5.0f : 0.0f); }
    // CPU upper Cholesky
    std::vector<double> U(n * n, 0.0);
    for (int i = 0; i < n; ++i) for (int j = i; j < n; ++j) { double s = G[i * n + j]; for (int k = 0; k < i; ++k) s -= U[k * n + i] * U[k * n + j]; U[i * n + j] = (i == j) ? sqrt(s) : s / U[i * n + i]; }
    float* d; int* st; hipMalloc(&d, n * n * 4); hipMalloc(&st, 4); hipMemset(st, 0, 4);
    hipMemcpy(d, G.data(), n * n * 4, hipMemcpyHostToDevice);
    const size_t lds = ((size_t)IB * POTRF_PLD + IB * IB + 4 + 8 * IB * (IB + 1)) * sizeof(float);
    hipLaunchKernelGGL(potrf_tile_kernel, dim3(1), dim3(512), lds, 0, d, (long long)n, 0, st);
    hipDeviceSynchronize();
    std::vector<float> R(n * n); int status = 0;
    hipMemcpy(R.data(), d, n * n * 4, hipMemcpyDeviceToHost); hipMemcpy(&status, st, 4, hipMemcpyDeviceToHost);
    printf("status %d err %s\n", status, hipGetErrorString(hipGetLastError()));
    int shown = 0; double maxe = 0;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
        const double want = j >= i ? U[i * n + j] : 0.0, e = fabs(R[i * n + j] - want);
        if (e > maxe) maxe = e;
        if (e > 1e-3 && shown < 6) { printf("(%d,%d) got %g want %g raw %g\n", i, j, R[i * n + j], want, G[i * n + j]); ++shown; }
    }
    printf("max abs err %g\n", maxe);
    for (int i = 0; i < 24; ++i) { int bad = 0; for (int j = 0; j < n; ++j) { const double want = j >= i ? U[i * n + j] : 0.0; if (fabs(R[i * n + j] - want) > 1e-3) ++bad; } printf("row %d: %d wrong\n", i, bad); }
    return 0;
}
