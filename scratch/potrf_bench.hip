// Micro-benchmark of the 64 x 64 diagonal-block kernel of the Cholesky panel (factor + inverse):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I3d_reconstruction_amd/csrc -Iinclude -o scratch/potrf_bench scratch/potrf_bench.hip
//   scratch/potrf_bench [variant]        (variant 0 = the library's kernel with phase stamps)
// Runs the kernel on 256 independent blocks one launch after the other, checks L L^T = A and Linv L = I on all of
// them, prints the time per launch (events around the whole sequence) and the phase stamps of the last launch.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "ba_device.h"

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

namespace ba {
constexpr int NB = 64;

#include "potrf_variants.inc"

}  // namespace ba

int main(int argc, char **argv) {
    const int variant = argc > 1 ? atoi(argv[1]) : 0;
    const int M = 256;
    std::vector<double> hA((size_t)M * 4096), hL((size_t)M * 4096), hX((size_t)M * 4096);
    srand(1);
    for (int b = 0; b < M; ++b) {
        // SPD: G G^T + 64 I with entries of G in [-1, 1]; the upper triangle is filled with NaN (never read)
        std::vector<double> G(4096);
        for (auto &g : G) g = 2.0 * rand() / RAND_MAX - 1.0;
        for (int r = 0; r < 64; ++r)
            for (int c = 0; c < 64; ++c) {
                double s = r == c ? 8.0 : 0.0;
                for (int m = 0; m < 64; ++m) s += G[r * 64 + m] * G[c * 64 + m];
                hA[(size_t)b * 4096 + r * 64 + c] = c <= r ? s : NAN;
            }
    }
    double *dA, *dX;
    int *dinfo;
    long long *dst;
    CK(hipMalloc(&dA, hA.size() * 8));
    CK(hipMalloc(&dX, hX.size() * 8));
    CK(hipMalloc(&dinfo, 4));
    CK(hipMalloc(&dst, 64 * 8));
    CK(hipMemset(dinfo, 0, 4));
    CK(hipMemset(dst, 0, 64 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemcpy(dA, hA.data(), hA.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemset(dX, 0xff, hX.size() * 8));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int b = 0; b < M; ++b) ba::launch_variant(variant, dA + (size_t)b * 4096, dX + (size_t)b * 4096, dinfo, dst);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        CK(hipGetLastError());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = fminf(best, ms);
    }
    CK(hipMemcpy(hL.data(), dA, hL.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hX.data(), dX, hX.size() * 8, hipMemcpyDeviceToHost));
    int info;
    long long st[64];
    CK(hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(st, dst, sizeof(st), hipMemcpyDeviceToHost));
    double worst_f = 0, worst_i = 0;
    for (int b = 0; b < M; ++b) {
        const double *A = &hA[(size_t)b * 4096], *L = &hL[(size_t)b * 4096], *X = &hX[(size_t)b * 4096];
        for (int r = 0; r < 64; ++r)
            for (int c = 0; c <= r; ++c) {
                double s = 0, t = 0;
                for (int m = 0; m <= c; ++m) s += L[r * 64 + m] * L[c * 64 + m];
                for (int m = c; m <= r; ++m) t += X[r * 64 + m] * L[m * 64 + c];
                worst_f = fmax(worst_f, fabs(s - A[r * 64 + c]) / A[r * 64 + r]);
                worst_i = fmax(worst_i, fabs(t - (r == c ? 1.0 : 0.0)));
                if (!(s == s) || !(t == t)) worst_f = worst_i = INFINITY;
            }
        for (int r = 0; r < 64; ++r)
            for (int c = r + 1; c < 64; ++c)
                if (X[r * 64 + c] != 0.0) worst_i = INFINITY;      // the inverse is written as a full lower-triangular block
    }
    printf("variant %d: %.2f us per launch (sequence of %d), info %d, |L L^T - A| / diag %.2e, |Linv L - I| %.2e\n", variant,
           best * 1e3 / M, M, info, worst_f, worst_i);
    printf("stamps of the last launch (10 ns ticks from the first):");
    for (int k = 1; k < 12 && st[k]; ++k) printf(" %lld", st[k] - st[0]);
    printf("\n");
    if (st[20]) printf("shader clock: %lld cycles over %lld ticks = %.0f MHz\n", st[20], st[6] - st[0], st[20] / ((st[6] - st[0]) * 1e-2));
    return 0;
}
