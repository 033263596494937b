// fp64 matrix-core peak: every wave issues independent MFMAs back to back.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4f64 __attribute__((ext_vector_type(4)));
template <int NACC, int KIND>
__global__ void __launch_bounds__(256) k(double *out, int iters) {
    v4f64 acc[NACC];
    double acc1[NACC];
    for (int i = 0; i < NACC; ++i) { acc[i] = (v4f64){0, 0, 0, 0}; acc1[i] = 0; }
    double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
            else acc1[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc1[i], 0, 0, 0);
        }
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + acc1[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, int KIND>
void run(int wgs_per_cu) {
    int cus = 256, iters = 2000;
    double *out;
    (void)hipMalloc(&out, sizeof(double) * cus * wgs_per_cu * 256);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NACC, KIND><<<cus * wgs_per_cu, 256>>>(out, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<NACC, KIND><<<cus * wgs_per_cu, 256>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double per = KIND == 0 ? 2.0 * 16 * 16 * 4 : 2.0 * 4 * 4 * 4 * 4;
    double flops = per * (double)NACC * iters * 4 * cus * wgs_per_cu;
    printf("%s NACC=%d wgs/cu=%d: %.3f ms, %.1f TFLOP/s, %.1f cycles/MFMA/SIMD at 2.4 GHz\n", KIND ? "4x4x4  " : "16x16x4", NACC,
           wgs_per_cu, ms, flops / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)NACC * iters * wgs_per_cu));
    (void)hipFree(out);
}
int main() {
    run<16, 0>(1);
    run<16, 0>(2);
    run<8, 0>(4);
    run<16, 1>(1);
    run<16, 1>(2);
    run<16, 1>(4);
    return 0;
}
