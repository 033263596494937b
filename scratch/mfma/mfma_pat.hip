// v_mfma_f64_4x4x4 in the register pattern of the Cholesky trailing update: acc[16][4] += af[I] x bf[c]
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int WPE>
__global__ void __launch_bounds__(256, WPE) k(double *out, const double *in, int iters) {
    double acc[16][4], af[16], bf[4];
#pragma unroll
    for (int I = 0; I < 16; ++I) { af[I] = in[threadIdx.x + I]; for (int c = 0; c < 4; ++c) acc[I][c] = 0.0; }
#pragma unroll
    for (int c = 0; c < 4; ++c) bf[c] = in[threadIdx.x + 32 + c];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int I = 0; I < 16; ++I)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[I][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[I], bf[c], acc[I][c], 0, 0, 0);
    }
    double s = 0;
    for (int I = 0; I < 16; ++I) for (int c = 0; c < 4; ++c) s += acc[I][c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int WPE>
void run(int wgs_per_cu) {
    int cus = 256, iters = 500;
    double *out, *in;
    (void)hipMalloc(&out, sizeof(double) * cus * wgs_per_cu * 256);
    (void)hipMalloc(&in, sizeof(double) * 1024);
    (void)hipMemset(in, 0, sizeof(double) * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<WPE><<<cus * wgs_per_cu, 256>>>(out, in, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<WPE><<<cus * wgs_per_cu, 256>>>(out, in, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double flops = 512.0 * 64 * iters * 4 * cus * wgs_per_cu;
    printf("trailing-update pattern, %d workgroups of 4 waves per CU: %.3f ms, %.1f TFLOP/s, %.1f cycles/MFMA/SIMD at 2.4 GHz\n", wgs_per_cu, ms,
           flops / ms / 1e9, ms * 1e-3 * 2.4e9 / (64.0 * iters * wgs_per_cu));
}
int main() {
    run<1>(1);
    run<2>(2);
    return 0;
}
