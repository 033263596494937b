#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double swap16_add(double a, double b) {
    const v2u lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const v2u hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}
__device__ __forceinline__ double swap32_add(double a, double b) {
    const v2u lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const v2u hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}
__global__ void k(double *out) {
    const int l = threadIdx.x;
    const double a = 1000.0 + l, b = 2000.0 + l;   // a: "plane k" value of lane l, b: "plane k+4" value
    out[l] = swap16_add(a, b);
    out[64 + l] = swap32_add(a, b);
}
int main() {
    double *d, h[128];
    (void)hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int row = l >> 4;
        const double e16 = (row & 1) ? (2000.0 + l) + (2000.0 + l - 16) : (1000.0 + l) + (1000.0 + l + 16);
        const double e32 = (l >= 32) ? (2000.0 + l) + (2000.0 + l - 32) : (1000.0 + l) + (1000.0 + l + 32);
        if (h[l] != e16 || h[64 + l] != e32) { ++bad; if (bad < 6) printf("lane %d: swap16 %g (want %g) swap32 %g (want %g)\n", l, h[l], e16, h[64 + l], e32); }
    }
    printf(bad ? "MISMATCH %d\n" : "permlane swap semantics as expected\n", bad);
    return bad != 0;
}
