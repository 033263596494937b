// accuracy of the hardware estimates v_rcp_f64 / v_rsq_f64 and of one / two refinement steps
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
__global__ void k(double *out) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long s = 88172645463325252ull + tid * 2654435761ull;
    double e_rcp = 0, e_rsq = 0, e_rcp1 = 0, e_rcp_cubic = 0, e_rsq1 = 0, e_rcp2 = 0;
    for (int it = 0; it < 2000; ++it) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double x = ldexp(1.0 + (double)(s >> 11) * 0x1p-53, (int)(s & 63) - 32);
        const double r = __builtin_amdgcn_rcp(x);
        const double q = __builtin_amdgcn_rsq(x);
        e_rcp = fmax(e_rcp, fabs(fma(-x, r, 1.0)));
        e_rsq = fmax(e_rsq, fabs(fma(-x * q, q, 1.0)) * 0.5);
        const double r1 = fma(fma(-x, r, 1.0), r, r);
        e_rcp1 = fmax(e_rcp1, fabs(r1 * x - 1.0));
        const double r2 = fma(fma(-x, r1, 1.0), r1, r1);
        e_rcp2 = fmax(e_rcp2, fabs(fma(-x, r2, 1.0)));
        const double e = fma(-x, r, 1.0);
        const double rc = fma(r, fma(e, e, e), r);
        e_rcp_cubic = fmax(e_rcp_cubic, fabs(fma(-x, rc, 1.0)));
        const double q1 = fma(0.5 * q, fma(-(x * q), q, 1.0), q);
        e_rsq1 = fmax(e_rsq1, fabs(fma(-x * q1, q1, 1.0)) * 0.5);
    }
    double v[6] = {e_rcp, e_rsq, e_rcp1, e_rcp_cubic, e_rsq1, e_rcp2};
    for (int m = 0; m < 6; ++m) atomicMax((unsigned long long *)(out + m), (unsigned long long)__double_as_longlong(v[m]));
}
int main() {
    double *d, h[6];
    hipMalloc(&d, 48); hipMemset(d, 0, 48);
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d);
    hipMemcpy(h, d, 48, hipMemcpyDeviceToHost);
    printf("max relative error: rcp %.3e  rsq %.3e  rcp+1 Newton %.3e  rcp+cubic %.3e  rsq+1 Newton %.3e  rcp+2 Newton %.3e\n", h[0], h[1], h[2], h[3], h[4], h[5]);
    return 0;
}
