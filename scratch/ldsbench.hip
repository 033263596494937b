// LDS instruction throughput probe: cycles per wave-instruction per CU for fp64 atomics, stores and loads
// at several lane -> address patterns.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef double v2d __attribute__((ext_vector_type(2)));
#define LDSP(T) volatile __attribute__((address_space(3))) T *
constexpr int BLOCK = 256;
constexpr int UNROLL = 12;

// MODE 0: ds_add_f64 no return   1: ds_write_b64   2: ds_read_b64   3: ds_write_b128   4: ds_add_f32   5: ds_read_b128
//      6: ds_add_u64 no return   7: ds_add_u32 no return   9 / 10: ds_add_f64 from half the lanes (lower half / even lanes)
//      11: ds_add_f64 with lanes l and l + 32 on the same address
// PAT  0: lane -> consecutive doubles   1: (p & 7) * S + (p >> 3) with S = 2 * odd   2: all lanes of 16 different, random permutation
template <int MODE, int PAT>
__global__ void __launch_bounds__(BLOCK) k(int iters, int S, const int *perm, double *out) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8 * S * 2; i += BLOCK) lds[i] = 0.0;
    __syncthreads();
    int p = tid;
    if (PAT == 2) p = perm[tid];
    int sw = (PAT == 1) ? (p & 7) * S + (p >> 3) : p;
    const int PS = 8 * S;
    double acc = 0.0;
    double2 acc2 = {0.0, 0.0};
    const double v = 1.0 + tid;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            double *a = lds + (u & 1) * PS + sw;
            if (MODE == 0) __hip_atomic_fetch_add(a, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 1) *(LDSP(double))a = v;
            if (MODE == 2) acc += *(LDSP(double))a;
            if (MODE == 3) {
                double2 *a2 = (double2 *)(lds + (u & 1) * PS) + sw;
                *(LDSP(v2d))a2 = (v2d){v, v};
            }
            if (MODE == 4) __hip_atomic_fetch_add((float *)a, (float)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 9 && (tid & 63) < 32) __hip_atomic_fetch_add(a, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 10 && (tid & 1) == 0) __hip_atomic_fetch_add(a, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 11) __hip_atomic_fetch_add(lds + (u & 1) * PS + (sw & ~32), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 6)
                __hip_atomic_fetch_add((unsigned long long *)a, (unsigned long long)tid + 1ull, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 7) __hip_atomic_fetch_add((unsigned int *)a, (unsigned int)tid + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 5) {
                double2 *a2 = (double2 *)(lds + (u & 1) * PS) + sw;
                const v2d t = *(LDSP(v2d))a2;
                acc2.x += t.x;
            }
        }
    }
    __syncthreads();
    if (acc + acc2.x == 12345.678) out[0] = lds[tid];
    if (iters < 0) out[tid] = lds[tid];
}

template <int MODE, int PAT>
void run(const char *name, int wgs_per_cu, int S, const int *dperm, double *dout) {
    const int iters = 2000;
    const size_t lds = sizeof(double) * 8 * S * 2 * (MODE == 3 || MODE == 5 ? 2 : 1);
    (void)hipFuncSetAttribute((const void *)k<MODE, PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    const int grid = 256 * wgs_per_cu;
    k<MODE, PAT><<<grid, BLOCK, lds>>>(10, S, dperm, dout);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    k<MODE, PAT><<<grid, BLOCK, lds>>>(iters, S, dperm, dout);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    const double instr_per_cu = (double)wgs_per_cu * (BLOCK / 64) * iters * UNROLL;
    printf("%-34s wgs/cu=%d: %.3f ms, %.1f cycles per wave-instruction per CU (2.4 GHz)\n", name, wgs_per_cu, ms,
           ms * 1e-3 * 2.4e9 / instr_per_cu);
}

int main() {
    int hperm[BLOCK];
    for (int i = 0; i < BLOCK; ++i) hperm[i] = i;
    srand(3);
    for (int i = BLOCK - 1; i > 0; --i) {
        int j = rand() % (i + 1), t = hperm[i];
        hperm[i] = hperm[j];
        hperm[j] = t;
    }
    int *dperm;
    double *dout;
    (void)hipMalloc(&dperm, sizeof(hperm));
    (void)hipMalloc(&dout, 4096);
    (void)hipMemcpy(dperm, hperm, sizeof(hperm), hipMemcpyHostToDevice);
    const int S = 66;   // 2 * 33
    for (int w = 1; w <= 4; w *= 2) {
        run<0, 0>("ds_add_f64 linear", w, S, dperm, dout);
        run<0, 1>("ds_add_f64 swizzled", w, S, dperm, dout);
        run<0, 2>("ds_add_f64 random perm", w, S, dperm, dout);
        run<4, 0>("ds_add_f32 linear (8B stride)", w, S, dperm, dout);
        run<9, 0>("ds_add_f64 lanes 0..31 only", w, S, dperm, dout);
        run<10, 0>("ds_add_f64 even lanes only", w, S, dperm, dout);
        run<11, 0>("ds_add_f64 lanes l, l+32 same addr", w, S, dperm, dout);
        run<6, 0>("ds_add_u64 linear", w, S, dperm, dout);
        run<6, 1>("ds_add_u64 swizzled", w, S, dperm, dout);
        run<7, 0>("ds_add_u32 linear (8B stride)", w, S, dperm, dout);
        run<1, 0>("ds_write_b64 linear", w, S, dperm, dout);
        run<1, 1>("ds_write_b64 swizzled", w, S, dperm, dout);
        run<3, 0>("ds_write_b128 linear", w, S, dperm, dout);
        run<2, 0>("ds_read_b64 linear", w, S, dperm, dout);
        run<2, 1>("ds_read_b64 swizzled", w, S, dperm, dout);
        run<5, 0>("ds_read_b128 linear", w, S, dperm, dout);
    }
    return 0;
}
