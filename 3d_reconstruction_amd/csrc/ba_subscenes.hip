// Joint alignment of subscenes by gradient descent (fit_subscenes_GD, p2w:729-949) on the present entries.
//
// P subscenes hold their own estimates x_ij of N points (an entry is absent where the reference's array holds
// NaN).  Parameters per subscene: scale S_j, shift T_j, angles Psi_j; x_new_ij = S_j R_j x_ij + T_j; the energy
// is the spread of every point's entries around their mean.  The parameter vector is [S (P) | T (3P) | Psi (3P)].
//   ba_subscenes_eval       energy sum and gradient sums              (p2w:813-818, 848-849, 886-924)
//   ba_subscenes_step       update rule and gauge (subscene 0 = identity)   (p2w:926-937)
//   ba_subscenes_transform  x_new of every entry                      (p2w:943-944)
// Entries are sorted by point; one thread walks the entries of one point (three short sweeps: mean, energy and
// mean of the differences, gradient sums), the per-subscene sums are added up in LDS first when 7P doubles fit.
#include "ba_device.h"

#include <math.h>

namespace ba {

// per-subscene table: [0] S, [1..3] T, [4..12] R = Ry Rx Rz row-major, [13 + 9 l + 3 k + n] = S dR[k][n]/dPsi_l
constexpr int SUB_ROW = 40;
constexpr int SUB_LDS_SUMS = 4096;      // doubles of LDS the evaluation may use for the per-subscene sums

__device__ __forceinline__ void mat3_mul(const double *A, const double *B, double *C) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}

__global__ void k_subscene_table(int64_t P, const double *__restrict__ params, double *__restrict__ tab) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= P) return;
    const double S = params[j];
    const double *T = params + P + 3 * j, *Psi = params + 4 * P + 3 * j;
    double sx, cx, sy, cy, sz, cz;
    sincos(Psi[0], &sx, &cx);
    sincos(Psi[1], &sy, &cy);
    sincos(Psi[2], &sz, &cz);
    // factor matrices (p2w:114-136) and their derivatives (p2w:905-915)
    const double Rx[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx}, dRx[9] = {0, 0, 0, 0, -sx, -cx, 0, cx, -sx};
    const double Ry[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy}, dRy[9] = {-sy, 0, cy, 0, 0, 0, -cy, 0, -sy};
    const double Rz[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1}, dRz[9] = {-sz, -cz, 0, cz, -sz, 0, 0, 0, 0};
    double *row = tab + SUB_ROW * j;
    row[0] = S;
    row[1] = T[0];
    row[2] = T[1];
    row[3] = T[2];
    double t[9], m[9];
    mat3_mul(Ry, Rx, t);
    mat3_mul(t, Rz, m);                        // R (p2w:808)
    for (int q = 0; q < 9; ++q) row[4 + q] = m[q];
    mat3_mul(Ry, dRx, t);
    mat3_mul(t, Rz, m);                        // p2w:917
    for (int q = 0; q < 9; ++q) row[13 + q] = S * m[q];
    mat3_mul(dRy, Rx, t);
    mat3_mul(t, Rz, m);                        // p2w:918
    for (int q = 0; q < 9; ++q) row[22 + q] = S * m[q];
    mat3_mul(Ry, Rx, t);
    mat3_mul(t, dRz, m);                       // p2w:919
    for (int q = 0; q < 9; ++q) row[31 + q] = S * m[q];
}

struct SubEntry {
    double xr[3], xn[3];
};
__device__ __forceinline__ SubEntry sub_forward(const double *__restrict__ row, const double *__restrict__ x) {
    SubEntry e;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        e.xr[k] = row[4 + 3 * k] * x[0] + row[5 + 3 * k] * x[1] + row[6 + 3 * k] * x[2];     // p2w:813
        e.xn[k] = row[0] * e.xr[k] + row[1 + k];                                             // p2w:814
    }
    return e;
}

// sums: [0] sum of squared differences, [1 + j] dE/dS_j, [1 + P + 3j + k] dE/dT_jk, [1 + 4P + 3j + l] dE/dPsi_jl,
// all without the reference's factors 1 / (2 n) and 2 / n.
template <bool LDS_SUMS>
__global__ void __launch_bounds__(256) k_subscene_eval(int64_t N, int64_t P, const int64_t *__restrict__ row_ptr,
                                                       const int32_t *__restrict__ ej, const double *__restrict__ ex,
                                                       const double *__restrict__ tab, double *__restrict__ sums) {
    extern __shared__ double acc[];
    const int n_acc = (int)(7 * P);
    if (LDS_SUMS) {
        for (int q = threadIdx.x; q < n_acc; q += 256) acc[q] = 0.0;
        __syncthreads();
    }
    double *dst = LDS_SUMS ? acc : sums + 1;
    double e_sum = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
        const int64_t e0 = row_ptr[i], e1 = row_ptr[i + 1];
        if (e1 <= e0) continue;
        const double inv = 1.0 / (double)(e1 - e0);
        double mean[3] = {0, 0, 0};
        for (int64_t e = e0; e < e1; ++e) {
            const SubEntry s = sub_forward(tab + SUB_ROW * (int64_t)ej[e], ex + 3 * e);
#pragma unroll
            for (int k = 0; k < 3; ++k) mean[k] += s.xn[k];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) mean[k] *= inv;                       // p2w:817
        double m2[3] = {0, 0, 0};
        for (int64_t e = e0; e < e1; ++e) {
            const SubEntry s = sub_forward(tab + SUB_ROW * (int64_t)ej[e], ex + 3 * e);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double d = s.xn[k] - mean[k];                        // p2w:818
                e_sum += d * d;                                            // p2w:848-849
                m2[k] += d;
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) m2[k] *= inv;
        for (int64_t e = e0; e < e1; ++e) {
            const int64_t j = ej[e];
            const double *row = tab + SUB_ROW * j, *x = ex + 3 * e;
            const SubEntry s = sub_forward(row, x);
            double dc[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) dc[k] = (s.xn[k] - mean[k]) - m2[k];                  // p2w:886
            atomicAdd(dst + j, dc[0] * s.xr[0] + dc[1] * s.xr[1] + dc[2] * s.xr[2]);         // p2w:888
#pragma unroll
            for (int k = 0; k < 3; ++k) atomicAdd(dst + P + 3 * j + k, dc[k]);               // p2w:890
#pragma unroll
            for (int l = 0; l < 3; ++l) {                                                     // p2w:921-924
                const double *D = row + 13 + 9 * l;
                double g = 0.0;
#pragma unroll
                for (int k = 0; k < 3; ++k) g += dc[k] * (D[3 * k] * x[0] + D[3 * k + 1] * x[1] + D[3 * k + 2] * x[2]);
                atomicAdd(dst + 4 * P + 3 * j + l, g);
            }
        }
    }
    // one atomic per workgroup on the energy's single address (the memory side serves the atomics of one address one
    // after the other: one per wave of 1 563 workgroups was 0.1 ms of the 0.14 ms this kernel took at 400 k points)
    __shared__ double e_wave[4];
    e_sum = wave_sum(e_sum);
    if ((threadIdx.x & 63) == 0) e_wave[threadIdx.x >> 6] = e_sum;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sums, (e_wave[0] + e_wave[1]) + (e_wave[2] + e_wave[3]));
    if (LDS_SUMS) {
        for (int q = threadIdx.x; q < n_acc; q += 256) {
            const double v = acc[q];
            if (v != 0.0) atomicAdd(sums + 1 + q, v);
        }
    }
}

// One workgroup.  Update rule on all 7P parameters (p2w:926-930), then the gauge (p2w:932-937):
//   T <- (T - T_0) R_0 / S_0,  R <- R_0^T R,  Psi <- angles of R,  S <- |S| / |S_0|.
__global__ void __launch_bounds__(256) k_subscene_step(int64_t P, UpdateRule u, const double *__restrict__ sums,
                                                       const double *__restrict__ params, double *__restrict__ params_out,
                                                       const double *__restrict__ s_in, double *__restrict__ s_out,
                                                       const double *__restrict__ r_in, double *__restrict__ r_out) {
    __shared__ double first[7];       // updated S_0, T_0, Psi_0
    const int64_t n = 7 * P;
    for (int64_t q = threadIdx.x; q < n; q += 256) {
        double s = 0.0, r = 0.0;
        if (u.mode == 2) {
            s = s_in[q];
            r = r_in[q];
        }
        params_out[q] = apply_rule(u, params[q], sums[1 + q], &s, &r);
        if (u.mode == 2) {
            s_out[q] = s;
            r_out[q] = r;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        first[0] = params_out[0];
        for (int k = 0; k < 3; ++k) {
            first[1 + k] = params_out[P + k];
            first[4 + k] = params_out[4 * P + k];
        }
    }
    __syncthreads();
    double R0[9], cz, sz;
    rotation_yxz(first[4], first[5], first[6], R0, &cz, &sz);
    const double S0 = first[0];
    for (int64_t j = threadIdx.x; j < P; j += 256) {
        double *T = params_out + P + 3 * j, *Psi = params_out + 4 * P + 3 * j;
        double R[9];
        rotation_yxz(Psi[0], Psi[1], Psi[2], R, &cz, &sz);                 // p2w:932
        const double d[3] = {T[0] - first[1], T[1] - first[2], T[2] - first[3]};
#pragma unroll
        for (int l = 0; l < 3; ++l) T[l] = (d[0] * R0[l] + d[1] * R0[3 + l] + d[2] * R0[6 + l]) / S0;     // p2w:933
        double Rn[9];
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
            for (int c = 0; c < 3; ++c) Rn[3 * l + c] = R0[l] * R[c] + R0[3 + l] * R[3 + c] + R0[6 + l] * R[6 + c];   // p2w:934
        const double Cx = sqrt(Rn[3] * Rn[3] + Rn[4] * Rn[4]);             // get_rotations, p2w:260-266
        Psi[0] = atan2(-Rn[5], Cx);
        Psi[1] = atan2(Rn[2] / Cx, Rn[8] / Cx);
        Psi[2] = atan2(Rn[3] / Cx, Rn[4] / Cx);
        params_out[j] = fabs(params_out[j]) / fabs(S0);                    // p2w:936-937
    }
}

__global__ void k_subscene_transform(int64_t n, const int32_t *__restrict__ ej, const double *__restrict__ ex,
                                     const double *__restrict__ tab, double *__restrict__ x_new) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const SubEntry s = sub_forward(tab + SUB_ROW * (int64_t)ej[e], ex + 3 * e);
#pragma unroll
    for (int k = 0; k < 3; ++k) x_new[3 * e + k] = s.xn[k];
}

static unsigned blocks_for(int64_t n, int64_t cap) {
    int64_t b = (n + 255) / 256;
    if (b > cap) b = cap;
    return (unsigned)(b < 1 ? 1 : b);
}

}  // namespace ba

using namespace ba;

extern "C" {

int ba_subscenes_eval(int64_t N, int64_t P, const int64_t *row_ptr, const int32_t *entry_subscene, const double *entry_x,
                      const double *params, double *table_work, double *sums, void *stream) {
    BA_REQUIRE(N > 0 && P > 0 && row_ptr && entry_subscene && entry_x && params && table_work && sums,
               "ba_subscenes_eval: bad argument");
    hipStream_t st = (hipStream_t)stream;
    BA_HIP(hipMemsetAsync(sums, 0, sizeof(double) * (size_t)(1 + 7 * P), st));
    hipLaunchKernelGGL(k_subscene_table, dim3(blocks_for(P, 1 << 20)), dim3(256), 0, st, P, params, table_work);
    const unsigned grid = blocks_for(N, 1024);      // (every workgroup ends with up to 7 P atomics on the same addresses)
    if (7 * P <= SUB_LDS_SUMS)
        hipLaunchKernelGGL(k_subscene_eval<true>, dim3(grid), dim3(256), sizeof(double) * (size_t)(7 * P), st, N, P, row_ptr,
                           entry_subscene, entry_x, table_work, sums);
    else
        hipLaunchKernelGGL(k_subscene_eval<false>, dim3(grid), dim3(256), 0, st, N, P, row_ptr, entry_subscene, entry_x,
                           table_work, sums);
    BA_HIP(hipGetLastError());
    return BA_OK;
}

int ba_subscenes_step(int64_t P, int rule, double lr, double bc1, double bc2, double n_entries, const double *sums,
                      const double *params, double *params_out, const double *s_in, double *s_out, const double *r_in,
                      double *r_out, void *stream) {
    BA_REQUIRE(P > 0 && sums && params && params_out && params_out != params, "ba_subscenes_step: bad argument");
    BA_REQUIRE(rule == 1 || rule == 2, "ba_subscenes_step: rule must be 1 (SGD) or 2 (Adam)");
    BA_REQUIRE(rule == 1 || (s_in && s_out && r_in && r_out), "ba_subscenes_step: Adam needs moment buffers");
    BA_REQUIRE(n_entries > 0.0, "ba_subscenes_step: no entries");
    UpdateRule u;
    u.mode = rule;
    u.lr = lr;
    u.gscale = 2.0 / n_entries;        // p2w:888, 890, 923
    u.ibc1 = 1.0 / bc1;
    u.ibc2 = 1.0 / bc2;
    hipLaunchKernelGGL(k_subscene_step, dim3(1), dim3(256), 0, (hipStream_t)stream, P, u, sums, params, params_out, s_in,
                       s_out, r_in, r_out);
    BA_HIP(hipGetLastError());
    return BA_OK;
}

int ba_subscenes_transform(int64_t P, int64_t n_entries, const int32_t *entry_subscene, const double *entry_x,
                           const double *params, double *table_work, double *x_new, void *stream) {
    BA_REQUIRE(P > 0 && n_entries > 0 && entry_subscene && entry_x && params && table_work && x_new,
               "ba_subscenes_transform: bad argument");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_subscene_table, dim3(blocks_for(P, 1 << 20)), dim3(256), 0, st, P, params, table_work);
    hipLaunchKernelGGL(k_subscene_transform, dim3(blocks_for(n_entries, (int64_t)1 << 30)), dim3(256), 0, st, n_entries,
                       entry_subscene, entry_x, table_work, x_new);
    BA_HIP(hipGetLastError());
    return BA_OK;
}

}  // extern "C"
