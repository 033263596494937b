// Stand-alone forms of the reference's rotation and gauge helpers (p2w:138-140, 254-333): ryxz,
// get_rotations, align_scene, normalize_scene as calls of their own, for callers that use them
// outside the fit loop (the loop itself runs the fused camera step of ba_engine.hip).
// One-shot elementwise kernels: one thread per camera or point, one workgroup for the search of the
// camera farthest from the reference camera.
#include "ba_device.h"

namespace ba {

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// R[j] = Ry(theta_y) Rx(theta_x) Rz(theta_z)                                          (p2w:138-140)
__global__ void k_rotation_matrices(int64_t K, const double *__restrict__ Theta, double *__restrict__ R) {
    const int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (j >= K) return;
    double Rj[9], cz, sz;
    rotation_yxz(Theta[3 * j], Theta[3 * j + 1], Theta[3 * j + 2], Rj, &cz, &sz);
#pragma unroll
    for (int k = 0; k < 9; ++k) R[9 * j + k] = Rj[k];
}

// Euler angles of a rotation matrix, theta_x in [-pi/2, pi/2]                         (p2w:260-266)
__device__ __forceinline__ void angles_of(const double *Rn, double *th) {
    const double Cx = sqrt(Rn[3] * Rn[3] + Rn[4] * Rn[4]);
    th[0] = atan2(-Rn[5], Cx);
    th[1] = atan2(Rn[2] / Cx, Rn[8] / Cx);
    th[2] = atan2(Rn[3] / Cx, Rn[4] / Cx);
}

__global__ void k_rotation_angles(int64_t K, const double *__restrict__ R, double *__restrict__ Theta) {
    const int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (j >= K) return;
    double Rn[9], th[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) Rn[k] = R[9 * j + k];
    angles_of(Rn, th);
#pragma unroll
    for (int k = 0; k < 3; ++k) Theta[3 * j + k] = th[k];
}

// Rotation of camera 0 (from its angles, or row 0 of R) into frame[0..8], its centre into frame[9..11]
__global__ void k_align_frame(const double *__restrict__ C, const double *__restrict__ Theta,
                              const double *__restrict__ R, double *__restrict__ frame) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double R0[9], cz, sz;
    if (Theta) {
        rotation_yxz(Theta[0], Theta[1], Theta[2], R0, &cz, &sz);
    } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) R0[k] = R[k];
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) frame[k] = R0[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) frame[9 + k] = C[k];
}

// rows <- (rows - C0) @ R0 for the n rows of `in` (points or centres)                  (p2w:285-294)
__global__ void k_align_rows(int64_t n, const double *__restrict__ in, const double *__restrict__ frame,
                             double *__restrict__ out) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d0 = in[3 * i] - frame[9], d1 = in[3 * i + 1] - frame[10], d2 = in[3 * i + 2] - frame[11];
#pragma unroll
    for (int k = 0; k < 3; ++k) out[3 * i + k] = d0 * frame[k] + d1 * frame[3 + k] + d2 * frame[6 + k];
}

// R_new[j] = R0^T R[j] (R[j] built from the angles when Theta is given), Theta_new[j] = its angles  (p2w:296-300)
__global__ void k_align_cameras(int64_t K, const double *__restrict__ Theta, const double *__restrict__ R,
                                const double *__restrict__ frame, double *__restrict__ Theta_out,
                                double *__restrict__ R_out) {
    const int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (j >= K) return;
    double Rj[9], Rn[9], th[3], cz, sz;
    if (Theta) {
        rotation_yxz(Theta[3 * j], Theta[3 * j + 1], Theta[3 * j + 2], Rj, &cz, &sz);
    } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) Rj[k] = R[9 * j + k];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) Rn[3 * a + b] = frame[a] * Rj[b] + frame[3 + a] * Rj[3 + b] + frame[6 + a] * Rj[6 + b];
    angles_of(Rn, th);
#pragma unroll
    for (int k = 0; k < 3; ++k) Theta_out[3 * j + k] = th[k];
    if (R_out) {
#pragma unroll
        for (int k = 0; k < 9; ++k) R_out[9 * j + k] = Rn[k];
    }
}

// scale[0] = dist_scale / |C[j1] - C[j0]|, scale[1] = j1; j1 < 0: the camera farthest from camera 0, the first
// one among equals as numpy.argmax picks it                                             (p2w:320-326)
__global__ void __launch_bounds__(1024) k_baseline(int64_t K, const double *__restrict__ C, int64_t j0, int64_t j1,
                                                   double dist_scale, double *__restrict__ scale) {
    __shared__ double best_d[1024];
    __shared__ long long best_j[1024];
    const int tid = threadIdx.x;
    if (j1 < 0) {
        double bd = -1.0;
        long long bj = 1;
        for (int64_t j = 1 + tid; j < K; j += 1024) {
            const double a = C[3 * j] - C[0], b = C[3 * j + 1] - C[1], c = C[3 * j + 2] - C[2];
            const double d = sqrt(a * a + b * b + c * c);
            if (d > bd) {
                bd = d;
                bj = j;
            }
        }
        best_d[tid] = bd;
        best_j[tid] = bj;
        __syncthreads();
        for (int s = 512; s > 0; s >>= 1) {
            if (tid < s) {
                const double od = best_d[tid + s];
                const long long oj = best_j[tid + s];
                if (od > best_d[tid] || (od == best_d[tid] && oj < best_j[tid])) {
                    best_d[tid] = od;
                    best_j[tid] = oj;
                }
            }
            __syncthreads();
        }
        j0 = 0;
        j1 = best_j[0];
    }
    if (tid == 0) {
        const double a = C[3 * j1] - C[3 * j0], b = C[3 * j1 + 1] - C[3 * j0 + 1], c = C[3 * j1 + 2] - C[3 * j0 + 2];
        scale[0] = dist_scale / sqrt(a * a + b * b + c * c);
        scale[1] = (double)j1;
    }
}

__global__ void k_scale_rows(int64_t n3, const double *__restrict__ in, const double *__restrict__ scale,
                             double *__restrict__ out) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n3) out[i] = in[i] * scale[0];
}

}  // namespace ba

using namespace ba;

extern "C" {

int ba_rotation_matrices(int64_t K, const double *Theta, double *R_out, void *stream) {
    BA_REQUIRE(K >= 0 && (K == 0 || (Theta && R_out)), "ba_rotation_matrices: null pointer");
    if (K == 0) return BA_OK;
    hipLaunchKernelGGL(k_rotation_matrices, dim3(cdiv64(K, 256)), dim3(256), 0, (hipStream_t)stream, K, Theta, R_out);
    BA_HIP(hipGetLastError());
    return BA_OK;
}

int ba_rotation_angles(int64_t K, const double *R, double *Theta_out, void *stream) {
    BA_REQUIRE(K >= 0 && (K == 0 || (R && Theta_out)), "ba_rotation_angles: null pointer");
    if (K == 0) return BA_OK;
    hipLaunchKernelGGL(k_rotation_angles, dim3(cdiv64(K, 256)), dim3(256), 0, (hipStream_t)stream, K, R, Theta_out);
    BA_HIP(hipGetLastError());
    return BA_OK;
}

int ba_align_scene(int64_t N, int64_t K, const double *X_in, const double *C_in, const double *Theta_in,
                   const double *R_in, double *X_out, double *C_out, double *Theta_out, double *R_out,
                   double *frame_work, void *stream) {
    BA_REQUIRE(K >= 1 && N >= 0, "ba_align_scene: needs at least one camera");
    BA_REQUIRE((Theta_in != nullptr) != (R_in != nullptr), "ba_align_scene: exactly one of Theta_in, R_in");
    BA_REQUIRE(C_in && C_out && Theta_out && frame_work && (N == 0 || (X_in && X_out)), "ba_align_scene: null pointer");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_align_frame, dim3(1), dim3(64), 0, st, C_in, Theta_in, R_in, frame_work);
    // cameras first: their kernel reads the input angles, which C_out / Theta_out may alias
    hipLaunchKernelGGL(k_align_cameras, dim3(cdiv64(K, 256)), dim3(256), 0, st, K, Theta_in, R_in, frame_work,
                       Theta_out, R_out);
    hipLaunchKernelGGL(k_align_rows, dim3(cdiv64(K, 256)), dim3(256), 0, st, K, C_in, frame_work, C_out);
    if (N > 0) hipLaunchKernelGGL(k_align_rows, dim3(cdiv64(N, 256)), dim3(256), 0, st, N, X_in, frame_work, X_out);
    BA_HIP(hipGetLastError());
    return BA_OK;
}

int ba_normalize_scene(int64_t N, int64_t K, const double *X_in, const double *C_in, int64_t j0, int64_t j1,
                       double dist_scale, double *X_out, double *C_out, double *scale_work, void *stream) {
    BA_REQUIRE(N >= 0 && C_in && C_out && scale_work && (N == 0 || (X_in && X_out)), "ba_normalize_scene: null pointer");
    if (j1 < 0) {
        BA_REQUIRE(K >= 2, "ba_normalize_scene: needs at least two cameras");
        j0 = 0;
    } else {
        BA_REQUIRE(j0 >= 0 && j0 < K && j1 < K, "ba_normalize_scene: camera index out of range");
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_baseline, dim3(1), dim3(1024), 0, st, K, C_in, j0, j1, dist_scale, scale_work);
    hipLaunchKernelGGL(k_scale_rows, dim3(cdiv64(3 * K, 256)), dim3(256), 0, st, 3 * K, C_in, scale_work, C_out);
    if (N > 0) hipLaunchKernelGGL(k_scale_rows, dim3(cdiv64(3 * N, 256)), dim3(256), 0, st, 3 * N, X_in, scale_work, X_out);
    BA_HIP(hipGetLastError());
    return BA_OK;
}

}  // extern "C"
