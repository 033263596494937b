// Homography misfit of every camera pair (get_H_diff_matrix, p2w:1945-1995) on the camera-sorted observation list.
//
// One wave per pair (j1 < j2).  The common points are found by binary search of the points of j1 in the sorted
// point list of j2 (pairs whose point ranges do not overlap leave at once: in video tracks that is nearly every
// pair).  With x1 = (u1, v1, 1) and the pixel (u2, v2) of the same point in j2, the rows of the reference's matrix
// are (-x1, 0, u2 x1), (0, -x1, v2 x1), so A^T A = [[M, 0, -Mu], [0, M, -Mv], [-Mu, -Mv, Mw]] with four weighted
// sums of x1 x1^T (weights 1, u2, v2, u2^2 + v2^2): 24 accumulators per lane.  The eigenvector of the smallest
// eigenvalue comes from a cyclic Jacobi iteration with the relative stopping rule (it keeps the accuracy of the
// small eigenpairs that forming A^T A would otherwise cost); a second sweep over the common points adds up the
// squared transfer error.
#include "ba_device.h"

#include <math.h>

namespace ba {

struct PairArgs {
    int64_t K;
    const int64_t *col_ptr;   // K + 1: observation range of every camera in the camera-sorted list
    const int32_t *oi;        // point of every observation, ascending within a camera
    const double2 *uv;        // pixel centre
    double scale, sx, sy;     // centre -> (centre - (sx, sy)) * scale   (p2w:1960-1964)
    int min_common;           // p2w:1974
    double *Hd;               // K x K, NaN-filled by the caller of the kernel
};

// index of `key` in the ascending list a[lo .. hi), or -1
__device__ __forceinline__ int64_t find_point(const int32_t *__restrict__ a, int64_t lo, int64_t hi, int32_t key) {
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        const int32_t v = a[mid];
        if (v == key) return mid;
        if (v < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    return -1;
}

__global__ void k_fill_nan(int64_t n, double *__restrict__ out) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) out[q] = __builtin_nan("");
}

__global__ void __launch_bounds__(64) k_pair_homography(PairArgs a) {
    __shared__ double G[9][10];
    __shared__ double V[9][10];
    const int64_t j2 = blockIdx.x, j1 = blockIdx.y;
    if (j1 >= j2) return;
    const int lane = threadIdx.x;
    const int64_t b0 = a.col_ptr[j1], b1 = a.col_ptr[j1 + 1], c0 = a.col_ptr[j2], c1 = a.col_ptr[j2 + 1];
    if (b1 - b0 < a.min_common || c1 - c0 < a.min_common) return;
    const int32_t lo2 = a.oi[c0], hi2 = a.oi[c1 - 1];
    if (a.oi[b1 - 1] < lo2 || a.oi[b0] > hi2) return;

    double acc[4][6];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[b][k] = 0.0;
    int count = 0;
    for (int64_t e = b0 + lane; e < b1; e += 64) {
        const int32_t pt = a.oi[e];
        if (pt < lo2 || pt > hi2) continue;
        const int64_t m = find_point(a.oi, c0, c1, pt);
        if (m < 0) continue;
        ++count;
        const double2 p1 = a.uv[e], p2 = a.uv[m];
        const double x1[3] = {(p1.x - a.sx) * a.scale, (p1.y - a.sy) * a.scale, 1.0};
        const double u = (p2.x - a.sx) * a.scale, v = (p2.y - a.sy) * a.scale;
        const double w = u * u + v * v;
        int idx = 0;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = p; q < 3; ++q) {
                const double s = x1[p] * x1[q];
                acc[0][idx] += s;
                acc[1][idx] += u * s;
                acc[2][idx] += v * s;
                acc[3][idx] += w * s;
                ++idx;
            }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) count += __shfl_xor(count, off);
    if (count < a.min_common) return;                                      // p2w:1974 (uniform over the wave)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[b][k] = wave_sum(acc[b][k]);
    if (lane == 0) {
        int idx = 0;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = p; q < 3; ++q) {
                const double s = acc[0][idx], su = acc[1][idx], sv = acc[2][idx], sw = acc[3][idx];
                ++idx;
                for (int t = 0; t < 2; ++t) {      // both (p, q) and (q, p)
                    const int r = t ? q : p, c = t ? p : q;
                    G[r][c] = s;
                    G[3 + r][3 + c] = s;
                    G[6 + r][6 + c] = sw;
                    G[r][3 + c] = 0.0;
                    G[3 + r][c] = 0.0;
                    G[r][6 + c] = -su;
                    G[6 + c][r] = -su;
                    G[3 + r][6 + c] = -sv;
                    G[6 + c][3 + r] = -sv;
                }
            }
    }
    if (lane < 9)
        for (int k = 0; k < 9; ++k) V[lane][k] = lane == k ? 1.0 : 0.0;
    __syncthreads();

    // cyclic Jacobi: lane k < 9 owns row / column k of the updates, every lane computes the same rotation
    // (skipped when |g_pq| <= 1e-16 sqrt(g_pp g_qq)), so the control flow is uniform
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < 8; ++p)
            for (int q = p + 1; q < 9; ++q) {
                const double apq = G[p][q], app = G[p][p], aqq = G[q][q];
                if (apq * apq <= 1e-32 * fabs(app * aqq)) continue;
                rotated = true;
                const double theta = (aqq - app) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                __syncthreads();
                if (lane < 9) {
                    const double akp = G[lane][p], akq = G[lane][q];
                    G[lane][p] = c * akp - s * akq;
                    G[lane][q] = s * akp + c * akq;
                    const double vkp = V[lane][p], vkq = V[lane][q];
                    V[lane][p] = c * vkp - s * vkq;
                    V[lane][q] = s * vkp + c * vkq;
                }
                __syncthreads();
                if (lane < 9) {
                    const double apk = G[p][lane], aqk = G[q][lane];
                    G[p][lane] = c * apk - s * aqk;
                    G[q][lane] = s * apk + c * aqk;
                }
                __syncthreads();
            }
        if (!rotated) break;
    }
    __syncthreads();
    int kmin = 0;
    double lmin = G[0][0];
    for (int k = 1; k < 9; ++k)
        if (G[k][k] < lmin) {
            lmin = G[k][k];
            kmin = k;
        }
    double h[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) h[k] = V[k][kmin];                          // p2w:1983-1987

    // mean squared transfer error of the common points (p2w:1989-1993)
    double err = 0.0;
    for (int64_t e = b0 + lane; e < b1; e += 64) {
        const int32_t pt = a.oi[e];
        if (pt < lo2 || pt > hi2) continue;
        const int64_t m = find_point(a.oi, c0, c1, pt);
        if (m < 0) continue;
        const double2 p1 = a.uv[e], p2 = a.uv[m];
        const double x = (p1.x - a.sx) * a.scale, y = (p1.y - a.sy) * a.scale;
        const double u = (p2.x - a.sx) * a.scale, v = (p2.y - a.sy) * a.scale;
        const double yw = h[6] * x + h[7] * y + h[8];
        const double du = u - (h[0] * x + h[1] * y + h[2]) / yw, dv = v - (h[3] * x + h[4] * y + h[5]) / yw;
        err += du * du + dv * dv;
    }
    err = wave_sum(err);
    if (lane == 0) {
        const double val = err / (double)count;
        a.Hd[j1 * a.K + j2] = val;
        a.Hd[j2 * a.K + j1] = val;
    }
}

}  // namespace ba

using namespace ba;

extern "C" {

int ba_homography_diff_matrix(int64_t K, const int64_t *col_ptr, const int32_t *oi_by_camera, const double *uv_by_camera,
                              double W, double H, int normalize, int min_common, double *H_diff, void *stream) {
    BA_REQUIRE(K > 0 && K <= 65535 && col_ptr && oi_by_camera && uv_by_camera && H_diff,
               "ba_homography_diff_matrix: bad argument (K must be within 1 .. 65535)");
    BA_REQUIRE(!normalize || (W > 0.0 && H > 0.0), "ba_homography_diff_matrix: normalize needs W and H");
    BA_REQUIRE(min_common >= 1, "ba_homography_diff_matrix: min_common must be positive");
    hipStream_t st = (hipStream_t)stream;
    PairArgs a;
    a.K = K;
    a.col_ptr = col_ptr;
    a.oi = oi_by_camera;
    a.uv = (const double2 *)uv_by_camera;
    a.scale = 1.0;
    a.sx = a.sy = 0.0;
    if (normalize) {
        const double m = W > H ? W : H;
        a.scale = 1.0 / m;
        a.sx = 0.5 * W;
        a.sy = 0.5 * H;
    }
    a.min_common = min_common;
    a.Hd = H_diff;
    const int64_t n = K * K;
    hipLaunchKernelGGL(k_fill_nan, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, H_diff);
    if (K > 1) hipLaunchKernelGGL(k_pair_homography, dim3((unsigned)K, (unsigned)K), dim3(64), 0, st, a);
    BA_HIP(hipGetLastError());
    return BA_OK;
}

}  // extern "C"
