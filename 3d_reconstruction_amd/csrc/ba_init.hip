// Linear steps on either side of the bundle adjustment, on the visible pairs only:
//   ba_triangulate  - every point from the cameras that see it      (get_X_from_C_Theta, p2w:1655-1721)
//   ba_resect       - every camera from the points it sees (DLT)     (get_C_Theta_from_X, p2w:1724-1808)
// Both stream the observation list once; the per-point 4x4 / per-camera 12x12 Gram matrices live
// in registers / LDS, the small dense solves are done in place (3x3 inverse; cyclic Jacobi).
//
// Rank tests: the reference applies numpy.linalg.matrix_rank to the row matrices themselves
// (tolerance: largest singular value x rows x eps).  Here only the Gram matrices exist, whose
// eigenvalues (squared singular values) carry an absolute error of a few eps x the largest one,
// so the test is  lambda_k > lambda_1 max((rows eps)^2, 1e-14)  (resection: eigenvalues from the
// Jacobi iteration; triangulation: pivots of a Cholesky factorisation with diagonal pivoting, which
// follow the eigenvalues within small factors):  the same verdict unless the k-th singular value
// is below about 1e-7 of the largest, where the solution has no correct digit anyway.
#include "ba_device.h"

#include <math.h>

namespace ba {

constexpr double RANK_FLOOR = 1e-14;

struct Intrinsics {
    double f, px, py;      // K matrix (p2w:1684-1695)
    double scale, sx, sy;  // pixel centre -> (centre - (sx, sy)) * scale  (p2w:1682, 1688)
};

static Intrinsics make_intrinsics(double phi_x, double W, double H, int normalize) {
    Intrinsics k;
    if (!normalize) {
        k.f = W / 2.0 / tan(phi_x / 2.0);
        k.px = W / 2.0;
        k.py = H / 2.0;
        k.scale = 1.0;
        k.sx = k.sy = 0.0;
    } else {
        const double m = W > H ? W : H;
        k.f = W / m / 2.0 / tan(phi_x / 2.0);
        k.px = k.py = 0.0;
        k.scale = 1.0 / m;
        k.sx = 0.5 * W;
        k.sy = 0.5 * H;
    }
    return k;
}

// P = K [R^T | -R^T C] per camera, 12 doubles row-major (p2w:1697-1700)
__global__ void k_projection_matrices(int64_t K, const double *__restrict__ C, const double *__restrict__ Theta,
                                      Intrinsics in, double *__restrict__ P) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= K) return;
    double R[9], cz, sz;
    rotation_yxz(Theta[3 * j], Theta[3 * j + 1], Theta[3 * j + 2], R, &cz, &sz);
    const double c0 = C[3 * j], c1 = C[3 * j + 1], c2 = C[3 * j + 2];
    double RT[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        RT[r][0] = R[r];          // row r of R^T = column r of R
        RT[r][1] = R[3 + r];
        RT[r][2] = R[6 + r];
        RT[r][3] = -(R[r] * c0 + R[3 + r] * c1 + R[6 + r] * c2);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        P[12 * j + q] = in.f * RT[0][q] + in.px * RT[2][q];
        P[12 * j + 4 + q] = in.f * RT[1][q] + in.py * RT[2][q];
        P[12 * j + 8 + q] = RT[2][q];
    }
}

// First three pivots of the Cholesky factorisation with diagonal pivoting of a symmetric positive
// semi-definite 4 x 4 matrix (fully unrolled, no indexed registers).  The pivots follow the eigenvalues
// within small factors, which is all the rank test needs: rank >= 3  <=>  d3 > threshold x d1.
// Nothing is swapped: the step that takes pivot p subtracts col_p col_p^T / a_pp from the WHOLE matrix, which
// zeroes row and column p (diagonal entry included), so that index is never the largest diagonal entry again;
// M holds the upper triangle row by row (00 01 02 03 11 12 13 22 23 33).
__device__ __forceinline__ void pivots_4x4(const double (&M)[10], double &d1, double &d3) {
    double a00 = M[0], a01 = M[1], a02 = M[2], a03 = M[3], a11 = M[4], a12 = M[5], a13 = M[6], a22 = M[7], a23 = M[8],
           a33 = M[9];
    double piv[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double m01 = fmax(a00, a11), m23 = fmax(a22, a33), best = fmax(m01, m23);
        const bool p0 = a00 == best, p1 = !p0 && a11 == best, p2 = !p0 && !p1 && a22 == best;
        // column p of the symmetric matrix
        const double c0 = p0 ? a00 : (p1 ? a01 : (p2 ? a02 : a03));
        const double c1 = p0 ? a01 : (p1 ? a11 : (p2 ? a12 : a13));
        const double c2 = p0 ? a02 : (p1 ? a12 : (p2 ? a22 : a23));
        const double c3 = p0 ? a03 : (p1 ? a13 : (p2 ? a23 : a33));
        piv[k] = best;
        const double inv = best > 0.0 ? fast_rcp_wide(best) : 0.0;
        const double s0 = c0 * inv, s1 = c1 * inv, s2 = c2 * inv, s3 = c3 * inv;
        a00 = fma(-s0, c0, a00);
        a01 = fma(-s0, c1, a01);
        a02 = fma(-s0, c2, a02);
        a03 = fma(-s0, c3, a03);
        a11 = fma(-s1, c1, a11);
        a12 = fma(-s1, c2, a12);
        a13 = fma(-s1, c3, a13);
        a22 = fma(-s2, c2, a22);
        a23 = fma(-s2, c3, a23);
        a33 = fma(-s3, c3, a33);
        // the pivot's own diagonal entry is zero by construction; rounding must not leave it a candidate
        a00 = p0 ? 0.0 : a00;
        a11 = p1 ? 0.0 : a11;
        a22 = p2 ? 0.0 : a22;
        a33 = (!p0 && !p1 && !p2) ? 0.0 : a33;
    }
    d1 = piv[0];
    d3 = piv[2];
}

// LPP lanes per point: lane q of a point's group takes its observations q, q + LPP, q + 2 LPP, ... (the group reads
// consecutive records, a wave's points one contiguous stretch of the list), TRIP of them per trip with their camera
// numbers requested together and then their projection rows and pixels together, and keeps its share of the ten sums
// of [A b]^T [A b] in registers.  The shares meet in LDS, where the workgroup's first 256 / LPP threads pick up one
// point each: the rank test and the 3 x 3 solve, some six hundred instructions per point, run on full waves.
// What the kernel costs at 1 M points / 10 M observations is a mixture, none of it dominant: the launch of its waves
// (15 us for the 62 500 waves of four lanes per point even when they return at once), 330 vector instructions per
// wave (half the kernel's time busy), and per wave a chain of memory latencies (ranges, camera numbers, then rows and
// pixels per trip) times the waves a SIMD holds.  Measured, caller's points in random order / in the engine's order
// (by first camera), us: one lane per point 155 / 89; four lanes, three observations per trip, shares added by DPP
// 91 / 82; four lanes through LDS, 3 / 2 / 1 per trip (126 / 96 / 66 registers) 91 / 75 / 70 and 89 at best in random
// order, where every row comes from L2 (a quarter of L2's bandwidth in 96-byte pieces); two lanes, 3 / 2 / 1 per trip
// 66 / 63 / 71 but 105-110 in random order (a lane's chain is five observations long); eight lanes 91-99.  Hence two
// lanes, two per trip where the points are known to be ordered by first camera (a problem's own list) and four lanes,
// one per trip otherwise.  Dropped: camera numbers of a whole track requested ahead of the trips (no change), the
// rows of a workgroup's camera window staged in LDS (100 us), larger workgroups (87 / 105 us at 512 / 1024 threads).
template <int LPP, int TRIP>
__global__ void __launch_bounds__(256) k_triangulate(int64_t N, const int64_t *__restrict__ row_ptr,
                                                     const int32_t *__restrict__ oj, const double2 *__restrict__ uv,
                                                     const double *__restrict__ P, Intrinsics in, double rank_thr,
                                                     double *__restrict__ X) {
    constexpr int PPB = 256 / LPP;
    constexpr int PSTRIDE = LPP * 10 + 2;          // doubles per point: a stride of 20 / 12 banks between the solver's lanes
    __shared__ double2 s_part[PPB * PSTRIDE / 2];
    __shared__ int s_deg[PPB];
    const int tid = threadIdx.x, ql = tid % LPP, pl = tid / LPP;
    const int64_t i_raw = (int64_t)blockIdx.x * PPB + pl;
    const int64_t i = i_raw < N ? i_raw : N - 1;
    const int64_t e0 = row_ptr[i], e1 = row_ptr[i + 1];
    double M[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};               // upper triangle of [A b]^T [A b], row-major
    for (int64_t base = e0 + ql; base < e1; base += LPP * TRIP) {
        int jn[TRIP];
#pragma unroll
        for (int t = 0; t < TRIP; ++t) jn[t] = base + LPP * t < e1 ? oj[base + LPP * t] : 0;
        double2 Pt[TRIP][6], ct[TRIP];
#pragma unroll
        for (int t = 0; t < TRIP; ++t) {
            if (base + LPP * t < e1) {
                const double2 *Pv = (const double2 *)(P + 12 * (int64_t)jn[t]);
#pragma unroll
                for (int q = 0; q < 6; ++q) Pt[t][q] = Pv[q];
                ct[t] = uv[base + LPP * t];
            }
        }
#pragma unroll
        for (int t = 0; t < TRIP; ++t) {
            if (base + LPP * t < e1) {
                double Pj[12];
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    Pj[2 * q] = Pt[t][q].x;
                    Pj[2 * q + 1] = Pt[t][q].y;
                }
                const double u = (ct[t].x - in.sx) * in.scale, v = (ct[t].y - in.sy) * in.scale;
                double a[4], b[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    a[q] = u * Pj[8 + q] - Pj[q];                 // p2w:1703
                    b[q] = v * Pj[8 + q] - Pj[4 + q];             // p2w:1704
                }
                int idx = 0;
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int q = p; q < 4; ++q) M[idx++] += a[p] * a[q] + b[p] * b[q];
            }
        }
    }
    {
        double2 *dst = s_part + (pl * PSTRIDE + ql * 10) / 2;
#pragma unroll
        for (int k = 0; k < 5; ++k) dst[k] = make_double2(M[2 * k], M[2 * k + 1]);
        if (ql == 0) s_deg[pl] = (int)(e1 - e0);
    }
    __syncthreads();
    if (tid >= PPB) return;
    const int64_t i_mine = (int64_t)blockIdx.x * PPB + tid;
    if (i_mine >= N) return;
    {
        const double2 *src = s_part + tid * PSTRIDE / 2;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            double2 v = src[k];
#pragma unroll
            for (int q = 1; q < LPP; ++q) {
                const double2 w = src[5 * q + k];
                v.x += w.x;
                v.y += w.y;
            }
            M[2 * k] = v.x;
            M[2 * k + 1] = v.y;
        }
    }
    const int deg = s_deg[tid];
    const double nan = __builtin_nan("");
    double x0 = nan, x1 = nan, x2 = nan;
    if (deg >= 2) {                                               // p2w:1671
        double d1, d3;
        pivots_4x4(M, d1, d3);
        if (d3 > d1 * rank_thr && d1 > 0.0) {                     // rank >= 3, p2w:1707
            // X = -(A3^T A3)^-1 A3^T b  (p2w:1716-1719): G = M[0,1,2; 4,5; 7], h = M[3], M[6], M[8]
            const double g00 = M[0], g01 = M[1], g02 = M[2], g11 = M[4], g12 = M[5], g22 = M[7];
            const double h0 = M[3], h1 = M[6], h2 = M[8];
            const double c00 = g11 * g22 - g12 * g12, c01 = g02 * g12 - g01 * g22, c02 = g01 * g12 - g02 * g11;
            const double det = g00 * c00 + g01 * c01 + g02 * c02;
            const double c11 = g00 * g22 - g02 * g02, c12 = g01 * g02 - g00 * g12, c22 = g00 * g11 - g01 * g01;
            const double id = fast_rcp_wide(det);
            x0 = -(c00 * h0 + c01 * h1 + c02 * h2) * id;
            x1 = -(c01 * h0 + c11 * h1 + c12 * h2) * id;
            x2 = -(c02 * h0 + c12 * h1 + c22 * h2) * id;
        }
    }
    X[3 * i_mine] = x0;
    X[3 * i_mine + 1] = x1;
    X[3 * i_mine + 2] = x2;
}

// cameras that see at least one point observed twice or more (p2w:1676): marked, then counted
__global__ void k_mark_selected_cameras(int64_t N, const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ oj,
                                        int *__restrict__ mark) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int64_t e0 = row_ptr[i], e1 = row_ptr[i + 1];
    if (e1 - e0 < 2) return;
    for (int64_t e = e0; e < e1; ++e) mark[oj[e]] = 1;
}

__global__ void k_count_marks(int64_t K, const int *__restrict__ mark, int *__restrict__ count) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long m = __ballot(j < K && mark[j] != 0);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, __popcll(m));
}

// ---------------------------------------------------------------------------------------------
// resection: one workgroup per camera
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void inv3(const double (&m)[3][3], double (&o)[3][3]) {
    const double c00 = m[1][1] * m[2][2] - m[1][2] * m[2][1], c01 = m[1][2] * m[2][0] - m[1][0] * m[2][2],
                 c02 = m[1][0] * m[2][1] - m[1][1] * m[2][0];
    const double id = 1.0 / (m[0][0] * c00 + m[0][1] * c01 + m[0][2] * c02);
    o[0][0] = c00 * id;
    o[1][0] = c01 * id;
    o[2][0] = c02 * id;
    o[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) * id;
    o[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) * id;
    o[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) * id;
    o[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) * id;
    o[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) * id;
    o[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) * id;
}

struct ResectArgs {
    int64_t K;
    const int64_t *col_ptr;   // K + 1: observation range of every camera in the camera-sorted list
    const int32_t *oi;        // point of every observation, camera-sorted
    const double2 *uv;        // pixel centre, camera-sorted
    const double *X;          // N x 3
    Intrinsics in;
    double rank_thr;
    int min_points;
    double *C, *Theta;        // K x 3 each
};

// NW waves share a camera's observations; the eigen-solver rotates six disjoint index pairs of the 12 x 12 matrix at a
// time (round-robin ordering: eleven rounds per sweep instead of 66 rotations one after the other, each a chain of a
// root, a reciprocal and a reciprocal root and three trips through LDS).  At 2 000 cameras x 5 000 observations, points
// numbered at random: 415 us with one rotation at a time -> 226 us, of which the sums take 150 (whatever NW is, 1 to
// 4: the gather of 24-byte point rows from all over a 24 MB array is what they wait for), the eigen-solver 60, the
// rest 14.
template <int NW>
__global__ void __launch_bounds__(64 * NW) k_resect(ResectArgs a) {
    constexpr int NT = 64 * NW;
    __shared__ double G[12][13];   // Gram matrix A^T A, overwritten by the Jacobi iteration
    __shared__ double V[12][13];   // accumulated rotations: eigenvectors in the columns
    __shared__ double part[NW][40];
    __shared__ double rc[12], rs[12];   // per index: cosine and signed sine of the rotation of its pair in this round
    __shared__ int partner[12];
    const int64_t j = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double nan = __builtin_nan("");
    const int64_t e0 = a.col_ptr[j], e1 = a.col_ptr[j + 1];
    if (e1 - e0 < a.min_points) {                                  // p2w:1739
        if (tid < 3) {
            a.C[3 * j + tid] = nan;
            a.Theta[3 * j + tid] = nan;
        }
        return;
    }
    // A^T A = [[S, 0, -Su], [0, S, -Sv], [-Su, -Sv, Sw]] with S = sum Xh Xh^T, Su = sum u Xh Xh^T, Sv = sum v Xh Xh^T,
    // Sw = sum (u^2 + v^2) Xh Xh^T over the camera's observations (rows p2w:1767-1770); 4 x 10 sums per lane
    double acc[4][10];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int k = 0; k < 10; ++k) acc[b][k] = 0.0;
    for (int64_t e = e0 + tid; e < e1; e += 2 * NT) {
        // two observations per trip, their loads in flight together
        const bool two = e + NT < e1;
        const int64_t i0 = a.oi[e], i1 = two ? a.oi[e + NT] : i0;
        const double2 c0 = a.uv[e], c1 = a.uv[two ? e + NT : e];
        const double x0[3] = {a.X[3 * i0], a.X[3 * i0 + 1], a.X[3 * i0 + 2]};
        const double x1[3] = {a.X[3 * i1], a.X[3 * i1 + 1], a.X[3 * i1 + 2]};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const double *x = h ? x1 : x0;
            const double2 c = h ? c1 : c0;
            // rows with NaN are zeroed (p2w:1771-1772)
            const bool live = (h == 0 || two) && x[0] == x[0] && x[1] == x[1] && x[2] == x[2];
            const double xh[4] = {live ? x[0] : 0.0, live ? x[1] : 0.0, live ? x[2] : 0.0, live ? 1.0 : 0.0};
            const double u = (c.x - a.in.sx) * a.in.scale, v = (c.y - a.in.sy) * a.in.scale;
            const double w = u * u + v * v;
            int idx = 0;
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int q = p; q < 4; ++q) {
                    const double s = xh[p] * xh[q];
                    acc[0][idx] += s;
                    acc[1][idx] += u * s;
                    acc[2][idx] += v * s;
                    acc[3][idx] += w * s;
                    ++idx;
                }
        }
    }
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const double t = wave_sum(acc[b][k]);
            if (lane == 0) part[wave][10 * b + k] = t;
        }
    __syncthreads();
    if (wave != 0) return;
    // From here on one wave is left.  The LDS serves a wave's instructions in order, so what its lanes hand to each
    // other through LDS needs no workgroup barrier, only that the compiler keeps the accesses in order.
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    if (lane < 40) {
        double t = part[0][lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) t += part[w][lane];
        part[0][lane] = t;
    }
    wave_sync();
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int el = lane + 64 * t;
        if (el < 144) {
            const int r = el / 12, c = el - 12 * r, br = r >> 2, bc = c >> 2;
            const int p = min(r & 3, c & 3), q = max(r & 3, c & 3);
            const int idx = 4 * p - p * (p - 1) / 2 + (q - p);          // (p, q), p <= q, in the row-major upper triangle
            double g;
            if (br == bc) g = part[0][(br == 2 ? 30 : 0) + idx];
            else if (br + bc == 1) g = 0.0;
            else g = -part[0][(br + bc == 2 ? 10 : 20) + idx];        // blocks (0, 2), (2, 0): -Su; (1, 2), (2, 1): -Sv
            G[r][c] = g;
            V[r][c] = r == c ? 1.0 : 0.0;
        }
    }
    wave_sync();

    // Jacobi iteration.  A rotation is skipped when |g_pq| <= 1e-16 sqrt(g_pp g_qq), the criterion that keeps the
    // relative accuracy of the small eigenpairs.  Round r pairs index 11 with r and (r + m) mod 11 with (r - m) mod 11,
    // m = 1 .. 5: every index is in exactly one pair per round and meets every other index once per sweep.  With
    // (c_x, s_x) the cosine and the signed sine of the pair of index x (minus for the pair's lower index) and x' its
    // partner, G <- J^T G J is  g'_rc = c_r (c_c g_rc + s_c g_rc') + s_r (c_c g_r'c + s_c g_r'c'),  V <- V J likewise.
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = false;
        for (int round = 0; round < 11; ++round) {
            bool did = false;
            if (lane < 6) {
                int p = lane == 0 ? 11 : (round + lane) % 11, q = lane == 0 ? round : (round + 11 - lane) % 11;
                if (p > q) {
                    const int t = p;
                    p = q;
                    q = t;
                }
                const double apq = G[p][q], app = G[p][p], aqq = G[q][q];
                double c = 1.0, s = 0.0;
                if (!(apq * apq <= 1e-32 * fabs(app * aqq))) {
                    did = true;
                    // t = sign(theta) / (|theta| + sqrt(theta^2 + 1)), theta = (g_qq - g_pp) / (2 g_pq), without the
                    // quotient theta: one root, one reciprocal, one reciprocal root, each a Newton iteration from the
                    // hardware's estimate (this chain is what a round waits for)
                    const double d = aqq - app, b = 2.0 * apq;
                    const double h = fma(d, d, b * b);
                    const double t = (d >= 0.0 ? b : -b) * fast_rcp_wide(fabs(d) + h * fast_rsqrt(h));
                    c = fast_rsqrt(fma(t, t, 1.0));
                    s = t * c;
                }
                rc[p] = c;
                rc[q] = c;
                rs[p] = -s;
                rs[q] = s;
                partner[p] = q;
                partner[q] = p;
            }
            rotated |= __ballot(did) != 0;
            wave_sync();
            double gn[3], vn[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int el = lane + 64 * t;
                if (el < 144) {
                    const int r = el / 12, c = el - 12 * r, pr = partner[r], pc = partner[c];
                    const double cr = rc[r], sr = rs[r], cc = rc[c], sc = rs[c];
                    gn[t] = cr * (cc * G[r][c] + sc * G[r][pc]) + sr * (cc * G[pr][c] + sc * G[pr][pc]);
                    vn[t] = cc * V[r][c] + sc * V[r][pc];
                }
            }
            wave_sync();
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int el = lane + 64 * t;
                if (el < 144) {
                    const int r = el / 12, c = el - 12 * r;
                    G[r][c] = gn[t];
                    V[r][c] = vn[t];
                }
            }
            wave_sync();
        }
        if (!rotated) break;
    }
    wave_sync();
    if (lane != 0) return;
    double lmax = 0.0, lmin = G[0][0];
    int kmin = 0;
    for (int k = 0; k < 12; ++k) {
        const double l = G[k][k];
        lmax = fmax(lmax, l);
        if (l < lmin) {
            lmin = l;
            kmin = k;
        }
    }
    int rank = 0;
    for (int k = 0; k < 12; ++k) rank += G[k][k] > lmax * a.rank_thr ? 1 : 0;
    if (rank < 11 || !(lmax > 0.0)) {                              // p2w:1774
        for (int k = 0; k < 3; ++k) {
            a.C[3 * j + k] = nan;
            a.Theta[3 * j + k] = nan;
        }
        return;
    }
    // P (3 x 4) = eigenvector of the smallest eigenvalue (p2w:1786-1792); RT_scaled = K^-1 P (p2w:1794)
    double Pm[3][4];
    for (int r = 0; r < 3; ++r)
        for (int q = 0; q < 4; ++q) Pm[r][q] = V[4 * r + q][kmin];
    const double fi = 1.0 / a.in.f;
    double M[3][3], T[3];
    for (int q = 0; q < 3; ++q) {
        M[0][q] = fi * (Pm[0][q] - a.in.px * Pm[2][q]);
        M[1][q] = fi * (Pm[1][q] - a.in.py * Pm[2][q]);
        M[2][q] = Pm[2][q];
    }
    T[0] = fi * (Pm[0][3] - a.in.px * Pm[2][3]);
    T[1] = fi * (Pm[1][3] - a.in.py * Pm[2][3]);
    T[2] = Pm[2][3];
    // orthogonal polar factor Q = U V^T of M (orthogonolize_matrix, p2w:241-242) by Newton's iteration
    // Q <- (Q + Q^-T) / 2; the mean singular value is trace(Q^T M) / 3
    double Q[3][3];
    for (int r = 0; r < 3; ++r)
        for (int q = 0; q < 3; ++q) Q[r][q] = M[r][q];
    for (int it = 0; it < 40; ++it) {
        double Qi[3][3];
        inv3(Q, Qi);
        double diff = 0.0, nrm = 0.0;
        for (int r = 0; r < 3; ++r)
            for (int q = 0; q < 3; ++q) {
                const double n = 0.5 * (Q[r][q] + Qi[q][r]);
                diff += (n - Q[r][q]) * (n - Q[r][q]);
                nrm += n * n;
                Q[r][q] = n;
            }
        if (diff <= 1e-32 * nrm) break;
    }
    double tr = 0.0;
    for (int r = 0; r < 3; ++r)
        for (int q = 0; q < 3; ++q) tr += Q[r][q] * M[r][q];
    const double sigma_mean = tr / 3.0;
    // R = Q^T (p2w:1801), T / mean(Sigma) (p2w:1802), C = -R T (p2w:1803)
    double R[3][3];
    for (int r = 0; r < 3; ++r)
        for (int q = 0; q < 3; ++q) R[r][q] = Q[q][r];
    for (int r = 0; r < 3; ++r)
        a.C[3 * j + r] = -(R[r][0] * T[0] + R[r][1] * T[1] + R[r][2] * T[2]) / sigma_mean;
    const double det = R[0][0] * (R[1][1] * R[2][2] - R[1][2] * R[2][1]) - R[0][1] * (R[1][0] * R[2][2] - R[1][2] * R[2][0]) +
                       R[0][2] * (R[1][0] * R[2][1] - R[1][1] * R[2][0]);
    const double sg = det > 0.0 ? 1.0 : (det < 0.0 ? -1.0 : 0.0);          // p2w:1805
    for (int r = 0; r < 3; ++r)
        for (int q = 0; q < 3; ++q) R[r][q] *= sg;
    const double Cx = sqrt(R[1][0] * R[1][0] + R[1][1] * R[1][1]);          // get_rotations, p2w:260-266
    a.Theta[3 * j + 0] = atan2(-R[1][2], Cx);
    a.Theta[3 * j + 1] = atan2(R[0][2] / Cx, R[2][2] / Cx);
    a.Theta[3 * j + 2] = atan2(R[1][0] / Cx, R[1][1] / Cx);
}

static int cdiv_n(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace ba

using namespace ba;

// points_by_first_camera: the caller vouches that neighbouring points see neighbouring cameras (a problem's own list)
static int triangulate_list(int64_t N, int64_t K, const int64_t *row_ptr, const int32_t *oj, const double *uv,
                            const double *C, const double *Theta, double phi_x, double W, double H, int normalize,
                            int64_t rank_rows, double *P_work, double *X_out, void *stream, bool points_by_first_camera) {
    BA_REQUIRE(N > 0 && K > 0 && row_ptr && oj && uv && C && Theta && P_work && X_out, "ba_triangulate: bad argument");
    BA_REQUIRE(phi_x > 0.0 && phi_x < 3.14159265358979323846, "ba_triangulate: phi_x outside (0, pi)");
    hipStream_t st = (hipStream_t)stream;
    const Intrinsics in = make_intrinsics(phi_x, W, H, normalize);
    const double tol = (double)(rank_rows > 4 ? rank_rows : 4) * 2.220446049250313e-16;
    const double thr = tol * tol > RANK_FLOOR ? tol * tol : RANK_FLOOR;
    hipLaunchKernelGGL(k_projection_matrices, dim3(cdiv_n(K, 256)), dim3(256), 0, st, K, C, Theta, in, P_work);
    if (points_by_first_camera)
        hipLaunchKernelGGL((k_triangulate<2, 2>), dim3(cdiv_n(N, 128)), dim3(256), 0, st, N, row_ptr, oj, (const double2 *)uv,
                           P_work, in, thr, X_out);
    else
        hipLaunchKernelGGL((k_triangulate<4, 1>), dim3(cdiv_n(N, 64)), dim3(256), 0, st, N, row_ptr, oj, (const double2 *)uv,
                           P_work, in, thr, X_out);
    BA_HIP(hipGetLastError());
    return BA_OK;
}

extern "C" {

int ba_triangulate(int64_t N, int64_t K, const int64_t *row_ptr, const int32_t *oj, const double *uv, const double *C,
                   const double *Theta, double phi_x, double W, double H, int normalize, int64_t rank_rows, double *P_work,
                   double *X_out, void *stream) {
    return triangulate_list(N, K, row_ptr, oj, uv, C, Theta, phi_x, W, H, normalize, rank_rows, P_work, X_out, stream, false);
}

int ba_problem_triangulate(ba_problem *p, const double *C, const double *Theta, double phi_x, int normalize,
                           double *P_work, double *X_out, void *stream) {
    BA_REQUIRE(p && C && Theta && P_work && X_out, "ba_problem_triangulate: null pointer");
    if (int rc = ensure_row_ptr(p)) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (p->tri_selected_cameras < 0) {      // once per problem: numpy's rank tolerance counts these cameras
        int *mark = nullptr;
        BA_HIP(hipMalloc(&mark, sizeof(int) * (p->K + 1)));
        hipError_t e = hipMemsetAsync(mark, 0, sizeof(int) * (p->K + 1), st);
        hipLaunchKernelGGL(k_mark_selected_cameras, dim3(cdiv_n(p->N, 256)), dim3(256), 0, st, p->N, p->row_ptr, p->oj, mark);
        hipLaunchKernelGGL(k_count_marks, dim3(cdiv_n(p->K, 256)), dim3(256), 0, st, p->K, mark, mark + p->K);
        int n = 0;
        if (e == hipSuccess) e = hipMemcpyAsync(&n, mark + p->K, sizeof(int), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        hipFree(mark);
        BA_HIP(e);
        p->tri_selected_cameras = n;
    }
    const int64_t rank_rows = 2 * (int64_t)p->tri_selected_cameras > 4 ? 2 * (int64_t)p->tri_selected_cameras : 4;
    return triangulate_list(p->N, p->K, p->row_ptr, p->oj, (const double *)p->ouv, C, Theta, phi_x, p->W, p->H, normalize,
                            rank_rows, P_work, X_out, stream, true);
}

int ba_resect(int64_t N, int64_t K, const int64_t *col_ptr, const int32_t *oi_by_camera, const double *uv_by_camera,
              const double *X, double phi_x, double W, double H, int normalize, int64_t rank_rows, double *C_out,
              double *Theta_out, void *stream) {
    BA_REQUIRE(N > 0 && K > 0 && col_ptr && oi_by_camera && uv_by_camera && X && C_out && Theta_out, "ba_resect: bad argument");
    BA_REQUIRE(phi_x > 0.0 && phi_x < 3.14159265358979323846, "ba_resect: phi_x outside (0, pi)");
    ResectArgs a;
    a.K = K;
    a.col_ptr = col_ptr;
    a.oi = oi_by_camera;
    a.uv = (const double2 *)uv_by_camera;
    a.X = X;
    a.in = make_intrinsics(phi_x, W, H, normalize);
    const double tol = (double)(rank_rows > 12 ? rank_rows : 12) * 2.220446049250313e-16;
    a.rank_thr = tol * tol > RANK_FLOOR ? tol * tol : RANK_FLOOR;
    a.min_points = 6;
    a.C = C_out;
    a.Theta = Theta_out;
    hipLaunchKernelGGL(k_resect<4>, dim3((unsigned)K), dim3(256), 0, (hipStream_t)stream, a);
    BA_HIP(hipGetLastError());
    return BA_OK;
}

}  // extern "C"
