// Second-order extension of the engine (no counterpart in the reference, which is first-order
// only): Gauss-Newton blocks J^T J in the same fused observation pass, the Schur complement onto
// the cameras and the back-substitution for the points.  The dense solve is in ba_chol.hip.
//
// Unknown order of the reduced system: camera j owns rows 6j .. 6j+5 (C_x, C_y, C_z, theta_x,
// theta_y, theta_z), phi_x is row 6K; n = 6K + 1, stored row-major with leading dimension
// ld = n rounded up to a multiple of 64 (padding rows/columns are the identity).
//
// Per observation (camera frame, see ba_engine.hip): u_0 = fi (1, 0, -m0), u_1 = fi (0, 1, -m1) are the
// rows of d model / d q (fi = f / q2).  Then
//   J_X   rows jx_a = R u_a                     (world frame)
//   J_cam rows G_a T,  G_a = (-u_a, u_a x q),  T = blockdiag(R^T, [a_x a_y a_z])
//   J_phi rows dphi m_a,  dphi = -f / sin(phi_x)
// so per camera the blocks are accumulated in the camera frame (B' = sum G^T G, b' = sum G^T m) and
// rotated once per camera afterwards.
#include "ba_device.h"
#include "ba_runpass.h"
#include "ba_blockpass.h"

#include <hip/hip_ext.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace ba {

constexpr int CAMV = 33;  // per camera: force (3), torque (3), B' upper triangle (21), b' (6)
constexpr int PTV = 9;    // per point: gradient (3), V upper triangle (6)

struct JtjArgs {
    const uint32_t *opack;
    const double2 *ouv;
    const TileMeta *meta;
    const int32_t *cam_list;
    int K;
    int cam_cap, pt_cap;
    const double *X;
    const double *T;
    const double *scalars;
    double *gX;        // N x 3
    double *V;         // N x 6
    double *cam_sums;  // CAMV x K, component-major (pre-zeroed)
    double *partials;  // per tile {sum r^2, sum m.r, sum m.m}
};

// per-observation quantities shared by the J^T J pass and the Schur kernels
struct ObsJac {
    double r0, r1, m0, m1, fi;
    double q0, q1, q2;
    double jx0[3], jx1[3];  // rows of J_X
    double G0[6], G1[6];    // rows of the camera-frame camera Jacobian
};

// valid = false (a zero-weight record, any point / camera pair): every Jacobian entry and m come out
// as zero (they carry the factor 1 / q2, which is replaced by 0)
// ROW_T: the last three entries of the row are t = -R^T C (stage_camera_row_rt) instead of C
template <bool ROW_T = false>
__device__ __forceinline__ ObsJac obs_jacobian(const double *x, const double2 *rw, double f, double pcx, double pcy,
                                               double2 uv, bool valid = true) {
    const double2 t0 = rw[0], t1 = rw[1], t2 = rw[2], t3 = rw[3], t4 = rw[4], t5 = rw[5];
    const double R00 = t0.x, R01 = t0.y, R02 = t1.x, R10 = t1.y, R11 = t2.x, R12 = t2.y, R20 = t3.x, R21 = t3.y,
                 R22 = t4.x;
    ObsJac o;
    if (ROW_T) {
        o.q0 = fma(R00, x[0], fma(R10, x[1], fma(R20, x[2], t4.y)));
        o.q1 = fma(R01, x[0], fma(R11, x[1], fma(R21, x[2], t5.x)));
        o.q2 = fma(R02, x[0], fma(R12, x[1], fma(R22, x[2], t5.y)));
    } else {
        const double d0 = x[0] - t4.y, d1 = x[1] - t5.x, d2 = x[2] - t5.y;
        o.q0 = R00 * d0 + R10 * d1 + R20 * d2;
        o.q1 = R01 * d0 + R11 * d1 + R21 * d2;
        o.q2 = R02 * d0 + R12 * d1 + R22 * d2;
    }
    const double inv = valid ? fast_rcp(o.q2) : 0.0;
    o.m0 = o.q0 * inv;
    o.m1 = o.q1 * inv;
    o.r0 = fma(f, o.m0, pcx - uv.x);
    o.r1 = fma(f, o.m1, pcy - uv.y);
    const double fi = f * inv;
    o.fi = fi;
    o.jx0[0] = fi * (R00 - o.m0 * R02);
    o.jx0[1] = fi * (R10 - o.m0 * R12);
    o.jx0[2] = fi * (R20 - o.m0 * R22);
    o.jx1[0] = fi * (R01 - o.m1 * R02);
    o.jx1[1] = fi * (R11 - o.m1 * R12);
    o.jx1[2] = fi * (R21 - o.m1 * R22);
    o.G0[0] = -fi;
    o.G0[1] = 0.0;
    o.G0[2] = fi * o.m0;
    o.G0[3] = fi * (o.m0 * o.q1);
    o.G0[4] = -fi * (o.m0 * o.q0 + o.q2);
    o.G0[5] = fi * o.q1;
    o.G1[0] = 0.0;
    o.G1[1] = -fi;
    o.G1[2] = fi * o.m1;
    o.G1[3] = fi * (o.q2 + o.m1 * o.q1);
    o.G1[4] = -fi * (o.m1 * o.q0);
    o.G1[5] = -fi * o.q0;
    return o;
}

// --------------------------------------------------------------------------------------------
// K3: fused residual + Jacobian + J^T J pass on LDS tiles
// --------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TILE_BLOCK) k_jtj_tiled(JtjArgs a) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int t = blockIdx.x;
    const TileMeta tm = a.meta[t];
    const int e0 = tm.e0, e1 = tm.e1, p0 = tm.p0, np = tm.np, nc = tm.nc;
    const int64_t cb = tm.cb;
    __shared__ double red[3][TILE_WAVES];
    if (e0 >= e1) {
        if (tid < 3) a.partials[3 * (int64_t)t + tid] = 0.0;
        return;
    }
    double *rows = lds;                                    // cam_cap x ROW_LDS
    double *cacc = rows + (size_t)a.cam_cap * ROW_LDS;     // CAMV x nc (component-major)
    double *xs = cacc + (size_t)a.cam_cap * CAMV;          // np x 3
    double *pacc = xs + (size_t)a.pt_cap * 3;              // np x PTV

    for (int idx = tid; idx < nc * (ROW_LDS / 2); idx += TILE_BLOCK) {
        const int row = idx / (ROW_LDS / 2), piece = idx - row * (ROW_LDS / 2);
        const int j = tm.j0 >= 0 ? tm.j0 + row : a.cam_list[cb + row];
        ((double2 *)rows)[idx] = ((const double2 *)(a.T + (int64_t)j * CAM_ROW))[piece];
    }
    for (int idx = tid; idx < CAMV * nc; idx += TILE_BLOCK) cacc[idx] = 0.0;
    for (int idx = tid; idx < 3 * np; idx += TILE_BLOCK) xs[idx] = a.X[3 * (int64_t)p0 + idx];
    for (int idx = tid; idx < PTV * np; idx += TILE_BLOCK) pacc[idx] = 0.0;
    const double f = a.scalars[SC_F], pcx = a.scalars[SC_CX], pcy = a.scalars[SC_CY];
    __syncthreads();

    double esum = 0.0, psum = 0.0, hsum = 0.0;
    for (int base = e0 + wid * 64; base < e1; base += TILE_BLOCK) {
        const int e = base + lane;
        const bool valid = e < e1;
        const int ec = valid ? e : e1 - 1;
        const uint32_t pack = a.opack[ec];
        const int li = valid ? (int)(pack >> 16) : -2;
        const int slot = (int)(pack & 0xffffu);
        const double2 uv = a.ouv[ec];
        ObsJac o = obs_jacobian(xs + 3 * (valid ? li : 0), (const double2 *)(rows + slot * ROW_LDS), f, pcx, pcy, uv);
        const double vm = valid ? 1.0 : 0.0;
        const double r0 = o.r0 * vm, r1 = o.r1 * vm;
        esum += r0 * r0 + r1 * r1;
        psum += o.m0 * r0 + o.m1 * r1;
        hsum += vm * (o.m0 * o.m0 + o.m1 * o.m1);

        if (valid) {
            double cv[CAMV];
            // force g = u0 r0 + u1 r1 = -(G0 r0 + G1 r1)[0..2]; torque q x g = -(G0 r0 + G1 r1)[3..5]
#pragma unroll
            for (int k = 0; k < 6; ++k) cv[k] = -(o.G0[k] * r0 + o.G1[k] * r1);
            int idx = 6;
#pragma unroll
            for (int p = 0; p < 6; ++p)
#pragma unroll
                for (int q = p; q < 6; ++q) cv[idx++] = o.G0[p] * o.G0[q] + o.G1[p] * o.G1[q];
#pragma unroll
            for (int k = 0; k < 6; ++k) cv[27 + k] = o.G0[k] * o.m0 + o.G1[k] * o.m1;
            double *acc = cacc + slot;
#pragma unroll
            for (int k = 0; k < CAMV; ++k)
                __hip_atomic_fetch_add(acc + k * nc, cv[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }

        // point side: 9 values, segmented scan inside 16-lane rows, run tails add to LDS
        double pv[PTV];
#pragma unroll
        for (int k = 0; k < 3; ++k) pv[k] = o.jx0[k] * r0 + o.jx1[k] * r1;
        pv[3] = vm * (o.jx0[0] * o.jx0[0] + o.jx1[0] * o.jx1[0]);
        pv[4] = vm * (o.jx0[0] * o.jx0[1] + o.jx1[0] * o.jx1[1]);
        pv[5] = vm * (o.jx0[0] * o.jx0[2] + o.jx1[0] * o.jx1[2]);
        pv[6] = vm * (o.jx0[1] * o.jx0[1] + o.jx1[1] * o.jx1[1]);
        pv[7] = vm * (o.jx0[1] * o.jx0[2] + o.jx1[1] * o.jx1[2]);
        pv[8] = vm * (o.jx0[2] * o.jx0[2] + o.jx1[2] * o.jx1[2]);

        const int lane16 = lane & 15;
        const int ip = row_shr<1>(li);
        const bool head = (lane16 == 0) || (ip != li);
        const unsigned long long hm = __ballot(head);
        const unsigned long long below = hm & (~0ull >> (63 - lane));
        const int dist = lane - (63 - __clzll(below));
#pragma unroll
        for (int k = 0; k < PTV; ++k) {
            double v = pv[k];
            double s = row_shr<1>(v);
            if (dist >= 1) v += s;
            s = row_shr<2>(v);
            if (dist >= 2) v += s;
            s = row_shr<4>(v);
            if (dist >= 4) v += s;
            s = row_shr<8>(v);
            if (dist >= 8) v += s;
            pv[k] = v;
        }
        const bool tail = (lane == 63) || ((hm >> (lane + 1)) & 1ull);
        if (tail && li >= 0) {
            double *pa = pacc + PTV * li;
#pragma unroll
            for (int k = 0; k < PTV; ++k)
                __hip_atomic_fetch_add(pa + k, pv[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    esum = wave_sum(esum);
    psum = wave_sum(psum);
    hsum = wave_sum(hsum);
    if (lane == 0) {
        red[0][wid] = esum;
        red[1][wid] = psum;
        red[2][wid] = hsum;
    }
    __syncthreads();
    if (tid < 3) a.partials[3 * (int64_t)t + tid] = (red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]);

    // every point is written by exactly one tile (unobserved ones get zeros)
    const int lo = tm.wlo - p0, hi = tm.whi - p0;
    for (int idx = 3 * lo + tid; idx < 3 * hi; idx += TILE_BLOCK) {
        const int pt = idx >= 0 ? idx / 3 : -1, k = idx - 3 * pt;
        const bool in = idx >= 0 && pt < np;
        a.gX[3 * (int64_t)p0 + idx] = in ? pacc[PTV * pt + k] : 0.0;
    }
    for (int idx = 6 * lo + tid; idx < 6 * hi; idx += TILE_BLOCK) {
        const int pt = idx >= 0 ? idx / 6 : -1, k = idx - 6 * pt;
        a.V[6 * (int64_t)p0 + idx] = (idx >= 0 && pt < np) ? pacc[PTV * pt + 3 + k] : 0.0;
    }
    for (int idx = tid; idx < CAMV * nc; idx += TILE_BLOCK) {
        const int comp = idx / nc, slot = idx - comp * nc;
        const int j = tm.j0 >= 0 ? tm.j0 + slot : a.cam_list[cb + slot];
        __hip_atomic_fetch_add(a.cam_sums + (int64_t)comp * a.K + j, cacc[idx], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The same pass on camera runs / dense blocks (see ba_runpass.h, ba_blockpass.h).  Camera sums in compact form:
// with a = m0, b = m1, s = a^2 + b^2 and fi = f / q2 the rows of the camera-frame Jacobian are
//   G0 = fi (-1, 0, a, q2 a b, -q2 (a^2 + 1), q2 b),   G1 = fi (0, -1, b, q2 (1 + b^2), -q2 a b, -q2 a),
// and fi q2 = f, so every entry of G^T G, G^T r and G^T m is a constant (1, f, f^2) times a sum of monomials in
// a, b weighted by 1, fi or fi^2.  The pass accumulates those 27 sums (CAMV_C) and k_jtj_finalize expands them
// to the 33 classic ones: 45 instead of 61 instructions per observation on the camera side, 6 fewer registers.
//   0..5   fi r0, fi r1, -fi m.r, b m.r + r1, a m.r + r0, a r1 - b r0        (force; torque / (-f, f, f))
//   6..9   w, w a, w b, w s                      with w = fi^2                 (centre x centre block)
//   10..17 u, u a, u b, u a b, u a^2, u b^2, u b (s + 1), u a (s + 1)   with u = fi   (centre x angle, / f)
//   18..24 a, b, s, a b (s + 2), a^2 b^2, (1 + b^2)^2, (1 + a^2)^2       (angle x angle, / f^2)
//   25..26 b (s + 1), a (s + 1)                 (camera x phi_x column; its third entry fi s is sums 14 + 15)
constexpr int CAMV_C = 27;
struct JtjOp {
    // register budget: three waves per SIMD with the 8-byte records, two with the long ones
    static constexpr int NCV = CAMV_C, NPV = PTV, NPART = 2, NOUT = 2, WAVES_PER_EU = 2, WAVES_PER_EU_COMPACT = 3;
    static constexpr int MOMENT_PREFETCH = 0;      // this pass never carries the point update
    // per-point sums: gradient (3) -> gX, V upper triangle (6) -> V
    __host__ __device__ static constexpr int out_width(int o) { return o == 1 ? 6 : 3; }
    __host__ __device__ static constexpr int out_offset(int o) { return o == 0 ? 0 : 3; }
    __host__ __device__ static constexpr bool cv_used(int) { return true; }
    static constexpr int NCV_LDS = CAMV_C;
    __host__ __device__ static constexpr int cv_lds(int k) { return k; }
    __host__ __device__ static constexpr int cv_from_lds(int c) { return c; }
    double f, pcx, pcy;
    __device__ explicit JtjOp(const double *scalars, double pix_shift = 0.0)
        : f(scalars[SC_F]), pcx(scalars[SC_CX] - pix_shift), pcy(scalars[SC_CY] - pix_shift) {}
    __device__ __forceinline__ double cv_flush(int, double v) const { return v; }
    __device__ __forceinline__ void obs(bool valid, const double (&x)[3], const double2 (&rw)[6], double2 uv,
                                        double (&cv)[CAMV_C], double (&pv)[PTV], double (&part)[2]) const {
        const ObsJac o = obs_jacobian<true>(x, rw, f, pcx, pcy, uv, valid);    // m = 0, fi = 0 for a zero-weight record
        const double r0 = valid ? o.r0 : 0.0, r1 = valid ? o.r1 : 0.0, vm = valid ? 1.0 : 0.0;
        const double a = o.m0, b = o.m1, fi = o.fi;
        const double a2 = a * a, b2 = b * b, s = a2 + b2, ab = a * b, s1 = s + 1.0, s2 = s1 + 1.0;
        const double a1 = vm + a2, b1 = vm + b2, a2b2 = a2 * b2, w = fi * fi, ua = fi * a, ub = fi * b;
        const double mr = fma(a, r0, b * r1);
        part[0] = fma(r0, r0, fma(r1, r1, part[0]));
        part[1] += mr;      // (the phi x phi entry, sum of m.m, is the sum over the cameras of sum 20)
        cv[0] = fma(fi, r0, cv[0]);
        cv[1] = fma(fi, r1, cv[1]);
        cv[2] = fma(-fi, mr, cv[2]);
        cv[3] = fma(b, mr, cv[3] + r1);
        cv[4] = fma(a, mr, cv[4] + r0);
        cv[5] = fma(a, r1, fma(-b, r0, cv[5]));
        cv[6] += w;
        cv[7] = fma(w, a, cv[7]);
        cv[8] = fma(w, b, cv[8]);
        cv[9] = fma(w, s, cv[9]);
        cv[10] += fi;
        cv[11] += ua;
        cv[12] += ub;
        cv[13] = fma(ua, b, cv[13]);
        cv[14] = fma(ua, a, cv[14]);
        cv[15] = fma(ub, b, cv[15]);
        cv[16] = fma(ub, s1, cv[16]);
        cv[17] = fma(ua, s1, cv[17]);
        cv[18] += a;
        cv[19] += b;
        cv[20] += s;
        cv[21] = fma(ab, s2, cv[21]);
        cv[22] += a2b2;
        cv[23] = fma(b1, b1, cv[23]);
        cv[24] = fma(a1, a1, cv[24]);
        cv[25] = fma(b, s1, cv[25]);
        cv[26] = fma(a, s1, cv[26]);
#pragma unroll
        for (int k = 0; k < 3; ++k) pv[k] = fma(o.jx0[k], r0, o.jx1[k] * r1);
        pv[3] = fma(o.jx0[0], o.jx0[0], o.jx1[0] * o.jx1[0]);
        pv[4] = fma(o.jx0[0], o.jx0[1], o.jx1[0] * o.jx1[1]);
        pv[5] = fma(o.jx0[0], o.jx0[2], o.jx1[0] * o.jx1[2]);
        pv[6] = fma(o.jx0[1], o.jx0[1], o.jx1[1] * o.jx1[1]);
        pv[7] = fma(o.jx0[1], o.jx0[2], o.jx1[1] * o.jx1[2]);
        pv[8] = fma(o.jx0[2], o.jx0[2], o.jx1[2] * o.jx1[2]);
    }
};

// T = blockdiag(R^T, A), A = [a_x a_y a_z] columns: row mu, column alpha of the 6x6 frame change
__device__ __forceinline__ void camera_frame_change(const double *row, double T[6][6]) {
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int q = 0; q < 6; ++q) T[p][q] = 0.0;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int q = 0; q < 3; ++q) T[p][q] = row[3 * q + p];  // R^T
    // a_x = (cos tz, -sin tz, 0), a_y = row 1 of R, a_z = e_z
    T[3][3] = row[12];
    T[4][3] = -row[13];
    T[5][3] = 0.0;
    T[3][4] = row[3];
    T[4][4] = row[4];
    T[5][4] = row[5];
    T[3][5] = 0.0;
    T[4][5] = 0.0;
    T[5][5] = 1.0;
}

// camera-frame sums -> gradient, B (21), bphi (6) per camera; scalars
__global__ void __launch_bounds__(256) k_jtj_finalize(int64_t K, int n_partials, const double *__restrict__ partials,
                                                      const double *__restrict__ cam_sums, const double *__restrict__ T,
                                                      const double *__restrict__ scalars, double *__restrict__ E_sum,
                                                      double *__restrict__ gphi, double *__restrict__ hphi,
                                                      double *__restrict__ gC, double *__restrict__ gTheta,
                                                      double *__restrict__ B, double *__restrict__ bphi, int compact) {
    const double dphi = scalars[SC_DPHI];
    if (blockIdx.x == 0) {
        __shared__ double red[3][4];
        double acc[3] = {0.0, 0.0, 0.0};
        if (compact) {      // tile sums {r.r, m.r}; m.m from the cameras' sums of s
            for (int t = threadIdx.x; t < n_partials; t += 256) {
                acc[0] += partials[2 * t];
                acc[1] += partials[2 * t + 1];
            }
            for (int64_t jj = threadIdx.x; jj < K; jj += 256) acc[2] += cam_sums[20 * K + jj];
        } else {
            for (int t = threadIdx.x; t < n_partials; t += 256)
#pragma unroll
                for (int k = 0; k < 3; ++k) acc[k] += partials[3 * t + k];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            acc[k] = wave_sum(acc[k]);
            if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = acc[k];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            E_sum[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
            gphi[0] = dphi * ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
            hphi[0] = dphi * dphi * ((red[2][0] + red[2][1]) + (red[2][2] + red[2][3]));
        }
    }
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= K) return;
    const double *row = T + j * CAM_ROW;
    double Tm[6][6];
    camera_frame_change(row, Tm);
    double c[CAMV];
    if (compact) {      // the 27 sums of JtjOp -> force, torque, upper triangle of the camera-frame block, phi column
        double t[CAMV_C];
#pragma unroll
        for (int k = 0; k < CAMV_C; ++k) t[k] = cam_sums[(int64_t)k * K + j];
        const double f = scalars[SC_F], f2 = f * f;
        c[0] = t[0];
        c[1] = t[1];
        c[2] = t[2];
        c[3] = -f * t[3];
        c[4] = f * t[4];
        c[5] = f * t[5];
        c[6] = t[6];                    // (0,0)
        c[7] = 0.0;                     // (0,1)
        c[8] = -t[7];                   // (0,2)
        c[9] = -f * t[13];              // (0,3)
        c[10] = f * (t[14] + t[10]);    // (0,4)
        c[11] = -f * t[12];             // (0,5)
        c[12] = t[6];                   // (1,1)
        c[13] = -t[8];                  // (1,2)
        c[14] = -f * (t[10] + t[15]);   // (1,3)
        c[15] = f * t[13];              // (1,4)
        c[16] = f * t[11];              // (1,5)
        c[17] = t[9];                   // (2,2)
        c[18] = f * t[16];              // (2,3)
        c[19] = -f * t[17];             // (2,4)
        c[20] = 0.0;                    // (2,5)
        c[21] = f2 * (t[22] + t[23]);   // (3,3)
        c[22] = -f2 * t[21];            // (3,4)
        c[23] = -f2 * t[18];            // (3,5)
        c[24] = f2 * (t[24] + t[22]);   // (4,4)
        c[25] = -f2 * t[19];            // (4,5)
        c[26] = f2 * t[20];             // (5,5)
        c[27] = -t[11];
        c[28] = -t[12];
        c[29] = t[14] + t[15];
        c[30] = f * t[25];
        c[31] = -f * t[26];
        c[32] = 0.0;
    } else {
#pragma unroll
        for (int k = 0; k < CAMV; ++k) c[k] = cam_sums[(int64_t)k * K + j];
    }
    // gradient: sum J_cam^T r = T^T sum G^T r = -T^T (force, torque)
#pragma unroll
    for (int al = 0; al < 6; ++al) {
        double v = 0.0;
#pragma unroll
        for (int mu = 0; mu < 6; ++mu) v += Tm[mu][al] * c[mu];
        if (al < 3) gC[3 * j + al] = -v;
        else gTheta[3 * j + al - 3] = -v;
    }
    double Bp[6][6];
    int idx = 6;
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int q = p; q < 6; ++q) {
            Bp[p][q] = c[idx];
            Bp[q][p] = c[idx];
            ++idx;
        }
    double M[6][6];  // B' T
#pragma unroll
    for (int mu = 0; mu < 6; ++mu)
#pragma unroll
        for (int be = 0; be < 6; ++be) {
            double v = 0.0;
#pragma unroll
            for (int nu = 0; nu < 6; ++nu) v += Bp[mu][nu] * Tm[nu][be];
            M[mu][be] = v;
        }
    idx = 0;
#pragma unroll
    for (int al = 0; al < 6; ++al)
#pragma unroll
        for (int be = al; be < 6; ++be) {
            double v = 0.0;
#pragma unroll
            for (int mu = 0; mu < 6; ++mu) v += Tm[mu][al] * M[mu][be];
            B[21 * j + idx] = v;
            ++idx;
        }
#pragma unroll
    for (int al = 0; al < 6; ++al) {
        double v = 0.0;
#pragma unroll
        for (int mu = 0; mu < 6; ++mu) v += Tm[mu][al] * c[27 + mu];
        bphi[6 * j + al] = dphi * v;
    }
}

// The block profile of the reduced system before any point has contributed: every row starts at its own camera's
// first column (a camera's six unknowns can lie across a block boundary), the row of phi_x at column 0, padding
// rows on the diagonal.
__device__ __forceinline__ int schur_first_of_block(int64_t ib, int64_t K) {
    const int64_t row = 64 * ib;
    if (row > 6 * K) return (int)ib;
    if (row + 63 >= 6 * K) return 0;
    return (int)((6 * (row / 6)) / 64);
}

// --------------------------------------------------------------------------------------------
// K4: Schur complement onto the cameras.  S starts as the damped camera / phi blocks; every point
// subtracts Y^T V'^-1 Y, where Y holds its 3x6 cross blocks W_ij and its 3x1 phi column.
// --------------------------------------------------------------------------------------------
__global__ void k_schur_init(int64_t K, int64_t n, int64_t ld, double lambda, const double *__restrict__ B,
                             const double *__restrict__ bphi, const double *__restrict__ hphi,
                             const double *__restrict__ gC, const double *__restrict__ gTheta,
                             const double *__restrict__ gphi, double *__restrict__ S, double *__restrict__ rhs,
                             int *__restrict__ first) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (first && j < ld / 64) first[j] = schur_first_of_block(j, K);       // (ld / 64 <= K: the grid covers it)
    if (j < K) {
        int idx = 0;
        for (int p = 0; p < 6; ++p)
            for (int q = p; q < 6; ++q) {
                double v = B[21 * j + idx++];
                if (p == q) v *= 1.0 + lambda;
                S[(6 * j + q) * ld + 6 * j + p] = v;       // lower triangle only: nothing reads the other one
            }
        for (int p = 0; p < 6; ++p) {
            S[(6 * K) * ld + 6 * j + p] = bphi[6 * j + p];
            rhs[6 * j + p] = -(p < 3 ? gC[3 * j + p] : gTheta[3 * j + p - 3]);
        }
    }
    if (j == 0) {
        S[(6 * K) * ld + 6 * K] = hphi[0] * (1.0 + lambda);
        rhs[6 * K] = -gphi[0];
        for (int64_t t = n; t < ld; ++t) {
            S[t * ld + t] = 1.0;
            rhs[t] = 0.0;
        }
    }
}

// inverse of the damped symmetric 3x3 point block
__device__ __forceinline__ void inv_sym3(const double *V6, double lambda, double *inv /*6, upper*/) {
    const double a = V6[0] * (1.0 + lambda), b = V6[1], c = V6[2], d = V6[3] * (1.0 + lambda), e = V6[4],
                 g = V6[5] * (1.0 + lambda);
    const double A = d * g - e * e, Bc = c * e - b * g, Cc = b * e - c * d;
    const double det = a * A + b * Bc + c * Cc;
    const double id = fast_rcp_wide(det);
    inv[0] = A * id;
    inv[1] = Bc * id;
    inv[2] = Cc * id;
    inv[3] = (a * g - c * c) * id;
    inv[4] = (b * c - a * e) * id;
    inv[5] = (a * d - b * b) * id;
}

struct SchurArgs {
    const int32_t *oj;
    const double2 *ouv;
    const int64_t *row_ptr;  // N + 1: observation range of every point
    int64_t N, K, ld;
    double lambda;
    const double *X, *T, *scalars;
    const double *V, *gX;
    double *Xphi;            // N x 3, written: point x phi column
    double *S, *rhs;
    int max_deg;
    int *first;              // optional: first occupied 64-column block of every 64-row block of S (atomic minimum)
};

// Tile form of the Schur kernel: a workgroup owns an LDS tile (a few hundred consecutive points
// and the ~10 cameras that see them), sums every point's contribution -Y^T V'^-1 Y to the tile's
// (6 nc + 1)^2 corner of S in LDS (atomics from lanes that hold different entries), and only then
// adds the lower block triangle of that corner to the global matrix: ~2.4 k global atomics per
// tile instead of ~2 k per point.
struct SchurTileArgs {
    const uint32_t *opack;   // point-major: (point - tile's first point) << 16 | camera slot
    const double2 *ouv;
    const int64_t *row_ptr;
    const TileMeta *meta;
    const int32_t *cam_list;
    int64_t K, ld;
    int cam_cap, max_deg;
    double lambda;
    const double *X, *T, *scalars;
    const double *V, *gX;
    double *Xphi;            // N x 3, written: point x phi column
    double *S, *rhs;
    int *first;              // optional: first occupied 64-column block of every 64-row block of S (atomic minimum)
    const int4 *sub = nullptr;   // k_schur_gemm: its ranges of points (one workgroup each)
    const int *only = nullptr;   // k_schur_tile: if given, the tiles it deals with (flag per tile)
};

__host__ __device__ inline size_t schur_tile_lds_doubles(int cam_cap, int max_deg, int waves = 16) {
    const size_t m = 6 * (size_t)cam_cap;
    // per wave: W, Z and the slots of one point of any degree and of a second one of at most ten observations
    return (size_t)cam_cap * (CAM_ROW + 18) + 18 * (size_t)cam_cap * (cam_cap + 1) + 2 * m + 2 +
           (size_t)waves * (max_deg + (max_deg < 10 ? max_deg : 10)) * 37 + (size_t)(cam_cap + 1) / 2;
}

// (the profile's starting values, for a rank that does not write the camera blocks: k_schur_init does it otherwise)
__global__ void k_schur_first_init(int *__restrict__ first, int64_t nb, int64_t K) {
    const int64_t ib = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ib < nb) first[ib] = schur_first_of_block(ib, K);
}

// rows of camera jr have an entry in the columns of camera jc <= jr
__device__ __forceinline__ void note_block_first(int *__restrict__ first, int64_t jr, int64_t jc) {
    const int cb = (int)((6 * jc) / 64), rb0 = (int)((6 * jr) / 64), rb1 = (int)((6 * jr + 5) / 64);
    atomicMin(first + rb0, cb);
    if (rb1 != rb0) atomicMin(first + rb1, cb);
}

// REG_PAIRS: a point's pairs of observations are dealt out one per lane (lane l <-> pair (o, o'), o' <= o, fixed for
// the whole kernel) and the lane keeps the 6 x 6 block of its pair in registers from one point to the next; it goes
// to the LDS tile only when the pair's cameras change.  The points of a tile come in the order of their cameras,
// so most of them see the same cameras as the point before: 36 LDS atomics per pair become 36 per change of the camera
// set, and the 165 LDS reads of the entry-per-lane loop 36.  Points with more than 10 observations (more pairs than
// lanes) or with their observations out of the order of the slots take the entry-per-lane loop.
template <bool REG_PAIRS, int SCHUR_TILE_BLOCK>
__global__ void __launch_bounds__(SCHUR_TILE_BLOCK) k_schur_tile(SchurTileArgs a) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const TileMeta tm = a.meta[blockIdx.x];
    const int nc = tm.nc, np = tm.np, p0 = tm.p0;
    if (tm.e0 >= tm.e1 || (a.only && !a.only[blockIdx.x])) return;
    const int m = 6 * nc;
    double *rows = lds;                                   // cam_cap x CAM_ROW
    double *Tt = rows + (size_t)a.cam_cap * CAM_ROW;      // cam_cap x 6 x 3: the three non-zero entries of every column of the frame change
    double *St = Tt + (size_t)a.cam_cap * 18;             // lower block triangle: block (sr, sc <= sr) at (sr (sr + 1) / 2 + sc) * 36
    double *sphi_t = St + 18 * (size_t)a.cam_cap * (a.cam_cap + 1);   // m
    double *rhs_t = sphi_t + 6 * a.cam_cap;               // m
    double *sc_t = rhs_t + 6 * a.cam_cap;                 // {phi-phi, rhs phi}
    // per wave: W (deg x 18), Z (deg x 18) and the slots (deg) of one point of any degree, then the same for a second
    // point of at most ten observations (the layout of either: W | Z | slots with max_deg resp. deg2 rows)
    const int deg2 = a.max_deg < 10 ? a.max_deg : 10;
    const int scratch = a.max_deg * 37, scratch_wave = scratch + deg2 * 37;
    double *scratch0 = sc_t + 2 + (size_t)wid * scratch_wave;
    int *low = (int *)(sc_t + 2 + (size_t)(SCHUR_TILE_BLOCK / 64) * scratch_wave);     // per camera slot: the lowest slot it shares a point with
    for (int idx = tid; idx < a.cam_cap; idx += SCHUR_TILE_BLOCK) low[idx] = 0x7fffffff;
    for (int idx = tid; idx < nc * (CAM_ROW / 2); idx += SCHUR_TILE_BLOCK) {
        const int row = idx / (CAM_ROW / 2), piece = idx - row * (CAM_ROW / 2);
        const int j = tm.j0 >= 0 ? tm.j0 + row : a.cam_list[tm.cb + row];
        ((double2 *)rows)[idx] = ((const double2 *)(a.T + (int64_t)j * CAM_ROW))[piece];
    }
    // column al of T = blockdiag(R^T, A) (camera_frame_change) has three non-zero entries, in rows 0..2 (al < 3) or 3..5
    for (int idx = tid; idx < nc * 18; idx += SCHUR_TILE_BLOCK) {
        const int row = idx / 18, rest = idx - 18 * row, al = rest / 3, k = rest - 3 * al;
        const int j = tm.j0 >= 0 ? tm.j0 + row : a.cam_list[tm.cb + row];
        const double *src = a.T + (int64_t)j * CAM_ROW;
        double v;
        if (al < 3) v = src[3 * al + k];
        else if (al == 3) v = k == 0 ? src[12] : (k == 1 ? -src[13] : 0.0);
        else if (al == 4) v = src[3 + k];
        else v = k == 2 ? 1.0 : 0.0;
        Tt[idx] = v;
    }
    const int n_st = 18 * nc * (nc + 1);
    for (int idx = tid; idx < n_st; idx += SCHUR_TILE_BLOCK) St[idx] = 0.0;
    for (int idx = tid; idx < 2 * 6 * a.cam_cap + 2; idx += SCHUR_TILE_BLOCK) sphi_t[idx] = 0.0;
    const double f = a.scalars[SC_F], pcx = a.scalars[SC_CX], pcy = a.scalars[SC_CY], dphi = a.scalars[SC_DPHI];
    __syncthreads();

    // lane -> its pair (po, po2 <= po); block of St the lane's registers belong to (-1: none)
    int po = (int)((sqrtf(8.0f * (float)lane + 1.0f) - 1.0f) * 0.5f);
    while (po * (po + 1) / 2 > lane) --po;
    while ((po + 1) * (po + 2) / 2 <= lane) ++po;
    const int po2 = lane - po * (po + 1) / 2;
    int held = -1;
    double acc[36];
#pragma unroll
    for (int e = 0; e < 36; ++e) acc[e] = 0.0;
    auto flush_pairs = [&](bool mine) {      // lanes with `mine` add their block to the tile and let go of it
        if (mine) {
            double *dst = St + (size_t)held * 36;
#pragma unroll
            for (int e = 0; e < 36; ++e) {
                __hip_atomic_fetch_add(dst + e, acc[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                acc[e] = 0.0;
            }
            held = -1;
        }
    };
    // ---- first phase, one point per trip of the wave.  Six lanes per observation: lane (o, al) forms column al of the
    // observation's 3 x 6 block W = J_X^T J_cam and of Z = V'^-1 W (ten observations per trip).
    const int lo = lane / 6, al = lane - 6 * lo;
    const bool lane_on = lane < 60;
    auto phase1_single = [&](int64_t i, int64_t eb, int deg, double *Wl, double *Zl, int *sl) {
        double Vi[6];
        inv_sym3(a.V + 6 * i, a.lambda, Vi);
        const double x[3] = {a.X[3 * i], a.X[3 * i + 1], a.X[3 * i + 2]};
        // point x phi column w = dphi sum_obs J_X^T m (not kept by the J^T J pass: summed here, stored for the back-substitution)
        // what the lane keeps of its observation: the rows of J_X and the three entries of either row of G that meet
        // the non-zero entries of column al of T (rows 0..2 of T for the centre, rows 3..5 for the angles)
        double jx0[3] = {0.0, 0.0, 0.0}, jx1[3] = {0.0, 0.0, 0.0}, g0[3] = {0.0, 0.0, 0.0}, g1[3] = {0.0, 0.0, 0.0}, om0 = 0.0, om1 = 0.0;
        int slot = 0;
        const bool centre = al < 3;
        auto load_obs = [&](int o) {
            const int64_t e = eb + o;
            slot = (int)(a.opack[e] & 0xffffu);
            const ObsJac ob = obs_jacobian(x, (const double2 *)(rows + slot * CAM_ROW), f, pcx, pcy, a.ouv[e]);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                jx0[k] = ob.jx0[k], jx1[k] = ob.jx1[k];
                g0[k] = centre ? ob.G0[k] : ob.G0[3 + k];
                g1[k] = centre ? ob.G1[k] : ob.G1[3 + k];
            }
            om0 = ob.m0, om1 = ob.m1;
        };
        const double gx[3] = {a.gX[3 * i], a.gX[3 * i + 1], a.gX[3 * i + 2]};
        const double zg[3] = {Vi[0] * gx[0] + Vi[1] * gx[1] + Vi[2] * gx[2], Vi[1] * gx[0] + Vi[3] * gx[1] + Vi[4] * gx[2],
                              Vi[2] * gx[0] + Vi[4] * gx[1] + Vi[5] * gx[2]};
        // column al of W and Z of observation o into the wave's scratch, its share of the right-hand side
        auto form_w = [&](int o, double &W0, double &W1, double &W2) {
            const double *tc = Tt + (slot * 6 + al) * 3;        // column al of J_cam = G T
            const double t0 = tc[0], t1 = tc[1], t2 = tc[2];
            const double jc0 = g0[0] * t0 + g0[1] * t1 + g0[2] * t2;
            const double jc1 = g1[0] * t0 + g1[1] * t1 + g1[2] * t2;
            W0 = jx0[0] * jc0 + jx1[0] * jc1;
            W1 = jx0[1] * jc0 + jx1[1] * jc1;
            W2 = jx0[2] * jc0 + jx1[2] * jc1;
            Wl[o * 18 + 0 * 6 + al] = W0;
            Wl[o * 18 + 1 * 6 + al] = W1;
            Wl[o * 18 + 2 * 6 + al] = W2;
            Zl[o * 18 + 0 * 6 + al] = Vi[0] * W0 + Vi[1] * W1 + Vi[2] * W2;
            Zl[o * 18 + 1 * 6 + al] = Vi[1] * W0 + Vi[3] * W1 + Vi[4] * W2;
            Zl[o * 18 + 2 * 6 + al] = Vi[2] * W0 + Vi[4] * W1 + Vi[5] * W2;
            const double rr = W0 * zg[0] + W1 * zg[1] + W2 * zg[2];
            __hip_atomic_fetch_add(rhs_t + 6 * slot + al, rr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (al == 0) sl[o] = slot;
        };
        double wp[3] = {0.0, 0.0, 0.0}, zw[3];
        auto finish_wp = [&]() {
#pragma unroll
            for (int k = 0; k < 3; ++k) wp[k] = dphi * wave_sum(wp[k]);
            if (lane < 3) a.Xphi[3 * i + lane] = lane == 0 ? wp[0] : (lane == 1 ? wp[1] : wp[2]);
            zw[0] = Vi[0] * wp[0] + Vi[1] * wp[1] + Vi[2] * wp[2];
            zw[1] = Vi[1] * wp[0] + Vi[3] * wp[1] + Vi[4] * wp[2];
            zw[2] = Vi[2] * wp[0] + Vi[4] * wp[1] + Vi[5] * wp[2];
        };
        auto add_sphi = [&](double W0, double W1, double W2) {
            const double sphi = W0 * zw[0] + W1 * zw[1] + W2 * zw[2];
            __hip_atomic_fetch_add(sphi_t + 6 * slot + al, -sphi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        if (deg <= 10) {
            // one trip: W and Z do not depend on the sums over the wave, so they are formed (and written) ahead of them
            // and the sums' latency passes behind that work; only the phi column's entry waits for them
            const bool on = lane_on && lo < deg;
            double W0 = 0.0, W1 = 0.0, W2 = 0.0;
            if (on) {
                load_obs(lo);
                if (al == 0) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) wp[k] = jx0[k] * om0 + jx1[k] * om1;
                }
                form_w(lo, W0, W1, W2);
            }
            finish_wp();
            if (on) add_sphi(W0, W1, W2);
        } else {
            for (int base = 0; base < deg; base += 10) {
                const int o = base + lo;
                if (lane_on && o < deg) {
                    load_obs(o);
                    if (al == 0) {
#pragma unroll
                        for (int k = 0; k < 3; ++k) wp[k] += jx0[k] * om0 + jx1[k] * om1;
                    }
                }
            }
            finish_wp();
            for (int base = 0; base < deg; base += 10) {
                const int o = base + lo;
                if (lane_on && o < deg) {
                    load_obs(o);
                    double W0, W1, W2;
                    form_w(o, W0, W1, W2);
                    add_sphi(W0, W1, W2);
                }
            }
        }
        if (lane == 0) {
            __hip_atomic_fetch_add(sc_t + 0, -(wp[0] * zw[0] + wp[1] * zw[1] + wp[2] * zw[2]), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(sc_t + 1, wp[0] * zg[0] + wp[1] * zg[1] + wp[2] * zg[2], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    // ---- first phase, two points per trip (neither with more than ten observations): the chain of dependent steps a
    // point costs - its data, the Jacobians, the sums over its observations, W and Z - is what the kernel waits for
    // (two waves per SIMD, the pair blocks take the registers), so one pass through it serves points pt and pt + 1
    // in the two halves of the wave: three lanes per observation, each forming two columns.
    const int half = lane >> 5, l32 = lane & 31, lo3 = l32 / 3, ac = l32 - 3 * lo3;
    auto phase1_pair = [&](int64_t iA, bool hasB, int64_t ebA, int degA, int64_t ebB, int degB) {
        const int64_t i = iA + half;
        const int64_t eb = half ? ebB : ebA;
        const int deg = half ? degB : degA;
        const bool exists = half == 0 || hasB;
        const int n_rows = half ? deg2 : a.max_deg;
        double *Wl = scratch0 + half * scratch, *Zl = Wl + (size_t)n_rows * 18;
        int *sl = (int *)(Zl + (size_t)n_rows * 18);
        double V6[6] = {1.0, 0.0, 0.0, 1.0, 0.0, 1.0}, x[3] = {0.0, 0.0, 0.0}, gx[3] = {0.0, 0.0, 0.0};
        if (deg > 0) {
#pragma unroll
            for (int k = 0; k < 6; ++k) V6[k] = a.V[6 * i + k];
#pragma unroll
            for (int k = 0; k < 3; ++k) x[k] = a.X[3 * i + k], gx[k] = a.gX[3 * i + k];
        }
        double Vi[6];
        inv_sym3(V6, a.lambda, Vi);
        const double zg[3] = {Vi[0] * gx[0] + Vi[1] * gx[1] + Vi[2] * gx[2], Vi[1] * gx[0] + Vi[3] * gx[1] + Vi[4] * gx[2],
                              Vi[2] * gx[0] + Vi[4] * gx[1] + Vi[5] * gx[2]};
        const bool on = l32 < 30 && lo3 < deg;
        double wp[3] = {0.0, 0.0, 0.0}, W[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
        int slot = 0;
        if (on) {
            const int64_t e = eb + lo3;
            slot = (int)(a.opack[e] & 0xffffu);
            const ObsJac ob = obs_jacobian(x, (const double2 *)(rows + slot * CAM_ROW), f, pcx, pcy, a.ouv[e]);
            if (ac == 0) {
#pragma unroll
                for (int k = 0; k < 3; ++k) wp[k] = ob.jx0[k] * ob.m0 + ob.jx1[k] * ob.m1;
            }
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const int c = 2 * ac + cc;
                const bool centre = c < 3;
                const double *tc = Tt + (slot * 6 + c) * 3;        // column c of J_cam = G T
                const double t0 = tc[0], t1 = tc[1], t2 = tc[2];
                const double jc0 = (centre ? ob.G0[0] : ob.G0[3]) * t0 + (centre ? ob.G0[1] : ob.G0[4]) * t1 +
                                   (centre ? ob.G0[2] : ob.G0[5]) * t2;
                const double jc1 = (centre ? ob.G1[0] : ob.G1[3]) * t0 + (centre ? ob.G1[1] : ob.G1[4]) * t1 +
                                   (centre ? ob.G1[2] : ob.G1[5]) * t2;
#pragma unroll
                for (int k = 0; k < 3; ++k) W[cc][k] = ob.jx0[k] * jc0 + ob.jx1[k] * jc1;
                Wl[lo3 * 18 + 0 * 6 + c] = W[cc][0];
                Wl[lo3 * 18 + 1 * 6 + c] = W[cc][1];
                Wl[lo3 * 18 + 2 * 6 + c] = W[cc][2];
                Zl[lo3 * 18 + 0 * 6 + c] = Vi[0] * W[cc][0] + Vi[1] * W[cc][1] + Vi[2] * W[cc][2];
                Zl[lo3 * 18 + 1 * 6 + c] = Vi[1] * W[cc][0] + Vi[3] * W[cc][1] + Vi[4] * W[cc][2];
                Zl[lo3 * 18 + 2 * 6 + c] = Vi[2] * W[cc][0] + Vi[4] * W[cc][1] + Vi[5] * W[cc][2];
                const double rr = W[cc][0] * zg[0] + W[cc][1] * zg[1] + W[cc][2] * zg[2];
                __hip_atomic_fetch_add(rhs_t + 6 * slot + c, rr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (ac == 0) sl[lo3] = slot;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) wp[k] = dphi * half_wave_sum(wp[k], half);
        if (exists && l32 < 3) a.Xphi[3 * i + l32] = l32 == 0 ? wp[0] : (l32 == 1 ? wp[1] : wp[2]);
        const double zw[3] = {Vi[0] * wp[0] + Vi[1] * wp[1] + Vi[2] * wp[2], Vi[1] * wp[0] + Vi[3] * wp[1] + Vi[4] * wp[2],
                              Vi[2] * wp[0] + Vi[4] * wp[1] + Vi[5] * wp[2]};
        if (on) {
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const double sphi = W[cc][0] * zw[0] + W[cc][1] * zw[1] + W[cc][2] * zw[2];
                __hip_atomic_fetch_add(sphi_t + 6 * slot + 2 * ac + cc, -sphi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if (l32 == 0 && deg > 0) {
            __hip_atomic_fetch_add(sc_t + 0, -(wp[0] * zw[0] + wp[1] * zw[1] + wp[2] * zw[2]), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(sc_t + 1, wp[0] * zg[0] + wp[1] * zg[1] + wp[2] * zg[2], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    // ---- second phase: the pairs of one point's observations
    auto pair_phase = [&](int deg, const double *Wl, const double *Zl, const int *sl) {
        // all ordered pairs (o, o') of the point's observations with slot >= slot': the pair loop is
        // wave-uniform, lane (al, be) < 36 owns one entry of the 6 x 6 block (no index arithmetic
        // with divisions inside the loop)
        // a point's observations usually come in the order of the camera slots: the pairs with slot >= slot' are then
        // the pairs o' <= o, and the inner loop needs no test (half its trips would fail it)
        // (up to ten observations: the lanes' own pairs (po, po2 <= po) cover every neighbouring pair, and the slots
        // they read for themselves answer the question; the lowest slot is then the first observation's)
        bool ascending = true;
        const bool pair_on = po < deg;
        int so = 0, so2 = 0;
        if (deg <= 10) {
            if (pair_on) so = sl[po], so2 = sl[po2];
            ascending = !pair_on || so2 <= so;
        } else {
            for (int o = lane; o + 1 < deg; o += 64) ascending = ascending && sl[o] <= sl[o + 1];
        }
        ascending = __all(ascending);
        if (a.first) {
            if (deg <= 10 && ascending) {
                const int lowest = __builtin_amdgcn_readfirstlane(so2);      // lane 0 holds the pair (0, 0)
                if (pair_on && po2 == 0) atomicMin(low + so, lowest);
            } else {
                int lowest = 0x7fffffff;
                for (int o = lane; o < deg; o += 64) lowest = min(lowest, sl[o]);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) lowest = min(lowest, __shfl_xor(lowest, off));
                for (int o = lane; o < deg; o += 64) atomicMin(low + sl[o], lowest);
            }
        }
        if (REG_PAIRS && deg <= 10 && ascending) {
            const bool active = pair_on;
            int blk = -1;
            if (active) blk = so * (so + 1) / 2 + so2;
            const bool moved = active && held >= 0 && blk != held;
            if (__any(moved)) flush_pairs(moved);
            if (active) {
                held = blk;
                const double *wr = Wl + po * 18, *zr = Zl + po2 * 18;
#pragma unroll
                for (int mm = 0; mm < 3; ++mm) {
                    double w[6], z[6];
#pragma unroll
                    for (int h = 0; h < 3; ++h) {
                        const double2 wv = *(const double2 *)(wr + 6 * mm + 2 * h), zv = *(const double2 *)(zr + 6 * mm + 2 * h);
                        w[2 * h] = wv.x, w[2 * h + 1] = wv.y;
                        z[2 * h] = zv.x, z[2 * h + 1] = zv.y;
                    }
#pragma unroll
                    for (int al = 0; al < 6; ++al)
#pragma unroll
                        for (int be = 0; be < 6; ++be) acc[6 * al + be] = fma(-w[al], z[be], acc[6 * al + be]);
                }
            }
        } else {
        if (REG_PAIRS && __any(held >= 0)) flush_pairs(held >= 0);
        if (lane < 36) {
            const int al = lane / 6, be = lane - 6 * al;
            for (int o = 0; o < deg; ++o) {
                const int so = sl[o];
                const double w0 = Wl[o * 18 + al], w1 = Wl[o * 18 + 6 + al], w2 = Wl[o * 18 + 12 + al];
                double *dst = St + (size_t)(so * (so + 1) / 2) * 36 + lane;
                const int o2_end = ascending ? o + 1 : deg;
                for (int o2 = 0; o2 < o2_end; ++o2) {
                    const int so2 = sl[o2];
                    if (so < so2) continue;
                    const double v = w0 * Zl[o2 * 18 + be] + w1 * Zl[o2 * 18 + 6 + be] + w2 * Zl[o2 * 18 + 12 + be];
                    __hip_atomic_fetch_add(dst + 36 * so2, -v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        }
    };
    auto lanes_sync = [&]() {
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
    };
    constexpr int PT_STEP = SCHUR_TILE_BLOCK / 64;
    for (int pt = 2 * wid; pt < np; pt += 2 * PT_STEP) {
        const int64_t iA = p0 + pt;
        const bool hasB = pt + 1 < np;
        const int64_t ebA = a.row_ptr[iA], ebB = a.row_ptr[iA + 1];
        const int degA = (int)(ebB - ebA), degB = hasB ? (int)(a.row_ptr[iA + 2] - ebB) : 0;
        double *W0 = scratch0, *Z0 = W0 + (size_t)a.max_deg * 18, *W1 = scratch0 + scratch, *Z1 = W1 + (size_t)deg2 * 18;
        int *s0 = (int *)(Z0 + (size_t)a.max_deg * 18), *s1 = (int *)(Z1 + (size_t)deg2 * 18);
        if (degA <= 10 && degB <= 10) {
            phase1_pair(iA, hasB, ebA, degA, ebB, degB);
            lanes_sync();
            if (degA > 0) pair_phase(degA, W0, Z0, s0);
            if (degB > 0) pair_phase(degB, W1, Z1, s1);
            lanes_sync();
        } else {
            for (int q = 0; q < (hasB ? 2 : 1); ++q) {
                const int64_t i = iA + q, eb = q ? ebB : ebA;
                const int deg = q ? degB : degA;
                if (deg == 0) {
                    if (lane < 3) a.Xphi[3 * i + lane] = 0.0;
                    continue;
                }
                phase1_single(i, eb, deg, W0, Z0, s0);
                lanes_sync();
                pair_phase(deg, W0, Z0, s0);
                lanes_sync();
            }
        }
    }
    if (REG_PAIRS) flush_pairs(held >= 0);
    __syncthreads();
    const int64_t nphi = 6 * a.K;
    if (a.first) {      // (the tile's camera slots are in the order of the camera numbers)
        for (int sr = tid; sr < nc; sr += SCHUR_TILE_BLOCK) {
            if (low[sr] > sr) continue;
            const int64_t jr = tm.j0 >= 0 ? tm.j0 + sr : a.cam_list[tm.cb + sr];
            const int64_t jc = tm.j0 >= 0 ? tm.j0 + low[sr] : a.cam_list[tm.cb + low[sr]];
            note_block_first(a.first, jr, jc);
        }
    }
    for (int idx = tid; idx < n_st; idx += SCHUR_TILE_BLOCK) {
        const double v = St[idx];
        if (v == 0.0) continue;
        const int blk = idx / 36, ent = idx - 36 * blk;
        int sr = (int)((sqrt(8.0 * (double)blk + 1.0) - 1.0) * 0.5);
        while (sr * (sr + 1) / 2 > blk) --sr;
        while ((sr + 1) * (sr + 2) / 2 <= blk) ++sr;
        const int scn = blk - sr * (sr + 1) / 2;
        if (sr == scn && ent / 6 < ent % 6) continue;       // a camera's own block: lower triangle only
        const int64_t jr = tm.j0 >= 0 ? tm.j0 + sr : a.cam_list[tm.cb + sr];
        const int64_t jc = tm.j0 >= 0 ? tm.j0 + scn : a.cam_list[tm.cb + scn];
        __hip_atomic_fetch_add(a.S + (6 * jr + ent / 6) * a.ld + 6 * jc + ent % 6, v, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int idx = tid; idx < m; idx += SCHUR_TILE_BLOCK) {
        const int sr = idx / 6;
        const int64_t jr = tm.j0 >= 0 ? tm.j0 + sr : a.cam_list[tm.cb + sr];
        const int64_t ra = 6 * jr + (idx - 6 * sr);
        __hip_atomic_fetch_add(a.S + nphi * a.ld + ra, sphi_t[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(a.rhs + ra, rhs_t[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0) {
        __hip_atomic_fetch_add(a.S + nphi * a.ld + nphi, sc_t[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(a.rhs + nphi, sc_t[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// --------------------------------------------------------------------------------------------
// The same on the matrix cores.  With V'^-1 = L^-T L^-1 (V' = L L^T, 3 x 3) a point's contribution is a Gram matrix:
// let Yt = L^-1 [ W(slot 0) | ... | W(slot nc-1) | w_phi | g ]  (3 x (6 nc + 2); W = 0 for the cameras that do not see
// the point), then  Yt^T Yt  holds  W^T V'^-1 W  (subtracted from the camera blocks),  W^T V'^-1 w_phi  (from the row of
// phi_x),  w_phi^T V'^-1 w_phi,  and in its last row  W^T V'^-1 g,  w_phi^T V'^-1 g  (added to the right-hand side).
// Summed over the points of a tile that is  A^T A  with the 3 np rows of all points stacked: a product of a
// (6 nc + 2) x 3 np matrix with its transpose, nine tenths of it non-zero in the tracks of a video.
//
// A wave takes four points at a time: one lane per observation forms the Jacobians and the six columns of L^-1 W
// (instead of six lanes per observation, each with the observation's Jacobians), the rows go to the wave's own
// 12 x Mp patch of LDS, and three steps of v_mfma_f64_4x4x4_4b_f64 (lane layout: see k_syrk_trailing in ba_chol.hip)
// add the patch's Gram matrix to the lower 16 x 16 tiles the wave keeps in registers for the whole tile of points:
// per step NBT column-operand registers and 4 NBT row-operand registers from LDS for 2 NBT (NBT + 1) instructions.
// No barrier between waves until the end, where the waves' sums meet in LDS and leave for S as in k_schur_tile.
// A workgroup's share is not a whole tile but a range of its points whose cameras lie within a window of cam_cap
// slots (ensure_schur_subtiles): NBT = ceil((6 cam_cap + 2) / 16) tile rows of accumulators must fit the registers
// at two waves per SIMD (NBT = 5, a window of 13 cameras: 214 registers; a sixth row spills, a seventh needs the
// whole register file of a SIMD for one wave and was slower).
// --------------------------------------------------------------------------------------------
constexpr int SCHUR_GEMM_BLOCK = 256;
// Points per trip of a wave: four (sixteen lanes each, 12 rows = three steps of four) or, where no point has more than
// ten observations, six (ten lanes each, 18 rows in five steps): the lanes that form the rows are then 60 of 64 busy
// instead of 40, for one matrix-core step in six spent on two rows of zeros.
constexpr int SCHUR_GEMM_ROWS_MAX = 20;

__host__ __device__ inline int schur_gemm_nbt(int cam_cap) { return (6 * cam_cap + 2 + 15) / 16; }
__host__ __device__ inline size_t schur_gemm_lds_doubles(int cam_cap) {
    const int nbt = schur_gemm_nbt(cam_cap) < 4 ? 4 : schur_gemm_nbt(cam_cap), mp = 16 * nbt;      // (the smallest kernel has NBT = 4)
    // (the waves' patches and, after the last trip, the tiles their sums meet in share one region)
    return (size_t)cam_cap * (CAM_ROW + 18) +
           std::max((size_t)(SCHUR_GEMM_BLOCK / 64) * (SCHUR_GEMM_ROWS_MAX * (mp + 1)), (size_t)nbt * (nbt + 1) / 2 * 256) +
           (size_t)(cam_cap + 1) / 2;
}

template <int NBT, int PTS>
__global__ void __launch_bounds__(SCHUR_GEMM_BLOCK, NBT <= 4 ? 3 : 2) k_schur_gemm(SchurTileArgs a) {
    extern __shared__ double lds[];
    constexpr int LPP = PTS == 4 ? 16 : 10;             // lanes per point
    constexpr int MP = 16 * NBT, NT = NBT * (NBT + 1) / 2, ROWS = (3 * PTS + 3) / 4 * 4, PITCH = MP;
    static_assert(PTS * LPP <= 64 && ROWS <= SCHUR_GEMM_ROWS_MAX, "points of a trip: one lane group each");
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // a range of points of a tile and the window of the tile's camera slots they see
    const int4 sub = a.sub[blockIdx.x];
    const TileMeta tm = a.meta[sub.x];
    const int slot_lo = sub.w & 0xffff, nc = sub.w >> 16, np = sub.z, p0 = tm.p0 + sub.y;
    auto camera_of = [&](int s) -> int64_t { return tm.j0 >= 0 ? tm.j0 + slot_lo + s : a.cam_list[tm.cb + slot_lo + s]; };
    const int m = 6 * nc;       // columns: 6 per camera slot of the window, then phi_x (m), then g (m + 1)
    double *rows = lds;                                   // cam_cap x CAM_ROW
    double *Tt = rows + (size_t)a.cam_cap * CAM_ROW;      // cam_cap x 6 x 3 (see k_schur_tile)
    double *patch = Tt + (size_t)a.cam_cap * 18 + (size_t)wid * (SCHUR_GEMM_ROWS_MAX * (PITCH + 1));   // ROWS x PITCH, then ROWS sums for w_phi
    double *wsum = patch + ROWS * PITCH;
    double *Gt = Tt + (size_t)a.cam_cap * 18;       // NT tiles of 16 x 16, in place of the patches once the trips are over
    constexpr size_t SHARED = (size_t)(SCHUR_GEMM_BLOCK / 64) * (SCHUR_GEMM_ROWS_MAX * (PITCH + 1)) > (size_t)NT * 256
                                  ? (size_t)(SCHUR_GEMM_BLOCK / 64) * (SCHUR_GEMM_ROWS_MAX * (PITCH + 1))
                                  : (size_t)NT * 256;
    int *low = (int *)(Gt + SHARED);
    for (int idx = tid; idx < a.cam_cap; idx += SCHUR_GEMM_BLOCK) low[idx] = 0x7fffffff;
    for (int idx = tid; idx < nc * (CAM_ROW / 2); idx += SCHUR_GEMM_BLOCK) {
        const int row = idx / (CAM_ROW / 2), piece = idx - row * (CAM_ROW / 2);
        ((double2 *)rows)[idx] = ((const double2 *)(a.T + camera_of(row) * CAM_ROW))[piece];
    }
    for (int idx = tid; idx < nc * 18; idx += SCHUR_GEMM_BLOCK) {
        const int row = idx / 18, rest = idx - 18 * row, al = rest / 3, k = rest - 3 * al;
        const double *src = a.T + camera_of(row) * CAM_ROW;
        double v;
        if (al < 3) v = src[3 * al + k];
        else if (al == 3) v = k == 0 ? src[12] : (k == 1 ? -src[13] : 0.0);
        else if (al == 4) v = src[3 + k];
        else v = k == 2 ? 1.0 : 0.0;
        Tt[idx] = v;
    }
    const double f = a.scalars[SC_F], pcx = a.scalars[SC_CX], pcy = a.scalars[SC_CY], dphi = a.scalars[SC_DPHI];
    __syncthreads();

    double acc[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int I = 0; I < 4; ++I) acc[t][I] = 0.0;
    // L^-1 of the damped point block (V6: upper triangle of V, row-major)
    auto inverse_factor = [&](const double *V6, double *li /* i00 i10 i11 i20 i21 i22 */) {
        const double va = V6[0] * (1.0 + a.lambda), vb = V6[1], vc = V6[2], vd = V6[3] * (1.0 + a.lambda), ve = V6[4],
                     vg = V6[5] * (1.0 + a.lambda);
        const double i00 = fast_rsqrt(va);
        const double l10 = vb * i00, l20 = vc * i00;
        const double i11 = fast_rsqrt(vd - l10 * l10);
        const double l21 = (ve - l20 * l10) * i11;
        const double i22 = fast_rsqrt(vg - l20 * l20 - l21 * l21);
        li[0] = i00;
        li[1] = -l10 * i00 * i11;
        li[2] = i11;
        li[4] = -l21 * i11 * i22;
        li[3] = -(li[4] * l10 + i22 * l20) * i00;
        li[5] = i22;
    };
    // The lanes of a wave hand data to each other through the wave's patch of LDS.  The LDS serves a wave's
    // instructions in order, so all that is needed is that the compiler keeps them in order too - a fence would also
    // wait for the loads of the next trip that are in flight on purpose.
    auto lanes_sync = [&]() {
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
    };
    const int nbt_tile = (m + 2 + 15) / 16;     // tile rows of this tile of points
    // LPP lanes per point, lane o16 of them for the point's observation o16 (more than LPP: further trips).
    // A trip starts with two dependent trips to memory (observation ranges, then the points' and observations' data)
    // that only the SIMD's other waves cover: requesting them a trip ahead costs registers the accumulators need
    // (the ranges alone: 14 spilled registers at three waves per SIMD, LM iteration 2.07 -> 2.13 ms).
    constexpr int GSTEP = PTS * (SCHUR_GEMM_BLOCK / 64);
    const int pl = lane / LPP, o16 = lane - LPP * pl;       // (pl == PTS: a lane without a point)
    struct GroupIn {
        uint32_t pk;
        double2 uv;
        double V6[6], x[3], g[3];
    };
    auto range_of = [&](int first_pt, int64_t &eb, int &deg) {
        eb = 0, deg = 0;
        if (pl < PTS && first_pt + pl < np) {
            eb = a.row_ptr[p0 + first_pt + pl];
            deg = (int)(a.row_ptr[p0 + first_pt + pl + 1] - eb);
        }
    };
    auto fetch = [&](int first_pt, int64_t eb, int deg, GroupIn &q) {
        q.pk = 0;
        q.uv = make_double2(0.0, 0.0);
#pragma unroll
        for (int k = 0; k < 6; ++k) q.V6[k] = k == 0 || k == 3 || k == 5 ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) q.x[k] = q.g[k] = 0.0;
        if (deg > 0) {
            const int64_t i = (int64_t)p0 + first_pt + pl;
#pragma unroll
            for (int k = 0; k < 6; ++k) q.V6[k] = a.V[6 * i + k];
#pragma unroll
            for (int k = 0; k < 3; ++k) q.x[k] = a.X[3 * i + k];
            if (o16 == 0) {
#pragma unroll
                for (int k = 0; k < 3; ++k) q.g[k] = a.gX[3 * i + k];
            }
            if (o16 < deg) {
                q.pk = a.opack[eb + o16];
                q.uv = a.ouv[eb + o16];
            }
        }
    };
    for (int pt0 = PTS * wid; pt0 < np; pt0 += GSTEP) {
        int64_t eb_c;
        int deg_c;
        GroupIn cur;
        range_of(pt0, eb_c, deg_c);
        fetch(pt0, eb_c, deg_c, cur);
        for (int idx = lane; idx < ROWS * (PITCH + 1); idx += 64) patch[idx] = 0.0;
        lanes_sync();
        double li[6];
        inverse_factor(cur.V6, li);
        const int64_t i = (int64_t)p0 + pt0 + pl;
        auto one_observation = [&](uint32_t pk, double2 uv) {
            const int slot = (int)(pk & 0xffffu) - slot_lo;
            const ObsJac ob = obs_jacobian(cur.x, (const double2 *)(rows + slot * CAM_ROW), f, pcx, pcy, uv);
#pragma unroll
            for (int k = 0; k < 3; ++k)
                __hip_atomic_fetch_add(wsum + 3 * pl + k, dphi * (ob.jx0[k] * ob.m0 + ob.jx1[k] * ob.m1), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
            double *dst = patch + (3 * pl) * PITCH + 6 * slot;
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const double *tc = Tt + (slot * 6 + c) * 3;         // column c of J_cam = G T
                const double jc0 = ob.G0[c < 3 ? 0 : 3] * tc[0] + ob.G0[c < 3 ? 1 : 4] * tc[1] + ob.G0[c < 3 ? 2 : 5] * tc[2];
                const double jc1 = ob.G1[c < 3 ? 0 : 3] * tc[0] + ob.G1[c < 3 ? 1 : 4] * tc[1] + ob.G1[c < 3 ? 2 : 5] * tc[2];
                const double W0 = ob.jx0[0] * jc0 + ob.jx1[0] * jc1, W1 = ob.jx0[1] * jc0 + ob.jx1[1] * jc1,
                             W2 = ob.jx0[2] * jc0 + ob.jx1[2] * jc1;
                // (added, not stored: a point seen twice by one camera contributes the sum of both blocks)
                __hip_atomic_fetch_add(dst + c, li[0] * W0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(dst + PITCH + c, li[1] * W0 + li[2] * W1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(dst + 2 * PITCH + c, li[3] * W0 + li[4] * W1 + li[5] * W2, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        };
        if (o16 < deg_c) one_observation(cur.pk, cur.uv);
        for (int o = o16 + LPP; o < deg_c; o += LPP) one_observation(a.opack[eb_c + o], a.ouv[eb_c + o]);
        lanes_sync();
        if (o16 == 0 && pl < PTS && pt0 + pl < np) {       // the point's own two columns: phi_x and g
            const bool seen = deg_c > 0;
            const double wp[3] = {wsum[3 * pl], wsum[3 * pl + 1], wsum[3 * pl + 2]};
#pragma unroll
            for (int k = 0; k < 3; ++k) a.Xphi[3 * i + k] = seen ? wp[k] : 0.0;
            if (seen) {
                double *dst = patch + (3 * pl) * PITCH + m;
                dst[0] = li[0] * wp[0];
                dst[PITCH] = li[1] * wp[0] + li[2] * wp[1];
                dst[2 * PITCH] = li[3] * wp[0] + li[4] * wp[1] + li[5] * wp[2];
                dst[1] = li[0] * cur.g[0];
                dst[PITCH + 1] = li[1] * cur.g[0] + li[2] * cur.g[1];
                dst[2 * PITCH + 1] = li[3] * cur.g[0] + li[4] * cur.g[1] + li[5] * cur.g[2];
            }
        }
        lanes_sync();
#pragma unroll
        for (int ks = 0; ks < ROWS / 4; ++ks) {
            const double *src = patch + (4 * ks + (lane >> 4)) * PITCH;
            double bop[NBT], aop[NBT][4];
#pragma unroll
            for (int t = 0; t < NBT; ++t) {
                bop[t] = src[16 * t + (lane & 15)];
#pragma unroll
                for (int I = 0; I < 4; ++I) aop[t][I] = src[16 * t + 4 * I + (lane & 3)];
            }
#pragma unroll
            for (int tr = 0; tr < NBT; ++tr) {
                if (tr < nbt_tile) {
#pragma unroll
                    for (int I = 0; I < 4; ++I) {
                        if (16 * tr + 4 * I >= m + 2) continue;      // rows of the Gram matrix beyond the tile's last column
#pragma unroll
                        for (int tc = 0; tc <= tr; ++tc)
                            acc[tr * (tr + 1) / 2 + tc][I] =
                                __builtin_amdgcn_mfma_f64_4x4x4f64(aop[tr][I], bop[tc], acc[tr * (tr + 1) / 2 + tc][I], 0, 0, 0);
                    }
                }
            }
        }
        lanes_sync();
    }
    // the waves' sums meet in LDS: entry (16 tr + 4 I + (lane >> 4), 16 tc + (lane & 15)) of the Gram matrix
    __syncthreads();        // every wave is through with its patch
    for (int idx = tid; idx < NT * 256; idx += SCHUR_GEMM_BLOCK) Gt[idx] = 0.0;
    __syncthreads();
#pragma unroll
    for (int tr = 0; tr < NBT; ++tr)
#pragma unroll
        for (int tc = 0; tc <= tr; ++tc)
#pragma unroll
            for (int I = 0; I < 4; ++I)
                __hip_atomic_fetch_add(Gt + (tr * (tr + 1) / 2 + tc) * 256 + (4 * I + (lane >> 4)) * 16 + (lane & 15),
                                       acc[tr * (tr + 1) / 2 + tc][I], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    const int64_t nphi = 6 * a.K;
    if (a.first) {      // (the tile's camera slots are in the order of the camera numbers)
        // two cameras share a point exactly when their block of the Gram matrix is not zero
        for (int idx = tid; idx < NT * 256; idx += SCHUR_GEMM_BLOCK) {
            if (Gt[idx] == 0.0) continue;
            const int t = idx >> 8, within = idx & 255;
            int tr = 0;
            while ((tr + 1) * (tr + 2) / 2 <= t) ++tr;
            const int r = 16 * tr + (within >> 4), c = 16 * (t - tr * (tr + 1) / 2) + (within & 15);
            if (c <= r && r < m) atomicMin(low + r / 6, c / 6);
        }
        __syncthreads();
        for (int sr = tid; sr < nc; sr += SCHUR_GEMM_BLOCK) {
            if (low[sr] > sr) continue;
            note_block_first(a.first, camera_of(sr), camera_of(low[sr]));
        }
    }
    // global row of local row / column r <= m: a camera's unknown or phi_x
    auto global_index = [&](int r) -> int64_t {
        if (r == m) return nphi;
        const int sr = r / 6;
        return 6 * camera_of(sr) + (r - 6 * sr);
    };
    for (int idx = tid; idx < NT * 256; idx += SCHUR_GEMM_BLOCK) {
        const double v = Gt[idx];
        if (v == 0.0) continue;
        const int t = idx >> 8, within = idx & 255;
        int tr = 0;
        while ((tr + 1) * (tr + 2) / 2 <= t) ++tr;
        const int tc = t - tr * (tr + 1) / 2;
        const int r = 16 * tr + (within >> 4), c = 16 * tc + (within & 15);
        if (c > r || r > m + 1 || c > m) continue;
        if (r == m + 1) {       // the row of g: right-hand side
            __hip_atomic_fetch_add(a.rhs + global_index(c), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_fetch_add(a.S + global_index(r) * a.ld + global_index(c), -v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// One wave per point.  LDS per wave: W (cap x 18) and Z = V'^-1 W (cap x 18), camera indices (2 x cap), with
// cap = a.max_deg rows.  A point with more observations than that (a keypoint tracked through hundreds of frames)
// goes through in chunks of cap: the W rows of chunk A against the Z rows of chunk B for every pair of chunks, the
// rows formed again for every pair (only such points pay for it).
__global__ void __launch_bounds__(256) k_schur_points(SchurArgs a) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + wid;
    if (i >= a.N) return;
    const int64_t e0 = a.row_ptr[i], e1 = a.row_ptr[i + 1];
    const int deg = (int)(e1 - e0);
    if (deg == 0) {
        if (lane < 3) a.Xphi[3 * i + lane] = 0.0;
        return;
    }
    const int cap = a.max_deg;
    double *Wl = lds + (size_t)wid * cap * 37;
    double *Zl = Wl + (size_t)cap * 18;
    int *jA = (int *)(Zl + (size_t)cap * 18), *jB = jA + cap;
    const double f = a.scalars[SC_F], pcx = a.scalars[SC_CX], pcy = a.scalars[SC_CY], dphi = a.scalars[SC_DPHI];
    double Vi[6];
    inv_sym3(a.V + 6 * i, a.lambda, Vi);
    const double x[3] = {a.X[3 * i], a.X[3 * i + 1], a.X[3 * i + 2]};
    // point x phi column w = dphi sum_obs J_X^T m (summed here, stored for the back-substitution)
    double wp[3] = {0.0, 0.0, 0.0};
    int lowest = 0x7fffffff;
    for (int o = lane; o < deg; o += 64) {
        const int64_t e = e0 + o;
        const int j = a.oj[e];
        lowest = min(lowest, j);
        const ObsJac ob = obs_jacobian(x, (const double2 *)(a.T + (int64_t)j * CAM_ROW), f, pcx, pcy, a.ouv[e]);
#pragma unroll
        for (int k = 0; k < 3; ++k) wp[k] += ob.jx0[k] * ob.m0 + ob.jx1[k] * ob.m1;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) wp[k] = dphi * wave_sum(wp[k]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lowest = min(lowest, __shfl_xor(lowest, off));
    if (lane < 3) a.Xphi[3 * i + lane] = lane == 0 ? wp[0] : (lane == 1 ? wp[1] : wp[2]);
    const double gx[3] = {a.gX[3 * i], a.gX[3 * i + 1], a.gX[3 * i + 2]};
    // V'^-1 wphi and V'^-1 gX
    const double zw[3] = {Vi[0] * wp[0] + Vi[1] * wp[1] + Vi[2] * wp[2], Vi[1] * wp[0] + Vi[3] * wp[1] + Vi[4] * wp[2],
                          Vi[2] * wp[0] + Vi[4] * wp[1] + Vi[5] * wp[2]};
    const double zg[3] = {Vi[0] * gx[0] + Vi[1] * gx[1] + Vi[2] * gx[2], Vi[1] * gx[0] + Vi[3] * gx[1] + Vi[4] * gx[2],
                          Vi[2] * gx[0] + Vi[4] * gx[1] + Vi[5] * gx[2]};
    const int64_t nphi = 6 * a.K;
    if (lane == 0) {
        const double spp = wp[0] * zw[0] + wp[1] * zw[1] + wp[2] * zw[2];
        const double rp = wp[0] * zg[0] + wp[1] * zg[1] + wp[2] * zg[2];
        __hip_atomic_fetch_add(a.S + nphi * a.ld + nphi, -spp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(a.rhs + nphi, rp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // rows of the observations c0 .. c0 + cn - 1: W into Wd and / or Z into Zd (either may be null), camera indices
    // into jd; emit: these observations' part of the camera x phi column, of the right-hand side and of the profile
    auto rows = [&](int c0, int cn, double *Wd, double *Zd, int *jd, bool emit) {
        for (int o = lane; o < cn; o += 64) {
            const int64_t e = e0 + c0 + o;
            const int j = a.oj[e];
            const double *row = a.T + (int64_t)j * CAM_ROW;
            ObsJac ob = obs_jacobian(x, (const double2 *)row, f, pcx, pcy, a.ouv[e]);
            double Tm[6][6];
            camera_frame_change(row, Tm);
            double jc0[6], jc1[6];  // rows of J_cam = G T
#pragma unroll
            for (int al = 0; al < 6; ++al) {
                double v0 = 0.0, v1 = 0.0;
#pragma unroll
                for (int mu = 0; mu < 6; ++mu) {
                    v0 += ob.G0[mu] * Tm[mu][al];
                    v1 += ob.G1[mu] * Tm[mu][al];
                }
                jc0[al] = v0;
                jc1[al] = v1;
            }
            double W[3][6];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int al = 0; al < 6; ++al) W[p][al] = ob.jx0[p] * jc0[al] + ob.jx1[p] * jc1[al];
#pragma unroll
            for (int al = 0; al < 6; ++al) {
                if (Wd) {
                    Wd[o * 18 + 0 * 6 + al] = W[0][al];
                    Wd[o * 18 + 1 * 6 + al] = W[1][al];
                    Wd[o * 18 + 2 * 6 + al] = W[2][al];
                }
                if (Zd) {
                    Zd[o * 18 + 0 * 6 + al] = Vi[0] * W[0][al] + Vi[1] * W[1][al] + Vi[2] * W[2][al];
                    Zd[o * 18 + 1 * 6 + al] = Vi[1] * W[0][al] + Vi[3] * W[1][al] + Vi[4] * W[2][al];
                    Zd[o * 18 + 2 * 6 + al] = Vi[2] * W[0][al] + Vi[4] * W[1][al] + Vi[5] * W[2][al];
                }
                if (emit) {     // camera x phi column, right-hand side
                    const double sphi = W[0][al] * zw[0] + W[1][al] * zw[1] + W[2][al] * zw[2];
                    const double rr = W[0][al] * zg[0] + W[1][al] * zg[1] + W[2][al] * zg[2];
                    const int64_t ra = 6 * (int64_t)j + al;
                    __hip_atomic_fetch_add(a.S + nphi * a.ld + ra, -sphi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(a.rhs + ra, rr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            jd[o] = j;
            if (emit && a.first) note_block_first(a.first, j, lowest);
        }
    };
    // all ordered pairs (o in chunk A, o' in chunk B) of observations, 36 entries each: S[j, j'] -= W_o^T Z_o'
    auto pairs = [&](int na, int nb) {
        const int total = na * nb * 36;
        for (int w = lane; w < total; w += 64) {
            const int pair = w / 36, ent = w - pair * 36;
            const int o = pair / nb, o2 = pair - o * nb;
            const int al = ent / 6, be = ent - al * 6;
            const double v = Wl[o * 18 + al] * Zl[o2 * 18 + be] + Wl[o * 18 + 6 + al] * Zl[o2 * 18 + 6 + be] +
                             Wl[o * 18 + 12 + al] * Zl[o2 * 18 + 12 + be];
            if (jA[o] < jB[o2] || (jA[o] == jB[o2] && al < be)) continue;  // lower triangle only: nothing reads the rest
            const int64_t ra = 6 * (int64_t)jA[o] + al, cbx = 6 * (int64_t)jB[o2] + be;
            __hip_atomic_fetch_add(a.S + ra * a.ld + cbx, -v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    if (deg <= cap) {
        rows(0, deg, Wl, Zl, jA, true);
        for (int o = lane; o < deg; o += 64) jB[o] = a.oj[e0 + o];
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        pairs(deg, deg);
        return;
    }
    for (int ca = 0; ca < deg; ca += cap) {
        const int na = min(cap, deg - ca);
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();          // the pairs of the last trip have read W
        rows(ca, na, Wl, nullptr, jA, true);
        for (int cb = 0; cb < deg; cb += cap) {
            const int nb = min(cap, deg - cb);
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();      // ... and Z
            rows(cb, nb, nullptr, Zl, jB, false);
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
            pairs(na, nb);
        }
    }
}

// back-substitution: dX_i = V'^-1 (-gX_i - sum_j W_ij dcam_j - wphi_i dphi); one thread per point
struct BacksubArgs {
    const int32_t *oj;
    const double2 *ouv;
    const int64_t *row_ptr;
    int64_t N, K;
    double lambda;
    const double *X, *T, *scalars;
    const double *V, *Xphi, *gX;
    const double *d;  // solution of the reduced system, camera-major order (6j + k), phi at 6K
    const double *dframe;   // K x 6: every camera's step in its own frame (k_backsub_cameras)
    double *dX;
    double *X_trial;  // optional: X + dX, the trial point of the damped step
};

// A camera's step taken to the camera frame once, T_j d_j (T = blockdiag(R^T, A), see camera_frame_change): an
// observation's J_cam d_j is then G (T_j d_j), 12 multiply-adds instead of the 6 x 6 frame change per observation.
__global__ void __launch_bounds__(256) k_backsub_cameras(int64_t K, const double *__restrict__ T, const double *__restrict__ d,
                                                         double *__restrict__ dframe) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= K) return;
    double Tm[6][6];
    camera_frame_change(T + j * CAM_ROW, Tm);
#pragma unroll
    for (int mu = 0; mu < 6; ++mu) {
        double v = 0.0;
#pragma unroll
        for (int al = 0; al < 6; ++al) v += Tm[mu][al] * d[6 * j + al];
        dframe[6 * j + mu] = v;
    }
}

__global__ void __launch_bounds__(256) k_backsub(BacksubArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.N) return;
    const int64_t e0 = a.row_ptr[i], e1 = a.row_ptr[i + 1];
    const double x[3] = {a.X[3 * i], a.X[3 * i + 1], a.X[3 * i + 2]};
    if (e0 == e1) {
        a.dX[3 * i] = a.dX[3 * i + 1] = a.dX[3 * i + 2] = 0.0;
        if (a.X_trial) {
            a.X_trial[3 * i] = x[0];
            a.X_trial[3 * i + 1] = x[1];
            a.X_trial[3 * i + 2] = x[2];
        }
        return;
    }
    const double f = a.scalars[SC_F], pcx = a.scalars[SC_CX], pcy = a.scalars[SC_CY];
    const double dphi_x = a.d[6 * a.K];
    double rh[3] = {-a.gX[3 * i] - a.Xphi[3 * i] * dphi_x, -a.gX[3 * i + 1] - a.Xphi[3 * i + 1] * dphi_x,
                    -a.gX[3 * i + 2] - a.Xphi[3 * i + 2] * dphi_x};
    for (int64_t e = e0; e < e1; ++e) {
        const int j = a.oj[e];
        const double *row = a.T + (int64_t)j * CAM_ROW;
        ObsJac ob = obs_jacobian(x, (const double2 *)row, f, pcx, pcy, a.ouv[e]);
        // s_a = J_cam row a . dcam_j = G row a . (T_j dcam_j)
        const double *dj = a.dframe + 6 * (int64_t)j;
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int mu = 0; mu < 6; ++mu) {
            const double t = dj[mu];
            s0 = fma(ob.G0[mu], t, s0);
            s1 = fma(ob.G1[mu], t, s1);
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) rh[p] -= ob.jx0[p] * s0 + ob.jx1[p] * s1;
    }
    double Vi[6];
    inv_sym3(a.V + 6 * i, a.lambda, Vi);
    const double dx[3] = {Vi[0] * rh[0] + Vi[1] * rh[1] + Vi[2] * rh[2], Vi[1] * rh[0] + Vi[3] * rh[1] + Vi[4] * rh[2],
                          Vi[2] * rh[0] + Vi[4] * rh[1] + Vi[5] * rh[2]};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        a.dX[3 * i + c] = dx[c];
        if (a.X_trial) a.X_trial[3 * i + c] = x[c] + dx[c];
    }
}

// observation range of every point from the sorted point indices
__global__ void k_row_ptr(const int32_t *__restrict__ oi, int64_t nV, int64_t N, int64_t *__restrict__ row_ptr) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e > nV) return;
    const int64_t cur = e < nV ? oi[e] : N, prev = e > 0 ? oi[e - 1] : -1;
    for (int64_t p = prev + 1; p <= cur; ++p) row_ptr[p] = e;
}

__global__ void k_max_degree(const int64_t *__restrict__ row_ptr, int64_t N, int *out) {
    int d = 0;      // (a few hundred atomics on the one address, not one per wave of a million points)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x)
        d = max(d, (int)(row_ptr[i + 1] - row_ptr[i]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) d = max(d, __shfl_xor(d, off));
    if ((threadIdx.x & 63) == 0 && d > 0) atomicMax(out, d);
}

// solution in camera-major order -> packed [dC (3K) | dTheta (3K) | dphi]
// out (optional) receives the packed step, trial (optional) the packed parameters cam + step
__global__ void k_unpack_solution(int64_t K, const double *__restrict__ d, double *__restrict__ out,
                                  const double *__restrict__ cam, double *__restrict__ trial) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > 6 * K) return;
    int64_t at = 6 * K;
    if (t < 6 * K) {
        const int64_t j = t / 6, k = t - 6 * j;
        at = (k < 3 ? 0 : 3 * K) + 3 * j + (k % 3);
    }
    if (out) out[at] = d[t];
    if (trial) trial[at] = cam[at] + d[t];
}

static int cdiv_i(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace ba

using namespace ba;

int ba::ensure_row_ptr(ba_problem *p) {
    if (p->row_ptr) return BA_OK;
    BA_HIP(hipMalloc(&p->row_ptr, sizeof(int64_t) * (p->N + 1)));
    hipLaunchKernelGGL(k_row_ptr, dim3(cdiv_i(p->nV + 1, 256)), dim3(256), 0, 0, p->oi, p->nV, p->N, p->row_ptr);
    int *dmax = p->flag + 3;
    BA_HIP(hipMemset(dmax, 0, sizeof(int)));
    hipLaunchKernelGGL(k_max_degree, dim3(std::min(cdiv_i(p->N, 256), 256)), dim3(256), 0, 0, p->row_ptr, p->N, dmax);
    BA_HIP(hipGetLastError());
    BA_HIP(hipMemcpy(&p->max_degree, dmax, sizeof(int), hipMemcpyDeviceToHost));
    return BA_OK;
}

extern "C" {

int ba_eval_jtj(ba_problem *p, const double *X, const double *C, const double *Theta, const double *phi, int flags,
                double *E_sum, double *gX, double *gC, double *gTheta, double *gphi, double *V, double *B,
                double *bphi, double *hphi, void *stream) {
    BA_REQUIRE(p && X && C && Theta && phi, "ba_eval_jtj: null pointer");
    BA_REQUIRE(E_sum && gX && gC && gTheta && gphi && V && B && bphi && hphi, "ba_eval_jtj: null output");
    if (!p->tiled) {
        set_error("ba_eval_jtj: the scene has no LDS tiles (camera lists too long for a workgroup)");
        return BA_ERR_STATE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int64_t K = p->K;
    const size_t lds_runs = sizeof(double) * run_pass_lds_doubles<JtjOp>(p->cam_cap, p->run_pt_stride, p->run_slots_jtj);
    const size_t lds_blocks = sizeof(double) * block_pass_lds_doubles(PTV, p->cam_cap, p->blk_pt_stride);
    const bool use_blocks = p->blk_ptr && (flags & BA_EVAL_DENSE_BLOCKS) && !(flags & BA_EVAL_POINT_MAJOR) && lds_blocks <= 160 * 1024 - 256;
    const bool use_runs = !use_blocks && p->run_ptr && !(flags & BA_EVAL_POINT_MAJOR) && lds_runs <= 160 * 1024 - 256;
    const size_t lds = use_blocks ? lds_blocks
                       : use_runs ? lds_runs
                                  : sizeof(double) * ((size_t)(ROW_LDS + CAMV) * p->cam_cap + (size_t)(3 + PTV) * p->pt_cap);
    if (lds > 160 * 1024 - 256) {
        set_error("ba_eval_jtj: tile accumulators exceed the LDS of a workgroup");
        return BA_ERR_STATE;
    }
    if (!p->cam_sums_jtj) BA_HIP(hipMalloc(&p->cam_sums_jtj, sizeof(double) * CAMV * K));
    if (!p->partials_jtj) BA_HIP(hipMalloc(&p->partials_jtj, sizeof(double) * 3 * (size_t)p->n_tiles));
    if (!(flags & BA_EVAL_REUSE_TABLE) || p->table_centres_stale)     // the Schur kernels read the rows' centres
        if (int rc = camera_table(p, C, Theta, phi, st)) return rc;
    BA_HIP(hipMemsetAsync(p->cam_sums_jtj, 0, sizeof(double) * CAMV * K, st));
    JtjArgs a;
    a.opack = p->opack;
    a.ouv = p->ouv;
    a.meta = (const TileMeta *)p->tile_meta;
    a.cam_list = p->cam_list;
    a.K = (int)K;
    a.cam_cap = p->cam_cap;
    a.pt_cap = p->pt_cap;
    a.X = X;
    a.T = p->cam_table;
    a.scalars = p->scalars;
    a.gX = gX;
    a.V = V;
    a.cam_sums = p->cam_sums_jtj;
    a.partials = p->partials_jtj;
    RunArgs ra;
    ra.rpack = p->rpack;
    ra.ruv = p->ruv;
    ra.rrec = p->rrec;
    ra.run_ptr = p->run_ptr;
    ra.meta = a.meta;
    ra.cam_list = p->cam_list;
    ra.K = (int)K;
    ra.cam_cap = p->cam_cap;
    ra.pt_stride = p->run_pt_stride;
    ra.cacc_slots = p->run_slots_jtj;
    ra.X = X;
    ra.T = p->cam_table;
    ra.C = C;
    ra.scalars = p->scalars;
    ra.out[0] = gX;
    ra.out[1] = V;
    ra.cam_sums = p->cam_sums_jtj;
    ra.partials = p->partials_jtj;
    BlockArgs bk;
    bk.brec = p->brec;
    bk.buv = p->buv;
    bk.blk_ptr = p->blk_ptr;
    bk.meta = a.meta;
    bk.cam_list = p->cam_list;
    bk.K = (int)K;
    bk.cam_cap = p->cam_cap;
    bk.pt_stride = p->blk_pt_stride;
    bk.X = X;
    bk.T = p->cam_table;
    bk.C = C;
    bk.scalars = p->scalars;
    bk.out[0] = gX;
    bk.out[1] = V;
    bk.cam_sums = p->cam_sums_jtj;
    bk.partials = p->partials_jtj;
    const void *fn = use_blocks ? (p->compact ? (const void *)k_block_pass<JtjOp, BlkRecI16> : (const void *)k_block_pass<JtjOp, BlkRecF64>)
                     : !use_runs ? (const void *)k_jtj_tiled
                     : p->compact ? (const void *)k_run_pass<JtjOp, RecI16> : (const void *)k_run_pass<JtjOp, RecF64>;
    if (lds > 64 * 1024) BA_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const bool timed = (flags & BA_EVAL_TIMED) != 0 && p->ev_count < TIMING_RING;
    // a timed launch hands its events to the dispatch itself: the kernel's own begin and end, no event packets
    hipEvent_t ev_b = timed ? p->ev_begin[p->ev_count] : nullptr, ev_e = timed ? p->ev_end[p->ev_count] : nullptr;
#define BA_LAUNCH_MAYBE_TIMED(kern, grid, block, arg)                                          \
    do {                                                                                       \
        if (timed) hipExtLaunchKernelGGL(kern, grid, block, (uint32_t)lds, st, ev_b, ev_e, 0, arg); \
        else hipLaunchKernelGGL(kern, grid, block, lds, st, arg);                              \
    } while (0)
    if (use_blocks && p->compact)
        BA_LAUNCH_MAYBE_TIMED((k_block_pass<JtjOp, BlkRecI16>), dim3(p->n_tiles), dim3(64 * p->blk_waves), bk);
    else if (use_blocks)
        BA_LAUNCH_MAYBE_TIMED((k_block_pass<JtjOp, BlkRecF64>), dim3(p->n_tiles), dim3(64 * p->blk_waves), bk);
    else if (use_runs && p->compact)
        BA_LAUNCH_MAYBE_TIMED((k_run_pass<JtjOp, RecI16>), dim3(p->n_tiles), dim3(TILE_BLOCK), ra);
    else if (use_runs)
        BA_LAUNCH_MAYBE_TIMED((k_run_pass<JtjOp, RecF64>), dim3(p->n_tiles), dim3(TILE_BLOCK), ra);
    else
        BA_LAUNCH_MAYBE_TIMED(k_jtj_tiled, dim3(p->n_tiles), dim3(TILE_BLOCK), a);
#undef BA_LAUNCH_MAYBE_TIMED
    BA_HIP(hipGetLastError());
    if (timed) p->ev_count++;
    hipLaunchKernelGGL(k_jtj_finalize, dim3(cdiv_i(K, 256)), dim3(256), 0, st, K, p->n_tiles, p->partials_jtj,
                       p->cam_sums_jtj, p->cam_table, p->scalars, E_sum, gphi, hphi, gC, gTheta, B, bphi,
                       (use_runs || use_blocks) ? 1 : 0);
    BA_HIP(hipGetLastError());
    return BA_OK;
}

int ba_build_camera_table(ba_problem *p, const double *C, const double *Theta, const double *phi, void *stream) {
    BA_REQUIRE(p && C && Theta && phi, "ba_build_camera_table: null pointer");
    return camera_table(p, C, Theta, phi, (hipStream_t)stream);
}


// Lowest and highest camera slot (of its tile) among a point's observations; {INT_MAX, -1} for a point nobody sees.
__global__ void k_point_slot_range(int64_t N, const int64_t *__restrict__ row_ptr, const uint32_t *__restrict__ opack,
                                   int2 *__restrict__ range) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    int lo = 0x7fffffff, hi = -1;
    for (int64_t e = row_ptr[i]; e < row_ptr[i + 1]; ++e) {
        const int s = (int)(opack[e] & 0xffffu);
        lo = min(lo, s);
        hi = max(hi, s);
    }
    range[i] = make_int2(lo, hi);
}

// Cuts every tile into ranges of consecutive points whose cameras lie within a window of SCHUR_GEMM_WINDOW slots of
// the tile (k_schur_gemm's accumulators hold that many cameras): in the tracks of a video a tile of 384 points with
// 18 cameras becomes two ranges with 13 and 12.  A tile with a point that alone sees cameras further apart (the seam
// of a closed camera path) is left to the pair kernel.
constexpr int SCHUR_GEMM_WINDOW = 13;       // 6 x 13 + 2 = 80 columns: five tile rows of 16
// Where no point has more than ten observations a window of 10 slots holds every point too, and its 62 columns are
// four tile rows: 80 accumulator registers instead of 120, three waves per SIMD instead of two (the kernel keeps
// neither the matrix cores nor the vector ALU nor the LDS busy half the time: it needs the waves).  LM iteration at
// 2 000 cameras 2.17 -> 2.08 ms; at 20 000 (17.7 cameras per tile, more ranges) unchanged.
static int schur_window(const ba_problem *p) {
    const char *env = getenv("BA_SCHUR_WINDOW");
    const int w = env ? atoi(env) : (p->max_degree <= 10 ? 10 : SCHUR_GEMM_WINDOW);
    return w >= 2 && w <= SCHUR_GEMM_WINDOW ? w : SCHUR_GEMM_WINDOW;
}
static int ensure_schur_subtiles(ba_problem *p) {
    if (p->n_schur_rest >= 0) return BA_OK;
    const int64_t N = p->N;
    const int nt = p->n_tiles;
    int2 *d_range = nullptr;
    BA_HIP(hipMalloc(&d_range, sizeof(int2) * N));
    hipLaunchKernelGGL(k_point_slot_range, dim3(cdiv_i(N, 256)), dim3(256), 0, 0, N, p->row_ptr, p->opack, d_range);
    std::vector<int2> range((size_t)N);
    std::vector<TileMeta> meta((size_t)nt);
    hipError_t e = hipMemcpy(range.data(), d_range, sizeof(int2) * N, hipMemcpyDeviceToHost);
    hipFree(d_range);
    BA_HIP(e);
    BA_HIP(hipMemcpy(meta.data(), p->tile_meta, sizeof(TileMeta) * nt, hipMemcpyDeviceToHost));
    std::vector<int4> subs;
    std::vector<int> rest((size_t)nt, 0);
    int n_rest = 0;
    auto cut = [&](int window) {
    subs.clear();
    std::fill(rest.begin(), rest.end(), 0);
    n_rest = 0;
    for (int t = 0; t < nt; ++t) {
        const TileMeta &tm = meta[t];
        if (tm.e0 >= tm.e1) continue;
        const size_t mark = subs.size();
        bool ok = true;
        int begin = 0, lo = 0x7fffffff, hi = -1;
        auto close = [&](int end) {
            if (end > begin) subs.push_back(make_int4(t, begin, end - begin, hi >= lo ? (lo | ((hi - lo + 1) << 16)) : (0 | (1 << 16))));
        };
        for (int pt = 0; pt < tm.np && ok; ++pt) {
            const int2 r = range[(size_t)tm.p0 + pt];
            if (r.y < r.x) continue;                                   // nobody sees it: it rides along
            if (r.y - r.x + 1 > window) ok = false;
            else if (hi >= lo && std::max(hi, r.y) - std::min(lo, r.x) + 1 > window) {
                close(pt);
                begin = pt, lo = r.x, hi = r.y;
            } else {
                lo = std::min(lo, r.x), hi = std::max(hi, r.y);
            }
        }
        if (ok) {
            close(tm.np);
        } else {
            subs.resize(mark);
            rest[t] = 1;
            ++n_rest;
        }
    }
    };
    // the narrow window (see schur_window) if it leaves no more tiles to the pair kernel than the wide one does:
    // tracks with gaps span more camera slots than they have observations
    int window = schur_window(p);
    cut(window);
    if (window < SCHUR_GEMM_WINDOW && !getenv("BA_SCHUR_WINDOW") && n_rest > 0) {
        const int narrow_rest = n_rest;
        cut(SCHUR_GEMM_WINDOW);
        if (n_rest < narrow_rest) window = SCHUR_GEMM_WINDOW;
        else cut(window);
    }
    p->schur_window = window;
    hipFree(p->schur_sub), hipFree(p->schur_rest);
    p->schur_sub = nullptr, p->schur_rest = nullptr;
    if (!subs.empty()) {
        BA_HIP(hipMalloc(&p->schur_sub, sizeof(int4) * subs.size()));
        BA_HIP(hipMemcpy(p->schur_sub, subs.data(), sizeof(int4) * subs.size(), hipMemcpyHostToDevice));
    }
    BA_HIP(hipMalloc(&p->schur_rest, sizeof(int) * (nt > 0 ? nt : 1)));
    BA_HIP(hipMemcpy(p->schur_rest, rest.data(), sizeof(int) * nt, hipMemcpyHostToDevice));
    p->n_schur_sub = (int)subs.size();
    p->n_schur_rest = n_rest;
    if (getenv("BA_VERBOSE"))
        fprintf(stderr, "[ba] schur: %d ranges of points for the matrix-core kernel, %d of %d tiles left to the pair kernel\n",
                p->n_schur_sub, n_rest, nt);
    return BA_OK;
}

int ba_schur_build(ba_problem *p, const double *X, double lambda, const double *V, double *Xphi, const double *gX,
                   const double *B, const double *bphi, const double *hphi, const double *gC, const double *gTheta,
                   const double *gphi, int with_camera_blocks, double *S, double *rhs, void *stream) {
    return ba_schur_build_tracked(p, X, lambda, V, Xphi, gX, B, bphi, hphi, gC, gTheta, gphi, with_camera_blocks, S, rhs,
                                  nullptr, 0, stream);
}

int ba_schur_build_tracked(ba_problem *p, const double *X, double lambda, const double *V, double *Xphi, const double *gX,
                           const double *B, const double *bphi, const double *hphi, const double *gC, const double *gTheta,
                           const double *gphi, int with_camera_blocks, double *S, double *rhs, int *block_first,
                           int S_is_zero, void *stream) {
    BA_REQUIRE(p && X && V && Xphi && gX && B && bphi && hphi && gC && gTheta && gphi && S && rhs,
               "ba_schur_build: null pointer");
    if (int rc = ensure_row_ptr(p)) return rc;
    hipStream_t st = (hipStream_t)stream;
    int64_t n, ld;
    ba_schur_dim(p->K, &n, &ld);
    // rows of W and Z a wave of the point kernel keeps in LDS; points with more observations go through in chunks
    constexpr int SCHUR_POINT_ROWS = 128;
    const int point_rows = getenv("BA_SCHUR_POINT_ROWS") ? std::max(1, atoi(getenv("BA_SCHUR_POINT_ROWS")))
                                                          : std::min(p->max_degree, SCHUR_POINT_ROWS);
    const size_t lds = (size_t)4 * point_rows * 37 * sizeof(double);
    const char *form = getenv("BA_SCHUR_FORM");      // 0: entry-per-lane pair loop; 1: pairs in registers, 16 waves; 2: 8 waves
    const int which = form ? atoi(form) : 2;
    const int tile_waves = which < 2 ? 16 : 8;
    const size_t lds_tile = p->tiled ? sizeof(double) * schur_tile_lds_doubles(p->cam_cap, p->max_degree, tile_waves) : ~(size_t)0;
    const bool use_tile = p->tiled && p->tiles_worth && lds_tile <= 150 * 1024 && !getenv("BA_SCHUR_POINTS");
    if (!use_tile && lds > 160 * 1024 - 256) {
        set_error("ba_schur_build: BA_SCHUR_POINT_ROWS asks for more LDS than a workgroup has");
        return BA_ERR_ARG;
    }
    if (!S_is_zero) BA_HIP(hipMemsetAsync(S, 0, sizeof(double) * ld * ld, st));
    if (with_camera_blocks) {
        hipLaunchKernelGGL(k_schur_init, dim3(cdiv_i(std::max<int64_t>(p->K, ld / 64), 256)), dim3(256), 0, st, p->K, n, ld,
                           lambda, B, bphi, hphi, gC, gTheta, gphi, S, rhs, block_first);
    } else {
        if (block_first)
            hipLaunchKernelGGL(k_schur_first_init, dim3(cdiv_i(ld / 64, 256)), dim3(256), 0, st, block_first, ld / 64, p->K);
        BA_HIP(hipMemsetAsync(rhs, 0, sizeof(double) * ld, st));
    }
    if (use_tile) {
        SchurTileArgs ta;
        ta.opack = p->opack;
        ta.ouv = p->ouv;
        ta.row_ptr = p->row_ptr;
        ta.meta = (const TileMeta *)p->tile_meta;
        ta.cam_list = p->cam_list;
        ta.K = p->K;
        ta.ld = ld;
        ta.cam_cap = p->cam_cap;
        ta.max_deg = p->max_degree;
        ta.lambda = lambda;
        ta.X = X;
        ta.T = p->cam_table;
        ta.scalars = p->scalars;
        ta.V = V;
        ta.Xphi = Xphi;
        ta.gX = gX;
        ta.S = S;
        ta.rhs = rhs;
        ta.first = block_first;
        // (the kernel needs about 200 registers: two waves per SIMD, one workgroup of 8 waves per CU whatever the LDS
        // allows; builds held to 168 or 128 registers spill and run 1.3 to 3.5 times as long)
        auto launch = [&](auto kernel, int block) -> int {
            if (lds_tile > 64 * 1024)
                BA_HIP(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_tile));
            hipLaunchKernelGGL(kernel, dim3(p->n_tiles), dim3(block), lds_tile, st, ta);
            return BA_OK;
        };
        // The Gram-matrix form on the matrix cores for every range of points whose cameras fit its window
        // (ensure_schur_subtiles); the tiles it cannot take (if any: in a closed camera path those at the seam, which
        // hold its first and last cameras) go to the pair kernel on a second stream beside it - a few workgroups that
        // each run for the length of a fifth of the other launch.  Both add to S with atomics.
        if (!form && !getenv("BA_SCHUR_NO_MFMA")) {
            if (int rc = ensure_schur_subtiles(p)) return rc;
            const bool rest = p->n_schur_rest > 0;
            if (rest) {
                if (!p->aux_stream) {
                    BA_HIP(hipStreamCreateWithFlags(&p->aux_stream, hipStreamNonBlocking));
                    BA_HIP(hipEventCreateWithFlags(&p->aux_fork, hipEventDisableTiming));
                    BA_HIP(hipEventCreateWithFlags(&p->aux_join, hipEventDisableTiming));
                }
                // (launched BEFORE the long kernel: behind it, its workgroups - a CU's whole LDS each - get no CU
                // until the other launch has drained, 0.14 ms later at 2 000 cameras)
                BA_HIP(hipEventRecord(p->aux_fork, st));
                BA_HIP(hipStreamWaitEvent(p->aux_stream, p->aux_fork, 0));
                SchurTileArgs tr = ta;
                tr.only = p->schur_rest;
                if (lds_tile > 64 * 1024)
                    BA_HIP(hipFuncSetAttribute((const void *)k_schur_tile<true, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_tile));
                hipLaunchKernelGGL((k_schur_tile<true, 512>), dim3(p->n_tiles), dim3(512), lds_tile, p->aux_stream, tr);
                BA_HIP(hipEventRecord(p->aux_join, p->aux_stream));
            }
            if (p->n_schur_sub > 0) {
                SchurTileArgs tg = ta;
                tg.sub = p->schur_sub;
                const int window = p->schur_window;
                tg.cam_cap = window;
                const bool four_rows = 6 * window + 2 <= 64;
                const size_t lds_gemm = sizeof(double) * schur_gemm_lds_doubles(tg.cam_cap);
                // six points per trip where no point has more than ten observations, else four
                const bool six = p->max_degree <= 10 && !getenv("BA_SCHUR_FOUR_POINTS");
                constexpr int NBT = (6 * SCHUR_GEMM_WINDOW + 2 + 15) / 16;
                const void *kernel = four_rows ? (six ? (const void *)k_schur_gemm<4, 6> : (const void *)k_schur_gemm<4, 4>)
                                               : (six ? (const void *)k_schur_gemm<NBT, 6> : (const void *)k_schur_gemm<NBT, 4>);
                if (lds_gemm > 64 * 1024)
                    BA_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_gemm));
                const dim3 grid(p->n_schur_sub), block(SCHUR_GEMM_BLOCK);
                if (four_rows && six) hipLaunchKernelGGL((k_schur_gemm<4, 6>), grid, block, lds_gemm, st, tg);
                else if (four_rows) hipLaunchKernelGGL((k_schur_gemm<4, 4>), grid, block, lds_gemm, st, tg);
                else if (six) hipLaunchKernelGGL((k_schur_gemm<NBT, 6>), grid, block, lds_gemm, st, tg);
                else hipLaunchKernelGGL((k_schur_gemm<NBT, 4>), grid, block, lds_gemm, st, tg);
            }
            if (rest) BA_HIP(hipStreamWaitEvent(st, p->aux_join, 0));
            BA_HIP(hipGetLastError());
            return BA_OK;
        }
        int rc;
        if (which == 0) rc = launch(k_schur_tile<false, 1024>, 1024);
        else if (which == 1) rc = launch(k_schur_tile<true, 1024>, 1024);
        else rc = launch(k_schur_tile<true, 512>, 512);
        if (rc) return rc;
        BA_HIP(hipGetLastError());
        return BA_OK;
    }
    SchurArgs a;
    a.oj = p->oj;
    a.ouv = p->ouv;
    a.row_ptr = p->row_ptr;
    a.N = p->N;
    a.K = p->K;
    a.ld = ld;
    a.lambda = lambda;
    a.X = X;
    a.T = p->cam_table;
    a.scalars = p->scalars;
    a.V = V;
    a.Xphi = Xphi;
    a.gX = gX;
    a.S = S;
    a.rhs = rhs;
    a.max_deg = point_rows;
    a.first = block_first;
    if (lds > 64 * 1024)
        BA_HIP(hipFuncSetAttribute((const void *)k_schur_points, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_schur_points, dim3(cdiv_i(p->N, 4)), dim3(256), lds, st, a);
    BA_HIP(hipGetLastError());
    return BA_OK;
}

int ba_backsubstitute(ba_problem *p, const double *X, double lambda, const double *V, const double *Xphi,
                      const double *gX, const double *d, double *dcam_packed, double *dX, const double *cam,
                      double *X_trial, double *cam_trial, void *stream) {
    BA_REQUIRE(p && X && V && Xphi && gX && d && dX, "ba_backsubstitute: null pointer");
    BA_REQUIRE(!cam_trial || cam, "ba_backsubstitute: cam_trial needs the current parameters (cam)");
    BA_REQUIRE(X_trial != X && X_trial != dX, "ba_backsubstitute: X_trial must be a buffer of its own");
    if (int rc = ensure_row_ptr(p)) return rc;
    hipStream_t st = (hipStream_t)stream;
    BacksubArgs a;
    a.oj = p->oj;
    a.ouv = p->ouv;
    a.row_ptr = p->row_ptr;
    a.N = p->N;
    a.K = p->K;
    a.lambda = lambda;
    a.X = X;
    a.T = p->cam_table;
    a.scalars = p->scalars;
    a.V = V;
    a.Xphi = Xphi;
    a.gX = gX;
    a.d = d;
    a.dX = dX;
    a.X_trial = X_trial;
    // (the camera sums of the J^T J pass serve as scratch between two passes: the pass clears them before it starts)
    if (!p->cam_sums_jtj) BA_HIP(hipMalloc(&p->cam_sums_jtj, sizeof(double) * CAMV * p->K));
    a.dframe = p->cam_sums_jtj;
    hipLaunchKernelGGL(k_backsub_cameras, dim3(cdiv_i(p->K, 256)), dim3(256), 0, st, p->K, p->cam_table, d, p->cam_sums_jtj);
    hipLaunchKernelGGL(k_backsub, dim3(cdiv_i(p->N, 256)), dim3(256), 0, st, a);
    if (dcam_packed || cam_trial)
        hipLaunchKernelGGL(k_unpack_solution, dim3(cdiv_i(6 * p->K + 1, 256)), dim3(256), 0, st, p->K, d, dcam_packed,
                           cam, cam_trial);
    BA_HIP(hipGetLastError());
    return BA_OK;
}

}  // extern "C"
