// Device-side pieces shared by the kernels of the engine (internal).
#pragma once

#include "ba_common.h"

namespace ba {

// --------------------------------------------------------------------------------------------
// small device helpers
// --------------------------------------------------------------------------------------------
// value of the lane N positions lower in the same 16-lane row (0.0 / 0 where there is none)
template <int N>
__device__ __forceinline__ int row_shr(int v) {
    return __builtin_amdgcn_update_dpp(0, v, 0x110 + N, 0xf, 0xf, true);
}
template <int N>
__device__ __forceinline__ double row_shr(double v) {
    const int lo = row_shr<N>(__double2loint(v)), hi = row_shr<N>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// Sum over the 64 lanes of a wave (all of them active), the same value in every lane: running sums inside the four
// 16-lane rows by DPP shifts, then the four row totals through scalar registers.  (Butterfly exchanges of doubles
// go through the LDS crossbar, twelve dependent trips per sum.)
__device__ __forceinline__ double wave_sum(double v) {
    v += row_shr<1>(v);
    v += row_shr<2>(v);
    v += row_shr<4>(v);
    v += row_shr<8>(v);
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 15), __builtin_amdgcn_readlane(lo, 15));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 31), __builtin_amdgcn_readlane(lo, 31));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 47), __builtin_amdgcn_readlane(lo, 47));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 63), __builtin_amdgcn_readlane(lo, 63));
    return (r0 + r1) + (r2 + r3);
}

// 1 / x without the range scaling and fix-up of the IEEE division sequence (x is a depth along the
// optical axis, a determinant, a moment's root: within the range of a float): single-precision hardware estimate
// (2^-23; the double-precision estimate is no better than that for the two steps that follow, and slower: the
// J^T J pass -2 %) and two Newton steps, relative error of a few 1e-16
__device__ __forceinline__ double fast_rcp(double x) {
    double r = (double)__builtin_amdgcn_rcpf((float)x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// the same with the double-precision estimate: for arguments that may leave the range of a float (a determinant)
__device__ __forceinline__ double fast_rcp_wide(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// 1 / sqrt(x) for x > 0 far from the ends of the exponent range: hardware estimate and two Newton steps (relative
// error of a few 1e-16), a dependent chain of 9 instructions where sqrt followed by a division has about 55
__device__ __forceinline__ double fast_rsqrt(double x) {
    double r = __builtin_amdgcn_rsq(x);
    r = fma(0.5 * r, fma(-(x * r), r, 1.0), r);
    r = fma(0.5 * r, fma(-(x * r), r, 1.0), r);
    return r;
}

// maximum over the 64 lanes of a wave of values >= 0 (lanes a row shift has no source for contribute 0)
__device__ __forceinline__ double wave_max_nonneg(double v) {
    v = fmax(v, row_shr<1>(v));
    v = fmax(v, row_shr<2>(v));
    v = fmax(v, row_shr<4>(v));
    v = fmax(v, row_shr<8>(v));
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 15), __builtin_amdgcn_readlane(lo, 15));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 31), __builtin_amdgcn_readlane(lo, 31));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 47), __builtin_amdgcn_readlane(lo, 47));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 63), __builtin_amdgcn_readlane(lo, 63));
    return fmax(fmax(r0, r1), fmax(r2, r3));
}

// wave_sum over the lanes of one half of the wave (half 0: lanes 0..31, half 1: lanes 32..63; all 64 lanes active)
__device__ __forceinline__ double half_wave_sum(double v, int half) {
    v += row_shr<1>(v);
    v += row_shr<2>(v);
    v += row_shr<4>(v);
    v += row_shr<8>(v);
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 15), __builtin_amdgcn_readlane(lo, 15));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 31), __builtin_amdgcn_readlane(lo, 31));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 47), __builtin_amdgcn_readlane(lo, 47));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 63), __builtin_amdgcn_readlane(lo, 63));
    return half ? r2 + r3 : r0 + r1;
}

// R = Ry(ty) Rx(tx) Rz(tz), row-major (p2w:114-140)
__device__ __forceinline__ void rotation_yxz(double tx, double ty, double tz, double *R, double *cz_out,
                                             double *sz_out) {
    double sx, cx, sy, cy, sz, cz;
    sincos(tx, &sx, &cx);
    sincos(ty, &sy, &cy);
    sincos(tz, &sz, &cz);
    const double sysx = sy * sx, cysx = cy * sx;
    R[0] = cy * cz + sysx * sz;
    R[1] = -cy * sz + sysx * cz;
    R[2] = sy * cx;
    R[3] = cx * sz;
    R[4] = cx * cz;
    R[5] = -sx;
    R[6] = -sy * cz + cysx * sz;
    R[7] = sy * sz + cysx * cz;
    R[8] = cy * cx;
    *cz_out = cz;
    *sz_out = sz;
}

// Scale factor of a gauge record (16 doubles: R0 (9), C0 (3), [12] factor, [13] j1, [14] squared longest baseline,
// [15] flag).  The single-launch camera steps know the longest baseline when they end and store the factor; the
// ticket-free step only collects the maximum ([14], by atomic max on the bit pattern) and leaves 1 / sqrt to the
// readers ([15] != 0): every one of them evaluates this same expression, so all see the same number (a few 1e-16
// from the correctly rounded quotient; the chain of nine instructions sits in the prologue of every tile).
__device__ __forceinline__ double gauge_factor(const double *__restrict__ G) {
    return G[15] != 0.0 ? fast_rsqrt(G[14]) : G[12];
}

// update rules (p2w:336-360)
struct UpdateRule {
    int mode;  // 0 = leave unchanged, 1 = SGD (p2w:336-342), 2 = Adam (p2w:344-360)
    double lr, gscale;
    double ibc1, ibc2;   // 1 / (1 - rho_1^t), 1 / (1 - rho_2^t): the bias corrections as factors
};

// One parameter: returns the new value, updates the moments in place.  The two bias corrections are
// multiplications by reciprocals formed once on the host and the final quotient is a multiplication by
// the Newton reciprocal of the denominator (>= 1e-12): a few 1e-16 relative from the reference's three divisions,
// and a third of the instructions on the path every parameter takes every iteration.  The denominator is the root
// of a squared gradient and leaves the range of a float when one observation has a depth near zero (gradients of
// 1e39 and more, from which the reference recovers in one step of length lr): double-precision estimate.
__device__ __forceinline__ double apply_rule(const UpdateRule &u, double x, double g, double *s, double *r) {
    const double rho1 = 0.9, rho2 = 0.999;
    const double gr = g * u.gscale;
    if (u.mode == 2) {
        const double sn = rho1 * *s + (1.0 - rho1) * gr;
        const double rn = rho2 * *r + (1.0 - rho2) * (gr * gr);
        *s = sn;
        *r = rn;
        // sqrt(y) as y / sqrt(y) by the Newton reciprocal square root: a few 1e-16 from the correctly rounded root in
        // 9 instructions instead of the library sequence's 20 odd (the update is a sixth of the fused pass's arithmetic)
        const double y = rn * u.ibc2;
        const double root = y > 0.0 ? y * fast_rsqrt(y) : 0.0;
        return x - u.lr * (sn * u.ibc1) * fast_rcp_wide(1e-12 + root);
    }
    return x - u.lr * gr;
}

// Everything a workgroup needs to know about its tile, fetched with one scalar load.
struct TileMeta {
    int32_t e0, e1;   // observation range
    int32_t p0, np;   // first point and point span
    int32_t nc;       // distinct cameras
    int32_t j0;       // first camera when the cameras are the contiguous range j0 .. j0+nc-1, else -1
    int64_t cb;       // offset of the tile's camera list in cam_list
    int32_t wlo, whi; // points whose gradient this tile writes: its own and the unobserved ones up to
                      // the next tile's first point (so that every point is written exactly once)
};


constexpr int TILE_BLOCK = 256;
constexpr int TILE_WAVES = TILE_BLOCK / 64;
constexpr int ROW_LDS = 12;  // doubles of a camera row kept in LDS: R (9), C (3) or t (3)

// Camera row of the table -> LDS row [R (9) | t = -R^T C (3)] for the run / block passes, which then form
// q = R^T (x - C) as R^T x + t: three instructions fewer per observation (one thread per row)
// (the centre comes from the parameter vector, not from the row: the merged camera step leaves the rows' centres
// to whoever needs them, see ba_problem::table_centres_stale)
// `cscale`: what the stored centre still has to be multiplied by (1 unless a scale of the gauge fix is pending)
__device__ __forceinline__ void stage_camera_row_rt(const double *__restrict__ src, const double *__restrict__ centre,
                                                    double cscale, double *__restrict__ dst) {
    const double2 *s2 = (const double2 *)src;
    double2 v[6];
#pragma unroll
    for (int k = 0; k < 5; ++k) v[k] = s2[k];
    const double c0 = centre[0] * cscale, c1 = centre[1] * cscale, c2 = centre[2] * cscale;
    const double t0 = -(v[0].x * c0 + v[1].y * c1 + v[3].x * c2);   // column 0 of R: R00, R10, R20
    const double t1 = -(v[0].y * c0 + v[2].x * c1 + v[3].y * c2);
    const double t2 = -(v[1].x * c0 + v[2].y * c1 + v[4].x * c2);
    v[4].y = t0;
    v[5].x = t1;
    v[5].y = t2;
    double2 *d2 = (double2 *)dst;
#pragma unroll
    for (int k = 0; k < 6; ++k) d2[k] = v[k];
}

}  // namespace ba
