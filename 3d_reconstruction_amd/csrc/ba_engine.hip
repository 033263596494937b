// MI355X (gfx950, CDNA4) bundle-adjustment engine: kernels and C ABI of the observation pass,
// the optimiser updates, the gauge fix and the scoring helpers.
//
// Reference behaviour followed: modules/pix2world.py ("p2w") fit_scene hot loop, p2w:480-726.
// The reference evaluates every one of the N*K (point, camera) pairs with dense NumPy
// temporaries; this engine streams only the visible pairs, 24 bytes each, from HBM.
//
// Math of one observation (i, j) with pixel centre (u, v)           [p2w:555-573, 654-702]
//   d = X_i - C_j,  q = R_j^T d,  m = (q0, q1)/q2,  r = f m + (W/2, H/2) - (u, v)
//   g = (f/q2) (r0, r1, -(m . r))          = (d model / d q)^T r      ("camera-frame force")
//   dE/dX_i      +=  R_j g
//   dE/dC_j       = -R_j  sum_i g
//   dE/dTheta_jl  = -a_l . sum_i (q x g),  a_x = (cos tz, -sin tz, 0), a_y = row 1 of R, a_z = e_z
//   dE/dphi_x     = -(f / sin phi_x) sum (m . r)
// (dq/dtheta_l = -a_l x q for R = Ry Rx Rz, so the 27 entries of dR/dTheta the reference builds
//  at p2w:678-692 are never formed.)  All sums are un-normalised; the caller applies 1/|V|.
#include "ba_common.h"

#include <math.h>
#include <string.h>

namespace ba {

static thread_local std::string g_error;

void set_error(const std::string &msg) { g_error = msg; }

int hip_fail(hipError_t e, const char *what, const char *file, int line) {
    g_error = std::string("HIP error ") + hipGetErrorString(e) + " in " + what + " at " + file + ":" +
              std::to_string(line);
    return BA_ERR_HIP;
}

// --------------------------------------------------------------------------------------------
// small device helpers
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
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

// --------------------------------------------------------------------------------------------
// K0: visibility extraction (p2w:506-508, 531) — one wave per point row
// --------------------------------------------------------------------------------------------
__global__ void k_vis_count(const int2 *__restrict__ pix, int64_t N, int64_t K, int64_t *__restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= N) return;
    const int2 *p = pix + row * K;
    int cnt = 0;
    for (int64_t j0 = 0; j0 < K; j0 += 64) {
        const int64_t j = j0 + lane;
        const bool vis = (j < K) && (p[j].x >= 0);
        cnt += __popcll(__ballot(vis));
    }
    if (lane == 0) counts[row] = cnt;
}

__global__ void k_vis_fill(const int2 *__restrict__ pix, int64_t N, int64_t K, const int64_t *__restrict__ row_ptr,
                           int32_t *__restrict__ oi, int32_t *__restrict__ oj, double2 *__restrict__ ouv) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= N) return;
    const int2 *p = pix + row * K;
    int64_t pos = row_ptr[row];
    for (int64_t j0 = 0; j0 < K; j0 += 64) {
        const int64_t j = j0 + lane;
        int2 xy = make_int2(-1, -1);
        if (j < K) xy = p[j];
        const bool vis = xy.x >= 0;
        const unsigned long long mask = __ballot(vis);
        if (vis) {
            const int rank = __popcll(mask & ((1ull << lane) - 1ull));
            const int64_t e = pos + rank;
            oi[e] = (int32_t)row;
            oj[e] = (int32_t)j;
            ouv[e] = make_double2((double)xy.x + 0.5, (double)xy.y + 0.5);
        }
        pos += __popcll(mask);
    }
}

// exclusive prefix sum of int64 counts, two levels (block of 1024 threads x 8 items)
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = 1024 * SCAN_ITEMS;

__device__ __forceinline__ int64_t block_exclusive_scan_1024(int64_t v, int64_t *total, int64_t *lds /*[16]*/) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int64_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int64_t t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    if (lane == 63) lds[wid] = inc;
    __syncthreads();
    if (wid == 0) {
        int64_t w = (lane < 16) ? lds[lane] : 0;
        int64_t winc = w;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            const int64_t t = __shfl_up(winc, off);
            if (lane >= off) winc += t;
        }
        if (lane < 16) lds[lane] = winc - w;
        if (lane == 15) *total = winc;
    }
    __syncthreads();
    return inc - v + lds[wid];
}

__global__ void __launch_bounds__(1024) k_scan_tiles(const int64_t *in, int64_t n, int64_t *out,
                                                     int64_t *__restrict__ tile_sums) {
    __shared__ int64_t lds[16];
    __shared__ int64_t total;
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int64_t v[SCAN_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0;
        s += v[k];
    }
    int64_t ex = block_exclusive_scan_1024(s, &total, lds);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (base + k < n) out[base + k] = ex;
        ex += v[k];
    }
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

// single block: exclusive scan of the tile sums in place, grand total to tile_sums[n_tiles]
__global__ void __launch_bounds__(1024) k_scan_tile_sums(int64_t *tile_sums, int n_tiles) {
    __shared__ int64_t lds[16];
    __shared__ int64_t total;
    int64_t carry = 0;
    for (int base = 0; base < n_tiles; base += 1024) {
        const int t = base + threadIdx.x;
        const int64_t v = (t < n_tiles) ? tile_sums[t] : 0;
        const int64_t ex = block_exclusive_scan_1024(v, &total, lds);
        if (t < n_tiles) tile_sums[t] = ex + carry;
        carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_sums[n_tiles] = carry;
}

__global__ void k_scan_add(int64_t *__restrict__ out, int64_t n, const int64_t *__restrict__ tile_sums, int n_tiles) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] += tile_sums[t / SCAN_TILE];
    if (t == 0) out[n] = tile_sums[n_tiles];
}

// --------------------------------------------------------------------------------------------
// problem set-up kernels
// --------------------------------------------------------------------------------------------
// flag bit 0: index out of range, bit 1: not sorted by point
__global__ void k_validate_obs(const int32_t *__restrict__ oi, const int32_t *__restrict__ oj, int64_t nV, int64_t N,
                               int64_t K, int *flag, unsigned long long *heads) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool head = false;
    if (e < nV) {
        const int i = oi[e], j = oj[e];
        if (i < 0 || i >= N || j < 0 || j >= K) atomicOr(flag, 1);
        if (e > 0) {
            const int ip = oi[e - 1];
            if (ip > i) atomicOr(flag, 2);
            head = ip != i;
        } else {
            head = true;
        }
    }
    const unsigned long long m = __ballot(head);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(heads, (unsigned long long)__popcll(m));
}

// chunk c starts at the first point boundary at or after c * chunk (so that every point's
// observations belong to exactly one chunk)
__global__ void k_chunk_bounds(const int32_t *__restrict__ oi, int64_t nV, int n_chunks, int64_t chunk,
                               int32_t *__restrict__ chunk_begin) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n_chunks) return;
    int64_t t = (int64_t)c * chunk;
    if (c == n_chunks || t >= nV) {
        chunk_begin[c] = (int32_t)nV;
        return;
    }
    if (t > 0) {
        while (t < nV && oi[t] == oi[t - 1]) ++t;
    }
    chunk_begin[c] = (int32_t)t;
}

// --------------------------------------------------------------------------------------------
// K1: camera table + scalars
// --------------------------------------------------------------------------------------------
__global__ void k_camera_table(int64_t K, const double *__restrict__ C, const double *__restrict__ Theta,
                               const double *__restrict__ phi, double W, double H, double *__restrict__ T,
                               double *__restrict__ scalars) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0 && scalars) {
        const double ph = phi[0];
        const double f = W / 2.0 / tan(ph / 2.0);
        scalars[SC_F] = f;
        scalars[SC_CX] = W / 2.0;
        scalars[SC_CY] = H / 2.0;
        scalars[SC_DPHI] = -f / sin(ph);
    }
    if (j >= K) return;
    double R[9], cz, sz;
    rotation_yxz(Theta[3 * j + 0], Theta[3 * j + 1], Theta[3 * j + 2], R, &cz, &sz);
    double *row = T + j * CAM_ROW;
#pragma unroll
    for (int k = 0; k < 9; ++k) row[k] = R[k];
    row[9] = C[3 * j + 0];
    row[10] = C[3 * j + 1];
    row[11] = C[3 * j + 2];
    row[12] = cz;
    row[13] = sz;
    row[14] = 0.0;
    row[15] = 0.0;
}

// --------------------------------------------------------------------------------------------
// K2: fused observation pass
// --------------------------------------------------------------------------------------------
struct EvalArgs {
    const int32_t *oi;
    const int32_t *oj;
    const double2 *ouv;
    const int32_t *chunk_begin;
    int n_chunks;
    int K;
    const double *X;
    const double *T;
    const double *scalars;
    double *gX;        // may be null
    double *cam_sums;  // 6 x K, component-major (pre-zeroed)
    double *partials;  // per-wave {sum r^2, sum m.r}
};

template <bool LDS_CAM>
__global__ void __launch_bounds__(EVAL_BLOCK) k_eval(EvalArgs a) {
    extern __shared__ double lds_cam[];  // LDS_CAM: 6 x K accumulators, component-major
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int K = a.K;

    if (LDS_CAM) {
        for (int t = threadIdx.x; t < 6 * K; t += EVAL_BLOCK) lds_cam[t] = 0.0;
        __syncthreads();
    }

    const double f = a.scalars[SC_F], pcx = a.scalars[SC_CX], pcy = a.scalars[SC_CY];
    double esum = 0.0, psum = 0.0;
    const int wave = blockIdx.x * EVAL_WAVES + wid;
    const int n_waves = gridDim.x * EVAL_WAVES;

    for (int c = wave; c < a.n_chunks; c += n_waves) {
        const int e0 = a.chunk_begin[c], e1 = a.chunk_begin[c + 1];
        int carry_i = -1;
        double c0 = 0.0, c1 = 0.0, c2 = 0.0;
        for (int base = e0; base < e1; base += 64) {
            const int e = base + lane;
            const bool valid = e < e1;
            int i = -2;
            double g0 = 0.0, g1 = 0.0, g2 = 0.0;  // world-frame contribution to dE/dX_i
            if (valid) {
                i = a.oi[e];
                const int j = a.oj[e];
                const double2 uv = a.ouv[e];
                const double *x = a.X + 3 * (int64_t)i;
                const double *t = a.T + (int64_t)j * CAM_ROW;
                const double2 t0 = *(const double2 *)(t + 0), t1 = *(const double2 *)(t + 2),
                              t2 = *(const double2 *)(t + 4), t3 = *(const double2 *)(t + 6),
                              t4 = *(const double2 *)(t + 8), t5 = *(const double2 *)(t + 10);
                const double R00 = t0.x, R01 = t0.y, R02 = t1.x, R10 = t1.y, R11 = t2.x, R12 = t2.y, R20 = t3.x,
                             R21 = t3.y, R22 = t4.x;
                const double d0 = x[0] - t4.y, d1 = x[1] - t5.x, d2 = x[2] - t5.y;
                const double q0 = R00 * d0 + R10 * d1 + R20 * d2;
                const double q1 = R01 * d0 + R11 * d1 + R21 * d2;
                const double q2 = R02 * d0 + R12 * d1 + R22 * d2;
                const double inv = 1.0 / q2;
                const double m0 = q0 * inv, m1 = q1 * inv;
                const double r0 = f * m0 + pcx - uv.x;
                const double r1 = f * m1 + pcy - uv.y;
                esum += r0 * r0 + r1 * r1;
                const double mr = m0 * r0 + m1 * r1;
                psum += mr;
                const double fi = f * inv;
                const double a0 = fi * r0, a1 = fi * r1, a2 = -fi * mr;  // camera-frame force g
                g0 = R00 * a0 + R01 * a1 + R02 * a2;
                g1 = R10 * a0 + R11 * a1 + R12 * a2;
                g2 = R20 * a0 + R21 * a1 + R22 * a2;
                // camera-frame sums: force and torque q x g
                const double w0 = q1 * a2 - q2 * a1, w1 = q2 * a0 - q0 * a2, w2 = q0 * a1 - q1 * a0;
                double *acc = LDS_CAM ? lds_cam : a.cam_sums;
                if (LDS_CAM) {
                    __hip_atomic_fetch_add(acc + 0 * K + j, a0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(acc + 1 * K + j, a1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(acc + 2 * K + j, a2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(acc + 3 * K + j, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(acc + 4 * K + j, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(acc + 5 * K + j, w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else {
                    __hip_atomic_fetch_add(acc + 0 * (int64_t)K + j, a0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(acc + 1 * (int64_t)K + j, a1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(acc + 2 * (int64_t)K + j, a2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(acc + 3 * (int64_t)K + j, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(acc + 4 * (int64_t)K + j, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(acc + 5 * (int64_t)K + j, w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }

            if (a.gX) {
                // segmented inclusive scan over the wave, keyed by point (segments are contiguous)
                const int ip = __shfl_up(i, 1);
                const bool head = (lane == 0) || (ip != i);
                const unsigned long long hm = __ballot(head);
                const int i_first = __shfl(i, 0);
                if (lane == 0) {
                    if (carry_i == i_first) {
                        g0 += c0;
                        g1 += c1;
                        g2 += c2;
                    } else if (carry_i >= 0) {
                        double *o = a.gX + 3 * (int64_t)carry_i;
                        o[0] = c0;
                        o[1] = c1;
                        o[2] = c2;
                    }
                }
                const unsigned long long below = hm & (~0ull >> (63 - lane));
                const int dist = lane - (63 - __clzll(below));
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const double s0 = __shfl_up(g0, off), s1 = __shfl_up(g1, off), s2 = __shfl_up(g2, off);
                    if (dist >= off) {
                        g0 += s0;
                        g1 += s1;
                        g2 += s2;
                    }
                }
                const bool tail = (lane < 63) && ((hm >> (lane + 1)) & 1ull);
                if (tail && i >= 0) {
                    double *o = a.gX + 3 * (int64_t)i;
                    o[0] = g0;
                    o[1] = g1;
                    o[2] = g2;
                }
                carry_i = __shfl(i, 63);
                c0 = __shfl(g0, 63);
                c1 = __shfl(g1, 63);
                c2 = __shfl(g2, 63);
            }
        }
        if (a.gX && lane == 0 && carry_i >= 0) {
            double *o = a.gX + 3 * (int64_t)carry_i;
            o[0] = c0;
            o[1] = c1;
            o[2] = c2;
        }
    }

    esum = wave_sum(esum);
    psum = wave_sum(psum);
    if (lane == 0) {
        a.partials[2 * (int64_t)wave + 0] = esum;
        a.partials[2 * (int64_t)wave + 1] = psum;
    }

    if (LDS_CAM) {
        __syncthreads();
        for (int t = threadIdx.x; t < 6 * K; t += EVAL_BLOCK) {
            const double v = lds_cam[t];
            if (v != 0.0) __hip_atomic_fetch_add(a.cam_sums + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// partial sums -> sums[2]; camera-frame sums -> dE/dC, dE/dTheta (un-normalised)
__global__ void __launch_bounds__(256) k_finalize(int64_t K, int n_partials, const double *__restrict__ partials,
                                                  const double *__restrict__ cam_sums, const double *__restrict__ T,
                                                  const double *__restrict__ scalars, double *__restrict__ E_sum,
                                                  double *__restrict__ gphi, double *__restrict__ gC,
                                                  double *__restrict__ gTheta) {
    if (blockIdx.x == 0) {
        __shared__ double red[2][4];
        double e = 0.0, p = 0.0;
        for (int t = threadIdx.x; t < n_partials; t += 256) {
            e += partials[2 * t + 0];
            p += partials[2 * t + 1];
        }
        e = wave_sum(e);
        p = wave_sum(p);
        if ((threadIdx.x & 63) == 0) {
            red[0][threadIdx.x >> 6] = e;
            red[1][threadIdx.x >> 6] = p;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            E_sum[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
            if (gphi) gphi[0] = scalars[SC_DPHI] * ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
        }
    }
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= K || !gC) return;
    const double *row = T + j * CAM_ROW;
    const double s0 = cam_sums[0 * K + j], s1 = cam_sums[1 * K + j], s2 = cam_sums[2 * K + j];
    const double w0 = cam_sums[3 * K + j], w1 = cam_sums[4 * K + j], w2 = cam_sums[5 * K + j];
    gC[3 * j + 0] = -(row[0] * s0 + row[1] * s1 + row[2] * s2);
    gC[3 * j + 1] = -(row[3] * s0 + row[4] * s1 + row[5] * s2);
    gC[3 * j + 2] = -(row[6] * s0 + row[7] * s1 + row[8] * s2);
    gTheta[3 * j + 0] = -(row[12] * w0 - row[13] * w1);
    gTheta[3 * j + 1] = -(row[3] * w0 + row[4] * w1 + row[5] * w2);
    gTheta[3 * j + 2] = -w2;
}

// --------------------------------------------------------------------------------------------
// K6: update rules (p2w:336-360)
// --------------------------------------------------------------------------------------------
__global__ void k_adam(int64_t n, double *__restrict__ x, const double *__restrict__ g, double *__restrict__ s,
                       double *__restrict__ r, double lr, double gscale, double bc1, double bc2) {
    const double rho1 = 0.9, rho2 = 0.999;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const double gr = g[t] * gscale;
        const double sn = rho1 * s[t] + (1.0 - rho1) * gr;
        const double rn = rho2 * r[t] + (1.0 - rho2) * (gr * gr);
        s[t] = sn;
        r[t] = rn;
        x[t] = x[t] - lr * (sn / bc1) / (1e-12 + sqrt(rn / bc2));
    }
}

__global__ void k_sgd(int64_t n, double *__restrict__ x, const double *__restrict__ g, double lr, double gscale) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
        x[t] = x[t] - lr * (g[t] * gscale);
}

// --------------------------------------------------------------------------------------------
// K7: gauge fix (p2w:270-333)
// --------------------------------------------------------------------------------------------
// single workgroup: cameras.  gauge[0..8] = R0, [9..11] = C0, [12] = scale factor, [13] = j1
__global__ void __launch_bounds__(1024) k_gauge_cameras(int64_t K, double *__restrict__ C, double *__restrict__ Theta,
                                                        double *__restrict__ R_out, double *__restrict__ gauge) {
    __shared__ double sR0[9], sC0[3];
    __shared__ double red_v[16];
    __shared__ long long red_j[16];
    __shared__ double s_factor;
    if (threadIdx.x == 0) {
        double cz, sz;
        rotation_yxz(Theta[0], Theta[1], Theta[2], sR0, &cz, &sz);
        sC0[0] = C[0];
        sC0[1] = C[1];
        sC0[2] = C[2];
    }
    __syncthreads();
    double R0[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) R0[k] = sR0[k];
    const double c00 = sC0[0], c01 = sC0[1], c02 = sC0[2];
    __syncthreads();

    double best = -1.0;
    long long best_j = K;  // larger than any index
    for (int64_t j = threadIdx.x; j < K; j += 1024) {
        const double d0 = C[3 * j + 0] - c00, d1 = C[3 * j + 1] - c01, d2 = C[3 * j + 2] - c02;
        // (C - C0) @ R0
        const double n0 = d0 * R0[0] + d1 * R0[3] + d2 * R0[6];
        const double n1 = d0 * R0[1] + d1 * R0[4] + d2 * R0[7];
        const double n2 = d0 * R0[2] + d1 * R0[5] + d2 * R0[8];
        C[3 * j + 0] = n0;
        C[3 * j + 1] = n1;
        C[3 * j + 2] = n2;
        double R[9], cz, sz;
        rotation_yxz(Theta[3 * j + 0], Theta[3 * j + 1], Theta[3 * j + 2], R, &cz, &sz);
        double Rn[9];  // R0^T R
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) Rn[3 * a + b] = R0[a] * R[b] + R0[3 + a] * R[3 + b] + R0[6 + a] * R[6 + b];
        // get_rotations (p2w:260-266)
        const double Cx = sqrt(Rn[3] * Rn[3] + Rn[4] * Rn[4]);
        const double Sx = -Rn[5];
        Theta[3 * j + 0] = atan2(Sx, Cx);
        Theta[3 * j + 1] = atan2(Rn[2] / Cx, Rn[8] / Cx);
        Theta[3 * j + 2] = atan2(Rn[3] / Cx, Rn[4] / Cx);
        if (R_out) {
#pragma unroll
            for (int k = 0; k < 9; ++k) R_out[9 * j + k] = Rn[k];
        }
        if (j >= 1) {
            const double nn = n0 * n0 + n1 * n1 + n2 * n2;
            if (nn > best) {  // ascending j per thread, strict '>' keeps the first maximum
                best = nn;
                best_j = j;
            }
        }
    }
    // workgroup argmax, ties to the smaller index (numpy.argmax semantics, p2w:322)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(best, off);
        const long long oj = __shfl_xor(best_j, off);
        if (ov > best || (ov == best && oj < best_j)) {
            best = ov;
            best_j = oj;
        }
    }
    if ((threadIdx.x & 63) == 0) {
        red_v[threadIdx.x >> 6] = best;
        red_j[threadIdx.x >> 6] = best_j;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double bv = red_v[0];
        long long bj = red_j[0];
        for (int w = 1; w < 16; ++w) {
            if (red_v[w] > bv || (red_v[w] == bv && red_j[w] < bj)) {
                bv = red_v[w];
                bj = red_j[w];
            }
        }
        const double factor = 1.0 / sqrt(bv);
        s_factor = factor;
        if (gauge) {
#pragma unroll
            for (int k = 0; k < 9; ++k) gauge[k] = R0[k];
            gauge[9] = c00;
            gauge[10] = c01;
            gauge[11] = c02;
            gauge[12] = factor;
            gauge[13] = (double)bj;
            gauge[14] = 0.0;
            gauge[15] = 0.0;
        }
    }
    __syncthreads();
    const double factor = s_factor;
    for (int64_t t = threadIdx.x; t < 3 * K; t += 1024) C[t] *= factor;
}

__global__ void k_gauge_points(int64_t N, double *__restrict__ X, const double *__restrict__ gauge) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const double d0 = X[3 * i + 0] - gauge[9], d1 = X[3 * i + 1] - gauge[10], d2 = X[3 * i + 2] - gauge[11];
    const double factor = gauge[12];
    X[3 * i + 0] = (d0 * gauge[0] + d1 * gauge[3] + d2 * gauge[6]) * factor;
    X[3 * i + 1] = (d0 * gauge[1] + d1 * gauge[4] + d2 * gauge[7]) * factor;
    X[3 * i + 2] = (d0 * gauge[2] + d1 * gauge[5] + d2 * gauge[8]) * factor;
}

// --------------------------------------------------------------------------------------------
// scoring helpers: transform (p2w:192-216) and error (p2w:98-111)
// --------------------------------------------------------------------------------------------
__global__ void k_project(int64_t K, int64_t n, const int32_t *__restrict__ pi, const int32_t *__restrict__ pj,
                          const double *__restrict__ X, const double *__restrict__ T, double w_over_tan, double W,
                          double H, int del_back, int del_boundary, double2 *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int64_t i = pi ? (int64_t)pi[e] : e / K;
    const int64_t j = pi ? (int64_t)pj[e] : e % K;
    const double *t = T + j * CAM_ROW;
    const double d0 = X[3 * i + 0] - t[9], d1 = X[3 * i + 1] - t[10], d2 = X[3 * i + 2] - t[11];
    double q0 = t[0] * d0 + t[3] * d1 + t[6] * d2;
    double q1 = t[1] * d0 + t[4] * d1 + t[7] * d2;
    double q2 = t[2] * d0 + t[5] * d1 + t[8] * d2;
    const double nan = __builtin_nan("");
    if (del_back && q2 <= 0.0) q0 = q1 = q2 = nan;
    double mx = (w_over_tan * q0 / q2 + W) / 2.0;
    double my = (w_over_tan * q1 / q2 + H) / 2.0;
    if (del_boundary && ((mx < 0.0) | (mx >= W) | (my < 0.0) | (my >= H))) mx = my = nan;
    out[e] = make_double2(mx, my);
}

__global__ void __launch_bounds__(256) k_reproj_error(int64_t n, const int2 *__restrict__ pix,
                                                      const double2 *__restrict__ model, double *__restrict__ sums) {
    double s = 0.0, c = 0.0;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int2 p = pix[e];
        const double2 m = model[e];
        if (p.x >= 0 && !isnan(m.x)) {
            const double a = m.x - ((double)p.x + 0.5), b = m.y - ((double)p.y + 0.5);
            s += a * a + b * b;
            c += 1.0;
        }
    }
    s = wave_sum(s);
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) {
        __hip_atomic_fetch_add(sums + 0, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(sums + 1, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

static int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace ba

using namespace ba;

// ============================================================================================
// C ABI
// ============================================================================================
extern "C" {

const char *ba_last_error(void) { return g_error.c_str(); }

int ba_version(void) { return 100; }

int ba_device_info(int device, int *compute_units, int *lds_bytes) {
    int cu = 0, lds = 0;
    BA_HIP(hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device));
    BA_HIP(hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device));
    if (compute_units) *compute_units = cu;
    if (lds_bytes) *lds_bytes = lds;
    return BA_OK;
}

int ba_visibility_count(const int32_t *X_pix, int64_t N, int64_t K, int64_t *row_ptr, int64_t *h_nV, void *stream) {
    BA_REQUIRE(X_pix && row_ptr && h_nV, "ba_visibility_count: null pointer");
    BA_REQUIRE(N >= 0 && K >= 0, "ba_visibility_count: negative shape");
    hipStream_t st = (hipStream_t)stream;
    if (N == 0) {
        BA_HIP(hipMemsetAsync(row_ptr, 0, sizeof(int64_t), st));
        *h_nV = 0;
        return BA_OK;
    }
    hipLaunchKernelGGL(k_vis_count, dim3(cdiv(N, 4)), dim3(256), 0, st, (const int2 *)X_pix, N, K, row_ptr);
    const int n_tiles = cdiv(N, SCAN_TILE);
    int64_t *tile_sums = nullptr;
    BA_HIP(hipMalloc(&tile_sums, sizeof(int64_t) * (n_tiles + 1)));
    // counts were written into row_ptr[0..N); scan them in place
    hipLaunchKernelGGL(k_scan_tiles, dim3(n_tiles), dim3(1024), 0, st, row_ptr, N, row_ptr, tile_sums);
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(1), dim3(1024), 0, st, tile_sums, n_tiles);
    hipLaunchKernelGGL(k_scan_add, dim3(cdiv(N, 256)), dim3(256), 0, st, row_ptr, N, tile_sums, n_tiles);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h_nV, row_ptr + N, sizeof(int64_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    hipFree(tile_sums);
    BA_HIP(e);
    return BA_OK;
}

int ba_visibility_fill(const int32_t *X_pix, int64_t N, int64_t K, const int64_t *row_ptr, int32_t *i, int32_t *j,
                       double *uv, void *stream) {
    BA_REQUIRE(X_pix && row_ptr, "ba_visibility_fill: null pointer");
    if (N == 0) return BA_OK;
    hipLaunchKernelGGL(k_vis_fill, dim3(cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, (const int2 *)X_pix, N, K,
                       row_ptr, i, j, (double2 *)uv);
    BA_HIP(hipGetLastError());
    return BA_OK;
}

int ba_problem_destroy(ba_problem *p) {
    if (!p) return BA_OK;
    hipSetDevice(p->device);
    hipFree(p->oi);
    hipFree(p->oj);
    hipFree(p->ouv);
    hipFree(p->chunk_begin);
    hipFree(p->cam_table);
    hipFree(p->cam_sums);
    hipFree(p->partials);
    hipFree(p->scalars);
    hipFree(p->flag);
    for (auto e : p->ev_begin) hipEventDestroy(e);
    for (auto e : p->ev_end) hipEventDestroy(e);
    delete p;
    return BA_OK;
}

int ba_problem_create(int64_t N, int64_t K, int64_t nV, const int32_t *i, const int32_t *j, const double *uv, double W,
                      double H, int device, ba_problem **out) {
    BA_REQUIRE(out, "ba_problem_create: null out pointer");
    *out = nullptr;
    BA_REQUIRE(N > 0 && K > 0 && nV > 0, "ba_problem_create: N, K and nV must be positive");
    BA_REQUIRE(nV < (1ll << 31) - 64 && N < (1ll << 31) && K < (1ll << 31) / 8,
               "ba_problem_create: shard too large for 32-bit indices");
    BA_REQUIRE(i && j && uv, "ba_problem_create: null observation array");
    BA_HIP(hipSetDevice(device));
    ba_problem *p = new ba_problem();
    p->device = device;
    p->N = N;
    p->K = K;
    p->nV = nV;
    p->W = W;
    p->H = H;
    int rc = BA_OK;
    auto fail = [&](int code) {
        ba_problem_destroy(p);
        return code;
    };
#define BA_TRY(expr)                                                              \
    do {                                                                          \
        hipError_t _e = (expr);                                                   \
        if (_e != hipSuccess) return fail(hip_fail(_e, #expr, __FILE__, __LINE__)); \
    } while (0)
    int lds = 0;
    BA_TRY(hipDeviceGetAttribute(&p->compute_units, hipDeviceAttributeMultiprocessorCount, device));
    BA_TRY(hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device));
    p->lds_limit = lds;

    BA_TRY(hipMalloc(&p->oi, sizeof(int32_t) * nV));
    BA_TRY(hipMalloc(&p->oj, sizeof(int32_t) * nV));
    BA_TRY(hipMalloc(&p->ouv, sizeof(double2) * nV));
    BA_TRY(hipMemcpy(p->oi, i, sizeof(int32_t) * nV, hipMemcpyDeviceToDevice));
    BA_TRY(hipMemcpy(p->oj, j, sizeof(int32_t) * nV, hipMemcpyDeviceToDevice));
    BA_TRY(hipMemcpy(p->ouv, uv, sizeof(double2) * nV, hipMemcpyDeviceToDevice));

    // validation + number of distinct points
    BA_TRY(hipMalloc(&p->flag, 16));
    BA_TRY(hipMemset(p->flag, 0, 16));
    hipLaunchKernelGGL(k_validate_obs, dim3(cdiv(nV, 256)), dim3(256), 0, 0, p->oi, p->oj, nV, N, K, p->flag,
                       (unsigned long long *)(p->flag + 2));
    BA_TRY(hipGetLastError());
    struct {
        int flag, pad;
        unsigned long long heads;
    } h;
    BA_TRY(hipMemcpy(&h, p->flag, 16, hipMemcpyDeviceToHost));
    if (h.flag & 1) {
        set_error("ba_problem_create: observation index out of range");
        return fail(BA_ERR_ARG);
    }
    if (h.flag & 2) {
        set_error("ba_problem_create: observations must be sorted by point index");
        return fail(BA_ERR_ARG);
    }
    p->n_points_seen = (int64_t)h.heads;

    // launch geometry: one workgroup of 16 waves per compute unit, every wave walks whole chunks
    // of ~768 observations whose boundaries fall on point boundaries
    const int cu = p->compute_units > 0 ? p->compute_units : 256;
    const int64_t waves_full = (int64_t)cu * EVAL_WAVES;
    int64_t n_chunks;
    if (nV <= waves_full * 64) {
        n_chunks = (nV + 63) / 64;
    } else {
        const int64_t per_wave = (nV + waves_full * 768 - 1) / (waves_full * 768);
        n_chunks = waves_full * per_wave;
    }
    const int64_t chunk = (nV + n_chunks - 1) / n_chunks;
    p->n_chunks = (int)n_chunks;
    p->grid = (int)((n_chunks + EVAL_WAVES - 1) / EVAL_WAVES);
    if (p->grid > cu) p->grid = cu;
    BA_TRY(hipMalloc(&p->chunk_begin, sizeof(int32_t) * (n_chunks + 1)));
    hipLaunchKernelGGL(k_chunk_bounds, dim3(cdiv(n_chunks + 1, 256)), dim3(256), 0, 0, p->oi, nV, p->n_chunks, chunk,
                       p->chunk_begin);
    BA_TRY(hipGetLastError());

    BA_TRY(hipMalloc(&p->cam_table, sizeof(double) * K * CAM_ROW));
    BA_TRY(hipMalloc(&p->cam_sums, sizeof(double) * K * 6));
    BA_TRY(hipMalloc(&p->partials, sizeof(double) * 2 * (size_t)p->grid * EVAL_WAVES));
    BA_TRY(hipMalloc(&p->scalars, sizeof(double) * SC_COUNT));
    p->ev_begin.resize(TIMING_RING);
    p->ev_end.resize(TIMING_RING);
    for (int t = 0; t < TIMING_RING; ++t) {
        p->ev_begin[t] = nullptr;
        p->ev_end[t] = nullptr;
    }
    for (int t = 0; t < TIMING_RING; ++t) {
        BA_TRY(hipEventCreate(&p->ev_begin[t]));
        BA_TRY(hipEventCreate(&p->ev_end[t]));
    }
    BA_TRY(hipDeviceSynchronize());
#undef BA_TRY
    (void)rc;
    *out = p;
    return BA_OK;
}

int ba_eval(ba_problem *p, const double *X, const double *C, const double *Theta, const double *phi, int flags,
            double *E_sum, double *gX, double *gC, double *gTheta, double *gphi, double *V, double *B, double *bphi,
            double *hphi, void *stream) {
    BA_REQUIRE(p, "ba_eval: null problem");
    BA_REQUIRE(X && C && Theta && phi && E_sum, "ba_eval: null parameter or E_sum pointer");
    BA_REQUIRE(!(flags & BA_EVAL_GRAD_PHI) || gphi, "ba_eval: BA_EVAL_GRAD_PHI needs gphi");
    BA_REQUIRE(!(flags & BA_EVAL_GRAD_X) || gX, "ba_eval: BA_EVAL_GRAD_X needs gX");
    BA_REQUIRE(!(flags & BA_EVAL_GRAD_CAM) || (gC && gTheta), "ba_eval: BA_EVAL_GRAD_CAM needs gC and gTheta");
    BA_REQUIRE(!(flags & BA_EVAL_JTJ), "ba_eval: BA_EVAL_JTJ is not available in this build");
    (void)V;
    (void)B;
    (void)bphi;
    (void)hphi;
    hipStream_t st = (hipStream_t)stream;
    const int64_t K = p->K;

    hipLaunchKernelGGL(k_camera_table, dim3(cdiv(K, 256)), dim3(256), 0, st, K, C, Theta, phi, p->W, p->H, p->cam_table,
                       p->scalars);
    BA_HIP(hipMemsetAsync(p->cam_sums, 0, sizeof(double) * 6 * K, st));
    const bool want_gx = (flags & BA_EVAL_GRAD_X) != 0;
    if (want_gx && p->n_points_seen < p->N) BA_HIP(hipMemsetAsync(gX, 0, sizeof(double) * 3 * p->N, st));

    EvalArgs a;
    a.oi = p->oi;
    a.oj = p->oj;
    a.ouv = p->ouv;
    a.chunk_begin = p->chunk_begin;
    a.n_chunks = p->n_chunks;
    a.K = (int)K;
    a.X = X;
    a.T = p->cam_table;
    a.scalars = p->scalars;
    a.gX = want_gx ? gX : nullptr;
    a.cam_sums = p->cam_sums;
    a.partials = p->partials;

    const size_t lds_bytes = sizeof(double) * 6 * (size_t)K;
    const bool use_lds = !(flags & BA_EVAL_CAM_ATOMICS) && lds_bytes <= (size_t)p->lds_limit;
    const bool timed = (flags & BA_EVAL_TIMED) != 0 && p->ev_count < TIMING_RING;
    if (timed) BA_HIP(hipEventRecord(p->ev_begin[p->ev_count], st));
    if (use_lds) {
        if (lds_bytes > 64 * 1024)
            BA_HIP(hipFuncSetAttribute((const void *)k_eval<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds_bytes));
        hipLaunchKernelGGL(k_eval<true>, dim3(p->grid), dim3(EVAL_BLOCK), lds_bytes, st, a);
    } else {
        hipLaunchKernelGGL(k_eval<false>, dim3(p->grid), dim3(EVAL_BLOCK), 0, st, a);
    }
    BA_HIP(hipGetLastError());
    if (timed) {
        BA_HIP(hipEventRecord(p->ev_end[p->ev_count], st));
        p->ev_count++;
    }
    const bool want_cam = (flags & BA_EVAL_GRAD_CAM) != 0;
    hipLaunchKernelGGL(k_finalize, dim3(want_cam ? cdiv(K, 256) : 1), dim3(256), 0, st, K, p->grid * EVAL_WAVES,
                       p->partials, p->cam_sums, p->cam_table, p->scalars, E_sum,
                       (flags & BA_EVAL_GRAD_PHI) ? gphi : nullptr, want_cam ? gC : nullptr,
                       want_cam ? gTheta : nullptr);
    BA_HIP(hipGetLastError());
    return BA_OK;
}

int ba_eval_timing(ba_problem *p, float *h_ms, int capacity, int *h_count) {
    BA_REQUIRE(p && h_count, "ba_eval_timing: null pointer");
    int n = p->ev_count < capacity ? p->ev_count : capacity;
    for (int t = 0; t < n; ++t) {
        BA_HIP(hipEventSynchronize(p->ev_end[t]));
        BA_HIP(hipEventElapsedTime(&h_ms[t], p->ev_begin[t], p->ev_end[t]));
    }
    *h_count = n;
    p->ev_count = 0;
    return BA_OK;
}

int ba_adam_step(int64_t n, double *x, const double *g, double *s, double *r, double lr, double gscale, double bc1,
                 double bc2, void *stream) {
    BA_REQUIRE(n >= 0 && (n == 0 || (x && g && s && r)), "ba_adam_step: null pointer");
    if (n == 0) return BA_OK;
    int grid = cdiv(n, 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(k_adam, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, x, g, s, r, lr, gscale, bc1, bc2);
    BA_HIP(hipGetLastError());
    return BA_OK;
}

int ba_sgd_step(int64_t n, double *x, const double *g, double lr, double gscale, void *stream) {
    BA_REQUIRE(n >= 0 && (n == 0 || (x && g)), "ba_sgd_step: null pointer");
    if (n == 0) return BA_OK;
    int grid = cdiv(n, 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(k_sgd, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, x, g, lr, gscale);
    BA_HIP(hipGetLastError());
    return BA_OK;
}

int ba_gauge_fix(int64_t N, int64_t K, double *X, double *C, double *Theta, double *R_out, double *gauge_out,
                 void *stream) {
    BA_REQUIRE(K >= 2, "ba_gauge_fix: needs at least two cameras");
    BA_REQUIRE(C && Theta && gauge_out && (N == 0 || X), "ba_gauge_fix: null pointer");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_gauge_cameras, dim3(1), dim3(1024), 0, st, K, C, Theta, R_out, gauge_out);
    if (N > 0) hipLaunchKernelGGL(k_gauge_points, dim3(cdiv(N, 256)), dim3(256), 0, st, N, X, gauge_out);
    BA_HIP(hipGetLastError());
    return BA_OK;
}

int ba_project(int64_t N, int64_t K, int64_t n, const int32_t *i, const int32_t *j, const double *X, const double *C,
               const double *Theta, double phi, double W, double H, int delete_back_points, int delete_boundary_points,
               double *out, void *stream) {
    BA_REQUIRE(X && C && Theta && out, "ba_project: null pointer");
    BA_REQUIRE((i == nullptr) == (j == nullptr), "ba_project: i and j must both be given or both be null");
    BA_REQUIRE(i || n == N * K, "ba_project: dense form needs n == N*K");
    if (n == 0) return BA_OK;
    hipStream_t st = (hipStream_t)stream;
    double *T = nullptr;
    BA_HIP(hipMalloc(&T, sizeof(double) * K * CAM_ROW));
    hipLaunchKernelGGL(k_camera_table, dim3(cdiv(K, 256)), dim3(256), 0, st, K, C, Theta, (const double *)nullptr, W, H,
                       T, (double *)nullptr);
    const double w_over_tan = W / tan(phi / 2.0);
    hipLaunchKernelGGL(k_project, dim3(cdiv(n, 256)), dim3(256), 0, st, K, n, i, j, X, T, w_over_tan, W, H,
                       delete_back_points, delete_boundary_points, (double2 *)out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    hipFree(T);
    BA_HIP(e);
    return BA_OK;
}

int ba_reprojection_error(int64_t n, const int32_t *pix_xy, const double *model, double *sums, void *stream) {
    BA_REQUIRE(pix_xy && model && sums, "ba_reprojection_error: null pointer");
    hipStream_t st = (hipStream_t)stream;
    BA_HIP(hipMemsetAsync(sums, 0, 2 * sizeof(double), st));
    if (n == 0) return BA_OK;
    int grid = cdiv(n, 256 * 8);
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_reproj_error, dim3(grid), dim3(256), 0, st, n, (const int2 *)pix_xy, (const double2 *)model,
                       sums);
    BA_HIP(hipGetLastError());
    return BA_OK;
}

}  // extern "C"
