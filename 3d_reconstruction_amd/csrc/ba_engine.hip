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

#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
#include "ba_device.h"

#include <hip/hip_ext.h>
#include "ba_runpass.h"
#include "ba_blockpass.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
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
// flag bit 0: index out of range, bit 1: not sorted by point, bit 2: some pixel centre is not
// (integer in (-32768, 32767]) + 0.5
// (one atomic per workgroup on the flag word and on the counter: the memory side serves the atomics of one address one
// after the other, and one per wave of a launch over ten million observations took 1.9 ms)
__global__ void __launch_bounds__(256) k_validate_obs(const int32_t *__restrict__ oi, const int32_t *__restrict__ oj,
                                                      const double2 *__restrict__ ouv, int64_t nV, int64_t N, int64_t K,
                                                      int *flag, unsigned long long *heads) {
    __shared__ unsigned s_heads[4];
    __shared__ int s_bits[4];
    int bits = 0;
    unsigned n_heads = 0;       // (kept by the wave's first lane)
    for (int64_t base = (int64_t)blockIdx.x * 256; base < nV; base += (int64_t)gridDim.x * 256) {
        const int64_t e = base + threadIdx.x;
        bool head = false;
        if (e < nV) {
            const int i = oi[e], j = oj[e];
            if (i < 0 || i >= N || j < 0 || j >= K) bits |= 1;
            const double2 uv = ouv[e];
            const double px = uv.x - 0.5, py = uv.y - 0.5;
            const bool fits = px == floor(px) && py == floor(py) && px > -32768.0 && px <= 32767.0 && py > -32768.0 &&
                              py <= 32767.0;
            if (!fits) bits |= 4;
            if (e > 0) {
                const int ip = oi[e - 1];
                if (ip > i) bits |= 2;
                head = ip != i;
            } else {
                head = true;
            }
        }
        n_heads += (unsigned)__popcll(__ballot(head));
    }
    const int wave_bits = (__ballot(bits & 1) ? 1 : 0) | (__ballot(bits & 2) ? 2 : 0) | (__ballot(bits & 4) ? 4 : 0);
    if ((threadIdx.x & 63) == 0) {
        s_heads[threadIdx.x >> 6] = n_heads;
        s_bits[threadIdx.x >> 6] = wave_bits;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int all = s_bits[0] | s_bits[1] | s_bits[2] | s_bits[3];
        if (all) atomicOr(flag, all);
        atomicAdd(heads, (unsigned long long)s_heads[0] + s_heads[1] + s_heads[2] + s_heads[3]);
    }
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

// Tile bounds from a list of nominal offsets: every bound moved forward to the next point's first observation.
__global__ void k_tile_bounds(const int32_t *__restrict__ oi, int64_t nV, int n_tiles, const int64_t *__restrict__ nominal,
                              int32_t *__restrict__ tile_begin) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n_tiles) return;
    int64_t t = c < n_tiles ? nominal[c] : nV;
    if (c == n_tiles || t >= nV) {
        tile_begin[c] = (int32_t)nV;
        return;
    }
    if (t > 0) {
        while (t < nV && oi[t] == oi[t - 1]) ++t;
    }
    tile_begin[c] = (int32_t)t;
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
    int debug;         // diagnostics: bit 0 skips the camera-side accumulation
};

// One observation record and the operands gathered for it.
struct ObsRec {
    int i, j;
    double2 uv;
};
struct ObsOps {
    double x0, x1, x2;            // X_i
    double2 t0, t1, t2, t3, t4, t5;  // camera-table row: R (9), C (3)
};

// Lanes past the end of the chunk read the chunk's last record (always a valid address) and
// are masked out of every sum.
__device__ __forceinline__ ObsRec load_record(const EvalArgs &a, int e, int e_last) {
    const int ec = e < e_last ? e : e_last;
    ObsRec r;
    r.i = a.oi[ec];
    r.j = a.oj[ec];
    r.uv = a.ouv[ec];
    return r;
}

__device__ __forceinline__ ObsOps gather_operands(const EvalArgs &a, const ObsRec &r) {
    ObsOps g;
    const bool fixed = (a.debug & 4) != 0;   // diagnostics: every lane reads point 0 / camera 0
    const double *x = a.X + 3 * (int64_t)(fixed ? 0 : r.i);
    const double2 *t = (const double2 *)(a.T + (int64_t)(fixed ? 0 : r.j) * CAM_ROW);
    g.x0 = x[0];
    g.x1 = x[1];
    g.x2 = x[2];
    g.t0 = t[0];
    g.t1 = t[1];
    g.t2 = t[2];
    g.t3 = t[3];
    g.t4 = t[4];
    g.t5 = t[5];
    return g;
}

// The observation stream is software-pipelined three deep per wave: while step k is computed,
// the gathers (X_i, camera row) of step k+1 and the record loads of step k+2 are in flight, so
// that a wave never sits on a dependent load chain (record -> gather -> math).
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
        if (e0 >= e1) continue;
        const int e_last = e1 - 1;
        int carry_i = -1;
        double c0 = 0.0, c1 = 0.0, c2 = 0.0;

        ObsRec rec0 = load_record(a, e0 + lane, e_last);
        ObsRec rec1 = load_record(a, e0 + 64 + lane, e_last);
        ObsOps ops0 = gather_operands(a, rec0);

        for (int base = e0; base < e1; base += 64) {
            const ObsRec rec2 = load_record(a, base + 128 + lane, e_last);
            const ObsOps ops1 = gather_operands(a, rec1);

            const bool valid = base + lane < e1;
            const int i = valid ? rec0.i : -2;
            const int j = rec0.j;
            const double R00 = ops0.t0.x, R01 = ops0.t0.y, R02 = ops0.t1.x, R10 = ops0.t1.y, R11 = ops0.t2.x,
                         R12 = ops0.t2.y, R20 = ops0.t3.x, R21 = ops0.t3.y, R22 = ops0.t4.x;
            const double d0 = ops0.x0 - ops0.t4.y, d1 = ops0.x1 - ops0.t5.x, d2 = ops0.x2 - ops0.t5.y;
            const double q0 = R00 * d0 + R10 * d1 + R20 * d2;
            const double q1 = R01 * d0 + R11 * d1 + R21 * d2;
            const double q2 = R02 * d0 + R12 * d1 + R22 * d2;
            const double inv = 1.0 / q2;
            const double m0 = q0 * inv, m1 = q1 * inv;
            const double r0 = valid ? f * m0 + pcx - rec0.uv.x : 0.0;
            const double r1 = valid ? f * m1 + pcy - rec0.uv.y : 0.0;
            esum += r0 * r0 + r1 * r1;
            const double mr = m0 * r0 + m1 * r1;
            psum += mr;
            const double fi = f * inv;
            const double a0 = fi * r0, a1 = fi * r1, a2 = -fi * mr;  // camera-frame force g
            double g0 = R00 * a0 + R01 * a1 + R02 * a2;            // world-frame dE/dX_i part
            double g1 = R10 * a0 + R11 * a1 + R12 * a2;
            double g2 = R20 * a0 + R21 * a1 + R22 * a2;
            // camera-frame sums: force and torque q x g
            const double w0 = q1 * a2 - q2 * a1, w1 = q2 * a0 - q0 * a2, w2 = q0 * a1 - q1 * a0;
            if (a.debug & 1) {
                asm volatile("" ::"v"(w0), "v"(w1), "v"(w2));
            } else if (valid) {
                if (LDS_CAM) {
                    double *acc = lds_cam + j;
                    __hip_atomic_fetch_add(acc + 0 * K, a0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(acc + 1 * K, a1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(acc + 2 * K, a2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(acc + 3 * K, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(acc + 4 * K, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(acc + 5 * K, w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else {
                    double *acc = a.cam_sums + j;
                    const int64_t Kl = K;
                    __hip_atomic_fetch_add(acc + 0 * Kl, a0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(acc + 1 * Kl, a1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(acc + 2 * Kl, a2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(acc + 3 * Kl, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(acc + 4 * Kl, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(acc + 5 * Kl, w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

            rec0 = rec1;
            rec1 = rec2;
            ops0 = ops1;
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

// --------------------------------------------------------------------------------------------
// K2 (tiled form): the observation list is cut into tiles of a few hundred observations whose
// boundaries fall on point boundaries.  A tile touches a contiguous range of points and a short
// list of distinct cameras (a video track is seen by consecutive frames), so one workgroup keeps
// everything a tile needs in LDS: the camera rows, X of its points, and the accumulators of both
// sides.  The per-observation operand fetch is then LDS reads instead of scattered global loads
// (which were bound by the vector-L1 rate), and both reductions are LDS atomics.
// Observation records carry the camera's slot in the tile's list instead of its index.
// --------------------------------------------------------------------------------------------
struct TileArgs {
    const uint32_t *opack;       // per observation: (point index - tile's first point) << 16 | camera slot
    const double2 *ouv;
    const TileMeta *meta;        // n_tiles
    const int32_t *cam_list;     // distinct cameras of every tile, ascending
    int n_tiles;
    int K;
    int cam_cap;                 // LDS capacity in cameras / points (max over tiles)
    int pt_cap;
    int cam_rep;                 // copies of the camera accumulators (power of two <= 16)
    const double *X;
    const double *T;
    const double *scalars;
    double *gX;        // may be null
    double *cam_sums;  // 6 x K, component-major (pre-zeroed)
    double *partials;  // per-tile {sum r^2, sum m.r}
    int debug;
};

// A tiled record is 20 bytes: the point (relative to the tile) and the camera slot share one
// 32-bit word (both are below 65536 inside a tile), the pixel centre is a double2.
struct TileRec {
    uint32_t pack;
    double2 uv;
};
// lanes past the end of the tile read its last record (a valid address) and are masked later
__device__ __forceinline__ TileRec load_tile_record(const TileArgs &a, int e, int e_last) {
    const int ec = e < e_last ? e : e_last;
    TileRec r;
    r.pack = a.opack[ec];
    r.uv = a.ouv[ec];
    return r;
}

__global__ void __launch_bounds__(TILE_BLOCK) k_eval_tiled(TileArgs a) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int t = blockIdx.x;
    const TileMeta tm = a.meta[t];
    const int e0 = tm.e0, e1 = tm.e1, p0 = tm.p0, np = tm.np, nc = tm.nc;
    const int64_t cb = tm.cb;
    __shared__ double red[2][TILE_WAVES];
    if (e0 >= e1) {
        if (tid == 0) {
            a.partials[2 * (int64_t)t + 0] = 0.0;
            a.partials[2 * (int64_t)t + 1] = 0.0;
        }
        return;
    }
    // records of this wave's first two steps: issued before anything else so that they travel
    // together with the tile's camera rows and points
    const int e_last = e1 - 1;
    int base = e0 + wid * 64;
    TileRec recA = load_tile_record(a, base + lane, e_last);
    TileRec recB = load_tile_record(a, base + TILE_BLOCK + lane, e_last);

    // Same-address LDS atomics serialise (about one conflict-free instruction per colliding lane),
    // and lanes of one wave share cameras (track after track covers the same frames).  The camera
    // accumulators are therefore kept in cam_rep copies selected by the low lane bits, which
    // separates the colliding lanes; the copies are added up when the tile is flushed.
    const int rep = a.cam_rep;
    double *rows = lds;                                        // cam_cap x ROW_LDS
    double *cacc = rows + (size_t)a.cam_cap * ROW_LDS;         // rep x 6 x nc (component-major)
    double *xs = cacc + (size_t)a.cam_cap * 6 * rep;           // np x 3
    double *pacc = xs + (size_t)a.pt_cap * 3;                  // np x 3

    for (int idx = tid; idx < nc * (ROW_LDS / 2); idx += TILE_BLOCK) {
        const int row = idx / (ROW_LDS / 2), piece = idx - row * (ROW_LDS / 2);
        const int j = tm.j0 >= 0 ? tm.j0 + row : a.cam_list[cb + row];
        ((double2 *)rows)[idx] = ((const double2 *)(a.T + (int64_t)j * CAM_ROW))[piece];
    }
    for (int idx = tid; idx < 6 * nc * rep; idx += TILE_BLOCK) cacc[idx] = 0.0;
    for (int idx = tid; idx < 3 * np; idx += TILE_BLOCK) {
        xs[idx] = a.X[3 * (int64_t)p0 + idx];
        pacc[idx] = 0.0;
    }
    double *cacc_lane = cacc + (size_t)(lane & (rep - 1)) * 6 * nc;
    const double f = a.scalars[SC_F], pcx = a.scalars[SC_CX], pcy = a.scalars[SC_CY];
    __syncthreads();

    double esum = 0.0, psum = 0.0;
    for (; base < e1; base += TILE_BLOCK) {
        const TileRec recC = load_tile_record(a, base + 2 * TILE_BLOCK + lane, e_last);
        const bool valid = base + lane < e1;
        const int li = valid ? (int)(recA.pack >> 16) : -2;
        const int slot = (int)(recA.pack & 0xffffu);
        const double2 uv = recA.uv;
        recA = recB;
        recB = recC;
        const double2 *rw = (const double2 *)(rows + slot * ROW_LDS);
        const double2 t0 = rw[0], t1 = rw[1], t2 = rw[2], t3 = rw[3], t4 = rw[4], t5 = rw[5];
        const double *x = xs + 3 * (valid ? li : 0);
        const double R00 = t0.x, R01 = t0.y, R02 = t1.x, R10 = t1.y, R11 = t2.x, R12 = t2.y, R20 = t3.x,
                     R21 = t3.y, R22 = t4.x;
        const double d0 = x[0] - t4.y, d1 = x[1] - t5.x, d2 = x[2] - t5.y;
        const double q0 = R00 * d0 + R10 * d1 + R20 * d2;
        const double q1 = R01 * d0 + R11 * d1 + R21 * d2;
        const double q2 = R02 * d0 + R12 * d1 + R22 * d2;
        const double inv = 1.0 / q2;
        const double m0 = q0 * inv, m1 = q1 * inv;
        const double r0 = valid ? f * m0 + pcx - uv.x : 0.0;
        const double r1 = valid ? f * m1 + pcy - uv.y : 0.0;
        esum += r0 * r0 + r1 * r1;
        const double mr = m0 * r0 + m1 * r1;
        psum += mr;
        const double fi = f * inv;
        const double a0 = fi * r0, a1 = fi * r1, a2 = -fi * mr;  // camera-frame force g
        double g0 = R00 * a0 + R01 * a1 + R02 * a2;            // world-frame dE/dX_i part
        double g1 = R10 * a0 + R11 * a1 + R12 * a2;
        double g2 = R20 * a0 + R21 * a1 + R22 * a2;
        const double w0 = q1 * a2 - q2 * a1, w1 = q2 * a0 - q0 * a2, w2 = q0 * a1 - q1 * a0;  // torque q x g
        if (a.debug & 1) {
            asm volatile("" ::"v"(w0), "v"(w1), "v"(w2));
        } else if (valid) {
            double *acc = cacc_lane + slot;
            __hip_atomic_fetch_add(acc + 0 * nc, a0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(acc + 1 * nc, a1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(acc + 2 * nc, a2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(acc + 3 * nc, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(acc + 4 * nc, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(acc + 5 * nc, w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (a.debug & 2) {
            asm volatile("" ::"v"(g0), "v"(g1), "v"(g2));
        } else {
            // Point side: observations of one point are adjacent lanes.  Segmented inclusive scan
            // inside every 16-lane row (DPP row shifts, no LDS traffic); the last lane of each run
            // then holds the run's sum and adds it to the tile's point accumulator.  A run cut by a
            // row or wave boundary simply contributes two partial sums.
            const int lane16 = lane & 15;
            const int ip = row_shr<1>(li);
            const bool head = (lane16 == 0) || (ip != li);
            const unsigned long long hm = __ballot(head);
            const unsigned long long below = hm & (~0ull >> (63 - lane));
            const int dist = lane - (63 - __clzll(below));
            {
                const double s0 = row_shr<1>(g0), s1 = row_shr<1>(g1), s2 = row_shr<1>(g2);
                if (dist >= 1) { g0 += s0; g1 += s1; g2 += s2; }
            }
            {
                const double s0 = row_shr<2>(g0), s1 = row_shr<2>(g1), s2 = row_shr<2>(g2);
                if (dist >= 2) { g0 += s0; g1 += s1; g2 += s2; }
            }
            {
                const double s0 = row_shr<4>(g0), s1 = row_shr<4>(g1), s2 = row_shr<4>(g2);
                if (dist >= 4) { g0 += s0; g1 += s1; g2 += s2; }
            }
            {
                const double s0 = row_shr<8>(g0), s1 = row_shr<8>(g1), s2 = row_shr<8>(g2);
                if (dist >= 8) { g0 += s0; g1 += s1; g2 += s2; }
            }
            const bool tail = (lane == 63) || ((hm >> (lane + 1)) & 1ull);
            if (tail && li >= 0) {
                double *pa = pacc + 3 * li;
                __hip_atomic_fetch_add(pa + 0, g0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(pa + 1, g1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(pa + 2, g2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    esum = wave_sum(esum);
    psum = wave_sum(psum);
    if (lane == 0) {
        red[0][wid] = esum;
        red[1][wid] = psum;
    }
    __syncthreads();
    if (tid == 0) {
        a.partials[2 * (int64_t)t + 0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        a.partials[2 * (int64_t)t + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
    if (a.gX) {
        const int lo = 3 * (tm.wlo - p0), hi = 3 * (tm.whi - p0);  // relative to the tile's first point
        double *o = a.gX + 3 * (int64_t)p0;
        for (int idx = lo + tid; idx < hi; idx += TILE_BLOCK) o[idx] = (idx >= 0 && idx < 3 * np) ? pacc[idx] : 0.0;
    }
    for (int idx = tid; idx < 6 * nc; idx += TILE_BLOCK) {
        const int comp = idx / nc, slot = idx - comp * nc;
        const int j = tm.j0 >= 0 ? tm.j0 + slot : a.cam_list[cb + slot];
        double v = cacc[idx];
        for (int r = 1; r < rep; ++r) v += cacc[idx + (size_t)r * 6 * nc];
        __hip_atomic_fetch_add(a.cam_sums + (int64_t)comp * a.K + j, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- tile construction -----------------------------------------------------------------------
// Pass 1: number of distinct cameras and point span of every tile (one workgroup per tile; a
// K-bit set in LDS).  Pass 2, after an exclusive scan of the counts: the ascending camera list of
// every tile and, for every observation, its camera's position in that list.
__device__ __forceinline__ void tile_camera_bitmap(const int32_t *oj, int e0, int e1, int n_words, unsigned *bits) {
    for (int w = threadIdx.x; w < n_words; w += TILE_BLOCK) bits[w] = 0u;
    __syncthreads();
    for (int e = e0 + threadIdx.x; e < e1; e += TILE_BLOCK) {
        const int j = oj[e];
        atomicOr(&bits[j >> 5], 1u << (j & 31));
    }
    __syncthreads();
}

__global__ void __launch_bounds__(TILE_BLOCK) k_tile_count(const int32_t *__restrict__ oi, const int32_t *__restrict__ oj,
                                                           const int32_t *__restrict__ tile_obs, int K,
                                                           int64_t *__restrict__ counts, int *__restrict__ maxima) {
    extern __shared__ unsigned bits[];
    __shared__ int wsum[TILE_WAVES];
    const int t = blockIdx.x;
    const int e0 = tile_obs[t], e1 = tile_obs[t + 1];
    const int n_words = (K + 31) >> 5;
    tile_camera_bitmap(oj, e0, e1, n_words, bits);
    int c = 0;
    for (int w = threadIdx.x; w < n_words; w += TILE_BLOCK) c += __popc(bits[w]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        counts[t] = total;
        atomicMax(&maxima[0], total);
        if (e1 > e0) atomicMax(&maxima[1], oi[e1 - 1] - oi[e0] + 1);
    }
}

__global__ void __launch_bounds__(TILE_BLOCK) k_tile_fill(const int32_t *__restrict__ oj, const int32_t *__restrict__ tile_obs,
                                                          const int64_t *__restrict__ tile_cam_ptr, int K,
                                                          const int32_t *__restrict__ oi, int64_t nV, int64_t N,
                                                          int32_t *__restrict__ cam_list, uint32_t *__restrict__ opack,
                                                          TileMeta *__restrict__ meta) {
    extern __shared__ unsigned bits[];  // n_words bit set, then n_words exclusive prefix counts
    __shared__ int wtot[TILE_WAVES];
    __shared__ int carry;
    const int t = blockIdx.x;
    const int e0 = tile_obs[t], e1 = tile_obs[t + 1];
    const int n_words = (K + 31) >> 5;
    unsigned *prefix = bits + n_words;
    tile_camera_bitmap(oj, e0, e1, n_words, bits);
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int base = 0; base < n_words; base += TILE_BLOCK) {
        const int w = base + threadIdx.x;
        const int c = (w < n_words) ? __popc(bits[w]) : 0;
        int inc = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(inc, off);
            if (lane >= off) inc += v;
        }
        if (lane == 63) wtot[wid] = inc;
        __syncthreads();
        int before = carry;
        for (int k = 0; k < wid; ++k) before += wtot[k];
        if (w < n_words) prefix[w] = (unsigned)(before + inc - c);
        __syncthreads();
        if (threadIdx.x == 0) carry += wtot[0] + wtot[1] + wtot[2] + wtot[3];
        __syncthreads();
    }
    const int64_t cb = tile_cam_ptr[t];
    if (threadIdx.x == 0) {
        // lowest / highest camera of the tile from the bit set
        const int nc = (int)(tile_cam_ptr[t + 1] - cb);
        int lo = -1, hi = -1;
        for (int w = 0; w < n_words && lo < 0; ++w)
            if (bits[w]) lo = (w << 5) + __ffs(bits[w]) - 1;
        for (int w = n_words - 1; w >= 0 && hi < 0; --w)
            if (bits[w]) hi = (w << 5) + 31 - __clz(bits[w]);
        TileMeta m;
        m.e0 = e0;
        m.e1 = e1;
        m.p0 = e1 > e0 ? oi[e0] : 0;
        m.np = e1 > e0 ? oi[e1 - 1] - m.p0 + 1 : 0;
        m.nc = nc;
        m.j0 = (nc > 0 && hi - lo + 1 == nc) ? lo : -1;
        m.cb = cb;
        m.wlo = e0 == 0 ? 0 : m.p0;
        m.whi = e1 < nV ? oi[e1] : (int32_t)N;
        meta[t] = m;
    }
    for (int w = threadIdx.x; w < n_words; w += TILE_BLOCK) {
        unsigned m = bits[w];
        int pos = (int)prefix[w];
        while (m) {
            const int b = __ffs(m) - 1;
            cam_list[cb + pos++] = (w << 5) + b;
            m &= m - 1;
        }
    }
    for (int e = e0 + threadIdx.x; e < e1; e += TILE_BLOCK) {
        const int j = oj[e];
        const unsigned m = bits[j >> 5] & ((1u << (j & 31)) - 1u);
        const unsigned slot = prefix[j >> 5] + (unsigned)__popc(m);
        opack[e] = ((unsigned)(oi[e] - oi[e0]) << 16) | slot;
    }
}

// ---- gradient pass on camera runs (see ba_runpass.h) --------------------------------------------
struct GradOp {
    static constexpr int NCV = 6, NPV = 3, NPART = 2, NOUT = 1, WAVES_PER_EU = 4, WAVES_PER_EU_COMPACT = 4;
#ifndef BA_MOMENT_PREFETCH
#define BA_MOMENT_PREFETCH 5
#endif
    static constexpr int MOMENT_PREFETCH = BA_MOMENT_PREFETCH;
    __host__ __device__ static constexpr int out_width(int) { return 3; }
    __host__ __device__ static constexpr int out_offset(int) { return 0; }
    __host__ __device__ static constexpr bool cv_used(int) { return true; }
    static constexpr int NCV_LDS = 6;
    __host__ __device__ static constexpr int cv_lds(int k) { return k; }
    __host__ __device__ static constexpr int cv_from_lds(int c) { return c; }
    double f, pcx, pcy;
    __device__ explicit GradOp(const double *scalars, double pix_shift = 0.0)
        : f(scalars[SC_F]), pcx(scalars[SC_CX] - pix_shift), pcy(scalars[SC_CY] - pix_shift) {}
    // the torque sums are kept without their common factor: q x a = f (m0, m1, 1) x (r0, r1, -m.r) exactly,
    // because a = (f / q2) (r0, r1, -m.r) and q = q2 (m0, m1, 1)
    __device__ __forceinline__ double cv_flush(int k, double v) const { return k < 3 ? v : (k == 3 ? -f * v : f * v); }
    // cv: camera-frame force (3) and torque / f (3; the first one negated); pv: world-frame dE/dX_i part;
    // part: r.r, m.r.  uv: pixel centre minus the shift given to the constructor
    __device__ __forceinline__ void obs(bool valid, const double (&x)[3], const double2 (&rw)[6], double2 uv,
                                        double (&cv)[6], double (&pv)[3], double (&part)[2]) const {
        const double R00 = rw[0].x, R01 = rw[0].y, R02 = rw[1].x, R10 = rw[1].y, R11 = rw[2].x, R12 = rw[2].y,
                     R20 = rw[3].x, R21 = rw[3].y, R22 = rw[4].x;
        // rw[4].y, rw[5]: t = -R^T C (stage_camera_row_rt), q = R^T x + t
        const double q0 = fma(R00, x[0], fma(R10, x[1], fma(R20, x[2], rw[4].y)));
        const double q1 = fma(R01, x[0], fma(R11, x[1], fma(R21, x[2], rw[5].x)));
        const double q2 = fma(R02, x[0], fma(R12, x[1], fma(R22, x[2], rw[5].y)));
        const double inv = fast_rcp(q2);
        const double m0 = q0 * inv, m1 = q1 * inv;
        const double r0 = valid ? fma(f, m0, pcx - uv.x) : 0.0;
        const double r1 = valid ? fma(f, m1, pcy - uv.y) : 0.0;
        part[0] = fma(r0, r0, fma(r1, r1, part[0]));
        const double mr = fma(m0, r0, m1 * r1);
        part[1] += mr;
        const double fi = f * inv;
        const double a0 = fi * r0, a1 = fi * r1, a2 = -fi * mr;
        pv[0] = R00 * a0 + R01 * a1 + R02 * a2;
        pv[1] = R10 * a0 + R11 * a1 + R12 * a2;
        pv[2] = R20 * a0 + R21 * a1 + R22 * a2;
        cv[0] += a0;
        cv[1] += a1;
        cv[2] += a2;
        cv[3] = fma(m1, mr, cv[3] + r1);
        cv[4] = fma(m0, mr, cv[4] + r0);
        cv[5] = fma(m0, r1, fma(-m1, r0, cv[5]));
    }
};

// Run layout, pass 1: runs per tile = sum over its cameras of ceil(observations / RUN_LEN).
__global__ void __launch_bounds__(TILE_BLOCK) k_run_count(const uint32_t *__restrict__ opack,
                                                          const TileMeta *__restrict__ meta,
                                                          int64_t *__restrict__ counts, int *__restrict__ max_obs) {
    extern __shared__ int hist[];
    __shared__ int total;
    const TileMeta tm = meta[blockIdx.x];
    for (int c = threadIdx.x; c < tm.nc; c += TILE_BLOCK) hist[c] = 0;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    for (int e = tm.e0 + threadIdx.x; e < tm.e1; e += TILE_BLOCK) atomicAdd(&hist[opack[e] & 0xffffu], 1);
    __syncthreads();
    int runs = 0;
    for (int c = threadIdx.x; c < tm.nc; c += TILE_BLOCK) runs += (hist[c] + RUN_LEN - 1) / RUN_LEN;
    if (runs) atomicAdd(&total, runs);
    __syncthreads();
    if (threadIdx.x == 0) {
        counts[blockIdx.x] = total;
        atomicMax(max_obs, tm.e1 - tm.e0);
    }
}

// Pass 2: every camera's observations go, in point order, into its runs (order of the runs and of the observations
// inside a run: see ba_runpass.h).
// Batches: full batches of 8 cameras, then the remaining cameras in batches of 4, 2 and 1 (binary_rest) - the point
// records' bank pairs are distinct over the 16 lanes served together when those hold 8, 4, 2 or 1 cameras times 2, 4, 8
// or 16 consecutive rounds, and not for 3, 5, 6 or 7 (a tile of 11 cameras as one batch of 8 and one of 3: the
// J^T J pass 7 % slower than with 10 or 12 cameras).
// A thread per observation: what it needs is the observation's rank among its camera's observations of the tile.  The
// four waves take a quarter of the tile's list each; a first walk counts every quarter's observations per camera (the
// quarters' counts, summed up, are where a wave's ranks start), a second walk ranks the 64 observations of a step by
// ballots, camera by camera, and writes the records.  (One thread per CAMERA walking the whole tile, a dependent
// load per observation, took 2.1 ms at ten million observations.)
// LDS (ints): runs per camera, runs before every camera (+ 1), observations per camera, rank bases per wave and
// camera (4 x), first word of every camera's last run.
__host__ __device__ inline size_t run_fill_lds_ints(int cam_cap) { return 8 * (size_t)cam_cap + 1; }

__global__ void __launch_bounds__(TILE_BLOCK) k_run_fill(const uint32_t *__restrict__ opack,
                                                         const double2 *__restrict__ ouv,
                                                         const TileMeta *__restrict__ meta,
                                                         const int64_t *__restrict__ run_ptr, int cam_cap, int pt_stride,
                                                         int binary_rest, uint32_t *__restrict__ rpack,
                                                         double2 *__restrict__ ruv, uint2 *__restrict__ rrec) {
    extern __shared__ int sm[];
    int *runs = sm, *cam_first = sm + cam_cap, *cnt = sm + 2 * cam_cap + 1, *wbase = cnt + cam_cap;
    uint32_t *firstw = (uint32_t *)(wbase + 4 * cam_cap);
    const TileMeta tm = meta[blockIdx.x];
    const int n = tm.e1 - tm.e0, nc = tm.nc;
    const int64_t rb = run_ptr[blockIdx.x];
    const int nruns = (int)(run_ptr[blockIdx.x + 1] - rb);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int idx = tid; idx < 4 * cam_cap; idx += TILE_BLOCK) wbase[idx] = 0;
    __syncthreads();
    const int share = ((n + 3) / 4 + 63) & ~63;
    const int e_lo = min(n, w * share), e_hi = min(n, e_lo + share);
    int *mine = wbase + w * cam_cap;
    for (int e = e_lo + lane; e < e_hi; e += 64) atomicAdd(&mine[opack[tm.e0 + e] & 0xffffu], 1);
    __syncthreads();
    for (int c = tid; c < nc; c += TILE_BLOCK) {
        int acc = 0;
        for (int q = 0; q < 4; ++q) {
            const int v = wbase[q * cam_cap + c];
            wbase[q * cam_cap + c] = acc;
            acc += v;
        }
        cnt[c] = acc;
        runs[c] = (acc + RUN_LEN - 1) / RUN_LEN;
        firstw[c] = 0;
    }
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int c = 0; c < nc; ++c) {
            cam_first[c] = acc;
            acc += runs[c];
        }
        cam_first[nc] = acc;
    }
    __syncthreads();
    // position of run rho of camera c: runs of its batch in earlier rounds + cameras before c still running in rho
    auto run_index = [&](int c, int rho) {
        int c_lo = c & ~7, c_hi = min(nc, c_lo + 8);
        if (binary_rest && c_hi - c_lo < 8) {
            int start = c_lo;
            const int rest = nc - c_lo;
            for (int sz = 4; sz >= 1; sz >>= 1)
                if (rest & sz) {
                    if (c < start + sz) {
                        c_lo = start, c_hi = start + sz;
                        break;
                    }
                    start += sz;
                }
        }
        int pos = cam_first[c_lo];
        for (int o = c_lo; o < c_hi; ++o) pos += min(runs[o], rho) + ((o < c && runs[o] > rho) ? 1 : 0);
        return pos;
    };
    for (int e_step = e_lo; e_step < e_hi; e_step += 64) {
        const int e = e_step + lane;
        const bool on = e < e_hi;
        const uint32_t pk = on ? opack[tm.e0 + e] : 0u;
        const int c = (int)(pk & 0xffffu);
        int q = 0;
        unsigned long long rem = __ballot(on);
        while (rem) {       // the cameras present in this step, one after the other (the loop is uniform over the wave)
            const int leader = __ffsll((long long)rem) - 1;
            const int cc = __builtin_amdgcn_readlane(c, leader);
            const unsigned long long m = __ballot(on && c == cc);
            const int base = mine[cc];
            if (on && c == cc) q = base + __popcll(m & ((1ull << lane) - 1ull));
            if (lane == leader) mine[cc] = base + __popcll(m);
            rem &= ~m;
        }
        if (on) {
            const int rho = q / RUN_LEN, k8 = q % RUN_LEN;
            const int run = run_index(c, rho);
            const int plane = (k8 - c) & (RUN_LEN - 1);
            const int64_t idx = rb * RUN_LEN + (int64_t)plane * nruns + run;
            // the point's position in the LDS record array (pt_swz), so that the pass needs no index arithmetic
            const uint32_t word = ((uint32_t)run_pt_swz((int)(pk >> 16), pt_stride) << 16) | ((uint32_t)(rho & 15) << RUN_SLOT_BITS) | (uint32_t)c;
            const double2 uv = ouv[tm.e0 + e];
            if (k8 == 0 && rho == runs[c] - 1) firstw[c] = word;
            if (rrec) {
                const int px = (int)(uv.x - 0.5), py = (int)(uv.y - 0.5);
                rrec[idx] = make_uint2(word, ((uint32_t)px & 0xffffu) | ((uint32_t)py << 16));
            } else {
                rpack[idx] = word;
                ruv[idx] = uv;
            }
        }
    }
    __syncthreads();
    // zero-weight records up to the end of every camera's last run
    for (int id = tid; id < nc * RUN_LEN; id += TILE_BLOCK) {
        const int c = id / RUN_LEN, q = cnt[c] + id % RUN_LEN, my_runs = runs[c];
        if (q >= my_runs * RUN_LEN) continue;
        const int rho = my_runs - 1, run = run_index(c, rho);
        const int plane = ((q % RUN_LEN) - c) & (RUN_LEN - 1);
        const int64_t idx = rb * RUN_LEN + (int64_t)plane * nruns + run;
        const uint32_t word = (firstw[c] & 0xffff0000u) | ((uint32_t)(rho & 15) << RUN_SLOT_BITS) | (uint32_t)c;
        if (rrec) {
            rrec[idx] = make_uint2(word, 0x8000u);
        } else {
            rpack[idx] = word;
            ruv[idx] = make_double2(__longlong_as_double(0x7ff8000000000000ll), 0.0);
        }
    }
}


// Dense block layout (see ba_blockpass.h), pass 1: units per tile = chunks of 128 points x batches of 4 cameras.
__global__ void k_block_count(const TileMeta *__restrict__ meta, int nt, int64_t *__restrict__ counts) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    const TileMeta tm = meta[t];
    counts[t] = tm.e0 < tm.e1 ? (int64_t)((tm.np + BLK_PTS - 1) / BLK_PTS) * ((tm.nc + BLK_CAMS - 1) / BLK_CAMS) : 0;
}

// every record slot starts as a zero-weight record
__global__ void k_block_clear(int64_t n, uint32_t *__restrict__ brec, double2 *__restrict__ buv) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    if (brec) brec[e] = 0x8000u;
    else buv[e] = make_double2(__longlong_as_double(0x7ff8000000000000ll), 0.0);
}

// Pass 2: every observation of the tile goes to the slot its point and camera name.
__global__ void __launch_bounds__(TILE_BLOCK) k_block_fill(const uint32_t *__restrict__ opack,
                                                           const double2 *__restrict__ ouv,
                                                           const TileMeta *__restrict__ meta,
                                                           const int64_t *__restrict__ blk_ptr,
                                                           uint32_t *__restrict__ brec, double2 *__restrict__ buv) {
    const TileMeta tm = meta[blockIdx.x];
    const int nchunk = (tm.np + BLK_PTS - 1) / BLK_PTS;
    const int64_t ub = blk_ptr[blockIdx.x];
    for (int e = tm.e0 + threadIdx.x; e < tm.e1; e += TILE_BLOCK) {
        const uint32_t pk = opack[e];
        const int pt = (int)(pk >> 16), slot = (int)(pk & 0xffffu);
        const int64_t unit = ub + (int64_t)(slot / BLK_CAMS) * nchunk + pt / BLK_PTS;
        const int64_t idx = (unit * 8 + (pt & 7)) * 64 + (slot % BLK_CAMS) * 16 + ((pt >> 3) & 15);
        const double2 uv = ouv[e];
        if (brec) {
            const int px = (int)(uv.x - 0.5), py = (int)(uv.y - 0.5);
            brec[idx] = ((uint32_t)px & 0xffffu) | ((uint32_t)py << 16);
        } else {
            buv[idx] = uv;
        }
    }
}

// partial sums -> sums[2]; camera-frame sums -> dE/dC, dE/dTheta (un-normalised).  The camera-frame
// sums are consumed here and left at zero for the next observation pass.  With a mailbox (host
// memory mapped into the device) the energy sum is also posted there, followed by `seq`, so that
// the host can pick it up without a copy command or a stream synchronisation.
__global__ void __launch_bounds__(256) k_finalize(int64_t K, int n_partials, const double *__restrict__ partials,
                                                  double *__restrict__ cam_sums, const double *__restrict__ T,
                                                  const double *__restrict__ scalars, double *__restrict__ E_sum,
                                                  double *__restrict__ gphi, double *__restrict__ gC,
                                                  double *__restrict__ gTheta, volatile double *mailbox, double seq) {
    if (blockIdx.x == 0) {
        __shared__ double red[2][4];
        double e = 0.0, p = 0.0;
        const double2 *pp = (const double2 *)partials;
        int t = threadIdx.x;
        for (; t + 7 * 256 < n_partials; t += 8 * 256) {  // 8 independent loads in flight per thread
            double2 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = pp[t + k * 256];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                e += v[k].x;
                p += v[k].y;
            }
        }
        for (; t < n_partials; t += 256) {
            e += pp[t].x;
            p += pp[t].y;
        }
        e = wave_sum(e);
        p = wave_sum(p);
        if ((threadIdx.x & 63) == 0) {
            red[0][threadIdx.x >> 6] = e;
            red[1][threadIdx.x >> 6] = p;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const double e_sum = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
            E_sum[0] = e_sum;
            if (gphi) gphi[0] = scalars[SC_DPHI] * ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
            if (mailbox) {
                mailbox[0] = e_sum;
                __threadfence_system();
                mailbox[1] = seq;
            }
        }
    }
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= K) return;
    const double s0 = cam_sums[0 * K + j], s1 = cam_sums[1 * K + j], s2 = cam_sums[2 * K + j];
    const double w0 = cam_sums[3 * K + j], w1 = cam_sums[4 * K + j], w2 = cam_sums[5 * K + j];
#pragma unroll
    for (int k = 0; k < 6; ++k) cam_sums[k * K + j] = 0.0;
    if (!gC) return;
    const double *row = T + j * CAM_ROW;
    gC[3 * j + 0] = -(row[0] * s0 + row[1] * s1 + row[2] * s2);
    gC[3 * j + 1] = -(row[3] * s0 + row[4] * s1 + row[5] * s2);
    gC[3 * j + 2] = -(row[6] * s0 + row[7] * s1 + row[8] * s2);
    gTheta[3 * j + 0] = -(row[12] * w0 - row[13] * w1);
    gTheta[3 * j + 1] = -(row[3] * w0 + row[4] * w1 + row[5] * w2);
    gTheta[3 * j + 2] = -w2;
}

// posts a device-resident energy sum to the host mailbox (used when the sum is completed outside the
// finalisation kernel, e.g. by a cross-GPU reduction)
__global__ void k_post_energy(const double *__restrict__ E_sum, volatile double *mailbox, double seq) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        mailbox[0] = E_sum[0];
        __threadfence_system();
        mailbox[1] = seq;
    }
}

// --------------------------------------------------------------------------------------------
// K6: update rules (p2w:336-360)
// --------------------------------------------------------------------------------------------
// elementwise update of n parameters; `release` (optional) receives the values before the update
__global__ void k_update(int64_t n, double *__restrict__ x, const double *__restrict__ g, double *__restrict__ s,
                         double *__restrict__ r, double *__restrict__ release, UpdateRule u) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const double xo = x[t];
        if (release) release[t] = xo;
        double sv = 0.0, rv = 0.0;
        if (u.mode == 2) {
            sv = s[t];
            rv = r[t];
        }
        x[t] = apply_rule(u, xo, g[t], &sv, &rv);
        if (u.mode == 2) {
            s[t] = sv;
            r[t] = rv;
        }
    }
}

// Points: update rule and the rigid part of the gauge fix X <- ((X - C0) @ R0) * factor
// (p2w:285-293, 327) in ONE pass over X and its moments.  Inputs and outputs are separate
// pointers, each pair either identical (in place) or disjoint (the caller keeps the inputs, e.g.
// as the snapshot the LR schedule rolls back to, p2w:397-398).
// A workgroup owns 256 consecutive points = 768 consecutive doubles of every array, which it
// reads and writes element-wise (fully coalesced); the 3-vectors meet in LDS for the rotation.
constexpr int PS_POINTS = 256;
__global__ void __launch_bounds__(PS_POINTS) k_points_step(int64_t N, const double *X, double *X_out,
                                                           const double *__restrict__ gX, const double *s,
                                                           double *s_out, const double *r, double *r_out,
                                                           UpdateRule u, const double *__restrict__ gauge) {
    __shared__ double xs[3 * PS_POINTS];
    const int64_t base = (int64_t)blockIdx.x * (3 * PS_POINTS);
    const int64_t n = 3 * N;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int l = k * PS_POINTS + threadIdx.x;
        const int64_t t = base + l;
        if (t < n) {
            const double xo = X[t];
            double xn = xo;
            if (u.mode) {
                double sv = 0.0, rv = 0.0;
                if (u.mode == 2) {
                    sv = s[t];
                    rv = r[t];
                }
                xn = apply_rule(u, xo, gX[t], &sv, &rv);
                if (u.mode == 2) {
                    s_out[t] = sv;
                    r_out[t] = rv;
                }
            }
            if (gauge) xs[l] = xn;
            else X_out[t] = xn;
        }
    }
    if (!gauge) return;
    __syncthreads();
    const double c0 = gauge[9], c1 = gauge[10], c2 = gauge[11], factor = gauge_factor(gauge);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int l = k * PS_POINTS + threadIdx.x;
        const int64_t t = base + l;
        if (t < n) {
            const int pt = l / 3, comp = l - 3 * pt;
            const double d0 = xs[3 * pt + 0] - c0, d1 = xs[3 * pt + 1] - c1, d2 = xs[3 * pt + 2] - c2;
            X_out[t] = (d0 * gauge[comp] + d1 * gauge[3 + comp] + d2 * gauge[6 + comp]) * factor;
        }
    }
}

// --------------------------------------------------------------------------------------------
// K7: camera side of one step: update rule on the free slice of
// [C | Theta | phi_x], then align_scene + normalize_scene for the cameras (p2w:270-333), then the
// camera table and scalars of the NEW parameters, so that the next observation pass needs no
// table kernel.  k_camera_step is the single-workgroup form without an update rule (in place:
// the stand-alone gauge fix); k_camera_step_mb below is the full step.  With the gauge fix the table holds R0^T R(Theta), the matrix align_scene returns
// and the reference keeps using (p2w:711), and cos/sin(theta_z) recovered from its entries the way
// p2w:665-676 does.  gauge[0..8] = R0, [9..11] = C0, [12] = scale factor, [13] = j1.
// --------------------------------------------------------------------------------------------
struct CameraStep {
    int64_t K;
    const double *cam_in; // [C (3K) | Theta (3K) | phi_x (1)] before the step
    double *cam;          // ... after the step; the same buffer as cam_in, or a disjoint one
    const double *g;      // gradient in the same packing (may be null when mode == 0)
    const double *s_in, *r_in;  // Adam moments, same packing
    double *s, *r;        // updated moments (same buffers as s_in / r_in, or disjoint ones)
    int64_t lo, hi;       // free slice
    UpdateRule rule;
    int do_gauge;
    double W, H;
    double *T;            // camera table (K x CAM_ROW), may be null
    int scale_later;      // 1: k_scale_centres follows and applies the scale factor (quad kernel only)
    double *scalars;      // may be null
    double *R_out;        // optional K x 9
    double *gauge;        // optional 16
};

__global__ void __launch_bounds__(1024) k_camera_step(CameraStep a) {
    __shared__ double sR0[9], sC0[3];
    __shared__ double red_v[16];
    __shared__ long long red_j[16];
    __shared__ double s_factor;
    const int64_t K = a.K;
    double *C = a.cam, *Theta = a.cam + 3 * K;

    if (threadIdx.x == 0 && a.scalars) {
        const double ph = a.cam[6 * K];
        const double f = a.W / 2.0 / tan(ph / 2.0);
        a.scalars[SC_F] = f;
        a.scalars[SC_CX] = a.W / 2.0;
        a.scalars[SC_CY] = a.H / 2.0;
        a.scalars[SC_DPHI] = -f / sin(ph);
    }

    if (!a.do_gauge) {
        if (a.T) {
            for (int64_t j = threadIdx.x; j < K; j += 1024) {
                double R[9], cz, sz;
                rotation_yxz(Theta[3 * j + 0], Theta[3 * j + 1], Theta[3 * j + 2], R, &cz, &sz);
                double *row = a.T + j * CAM_ROW;
#pragma unroll
                for (int k = 0; k < 9; ++k) row[k] = R[k];
                row[9] = C[3 * j + 0];
                row[10] = C[3 * j + 1];
                row[11] = C[3 * j + 2];
                row[12] = cz;
                row[13] = sz;
            }
        }
        return;
    }

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
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = 0; q < 3; ++q) Rn[3 * p + q] = R0[p] * R[q] + R0[3 + p] * R[3 + q] + R0[6 + p] * R[6 + q];
        // get_rotations (p2w:260-266)
        const double Cx = sqrt(Rn[3] * Rn[3] + Rn[4] * Rn[4]);
        const double Sx = -Rn[5];
        Theta[3 * j + 0] = atan2(Sx, Cx);
        Theta[3 * j + 1] = atan2(Rn[2] / Cx, Rn[8] / Cx);
        Theta[3 * j + 2] = atan2(Rn[3] / Cx, Rn[4] / Cx);
        if (a.R_out) {
#pragma unroll
            for (int k = 0; k < 9; ++k) a.R_out[9 * j + k] = Rn[k];
        }
        if (a.T) {
            double *row = a.T + j * CAM_ROW;
#pragma unroll
            for (int k = 0; k < 9; ++k) row[k] = Rn[k];
            row[12] = Rn[4] / Cx;  // cos(theta_z), p2w:674
            row[13] = Rn[3] / Cx;  // sin(theta_z), p2w:675
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
    __threadfence_block();
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
        if (a.gauge) {
#pragma unroll
            for (int k = 0; k < 9; ++k) a.gauge[k] = R0[k];
            a.gauge[9] = c00;
            a.gauge[10] = c01;
            a.gauge[11] = c02;
            a.gauge[12] = factor;
            a.gauge[13] = (double)bj;
            a.gauge[14] = 0.0;
            a.gauge[15] = 0.0;
        }
    }
    __syncthreads();
    const double factor = s_factor;
    for (int64_t j = threadIdx.x; j < K; j += 1024) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double c = C[3 * j + k] * factor;
            C[3 * j + k] = c;
            if (a.T) a.T[j * CAM_ROW + 9 + k] = c;
        }
    }
}

// The same step spread over ceil(K/256) workgroups.  The only cross-workgroup dependencies are
// camera 0 (whose new pose defines the rigid motion) and the longest baseline (which defines the
// scale):
//   * every workgroup recomputes camera 0's update from its OLD parameters, which nobody
//     overwrites before the end (also when the step runs in place): camera 0's own results are
//     staged in `stage` by workgroup 0;
//   * every workgroup publishes its best baseline and takes a ticket; the workgroup that draws
//     the last ticket reduces the baselines, scales all centres, writes camera 0 and phi_x.
// Publication follows the agent-scope release / acquire protocol (stores, vmcnt(0), barrier,
// release fence, relaxed ticket; last arriver: acquire fence, barrier, plain loads).
struct CameraStepShared {
    int *counter;         // zero before the launch; reset by the last workgroup
    double *block_best;   // 2 per workgroup: squared baseline, camera index
    double *stage;        // 16: camera 0's Theta (3), moments s (6), r (6)
};

__device__ __forceinline__ void camera_pose_update(const CameraStep &a, int64_t j, bool write, double *c, double *th,
                                                   double *s_out, double *r_out) {
    const int64_t K = a.K;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int64_t t = (k < 3) ? 3 * j + k : 3 * K + 3 * j + (k - 3);
        const double xo = a.cam_in[t];
        double xn = xo, sv = 0.0, rv = 0.0;
        if (a.rule.mode && t >= a.lo && t < a.hi) {
            if (a.rule.mode == 2) {
                sv = a.s_in[t];
                rv = a.r_in[t];
            }
            xn = apply_rule(a.rule, xo, a.g[t], &sv, &rv);
            if (write && a.rule.mode == 2) {
                a.s[t] = sv;
                a.r[t] = rv;
            }
        }
        if (s_out) {
            s_out[k] = sv;
            r_out[k] = rv;
        }
        if (k < 3) c[k] = xn;
        else th[k - 3] = xn;
    }
}

__global__ void __launch_bounds__(256) k_camera_step_mb(CameraStep a, CameraStepShared sh) {
    __shared__ double sR0[9], sC0[3];
    __shared__ double red_v[4];
    __shared__ long long red_j[4];
    __shared__ int s_last;
    __shared__ double s_factor;
    const int64_t K = a.K;
    double *C = a.cam, *Theta = a.cam + 3 * K;
    const int tid = threadIdx.x;
    const int64_t j = (int64_t)blockIdx.x * 256 + tid;

    if (!a.do_gauge) {
        // independent cameras: update, then the table row of the new pose
        if (j < K) {
            double c[3], th[3];
            camera_pose_update(a, j, true, c, th, nullptr, nullptr);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                C[3 * j + k] = c[k];
                Theta[3 * j + k] = th[k];
            }
            if (a.T) {
                double R[9], cz, sz;
                rotation_yxz(th[0], th[1], th[2], R, &cz, &sz);
                double *row = a.T + j * CAM_ROW;
#pragma unroll
                for (int k = 0; k < 9; ++k) row[k] = R[k];
                row[9] = c[0];
                row[10] = c[1];
                row[11] = c[2];
                row[12] = cz;
                row[13] = sz;
            }
        }
        if (j == 0) {
            const int64_t t = 6 * K;
            double ph = a.cam_in[t];
            if (a.rule.mode && t >= a.lo && t < a.hi) {
                double sv = 0.0, rv = 0.0;
                if (a.rule.mode == 2) {
                    sv = a.s_in[t];
                    rv = a.r_in[t];
                }
                ph = apply_rule(a.rule, ph, a.g[t], &sv, &rv);
                if (a.rule.mode == 2) {
                    a.s[t] = sv;
                    a.r[t] = rv;
                }
            }
            a.cam[t] = ph;
            if (a.scalars) {
                const double f = a.W / 2.0 / tan(ph / 2.0);
                a.scalars[SC_F] = f;
                a.scalars[SC_CX] = a.W / 2.0;
                a.scalars[SC_CY] = a.H / 2.0;
                a.scalars[SC_DPHI] = -f / sin(ph);
            }
        }
        return;
    }

    // camera 0's new pose, from its old parameters (identical in every workgroup)
    if (tid == 0) {
        double c[3], th[3], s6[6], r6[6], cz, sz;
        camera_pose_update(a, 0, false, c, th, s6, r6);
        rotation_yxz(th[0], th[1], th[2], sR0, &cz, &sz);
        sC0[0] = c[0];
        sC0[1] = c[1];
        sC0[2] = c[2];
        if (blockIdx.x == 0) {
            // camera 0 after the gauge fix: centre 0, angles of R0^T R0; staged until the end
            double Rn[9];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    Rn[3 * p + q] = sR0[p] * sR0[q] + sR0[3 + p] * sR0[3 + q] + sR0[6 + p] * sR0[6 + q];
            const double Cx = sqrt(Rn[3] * Rn[3] + Rn[4] * Rn[4]);
            sh.stage[0] = atan2(-Rn[5], Cx);
            sh.stage[1] = atan2(Rn[2] / Cx, Rn[8] / Cx);
            sh.stage[2] = atan2(Rn[3] / Cx, Rn[4] / Cx);
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                sh.stage[3 + k] = s6[k];
                sh.stage[9 + k] = r6[k];
            }
            if (a.R_out) {
#pragma unroll
                for (int k = 0; k < 9; ++k) a.R_out[k] = Rn[k];
            }
            if (a.T) {
#pragma unroll
                for (int k = 0; k < 9; ++k) a.T[k] = Rn[k];
                a.T[9] = 0.0;
                a.T[10] = 0.0;
                a.T[11] = 0.0;
                a.T[12] = Rn[4] / Cx;
                a.T[13] = Rn[3] / Cx;
            }
        }
    }
    __syncthreads();
    double R0[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) R0[k] = sR0[k];
    const double c00 = sC0[0], c01 = sC0[1], c02 = sC0[2];

    double best = -1.0;
    long long best_j = K;
    if (j >= 1 && j < K) {
        double c[3], th[3];
        camera_pose_update(a, j, true, c, th, nullptr, nullptr);
        const double d0 = c[0] - c00, d1 = c[1] - c01, d2 = c[2] - c02;
        const double n0 = d0 * R0[0] + d1 * R0[3] + d2 * R0[6];  // (C - C0) @ R0
        const double n1 = d0 * R0[1] + d1 * R0[4] + d2 * R0[7];
        const double n2 = d0 * R0[2] + d1 * R0[5] + d2 * R0[8];
        C[3 * j + 0] = n0;
        C[3 * j + 1] = n1;
        C[3 * j + 2] = n2;
        double R[9], cz, sz;
        rotation_yxz(th[0], th[1], th[2], R, &cz, &sz);
        double Rn[9];  // R0^T R
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = 0; q < 3; ++q) Rn[3 * p + q] = R0[p] * R[q] + R0[3 + p] * R[3 + q] + R0[6 + p] * R[6 + q];
        const double Cx = sqrt(Rn[3] * Rn[3] + Rn[4] * Rn[4]);  // get_rotations (p2w:260-266)
        Theta[3 * j + 0] = atan2(-Rn[5], Cx);
        Theta[3 * j + 1] = atan2(Rn[2] / Cx, Rn[8] / Cx);
        Theta[3 * j + 2] = atan2(Rn[3] / Cx, Rn[4] / Cx);
        if (a.R_out) {
#pragma unroll
            for (int k = 0; k < 9; ++k) a.R_out[9 * j + k] = Rn[k];
        }
        if (a.T) {
            double *row = a.T + j * CAM_ROW;
#pragma unroll
            for (int k = 0; k < 9; ++k) row[k] = Rn[k];
            row[12] = Rn[4] / Cx;  // cos(theta_z), p2w:674
            row[13] = Rn[3] / Cx;  // sin(theta_z), p2w:675
        }
        best = n0 * n0 + n1 * n1 + n2 * n2;
        best_j = j;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {  // argmax, ties to the smaller index (p2w:322)
        const double ov = __shfl_xor(best, off);
        const long long oj = __shfl_xor(best_j, off);
        if (ov > best || (ov == best && oj < best_j)) {
            best = ov;
            best_j = oj;
        }
    }
    if ((tid & 63) == 0) {
        red_v[tid >> 6] = best;
        red_j[tid >> 6] = best_j;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) {
            if (red_v[w] > best || (red_v[w] == best && red_j[w] < best_j)) {
                best = red_v[w];
                best_j = red_j[w];
            }
        }
        sh.block_best[2 * blockIdx.x + 0] = best;
        sh.block_best[2 * blockIdx.x + 1] = (double)best_j;
    }
    // publish this workgroup's results and take a ticket
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int ticket = __hip_atomic_fetch_add(sh.counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = ticket == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;

    // last workgroup: everything the others wrote is visible after the acquire
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (tid == 0) {
        double bv = -1.0;
        long long bj = K;
        for (int b = 0; b < (int)gridDim.x; ++b) {
            const double v = sh.block_best[2 * b + 0];
            const long long vj = (long long)sh.block_best[2 * b + 1];
            if (v > bv || (v == bv && vj < bj)) {
                bv = v;
                bj = vj;
            }
        }
        const double factor = 1.0 / sqrt(bv);
        s_factor = factor;
        if (a.gauge) {
#pragma unroll
            for (int k = 0; k < 9; ++k) a.gauge[k] = R0[k];
            a.gauge[9] = c00;
            a.gauge[10] = c01;
            a.gauge[11] = c02;
            a.gauge[12] = factor;
            a.gauge[13] = (double)bj;
            a.gauge[14] = 0.0;
            a.gauge[15] = 0.0;
        }
        // camera 0: new moments, gauge-fixed pose
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int64_t t = (k < 3) ? k : 3 * K + (k - 3);
            if (a.rule.mode == 2 && t >= a.lo && t < a.hi) {
                a.s[t] = sh.stage[3 + k];
                a.r[t] = sh.stage[9 + k];
            }
            a.cam[t] = (k < 3) ? 0.0 : sh.stage[k - 3];
        }
        *sh.counter = 0;
    }
    if (tid == 64) {  // phi_x and the scalars that depend on it
        const int64_t t = 6 * K;
        double ph = a.cam_in[t];
        if (a.rule.mode && t >= a.lo && t < a.hi) {
            double sv = 0.0, rv = 0.0;
            if (a.rule.mode == 2) {
                sv = a.s_in[t];
                rv = a.r_in[t];
            }
            ph = apply_rule(a.rule, ph, a.g[t], &sv, &rv);
            if (a.rule.mode == 2) {
                a.s[t] = sv;
                a.r[t] = rv;
            }
        }
        a.cam[t] = ph;
        if (a.scalars) {
            const double f = a.W / 2.0 / tan(ph / 2.0);
            a.scalars[SC_F] = f;
            a.scalars[SC_CX] = a.W / 2.0;
            a.scalars[SC_CY] = a.H / 2.0;
            a.scalars[SC_DPHI] = -f / sin(ph);
        }
    }
    __syncthreads();
    const double factor = s_factor;
    for (int64_t jj = 1 + tid; jj < K; jj += 256) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double c = C[3 * jj + k] * factor;
            C[3 * jj + k] = c;
            if (a.T) a.T[jj * CAM_ROW + 9 + k] = c;
        }
    }
}

// --------------------------------------------------------------------------------------------
// Camera step with the gauge fix, four lanes per camera.  The step is a chain of long fp64
// routines per camera (update rule, three sincos, three atan2): one thread per camera leaves a
// wave waiting on that chain, so lane `comp` (0..2) of a quad handles component `comp` of the
// centre and of the angles, runs ONE sincos and ONE atan2, and the quad exchanges the results
// with DPP quad broadcasts.  Quad 63 of EVERY workgroup recomputes camera 0 (whose new pose
// defines the rigid motion) from its old parameters; quads 0..62 own cameras 1 + 63 b + q.
// Camera 0's own results are staged and written by the workgroup that draws the last ticket,
// together with the scale factor (see k_camera_step_mb for the protocol), because every
// workgroup reads camera 0's OLD values.
// (The step that also finishes a deferred observation pass is k_camera_step_free below.)
// --------------------------------------------------------------------------------------------
// what a camera step that finishes the observation pass needs besides CameraStep
struct FinalizeArgs {
    double *partials = nullptr;         // 2 per tile
    int n_partials = 0;                 // entries to add up
    int n_own = 0;                      // entries the pass of this GPU writes; the others (sums from other GPUs in a
                                        // shared sum buffer) are cleared once they have been added
    double *cam_sums = nullptr;         // 6 x K
    double *g_out = nullptr;            // [gC (3K) | gTheta (3K) | gphi | E_sum]
    volatile double *mailbox = nullptr;
    double seq = 0.0;
};

template <int Q>
__device__ __forceinline__ double quad_bcast(double v) {
    constexpr int ctrl = Q | (Q << 2) | (Q << 4) | (Q << 6);   // quad_perm:[Q,Q,Q,Q]
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

constexpr int STAGE_DOUBLES = 32;   // camera 0: Theta (3), s (6), r (6), table row R (9) + cos/sin theta_z (2)

__global__ void __launch_bounds__(256) k_camera_step_quad(CameraStep a, CameraStepShared sh) {
    __shared__ double sR0[9], sC0[3];
    __shared__ double red_v[4];
    __shared__ long long red_j[4];
    __shared__ int s_last;
    __shared__ double s_factor;
    const int64_t K = a.K;
    const int tid = threadIdx.x;
    const int quad = tid >> 2, comp = tid & 3;
    const bool is_c0 = quad == 63;
    const int64_t j = is_c0 ? 0 : 1 + (int64_t)blockIdx.x * 63 + quad;
    const bool live = j < K && comp < 3;
    const bool own = live && !is_c0;                 // writes its results directly
    const bool stage0 = live && is_c0 && blockIdx.x == 0;   // camera 0's results, staged
    const int64_t jj = j < K ? j : 0;
    const int cc = comp < 3 ? comp : 0;

    // ---- everything this lane reads, requested up front
    double in_x[2], in_s[2] = {0.0, 0.0}, in_r[2] = {0.0, 0.0}, in_g[2] = {0.0, 0.0};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int64_t t = (h == 0 ? 0 : 3 * K) + 3 * jj + cc;
        in_x[h] = a.cam_in[t];
        if (a.rule.mode == 2 && t >= a.lo && t < a.hi) {
            in_s[h] = a.s_in[t];
            in_r[h] = a.r_in[t];
        }
        if (a.g) in_g[h] = a.g[t];
    }

    // phi_x and the scalars that depend on it: one thread of workgroup 0, off the tail of the kernel (nothing
    // else in this launch reads them)
    if (blockIdx.x == 0 && tid == 64) {
        const int64_t t = 6 * K;
        double ph = a.cam_in[t];
        if (a.rule.mode && t >= a.lo && t < a.hi) {
            double sv = 0.0, rv = 0.0;
            if (a.rule.mode == 2) {
                sv = a.s_in[t];
                rv = a.r_in[t];
            }
            ph = apply_rule(a.rule, ph, a.g[t], &sv, &rv);
            if (a.rule.mode == 2) {
                a.s[t] = sv;
                a.r[t] = rv;
            }
        }
        a.cam[t] = ph;
        if (a.scalars) {
            const double f = a.W / 2.0 / tan(ph / 2.0);
            a.scalars[SC_F] = f;
            a.scalars[SC_CX] = a.W / 2.0;
            a.scalars[SC_CY] = a.H / 2.0;
            a.scalars[SC_DPHI] = -f / sin(ph);
        }
    }

    // ---- un-normalised gradient of C[comp], Theta[comp] of this quad's camera
    const double gc = in_g[0], gt = in_g[1];

    // ---- update rule (p2w:704-708) on the two parameters of this lane
    double pnew[2], snew[2] = {0.0, 0.0}, rnew[2] = {0.0, 0.0};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int64_t t = (h == 0 ? 0 : 3 * K) + 3 * jj + cc;
        const double xo = in_x[h];
        double xn = xo;
        if (a.rule.mode && t >= a.lo && t < a.hi) {
            if (a.rule.mode == 2) {
                snew[h] = in_s[h];
                rnew[h] = in_r[h];
            }
            xn = apply_rule(a.rule, xo, h == 0 ? gc : gt, &snew[h], &rnew[h]);
            if (own && a.rule.mode == 2) {
                a.s[t] = snew[h];
                a.r[t] = rnew[h];
            }
        }
        pnew[h] = xn;
    }
    double sn, cs;
    sincos(pnew[1], &sn, &cs);
    const double sx = quad_bcast<0>(sn), cx = quad_bcast<0>(cs), sy = quad_bcast<1>(sn), cy = quad_bcast<1>(cs),
                 sz = quad_bcast<2>(sn), cz = quad_bcast<2>(cs);
    const double c0v = quad_bcast<0>(pnew[0]), c1v = quad_bcast<1>(pnew[0]), c2v = quad_bcast<2>(pnew[0]);
    double R[9];   // Ry Rx Rz (p2w:114-140)
    {
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
    }
    if (is_c0 && comp == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) sR0[k] = R[k];
        sC0[0] = c0v;
        sC0[1] = c1v;
        sC0[2] = c2v;
    }
    __syncthreads();
    double R0[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) R0[k] = sR0[k];
    const double c00 = sC0[0], c01 = sC0[1], c02 = sC0[2];

    // ---- gauge fix of this camera (p2w:270-333): centre (C - C0) @ R0, rotation R0^T R
    const double d0 = c0v - c00, d1 = c1v - c01, d2 = c2v - c02;
    const double n0 = d0 * R0[0] + d1 * R0[3] + d2 * R0[6];
    const double n1 = d0 * R0[1] + d1 * R0[4] + d2 * R0[7];
    const double n2 = d0 * R0[2] + d1 * R0[5] + d2 * R0[8];
    double Rn[9];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int q = 0; q < 3; ++q) Rn[3 * p + q] = R0[p] * R[q] + R0[3 + p] * R[3 + q] + R0[6 + p] * R[6 + q];
    const double Cx = sqrt(Rn[3] * Rn[3] + Rn[4] * Rn[4]);   // get_rotations (p2w:260-266)
    const double ay = cc == 0 ? -Rn[5] : (cc == 1 ? Rn[2] / Cx : Rn[3] / Cx);
    const double ax = cc == 0 ? Cx : (cc == 1 ? Rn[8] / Cx : Rn[4] / Cx);
    const double th_g = atan2(ay, ax);
    const double n_own = cc == 0 ? n0 : (cc == 1 ? n1 : n2);
    if (own) {
        a.cam[3 * j + cc] = n_own;              // scaled by the last workgroup
        a.cam[3 * K + 3 * j + cc] = th_g;
        if (a.R_out) {
#pragma unroll
            for (int k = 0; k < 3; ++k) a.R_out[9 * j + 3 * cc + k] = Rn[3 * cc + k];
        }
        if (a.T) {
            double *row = a.T + j * CAM_ROW;
#pragma unroll
            for (int k = 0; k < 3; ++k) row[3 * cc + k] = Rn[3 * cc + k];
            if (cc == 0) {
                row[12] = Rn[4] / Cx;  // cos(theta_z), p2w:674
                row[13] = Rn[3] / Cx;  // sin(theta_z), p2w:675
            }
        }
    }
    if (stage0) {
        sh.stage[cc] = th_g;
        sh.stage[3 + cc] = snew[0];
        sh.stage[6 + cc] = snew[1];
        sh.stage[9 + cc] = rnew[0];
        sh.stage[12 + cc] = rnew[1];
#pragma unroll
        for (int k = 0; k < 3; ++k) sh.stage[15 + 3 * cc + k] = Rn[3 * cc + k];
        if (cc == 0) {
            sh.stage[24] = Rn[4] / Cx;
            sh.stage[25] = Rn[3] / Cx;
        }
        if (a.R_out) {
#pragma unroll
            for (int k = 0; k < 3; ++k) a.R_out[3 * cc + k] = Rn[3 * cc + k];
        }
    }

    // ---- longest baseline from camera 0: argmax with ties to the smaller index (p2w:322)
    double best = (own && cc == 0) ? n0 * n0 + n1 * n1 + n2 * n2 : -1.0;
    long long best_j = (own && cc == 0) ? j : K;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(best, off);
        const long long oj = __shfl_xor(best_j, off);
        if (ov > best || (ov == best && oj < best_j)) {
            best = ov;
            best_j = oj;
        }
    }
    if ((tid & 63) == 0) {
        red_v[tid >> 6] = best;
        red_j[tid >> 6] = best_j;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) {
            if (red_v[w] > best || (red_v[w] == best && red_j[w] < best_j)) {
                best = red_v[w];
                best_j = red_j[w];
            }
        }
        sh.block_best[2 * blockIdx.x + 0] = best;
        sh.block_best[2 * blockIdx.x + 1] = (double)best_j;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int ticket = __hip_atomic_fetch_add(sh.counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = ticket == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;

    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    // every load below is issued before its first use: the caches were just invalidated, and a
    // chain of dependent misses is what this tail costs
    // first batch of the centres to be scaled below: requested together with the workgroups' maxima
    const int n3 = (int)(3 * K);
    double v[24];
#pragma unroll
    for (int u = 0; u < 24; ++u) {
        const int idx = 3 + u * 256 + tid;
        v[u] = (idx < n3 && !a.scale_later) ? a.cam[idx] : 0.0;
    }
    double bv = -1.0;
    long long bj = K;
    for (int b = tid; b < (int)gridDim.x; b += 256) {
        const double v = sh.block_best[2 * b + 0];
        const long long vj = (long long)sh.block_best[2 * b + 1];
        if (v > bv || (v == bv && vj < bj)) {
            bv = v;
            bj = vj;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(bv, off);
        const long long oj = __shfl_xor(bj, off);
        if (ov > bv || (ov == bv && oj < bj)) {
            bv = ov;
            bj = oj;
        }
    }
    if ((tid & 63) == 0) {
        red_v[tid >> 6] = bv;
        red_j[tid >> 6] = bj;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) {
            if (red_v[w] > bv || (red_v[w] == bv && red_j[w] < bj)) {
                bv = red_v[w];
                bj = red_j[w];
            }
        }
        const double factor = 1.0 / sqrt(bv);
        s_factor = factor;
        if (a.gauge) {
#pragma unroll
            for (int k = 0; k < 9; ++k) a.gauge[k] = R0[k];
            a.gauge[9] = c00;
            a.gauge[10] = c01;
            a.gauge[11] = c02;
            a.gauge[12] = factor;
            a.gauge[13] = (double)bj;
            a.gauge[14] = 0.0;
            a.gauge[15] = 0.0;
        }
        *sh.counter = 0;
    }
    if (tid >= 32 && tid < 35) {   // camera 0: new moments, gauge-fixed pose, table row
        const int k = tid - 32;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t t = (h == 0 ? 0 : 3 * K) + k;
            if (a.rule.mode == 2 && t >= a.lo && t < a.hi) {
                a.s[t] = sh.stage[3 + 3 * h + k];
                a.r[t] = sh.stage[9 + 3 * h + k];
            }
        }
        a.cam[k] = 0.0;
        a.cam[3 * K + k] = sh.stage[k];
        if (a.T) {
#pragma unroll
            for (int q = 0; q < 3; ++q) a.T[3 * k + q] = sh.stage[15 + 3 * k + q];
            a.T[9 + k] = 0.0;
            if (k == 0) {
                a.T[12] = sh.stage[24];
                a.T[13] = sh.stage[25];
            }
        }
    }
    __syncthreads();
    // scale of the centres: 24 loads per thread in flight at once (the caches were just invalidated, so a batch
    // costs a round trip to memory; K = 2000 is one batch, the one requested above)
    const double factor = s_factor;
    if (a.scale_later) return;
    for (int base = 3; base < n3; base += 24 * 256) {
        if (base > 3) {
#pragma unroll
            for (int u = 0; u < 24; ++u) {
                const int idx = base + u * 256 + tid;
                v[u] = idx < n3 ? a.cam[idx] : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < 24; ++u) {
            const int idx = base + u * 256 + tid;
            if (idx < n3) {
                const double c = v[u] * factor;
                a.cam[idx] = c;
                const int jc = idx / 3;
                if (a.T) a.T[(int64_t)jc * CAM_ROW + 9 + (idx - 3 * jc)] = c;
            }
        }
    }
}

// --------------------------------------------------------------------------------------------
// The same step without any hand-over between workgroups (k_camera_step_free): the form the fused loop uses.
// In k_camera_step_quad every workgroup drains its stores, draws a ticket, and the last one reduces the workgroups'
// maxima, scales every centre and writes camera 0's staged results: 8 of that kernel's 15.5 us at 2 000 cameras
// (store drain, returning atomic, a dependent trip for the tail).  Here nothing waits for anything:
//   * the longest baseline is collected by an atomic max on the bit pattern of its square (non-negative doubles
//     order like unsigned integers) in word 14 of the gauge record, which the PREVIOUS step (or the host, for the
//     first one) left at zero; the readers of the record divide (gauge_factor);
//   * the centres are written unscaled: the scale stays pending on them exactly as the rigid motion stays pending on
//     the points, and the same record carries both (the next pass multiplies while it stages the camera rows, the
//     next step while it loads the centres: the products are the ones the scaled store would have held);
//   * camera 0, whose old values every workgroup reads, is written by workgroup 0 alone: the step writes new buffers
//     (inputs and outputs must not overlap), the rotations come from one camera table and go to the other, and
//     camera 0's camera-frame sums live in two copies used alternately (the step reads one and zeroes the other).
// It always finishes a pass (energy accumulators -> energy, d E / d phi_x; camera-frame sums -> d E / d C,
// d E / d Theta; energy mailbox).
// --------------------------------------------------------------------------------------------
struct FreeStep {
    const double *T_in = nullptr;       // camera table the pass used (rotations of cam_in)
    double *T_out = nullptr;            // table of the new parameters (rotations and cos/sin theta_z; centres unwritten)
    const double *cam0_in = nullptr;    // camera 0's six camera-frame sums of the pass
    double *cam0_zero = nullptr;        // the other copy: zeroed for the pass after this step
    const double *in_scale = nullptr;   // gauge record whose scale is still pending on cam_in's centres, or null
    double *next_gauge = nullptr;       // record the NEXT step will use: its word 14 is zeroed here (may be null)
};

__global__ void __launch_bounds__(256) k_camera_step_free(CameraStep a, FinalizeArgs fz, FreeStep fs) {
    __shared__ double sR0[9], sC0[3];
    __shared__ double red_e[2], red_best[4];
    const int64_t K = a.K;
    const int tid = threadIdx.x;
    const int quad = tid >> 2, comp = tid & 3;
    const bool is_c0 = quad == 63;
    const int64_t j = is_c0 ? 0 : 1 + (int64_t)blockIdx.x * 63 + quad;
    const bool live = j < K && comp < 3;
    const bool own = live && !is_c0;                          // writes its results
    const bool own0 = live && is_c0 && blockIdx.x == 0;       // camera 0: workgroup 0 writes
    const bool writes = own || own0;
    const int64_t jj = j < K ? j : 0;
    const int cc = comp < 3 ? comp : 0;

    // ---- everything this lane reads, requested at once
    double in_sums[6], in_row[8];
    {
        const double *row = fs.T_in + jj * CAM_ROW;
#pragma unroll
        for (int k = 0; k < 6; ++k) in_sums[k] = is_c0 ? fs.cam0_in[k] : fz.cam_sums[k * K + jj];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            in_row[k] = row[3 * cc + k];
            in_row[3 + k] = row[3 + k];
        }
        in_row[6] = row[12];
        in_row[7] = row[13];
    }
    const double in_factor = fs.in_scale ? gauge_factor(fs.in_scale) : 1.0;
    double in_x[2], in_s[2] = {0.0, 0.0}, in_r[2] = {0.0, 0.0};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int64_t t = (h == 0 ? 0 : 3 * K) + 3 * jj + cc;
        in_x[h] = a.cam_in[t];
        if (a.rule.mode == 2 && t >= a.lo && t < a.hi) {
            in_s[h] = a.s_in[t];
            in_r[h] = a.r_in[t];
        }
    }
    in_x[0] *= in_factor;

    // ---- workgroup 0: the pass's energy accumulators -> energy and d E / d phi_x, mailbox, phi_x and its scalars
    if (blockIdx.x == 0) {
        const double dphi_old = a.scalars[SC_DPHI];
        if (tid < 64) {
            const double2 v = tid < fz.n_partials ? ((const double2 *)fz.partials)[tid] : make_double2(0.0, 0.0);
            if (tid < fz.n_partials) ((double2 *)fz.partials)[tid] = make_double2(0.0, 0.0);
            const double e = wave_sum(v.x), ps = wave_sum(v.y);
            if (tid == 0) {
                red_e[0] = e;
                red_e[1] = ps;
                fz.g_out[6 * K + 1] = e;
                fz.g_out[6 * K] = dphi_old * ps;
                if (fz.mailbox) {
                    fz.mailbox[0] = e;
                    __threadfence_system();
                    fz.mailbox[1] = fz.seq;
                }
            }
        }
        __syncthreads();
        if (tid == 64) {
            const int64_t t = 6 * K;
            double ph = a.cam_in[t];
            if (a.rule.mode && t >= a.lo && t < a.hi) {
                double sv = 0.0, rv = 0.0;
                if (a.rule.mode == 2) {
                    sv = a.s_in[t];
                    rv = a.r_in[t];
                }
                ph = apply_rule(a.rule, ph, dphi_old * red_e[1], &sv, &rv);
                if (a.rule.mode == 2) {
                    a.s[t] = sv;
                    a.r[t] = rv;
                }
            } else if (a.rule.mode == 2) {
                a.s[t] = a.s_in[t];
                a.r[t] = a.r_in[t];
            }
            a.cam[t] = ph;
            const double f = a.W / 2.0 / tan(ph / 2.0);
            a.scalars[SC_F] = f;
            a.scalars[SC_CX] = a.W / 2.0;
            a.scalars[SC_CY] = a.H / 2.0;
            a.scalars[SC_DPHI] = -f / sin(ph);
        }
        if (tid == 65 && fs.next_gauge) fs.next_gauge[14] = 0.0;
        if (tid >= 66 && tid < 72) fs.cam0_zero[tid - 66] = 0.0;
    }

    // ---- un-normalised gradient of C[comp], Theta[comp] of this quad's camera
    const double s0 = in_sums[0], s1 = in_sums[1], s2 = in_sums[2], w0 = in_sums[3], w1 = in_sums[4], w2 = in_sums[5];
    const double gc = -(in_row[0] * s0 + in_row[1] * s1 + in_row[2] * s2);
    const double gt = cc == 0 ? -(in_row[6] * w0 - in_row[7] * w1) : (cc == 1 ? -(in_row[3] * w0 + in_row[4] * w1 + in_row[5] * w2) : -w2);
    if (writes) {
        fz.g_out[3 * jj + cc] = gc;
        fz.g_out[3 * K + 3 * jj + cc] = gt;
    }
    if (own) {      // the sums are consumed: leave them at zero for the next pass
        fz.cam_sums[cc * K + jj] = 0.0;
        fz.cam_sums[(3 + cc) * K + jj] = 0.0;
    }

    // ---- update rule (p2w:704-708) on the two parameters of this lane
    double pnew[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int64_t t = (h == 0 ? 0 : 3 * K) + 3 * jj + cc;
        double xn = in_x[h], sn_ = in_s[h], rn_ = in_r[h];
        const bool free_t = a.rule.mode && t >= a.lo && t < a.hi;
        if (free_t) xn = apply_rule(a.rule, in_x[h], h == 0 ? gc : gt, &sn_, &rn_);
        if (writes && a.rule.mode == 2) {
            if (free_t) {
                a.s[t] = sn_;
                a.r[t] = rn_;
            } else {        // outside the free slice the moments are carried over
                a.s[t] = a.s_in[t];
                a.r[t] = a.r_in[t];
            }
        }
        pnew[h] = xn;
    }
    double sn, cs;
    sincos(pnew[1], &sn, &cs);
    const double sx = quad_bcast<0>(sn), cx = quad_bcast<0>(cs), sy = quad_bcast<1>(sn), cy = quad_bcast<1>(cs),
                 sz = quad_bcast<2>(sn), cz = quad_bcast<2>(cs);
    const double c0v = quad_bcast<0>(pnew[0]), c1v = quad_bcast<1>(pnew[0]), c2v = quad_bcast<2>(pnew[0]);
    double R[9];   // Ry Rx Rz (p2w:114-140)
    {
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
    }
    if (is_c0 && comp == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) sR0[k] = R[k];
        sC0[0] = c0v;
        sC0[1] = c1v;
        sC0[2] = c2v;
    }
    __syncthreads();
    double R0[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) R0[k] = sR0[k];
    const double c00 = sC0[0], c01 = sC0[1], c02 = sC0[2];

    // ---- gauge fix of this camera (p2w:270-333): centre (C - C0) @ R0, rotation R0^T R
    const double d0 = c0v - c00, d1 = c1v - c01, d2 = c2v - c02;
    const double n0 = d0 * R0[0] + d1 * R0[3] + d2 * R0[6];
    const double n1 = d0 * R0[1] + d1 * R0[4] + d2 * R0[7];
    const double n2 = d0 * R0[2] + d1 * R0[5] + d2 * R0[8];
    double Rn[9];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int q = 0; q < 3; ++q) Rn[3 * p + q] = R0[p] * R[q] + R0[3 + p] * R[3 + q] + R0[6 + p] * R[6 + q];
    // get_rotations (p2w:260-266): theta_x = atan2(-R12, Cx), theta_y = atan2(R02 / Cx, R22 / Cx), theta_z =
    // atan2(R10 / Cx, R11 / Cx) with Cx = cos(theta_x) = |(R10, R11)| > 0.  The common factor 1 / Cx does not change
    // an angle, so the two quotients per angle are left out, and Cx comes from the Newton reciprocal square root
    // (this kernel is one chain of dependent fp64 routines per camera: the square root and the four divisions were
    // 100 instructions of it)
    const double c2 = Rn[3] * Rn[3] + Rn[4] * Rn[4];
    const double iCx = fast_rsqrt(c2), Cx = c2 * iCx;
    const double ay = cc == 0 ? -Rn[5] : (cc == 1 ? Rn[2] : Rn[3]);
    const double ax = cc == 0 ? Cx : (cc == 1 ? Rn[8] : Rn[4]);
    const double th_g = atan2(ay, ax);
    const double n_own = cc == 0 ? n0 : (cc == 1 ? n1 : n2);
    if (writes) {
        a.cam[3 * jj + cc] = is_c0 ? 0.0 : n_own;      // the scale of the record below stays pending on the centres
        a.cam[3 * K + 3 * jj + cc] = th_g;
        if (a.R_out) {
#pragma unroll
            for (int k = 0; k < 3; ++k) a.R_out[9 * jj + 3 * cc + k] = Rn[3 * cc + k];
        }
        double *row = fs.T_out + jj * CAM_ROW;
#pragma unroll
        for (int k = 0; k < 3; ++k) row[3 * cc + k] = Rn[3 * cc + k];
        if (cc == 0) {
            row[12] = Rn[4] * iCx;  // cos(theta_z), p2w:674
            row[13] = Rn[3] * iCx;  // sin(theta_z), p2w:675
        }
    }
    if (own0 && cc == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) a.gauge[k] = R0[k];
        a.gauge[9] = c00;
        a.gauge[10] = c01;
        a.gauge[11] = c02;
        a.gauge[12] = 0.0;
        a.gauge[13] = -1.0;     // the camera at the far end of the longest baseline is not tracked here
        a.gauge[15] = 1.0;      // the factor is 1 / sqrt(word 14)
    }

    // ---- longest baseline from camera 0 (p2w:322-326): maximum of the squared lengths, one atomic per workgroup
    // (atomics on one address are served one after the other at the memory side, some 16 ns each: one per wave is
    // 2 us at 2 000 cameras and was 20 us at 20 000)
    double best = wave_max_nonneg((own && cc == 0) ? n0 * n0 + n1 * n1 + n2 * n2 : 0.0);
    if ((tid & 63) == 0) red_best[tid >> 6] = best;
    __syncthreads();
    if (tid == 0) {
        best = fmax(fmax(red_best[0], red_best[1]), fmax(red_best[2], red_best[3]));
        __hip_atomic_fetch_max((unsigned long long *)(a.gauge + 14), (unsigned long long)__double_as_longlong(best),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Scale of the camera centres by the factor of a gauge record, for camera counts at which one workgroup (the
// tail of k_camera_step_quad) would spend tens of microseconds on it: cam[3 .. 3K) and, if T is given, the rows'
// centres.
__global__ void __launch_bounds__(256) k_scale_centres(int64_t K, double *__restrict__ cam, double *__restrict__ T,
                                                       const double *__restrict__ gauge) {
    const int64_t idx = 3 + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= 3 * K) return;
    const double c = cam[idx] * gauge_factor(gauge);
    cam[idx] = c;
    if (T) {
        const int64_t jc = idx / 3;
        T[jc * CAM_ROW + 9 + (idx - 3 * jc)] = c;
    }
}

// out = in with the centres (entries 3 .. 3K) multiplied by the record's factor; the rest is copied
__global__ void __launch_bounds__(256) k_scale_centres_copy(int64_t K, const double *__restrict__ in,
                                                            const double *__restrict__ gauge, double *__restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx > 6 * K) return;
    const double v = in[idx];
    out[idx] = (idx >= 3 && idx < 3 * K) ? v * gauge_factor(gauge) : v;
}

// X_out = ((X - C0) @ R0) * factor; X_out may be X
__global__ void k_gauge_points(int64_t N, const double *X, double *X_out, const double *__restrict__ gauge) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const double d0 = X[3 * i + 0] - gauge[9], d1 = X[3 * i + 1] - gauge[10], d2 = X[3 * i + 2] - gauge[11];
    const double factor = gauge_factor(gauge);
    X_out[3 * i + 0] = (d0 * gauge[0] + d1 * gauge[3] + d2 * gauge[6]) * factor;
    X_out[3 * i + 1] = (d0 * gauge[1] + d1 * gauge[4] + d2 * gauge[7]) * factor;
    X_out[3 * i + 2] = (d0 * gauge[2] + d1 * gauge[5] + d2 * gauge[8]) * factor;
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
    // one pair of atomics per workgroup (those of one address are served one after the other at the memory side)
    __shared__ double part[2][4];
    s = wave_sum(s);
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) {
        part[0][threadIdx.x >> 6] = s;
        part[1][threadIdx.x >> 6] = c;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        const double *v = part[threadIdx.x];
        __hip_atomic_fetch_add(sums + threadIdx.x, (v[0] + v[1]) + (v[2] + v[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

static int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace ba

using namespace ba;

int ba::camera_table(ba_problem *p, const double *C, const double *Theta, const double *phi, hipStream_t st) {
    p->table_centres_stale = 0;
    hipLaunchKernelGGL(k_camera_table, dim3(cdiv(p->K, 256)), dim3(256), 0, st, p->K, C, Theta, phi, p->W, p->H,
                       p->cam_table, p->scalars);
    BA_HIP(hipGetLastError());
    return BA_OK;
}

// Regroups every tile's observations into camera runs (see ba_runpass.h).
static int build_runs(ba_problem *p) {
    const int nt = p->n_tiles;
    int *max_obs = p->flag;
    BA_HIP(hipMemset(max_obs, 0, 4));
    BA_HIP(hipMalloc(&p->run_ptr, sizeof(int64_t) * (nt + 1)));
    hipLaunchKernelGGL(k_run_count, dim3(nt), dim3(TILE_BLOCK), sizeof(int) * p->cam_cap, 0, p->opack,
                       (const TileMeta *)p->tile_meta, p->run_ptr, max_obs);
    BA_HIP(hipGetLastError());
    int h_max_obs = 0;
    BA_HIP(hipMemcpy(&h_max_obs, max_obs, 4, hipMemcpyDeviceToHost));
    const int n_scan_tiles = cdiv(nt, SCAN_TILE);
    int64_t *tile_sums = nullptr;
    BA_HIP(hipMalloc(&tile_sums, sizeof(int64_t) * (n_scan_tiles + 1)));
    hipLaunchKernelGGL(k_scan_tiles, dim3(n_scan_tiles), dim3(1024), 0, 0, p->run_ptr, (int64_t)nt, p->run_ptr, tile_sums);
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(1), dim3(1024), 0, 0, tile_sums, n_scan_tiles);
    hipLaunchKernelGGL(k_scan_add, dim3(cdiv(nt, 256)), dim3(256), 0, 0, p->run_ptr, (int64_t)nt, tile_sums, n_scan_tiles);
    hipError_t e = hipMemcpy(&p->total_runs, p->run_ptr + nt, sizeof(int64_t), hipMemcpyDeviceToHost);
    hipFree(tile_sums);
    BA_HIP(e);
    const size_t fill_lds = sizeof(int) * (2 * (size_t)p->cam_cap + 1) + sizeof(uint32_t) * (size_t)h_max_obs;
    if (fill_lds > 64 * 1024 || p->total_runs == 0 || p->cam_cap > RUN_MAX_CAMS || p->pt_cap > 65000) {   // leave the point-major tiles in charge
        hipFree(p->run_ptr);
        p->run_ptr = nullptr;
        p->total_runs = 0;
        return BA_OK;
    }
    const size_t n_rec = (size_t)p->total_runs * RUN_LEN;
    if (p->compact) {
        BA_HIP(hipMalloc(&p->rrec, sizeof(uint2) * n_rec));
    } else {
        BA_HIP(hipMalloc(&p->rpack, sizeof(uint32_t) * n_rec));
        BA_HIP(hipMalloc(&p->ruv, sizeof(double2) * n_rec));
    }
    p->run_pt_stride = (p->pt_cap + RUN_LEN - 1) / RUN_LEN;
    while ((p->run_pt_stride & 3) != 2) ++p->run_pt_stride;   // 2 * odd, see pt_swz
    hipLaunchKernelGGL(k_run_fill, dim3(nt), dim3(TILE_BLOCK), sizeof(int) * run_fill_lds_ints(p->cam_cap), 0, p->opack, p->ouv,
                       (const TileMeta *)p->tile_meta, p->run_ptr, p->cam_cap, p->run_pt_stride,
                       getenv("BA_RUN_BATCHES_OF_8") ? 0 : 1, p->rpack, p->ruv, p->rrec);
    BA_HIP(hipGetLastError());
    // LDS accumulator space for the camera sums: two copies at the largest tile (one if that is too much), plus
    // whatever the workgroup's LDS allocation leaves unused: the LDS of a CU (160 KB) is handed out in granules of
    // 1280 bytes, so a workgroup can grow to the largest size that keeps the number of workgroups per CU
    const size_t budget = 64 * 1024;
    auto pick_slots = [&](int ncv, int npv, size_t limit, const char *env_name, size_t wg_per_cu_regs) {
        int rep = 2;
        if (const char *env = getenv(env_name)) rep = atoi(env);
        if (rep < 1) rep = 1;
        while (rep > 1 && sizeof(double) * run_pass_lds_doubles(ncv, npv, p->cam_cap, p->run_pt_stride, p->cam_cap * rep) > limit)
            rep >>= 1;
        int slots = p->cam_cap * rep;
        const size_t lds_cu = 160 * 1024, granule = 1280, fixed = 256;     // fixed: the kernel's static arrays
        const size_t base = sizeof(double) * run_pass_lds_doubles(ncv, npv, p->cam_cap, p->run_pt_stride, slots) + fixed;
        // workgroups a CU holds: by LDS, and by the registers of the kernel (four waves each)
        const size_t per_cu = std::min(lds_cu / base, wg_per_cu_regs);
        if (per_cu >= 1 && !getenv("BA_NO_LDS_SLACK")) {
            const size_t room = (lds_cu / per_cu) / granule * granule;
            if (room > base) slots += (int)((room - base) / (sizeof(double) * ncv));
        }
        return slots;
    };
    p->run_slots_grad = pick_slots(6, 3, budget / 2, "BA_CAM_REP", 1000);      // by LDS alone: more copies than that measured slower
    p->run_slots_jtj = pick_slots(27, 9, 78 * 1024, "BA_CAM_REP_JTJ", p->compact ? 3 : 2);   // two workgroups per CU at least
    if (getenv("BA_VERBOSE"))
        fprintf(stderr, "[ba] runs: records=%s total=%lld (%.3f x nV/8) pt_stride=%d accumulator columns grad=%d jtj=%d (largest tile: %d cameras)\n",
                p->compact ? "i16" : "f64", (long long)p->total_runs, (double)p->total_runs * RUN_LEN / (double)p->nV, p->run_pt_stride,
                p->run_slots_grad, p->run_slots_jtj, p->cam_cap);
    return BA_OK;
}

// Lays the tiles out as dense blocks as well (see ba_blockpass.h) when that costs little padding.
static int build_blocks(ba_problem *p) {
    const int nt = p->n_tiles;
    BA_HIP(hipMalloc(&p->blk_ptr, sizeof(int64_t) * (nt + 1)));
    hipLaunchKernelGGL(k_block_count, dim3(cdiv(nt, 256)), dim3(256), 0, 0, (const TileMeta *)p->tile_meta, nt, p->blk_ptr);
    const int n_scan_tiles = cdiv(nt, SCAN_TILE);
    int64_t *tile_sums = nullptr;
    BA_HIP(hipMalloc(&tile_sums, sizeof(int64_t) * (n_scan_tiles + 1)));
    hipLaunchKernelGGL(k_scan_tiles, dim3(n_scan_tiles), dim3(1024), 0, 0, p->blk_ptr, (int64_t)nt, p->blk_ptr, tile_sums);
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(1), dim3(1024), 0, 0, tile_sums, n_scan_tiles);
    hipLaunchKernelGGL(k_scan_add, dim3(cdiv(nt, 256)), dim3(256), 0, 0, p->blk_ptr, (int64_t)nt, tile_sums, n_scan_tiles);
    hipError_t e = hipMemcpy(&p->total_units, p->blk_ptr + nt, sizeof(int64_t), hipMemcpyDeviceToHost);
    hipFree(tile_sums);
    BA_HIP(e);
    const double fill = p->total_units > 0 ? (double)p->nV / ((double)p->total_units * BLK_UNIT) : 0.0;
    double min_fill = 0.6;      // below this the padding costs more than the cheaper reductions save
    if (const char *env = getenv("BA_BLOCK_MIN_FILL")) min_fill = atof(env);
    p->blk_fill = fill;
    if (fill < min_fill) {
        hipFree(p->blk_ptr);
        p->blk_ptr = nullptr;
        p->total_units = 0;
        return BA_OK;
    }
    const int64_t n_rec = p->total_units * BLK_UNIT;
    if (p->compact) BA_HIP(hipMalloc(&p->brec, sizeof(uint32_t) * n_rec));
    else BA_HIP(hipMalloc(&p->buv, sizeof(double2) * n_rec));
    hipLaunchKernelGGL(k_block_clear, dim3(cdiv(n_rec, 256)), dim3(256), 0, 0, n_rec, p->brec, p->buv);
    hipLaunchKernelGGL(k_block_fill, dim3(nt), dim3(TILE_BLOCK), 0, 0, p->opack, p->ouv, (const TileMeta *)p->tile_meta,
                       p->blk_ptr, p->brec, p->buv);
    BA_HIP(hipGetLastError());
    // LDS point capacity 8 S: S = 4 mod 8 (see ba_blockpass.h), at least 16 slots per chunk of the largest tile
    p->blk_pt_stride = 16 * ((p->pt_cap + BLK_PTS - 1) / BLK_PTS);
    while ((p->blk_pt_stride & 7) != 4) ++p->blk_pt_stride;
    const int nb_max = (p->cam_cap + BLK_CAMS - 1) / BLK_CAMS;
    p->blk_waves = nb_max < 4 ? nb_max : 4;
    if (getenv("BA_VERBOSE"))
        fprintf(stderr, "[ba] blocks: records=%s units=%lld fill=%.3f pt_stride=%d waves=%d\n", p->compact ? "i16" : "f64",
                (long long)p->total_units, fill, p->blk_pt_stride, p->blk_waves);
    return BA_OK;
}

// Cuts the observation list into LDS tiles (see k_eval_tiled).  The tile is halved until the
// camera rows, the points and both accumulator sets of the largest tile fit the LDS budget of a
// workgroup; scenes whose tiles would each need most of the cameras keep p->tiled == 0 and run
// the wave-chunk kernel instead.
static int build_tiles(ba_problem *p) {
    const int64_t K = p->K, nV = p->nV;
    const int n_words = (int)((K + 31) >> 5);
    const size_t lds_budget = 64 * 1024;
    int tobs = 3840;   // two camera runs per thread of the run pass (at most (2 * 256 - cameras of a tile) * 8 observations)
    if (const char *env = getenv("BA_TILE_OBS")) tobs = atoi(env);
    if (tobs < 64 || (size_t)n_words * 8 > lds_budget) return BA_OK;
    int *maxima = p->flag;  // 2 ints of the handle's flag buffer
    int nt = 0;
    for (;; tobs /= 2) {
        if (tobs < 64) return BA_OK;
        // Tile sizes are not uniform.  Head (BA_TILE_HEAD="count:size,count:size,..."): the first workgroups of a launch
        // start together; tiles of different sizes take them apart, so that later one workgroup's set-up and output
        // phases overlap with the others' main loops.  Tail (BA_TILE_TAIL=fraction of the observations): when the launch
        // runs out of tiles every CU is left with workgroups nothing overlaps with; the last observations are cut into
        // tiles of half the size.
        std::vector<int64_t> nominal;
        {
            double tail = 0.15;
            if (const char *env = getenv("BA_TILE_TAIL")) tail = atof(env);
            if (!(tail >= 0.0 && tail <= 1.0)) tail = 0.0;
            const bool graded = nV >= (int64_t)tobs * 1024;       // fewer tiles than workgroup slots: nothing to grade
            int64_t at = 0;
            if (graded)
                if (const char *env = getenv("BA_TILE_HEAD")) {
                    for (const char *q = env; *q && at < nV;) {
                        int count = 0, size = 0, used = 0;
                        if (sscanf(q, "%d:%d%n", &count, &size, &used) < 2 || size < 64 || size > tobs) break;
                        for (int k = 0; k < count && at < nV; ++k, at += size) nominal.push_back(at);
                        q += used;
                        if (*q == ',') ++q;
                    }
                }
            const int64_t tail_obs = graded ? (int64_t)(tail * (double)nV) : 0;
            for (; at < nV - tail_obs; at += tobs) nominal.push_back(at);
            for (; at < nV; at += tobs / 2) nominal.push_back(at);
        }
        nt = (int)nominal.size();
        hipFree(p->tile_obs);
        hipFree(p->tile_cam_ptr);
        p->tile_obs = nullptr;
        p->tile_cam_ptr = nullptr;
        BA_HIP(hipMalloc(&p->tile_obs, sizeof(int32_t) * (nt + 1)));
        BA_HIP(hipMalloc(&p->tile_cam_ptr, sizeof(int64_t) * (nt + 1)));
        BA_HIP(hipMemset(maxima, 0, 8));
        int64_t *d_nominal = nullptr;
        BA_HIP(hipMalloc(&d_nominal, sizeof(int64_t) * nominal.size()));
        if (hipError_t e = hipMemcpy(d_nominal, nominal.data(), sizeof(int64_t) * nominal.size(), hipMemcpyHostToDevice)) {
            hipFree(d_nominal);
            BA_HIP(e);
        }
        hipLaunchKernelGGL(k_tile_bounds, dim3(cdiv(nt + 1, 256)), dim3(256), 0, 0, p->oi, nV, nt, (const int64_t *)d_nominal,
                           p->tile_obs);
        hipLaunchKernelGGL(k_tile_count, dim3(nt), dim3(TILE_BLOCK), sizeof(unsigned) * n_words, 0, p->oi, p->oj,
                           p->tile_obs, (int)K, p->tile_cam_ptr, maxima);
        int hmax[2];
        hipError_t e_tiles = hipGetLastError();
        if (e_tiles == hipSuccess) e_tiles = hipMemcpy(hmax, maxima, 8, hipMemcpyDeviceToHost);
        hipFree(d_nominal);
        BA_HIP(e_tiles);
        int rep = 16;
        if (const char *env = getenv("BA_CAM_REP")) rep = atoi(env);
        size_t need = 0;
        for (;; rep >>= 1) {  // fewer accumulator copies rather than fewer workgroups per CU
            need = sizeof(double) * ((size_t)(ROW_LDS + 6 * rep) * hmax[0] + 6 * (size_t)hmax[1]);
            if (need <= lds_budget / 4 || rep <= 1) break;
        }
        if (need <= lds_budget && hmax[0] < 65536 && hmax[1] < 65536) {
            p->cam_rep = rep < 1 ? 1 : rep;
            p->n_tiles = nt;
            p->tile_obs_target = tobs;
            p->cam_cap = hmax[0];
            p->pt_cap = hmax[1];
            p->tile_lds_bytes = need;
            break;
        }
    }
    // exclusive scan of the per-tile camera counts -> offsets into cam_list
    const int n_scan_tiles = cdiv(nt, SCAN_TILE);
    int64_t *tile_sums = nullptr;
    BA_HIP(hipMalloc(&tile_sums, sizeof(int64_t) * (n_scan_tiles + 1)));
    hipLaunchKernelGGL(k_scan_tiles, dim3(n_scan_tiles), dim3(1024), 0, 0, p->tile_cam_ptr, (int64_t)nt, p->tile_cam_ptr,
                       tile_sums);
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(1), dim3(1024), 0, 0, tile_sums, n_scan_tiles);
    hipLaunchKernelGGL(k_scan_add, dim3(cdiv(nt, 256)), dim3(256), 0, 0, p->tile_cam_ptr, (int64_t)nt, tile_sums,
                       n_scan_tiles);
    int64_t total_cams = 0;
    hipError_t e = hipMemcpy(&total_cams, p->tile_cam_ptr + nt, sizeof(int64_t), hipMemcpyDeviceToHost);
    hipFree(tile_sums);
    BA_HIP(e);
    // tiles pay off only while fetching every tile's camera rows stays small next to the
    // observation stream - or when the scene is so small that bytes do not matter and the number of launches per
    // iteration does (the tile form fuses the point update into the pass: two launches instead of four; a scene of
    // 100 points x 300 frames: 28 -> 23 us per iteration)
    p->tiles_worth = getenv("BA_FORCE_TILES") || (double)total_cams * 128.0 <= 0.5 * 24.0 * (double)nV || nV <= 200000;
    {
        BA_HIP(hipMalloc(&p->cam_list, sizeof(int32_t) * (total_cams > 0 ? total_cams : 1)));
        BA_HIP(hipMalloc(&p->opack, sizeof(uint32_t) * nV));
        BA_HIP(hipMalloc(&p->tile_meta, sizeof(TileMeta) * nt));
        hipLaunchKernelGGL(k_tile_fill, dim3(nt), dim3(TILE_BLOCK), sizeof(unsigned) * 2 * n_words, 0, p->oj, p->tile_obs,
                           p->tile_cam_ptr, (int)K, p->oi, nV, p->N, p->cam_list, p->opack, (TileMeta *)p->tile_meta);
        BA_HIP(hipGetLastError());
        p->tiled = 1;
    }
    if (p->tiles_worth) {
        if (int rc = build_runs(p)) return rc;
        if (int rc = build_blocks(p)) return rc;
    }
    if (getenv("BA_VERBOSE"))
        fprintf(stderr, "[ba] tiles: tiled=%d worth=%d n_tiles=%d target_obs=%d cam_cap=%d pt_cap=%d rep=%d lds=%zu B cams/tile=%.1f\n",
                p->tiled, p->tiles_worth, p->n_tiles, p->tile_obs_target, p->cam_cap, p->pt_cap, p->cam_rep, p->tile_lds_bytes,
                nt ? (double)total_cams / nt : 0.0);
    return BA_OK;
}

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
    hipFree(p->tile_obs);
    hipFree(p->tile_cam_ptr);
    hipFree(p->cam_list);
    hipFree(p->opack);
    hipFree(p->tile_meta);
    hipFree(p->rpack);
    hipFree(p->rrec);
    hipFree(p->ruv);
    hipFree(p->run_ptr);
    hipFree(p->brec);
    hipFree(p->buv);
    hipFree(p->blk_ptr);
    hipFree(p->cam_table);
    hipFree(p->cam_table_alt);
    hipFree(p->sum_block);
    hipFree(p->scalars);
    hipFree(p->step_counter);
    hipFree(p->step_best);
    hipFree(p->step_stage);
    hipFree(p->cam_sums_jtj);
    hipFree(p->partials_jtj);
    hipFree(p->row_ptr);
    hipFree(p->schur_sub);
    hipFree(p->schur_rest);
    hipFree(p->flag);
    if (p->mailbox_host) hipHostFree((void *)p->mailbox_host);
    if (p->aux_stream) {
        hipStreamDestroy(p->aux_stream);
        hipEventDestroy(p->aux_fork);
        hipEventDestroy(p->aux_join);
    }
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
    BA_REQUIRE(nV < (1ll << 31) - 64 && N < (1ll << 31) / 3 && K < (1ll << 31) / 8,
               "ba_problem_create: shard too large for 32-bit indices (3 N, 8 K and nV must stay below 2^31)");
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
    hipLaunchKernelGGL(k_validate_obs, dim3(std::min(cdiv(nV, 256), 2048)), dim3(256), 0, 0, p->oi, p->oj, p->ouv, nV, N, K, p->flag,
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
    p->compact = !(h.flag & 4);
    if (const char *env = getenv("BA_RECORDS")) {   // "f64" forces the long records (diagnostics)
        if (!strcmp(env, "f64")) p->compact = 0;
    }

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

    if (int rc_tiles = build_tiles(p)) return fail(rc_tiles);

    BA_TRY(hipMalloc(&p->cam_table, sizeof(double) * K * CAM_ROW));
    BA_TRY(hipMalloc(&p->cam_table_alt, sizeof(double) * K * CAM_ROW));
    BA_TRY(hipMemset(p->cam_table_alt, 0, sizeof(double) * K * CAM_ROW));
    {
        // camera-frame sums (kept at zero between passes by the finalisation) and per-tile partial sums in one
        // block, so that several GPUs can add both up with one collective
        size_t n_part = (size_t)p->grid * EVAL_WAVES;
        if ((size_t)p->n_tiles > n_part) n_part = p->n_tiles;
        p->partials_cap = (int64_t)n_part;
        BA_TRY(hipMalloc(&p->sum_block, sizeof(double) * (K * 6 + SUM_EXTRA + 2 * n_part)));
        BA_TRY(hipMemset(p->sum_block, 0, sizeof(double) * (K * 6 + SUM_EXTRA + 2 * n_part)));
        p->cam_sums = p->sum_block;
        p->esum = p->sum_block + K * 6;
        p->cam0_sums = p->esum + 2 * ESUM_SLOTS;
        p->partials = p->sum_block + K * 6 + SUM_EXTRA;
    }
    {
        void *host = nullptr, *dev = nullptr;
        BA_TRY(hipHostMalloc(&host, 64, hipHostMallocMapped | hipHostMallocCoherent));
        memset(host, 0, 64);
        p->mailbox_host = (double *)host;
        BA_TRY(hipHostGetDevicePointer(&dev, host, 0));
        p->mailbox_dev = (double *)dev;
    }
    BA_TRY(hipMalloc(&p->scalars, sizeof(double) * SC_COUNT));
    BA_TRY(hipMalloc(&p->step_counter, 16));
    BA_TRY(hipMemset(p->step_counter, 0, 16));
    BA_TRY(hipMalloc(&p->step_best, sizeof(double) * 2 * (cdiv(K, 63) + 1)));
    BA_TRY(hipMalloc(&p->step_stage, sizeof(double) * STAGE_DOUBLES));
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

}  // extern "C"

static UpdateRule make_rule(int mode, double lr, double gscale, double bc1, double bc2);

// optional extras of the camera-run pass: pending gauge of X and the fused point update
struct FusedUpdate {
    const double *X_gauge = nullptr;
    const double *C_scale = nullptr;    // gauge record whose scale is pending on the centres in C
    UpdateRule rule = {0, 0.0, 0.0, 1.0, 1.0};
    const double *s_in = nullptr, *r_in = nullptr;
    double *X_out = nullptr, *s_out = nullptr, *r_out = nullptr;
};

static int eval_impl(ba_problem *p, const double *X, const double *C, const double *Theta, const double *phi, int flags,
                     double *E_sum, double *gX, double *gC, double *gTheta, double *gphi, const FusedUpdate *fu,
                     void *stream) {
    BA_REQUIRE(p, "ba_eval: null problem");
    BA_REQUIRE(X && C && Theta && phi && E_sum, "ba_eval: null parameter or E_sum pointer");
    BA_REQUIRE(!(flags & BA_EVAL_GRAD_PHI) || gphi, "ba_eval: BA_EVAL_GRAD_PHI needs gphi");
    BA_REQUIRE(!(flags & BA_EVAL_GRAD_X) || gX, "ba_eval: BA_EVAL_GRAD_X needs gX");
    BA_REQUIRE(!(flags & BA_EVAL_GRAD_CAM) || (gC && gTheta), "ba_eval: BA_EVAL_GRAD_CAM needs gC and gTheta");
    hipStream_t st = (hipStream_t)stream;
    const int64_t K = p->K;

    // a table whose rows lack the centres (merged camera step) serves the run pass only
    const bool runs_expected = p->tiled && p->tiles_worth && p->run_ptr && ((flags >> 6) & 7) == 0 &&
                               !(flags & (BA_EVAL_CAM_ATOMICS | BA_EVAL_UNTILED | BA_EVAL_POINT_MAJOR));
    if (fu && fu->C_scale)
        BA_REQUIRE(runs_expected && (flags & BA_EVAL_REUSE_TABLE),
                   "ba_eval_update: centres with a pending scale need the camera-run pass and the table their camera step left");
    if (!(flags & BA_EVAL_REUSE_TABLE) || (p->table_centres_stale && !runs_expected))
        if (int rc = camera_table(p, C, Theta, phi, st)) return rc;
    const bool want_gx = (flags & BA_EVAL_GRAD_X) != 0;
    const bool use_tiles = p->tiled && p->tiles_worth && !(flags & (BA_EVAL_CAM_ATOMICS | BA_EVAL_UNTILED));
    // the tiled pass writes every point's gradient itself; the wave-chunk pass only the observed ones
    if (want_gx && !use_tiles && p->n_points_seen < p->N)
        BA_HIP(hipMemsetAsync(gX, 0, sizeof(double) * 3 * p->N, st));

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
    a.debug = (flags >> 6) & 7;
    if (a.debug & 2) a.gX = nullptr;

    const size_t lds_bytes = sizeof(double) * 6 * (size_t)K;
    const bool use_lds = !(flags & BA_EVAL_CAM_ATOMICS) && lds_bytes <= (size_t)p->lds_limit;
    const bool timed = (flags & BA_EVAL_TIMED) != 0 && p->ev_count < TIMING_RING;
    int n_partials = p->grid * EVAL_WAVES;
    const bool use_runs = use_tiles && p->run_ptr && !(flags & BA_EVAL_POINT_MAJOR) && a.debug == 0;
    if (fu && !use_runs) {
        set_error("ba_eval_update: needs the camera-run layout of the observation pass");
        return BA_ERR_STATE;
    }
    // a timed launch hands its two events to the dispatch itself (hipExtLaunchKernelGGL): they receive the kernel's
    // own begin and end, and no event packets sit in the queue (recording events around the launch stalls it)
    hipEvent_t ev_b = timed ? p->ev_begin[p->ev_count] : nullptr, ev_e = timed ? p->ev_end[p->ev_count] : nullptr;
#define BA_LAUNCH_MAYBE_TIMED(kern, grid, block, lds, arg)                                   \
    do {                                                                                     \
        if (timed) hipExtLaunchKernelGGL(kern, grid, block, (uint32_t)(lds), st, ev_b, ev_e, 0, arg); \
        else hipLaunchKernelGGL(kern, grid, block, lds, st, arg);                            \
    } while (0)
    if (use_runs) {
        RunArgs ra;
        ra.rpack = p->rpack;
        ra.ruv = p->ruv;
        ra.rrec = p->rrec;
        ra.run_ptr = p->run_ptr;
        ra.meta = (const TileMeta *)p->tile_meta;
        ra.cam_list = p->cam_list;
        ra.K = (int)K;
        ra.cam_cap = p->cam_cap;
        ra.pt_stride = p->run_pt_stride;
        ra.cacc_slots = p->run_slots_grad;
        ra.X = X;
        ra.X_gauge = fu ? fu->X_gauge : nullptr;
        ra.upd = fu ? fu->rule : UpdateRule{0, 0.0, 0.0, 1.0, 1.0};
        ra.s_in = fu ? fu->s_in : nullptr;
        ra.r_in = fu ? fu->r_in : nullptr;
        ra.X_out = fu ? fu->X_out : nullptr;
        ra.s_out = fu ? fu->s_out : nullptr;
        ra.r_out = fu ? fu->r_out : nullptr;
        ra.T = p->cam_table;
        ra.C = C;
        ra.scalars = p->scalars;
        ra.out[0] = a.gX;
        ra.out[1] = ra.out[2] = nullptr;
        ra.cam_sums = p->cam_sums;
        ra.partials = p->partials;
        if (flags & BA_EVAL_DEFER_FINALIZE) {     // the camera step that finishes the pass reads these
            ra.esum = p->esum;
            p->cam0_parity ^= 1;
            ra.cam0 = p->cam0_sums + 6 * p->cam0_parity;
        }
        ra.C_scale = fu ? fu->C_scale : nullptr;
        n_partials = p->n_tiles;
        const size_t lds = sizeof(double) * run_pass_lds_doubles<GradOp>(p->cam_cap, p->run_pt_stride, p->run_slots_grad);
        const void *fn = p->compact ? (const void *)k_run_pass<GradOp, RecI16> : (const void *)k_run_pass<GradOp, RecF64>;
        if (lds > 64 * 1024) BA_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (p->compact) BA_LAUNCH_MAYBE_TIMED((k_run_pass<GradOp, RecI16>), dim3(p->n_tiles), dim3(TILE_BLOCK), lds, ra);
        else BA_LAUNCH_MAYBE_TIMED((k_run_pass<GradOp, RecF64>), dim3(p->n_tiles), dim3(TILE_BLOCK), lds, ra);
    } else if (use_tiles) {
        TileArgs ta;
        ta.opack = p->opack;
        ta.ouv = p->ouv;
        ta.meta = (const TileMeta *)p->tile_meta;
        ta.cam_list = p->cam_list;
        ta.n_tiles = p->n_tiles;
        ta.K = (int)K;
        ta.cam_cap = p->cam_cap;
        ta.pt_cap = p->pt_cap;
        ta.cam_rep = p->cam_rep;
        ta.X = X;
        ta.T = p->cam_table;
        ta.scalars = p->scalars;
        ta.gX = a.gX;
        ta.cam_sums = p->cam_sums;
        ta.partials = p->partials;
        ta.debug = a.debug;
        n_partials = p->n_tiles;
        BA_LAUNCH_MAYBE_TIMED(k_eval_tiled, dim3(p->n_tiles), dim3(TILE_BLOCK), p->tile_lds_bytes, ta);
    } else if (use_lds) {
        if (lds_bytes > 64 * 1024)
            BA_HIP(hipFuncSetAttribute((const void *)k_eval<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds_bytes));
        BA_LAUNCH_MAYBE_TIMED(k_eval<true>, dim3(p->grid), dim3(EVAL_BLOCK), lds_bytes, a);
    } else {
        BA_LAUNCH_MAYBE_TIMED(k_eval<false>, dim3(p->grid), dim3(EVAL_BLOCK), 0, a);
    }
#undef BA_LAUNCH_MAYBE_TIMED
    BA_HIP(hipGetLastError());
    if (timed) p->ev_count++;
    const bool want_cam = (flags & BA_EVAL_GRAD_CAM) != 0;
    if (flags & BA_EVAL_DEFER_FINALIZE) {   // ba_camera_step_finalize finishes the pass
        BA_REQUIRE(want_cam && (flags & BA_EVAL_GRAD_PHI), "ba_eval: BA_EVAL_DEFER_FINALIZE needs the camera and phi gradients");
        BA_REQUIRE(use_runs, "ba_eval: BA_EVAL_DEFER_FINALIZE needs the camera-run layout of the observation pass");
        p->pending_partials = ESUM_SLOTS;        // the pairs of energy accumulators
        return BA_OK;
    }
    double *mailbox = nullptr;
    double seq = 0.0;
    if (flags & BA_EVAL_POST) {
        mailbox = p->mailbox_dev;
        seq = (double)++p->post_seq;
    }
    hipLaunchKernelGGL(k_finalize, dim3(cdiv(K, 256)), dim3(256), 0, st, K, n_partials, p->partials, p->cam_sums,
                       p->cam_table, p->scalars, E_sum, (flags & BA_EVAL_GRAD_PHI) ? gphi : nullptr,
                       want_cam ? gC : nullptr, want_cam ? gTheta : nullptr, mailbox, seq);
    BA_HIP(hipGetLastError());
    return BA_OK;
}

extern "C" {

int ba_eval(ba_problem *p, const double *X, const double *C, const double *Theta, const double *phi, int flags,
            double *E_sum, double *gX, double *gC, double *gTheta, double *gphi, void *stream) {
    return eval_impl(p, X, C, Theta, phi, flags, E_sum, gX, gC, gTheta, gphi, nullptr, stream);
}

int ba_can_fuse_update(const ba_problem *p) { return p && p->tiled && p->tiles_worth && p->run_ptr ? 1 : 0; }

int ba_eval_update(ba_problem *p, const double *X, const double *X_gauge, const double *C, const double *C_gauge,
                   const double *Theta, const double *phi, int flags, double *E_sum, double *gX, double *gC,
                   double *gTheta, double *gphi, int rule, double lr, double gscale, double bc1, double bc2,
                   const double *s_in, double *s_out, const double *r_in, double *r_out, double *X_out, void *stream) {
    BA_REQUIRE(p, "ba_eval_update: null problem");
    BA_REQUIRE(rule == 1 || rule == 2, "ba_eval_update: rule must be 1 (SGD) or 2 (Adam)");
    BA_REQUIRE(X_out && (rule != 2 || (s_in && s_out && r_in && r_out)), "ba_eval_update: null pointer");
    BA_REQUIRE(X_out != X || !X_gauge, "ba_eval_update: a pending gauge needs X_out different from X");
    BA_REQUIRE(!(flags & (BA_EVAL_CAM_ATOMICS | BA_EVAL_UNTILED | BA_EVAL_POINT_MAJOR | BA_EVAL_DEBUG_MASK)),
               "ba_eval_update: only the camera-run pass can fuse the update");
    FusedUpdate fu;
    fu.X_gauge = X_gauge;
    fu.C_scale = C_gauge;
    fu.rule = make_rule(rule, lr, gscale, bc1, bc2);
    fu.s_in = s_in;
    fu.r_in = r_in;
    fu.X_out = X_out;
    fu.s_out = s_out;
    fu.r_out = r_out;
    const int f = gX ? (flags | BA_EVAL_GRAD_X) : (flags & ~BA_EVAL_GRAD_X);   // gX is optional here
    return eval_impl(p, X, C, Theta, phi, f, E_sum, gX, gC, gTheta, gphi, &fu, stream);
}

int ba_gauge_points(int64_t N, const double *X_in, double *X_out, const double *gauge, void *stream) {
    BA_REQUIRE(N >= 0 && (N == 0 || (X_in && X_out)) && gauge, "ba_gauge_points: null pointer");
    if (N > 0) hipLaunchKernelGGL(k_gauge_points, dim3(cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, N, X_in, X_out, gauge);
    BA_HIP(hipGetLastError());
    return BA_OK;
}

int ba_post_energy(ba_problem *p, const double *E_sum, void *stream) {
    BA_REQUIRE(p && E_sum, "ba_post_energy: null pointer");
    const double seq = (double)++p->post_seq;
    hipLaunchKernelGGL(k_post_energy, dim3(1), dim3(64), 0, (hipStream_t)stream, E_sum, p->mailbox_dev, seq);
    BA_HIP(hipGetLastError());
    return BA_OK;
}

int ba_wait_energy(ba_problem *p, double *h_E_sum, void *stream) {
    BA_REQUIRE(p && h_E_sum, "ba_wait_energy: null pointer");
    BA_REQUIRE(p->post_seq > 0, "ba_wait_energy: nothing was posted");
    volatile double *box = p->mailbox_host;
    const double want = (double)p->post_seq;
    const auto t0 = std::chrono::steady_clock::now();
    bool drained = false;
    for (uint64_t spin = 1;; ++spin) {
        if (box[1] == want) break;
        __builtin_ia32_pause();
        if ((spin & 0xffff) == 0) {
            // the stream ran dry without the post arriving, or nothing for a long time: a kernel
            // failed; report instead of spinning forever
            const hipError_t q = hipStreamQuery((hipStream_t)stream);
            if (q != hipSuccess && q != hipErrorNotReady) return hip_fail(q, "hipStreamQuery", __FILE__, __LINE__);
            if (q == hipSuccess) {
                if (drained) {
                    set_error("ba_wait_energy: the stream finished without posting the energy");
                    return BA_ERR_HIP;
                }
                drained = true;   // one more look at the mailbox after the stream drained
            }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) {
                set_error("ba_wait_energy: timed out");
                return BA_ERR_HIP;
            }
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    *h_E_sum = box[0];
    return BA_OK;
}

int ba_eval_update_arm(ba_problem *p, const double *X, const double *X_gauge, const double *C, const double *C_gauge,
                       const double *Theta, const double *phi, int flags, double *E_sum, double *gX, double *gC,
                       double *gTheta, double *gphi, int rule, double lr, double gscale, double bc1, double bc2,
                       const double *s_in, double *s_out, const double *r_in, double *r_out, double *X_out,
                       double *X_out_alt) {
    BA_REQUIRE(p, "ba_eval_update_arm: null problem");
    BA_REQUIRE(rule == 1 || rule == 2, "ba_eval_update_arm: rule must be 1 (SGD) or 2 (Adam)");
    BA_REQUIRE(X && C && Theta && phi && X_out && (rule != 2 || (s_in && s_out && r_in && r_out)),
               "ba_eval_update_arm: null pointer");
    BA_REQUIRE(ba_can_fuse_update(p), "ba_eval_update_arm: needs the camera-run layout of the observation pass");
    ba_problem::ArmedUpdate &a = p->armed_update;
    a.X = X, a.X_gauge = X_gauge, a.C = C, a.C_gauge = C_gauge, a.Theta = Theta, a.phi = phi, a.flags = flags;
    a.E_sum = E_sum, a.gX = gX, a.gC = gC, a.gTheta = gTheta, a.gphi = gphi;
    a.rule = rule, a.lr = lr, a.gscale = gscale, a.bc1 = bc1, a.bc2 = bc2;
    a.s_in = s_in, a.s_out = s_out, a.r_in = r_in, a.r_out = r_out, a.X_out = X_out, a.X_out_alt = X_out_alt;
    a.armed = true;
    return BA_OK;
}

int ba_wait_energy_fire(ba_problem *p, double *h_E_sum, double divisor, double lo, double hi, double alt_lo,
                        int *h_fired, void *stream) {
    BA_REQUIRE(p && h_E_sum && h_fired, "ba_wait_energy_fire: null pointer");
    *h_fired = 0;
    ba_problem::ArmedUpdate a = p->armed_update;
    p->armed_update.armed = false;
    const int rc = ba_wait_energy(p, h_E_sum, stream);
    if (rc != BA_OK) return rc;
    const double e = *h_E_sum / divisor;      // the caller's own arithmetic: its schedule sees exactly this value
    if (!a.armed || !(e >= lo)) return BA_OK;                 // a NaN energy fails every comparison
    int which = 0;
    if (e < hi) which = 1;
    else if (a.X_out_alt && e >= alt_lo) which = 2;
    if (!which) return BA_OK;
    const int rl = ba_eval_update(p, a.X, a.X_gauge, a.C, a.C_gauge, a.Theta, a.phi, a.flags, a.E_sum, a.gX, a.gC, a.gTheta, a.gphi,
                                  a.rule, a.lr, a.gscale, a.bc1, a.bc2, a.s_in, a.s_out, a.r_in, a.r_out,
                                  which == 1 ? a.X_out : a.X_out_alt, stream);
    if (rl != BA_OK) return rl;
    *h_fired = which;
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

static UpdateRule make_rule(int mode, double lr, double gscale, double bc1, double bc2) {
    UpdateRule u;
    u.mode = mode;
    u.lr = lr;
    u.gscale = gscale;
    u.ibc1 = 1.0 / bc1;
    u.ibc2 = 1.0 / bc2;
    return u;
}

int ba_adam_step(int64_t n, double *x, const double *g, double *s, double *r, double lr, double gscale, double bc1,
                 double bc2, double *release, void *stream) {
    BA_REQUIRE(n >= 0 && (n == 0 || (x && g && s && r)), "ba_adam_step: null pointer");
    if (n == 0) return BA_OK;
    int grid = cdiv(n, 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(k_update, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, x, g, s, r, release,
                       make_rule(2, lr, gscale, bc1, bc2));
    BA_HIP(hipGetLastError());
    return BA_OK;
}

int ba_sgd_step(int64_t n, double *x, const double *g, double lr, double gscale, double *release, void *stream) {
    BA_REQUIRE(n >= 0 && (n == 0 || (x && g)), "ba_sgd_step: null pointer");
    if (n == 0) return BA_OK;
    int grid = cdiv(n, 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(k_update, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, x, g, (double *)nullptr,
                       (double *)nullptr, release, make_rule(1, lr, gscale, 1.0, 1.0));
    BA_HIP(hipGetLastError());
    return BA_OK;
}

// true when [a, a+n) and [b, b+n) are the same range or do not overlap
static bool same_or_disjoint(const double *a, const double *b, int64_t n) {
    return a == b || a + n <= b || b + n <= a;
}

int ba_points_step(int64_t N, const double *X_in, double *X_out, const double *gX, const double *s_in, double *s_out,
                   const double *r_in, double *r_out, int rule, double lr, double gscale, double bc1, double bc2,
                   const double *gauge, void *stream) {
    BA_REQUIRE(N >= 0 && rule >= 0 && rule <= 2, "ba_points_step: bad size or rule");
    if (N == 0) return BA_OK;
    BA_REQUIRE(X_in && X_out && (rule == 0 || gX) && (rule != 2 || (s_in && s_out && r_in && r_out)),
               "ba_points_step: null pointer");
    BA_REQUIRE(same_or_disjoint(X_in, X_out, 3 * N) && (rule != 2 || (same_or_disjoint(s_in, s_out, 3 * N) &&
                                                                       same_or_disjoint(r_in, r_out, 3 * N))),
               "ba_points_step: an output must be its input or not overlap it");
    if (rule == 0 && !gauge && X_in == X_out) return BA_OK;
    hipLaunchKernelGGL(k_points_step, dim3(cdiv(N, PS_POINTS)), dim3(PS_POINTS), 0, (hipStream_t)stream, N, X_in,
                       X_out, gX, s_in, s_out, r_in, r_out, make_rule(rule, lr, gscale, bc1, bc2), gauge);
    BA_HIP(hipGetLastError());
    return BA_OK;
}

static int camera_step_impl(ba_problem *p, const double *cam_in, double *cam_out, const double *g, const double *s_in,
                            double *s_out, const double *r_in, double *r_out, int64_t lo, int64_t hi, int rule,
                            double lr, double gscale, double bc1, double bc2, int gauge, double *R_out,
                            double *gauge_out, void *stream) {
    BA_REQUIRE(p && cam_in && cam_out, "ba_camera_step: null pointer");
    BA_REQUIRE(rule >= 0 && rule <= 2 && lo >= 0 && hi <= 6 * p->K + 1, "ba_camera_step: bad rule or range");
    BA_REQUIRE(rule == 0 || lo >= hi || (g && (rule != 2 || (s_in && s_out && r_in && r_out))),
               "ba_camera_step: null pointer");
    BA_REQUIRE(!gauge || (p->K >= 2 && gauge_out), "ba_camera_step: the gauge fix needs two cameras and gauge_out");
    BA_REQUIRE(same_or_disjoint(cam_in, cam_out, 6 * p->K + 1),
               "ba_camera_step: cam_out must be cam_in or not overlap it");
    CameraStep a;
    a.K = p->K;
    a.cam_in = cam_in;
    a.cam = cam_out;
    a.g = g;
    a.s_in = s_in;
    a.r_in = r_in;
    a.s = s_out;
    a.r = r_out;
    a.lo = lo;
    a.hi = hi;
    a.rule = make_rule(lo < hi ? rule : 0, lr, gscale, bc1, bc2);
    a.do_gauge = gauge;
    a.W = p->W;
    a.H = p->H;
    a.T = p->cam_table;
    a.scale_later = (gauge && p->K > 2048) ? 1 : 0;     // beyond one batch of the kernel's last workgroup
    a.scalars = p->scalars;
    a.R_out = R_out;
    a.gauge = gauge_out;
    CameraStepShared sh;
    sh.counter = p->step_counter;
    sh.block_best = p->step_best;
    sh.stage = p->step_stage;
    if (gauge) {
        hipLaunchKernelGGL(k_camera_step_quad, dim3(cdiv(p->K - 1, 63)), dim3(256), 0, (hipStream_t)stream, a, sh);
        if (a.scale_later)
            hipLaunchKernelGGL(k_scale_centres, dim3(cdiv(3 * p->K - 3, 256)), dim3(256), 0, (hipStream_t)stream, p->K,
                               a.cam, a.T, a.gauge);
    } else {
        hipLaunchKernelGGL(k_camera_step_mb, dim3(cdiv(p->K, 256)), dim3(256), 0, (hipStream_t)stream, a, sh);
    }
    BA_HIP(hipGetLastError());
    return BA_OK;
}

}  // extern "C"  (closed around the shared implementation above)

extern "C" {

int ba_camera_step(ba_problem *p, const double *cam_in, double *cam_out, const double *g, const double *s_in,
                   double *s_out, const double *r_in, double *r_out, int64_t lo, int64_t hi, int rule, double lr,
                   double gscale, double bc1, double bc2, int gauge, double *R_out, double *gauge_out, void *stream) {
    return camera_step_impl(p, cam_in, cam_out, g, s_in, s_out, r_in, r_out, lo, hi, rule, lr, gscale, bc1, bc2, gauge,
                            R_out, gauge_out, stream);
}

int ba_camera_step_finalize(ba_problem *p, const double *cam_in, const double *cam_in_gauge, double *cam_out,
                            double *g_out, int post, const double *s_in, double *s_out, const double *r_in,
                            double *r_out, int64_t lo, int64_t hi, int rule, double lr, double gscale, double bc1,
                            double bc2, double *R_out, double *gauge_out, double *next_gauge, void *stream) {
    BA_REQUIRE(p && g_out && cam_in && cam_out && gauge_out, "ba_camera_step_finalize: null pointer");
    BA_REQUIRE(p->pending_partials > 0,
               "ba_camera_step_finalize: no observation pass with BA_EVAL_DEFER_FINALIZE is waiting");
    BA_REQUIRE(rule >= 0 && rule <= 2 && lo >= 0 && hi <= 6 * p->K + 1, "ba_camera_step_finalize: bad rule or range");
    BA_REQUIRE(rule != 2 || (s_in && s_out && r_in && r_out), "ba_camera_step_finalize: null pointer");
    BA_REQUIRE(p->K >= 2, "ba_camera_step_finalize: the gauge fix needs two cameras");
    const int64_t n = 6 * p->K + 1;
    // no workgroup waits for another: camera 0's old values, which all of them read, must survive the launch
    BA_REQUIRE(cam_out + n <= cam_in || cam_in + n <= cam_out, "ba_camera_step_finalize: cam_out must not overlap cam_in");
    BA_REQUIRE(rule != 2 || ((s_out + n <= s_in || s_in + n <= s_out) && (r_out + n <= r_in || r_in + n <= r_out)),
               "ba_camera_step_finalize: the moments need output buffers of their own");
    CameraStep a;
    memset(&a, 0, sizeof(a));
    a.K = p->K;
    a.cam_in = cam_in;
    a.cam = cam_out;
    a.s_in = s_in;
    a.r_in = r_in;
    a.s = s_out;
    a.r = r_out;
    a.lo = lo;
    a.hi = hi;
    a.rule = make_rule(lo < hi ? rule : 0, lr, gscale, bc1, bc2);
    if (rule == 2 && lo >= hi) a.rule.mode = 0;
    a.do_gauge = 1;
    a.W = p->W;
    a.H = p->H;
    a.scalars = p->scalars;
    a.R_out = R_out;
    a.gauge = gauge_out;
    FinalizeArgs fz;
    fz.partials = p->esum;      // the tiles' sums (over the GPUs too, if a collective ran in between); zeroed once read
    fz.n_own = 0;
    fz.n_partials = ESUM_SLOTS;
    fz.cam_sums = p->cam_sums;
    fz.g_out = g_out;
    if (post) {
        fz.mailbox = p->mailbox_dev;
        fz.seq = (double)++p->post_seq;
    }
    FreeStep fs;
    fs.T_in = p->cam_table;
    fs.T_out = p->cam_table_alt;
    fs.cam0_in = p->cam0_sums + 6 * p->cam0_parity;
    fs.cam0_zero = p->cam0_sums + 6 * (p->cam0_parity ^ 1);
    fs.in_scale = cam_in_gauge;
    fs.next_gauge = next_gauge;
    p->pending_partials = 0;
    hipLaunchKernelGGL(k_camera_step_free, dim3(cdiv(p->K - 1, 63)), dim3(256), 0, (hipStream_t)stream, a, fz, fs);
    BA_HIP(hipGetLastError());
    std::swap(p->cam_table, p->cam_table_alt);      // the table of the new parameters is the current one
    p->table_centres_stale = 1;                     // its rows hold rotations only
    return BA_OK;
}

/* cam_out = cam_in with the scale of a gauge record applied to the centres (entries 3 .. 3K; camera 0 sits at the
 * origin): what a caller outside the fused loop needs when a ticket-free camera step left that scale pending. */
int ba_scale_centres(int64_t K, const double *cam_in, const double *gauge, double *cam_out, void *stream) {
    BA_REQUIRE(K >= 1 && cam_in && gauge && cam_out && cam_in != cam_out, "ba_scale_centres: null or aliased pointer");
    hipLaunchKernelGGL(k_scale_centres_copy, dim3(cdiv(6 * K + 1, 256)), dim3(256), 0, (hipStream_t)stream, K, cam_in,
                       gauge, cam_out);
    BA_HIP(hipGetLastError());
    return BA_OK;
}

int ba_sum_buffer_size(ba_problem *p, int64_t *n_partials) {
    BA_REQUIRE(p && n_partials, "ba_sum_buffer_size: null pointer");
    *n_partials = SUM_EXTRA / 2;
    return BA_OK;
}

int ba_set_sum_buffer(ba_problem *p, double *buf, int64_t n_partials) {
    BA_REQUIRE(p && buf, "ba_set_sum_buffer: null pointer");
    BA_REQUIRE(n_partials >= SUM_EXTRA / 2, "ba_set_sum_buffer: room for fewer pairs than ba_sum_buffer_size reports");
    BA_REQUIRE(p->pending_partials == 0, "ba_set_sum_buffer: an observation pass is waiting for its finalisation");
    p->cam_sums = buf;
    p->esum = buf + p->K * 6;
    p->cam0_sums = p->esum + 2 * ESUM_SLOTS;
    p->ext_partials_cap = n_partials;
    return BA_OK;
}

int ba_fetch_double(const double *src, double *h_out, void *stream) {
    BA_REQUIRE(src && h_out, "ba_fetch_double: null pointer");
    hipStream_t st = (hipStream_t)stream;
    BA_HIP(hipMemcpyAsync(h_out, src, sizeof(double), hipMemcpyDeviceToHost, st));
    BA_HIP(hipStreamSynchronize(st));
    return BA_OK;
}

int ba_gauge_fix(int64_t N, int64_t K, double *X, double *C, double *Theta, double *R_out, double *gauge_out,
                 void *stream) {
    BA_REQUIRE(K >= 2, "ba_gauge_fix: needs at least two cameras");
    BA_REQUIRE(C && Theta && gauge_out && (N == 0 || X), "ba_gauge_fix: null pointer");
    BA_REQUIRE(Theta == C + 3 * K, "ba_gauge_fix: Theta must follow C in one buffer ([C | Theta])");
    hipStream_t st = (hipStream_t)stream;
    CameraStep a;
    memset(&a, 0, sizeof(a));
    a.K = K;
    a.cam_in = C;
    a.cam = C;
    a.do_gauge = 1;
    a.R_out = R_out;
    a.gauge = gauge_out;
    hipLaunchKernelGGL(k_camera_step, dim3(1), dim3(1024), 0, st, a);
    if (N > 0) hipLaunchKernelGGL(k_gauge_points, dim3(cdiv(N, 256)), dim3(256), 0, st, N, X, X, gauge_out);
    BA_HIP(hipGetLastError());
    return BA_OK;
}

}  // extern "C"

// Camera table of ba_project: one buffer per (GPU, stream), grown on demand and kept until the process ends, so that a
// call neither allocates (after the first of its size) nor waits for the stream; calls on one stream are ordered by
// the stream, calls on different streams have tables of their own.
static int project_table(hipStream_t st, int64_t K, double **out) {
    struct Table {
        double *p = nullptr;
        int64_t rows = 0;
    };
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, Table> tables;
    int dev = -1;
    BA_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    Table &t = tables[std::make_pair(dev, st)];
    if (t.rows < K) {
        if (t.p) {
            BA_HIP(hipStreamSynchronize(st));       // the last call on this stream may still read the old table
            BA_HIP(hipFree(t.p));
            t.p = nullptr, t.rows = 0;
        }
        BA_HIP(hipMalloc(&t.p, sizeof(double) * K * CAM_ROW));
        t.rows = K;
    }
    *out = t.p;
    return BA_OK;
}

extern "C" {

int ba_project(int64_t N, int64_t K, int64_t n, const int32_t *i, const int32_t *j, const double *X, const double *C,
               const double *Theta, double phi, double W, double H, int delete_back_points, int delete_boundary_points,
               double *out, void *stream) {
    BA_REQUIRE(X && C && Theta && out, "ba_project: null pointer");
    BA_REQUIRE((i == nullptr) == (j == nullptr), "ba_project: i and j must both be given or both be null");
    BA_REQUIRE(i || n == N * K, "ba_project: dense form needs n == N*K");
    if (n == 0) return BA_OK;
    hipStream_t st = (hipStream_t)stream;
    double *T = nullptr;
    if (int rc = project_table(st, K, &T)) return rc;
    hipLaunchKernelGGL(k_camera_table, dim3(cdiv(K, 256)), dim3(256), 0, st, K, C, Theta, (const double *)nullptr, W, H,
                       T, (double *)nullptr);
    const double w_over_tan = W / tan(phi / 2.0);
    hipLaunchKernelGGL(k_project, dim3(cdiv(n, 256)), dim3(256), 0, st, K, n, i, j, X, T, w_over_tan, W, H,
                       delete_back_points, delete_boundary_points, (double2 *)out);
    BA_HIP(hipGetLastError());
    return BA_OK;
}

}  // extern "C"

// ---- ba_project for callers that cannot allocate (HIP graph capture, serving loops): the camera table, which
// ba_project keeps in buffers it allocates on demand, lives in a fixed array of the code object.  The FIRST call on
// a GPU looks the array's address up (hipGetSymbolAddress: a runtime query that may load the code object, not a
// stream operation) and so has to happen outside a graph capture; every later call only launches two kernels.
// Every GPU holds its own copy at its own address; host code reaches it through the pointer
// hipGetSymbolAddress returns for the current device, never through the array's name.
constexpr int64_t NOALLOC_MAX_CAMERAS = 16384;                       // 2 MiB of table per GPU
constexpr int NOALLOC_MAX_DEVICES = 64;
static __device__ __align__(256) double scratch[NOALLOC_MAX_CAMERAS * CAM_ROW];
static double *g_scratch_ptr[NOALLOC_MAX_DEVICES] = {};
static std::mutex g_scratch_mutex;

// Device address of `scratch` on the current GPU (looked up once per GPU).
static int noalloc_scratch(double **out) {
    int dev = -1;
    BA_HIP(hipGetDevice(&dev));
    BA_REQUIRE(dev >= 0 && dev < NOALLOC_MAX_DEVICES, "ba_project_noalloc: device index out of range");
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        double *ptr = nullptr;
        BA_HIP(hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)));
        BA_REQUIRE(ptr != nullptr, "ba_project_noalloc: no address for the scratch array on this device");
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return BA_OK;
}

extern "C" {

int ba_project_noalloc(int64_t N, int64_t K, int64_t n, const int32_t *i, const int32_t *j, const double *X,
                       const double *C, const double *Theta, double phi, double W, double H, int delete_back_points,
                       int delete_boundary_points, double *out, void *stream) {
    BA_REQUIRE(X && C && Theta && out, "ba_project_noalloc: null pointer");
    BA_REQUIRE((i == nullptr) == (j == nullptr), "ba_project_noalloc: i and j must both be given or both be null");
    BA_REQUIRE(i || n == N * K, "ba_project_noalloc: dense form needs n == N*K");
    BA_REQUIRE(K >= 1 && K <= NOALLOC_MAX_CAMERAS,
               "ba_project_noalloc: more than 16384 cameras do not fit the fixed camera table; use ba_project");
    if (n == 0) return BA_OK;
    hipStream_t st = (hipStream_t)stream;
    double *T = nullptr;
    const int rc = noalloc_scratch(&T);
    if (rc != BA_OK) return rc;
    // every row a call reads is written by the same call first: nothing of an earlier call's table is used
    hipLaunchKernelGGL(k_camera_table, dim3(cdiv(K, 256)), dim3(256), 0, st, K, C, Theta, (const double *)nullptr, W, H,
                       T, (double *)nullptr);
    const double w_over_tan = W / tan(phi / 2.0);
    hipLaunchKernelGGL(k_project, dim3(cdiv(n, 256)), dim3(256), 0, st, K, n, i, j, X, (const double *)T, w_over_tan, W,
                       H, delete_back_points, delete_boundary_points, (double2 *)out);
    BA_HIP(hipGetLastError());
    return BA_OK;
}

int ba_reprojection_error(int64_t n, const int32_t *pix_xy, const double *model, double *sums, void *stream) {
    BA_REQUIRE(pix_xy && model && sums, "ba_reprojection_error: null pointer");
    hipStream_t st = (hipStream_t)stream;
    BA_HIP(hipMemsetAsync(sums, 0, 2 * sizeof(double), st));
    if (n == 0) return BA_OK;
    int grid = cdiv(n, 256 * 8);
    if (grid > 1024) grid = 1024;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_reproj_error, dim3(grid), dim3(256), 0, st, n, (const int2 *)pix_xy, (const double2 *)model,
                       sums);
    BA_HIP(hipGetLastError());
    return BA_OK;
}

}  // extern "C"
