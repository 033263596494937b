// Observation pass on LDS tiles in "dense block" layout (internal).
//
// For scenes whose tiles are nearly dense - every camera of a tile sees nearly every point of the
// tile, as in keypoint tracks of a video - the tile is stored as a grid of blocks of 4 cameras x 8
// points, one block per lane group {l, l + 16, l + 32, l + 48} of a wave:
//   lane l of a wave:  camera slot 4 b + (l >> 4)  (b: camera batch of the wave),
//                      point group (l & 15) of the current chunk of 128 points,
//                      plane k = 0..7: point 8 * group + k.
// A record holds only the pixel (its camera and point follow from its position); pairs that are not
// observations are zero-weight records.  This layout puts both reductions of the pass into
// registers:
//   * per camera: a thread keeps ONE camera for the whole tile, so the 6 / 33 camera sums stay in
//     its registers until the tile ends; the 16 lanes of a row then add up with DPP row shifts
//     and one lane per row adds to the global sums - no LDS accumulators at all;
//   * per point: the four lanes of a block hold the same 8 points, so the 3 / 12 point sums of
//     the 8 planes are summed across the block by a transposing reduction with
//     v_permlane16_swap / v_permlane32_swap (a swap of two registers between lanes 16 / 32
//     apart followed by one add halves the data: 3 instructions per pair of values, no selects).
//     Afterwards every lane holds the complete block sums of 2 of the 8 points and adds them to
//     the point's LDS record: 24 / 6 LDS atomics per 8 observations instead of 96 / 24.
// Records are stored per (tile, batch, chunk) unit as 8 planes of 64 entries: a wave reads 256
// (compact) or 1024 consecutive bytes per plane.
//
// Point records in LDS ([x (3) | sums (NPV) | pad to an odd length]) sit at index
// (p & 7) * S + (p >> 3): lanes of a row read consecutive records (odd length: different bank
// pairs), the four rows read the same address (broadcast); with S = 4 mod 8 the atomics of rows
// 0 / 1 (planes k and k + 4) fall into different halves of the banks.
#pragma once

#include "ba_device.h"
#include "ba_runpass.h"

namespace ba {

constexpr int BLK_PTS = 128;     // points per chunk (16 groups of 8)
constexpr int BLK_CAMS = 4;      // cameras per batch
constexpr int BLK_UNIT = 512;    // record slots per (batch, chunk) unit: 8 planes x 64 lanes

struct BlockArgs;
// compact record: pixel x (low half) and y (high half) as int16, centre = integer + 0.5; x = -32768: zero weight
struct BlkRecI16 {
    typedef uint32_t raw_t;
    static __device__ __forceinline__ raw_t load(const BlockArgs &a, int64_t idx);
    static __device__ __forceinline__ bool valid(raw_t r) { return (r & 0xffffu) != 0x8000u; }
    static __device__ __forceinline__ double2 uv(raw_t r) {
        return make_double2((double)(int)(short)(r & 0xffffu) + 0.5, (double)((int)r >> 16) + 0.5);
    }
};
// general record: pixel centre as two doubles; x = NaN: zero weight
struct BlkRecF64 {
    typedef double2 raw_t;
    static __device__ __forceinline__ raw_t load(const BlockArgs &a, int64_t idx);
    static __device__ __forceinline__ bool valid(const raw_t &r) { return r.x == r.x; }
    static __device__ __forceinline__ double2 uv(const raw_t &r) { return r; }
};

struct BlockArgs {
    const uint32_t *brec = nullptr;     // BlkRecI16 records
    const double2 *buv = nullptr;       // BlkRecF64 records
    const int64_t *blk_ptr = nullptr;   // n_tiles + 1: first unit of every tile
    const TileMeta *meta = nullptr;
    const int32_t *cam_list = nullptr;
    int K = 0;
    int cam_cap = 0;
    int pt_stride = 0;                  // S: LDS capacity in points is 8 * S
    const double *X = nullptr;
    const double *T = nullptr;          // camera table: rotations
    const double *C = nullptr;          // camera centres (K x 3)
    const double *scalars = nullptr;
    double *out[3] = {nullptr, nullptr, nullptr};   // per-point outputs (see the Op), any may be null
    double *cam_sums = nullptr;         // Op::NCV x K, component-major, accumulated with atomics
    double *partials = nullptr;         // Op::NPART per tile
};

__device__ __forceinline__ BlkRecI16::raw_t BlkRecI16::load(const BlockArgs &a, int64_t idx) { return a.brec[idx]; }
__device__ __forceinline__ BlkRecF64::raw_t BlkRecF64::load(const BlockArgs &a, int64_t idx) { return a.buv[idx]; }

__host__ __device__ inline size_t block_pass_lds_doubles(int npv, int cam_cap, int pt_stride) {
    return (size_t)cam_cap * ROW_LDS + (size_t)pt_record_doubles(npv) * 8 * pt_stride;
}

typedef unsigned blk_v2u __attribute__((ext_vector_type(2)));

// a + (a of the lane 16 above) on even rows, b + (b of the lane 16 below) on odd rows
__device__ __forceinline__ double swap16_add(double a, double b) {
    const blk_v2u lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const blk_v2u hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}
// a + (a of the lane 32 above) on lanes 0..31, b + (b of the lane 32 below) on lanes 32..63
__device__ __forceinline__ double swap32_add(double a, double b) {
    const blk_v2u lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const blk_v2u hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}

// Op: the same interface as for k_run_pass.  Launch with 64 * min(4, camera batches of the largest tile) threads.
template <class Op, class Rec>
__global__ void __launch_bounds__(256, Op::WAVES_PER_EU) k_block_pass(BlockArgs a) {
    constexpr int NCV = Op::NCV, NPV = Op::NPV, NPART = Op::NPART;
    constexpr int PREC = pt_record_doubles(NPV);
    extern __shared__ double lds[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nthr = blockDim.x, nw = nthr >> 6;
    const int t = blockIdx.x;
    const TileMeta tm = a.meta[t];
    const int p0 = tm.p0, np = tm.np, nc = tm.nc;
    const int64_t cb = tm.cb;
    __shared__ double red[NPART][4];
    if (tm.e0 >= tm.e1) {
        if (tid < NPART) a.partials[NPART * (int64_t)t + tid] = 0.0;
        return;
    }
    const int nchunk = (np + BLK_PTS - 1) / BLK_PTS, nb = (nc + BLK_CAMS - 1) / BLK_CAMS;
    const int64_t ub = a.blk_ptr[t];
    const int S = a.pt_stride;
    double *rows = lds;                                   // cam_cap x ROW_LDS
    double *pts = rows + (size_t)a.cam_cap * ROW_LDS;     // 8 S records of PREC doubles

    for (int row = tid; row < nc; row += nthr) {
        const int j = tm.j0 >= 0 ? tm.j0 + row : a.cam_list[cb + row];
        stage_camera_row_rt(a.T + (int64_t)j * CAM_ROW, a.C + 3 * (int64_t)j, 1.0, rows + row * ROW_LDS);
    }
    // every point slot of the tile's chunks: x (zero past the tile's last point: such slots only
    // meet zero-weight records, which need finite coordinates) and zeroed sums
    for (int pt = tid; pt < nchunk * BLK_PTS; pt += nthr) {
        double *rec_pt = pts + pt_swz(pt, S) * PREC;
        const double *x = a.X + 3 * (int64_t)(p0 + pt);
#pragma unroll
        for (int comp = 0; comp < 3; ++comp) rec_pt[comp] = pt < np ? x[comp] : 0.0;
#pragma unroll
        for (int v = 0; v < NPV; ++v) rec_pt[3 + v] = 0.0;
    }
    const Op op(a.scalars);
    __syncthreads();

    double part[NPART];
#pragma unroll
    for (int k = 0; k < NPART; ++k) part[k] = 0.0;

    const int g = lane & 15, crow = lane >> 4;
    // after the two swap stages the lane holds the sums of planes (0 | 1) + 2 (lane >> 5) + 4 ((lane >> 4) & 1)
    const int plane_mine = 2 * (lane >> 5) + 4 * (crow & 1);
    for (int b = wid; b < nb; b += nw) {
        const int slot_raw = BLK_CAMS * b + crow;
        const bool cam_ok = slot_raw < nc;
        const int slot = cam_ok ? slot_raw : nc - 1;
        double2 rw[6];
        {
            const double2 *src = (const double2 *)(rows + slot * ROW_LDS);
#pragma unroll
            for (int k = 0; k < 6; ++k) rw[k] = src[k];
        }
        double cv[NCV];
#pragma unroll
        for (int k = 0; k < NCV; ++k) cv[k] = 0.0;
        typename Rec::raw_t rec[8];
        {
            const int64_t base = (ub + (int64_t)b * nchunk) * BLK_UNIT + lane;
#pragma unroll
            for (int k = 0; k < 8; ++k) rec[k] = Rec::load(a, base + 64 * k);
        }
        for (int c = 0; c < nchunk; ++c) {
            const int64_t base_next = (ub + (int64_t)b * nchunk + (c + 1 < nchunk ? c + 1 : c)) * BLK_UNIT + lane;
            double *pbase = pts + (size_t)(c * 16 + g) * PREC;     // plane k: + k * S * PREC
            const int SP = S * PREC;
            double xn[3] = {pbase[0], pbase[1], pbase[2]};
            double pvA[NPV], pvB[NPV], s1[2][NPV];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                // plane order 0 4 2 6 | 1 5 3 7: partners of the first swap stage are neighbours
                const int k = ((i & 1) << 2) | (i & 2) | (i >> 2);
                const int kn = (((i + 1) & 1) << 2) | ((i + 1) & 2) | ((i + 1) >> 2);
                const double x[3] = {xn[0], xn[1], xn[2]};
                const bool valid = Rec::valid(rec[k]);
                const double2 uv = Rec::uv(rec[k]);
                if (i + 1 < 8) {
                    const double *pn = pbase + kn * SP;
                    xn[0] = pn[0];
                    xn[1] = pn[1];
                    xn[2] = pn[2];
                }
                rec[k] = Rec::load(a, base_next + 64 * k);      // this plane of the next chunk (no branch)
                if (i & 1) op.obs(valid, x, rw, uv, cv, pvB, part);
                else op.obs(valid, x, rw, uv, cv, pvA, part);
                if (i & 1) {
#pragma unroll
                    for (int v = 0; v < NPV; ++v) s1[(i >> 1) & 1][v] = swap16_add(pvA[v], pvB[v]);
                }
                if ((i & 3) == 3) {
                    double *dst = pbase + (plane_mine + (i >> 2)) * SP + 3;
#pragma unroll
                    for (int v = 0; v < NPV; ++v)
                        __hip_atomic_fetch_add(dst + v, swap32_add(s1[0][v], s1[1][v]), __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the 16 lanes of a row hold the same camera: add up along the row, lane 15 of the row owns the total
        const int j = tm.j0 >= 0 ? tm.j0 + slot : a.cam_list[cb + slot];
#pragma unroll
        for (int k = 0; k < NCV; ++k) {
            if (!Op::cv_used(k)) continue;
            double v = op.cv_flush(k, cv[k]);
            v += row_shr<1>(v);
            v += row_shr<2>(v);
            v += row_shr<4>(v);
            v += row_shr<8>(v);
            if (g == 15 && cam_ok)
                __hip_atomic_fetch_add(a.cam_sums + (int64_t)k * a.K + j, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

#pragma unroll
    for (int k = 0; k < NPART; ++k) {
        const double v = wave_sum(part[k]);
        if (lane == 0) red[k][wid] = v;
    }
    __syncthreads();
    if (tid < NPART) {
        double v = 0.0;
        for (int w = 0; w < nw; ++w) v += red[tid][w];
        a.partials[NPART * (int64_t)t + tid] = v;
    }

    // every point is written by exactly one tile: its own points and the unobserved ones up to the
    // next tile's first point (zeros)
    const int lo = tm.wlo - p0, hi = tm.whi - p0;
#pragma unroll
    for (int o = 0; o < Op::NOUT; ++o) {
        double *dst = a.out[o];
        if (!dst) continue;
        const int w = Op::out_width(o), off = Op::out_offset(o);
        for (int idx = w * lo + tid; idx < w * hi; idx += nthr) {
            const int pt = idx >= 0 ? idx / w : -1, k = idx - w * pt;
            dst[(int64_t)w * p0 + idx] = (idx >= 0 && pt < np) ? pts[pt_swz(pt, S) * PREC + 3 + off + k] : 0.0;
        }
    }
}

}  // namespace ba
