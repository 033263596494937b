// Observation pass on LDS tiles in "camera-run" layout (internal).
//
// A tile is a range of consecutive points with all their observations (about 2000 of them), its
// points and camera rows staged in LDS.  Inside the tile the observations are regrouped by camera
// and cut into RUNS of up to 8 observations of ONE camera; a thread owns a run.  That choice
// decides where the two reductions of the pass happen:
//   * per camera (6 sums for the gradient, 27 with the J^T J blocks): in the owning thread's
//     registers over the run, flushed once per run to the tile's LDS accumulators;
//   * per point (3 / 9 sums): LDS atomics straight from the lanes.  Lanes of a wave hold
//     different points of (mostly) the same camera, so the addresses differ and the atomics do not
//     serialise - the opposite of the point-major order, where the lanes of a wave share cameras
//     and the 6 / 27 camera sums collide.
// Records are stored per tile as 8 planes of n_runs entries (plane k holds one observation of
// every run), so lane r of a wave reads consecutive addresses in every plane: the record stream
// stays fully coalesced although each thread walks its own run.  A camera's last run is padded
// with records of zero weight (a copy of a real record's point and camera, pixel x = NaN), so the
// loop over a run has no branches and the eight bodies can overlap.
//
// Order of the runs inside a tile: cameras are taken in batches of 8; inside a batch the runs go
// round by round (round rho holds the rho-th run of each of the 8 cameras), and a run of camera c
// stores its observations rotated by c (plane k holds observation (k + c) mod 8).  For tracks of
// a video (every camera of the tile sees nearly every point of the tile, in order) the 16 lanes
// the LDS serves per cycle then hold 8 cameras x 2 rounds and, in every plane, 16 different
// points: the point atomics never collide, and two copies of the camera accumulators (selected by
// the round's parity, stored in the record) keep the flush collision-free as well.
//
// Point data in LDS: one record per point, [x (3) | per-point sums (NPV) | pad to an odd number of
// doubles], at record index pt_swz(p) = (p mod 8) * S + p / 8 with S = 2 * odd, which maps those 16
// points to 16 different bank pairs (an odd record length keeps that property; an odd S measured
// slower).  Every field of a point is then a constant offset from one address register.
//
// Latencies: the LDS returns results in order, so a read issued behind the 3 / 9 atomics of the
// previous observation waits for them; the point of observation k + 1 is therefore fetched between
// the arithmetic of observation k and its atomics, and the records of the thread's next run are requested
// as soon as those of the current run have been decoded.
#pragma once

#include "ba_device.h"

namespace ba {

#ifndef BA_RUN_LEN
#define BA_RUN_LEN 8
#endif
constexpr int RUN_LEN = BA_RUN_LEN;  // observations of a run (8 or 16)
#ifndef BA_RUN_PF
#define BA_RUN_PF 4
#endif
constexpr int RUN_PF = BA_RUN_PF;    // records a thread holds ahead of the one it works on (4 observation bodies of three
                                     // interleaved waves cover the latency of HBM; 8 cost eight more registers)
static_assert(RUN_LEN == 8 || RUN_LEN == 16, "runs of 8 or 16 observations");
constexpr int RUN_SLOT_BITS = 12;    // record: pt_swz(point in tile) << 16 | (round & 15) << 12 | camera slot
constexpr int RUN_MAX_CAMS = 1 << RUN_SLOT_BITS;

// Two record formats.  RecF64: the pack word and the pixel centre as two doubles (20 bytes).
// RecI16: for pixel centres of the form integer + 0.5 with the integer in (-32768, 32767] - what
// floor()-ed pixel tracks give - the pack word and both integers as int16 in one 8-byte word; the
// stream is 2.5 times shorter.  Zero-weight records: x = NaN, resp. x = -32768.
struct RunArgs;
struct RecF64 {
    struct raw_t {
        uint32_t pack;
        double2 uv;
    };
    // A tile's records as buffer resources (scalar registers): a load names the plane by a scalar offset and the
    // run by a 32-bit per-lane offset, so no plane costs a 64-bit address register pair per lane; reads beyond the
    // tile's records return zeros instead of faulting.
    struct stream_t {
        __amdgpu_buffer_rsrc_t pack, uv;
    };
    static __device__ __forceinline__ stream_t stream(const RunArgs &a, int64_t first, int count);
    // record `lane` of the plane that starts `plane0` records into the tile (plane0: the same for the whole wave)
    static __device__ __forceinline__ raw_t load(const stream_t &s, int plane0, uint32_t lane);
    static __device__ __forceinline__ uint32_t pack(const raw_t &r) { return r.pack; }
    static __device__ __forceinline__ bool valid(const raw_t &r) { return r.uv.x == r.uv.x; }
    static __device__ __forceinline__ double2 uv(const raw_t &r) { return r.uv; }
    // pixel centre minus PIX_SHIFT in both components: what the pass hands to the Op, which folds the shift into
    // its principal-point constants
    static constexpr double PIX_SHIFT = 0.0;
    static __device__ __forceinline__ double2 pix(const raw_t &r) { return r.uv; }
};
struct RecI16 {
    typedef uint2 raw_t;    // x: pack, y: pixel x (low half) and y (high half) as int16
    struct stream_t {
        __amdgpu_buffer_rsrc_t rec;
    };
    static __device__ __forceinline__ stream_t stream(const RunArgs &a, int64_t first, int count);
    static __device__ __forceinline__ raw_t load(const stream_t &s, int plane0, uint32_t lane);
    static __device__ __forceinline__ uint32_t pack(const raw_t &r) { return r.x; }
    static __device__ __forceinline__ bool valid(const raw_t &r) { return (r.y & 0xffffu) != 0x8000u; }
    static __device__ __forceinline__ double2 uv(const raw_t &r) {
        return make_double2((double)(int)(short)(r.y & 0xffffu) + 0.5, (double)((int)r.y >> 16) + 0.5);
    }
    static constexpr double PIX_SHIFT = 0.5;      // the integer pixel itself: no addition per record
    static __device__ __forceinline__ double2 pix(const raw_t &r) {
        return make_double2((double)(int)(short)(r.y & 0xffffu), (double)((int)r.y >> 16));
    }
};

struct RunArgs {
    const uint32_t *rpack = nullptr;    // RecF64: pt_swz(point index - tile's first point) << 16 | (round & 15) << 12 | camera slot
    const double2 *ruv = nullptr;       // RecF64: pixel centre
    const uint2 *rrec = nullptr;        // RecI16 records
    const int64_t *run_ptr = nullptr;   // n_tiles + 1: first run of every tile
    const TileMeta *meta = nullptr;     // n_tiles
    const int32_t *cam_list = nullptr;  // distinct cameras of every tile, ascending
    int K = 0;
    int cam_cap = 0;              // LDS capacity in cameras (max over tiles)
    int pt_stride = 0;            // S: LDS capacity in points is 8 * S
    int cacc_slots = 0;           // camera columns the LDS accumulator space holds (each Op::NCV_LDS doubles): a tile
                                  // with nc cameras uses cacc_slots / nc copies
    const double *X = nullptr;
    const double *X_gauge = nullptr;    // optional: the true points are ((X - C0) @ R0) * factor of this record
                              // (R0 row-major [0..8], C0 [9..11], factor [12]); applied while loading
    const double *T = nullptr;          // camera table: rotations
    const double *C = nullptr;          // camera centres (K x 3)
    const double *scalars = nullptr;
    // optional fused point update (upd.mode != 0): every point's gradient is complete when its
    // tile ends, so the tile applies the update rule right there and writes X_out / s_out / r_out
    UpdateRule upd = {0, 0.0, 0.0, 1.0, 1.0};
    const double *s_in = nullptr, *r_in = nullptr;
    double *X_out = nullptr, *s_out = nullptr, *r_out = nullptr;
    double *out[3] = {nullptr, nullptr, nullptr};           // per-point outputs (see the Op), any may be null
    double *cam_sums = nullptr;         // Op::NCV x K, component-major, accumulated with atomics
    double *cam0 = nullptr;             // optional: camera 0's sums go here (Op::NCV doubles) instead of cam_sums
    const double *C_scale = nullptr;    // optional gauge record: the centres in C still lack its scale factor
    double *partials = nullptr;         // Op::NPART per tile
    double *esum = nullptr;             // optional: ESUM_SLOTS x Op::NPART accumulators the tiles add their partial sums to
                                        // instead (atomically; whoever finishes the pass reads 64 pairs, not one per tile)
};

constexpr int BUFFER_RSRC_WORD3 = 0x00020000;      // raw buffer of 32-bit data on gfx950
__device__ __forceinline__ RecF64::stream_t RecF64::stream(const RunArgs &a, int64_t first, int count) {
    stream_t s;
    s.pack = __builtin_amdgcn_make_buffer_rsrc((void *)(a.rpack + first), 0, count * 4, BUFFER_RSRC_WORD3);
    s.uv = __builtin_amdgcn_make_buffer_rsrc((void *)(a.ruv + first), 0, count * 16, BUFFER_RSRC_WORD3);
    return s;
}
__device__ __forceinline__ RecF64::raw_t RecF64::load(const stream_t &s, int plane0, uint32_t lane) {
    raw_t r;
    r.pack = __builtin_amdgcn_raw_buffer_load_b32(s.pack, lane * 4u, plane0 * 4, 0);
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(s.uv, lane * 16u, plane0 * 16, 0);
    r.uv = make_double2(__hiloint2double((int)v[1], (int)v[0]), __hiloint2double((int)v[3], (int)v[2]));
    return r;
}
__device__ __forceinline__ RecI16::stream_t RecI16::stream(const RunArgs &a, int64_t first, int count) {
    stream_t s;
    s.rec = __builtin_amdgcn_make_buffer_rsrc((void *)(a.rrec + first), 0, count * 8, BUFFER_RSRC_WORD3);
    return s;
}
__device__ __forceinline__ RecI16::raw_t RecI16::load(const stream_t &s, int plane0, uint32_t lane) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b64(s.rec, lane * 8u, plane0 * 8, 0);
    return make_uint2(v[0], v[1]);
}

__device__ __forceinline__ int pt_swz(int p, int S) { return (p & 7) * S + (p >> 3); }
// the same for the run pass: RUN_LEN rows of S records (the 16 lanes the LDS serves together hold RUN_LEN-periodic
// point numbers of two consecutive rounds: row = p mod RUN_LEN differs inside a round, p / RUN_LEN between the
// rounds, and with S = 2 * odd and an odd record length the two sets fall on bank pairs of different parity)
__host__ __device__ __forceinline__ int run_pt_swz(int p, int S) { return (p & (RUN_LEN - 1)) * S + p / RUN_LEN; }

// doubles of a point's LDS record: x (3), the per-point sums, padded to an odd count
__host__ __device__ constexpr int pt_record_doubles(int npv) { return (3 + npv) | 1; }

// LDS doubles a workgroup of the pass needs
__host__ __device__ inline size_t run_pass_lds_doubles(int ncv, int npv, int cam_cap, int pt_stride, int cacc_slots) {
    return (size_t)cam_cap * ROW_LDS + (size_t)cacc_slots * ncv + (size_t)pt_record_doubles(npv) * RUN_LEN * pt_stride;
}
template <class Op>
__host__ __device__ inline size_t run_pass_lds_doubles(int cam_cap, int pt_stride, int cacc_slots) {
    return run_pass_lds_doubles(Op::NCV_LDS, Op::NPV, cam_cap, pt_stride, cacc_slots);
}

// Op supplies:  Op(scalars, pix_shift) (pix_shift: what the records' pixels lack to the pixel centre), cv_flush(k, v)
//   (the value camera sum k contributes when a run's sums are flushed: constant factors left out of obs()), WAVES_PER_EU / WAVES_PER_EU_COMPACT (register budget with long / 8-byte records),
//   cv_used(k) (false: camera sum k is never accumulated and is not flushed), NCV_LDS (number of used sums),
//   cv_lds(k) / cv_from_lds(c) (position of sum k among the used ones and back; the LDS accumulators hold only
//   those: with 160 KB of LDS per CU in granules of 1280 bytes, three workgroups fit if each stays within 53 760
//   bytes, and the J^T J pass at ~380 points and up to 20 cameras per tile is within a few hundred bytes of that), NCV, NPV, NPART, NOUT, out_width(o), out_offset(o) (slices of the per-point sums),
//   MOMENT_PREFETCH (rounds of the fused point update whose moments are loaded ahead of the main loop; 0 where the registers do not allow it),
//   obs(valid, x[3], rw[6], uv, cv[NCV] (+=), pv[NPV] (=), part[NPART] (+=)); everything it adds is zero when !valid
template <class Op, class Rec>
__global__ void __launch_bounds__(TILE_BLOCK, sizeof(typename Rec::raw_t) <= 8 ? Op::WAVES_PER_EU_COMPACT : Op::WAVES_PER_EU)
    k_run_pass(RunArgs a) {
    constexpr int NCV = Op::NCV, NCVL = Op::NCV_LDS, NPV = Op::NPV, NPART = Op::NPART;
    extern __shared__ double lds[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int t = blockIdx.x;
    const TileMeta tm = a.meta[t];
    const int p0 = tm.p0, np = tm.np, nc = tm.nc;
    const int64_t cb = tm.cb;
    const int64_t rb = a.run_ptr[t];
    const int nruns = (int)(a.run_ptr[t + 1] - rb);
    __shared__ double red[NPART][TILE_WAVES];
    if (nruns == 0) {
        if (tid < NPART && !a.esum) a.partials[NPART * (int64_t)t + tid] = 0.0;
        return;
    }
    // this thread's first run: issue the record loads before anything else
    const typename Rec::stream_t rs = Rec::stream(a, rb * RUN_LEN, nruns * RUN_LEN);
    typename Rec::raw_t rec[RUN_PF];
    {
        const uint32_t r = tid < nruns ? tid : nruns - 1;
#pragma unroll
        for (int k = 0; k < RUN_PF; ++k) rec[k] = Rec::load(rs, k * nruns, r);
    }

    // the fused point update's moments of this thread's first MPF components: requested now, used after the main loop
    constexpr int MPF = sizeof(typename Rec::raw_t) <= 8 ? Op::MOMENT_PREFETCH : 0;     // the long records leave no registers for it
    const int n_comp = 3 * (tm.whi - tm.wlo);
    const int64_t g0 = 3 * (int64_t)tm.wlo;
    double spf[MPF > 0 ? MPF : 1], rpf[MPF > 0 ? MPF : 1];
#pragma unroll
    for (int m = 0; m < MPF; ++m) {
        const int li = tid + m * TILE_BLOCK;
        spf[m] = rpf[m] = 0.0;
        if (a.upd.mode == 2 && li < n_comp) {
            spf[m] = a.s_in[g0 + li];
            rpf[m] = a.r_in[g0 + li];
        }
    }

    constexpr int PREC = pt_record_doubles(NPV);                // doubles of a point's LDS record
    // copies of the camera accumulators: a tile spreads its flushes over as many copies as the accumulator space
    // holds for its camera count (copy = round mod rep, by multiply and shift)
    // The copies lie 8 (mod 32) doubles apart when there is room: the 32 lanes the LDS serves together hold 8 cameras
    // x 4 rounds = 4 consecutive copies, which then fall on 32 different bank pairs.
    const int csz = NCVL * nc, cspace = a.cacc_slots * NCVL;
    int cstride = csz + ((8 - csz) & 31), rep = 1 + (cspace - csz) / cstride;
    if (rep < 2 || rep < a.cacc_slots / nc) {       // the padding would cost a copy: pack them
        cstride = csz;
        rep = a.cacc_slots / nc;
    }
    rep = min(15, rep);
    const int rep_inv = (256 + rep - 1) / rep;
    const int S = a.pt_stride;
    double *rows = lds;                                         // cam_cap x ROW_LDS
    double *cacc = rows + (size_t)a.cam_cap * ROW_LDS;          // rep x NCVL x nc (component-major, used sums only)
    double *pts = cacc + (size_t)a.cacc_slots * NCVL;         // 8 S records of PREC doubles: x (3), sums (NPV)

    for (int row = tid; row < nc; row += TILE_BLOCK) {
        const int j = tm.j0 >= 0 ? tm.j0 + row : a.cam_list[cb + row];
        stage_camera_row_rt(a.T + (int64_t)j * CAM_ROW, a.C + 3 * (int64_t)j, a.C_scale ? gauge_factor(a.C_scale) : 1.0,
                            rows + row * ROW_LDS);
    }
    for (int idx = tid; idx < (rep - 1) * cstride + csz; idx += TILE_BLOCK) cacc[idx] = 0.0;
    {
        const double *G = a.X_gauge;
        double g[13] = {1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0};
        if (G) {
#pragma unroll
            for (int k = 0; k < 12; ++k) g[k] = G[k];
            g[12] = gauge_factor(G);
        }
        for (int pt = tid; pt < np; pt += TILE_BLOCK) {
            const double *x = a.X + 3 * (int64_t)(p0 + pt);
            double *rec_pt = pts + run_pt_swz(pt, S) * PREC;
            if (G) {
                const double d0 = x[0] - g[9], d1 = x[1] - g[10], d2 = x[2] - g[11];
#pragma unroll
                for (int comp = 0; comp < 3; ++comp)
                    rec_pt[comp] = (d0 * g[comp] + d1 * g[3 + comp] + d2 * g[6 + comp]) * g[12];
            } else {
#pragma unroll
                for (int comp = 0; comp < 3; ++comp) rec_pt[comp] = x[comp];
            }
#pragma unroll
            for (int v = 0; v < NPV; ++v) rec_pt[3 + v] = 0.0;
        }
    }
    const Op op(a.scalars, Rec::PIX_SHIFT);
    __syncthreads();

    double part[NPART];
#pragma unroll
    for (int k = 0; k < NPART; ++k) part[k] = 0.0;

    for (int r = tid; r < nruns; r += TILE_BLOCK) {
        const bool more = r + TILE_BLOCK < nruns;
        const uint32_t r_next = more ? r + TILE_BLOCK : r;
        const int slot = (int)(Rec::pack(rec[0]) & (RUN_MAX_CAMS - 1));   // one camera per run
        const int round = (int)(Rec::pack(rec[0]) >> RUN_SLOT_BITS) & 15;
        double2 rw[6];
        {
            const double2 *src = (const double2 *)(rows + slot * ROW_LDS);
#pragma unroll
            for (int k = 0; k < 6; ++k) rw[k] = src[k];
        }
        double cv[NCV];
#pragma unroll
        for (int k = 0; k < NCV; ++k) cv[k] = 0.0;
        double *pt_next = pts + (Rec::pack(rec[0]) >> 16) * PREC;
        double xn[3] = {pt_next[0], pt_next[1], pt_next[2]};
#pragma unroll
        for (int k = 0; k < RUN_LEN; ++k) {
            double *pt_cur = pt_next;
            const double x[3] = {xn[0], xn[1], xn[2]};
            const bool valid = Rec::valid(rec[k % RUN_PF]);
            const double2 uv = Rec::pix(rec[k % RUN_PF]);
            const uint32_t next_pack = k + 1 < RUN_LEN ? Rec::pack(rec[(k + 1) % RUN_PF]) : 0u;
            // the record RUN_PF positions ahead in this thread's stream: a later one of this run, or one of the
            // thread's next run (of this run again after the last one: no branch)
            rec[k % RUN_PF] = k + RUN_PF < RUN_LEN
                                  ? Rec::load(rs, (k + RUN_PF) * nruns, (uint32_t)r)
                                  : Rec::load(rs, (k + RUN_PF - RUN_LEN) * nruns, r_next);
            double pv[NPV];
            op.obs(valid, x, rw, uv, cv, pv, part);
            if (k + 1 < RUN_LEN) {
                // the next observation's point: requested behind this observation's arithmetic (its coordinates are
                // then not live at the register peak: the J^T J pass fits 168 registers without spills) but ahead
                // of its atomics, which the LDS would otherwise finish first
                __builtin_amdgcn_sched_barrier(0);
                pt_next = pts + (next_pack >> 16) * PREC;
                xn[0] = pt_next[0];
                xn[1] = pt_next[1];
                xn[2] = pt_next[2];
            }
#pragma unroll
            for (int v = 0; v < NPV; ++v)
                __hip_atomic_fetch_add(pt_cur + 3 + v, pv[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_sched_barrier(0);     // keep the eight bodies apart: their live ranges add up otherwise
        }
        double *acc = cacc + (size_t)(round - rep * ((round * rep_inv) >> 8)) * cstride + slot;
#pragma unroll
        for (int k = 0; k < NCV; ++k)
            if (Op::cv_used(k))
                __hip_atomic_fetch_add(acc + Op::cv_lds(k) * nc, op.cv_flush(k, cv[k]), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
    }

#pragma unroll
    for (int k = 0; k < NPART; ++k) {
        const double v = wave_sum(part[k]);
        if (lane == 0) red[k][wid] = v;
    }
    __syncthreads();
    if (tid < NPART) {
        const double v = (red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]);
        if (a.esum) __hip_atomic_fetch_add(a.esum + NPART * (t % ESUM_SLOTS) + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else a.partials[NPART * (int64_t)t + tid] = v;
    }

    // every point is written by exactly one tile: its own points and the unobserved ones up to the
    // next tile's first point (zeros)
    const int lo = tm.wlo - p0, hi = tm.whi - p0;
    if (a.upd.mode) {
        // component li of the tile's write range (32-bit index arithmetic: the loop runs 3 x points / 256 times).
        // The first MPF rounds are unrolled: their moments were requested before the main loop, and their LDS
        // reads are issued together (each would otherwise wait in turn behind the other workgroups' atomics).
        auto fetch = [&](int li, double &x, double &g) {
            const int lpt = li / 3, comp = li - 3 * lpt, pt = lpt + lo;
            const int64_t gi = g0 + li;
            g = 0.0;
            if (pt >= 0 && pt < np) {
                const double *rec_pt = pts + run_pt_swz(pt, S) * PREC;
                x = rec_pt[comp];
                g = rec_pt[3 + comp];     // the gradient is the first slice of the per-point sums
            } else if (a.X_gauge) {   // a point nobody observes: only the stored value changes frame
                const double *G = a.X_gauge, *xp = a.X + (gi - comp);
                x = ((xp[0] - G[9]) * G[comp] + (xp[1] - G[10]) * G[3 + comp] + (xp[2] - G[11]) * G[6 + comp]) * gauge_factor(G);
            } else {
                x = a.X[gi];
            }
        };
        auto store = [&](int li, double x, double g, double sv, double rv) {
            const int64_t gi = g0 + li;
            a.X_out[gi] = apply_rule(a.upd, x, g, &sv, &rv);
            if (a.upd.mode == 2) {
                a.s_out[gi] = sv;
                a.r_out[gi] = rv;
            }
        };
        if (MPF > 0) {
            double xs[MPF > 0 ? MPF : 1], gs[MPF > 0 ? MPF : 1];
#pragma unroll
            for (int m = 0; m < MPF; ++m) {
                const int li = tid + m * TILE_BLOCK;
                if (li < n_comp) fetch(li, xs[m], gs[m]);
            }
#pragma unroll
            for (int m = 0; m < MPF; ++m) {
                const int li = tid + m * TILE_BLOCK;
                if (li < n_comp) store(li, xs[m], gs[m], spf[m], rpf[m]);
            }
        }
        for (int li = tid + MPF * TILE_BLOCK; li < n_comp; li += TILE_BLOCK) {
            double x, g, sv = 0.0, rv = 0.0;
            fetch(li, x, g);
            if (a.upd.mode == 2) {
                sv = a.s_in[g0 + li];
                rv = a.r_in[g0 + li];
            }
            store(li, x, g, sv, rv);
        }
    }
#pragma unroll
    for (int o = 0; o < Op::NOUT; ++o) {
        double *dst = a.out[o];
        if (!dst) continue;
        const int w = Op::out_width(o), off = Op::out_offset(o);
        // four values per thread and trip, their LDS reads in flight together
        constexpr int UNR = 4;
        for (int base = w * lo + tid; base < w * hi; base += UNR * TILE_BLOCK) {
            double v[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int idx = base + u * TILE_BLOCK;
                const int pt = idx >= 0 ? idx / w : -1, k = idx - w * pt;
                v[u] = (idx < w * hi && idx >= 0 && pt < np) ? pts[run_pt_swz(pt, S) * PREC + 3 + off + k] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int idx = base + u * TILE_BLOCK;
                if (idx < w * hi) dst[(int64_t)w * p0 + idx] = v[u];
            }
        }
    }
    for (int idx = tid; idx < NCVL * nc; idx += TILE_BLOCK) {
        const int cl = idx / nc, slot = idx - cl * nc;
        const int comp = Op::cv_from_lds(cl);
        const int j = tm.j0 >= 0 ? tm.j0 + slot : a.cam_list[cb + slot];
        double v = cacc[idx];
        for (int r = 1; r < rep; ++r) v += cacc[idx + (size_t)r * cstride];
        double *dst = (j == 0 && a.cam0) ? a.cam0 + comp : a.cam_sums + (int64_t)comp * a.K + j;
        __hip_atomic_fetch_add(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace ba
