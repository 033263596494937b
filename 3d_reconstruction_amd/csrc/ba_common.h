// Shared declarations of the MI355X bundle-adjustment engine (internal; the public C ABI is
// include/ba_mi355x.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "ba_mi355x.h"

namespace ba {

// One row of the per-camera table (16 doubles = one 128-byte L2 line):
//   [0..8]  R = Ry(theta_y) Rx(theta_x) Rz(theta_z), row-major      (p2w:138-140)
//   [9..11] camera centre C
//   [12]    cos(theta_z)   [13] sin(theta_z)   (rotation axis of theta_x in the camera frame)
//   [14..15] unused
constexpr int CAM_ROW = 16;

// Device-resident scalars shared by the kernels of one evaluation.
//   [0] f = W / (2 tan(phi_x / 2))   (p2w:567)
//   [1] W/2   [2] H/2
//   [3] -f / sin(phi_x)              (p2w:701)
constexpr int SC_F = 0, SC_CX = 1, SC_CY = 2, SC_DPHI = 3, SC_COUNT = 8;

constexpr int EVAL_BLOCK = 1024;          // threads per workgroup of the observation pass
constexpr int EVAL_WAVES = EVAL_BLOCK / 64;
constexpr int ESUM_SLOTS = 64;            // pairs of energy accumulators of a deferred pass
constexpr int SUM_EXTRA = 2 * ESUM_SLOTS + 12;   // doubles behind the camera-frame sums: those pairs, camera 0's sums twice
constexpr int TIMING_RING = 256;          // event pairs kept for ba_eval_timing

void set_error(const std::string &msg);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

}  // namespace ba

#define BA_HIP(expr)                                                        \
    do {                                                                    \
        hipError_t _e = (expr);                                             \
        if (_e != hipSuccess) return ba::hip_fail(_e, #expr, __FILE__, __LINE__); \
    } while (0)

#define BA_REQUIRE(cond, msg)                                               \
    do {                                                                    \
        if (!(cond)) { ba::set_error(msg); return BA_ERR_ARG; }             \
    } while (0)

struct ba_problem {
    int device = 0;
    int64_t N = 0, K = 0, nV = 0;
    double W = 0, H = 0;
    int compute_units = 0;
    int lds_limit = 0;            // bytes of LDS one workgroup may use
    // observation list, point-major (device)
    int32_t *oi = nullptr, *oj = nullptr;
    double2 *ouv = nullptr;
    // per-wave chunk boundaries aligned to point starts (device, n_chunks + 1 entries)
    int32_t *chunk_begin = nullptr;
    int n_chunks = 0;
    int grid = 0;                 // workgroups of the observation pass
    int64_t n_points_seen = 0;    // points with at least one observation
    // tiled form of the observation pass (null / 0 when the scene does not tile)
    int tiled = 0;                // tiles exist
    int tiles_worth = 0;          // ... and the gradient pass is faster on them
    int n_tiles = 0, tile_obs_target = 0;
    int cam_cap = 0, pt_cap = 0;  // largest camera list / point span of a tile
    int cam_rep = 1;              // copies of the camera accumulators in LDS
    size_t tile_lds_bytes = 0;
    int32_t *tile_obs = nullptr;      // n_tiles + 1
    int64_t *tile_cam_ptr = nullptr;  // n_tiles + 1
    int32_t *cam_list = nullptr;
    uint32_t *opack = nullptr;        // nV: (point - tile's first point) << 16 | camera slot in the tile's list
    void *tile_meta = nullptr;        // n_tiles x TileMeta
    // camera-run layout of the tiles (see ba_runpass.h); null when the scene does not tile
    int compact = 0;                  // every pixel centre is int16 + 0.5: 8-byte records (rrec), else rpack + ruv
    uint32_t *rpack = nullptr;        // total_runs x 8 records
    double2 *ruv = nullptr;
    uint2 *rrec = nullptr;
    int64_t *run_ptr = nullptr;       // n_tiles + 1
    int64_t total_runs = 0;
    int run_pt_stride = 0;            // LDS point capacity / 8
    int run_slots_grad = 0, run_slots_jtj = 0;   // camera columns of the LDS accumulator space per pass (see RunArgs)
    // dense block layout of the tiles (see ba_blockpass.h); null when its padding would cost too much
    uint32_t *brec = nullptr;         // total_units x 512 compact records
    double2 *buv = nullptr;           // ... or pixel centres as doubles
    int64_t *blk_ptr = nullptr;       // n_tiles + 1
    int64_t total_units = 0;
    double blk_fill = 0.0;            // observations / record slots
    int blk_pt_stride = 0;            // LDS point capacity / 8
    int blk_waves = 0;                // waves per workgroup of the block pass
    // scratch owned by the handle (device)
    double *cam_table = nullptr;  // K x CAM_ROW
    int table_centres_stale = 0;  // the merged camera step left the centres of the rows unwritten: only the run
                                  // pass (which takes the centres from C itself) may reuse the table as it is
    double *sum_block = nullptr;  // owned: [cam_sums | partials]
    double *cam_sums = nullptr;   // 6 x K camera-frame sums (SoA: component-major), zero between passes
    double *esum = nullptr;       // ESUM_SLOTS pairs of accumulators {sum r^2, sum m.r} of a pass whose finalisation is
                                  // deferred: tile t adds to pair t mod ESUM_SLOTS atomically (one pair for all tiles
                                  // serialises at the memory side), the camera step that finishes the pass zeroes them
    double *cam0_sums = nullptr;  // camera 0's six camera-frame sums of a deferred pass, twice: the pass adds to the copy
                                  // of its parity, the camera step reads that copy and zeroes the other one (every
                                  // workgroup of the step reads camera 0's sums, so none of them can zero them)
    int cam0_parity = 0;          // parity of the latest deferred pass
    double *cam_table_alt = nullptr;   // second camera table: the ticket-free camera step reads one and writes the other
    double *partials = nullptr;   // 2 per wave / tile (passes finished by k_finalize)
    int64_t partials_cap = 0;     // partial-sum pairs this problem may write
    int64_t ext_partials_cap = 0; // > 0: cam_sums and esum live in a caller's buffer (ba_set_sum_buffer)
    double *scalars = nullptr;    // SC_COUNT
    int *flag = nullptr;          // 4 ints: validation flags, tile maxima
    int *step_counter = nullptr;  // arrival counter of the multi-workgroup camera step (kept at 0)
    double *step_best = nullptr;  // 2 per workgroup of that kernel
    double *step_stage = nullptr; // 16
    // energy mailbox: 2 doubles of host memory mapped into the device, [0] = energy sum, [1] = post number
    double *mailbox_host = nullptr, *mailbox_dev = nullptr;
    uint64_t post_seq = 0;        // posts requested so far
    int pending_partials = 0;     // partial sums of a pass whose finalisation was deferred to the camera step
    // second-order extension (allocated on first use)
    double *cam_sums_jtj = nullptr;  // 33 x K
    double *partials_jtj = nullptr;  // 3 per tile
    int64_t *row_ptr = nullptr;      // N + 1
    int max_degree = 0;              // most observations of one point
    int tri_selected_cameras = -1;   // cameras that see a point observed at least twice (ba_problem_triangulate), -1: not counted yet
    // Schur build on the matrix cores: ranges of points of a tile whose cameras lie within a window of slots
    // ({tile, first point in the tile, points, first slot | cameras << 16}), and per tile whether the pair kernel
    // takes it instead (a point of it sees cameras further apart than the window); built on first use
    int4 *schur_sub = nullptr;
    int *schur_rest = nullptr;
    int n_schur_sub = 0, n_schur_rest = -1;
    int schur_window = 13;        // camera slots the Gram kernel's accumulators hold for these ranges
    // timing ring
    std::vector<hipEvent_t> ev_begin, ev_end;
    int ev_count = 0;
    // a pass armed by ba_eval_update_arm: launched by ba_wait_energy_fire the moment the awaited energy allows it
    struct ArmedUpdate {
        bool armed = false;
        const double *X = nullptr, *X_gauge = nullptr, *C = nullptr, *C_gauge = nullptr, *Theta = nullptr, *phi = nullptr;
        int flags = 0;
        double *E_sum = nullptr, *gX = nullptr, *gC = nullptr, *gTheta = nullptr, *gphi = nullptr;
        int rule = 0;
        double lr = 0, gscale = 0, bc1 = 1, bc2 = 1;
        const double *s_in = nullptr, *r_in = nullptr;
        double *s_out = nullptr, *r_out = nullptr, *X_out = nullptr, *X_out_alt = nullptr;
    } armed_update;
    // second stream of the Schur build (tiles with many cameras beside the others), created on first use
    hipStream_t aux_stream = nullptr;
    hipEvent_t aux_fork = nullptr, aux_join = nullptr;
};

namespace ba {
// builds the camera table and scalars of the handle from C, Theta, phi (K1)
int camera_table(ba_problem *p, const double *C, const double *Theta, const double *phi, hipStream_t st);
// observation range of every point (row_ptr) and the longest track (max_degree) of the handle, built on first use
int ensure_row_ptr(ba_problem *p);
}  // namespace ba
