/*
 * ba_mi355x.h — C ABI of libba_mi355x.so, the MI355X (gfx950) bundle-adjustment engine.
 *
 * The reference (modules/pix2world.py, "p2w" below) is pure Python/NumPy and has no FFI
 * layer; its boundary for this path is the Python function fit_scene (p2w:480-500) and
 * the optimiser registry gd_func_dict (p2w:362).  The entry points declared here are what a
 * ctypes binding of that path calls: each one names the reference lines it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name starts with h_;
 *   - every function returns 0 on success or a negative ba_status; ba_last_error() returns
 *     the message of the calling thread's last failure; nothing throws;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *   - one ba_problem belongs to one GPU; a handle is not thread-safe, independent handles are;
 *   - all floating point is IEEE binary64, indices are int32, observation counts int64;
 *   - sums returned by ba_eval are UN-normalised (no 1/|V| factor) so that partial results of
 *     several GPUs can be added before scaling.
 */
#ifndef BA_MI355X_H
#define BA_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ba_problem ba_problem;

enum ba_status {
    BA_OK = 0,
    BA_ERR_ARG = -1,        /* bad argument (null pointer, negative size, unsorted input) */
    BA_ERR_HIP = -2,        /* a HIP runtime call failed */
    BA_ERR_STATE = -3,      /* call not valid for this handle */
    BA_ERR_NUMERIC = -4     /* factorisation met a non-positive pivot */
};

/* flags of ba_eval */
enum ba_eval_flags {
    BA_EVAL_GRAD_X = 1,     /* write gX                     (p2w:661-662) */
    BA_EVAL_GRAD_CAM = 2,   /* write gC, gTheta             (p2w:664-698) */
    BA_EVAL_GRAD_PHI = 4,   /* write gphi                   (p2w:700-702) */
    BA_EVAL_CAM_ATOMICS = 16, /* force the global-atomic camera reduction (default: LDS-private when it fits) */
    BA_EVAL_TIMED = 32,     /* record HIP events around the observation-pass kernel (see ba_eval_timing) */
    BA_EVAL_DEBUG_MASK = 64 | 128 | 256, /* diagnostics only: skip parts of the pass (results are wrong) */
    BA_EVAL_UNTILED = 512,  /* use the wave-chunk kernel even when the scene has LDS tiles */
    BA_EVAL_REUSE_TABLE = 1024, /* the handle's camera table already describes C, Theta, phi (left there by
                                  ba_camera_step); skip rebuilding it */
    BA_EVAL_POST = 2048,    /* also post E_sum to the handle's host mailbox (see ba_wait_energy) */
    BA_EVAL_POINT_MAJOR = 4096, /* use the point-major tile kernel instead of the camera-run one */
    BA_EVAL_DEFER_FINALIZE = 8192, /* leave E_sum and the camera-side gradient to ba_camera_step_finalize */
    BA_EVAL_DENSE_BLOCKS = 16384 /* ba_eval_jtj: use the dense block layout (if the scene has one) instead of camera runs */
};

const char *ba_last_error(void);
int ba_version(void);

/* Number of compute units / bytes of LDS per workgroup of `device` (launch sizing). */
int ba_device_info(int device, int *compute_units, int *lds_bytes);

/* ---- visibility set: X_visible = X_pix[:,:,0] >= 0, X_pix[X_visible] + 0.5 (p2w:506-508, 531)
 * X_pix is the dense (N,K,2) int32 track array in C order.  ba_visibility_count writes the
 * per-point counts' exclusive prefix sum row_ptr[N+1] (int64) and returns |V| in *h_nV;
 * ba_visibility_fill then writes the COO list in the order numpy.nonzero walks it
 * (i-major, j ascending): i[|V|], j[|V|], uv[|V|][2] = pixel + 0.5. */
int ba_visibility_count(const int32_t *X_pix, int64_t N, int64_t K, int64_t *row_ptr,
                        int64_t *h_nV, void *stream);
int ba_visibility_fill(const int32_t *X_pix, int64_t N, int64_t K, const int64_t *row_ptr,
                       int32_t *i, int32_t *j, double *uv, void *stream);

/* ---- the bipartite point / camera visibility graph of a COO list (p2w:519-525: the reference builds the dense
 * (N + K)^2 adjacency and asks scipy for its number of connected components; fit_scene needs it to be 1).
 * Label propagation on the device copy of the list: *h_camera_components = components that contain a camera (a
 * camera that sees nothing is one), *h_point_components = components of points alone (points nobody
 * sees); their sum is the reference's count.  camera_label (optional, K entries) receives for every camera the
 * smallest camera index of its component: what ranks that hold different point ranges exchange to merge their
 * verdicts. */
int ba_visibility_components(int64_t N, int64_t K, int64_t nV, const int32_t *i, const int32_t *j,
                             int32_t *camera_label, int64_t *h_camera_components, int64_t *h_point_components,
                             void *stream);

/* ---- the engine's internal point order: ascending first observing camera, ties in the caller's order (no
 * counterpart in the reference, whose arrays are dense).  Observations sorted by point.  *h_identity = 1: the list
 * already has the property, nothing is written.  Otherwise perm[N] (row of the new order -> caller's point index)
 * and the list rewritten in the new order (i_out holds new point numbers, sorted; every point's observations keep
 * their order). */
int ba_order_points_by_first_camera(int64_t N, int64_t K, int64_t nV, const int32_t *i, const int32_t *j,
                                    const double *uv, int32_t *perm, int32_t *i_out, int32_t *j_out, double *uv_out,
                                    int *h_identity, void *stream);

/* ---- order[nV]: the observations sorted by camera, points ascending inside a camera, ties in the caller's order (the
 * order ba_resect and ba_homography_diff_matrix take their lists in).  An own stable radix sort on the device; a list
 * that is already sorted by point needs one sort, any other two. */
int ba_argsort_by_camera(int64_t nV, int64_t N, int64_t K, const int32_t *i, const int32_t *j, int32_t *order,
                         void *stream);
/* ranges[n_keys + 1]: ranges[k] .. ranges[k + 1] = the run of key k in an ascending key list (keys 0 .. n_keys - 1;
 * row_ptr / col_ptr of the calls below from a sorted index list). */
int ba_ranges_of_sorted_keys(int64_t n, int64_t n_keys, const int32_t *keys, int64_t *ranges, void *stream);

/* ---- problem handle: the observation list of one GPU's shard (its points, all cameras).
 * Observations must be sorted by point (i non-decreasing); the arrays are copied.
 * Replaces the dense (N,K) temporaries fit_scene allocates at p2w:538-542, 555-573. */
int ba_problem_create(int64_t N, int64_t K, int64_t nV, const int32_t *i, const int32_t *j,
                      const double *uv, double W, double H, int device, ba_problem **out);
int ba_problem_destroy(ba_problem *p);

/* ---- fused per-observation pass: inner_transform (p2w:544-575), energy (p2w:612) and the
 * analytic gradient (p2w:654-702) on the visible pairs only.
 *   X[N][3], C[K][3], Theta[K][3] (columns theta_x, theta_y, theta_z), phi[1]
 *   E_sum[1]  : sum of squared residuals (E = E_sum / (2|V|))
 *   gX[N][3], gC[K][3], gTheta[K][3], gphi[1] : un-normalised gradients (multiply by 1/|V|)
 * Outputs not selected by `flags` may be NULL. */
int ba_eval(ba_problem *p, const double *X, const double *C, const double *Theta, const double *phi,
            int flags, double *E_sum, double *gX, double *gC, double *gTheta, double *gphi, void *stream);

/* ---- observation pass with the point update fused in (p2w:704-708 for the group X inside the pass
 * of p2w:544-702).  Available when ba_can_fuse_update(p) (the scene cuts into LDS tiles): every
 * point's gradient is complete when its tile ends, so the tile applies the update rule (1 SGD,
 * 2 Adam, as ba_adam_step) there and writes X_out (and s_out, r_out), saving the gX round trip.
 * X_gauge (optional, 16 doubles as written by ba_camera_step's gauge_out) says that the true points
 * are ((X - C0) @ R0) * factor: it is applied while X is loaded, so the rigid part of the gauge
 * fix (p2w:285-293, 327) needs no pass of its own either.  X_out then holds the updated points
 * BEFORE the next gauge fix; pair it with the gauge record of the camera step that follows.
 * C_gauge (optional; needs BA_EVAL_REUSE_TABLE): a gauge record whose scale factor is still pending on the camera
 * centres in C, as ba_camera_step_finalize leaves them; applied while the camera rows are staged.
 * gX may be NULL.  ba_gauge_points applies a gauge record: X_out = ((X_in - C0) @ R0) * factor.
 * ba_scale_centres applies a record's pending scale to a parameter vector: cam_out[6K+1] = cam_in with the centres
 * (entries 3 .. 3K) multiplied by the factor. */
int ba_can_fuse_update(const ba_problem *p);
int ba_eval_update(ba_problem *p, const double *X, const double *X_gauge, const double *C, const double *C_gauge,
                   const double *Theta, const double *phi, int flags, double *E_sum, double *gX, double *gC,
                   double *gTheta, double *gphi, int rule, double lr, double gscale, double bc1, double bc2,
                   const double *s_in, double *s_out, const double *r_in, double *r_out, double *X_out, void *stream);
int ba_gauge_points(int64_t N, const double *X_in, double *X_out, const double *gauge, void *stream);
int ba_scale_centres(int64_t K, const double *cam_in, const double *gauge, double *cam_out, void *stream);

/* Copies one double from the device to *h_out and waits for it: the energy read-back the LR schedule
 * branches on (E[iters] at p2w:612 is a host value in the reference). */
int ba_fetch_double(const double *src, double *h_out, void *stream);

/* Energy mailbox: two doubles of host memory mapped into the device.  A ba_eval with BA_EVAL_POST
 * (or ba_post_energy, for a sum completed elsewhere, e.g. by a cross-GPU reduction) makes the GPU
 * store the energy sum there; ba_wait_energy spins on the host until the LATEST post has landed
 * and returns it.  Unlike ba_fetch_double it does not wait for work queued behind the post, so the
 * caller can queue the next update before it reads the energy that the LR schedule (p2w:365-454)
 * branches on.  Fails (instead of spinning forever) if the stream drains without the post. */
int ba_post_energy(ba_problem *p, const double *E_sum, void *stream);
int ba_wait_energy(ba_problem *p, double *h_E_sum, void *stream);

/* The same wait with a pass launched from inside it: ba_eval_update_arm stores the arguments of a ba_eval_update call
 * (nothing is launched, nothing is read or written yet); ba_wait_energy_fire waits for the latest post like
 * ba_wait_energy, forms e = *h_E_sum / divisor (divisor: 2 |V|, the caller's normalisation, so that the comparisons
 * see the very number the caller's schedule will see) and launches the armed pass at once
 *   with X_out      if lo <= e < hi          (*h_fired = 1: an energy that only moves the schedule's snapshot forward),
 *   with X_out_alt  if hi <= e, alt_lo <= e  (*h_fired = 2: no improvement, patience left; X_out_alt may be NULL),
 * otherwise not at all (*h_fired = 0) - in the fired cases the caller's LR schedule (p2w:365-454) would not have
 * intervened, so the next pass starts without the round trip through the host interpreter.  Arming is consumed by
 * the wait either way.  The caller guarantees that the output buffers are free in the case they are used in (each
 * may still be referenced by state that the other outcomes keep alive). */
int ba_eval_update_arm(ba_problem *p, const double *X, const double *X_gauge, const double *C, const double *C_gauge,
                       const double *Theta, const double *phi, int flags, double *E_sum, double *gX, double *gC,
                       double *gTheta, double *gphi, int rule, double lr, double gscale, double bc1, double bc2,
                       const double *s_in, double *s_out, const double *r_in, double *r_out, double *X_out,
                       double *X_out_alt);
int ba_wait_energy_fire(ba_problem *p, double *h_E_sum, double divisor, double lo, double hi, double alt_lo,
                        int *h_fired, void *stream);

/* Durations in milliseconds of the observation-pass kernel of the ba_eval calls made with
 * BA_EVAL_TIMED since the last call of this function (at most 256 are kept), measured with HIP
 * events recorded on the launch stream around that kernel.  Synchronises on the events. */
int ba_eval_timing(ba_problem *p, float *h_ms, int capacity, int *h_count);

/* ---- update rules (p2w:336-360), in place, n doubles; g is multiplied by gscale (= 1/|V|) first.
 * Adam: s = rho1 s + (1-rho1) g; r = rho2 r + (1-rho2) g^2;
 *       x -= lr (s/bc1) / (1e-12 + sqrt(r/bc2)),  bc1 = 1-rho1^t, bc2 = 1-rho2^t (host supplies).
 * `release` (optional, n doubles) receives x before the update: the snapshot lr_manager takes at
 * p2w:397-398, written in the same pass. */
int ba_adam_step(int64_t n, double *x, const double *g, double *s, double *r, double lr,
                 double gscale, double bc1, double bc2, double *release, void *stream);
int ba_sgd_step(int64_t n, double *x, const double *g, double lr, double gscale, double *release,
                void *stream);

/* ---- gauge fix: align_scene + normalize_scene (p2w:270-333) in place.
 * Camera 0 goes to the origin with identity rotation; the longest baseline from camera 0
 * gets length 1.  C and Theta must be one buffer [C (3K) | Theta (3K)].  R_out[K][9] (optional)
 * receives R0^T R_j as align_scene(ret_R=True) does.
 * gauge_out[16]: R0 (9, row-major), C0 (3), scale factor, j1, squared longest baseline or 0, flag (0: the factor is
 * word 12; 1: it is 1 / sqrt(word 14), see ba_camera_step_finalize). */
int ba_gauge_fix(int64_t N, int64_t K, double *X, double *C, double *Theta, double *R_out,
                 double *gauge_out, void *stream);

/* ---- the rotation and gauge helpers as calls of their own (the fit loop uses the fused steps below).
 * ba_rotation_matrices: ryxz (p2w:138-140), R_out[K][9] row-major = Ry(theta_y) Rx(theta_x) Rz(theta_z).
 * ba_rotation_angles:   get_rotations (p2w:254-267), Theta_out[K][3] with theta_x in [-pi/2, pi/2].
 * ba_align_scene:       align_scene (p2w:270-303): camera 0 to the origin with identity rotation.  Exactly one of
 *   Theta_in[K][3], R_in[K][9]; X_out = (X - C0) @ R0, C_out likewise, R_out (optional) = R0^T R, Theta_out its
 *   angles.  Outputs may alias the inputs.  frame_work: 12 doubles of scratch.
 * ba_normalize_scene:   normalize_scene (p2w:306-333): X_out = X * f, C_out = C * f, f = dist_scale / |C[j1] - C[j0]|;
 *   j1 < 0 selects j0 = 0 and the camera farthest from camera 0 (the first among equals, as numpy.argmax).
 *   scale_work[2] receives f and the j1 used. */
int ba_rotation_matrices(int64_t K, const double *Theta, double *R_out, void *stream);
int ba_rotation_angles(int64_t K, const double *R, double *Theta_out, void *stream);
int ba_align_scene(int64_t N, int64_t K, const double *X_in, const double *C_in, const double *Theta_in,
                   const double *R_in, double *X_out, double *C_out, double *Theta_out, double *R_out,
                   double *frame_work, void *stream);
int ba_normalize_scene(int64_t N, int64_t K, const double *X_in, const double *C_in, int64_t j0, int64_t j1,
                       double dist_scale, double *X_out, double *C_out, double *scale_work, void *stream);

/* ---- one optimiser step, fused (p2w:704-712): the same arithmetic as ba_adam_step / ba_sgd_step
 * followed by ba_gauge_fix, in two launches and one pass over each array.  Parameters and moments
 * have an input and an output pointer each: the same buffer (in place) or disjoint ones.  With
 * disjoint buffers the inputs survive, which gives the caller the snapshot lr_manager takes at
 * p2w:397-398 for free, and lets it drop a step it queued before it knew the energy.
 * ba_camera_step: update rule (0 none, 1 SGD, 2 Adam) on the slice [lo, hi) of
 *   cam = [C (3K) | Theta (3K) | phi_x (1)] (g, s, r in the same packing; entries outside the slice
 *   are copied), then, if `gauge`, the camera half of the gauge fix; it also leaves the handle's
 *   camera table and scalars valid for cam_out (see BA_EVAL_REUSE_TABLE).
 * ba_points_step: update rule on X, then, if `gauge` (the record written by ba_camera_step) is
 *   given, X <- ((X - C0) @ R0) * factor. */
int ba_camera_step(ba_problem *p, const double *cam_in, double *cam_out, const double *g, const double *s_in,
                   double *s_out, const double *r_in, double *r_out, int64_t lo, int64_t hi, int rule,
                   double lr, double gscale, double bc1, double bc2, int gauge, double *R_out,
                   double *gauge_out, void *stream);
/* The camera step with the gauge fix that also FINISHES a pass made with BA_EVAL_DEFER_FINALIZE: the pass's energy
 * accumulators -> energy and d E / d phi_x, camera-frame sums -> d E / d C, d E / d Theta, all written to
 * g_out = [gC (3K) | gTheta (3K) | gphi | E_sum]; `post` also posts the energy to the mailbox (see ba_wait_energy).
 * One launch for the whole camera side of an iteration (with several GPUs the sum over the ranks sits between the
 * pass and this call), and no workgroup of it waits for another one:
 *   - cam_out, s_out, r_out must not overlap their inputs (every workgroup reads camera 0's old values);
 *   - the longest baseline is only COLLECTED: gauge_out[14] receives the maximum of the squared baselines by atomic
 *     max and gauge_out[15] = 1, i.e. the factor of this record is 1 / sqrt(gauge_out[14]) (ba_gauge_points,
 *     ba_scale_centres and ba_eval_update read records of both forms).  gauge_out[14] must be ZERO when the kernel
 *     starts: a step zeroes that word of `next_gauge`, the record the caller will pass to the following step
 *     (optional), the very first record is zeroed by the caller;
 *   - the centres in cam_out are left WITHOUT that factor: pass gauge_out as C_gauge to the next ba_eval_update and
 *     as cam_in_gauge to the next ba_camera_step_finalize (NULL: cam_in's centres are final), or apply it with
 *     ba_scale_centres;
 *   - it leaves the rotations of the camera table valid for cam_out but not the rows' centres: the camera-run pass
 *     (ba_eval_update with BA_EVAL_REUSE_TABLE) takes the centres from its C argument, every other user of the
 *     table rebuilds it first by itself. */
int ba_camera_step_finalize(ba_problem *p, const double *cam_in, const double *cam_in_gauge, double *cam_out,
                            double *g_out, int post, const double *s_in, double *s_out, const double *r_in,
                            double *r_out, int64_t lo, int64_t hi, int rule, double lr, double gscale, double bc1,
                            double bc2, double *R_out, double *gauge_out, double *next_gauge, void *stream);

/* Several GPUs, no counterpart in the reference (single process): the sums a pass with BA_EVAL_DEFER_FINALIZE
 * leaves for ba_camera_step_finalize - camera-frame sums (6K doubles) followed by n_partials pairs of energy
 * accumulators {sum r^2, sum m.r} the pass's tiles add to (n_partials as ba_sum_buffer_size reports: 64) - can live
 * in a caller's device buffer of 6K + 2 * n_partials doubles, zeroed once, so that ONE all-reduce of that buffer
 * between the pass and ba_camera_step_finalize adds up the GPUs' shares (the camera step leaves the buffer at zero
 * again).  The buffer stays the caller's. */
int ba_sum_buffer_size(ba_problem *p, int64_t *n_partials);
int ba_set_sum_buffer(ba_problem *p, double *buf, int64_t n_partials);
int ba_points_step(int64_t N, const double *X_in, double *X_out, const double *gX, const double *s_in,
                   double *s_out, const double *r_in, double *r_out, int rule, double lr, double gscale,
                   double bc1, double bc2, const double *gauge, void *stream);

/* ---- linear steps on either side of the bundle adjustment, on the visible pairs only.
 * ba_triangulate replaces get_X_from_C_Theta (p2w:1655-1721): X_out[N][3] from the cameras; points seen
 *   fewer than twice or whose rows have rank < 3 get NaN.  Observations sorted by point: row_ptr[N+1]
 *   (int64) gives every point's range in oj / uv.  rank_rows = max(2 K0, 4) of numpy's rank tolerance
 *   (K0: cameras that see a point of the selection, p2w:1676, 1707).  P_work: 12 K doubles of scratch.
 * ba_resect replaces get_C_Theta_from_X (p2w:1724-1808): C_out[K][3], Theta_out[K][3] from the points
 *   by DLT; cameras that see fewer than 6 points or whose rows have rank < 11 get NaN; observations of
 *   points with NaN coordinates are skipped (p2w:1771-1772).  Observations sorted by camera:
 *   col_ptr[K+1] (int64) gives every camera's range in oi_by_camera / uv_by_camera.  P is the
 *   eigenvector of the smallest eigenvalue of A^T A, the reference's own method for large inputs
 *   (p2w:1789-1792; for small inputs it takes the SVD of A), found by cyclic Jacobi rotations, which
 *   keep the relative accuracy of the small eigenpairs: measured within 2e-11 of the SVD of A in pixel
 *   units.  rank_rows = max(2 N0, 12) (p2w:1743, 1774).
 * phi_x, W, H, normalize as in the reference (normalize: pixel centres shifted to the image centre
 * and divided by max(W, H), p2w:1686-1691). */
int ba_triangulate(int64_t N, int64_t K, const int64_t *row_ptr, const int32_t *oj, const double *uv,
                   const double *C, const double *Theta, double phi_x, double W, double H, int normalize,
                   int64_t rank_rows, double *P_work, double *X_out, void *stream);
/* The same on the observation list a problem handle holds (its points in the engine's internal order, in which a
 * workgroup's points see a short window of cameras whose projection rows are staged in LDS): X_out[N][3] in that
 * order.  rank_rows is counted from the list on first use.  P_work: 12 K doubles of scratch. */
int ba_problem_triangulate(ba_problem *p, const double *C, const double *Theta, double phi_x, int normalize,
                           double *P_work, double *X_out, void *stream);
int ba_resect(int64_t N, int64_t K, const int64_t *col_ptr, const int32_t *oi_by_camera,
              const double *uv_by_camera, const double *X, double phi_x, double W, double H, int normalize,
              int64_t rank_rows, double *C_out, double *Theta_out, void *stream);

/* ---- second-order extension: NO counterpart in the reference (its optimiser is first-order, p2w:336-362).
 * Gauss-Newton / Levenberg-Marquardt on the same energy.  All quantities un-normalised (J^T J, J^T r).
 * ba_eval_jtj: the fused pass of ba_eval plus the blocks of J^T J:
 *   V[N][6]    upper triangle (xx,xy,xz,yy,yz,zz) of sum J_X^T J_X per point
 *   B[K][21]   upper triangle (row-major) of the 6x6 block over (C, Theta) per camera,
 *   bphi[K][6] camera x phi_x column,  hphi[1].
 *   The cross blocks (point x camera, point x phi_x) are not stored: ba_schur_build and
 *   ba_backsubstitute form them from the observations.
 * ba_schur_dim: n = 6K+1 unknowns of the reduced camera system, ld = n rounded up to 64.
 * ba_schur_build: S (ld x ld, row-major; only the lower triangle is written) and rhs (ld) of
 *   (D' - Y^T V'^-1 Y) d = -g_cam + Y^T V'^-1 g_X, primes = diagonal scaled by (1 + lambda);
 *   unknown order: camera j at 6j..6j+5 (C then Theta), phi_x at 6K.  Uses the camera table left by
 *   the last ba_eval_jtj.  Also WRITES Xphi[N][3], the point x phi_x column of J^T J, for
 *   ba_backsubstitute.  with_camera_blocks = 0 leaves out D', -g_cam and the identity padding
 *   and gives only this shard's sum over its points: with several GPUs one rank passes 1, the
 *   others 0, and S, rhs are summed over the ranks (B, bphi, hphi, g_cam already being the
 *   cross-GPU sums).
 * ba_cholesky_solve: blocked Cholesky of S in place (trailing update on the fp64 matrix cores) and
 *   solution of S d = rhs in place; *h_info = 0 or 1 + index of the first non-positive pivot.
 *   Rows and columns n .. ld-1 of S must be the identity and rhs[n .. ld) zero on entry (ba_schur_build leaves
 *   them so).  When ld > n the last of these rows is used as work space (it carries the right-hand side through
 *   the factorisation) and does not hold the identity afterwards.  Blocks that are structurally zero are not
 *   computed: the first occupied 64-column block of every 64-row block is read off S before the factorisation
 *   (Cholesky creates no entries left of it), which costs one pass over the lower triangle and one host
 *   synchronisation; the environment variable BA_CHOL_DENSE=1 computes every tile.  A long narrow band is cut
 *   into independent chains that are factorised side by side (the 64-blocks of S are reordered in place: chains
 *   first, then the separators between them, then the border; the separators, a band again, are cut the same way
 *   while that saves dependent steps - ba_cholesky_last_plan reports the split): what S holds on return is then
 *   the factor of the reordered matrix, and a pivot index in *h_info counts in that order; the solution comes back
 *   in the caller's order.  The sums into the blocks shared by the chains are atomic, so the last bits of the solution can differ
 *   from call to call.  BA_CHOL_NO_CHAINS=1 keeps the natural order (S = the factor L on return).
 * ba_backsubstitute: dX[N][3] = V'^-1 (-g_X - W d_cam - Xphi d_phi); dcam_packed (optional, 6K+1)
 *   receives d reordered to [dC (3K) | dTheta (3K) | dphi].  The trial point of the damped step comes out of
 *   the same two launches when asked for: X_trial[N][3] (optional) = X + dX, cam_trial (optional, 6K+1) =
 *   cam + dcam with cam = the current packed [C | Theta | phi_x]; both are buffers of their own, so the
 *   current parameters stay what a rejected step goes back to.
 * ba_build_camera_table: rebuilds the handle's camera table (and the pinhole scalars) from the given
 *   parameters without a pass: after a rejected trial the table holds the trial's cameras, while the
 *   gradient and the blocks of the accepted point are still in the caller's buffers. */
int ba_eval_jtj(ba_problem *p, const double *X, const double *C, const double *Theta, const double *phi,
                int flags, double *E_sum, double *gX, double *gC, double *gTheta, double *gphi,
                double *V, double *B, double *bphi, double *hphi, void *stream);
int ba_schur_dim(int64_t K, int64_t *n, int64_t *ld);
int ba_schur_build(ba_problem *p, const double *X, double lambda, const double *V, double *Xphi,
                   const double *gX, const double *B, const double *bphi, const double *hphi,
                   const double *gC, const double *gTheta, const double *gphi, int with_camera_blocks,
                   double *S, double *rhs, void *stream);
int ba_cholesky_solve(int64_t n, int64_t ld, double *S, double *rhs, int *h_info, void *stream);
/* The same two steps for a caller that builds and solves one system after the other in the same S (an LM loop):
 * at 20 000 cameras S has 115 GB, and clearing it for the next build and reading it for the block profile cost
 * more than building and factorising the band that it holds.
 * ba_schur_build_tracked: also writes block_first[ld / 64] (device): a lower bound of the first occupied
 *   64-column block of every 64-row block of S, collected while S is built (with several GPUs: the minimum over
 *   the ranks belongs to the summed S); S_is_zero != 0: S holds zeros on entry and is not cleared.
 * ba_cholesky_solve_tracked: block_first (or NULL) takes the place of the pass over S that reads the profile;
 *   clear_on_return != 0: S holds zeros again on return (only the blocks the factor occupied are written). */
int ba_schur_build_tracked(ba_problem *p, const double *X, double lambda, const double *V, double *Xphi,
                           const double *gX, const double *B, const double *bphi, const double *hphi,
                           const double *gC, const double *gTheta, const double *gphi, int with_camera_blocks,
                           double *S, double *rhs, int *block_first, int S_is_zero, void *stream);
int ba_cholesky_solve_tracked(int64_t n, int64_t ld, double *S, double *rhs, const int *block_first,
                              int clear_on_return, int *h_info, void *stream);
/* The same solve for a loop whose systems all have ONE block profile (an LM loop on one scene), without the two
 * waits of the call above (for the profile before, for the pivot verdict after): everything is queued on `stream`
 * and the call returns.  The first call on a device (and the first after n, ld or clear_on_return changed, or after
 * a verdict of -1) reads block_first and waits for it once; later calls take that profile for granted and a kernel
 * compares block_first with it before anything is factorised.
 * ba_cholesky_queued_result: the verdict of the last queued call on the current device, to be called once `stream`
 *   has been waited for by any means (the verdict travels through a word of mapped host memory, nothing is copied
 *   or waited for here): *h_info = 0, or 1 + index of the first non-positive pivot, or -1: a row of S started left
 *   of the assumed profile - S and rhs are then undefined, S must be cleared and built again, and the next queued
 *   call reads the profile afresh.  Calls on one device share a workspace: queue them on one stream. */
int ba_cholesky_solve_queued(int64_t n, int64_t ld, double *S, double *rhs, const int *block_first,
                             int clear_on_return, void *stream);
int ba_cholesky_queued_result(int *h_info);
/* Several GPUs add up S.  Only its envelope holds anything (block row ib: the 64-blocks block_first[ib] .. ib, with
 * block_first the minimum over the ranks of what ba_schur_build_tracked reported), so that is what travels:
 * ba_schur_envelope: offsets[ld / 64 + 1] (device) = blocks before every block row; *h_blocks = their total (waits).
 * ba_schur_envelope_copy: the envelope's blocks to packed[blocks][64][64] (to_packed != 0) or back into S. */
int ba_schur_envelope(int64_t ld, const int *block_first, int64_t *offsets, int64_t *h_blocks, void *stream);
int ba_schur_envelope_copy(int64_t ld, double *S, const int *block_first, const int64_t *offsets, int64_t blocks,
                           double *packed, int to_packed, void *stream);
/* How the last solve on the current device cut its band (for tests and reports): *n_levels levels of independent
 * chains, one nested behind the other's chains; levels[4 k .. 4 k + 3] = first 64-block of level k's matrix, number of
 * chains, 64-blocks per chain, 64-blocks per separator (at most `capacity` levels are written).  0 levels: the natural
 * order was kept. */
int ba_cholesky_last_plan(int64_t *levels, int capacity, int *n_levels);
int ba_backsubstitute(ba_problem *p, const double *X, double lambda, const double *V, const double *Xphi,
                      const double *gX, const double *d, double *dcam_packed, double *dX, const double *cam,
                      double *X_trial, double *cam_trial, void *stream);
int ba_build_camera_table(ba_problem *p, const double *C, const double *Theta, const double *phi, void *stream);

/* ---- transform (p2w:192-216) on a list of pairs: model pixels out[n][2], NaN behind the camera
 * (delete_back_points) and optionally outside [0,W)x[0,H) (delete_boundary_points).
 * i == NULL means the dense N x K grid in C order (n = N*K). */
int ba_project(int64_t N, int64_t K, int64_t n, const int32_t *i, const int32_t *j,
               const double *X, const double *C, const double *Theta, double phi, double W, double H,
               int delete_back_points, int delete_boundary_points, double *out, void *stream);

/* ---- ba_project without any allocation.  No counterpart in the reference: it exists for callers that cannot
 * allocate device memory at the time of the call - code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  Same arguments and results as ba_project; the camera table (K x 16 doubles), which ba_project
 * keeps in buffers it allocates on demand (one per GPU and stream, grown when K grows), lives in a fixed array of the
 * library instead (2 MiB per GPU, held until the process ends), so K is limited to 16384 cameras: a larger K returns
 * BA_ERR_ARG and nothing is launched.  The call queues two kernels on `stream` and returns without waiting for them.
 * All calls on one GPU share that array: calls that overlap on one GPU (different streams, or threads) race; queue
 * them on one stream.  The GPU the work is meant for must be the current device of the calling thread.  The first
 * call on a GPU looks the array's address up (a runtime query, not a stream operation): make it outside any graph
 * capture. */
int ba_project_noalloc(int64_t N, int64_t K, int64_t n, const int32_t *i, const int32_t *j,
                       const double *X, const double *C, const double *Theta, double phi, double W, double H,
                       int delete_back_points, int delete_boundary_points, double *out, void *stream);

/* ---- error (p2w:98-111): sums[0] = sum of squared differences over pairs visible in the data
 * (pix_xy[n][2] int32, x >= 0) whose model pixel is not NaN; sums[1] = number of such pairs. */
int ba_reprojection_error(int64_t n, const int32_t *pix_xy, const double *model, double *sums,
                          void *stream);

/* ---- get_H_diff_matrix (p2w:1945-1995): for every pair of cameras with at least min_common (reference: 10)
 * common points, the homography between the two views by DLT (p2w:1976-1987) and the mean squared transfer
 * error of the common points (p2w:1989-1993) -> H_diff[K][K], symmetric, NaN for all other pairs and the
 * diagonal.  Observations sorted by camera, points ascending within a camera: col_ptr[K+1], oi_by_camera[n],
 * uv_by_camera[n][2] (pixel centres); normalize: centres -> (centre - (W, H)/2) / max(W, H) (p2w:1963-1964).
 * K <= 65535. */
int ba_homography_diff_matrix(int64_t K, const int64_t *col_ptr, const int32_t *oi_by_camera,
                              const double *uv_by_camera, double W, double H, int normalize, int min_common,
                              double *H_diff, void *stream);

/* ---- joint alignment of subscenes by gradient descent (fit_subscenes_GD, p2w:729-949) on the present entries.
 * Input of the reference: X[N][P][3], NaN where subscene j does not hold point i.  Here: the present entries
 * sorted by point: row_ptr[N+1], entry_subscene[n], entry_x[n][3].  params = [S (P) | T (3P) | Psi (3P)].
 * ba_subscenes_eval (p2w:813-818, 848-849, 886-924): sums[1 + 7P] = [sum of squared differences | dE/dS |
 *   dE/dT | dE/dPsi], without the reference's factors 1/(2n) and 2/n; table_work: 40 P doubles of scratch.
 * ba_subscenes_step (p2w:926-937): SGD (rule 1) or Adam (rule 2; moments s, r of 7P doubles each, bc = 1 - rho^t)
 *   with the gradient 2/n x sums, then the gauge of the reference (subscene 0 becomes the identity); writes
 *   new buffers, the inputs stay valid (a rejected step is dropped by the caller).
 * ba_subscenes_transform (p2w:943-944): x_new[n][3] = S_j R_j x + T_j for every entry. */
int ba_subscenes_eval(int64_t N, int64_t P, const int64_t *row_ptr, const int32_t *entry_subscene,
                      const double *entry_x, const double *params, double *table_work, double *sums,
                      void *stream);
int ba_subscenes_step(int64_t P, int rule, double lr, double bc1, double bc2, double n_entries,
                      const double *sums, const double *params, double *params_out, const double *s_in,
                      double *s_out, const double *r_in, double *r_out, void *stream);
int ba_subscenes_transform(int64_t P, int64_t n_entries, const int32_t *entry_subscene, const double *entry_x,
                           const double *params, double *table_work, double *x_new, void *stream);

/* ---- sum over the GPUs of one node (the exchange step of the sharded fit loop; the reference runs on one
 * device and has no counterpart; p2w:654-702 are the sums that are added up).  Every rank creates an exchange
 * (a window of its GPU's memory: four slots of `capacity_doubles`), hands the 64-byte handle of its
 * window to the other ranks of the node (any host channel) and maps theirs; after a host barrier
 * ba_exchange_sum replaces buf[0..n) by the sum over the ranks, in rank order (identical bits on every rank),
 * as ONE kernel on `stream`: no collective library, no second stream.  All ranks must call it the same number of
 * times with the same n.  A value with the bit pattern 0x7ff4a5a57ff4a5a5 (a signalling NaN, the windows' mark for
 * "not yet published") is summed as a quiet NaN.  A rank that waits longer than the limit (default 2 s) for another gives up: the
 * contents of its buffer are then UNDEFINED (the chunks of workgroups that had finished their wait already hold
 * sums, the others the rank's own values), ba_exchange_status (and every later ba_exchange_sum) returns BA_ERR_STATE
 * (ba_exchange_set_wait changes the limit, at most 30 s).  Calls of one exchange must not overlap on a GPU: queue
 * them on one stream.  ba_exchange_destroy frees the window the other ranks read from inside their kernels: call
 * it only after every rank has finished its last sum (device synchronisation, then a host barrier). */
typedef struct ba_exchange ba_exchange;
int ba_exchange_create(int rank, int world, int64_t capacity_doubles, ba_exchange **out);
int ba_exchange_export(ba_exchange *x, void *h_handle64);
int ba_exchange_import(ba_exchange *x, int rank, const void *h_handle64);
int ba_exchange_sum(ba_exchange *x, double *buf, int64_t n, void *stream);
int ba_exchange_status(ba_exchange *x);
int ba_exchange_set_wait(ba_exchange *x, double seconds);
void ba_exchange_destroy(ba_exchange *x);

#ifdef __cplusplus
}
#endif
#endif /* BA_MI355X_H */
