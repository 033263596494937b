/*
 * C restatement of one fit_scene iteration of the reference (modules/pix2world.py, "p2w"),
 * restricted to the visible (point, camera) pairs.
 *
 * TEST INFRASTRUCTURE ONLY: this file is the checker and the CPU baseline.  It is compiled
 * into oracle/libba_oracle.so and may be loaded from tests/, from __graft_entry__.smoke() and
 * from the cpu_baseline leg of bench.py, never from the product package.  It is pinned by
 * tests/test_oracle_golden.py against vectors captured from the reference itself.
 *
 * Scalar double precision; follows the reference's own formulas (the 27 entries of dR/dTheta are
 * formed per camera exactly as p2w:678-692 does) rather than the rearrangement the GPU kernels
 * use, so that the two are independent derivations.  One thread by default (what the parity tests
 * use); orc_set_threads(n) spreads the observation loop, the update rule and the gauge fix over n
 * OpenMP threads for the CPU baseline (contiguous ranges of points per thread, camera sums per
 * thread added up in thread order: deterministic for a given n).
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int g_threads = 1;
void orc_set_threads(int n) { g_threads = n > 0 ? n : 1; }
int orc_get_threads(void) { return g_threads; }

static void mat3_mul(const double *a, const double *b, double *c) {
    for (int r = 0; r < 3; ++r)
        for (int s = 0; s < 3; ++s)
            c[3 * r + s] = a[3 * r] * b[s] + a[3 * r + 1] * b[3 + s] + a[3 * r + 2] * b[6 + s];
}

/* Rx, Ry, Rz (p2w:114-136) */
static void rot_factors(const double *th, double *Rx, double *Ry, double *Rz) {
    const double cx = cos(th[0]), sx = sin(th[0]), cy = cos(th[1]), sy = sin(th[1]), cz = cos(th[2]),
                 sz = sin(th[2]);
    memset(Rx, 0, 72);
    memset(Ry, 0, 72);
    memset(Rz, 0, 72);
    Rx[0] = 1.0; Rx[4] = cx; Rx[8] = cx; Rx[7] = sx; Rx[5] = -sx;
    Ry[4] = 1.0; Ry[0] = cy; Ry[8] = cy; Ry[2] = sy; Ry[6] = -sy;
    Rz[8] = 1.0; Rz[0] = cz; Rz[4] = cz; Rz[3] = sz; Rz[1] = -sz;
}

/* R = Ry Rx Rz (p2w:138-140) */
void orc_ryxz(int64_t K, const double *Theta, double *R) {
    double Rx[9], Ry[9], Rz[9], t[9];
    for (int64_t j = 0; j < K; ++j) {
        rot_factors(Theta + 3 * j, Rx, Ry, Rz);
        mat3_mul(Ry, Rx, t);
        mat3_mul(t, Rz, R + 9 * j);
    }
}

/* dR/dtheta_l, l = x, y, z (p2w:678-692): out[j][l] is a 3x3 matrix */
static void d_rot(const double *th, double *dR /* [3][9] */) {
    double Rx[9], Ry[9], Rz[9], D[9], t[9];
    rot_factors(th, Rx, Ry, Rz);
    memset(D, 0, 72);
    D[4] = D[8] = Rx[5]; D[7] = Rx[4]; D[5] = -D[7];
    mat3_mul(Ry, D, t);
    mat3_mul(t, Rz, dR);
    memset(D, 0, 72);
    D[0] = D[8] = Ry[6]; D[2] = Ry[0]; D[6] = -D[2];
    mat3_mul(D, Rx, t);
    mat3_mul(t, Rz, dR + 9);
    memset(D, 0, 72);
    D[0] = D[4] = Rz[1]; D[3] = Rz[0]; D[1] = -D[3];
    mat3_mul(Ry, Rx, t);
    mat3_mul(t, D, dR + 18);
}

/*
 * Energy (p2w:612) and gradients (p2w:654-702) on the visible pairs.
 * i, j: int32[nV]; uv: double[nV][2] pixel centres.  Outputs are NORMALISED by n_visible like the
 * reference's D['E', *]; gX[N][3], gC[K][3], gTheta[K][3], gphi[1].  Returns E.
 */
double orc_energy_and_gradients(int64_t N, int64_t K, int64_t nV, int64_t n_visible, const int32_t *oi,
                                const int32_t *oj, const double *uv, const double *X, const double *C,
                                const double *Theta, double phi_x, double W, double H, double *gX, double *gC,
                                double *gTheta, double *gphi) {
    double *R = (double *)malloc(sizeof(double) * 9 * K);
    double *dR = (double *)malloc(sizeof(double) * 27 * K);
    orc_ryxz(K, Theta, R);
    for (int64_t j = 0; j < K; ++j) d_rot(Theta + 3 * j, dR + 27 * j);
    memset(gX, 0, sizeof(double) * 3 * N);
    memset(gC, 0, sizeof(double) * 3 * K);
    memset(gTheta, 0, sizeof(double) * 3 * K);
    const double f = W / 2.0 / tan(phi_x / 2.0);
    const double dphi = -f / sin(phi_x);
    const double inv_nv = 1.0 / (double)n_visible;
    double e_sum = 0.0, gp = 0.0;
    const int T = g_threads;
    /* thread t takes the observations [cut[t], cut[t + 1]): equal shares moved forward to the next point's first
     * observation, so that no two threads touch one row of gX */
    int64_t *cut = (int64_t *)malloc(sizeof(int64_t) * (T + 1));
    for (int t = 0; t <= T; ++t) {
        int64_t e = nV * t / T;
        while (e > 0 && e < nV && oi[e] == oi[e - 1]) ++e;
        cut[t] = e;
    }
    double *cam_part = T > 1 ? (double *)calloc((size_t)T * 6 * K, sizeof(double)) : NULL;
    double *sc_part = (double *)calloc((size_t)T * 2, sizeof(double));
#pragma omp parallel num_threads(T)
    {
        const int t = omp_get_thread_num();
        double *gC_t = T > 1 ? cam_part + (size_t)t * 6 * K : gC;
        double *gTheta_t = T > 1 ? gC_t + 3 * K : gTheta;
        double e_sum_t = 0.0, gp_t = 0.0;
        for (int64_t e = cut[t]; e < cut[t + 1]; ++e) {
            const int64_t i = oi[e], j = oj[e];
            const double *Rj = R + 9 * j, *x = X + 3 * i, *c = C + 3 * j;
            const double d[3] = {x[0] - c[0], x[1] - c[1], x[2] - c[2]};
            double q[3];
            for (int k = 0; k < 3; ++k) q[k] = d[0] * Rj[k] + d[1] * Rj[3 + k] + d[2] * Rj[6 + k];
            const double m0 = q[0] / q[2], m1 = q[1] / q[2];
            const double r0 = f * m0 + W / 2.0 - uv[2 * e], r1 = f * m1 + H / 2.0 - uv[2 * e + 1];
            e_sum_t += r0 * r0 + r1 * r1;
            /* A = d model / d q (p2w:654-655) */
            const double A[2][3] = {{f / q[2], 0.0, -f / (q[2] * q[2]) * q[0]}, {0.0, f / q[2], -f / (q[2] * q[2]) * q[1]}};
            const double w0 = r0 * inv_nv, w1 = r1 * inv_nv;
            /* J_X = A R^T (p2w:657), dE/dX += w J_X, dE/dC -= w J_X */
            for (int k = 0; k < 3; ++k) {
                const double jx0 = A[0][0] * Rj[3 * k] + A[0][1] * Rj[3 * k + 1] + A[0][2] * Rj[3 * k + 2];
                const double jx1 = A[1][0] * Rj[3 * k] + A[1][1] * Rj[3 * k + 1] + A[1][2] * Rj[3 * k + 2];
                const double g = w0 * jx0 + w1 * jx1;
                gX[3 * i + k] += g;
                gC_t[3 * j + k] -= g;
            }
            /* J_Theta = A (dR/dtheta)^T d (p2w:694-695) */
            for (int l = 0; l < 3; ++l) {
                const double *D = dR + 27 * j + 9 * l;
                double dq[3];
                for (int k = 0; k < 3; ++k) dq[k] = d[0] * D[k] + d[1] * D[3 + k] + d[2] * D[6 + k];
                const double jt0 = A[0][0] * dq[0] + A[0][2] * dq[2];
                const double jt1 = A[1][1] * dq[1] + A[1][2] * dq[2];
                gTheta_t[3 * j + l] += w0 * jt0 + w1 * jt1;
            }
            gp_t += w0 * (dphi * m0) + w1 * (dphi * m1); /* p2w:701-702 */
        }
        sc_part[2 * t] = e_sum_t;
        sc_part[2 * t + 1] = gp_t;
    }
    for (int t = 0; t < T; ++t) {
        e_sum += sc_part[2 * t];
        gp += sc_part[2 * t + 1];
        if (T > 1)
            for (int64_t k = 0; k < 3 * K; ++k) {
                gC[k] += cam_part[(size_t)t * 6 * K + k];
                gTheta[k] += cam_part[(size_t)t * 6 * K + 3 * K + k];
            }
    }
    free(cut);
    free(cam_part);
    free(sc_part);
    gphi[0] = gp;
    free(R);
    free(dR);
    return e_sum / (2.0 * (double)n_visible);
}

/* Adam (p2w:344-360), in place; t is the step counter before the update */
void orc_adam(int64_t n, double *x, const double *g, double *s, double *r, double lr, int64_t t) {
    const double rho1 = 0.9, rho2 = 0.999;
    const double bc1 = 1.0 - pow(rho1, (double)t), bc2 = 1.0 - pow(rho2, (double)t);
#pragma omp parallel for num_threads(g_threads) schedule(static) if (n > 4096)
    for (int64_t k = 0; k < n; ++k) {
        s[k] = rho1 * s[k] + (1.0 - rho1) * g[k];
        r[k] = rho2 * r[k] + (1.0 - rho2) * (g[k] * g[k]);
        x[k] = x[k] - lr * (s[k] / bc1) / (1e-12 + sqrt(r[k] / bc2));
    }
}

/* align_scene + normalize_scene (p2w:270-333), in place */
void orc_gauge_fix(int64_t N, int64_t K, double *X, double *C, double *Theta) {
    double R0[9], Rj[9], Rn[9], c0[3] = {C[0], C[1], C[2]};
    orc_ryxz(1, Theta, R0);
    for (int64_t j = 0; j < K; ++j) {
        orc_ryxz(1, Theta + 3 * j, Rj);
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) Rn[3 * a + b] = R0[a] * Rj[b] + R0[3 + a] * Rj[3 + b] + R0[6 + a] * Rj[6 + b];
        const double Cx = sqrt(Rn[3] * Rn[3] + Rn[4] * Rn[4]);
        Theta[3 * j] = atan2(-Rn[5], Cx);
        Theta[3 * j + 1] = atan2(Rn[2] / Cx, Rn[8] / Cx);
        Theta[3 * j + 2] = atan2(Rn[3] / Cx, Rn[4] / Cx);
        const double d[3] = {C[3 * j] - c0[0], C[3 * j + 1] - c0[1], C[3 * j + 2] - c0[2]};
        for (int k = 0; k < 3; ++k) C[3 * j + k] = d[0] * R0[k] + d[1] * R0[3 + k] + d[2] * R0[6 + k];
    }
    double best = -1.0;
    for (int64_t j = 1; j < K; ++j) {
        const double nn = C[3 * j] * C[3 * j] + C[3 * j + 1] * C[3 * j + 1] + C[3 * j + 2] * C[3 * j + 2];
        if (nn > best) best = nn;
    }
    const double factor = 1.0 / sqrt(best);
    for (int64_t k = 0; k < 3 * K; ++k) C[k] *= factor;
#pragma omp parallel for num_threads(g_threads) schedule(static) if (N > 4096)
    for (int64_t i = 0; i < N; ++i) {
        const double d[3] = {X[3 * i] - c0[0], X[3 * i + 1] - c0[1], X[3 * i + 2] - c0[2]};
        for (int k = 0; k < 3; ++k) X[3 * i + k] = (d[0] * R0[k] + d[1] * R0[3 + k] + d[2] * R0[6 + k]) * factor;
    }
}

/* One full iteration as the reference orders it: evaluate, update X, C, Theta, phi_x with Adam,
 * gauge fix (p2w:606-712 without the learning-rate bookkeeping).  State arrays are caller-owned:
 * s*, r* Adam moments; work arrays g* gradients.  Returns E before the update. */
double orc_iteration(int64_t N, int64_t K, int64_t nV, const int32_t *oi, const int32_t *oj, const double *uv,
                     double *X, double *C, double *Theta, double *phi_x, double W, double H, double lr, int64_t t,
                     double *gX, double *gC, double *gTheta, double *gphi, double *sX, double *rX, double *sC,
                     double *rC, double *sT, double *rT, double *sP, double *rP) {
    const double E = orc_energy_and_gradients(N, K, nV, nV, oi, oj, uv, X, C, Theta, phi_x[0], W, H, gX, gC, gTheta,
                                              gphi);
    orc_adam(3 * N, X, gX, sX, rX, lr, t);
    orc_adam(3 * K, C, gC, sC, rC, lr, t);
    orc_adam(3 * K, Theta, gTheta, sT, rT, lr, t);
    orc_adam(1, phi_x, gphi, sP, rP, lr, t);
    orc_gauge_fix(N, K, X, C, Theta);
    return E;
}
