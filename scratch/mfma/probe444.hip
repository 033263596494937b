// Finds the lane layout of v_mfma_f64_4x4x4_4b_f64 by brute force: every operand's lane index is split
// into three 2-bit fields (bits 0-1, 2-3, 4-5) and every assignment of (row/col, k, block) to the fields is tried.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
__global__ void k(const double *a, const double *b, double *d) {
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}
static const int perms[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
// lane holding the element whose three coordinates are c[0], c[1], c[2], coordinate q sitting in field perm[q]
static int lane_of(const int *perm, int c0, int c1, int c2) { return (c0 << (2 * perm[0])) | (c1 << (2 * perm[1])) | (c2 << (2 * perm[2])); }
int main() {
    double ha[64], hb[64], hd[64], *da, *db, *dd;
    srand(1);
    for (int l = 0; l < 64; ++l) { ha[l] = rand() % 17 - 8; hb[l] = rand() % 13 - 6; }
    (void)hipMalloc(&da, 512); (void)hipMalloc(&db, 512); (void)hipMalloc(&dd, 512);
    (void)hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
    k<<<1, 64>>>(da, db, dd);
    (void)hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
    int found = 0;
    // A: coordinates (i, k, blk); B: (j, k, blk); D: (i, j, blk)
    for (int pa = 0; pa < 6; ++pa)
        for (int pb = 0; pb < 6; ++pb)
            for (int pd = 0; pd < 6; ++pd) {
                bool ok = true;
                for (int blk = 0; blk < 4 && ok; ++blk)
                    for (int i = 0; i < 4 && ok; ++i)
                        for (int j = 0; j < 4 && ok; ++j) {
                            double s = 0;
                            for (int kk = 0; kk < 4; ++kk) s += ha[lane_of(perms[pa], i, kk, blk)] * hb[lane_of(perms[pb], j, kk, blk)];
                            ok = fabs(s - hd[lane_of(perms[pd], i, j, blk)]) < 1e-9;
                        }
                if (ok) {
                    ++found;
                    printf("MATCH: A[i][k] of block b at lane (i << %d) | (k << %d) | (b << %d);  B[k][j] at (j << %d) | (k << %d) | (b << %d);  D[i][j] at (i << %d) | (j << %d) | (b << %d)\n",
                           2 * perms[pa][0], 2 * perms[pa][1], 2 * perms[pa][2], 2 * perms[pb][0], 2 * perms[pb][1], 2 * perms[pb][2],
                           2 * perms[pd][0], 2 * perms[pd][1], 2 * perms[pd][2]);
                }
            }
    if (!found) {
        printf("no layout of this family matches; raw results:\n");
        for (int l = 0; l < 64; ++l) printf("%g%c", hd[l], (l & 15) == 15 ? '\n' : ' ');
        printf("a:\n");
        for (int l = 0; l < 64; ++l) printf("%g%c", ha[l], (l & 15) == 15 ? '\n' : ' ');
        printf("b:\n");
        for (int l = 0; l < 64; ++l) printf("%g%c", hb[l], (l & 15) == 15 ? '\n' : ' ');
    }
    return 0;
}
