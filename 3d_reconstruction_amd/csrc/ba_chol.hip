// Dense fp64 Cholesky for the reduced camera system (no counterpart in the reference, whose optimiser
// is first-order only, p2w:336-362).
//
// Storage: lower triangle, row-major, leading dimension ld (a multiple of 64); rows/columns past n
// are the identity.  Three-level right-looking factorisation:
//   panel  (256 columns): its diagonal block per 64-column block (potrf64, then two small launches for the
//          blocks below and behind inside the diagonal block), then ONE launch for all rows below the
//          diagonal block (k_panel_rows: every wave keeps a strip of 16 rows in registers across the
//          whole blocked triangular solve).  O(n 256^2) flops per panel, latency-bound, small next to the
//          updates below.
//   outer panel (1024 columns = 4 panels): after each panel only the remaining columns of the outer
//          panel are updated with it; everything behind the outer panel is updated once, with all
//          1024 columns as the inner dimension.  What this buys is traffic: an update reads and
//          writes every entry of C it touches once, i.e. 16 bytes per 2 x (inner dimension) flops;
//          with 256 columns that is 1 byte per 32 flop, 1.9 TB/s at the rate the matrix cores reach
//          and not overlapped with them (measured: the read-modify-write of C took a quarter of
//          the kernel's time), with 1024 columns a quarter of that.
//   update kernel: C -= P P^T on 128 x 128 tiles, operands staged through LDS in 16-column chunks,
//          on v_mfma_f64_4x4x4_4b_f64.
// The inverse of every diagonal block is kept (trsm is a product with it, and so are the two
// triangular solves of the right-hand side).
//
// Structural zeros.  The storage is dense, the work is not: the reduced camera system of a video is block-banded
// (two cameras are coupled only if they see a common point) plus the dense row of the shared field of view.
// Cholesky factorisation creates no entries left of a row's first entry (the envelope), so before factorising,
// the first occupied 64-column block of every 64-row block is read off the matrix (k_block_profile), and from it
// reach[k] = the last block row that has an entry at or left of column block k.  Rows behind reach[k] hold zeros
// in a panel that ends at block k: the rows kernel and the update kernel leave them out, by grid size (their block
// index skips from the band to the dense border - the block rows of the last unknown and the padding, and before
// them any last rows that reach far back, as the last cameras of a closed path do - through RowSel).  A dense
// matrix gives reach = everything and the plain factorisation.  BA_CHOL_DENSE=1 turns the skipping off.
//
// Independent chains.  What a banded factorisation costs is not flops but its chain of dependent launches: about
// 45 us per 64 columns (diagonal block, the two steps inside the panel's diagonal block, rows, update), 8.5 ms for the
// 188 blocks of n = 12 001 whatever the band holds.  A band of width bw blocks falls apart when bw consecutive
// blocks are taken out of it: the pieces on either side share no entry.  So a long narrow band is cut into P chains
// of s blocks with a separator of bw blocks behind each, and the blocks are reordered - in place, only the occupied
// ones move (k_move_blocks) - to
//     chain 0 | chain 1 | ... | chain P-1 | separator 0 | ... | separator P-1 | rest of the band | border.
// The chains are factorised side by side: every launch of the panel loop covers the same block of all P chains
// (grid dimension y, or z for the update), so the loop has s steps instead of P s.  The rows a chain's panel
// touches behind the chain are its two neighbouring separators and the border (RowSel: band, window, tail);
// everything behind the chains is then factorised as a dense matrix of P bw + (rest) + (border) blocks.  Tiles of
// the update that lie in the columns of separators or border collect the contributions of several chains in one
// launch: those are added with atomics.  P and s minimise the number of dependent block steps,
// s + (blocks behind the chains); n = 12 001 with bw = 1: 14 chains of 12 blocks and 20 blocks behind them, 32 steps
// instead of 188 (9.1 -> 2.1 ms); n = 120 001: 5.5 ms.  The right-hand side enters and the solution leaves in the
// caller's order (k_rhs_into_row, k_solution_back).  BA_CHOL_NO_CHAINS=1 keeps the natural order.
//
// Levels.  What is behind the chains is again a band with a border: a separator is coupled to its two neighbours only
// (through the chain between them), the border stays the border.  Its block profile follows from the matrix's by
// symbolic elimination on the host, and it is cut into chains of its own - a second symmetric reordering, of the
// matrix behind the first level's chains only, done on the device when the first level's chains are finished - and
// so on while a level saves four block steps or more (BA_CHOL_LEVELS=1: one level).  n = 12 001: 14 x 12, then 3 x 4
// of the 14 separators, 8 blocks behind them: 24 dependent steps; n = 120 001: 48 x 38, then 5 x 8: 58 steps
// instead of 90.  The backward solve walks the levels from the inside out (k_back_tail, then per level
// k_back_tail_to_chains and k_back_chains).
//
#include "ba_device.h"

#include <stdlib.h>
#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

namespace ba {

// What a call needs beyond S and rhs - the inverses of the diagonal blocks, the solution vector, the lists that steer
// the block moves and the clearing - lives in a workspace per device that grows on demand and is kept until the process
// ends (at n = 120 001: 61 MB of inverses, 123 MB of gathered blocks): a Levenberg-Marquardt loop calls once per
// iteration with the same matrix, and allocating and freeing each time cost more than the short launches in between.
// The lists are uploaded again only when the block profile they were built from has changed.
// (Replaying the launches of a call - some 130 at n = 12 001, 340 at n = 120 001 - as a captured HIP graph was
// measured too: 4.0 against 3.8 ms and 13.7 against 13.4 ms per iteration, slower than launching them one by one.)
struct CholCache {
    struct Buf {
        void *p = nullptr;
        size_t bytes = 0;
    };
    std::mutex mu;
    Buf linv, tmp, info, newpos, moves, T, runs, first, guard;
    // what the lists in `newpos`, `moves` and `runs` were built for
    bool lists_valid = false;
    bool guard_valid = false;       // `guard` holds `profile` (the queued form compares the caller's profile with it)
    int *result_host = nullptr, *result_dev = nullptr;      // one word of mapped host memory: verdict of the queued form
    // second stream of the chains' look-ahead (see factor() in solve_impl) and the events that tie it to the caller's
    hipStream_t aux = nullptr, aux2 = nullptr;
    hipEvent_t ev_diag = nullptr, ev_rows = nullptr, ev_join = nullptr, ev_nc_rows = nullptr, ev_join2 = nullptr;
    int64_t n = 0, ld = 0;
    std::vector<int64_t> split;     // (first block, chains, blocks of a chain, blocks of a separator) of every level
    int clear = 0;
    std::vector<int> profile;
    hipError_t grow(Buf &b, size_t bytes) {
        if (b.bytes >= bytes) return hipSuccess;
        lists_valid = false;
        hipFree(b.p);
        b.p = nullptr, b.bytes = 0;
        const hipError_t e = hipMalloc(&b.p, bytes);
        if (e == hipSuccess) b.bytes = bytes;
        return e;
    }
};
static CholCache g_chol_cache[16];

constexpr int NB = 64;    // block of the panel factorisation
constexpr int PB = 256;   // panel width
constexpr int OB = 1024;  // outer panel = inner dimension of the update of everything behind it
constexpr int TS = 128;   // tile of the trailing update
constexpr int KC = 16;    // columns of the panel staged in LDS at a time
constexpr int LDT = TS + 4;    // row pitch of the staged (column-major) panel chunks in LDS

static int cdiv_c(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Which rows (and, for the update kernel, columns) a launch covers: logical unit l (a 64-row block or a 128-row tile,
// counted from the launch's first row) -> row offset from that first row.  Three runs of consecutive units:
//   band    l < n1:            unit l of the band that follows the panel; a chain's band lies chain_rows rows (and
//                              columns) behind the band of the chain before it (see "Independent chains" above)
//   window  l < n1 + n2:       the separator rows on either side of the chain, win_rows + win_step max(chain - 1, 0)
//   tail    the rest:          the dense border, from tail_rows on
struct RowSel {
    int n1 = 0x7fffffff, n2 = 0;
    int64_t chain_rows = 0, win_rows = 0, win_step = 0, tail_rows = 0;
    __host__ __device__ int64_t operator()(int64_t l, int unit_rows, int chain) const {
        if (l < n1) return unit_rows * l + chain * chain_rows;
        if (l < (int64_t)n1 + n2) return win_rows + win_step * (chain > 0 ? chain - 1 : 0) + unit_rows * (l - n1);
        return tail_rows + unit_rows * (l - n1 - n2);
    }
};

// first[ib] (preset to ib) = the first 64-column block of block row ib of the lower triangle that holds a nonzero
// entry.  Workgroup (ib, chunk): scans PROFILE_CHUNK column blocks from the left, stops at the first hit.
constexpr int PROFILE_CHUNK = 32;
__global__ void __launch_bounds__(256) k_block_profile(const double *__restrict__ A, int64_t ld, int *__restrict__ first) {
    __shared__ int s_first;
    const int ib = blockIdx.x, j_begin = blockIdx.y * PROFILE_CHUNK;
    const int j_end = min(ib, j_begin + PROFILE_CHUNK);
    const int tid = threadIdx.x;
    if (j_begin >= j_end) return;
    if (tid == 0) s_first = first[ib];
    __syncthreads();
    if (j_begin >= s_first) return;       // a chunk further left already found one (the others settle it by atomicMin)
    const double *row = A + ((int64_t)NB * ib + (tid >> 2)) * ld + 16 * (tid & 3);
    for (int jb = j_begin; jb < j_end; ++jb) {
        const double2 *p = (const double2 *)(row + (int64_t)NB * jb);
        int nz = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double2 v = p[u];
            nz |= (v.x != 0.0) | (v.y != 0.0);
        }
        if (__syncthreads_or(nz)) {
            if (tid == 0) atomicMin(first + ib, jb);
            return;
        }
    }
}

// 1 / x for x far from the ends of the exponent range: hardware estimate (relative error 5e-8, measured) and one
// third-order correction step, three dependent instructions behind the estimate (relative error 1.1e-16)
__device__ __forceinline__ double rcp_cubic(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r, 1.0);
    return fma(r, fma(e, e, e), r);
}

// ---------------------------------------------------------------------------------------------------------
// Panel factorisation.
//
// Everything in a panel acts on rows independently except the factorisation of its diagonal block: with
// L_d = chol(A_dd) (pw x pw, pw <= 256) the rows below become A_below L_d^-T.  The diagonal block is factorised
// per 64 columns (k_potrf64, then k_diag_trsm / k_diag_syrk for what lies below and behind inside the diagonal block);
// k_panel_rows then gives every wave a strip of 16 rows below, held in registers across the whole blocked
// solve (no barrier between waves is needed for the data, only for the operand blocks staged in LDS).
// A launch per 64 columns for the rows below (a product with the inverse, then the update of the panel's
// remaining columns, one wave per 64 x 64 block on v_mfma_f64_16x16x4_f64) took 167 us per panel at n = 12 001,
// bound by the matrix-core time of the one wave that owns a block; the rows kernel takes 50 us.
//
// Strip products run on v_mfma_f64_4x4x4_4b_f64 (lane layout in the header of k_syrk_trailing).  For a strip
// S (16 rows) and a 64 x 64 block M the instruction sequence computes (M S^T), whose result registers hold
// exactly the layout the next product wants as its operand: lane l keeps, of row (l & 15) of the strip, the
// columns 4 e + (l >> 4), e = 0..15 - as loaded from memory, as operand, and as result.  Nothing is ever
// transposed or exchanged between lanes.
// ---------------------------------------------------------------------------------------------------------
constexpr int LP = NB + 4;  // row pitch of an operand block in LDS: the 4 rows x 4 columns an operand read touches
                            // fall on 16 different bank pairs

// acc (+)= S M^T for the strip operand b (b[e]: S[row lane & 15][column 4 e + (lane >> 4)]) and M in LDS at pitch LP
__device__ __forceinline__ void strip_mma(const double *__restrict__ M, const double (&b)[16], double (&acc)[16], int lane) {
    const double *m = M + (lane & 3) * LP + (lane >> 4);
    // The operands of column group ks + 1 are read while the matrix cores work on group ks (sixteen instructions
    // = 256 cycles for the reads to return): left to the compiler the reads sit a few instructions ahead of their
    // use, and with one wave per SIMD nothing else hides their latency.
    double a0[16], a1[16];
#pragma unroll
    for (int I = 0; I < 16; ++I) a0[I] = m[4 * I * LP];
#pragma unroll
    for (int ks = 0; ks < 16; ks += 2) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int I = 0; I < 16; ++I) a1[I] = m[4 * I * LP + 4 * (ks + 1)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int I = 0; I < 16; ++I) acc[I] = __builtin_amdgcn_mfma_f64_4x4x4f64(a0[I], b[ks], acc[I], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 2 < 16) {
#pragma unroll
            for (int I = 0; I < 16; ++I) a0[I] = m[4 * I * LP + 4 * (ks + 2)];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int I = 0; I < 16; ++I) acc[I] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1[I], b[ks + 1], acc[I], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void strip_load(const double *__restrict__ src, int64_t pitch, double (&v)[16], int lane) {
    const double *p = src + (int64_t)(lane & 15) * pitch + (lane >> 4);
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = p[4 * e];
}

__device__ __forceinline__ void strip_store(double *__restrict__ dst, int64_t pitch, const double (&v)[16], int lane) {
    double *p = dst + (int64_t)(lane & 15) * pitch + (lane >> 4);
#pragma unroll
    for (int e = 0; e < 16; ++e) p[4 * e] = v[e];
}

// The same through a 16 x 64 patch of LDS owned by the wave (pitch TP): in memory a wave instruction then covers 512
// consecutive bytes of one row instead of 32 bytes of each of 16 rows a whole matrix row apart (rows kernel with
// this and the row-wise staging of the operand blocks: 62 -> 47 us).  Only the wave itself touches its patch: no
// barrier.
constexpr int TP = NB + 2;  // 2 mod 32: the operand-layout reads of rows 0..15 x columns q, q + 1 hit 32 different bank pairs
__device__ __forceinline__ void strip_request(const double *__restrict__ src, int64_t pitch, double (&t)[16], int lane) {
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = src[(int64_t)r * pitch + lane];
}

// barrier of the workgroup that orders LDS accesses only: __syncthreads() also waits until every global store of
// the wave has completed (a round trip to memory per step of the rows kernel)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void strip_from_rows(const double (&t)[16], double *__restrict__ patch, double (&v)[16], int lane) {
#pragma unroll
    for (int r = 0; r < 16; ++r) patch[r * TP + lane] = t[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const double *p = patch + (lane & 15) * TP + (lane >> 4);
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = p[4 * e];
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void strip_store_lds(double *__restrict__ dst, int64_t pitch, double *__restrict__ patch,
                                                const double (&v)[16], int lane) {
    double *p = patch + (lane & 15) * TP + (lane >> 4);
#pragma unroll
    for (int e = 0; e < 16; ++e) p[4 * e] = v[e];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = patch[r * TP + lane];
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[(int64_t)r * pitch + lane] = t[r];
    __builtin_amdgcn_wave_barrier();
}

// 64 x 64 block, 256 threads: thread -> columns 2 (tid % 32), +1 of rows tid / 32 + 8 u (a wave instruction reads two
// whole 512-byte rows)
__device__ __forceinline__ void block_fetch(const double *__restrict__ g, int64_t pitch, double2 (&st)[8], int tid) {
    const double *p = g + (int64_t)(tid >> 5) * pitch + 2 * (tid & 31);
#pragma unroll
    for (int u = 0; u < 8; ++u) st[u] = *(const double2 *)(p + 8 * u * pitch);
}

__device__ __forceinline__ void block_put(double *__restrict__ d, const double2 (&st)[8], int tid) {
    double *p = d + (tid >> 5) * LP + 2 * (tid & 31);
#pragma unroll
    for (int u = 0; u < 8; ++u) *(double2 *)(p + 8 * u * LP) = st[u];
}

constexpr int DIAG_LDS_DOUBLES = NB * LP + 4 * 16 * TP;     // one operand block + a patch per wave

__device__ __forceinline__ void stage_block(const double *__restrict__ g, int64_t pitch, double *__restrict__ sm, int tid) {
    double2 st[8];
    block_fetch(g, pitch, st, tid);
    block_put(sm, st, tid);
}

// L_kk = chol(A_kk) in place and Linv = L_kk^-1 (64 x 64, row-major), one workgroup of four waves.
//
// What a column costs here is not arithmetic but the trip of the finished column from the wave that owns it to the
// other three (LDS write, barrier, LDS read: about 440 cycles, measured) and the chain of dependent instructions from
// the pivot to the next column.  So the kernel takes TWO columns per barrier, and keeps square roots and divisions off
// the chain: the columns stay unscaled (a_ik as the elimination leaves it; L_ik = a_ik / sqrt(d_k) is formed once, at
// the end), the update of a later entry is a_ij -= a_ik a_jk / d_k, the reciprocal of the second pivot of a pair comes
// from the determinant of the 2 x 2 pivot block (d_k' = det / d_k), so that both reciprocals start at once.
// The inverse rides along: the 64 rows of the identity go through the same column operations as the rows of A
// (row i of Y = I L^-T, i.e. column i of the inverse), two more multiply-adds per entry and step instead of a
// separate substitution phase (which took a third of the kernel before).
// Thread (row i = tid % 64, wave q = tid / 64) keeps, of row i of A and of row i of Y, the columns
// j(u) = 8 (u / 2) + 2 q + (u & 1), u = 0..15: pairs of neighbouring columns dealt out to the waves in turn.
// Entries above the diagonal are never used as inputs (they may hold anything, NaN included): what is computed from
// them stays above the diagonal.
constexpr int POTRF_LDS_DOUBLES = 2 * NB * (NB + 1) + 2 * 2 * 2 * NB + 3 * NB;
//
// with_prev (a block-tridiagonal chain: block row k holds nothing left of block k - 1): the kernel first does what
// two launches of their own do otherwise, L_k,k-1 = A_k,k-1 L_k-1,k-1^-T (stored) and A_kk -= L_k,k-1 L_k,k-1^T (kept in
// LDS for the factorisation), a wave per strip of 16 rows on the matrix cores; Linv - 4096 is the inverse the launch
// for block k - 1 left.
__global__ void __launch_bounds__(256) k_potrf64(double *__restrict__ A, int64_t ld, int64_t k0, double *__restrict__ Linv,
                                                 int *__restrict__ info, int64_t chain_rows, int with_prev) {
    extern __shared__ double sm[];
    k0 += blockIdx.y * chain_rows;                      // workgroup y: the same block of chain y
    Linv += blockIdx.y * chain_rows * NB;
    double(*Ls)[NB + 1] = (double(*)[NB + 1])sm;                      // the block on its way in and out
    double(*Xs)[NB + 1] = (double(*)[NB + 1])(sm + NB * (NB + 1));    // the inverse on its way out
    double2(*rawA)[NB] = (double2(*)[NB])(sm + 2 * NB * (NB + 1));    // [step & 1][row]: the step's two columns of A
    double2(*rawY)[NB] = rawA + 2;                                    // ... and of Y
    double *piv = (double *)(rawY + 2), *invs = piv + NB, *dsq = invs + NB;
    const int tid = threadIdx.x, i = tid & 63, q = tid >> 6;
    if (with_prev) {
        const int lane = tid & 63, w = tid >> 6;
        double *patch = sm + NB * LP + w * 16 * TP;     // (operand block and patches lie inside the kernel's own LDS)
        double *prev = A + (k0 + 16 * w) * ld + k0 - NB, *mine = A + (k0 + 16 * w) * ld + k0;
        double t0[16], t1[16], b[16], acc[16], d[16];
        strip_request(prev, ld, t0, lane);
        strip_request(mine, ld, t1, lane);
        stage_block(Linv - NB * NB, NB, sm, tid);
        strip_from_rows(t0, patch, b, lane);
        strip_from_rows(t1, patch, d, lane);
        lds_barrier();
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0;
        strip_mma(sm, b, acc, lane);                    // this strip of L_k,k-1
        lds_barrier();                                  // every wave has read the inverse
#pragma unroll
        for (int e = 0; e < 16; ++e) sm[(16 * w + (lane & 15)) * LP + 4 * e + (lane >> 4)] = acc[e];
        strip_store_lds(prev, ld, patch, acc, lane);
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = -acc[e];
        lds_barrier();                                  // L_k,k-1 stands in LDS, whole
        strip_mma(sm, acc, d, lane);                    // A_kk - L_k,k-1 L_k,k-1^T, this strip
        __syncthreads();                                // ... and nobody reads it any more: the block takes its place
#pragma unroll
        for (int e = 0; e < 16; ++e) Ls[16 * w + (lane & 15)][4 * e + (lane >> 4)] = d[e];
    } else {
        double2 tmp[8];
        const int r0 = tid >> 5, c0 = 2 * (tid & 31);
#pragma unroll
        for (int m = 0; m < 8; ++m) tmp[m] = *(const double2 *)(A + (k0 + r0 + 8 * m) * ld + k0 + c0);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            Ls[r0 + 8 * m][c0] = tmp[m].x;
            Ls[r0 + 8 * m][c0 + 1] = tmp[m].y;
        }
    }
    __syncthreads();
    double reg[16], y[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int j = 8 * (u >> 1) + 2 * q + (u & 1);
        reg[u] = Ls[i][j];
        y[u] = (j == i) ? 1.0 : 0.0;
    }
    if (q == 0) {
        rawA[0][i] = make_double2(reg[0], reg[1]);
        rawY[0][i] = make_double2(y[0], y[1]);
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < NB / 2; ++s) {
        // columns k, k + 1, owned by wave qs in its registers u0, u0 + 1; the next pair belongs to wave qn
        const int k = 2 * s, qs = s & 3, u0 = 2 * (s >> 2), buf = s & 1;
        const int qn = (s + 1) & 3, un = 2 * ((s + 1) >> 2);
        // every read of the step goes out right behind the barrier (left to itself the compiler sinks them to their
        // uses, one LDS latency after the other)
        const double2 P = rawA[buf][k], Q = rawA[buf][k + 1];       // pivot block: P.x = d_k, Q = (a_k+1,k , a_k+1,k+1)
        const double2 Ri = rawA[buf][i], Yi = rawY[buf][i];
        double2 Rj[16];
#pragma unroll
        for (int u = u0; u < 16; ++u) Rj[u] = rawA[buf][8 * (u >> 1) + 2 * q + (u & 1)];
        __builtin_amdgcn_sched_barrier(0);
        const double det = fma(P.x, Q.y, -(Q.x * Q.x));
        const double r0 = rcp_cubic(P.x);           // 1 / d_k
        const double r1 = P.x * rcp_cubic(det);     // 1 / d_k+1
        const double t = Q.x * r0;
        const double ai1 = fma(-Ri.x, t, Ri.y), yi1 = fma(-Yi.x, t, Yi.y);     // column k + 1 after column k
        const double w0 = Ri.x * r0, w1 = ai1 * r1, v0 = Yi.x * r0, v1 = yi1 * r1;
        const bool behind = q > qs, own = q == qs;
#pragma unroll
        for (int u = u0; u < 16; ++u) {
            const double aj1 = fma(-Rj[u].x, t, Rj[u].y);
            const double rn = fma(-w1, aj1, fma(-w0, Rj[u].x, reg[u]));
            const double yn = fma(-v1, aj1, fma(-v0, Rj[u].x, y[u]));
            if (u >= u0 + 2) {
                reg[u] = rn;
                y[u] = yn;
            } else if (u == u0) {       // registers of the current pair: columns behind the step's in the waves after qs
                reg[u] = behind ? rn : reg[u];
                y[u] = behind ? yn : y[u];
            } else {                    // the finished column k + 1 stays in its owner's registers, unscaled
                reg[u] = behind ? rn : (own ? ai1 : reg[u]);
                y[u] = behind ? yn : (own ? yi1 : y[u]);
            }
            if (s + 1 < NB / 2 && u == un + 1 && q == qn) {     // the next pair leaves as soon as it is complete
                rawA[buf ^ 1][i] = make_double2(reg[un], reg[un + 1]);
                rawY[buf ^ 1][i] = make_double2(y[un], y[un + 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }
    // pivots = the unscaled diagonal entries; their square roots and reciprocal square roots, once
#pragma unroll
    for (int u = 0; u < 16; ++u)
        if (8 * (u >> 1) + 2 * q + (u & 1) == i) piv[i] = reg[u];
    __syncthreads();
    if (tid < NB) {
        const double p = piv[tid];
        const double inv = fast_rsqrt(p);
        double d = p * inv;
        d = fma(0.5 * inv, fma(-d, d, p), d);
        invs[tid] = inv;
        dsq[tid] = d;
        const unsigned long long bad = __ballot(!(p > 0.0));
        if (bad && tid == 0) atomicCAS(info, 0, (int)(k0 + __ffsll((long long)bad)));
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int j = 8 * (u >> 1) + 2 * q + (u & 1);
        const double sc = invs[j];
        Ls[i][j] = (j < i) ? reg[u] * sc : (j == i ? dsq[i] : 0.0);
        Xs[j][i] = y[u] * sc;
    }
    __syncthreads();
    for (int idx = tid; idx < NB * NB; idx += 256) {
        const int rr = idx >> 6, cc = idx & 63;
        if (cc <= rr) A[(k0 + rr) * ld + k0 + cc] = Ls[rr][cc];
        Linv[idx] = Xs[rr][cc];
    }
}

// Inside the diagonal block of a panel (64 nbk columns from p0), after potrf64 of its block kb: the blocks below it
// times the inverse, A_jk <- A_jk L_kk^-T.  Workgroup = block row kb + 1 + blockIdx.x, wave = 16-row strip.
__global__ void __launch_bounds__(256) k_diag_trsm(double *__restrict__ A, int64_t ld, int64_t p0, int kb,
                                                   const double *__restrict__ Li, int64_t chain_rows) {
    extern __shared__ double sm[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t k0 = p0 + NB * kb + blockIdx.y * chain_rows;
    Li += blockIdx.y * chain_rows * NB;
    double *strip = A + (k0 + NB * (1 + blockIdx.x) + 16 * w) * ld + k0;
    double *patch = sm + NB * LP + w * 16 * TP;
    double t[16], b[16], acc[16];
    strip_request(strip, ld, t, lane);
    stage_block(Li, NB, sm, tid);
    strip_from_rows(t, patch, b, lane);
    lds_barrier();
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0;
    strip_mma(sm, b, acc, lane);
    strip_store_lds(strip, ld, patch, acc, lane);
}

// ... then the update of the blocks behind, A_ij -= A_ik A_jk^T for the pairs j <= i of block rows behind kb.
// Workgroup = pair (ti, tj) counted from block row kb + 1, wave = 16-row strip of block (i, j).
// (Both steps and potrf64 as phases of one workgroup - one launch per panel for its whole diagonal block - were
// measured: potrf64 runs at 37 us there instead of 26 us, eight waves walking through its 64 barriers, and the
// strips of a phase queue up on the one CU: 231 us per panel against 104 us of potrf64 plus 14 us per step here.)
__global__ void __launch_bounds__(256) k_diag_syrk(double *__restrict__ A, int64_t ld, int64_t p0, int kb, int64_t chain_rows) {
    extern __shared__ double sm[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t k0 = p0 + NB * kb + blockIdx.y * chain_rows;
    const int pair = blockIdx.x;
    const int ti = pair < 1 ? 0 : (pair < 3 ? 1 : 2), tj = pair - ti * (ti + 1) / 2;
    double *strip = A + (k0 + NB * (1 + ti) + 16 * w) * ld + k0 + NB * (1 + tj);
    const double *mine = A + (k0 + NB * (1 + ti) + 16 * w) * ld + k0;       // this strip's rows of A_ik
    double *patch = sm + NB * LP + w * 16 * TP;
    double t0[16], t1[16], b[16], acc[16];
    strip_request(mine, ld, t0, lane);
    strip_request(strip, ld, t1, lane);
    stage_block(A + (k0 + NB * (1 + tj)) * ld + k0, ld, sm, tid);
    strip_from_rows(t0, patch, b, lane);
    strip_from_rows(t1, patch, acc, lane);
#pragma unroll
    for (int e = 0; e < 16; ++e) b[e] = -b[e];
    lds_barrier();
    strip_mma(sm, b, acc, lane);
    strip_store_lds(strip, ld, patch, acc, lane);
}

// Rows below the diagonal block of a panel: A[r, p0 : p0 + 64 NBK) <- the same times L_d^-T, by blocks: for every
// column block kb, T_kb = S_kb L_kk^-T, then S_jb -= T_kb L_jk^T for the blocks jb behind it.  Workgroup = 64 rows,
// wave = 16 rows; the operand blocks (inverses from Linv, L_jk from the factorised diagonal block) pass through two
// LDS buffers, the next one loaded while the matrix cores work on the current one.
// l_begin: the launch's first 64-row unit of `sel` (a launch can take a part of the rows).
template <int NBK>
__global__ void __launch_bounds__(256) k_panel_rows(double *__restrict__ A, int64_t ld, int64_t p0,
                                                    const double *__restrict__ Linv, RowSel sel, int l_begin) {
    extern __shared__ double sm[];
    constexpr int NOPS = NBK * (NBK + 1) / 2;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int chain = blockIdx.y;
    double *rows = A + (p0 + NB * NBK + sel(blockIdx.x + l_begin, NB, chain) + 16 * w) * ld + p0 + chain * sel.chain_rows;
    p0 += chain * sel.chain_rows;           // from here on: the panel's diagonal block, rows and columns of the chain
    Linv += chain * sel.chain_rows * NB;
    double *patch = sm + 2 * NB * LP + w * 16 * TP;
    double S[NBK][16];
#pragma unroll
    for (int kb = 0; kb < NBK; kb += 2) {       // two blocks' rows in flight at a time
        double t0[16], t1[16];
        strip_request(rows + NB * kb, ld, t0, lane);
        if (kb + 1 < NBK) strip_request(rows + NB * (kb + 1), ld, t1, lane);
        strip_from_rows(t0, patch, S[kb], lane);
        if (kb + 1 < NBK) strip_from_rows(t1, patch, S[kb + 1], lane);
    }
    double2 st[8];
    // jb == kb: the inverse of diagonal block kb, else L_jk of the factorised diagonal block
#define BA_FETCH(kb_, jb_)                                                                                         \
    block_fetch((jb_) == (kb_) ? Linv + (int64_t)(kb_) * NB * NB : A + (p0 + NB * (jb_)) * ld + p0 + NB * (kb_),  \
                (jb_) == (kb_) ? (int64_t)NB : ld, st, tid)
#define BA_PUT(n_) block_put(sm + ((n_) & 1) * NB * LP, st, tid)
    BA_FETCH(0, 0);
    BA_PUT(0);
    lds_barrier();
    int n = 0;
#pragma unroll
    for (int kb = 0; kb < NBK; ++kb) {
#pragma unroll
        for (int jb = kb; jb < NBK; ++jb) {      // jb == kb: the product with the inverse; behind it the updates
            if (n + 1 < NOPS) BA_FETCH(jb + 1 < NBK ? kb : kb + 1, jb + 1 < NBK ? jb + 1 : kb + 1);
            const double *M = sm + (n & 1) * NB * LP;
            if (jb == kb) {
                double acc[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.0;
                strip_mma(M, S[kb], acc, lane);
                strip_store_lds(rows + NB * kb, ld, patch, acc, lane);
#pragma unroll
                for (int e = 0; e < 16; ++e) S[kb][e] = -acc[e];    // from here on the operand of the updates
            } else {
                strip_mma(M, S[kb], S[jb], lane);
            }
            if (n + 1 < NOPS) BA_PUT(n + 1);    // the other buffer: its readers passed the barrier of the step before
            lds_barrier();
            ++n;
        }
    }
#undef BA_FETCH
#undef BA_PUT
}

// Update behind a panel: C[i0:+128, j0:+128] -= P_i P_j^T with P = A[:, k0 : k0 + kw], for the 128-tiles of
// the region that starts at row / column c0 and is nct tile columns wide (lower triangle of its first
// nct tile rows, whole rows below).  Four waves in a 2 x 2 arrangement of 64 x 64 sub-tiles.
//
// Matrix-core instruction: v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 products per
// instruction), measured at 70 TFLOP/s on MI355X against 36 TFLOP/s for v_mfma_f64_16x16x4_f64.
// Lane layout (found by probing the instruction): block b = (lane >> 2) & 3; lane supplies
// A_b[lane & 3][lane >> 4] and B_b[lane >> 4][lane & 3] and holds D_b[lane >> 4][lane & 3].
// A wave's 64 x 64 sub-tile is 16 x 16 blocks of 4 x 4.  Per 4 columns of the panel it reads 16
// A registers (register I: rows 4 I + (lane & 3) of the row operand, the same for every b) and 4
// B registers (register c: rows 16 c + (lane & 15) of the column operand, i.e. the four block
// columns 4 c + b), and issues 64 instructions, instruction (I, c) updating the blocks (I, 4 c + b),
// b = 0..3; its 64 results are 4 rows of 16 consecutive columns of C (128-byte segments).
//
// The panel rows of both operands pass through LDS in chunks of KC columns, stored column-major
// ([column][row], row pitch 132 = 4 mod 32 doubles: the 16 addresses of an A register read fall on 16
// different bank pairs, and the transposing stores of the loader - 8 columns x 8 rows per wave
// instruction - on 32 bank pairs twice); global loads of the next chunk are in flight while the
// matrix cores work on the current one.  Loads are whole 128-byte rows per 8 lanes: a mapping with one
// row per lane makes every load instruction touch 64 cache lines and leaves the kernel bound by the
// address pipeline of the vector memory unit.
__global__ void __launch_bounds__(256, 2) k_syrk_trailing(double *__restrict__ A, int64_t ld, int64_t k0, int kw, int64_t c0,
                                                          int64_t nct, RowSel sel, int64_t t_begin, int64_t bj_lo, int64_t bj_hi) {
    extern __shared__ double smem[];
    double(*As)[LDT] = (double(*)[LDT])smem;
    double(*Bs)[LDT] = (double(*)[LDT])(smem + KC * LDT);
    // tile index -> (bi, bj), bj <= bi and bj < nct: the lower triangle of the first nct tile columns, then the rows
    // below it, nct tiles each.  (Grouping the tiles of an 8 x 8 super-tile on one XCD, to share panel rows through
    // its L2, was measured: no gain - the panel is served from the infinity cache either way.)
    const int64_t t = blockIdx.x + t_begin, tri = nct * (nct + 1) / 2;     // (t_begin: a launch can take a part of the tiles)
    int64_t bi, bj;
    if (t < tri) {
        bi = (int64_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while (bi * (bi + 1) / 2 > t) --bi;
        while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
        bj = t - bi * (bi + 1) / 2;
    } else {
        bi = nct + (t - tri) / nct;
        bj = (t - tri) % nct;
    }
    if (bj < bj_lo || bj >= bj_hi) return;      // (a launch can take the tiles of some tile columns only)
    const int chain = blockIdx.z;
    const int64_t i0 = c0 + sel(bi, TS, chain), j0 = c0 + sel(bj, TS, chain);     // tiles counted without the rows left out
    // With few tiles (a banded matrix) the panel's columns are split over gridDim.y workgroups per tile, kw columns
    // each, which add their parts to C with atomics: one workgroup walking through 1024 columns alone is 200 us of
    // latency with the rest of the chip idle.  Tiles in the columns of the separators and of the border receive the
    // updates of several chains at once: atomics as well.
    k0 += (int64_t)blockIdx.y * kw + chain * sel.chain_rows;
    const bool shared_tile = gridDim.y > 1 || (gridDim.z > 1 && bj >= sel.n1);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 1, wc = w & 1;
    const int r16 = lane & 15, k4 = lane >> 4, j4 = lane & 3;
    const bool active = !(bi == bj && wc > wr);     // the sub-tile above the diagonal is never used

    // loader mapping: 8 consecutive lanes read the 128 bytes (KC columns) of one row, a wave 8 whole rows per
    // instruction; thread -> columns 2 (tid % 8), +1 of rows tid / 8 + 32 q, q = 0..3, of both operands
    const int lcol = 2 * (tid & 7), lrow = tid >> 3;
    const double2 *srcA[4], *srcB[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int64_t ra = i0 + lrow + 32 * q, rb = j0 + lrow + 32 * q;
        if (ra >= ld) ra = ld - 1;   // rows past the matrix: a valid address, results never stored
        if (rb >= ld) rb = ld - 1;
        srcA[q] = (const double2 *)(A + ra * ld + k0 + lcol);
        srcB[q] = (const double2 *)(A + rb * ld + k0 + lcol);
    }
    double2 pa[4], pb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        pa[q] = srcA[q][0];
        pb[q] = srcB[q][0];
    }
    double acc[16][4];
#pragma unroll
    for (int I = 0; I < 16; ++I)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[I][c] = 0.0;
    for (int kc = 0; kc < kw; kc += KC) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            As[lcol][lrow + 32 * q] = pa[q].x;
            As[lcol + 1][lrow + 32 * q] = pa[q].y;
            Bs[lcol][lrow + 32 * q] = pb[q].x;
            Bs[lcol + 1][lrow + 32 * q] = pb[q].y;
        }
        __syncthreads();
        if (kc + KC < kw) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                pa[q] = srcA[q][(kc + KC) / 2];
                pb[q] = srcB[q][(kc + KC) / 2];
            }
        }
        if (active) {
            // Software pipeline over the KC / 4 slabs of the chunk: the fragments of slab s + 1 are read from LDS
            // while the matrix cores work on slab s, in two halves so that no extra A registers are needed (the
            // A registers of block rows 0..7 are refilled during the instructions of block rows 8..15 and
            // vice versa); only the first slab of a chunk waits for its reads.
            double af[16], bf[2][4];
            auto load_b = [&](int kk, int buf) {
#pragma unroll
                for (int c = 0; c < 4; ++c) bf[buf][c] = Bs[kk + k4][64 * wc + 16 * c + r16];
            };
            auto load_a = [&](int kk, int half) {
#pragma unroll
                for (int I = 8 * half; I < 8 * half + 8; ++I) af[I] = As[kk + k4][64 * wr + 4 * I + j4];
            };
            auto mma = [&](int half, int buf) {
#pragma unroll
                for (int I = 8 * half; I < 8 * half + 8; ++I)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        acc[I][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[I], bf[buf][c], acc[I][c], 0, 0, 0);
            };
            load_b(0, 0);
            load_a(0, 0);
            load_a(0, 1);
#pragma unroll
            for (int sl = 0; sl < KC / 4; ++sl) {
                const int buf = sl & 1;
                const bool more = sl + 1 < KC / 4;
                __builtin_amdgcn_sched_barrier(0);
                mma(0, buf);
                __builtin_amdgcn_sched_barrier(0);
                if (more) {
                    load_b(4 * sl + 4, buf ^ 1);
                    load_a(4 * sl + 4, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                mma(1, buf);
                __builtin_amdgcn_sched_barrier(0);
                if (more) load_a(4 * sl + 4, 1);
            }
        }
        __syncthreads();
    }
    if (!active) return;
    // block (I, 4 c + b), b = (lane >> 2) & 3, entry [lane >> 4][lane & 3]: row 4 I + (lane >> 4), column 16 c + (lane & 15)
#pragma unroll
    for (int I = 0; I < 16; ++I) {
        const int64_t row = i0 + 64 * wr + 4 * I + k4;
        if (row >= ld) continue;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int64_t cc = j0 + 64 * wc + 16 * c + r16;
            if (cc >= ld) continue;
            if (shared_tile) __hip_atomic_fetch_add(A + row * ld + cc, -acc[I][c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else A[row * ld + cc] -= acc[I][c];
        }
    }
}

// The forward solve L y = b rides along with the factorisation: with b^T stored as the last (padding) row of the
// matrix, the row that the factorisation leaves there is y^T (the row's entries solve  row . L^T = b^T), computed by
// the panel and update kernels like any other row.  Only the row's own pivot is meaningless: nothing depends on
// it, and a diagonal entry of 1e200 keeps it positive and finite whatever |y|^2 is taken away from it (and its product
// with the pivot before it, which the diagonal-block kernel forms, finite as well).  That
// replaces ld / 64 dependent launches of k_solve_step.
// (newpos: the position of every 64-block in the order the factorisation works in, or null)
__global__ void k_rhs_into_row(double *__restrict__ A, int64_t ld, int64_t n, const double *__restrict__ rhs,
                               const int *__restrict__ newpos) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ld) return;
    const int64_t cn = newpos ? (int64_t)newpos[c / NB] * NB + c % NB : c;
    A[(ld - 1) * ld + cn] = c < n ? rhs[c] : (c == ld - 1 ? 1e200 : 0.0);
}

__global__ void k_row_to_rhs(const double *__restrict__ A, int64_t ld, int64_t n, double *__restrict__ rhs) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < ld) rhs[c] = c < n ? A[(ld - 1) * ld + c] : 0.0;
}

// solution in the order the factorisation worked in -> the caller's order
__global__ void k_solution_back(const double *__restrict__ x, int64_t ld, const int *__restrict__ newpos, double *__restrict__ out) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < ld) out[c] = x[(int64_t)newpos[c / NB] * NB + c % NB];
}

// Symmetric permutation of the 64-blocks of the lower triangle, in place, by way of a copy of the occupied blocks:
// workgroup b takes block (from[b].x, from[b].y) to T (mode 0), clears it (mode 1), or stores T[b] as block
// (to[b].x, to[b].y) - transposed when to[b].z says that the block crossed the diagonal (mode 2).
__global__ void __launch_bounds__(256) k_move_blocks(double *__restrict__ A, int64_t ld, const int4 *__restrict__ moves,
                                                     double *__restrict__ T, int mode) {
    __shared__ double tile[NB][NB + 1];
    const int4 m = moves[blockIdx.x];
    const int tid = threadIdx.x;
    double *t = T + (int64_t)blockIdx.x * NB * NB;
    if (mode == 0) {
        const double *src = A + (int64_t)m.x * NB * ld + (int64_t)m.y * NB;
        for (int idx = tid; idx < NB * NB; idx += 256) t[idx] = src[(int64_t)(idx >> 6) * ld + (idx & 63)];
    } else if (mode == 1) {
        double *dst = A + (int64_t)m.x * NB * ld + (int64_t)m.y * NB;
        for (int idx = tid; idx < NB * NB; idx += 256) dst[(int64_t)(idx >> 6) * ld + (idx & 63)] = 0.0;
    } else {
        double *dst = A + (int64_t)m.z * NB * ld + (int64_t)m.w * NB;
        const bool flip = m.z < 0;
        if (!flip) {
            for (int idx = tid; idx < NB * NB; idx += 256) dst[(int64_t)(idx >> 6) * ld + (idx & 63)] = t[idx];
        } else {
            dst = A + (int64_t)(-m.z - 1) * NB * ld + (int64_t)m.w * NB;
            for (int idx = tid; idx < NB * NB; idx += 256) tile[idx >> 6][idx & 63] = t[idx];
            __syncthreads();
            for (int idx = tid; idx < NB * NB; idx += 256) dst[(int64_t)(idx >> 6) * ld + (idx & 63)] = tile[idx & 63][idx >> 6];
        }
    }
}

// Clears runs of 64-blocks: workgroup b the blocks (runs[b].x, runs[b].y .. runs[b].z) of block row runs[b].x.
__global__ void __launch_bounds__(256) k_clear_runs(double *__restrict__ A, int64_t ld, const int4 *__restrict__ runs) {
    const int4 r = runs[blockIdx.x];
    const int width = (r.z - r.y + 1) * NB / 2;     // double2 per row
    double2 *row0 = (double2 *)(A + (int64_t)r.x * NB * ld + (int64_t)r.y * NB);
    for (int rr = 0; rr < NB; ++rr) {
        double2 *row = row0 + (int64_t)rr * (ld / 2);
        for (int c = threadIdx.x; c < width; c += 256) row[c] = make_double2(0.0, 0.0);
    }
}

// One block step of a triangular solve with the stored inverses of the diagonal blocks.
// forward  (L y = b):   y_k = Linv_k b_k, then b_i -= L_ik y_k for the rows i behind block k
// backward (L^T x = y): x_k = Linv_k^T b_k, then b_i -= L_ki^T x_k for the rows i before block k
// Every workgroup forms the 64-vector itself (b_k is not overwritten by this launch: the result
// goes to `out`), then updates its own 256 rows.  Workgroup 0 stores the result.
// transpose: rows before i_begin are left out (the block row k has no entries there).
__global__ void __launch_bounds__(256) k_solve_step(const double *__restrict__ A, int64_t ld, int64_t k0,
                                                    const double *__restrict__ Linv, double *__restrict__ b,
                                                    double *__restrict__ out, int transpose, int64_t i_begin,
                                                    int64_t chain_rows) {
    __shared__ double bk[NB], yk[NB], part[4][NB];
    const int tid = threadIdx.x;
    k0 += blockIdx.y * chain_rows;
    i_begin += blockIdx.y * chain_rows;
    Linv += blockIdx.y * chain_rows * NB;
    if (tid < NB) bk[tid] = b[k0 + tid];
    __syncthreads();
    {
        // y[r] = sum_c M[r][c] bk[c], M = Linv (forward) or Linv^T (backward); 4 threads per entry
        const int r = tid & 63, q = tid >> 6;
        double s = 0.0;
        for (int c = 16 * q; c < 16 * q + 16; ++c) s += (transpose ? Linv[c * NB + r] : Linv[r * NB + c]) * bk[c];
        part[q][r] = s;
    }
    __syncthreads();
    if (tid < NB) {
        const double y = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
        yk[tid] = y;
        if (blockIdx.x == 0) out[k0 + tid] = y;
    }
    __syncthreads();
    const int64_t row = (int64_t)blockIdx.x * 256 + tid;
    if (!transpose) {
        const int64_t i = k0 + NB + row;
        if (i >= ld) return;
        const double *Ar = A + i * ld + k0;
        double s = 0.0;
#pragma unroll 8
        for (int c = 0; c < NB; ++c) s += Ar[c] * yk[c];
        b[i] -= s;
    } else {
        const int64_t i = i_begin + row;
        if (i >= k0) return;
        double s = 0.0;
#pragma unroll 8
        for (int c = 0; c < NB; ++c) s += A[(k0 + c) * ld + i] * yk[c];
        b[i] -= s;
    }
}

// ---- backward solve L^T x = y of the reordered factorisation in three launches instead of one per block ----------
// A block step is a chain of two trips to memory and next to no arithmetic, so a launch per step costs its launch
// latency 32 times at n = 12 001.  What is behind the chains (separators, rest of the band, border: `tail`) is a small
// dense triangle that ONE workgroup walks through with the right-hand side in LDS (k_back_tail); its effect on the
// chains' unknowns is one product over the whole chip (k_back_tail_to_chains); then every chain is walked by a
// workgroup of its own (k_back_chains).
constexpr int BACK_TAIL_MAX_BLOCKS = 96;      // tail blocks one workgroup takes (right-hand side: 48 KB of LDS)
constexpr int BACK_CHAIN_MAX_BLOCKS = 96;     // blocks of a chain (48 KB of LDS)

// x_k = Linv_k^T b_k for k = nb - 1 .. B0, b_i -= L_ki^T x_k for the tail's rows i before k; x -> out.
// tfirst[k - B0]: the first block (counted from B0) of block row k of L that holds anything.
// One workgroup of 1024 threads; dynamic LDS: (nb - B0) * 64 doubles.
__global__ void __launch_bounds__(1024) k_back_tail(const double *__restrict__ A, int64_t ld, const double *__restrict__ Linv,
                                                     const double *__restrict__ b, double *__restrict__ out, int64_t B0,
                                                     int64_t nb, const int *__restrict__ tfirst) {
    extern __shared__ double bt[];
    __shared__ double xk[NB], part[16][NB + 1];
    const int tid = threadIdx.x;
    const int64_t base = B0 * NB;
    const int nD = (int)(nb - B0);
    for (int idx = tid; idx < nD * NB; idx += 1024) bt[idx] = b[base + idx];
    const int r = tid & 63, q = tid >> 6;       // the product with the inverse: entry r, columns 4 q .. 4 q + 3
    double lv[4];
    {
        const double *Lk = Linv + (nb - 1) * NB * NB;
#pragma unroll
        for (int c = 0; c < 4; ++c) lv[c] = Lk[(4 * q + c) * NB + r];
    }
    __syncthreads();
    for (int64_t k = nb - 1; k >= B0; --k) {
        const int lk = (int)(k - B0) * NB;
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < 4; ++c) s += lv[c] * bt[lk + 4 * q + c];
        part[q][r] = s;
        if (k > B0) {       // the next block's inverse is on its way while this step's update runs
            const double *Lk = Linv + (k - 1) * NB * NB;
#pragma unroll
            for (int c = 0; c < 4; ++c) lv[c] = Lk[(4 * q + c) * NB + r];
        }
        __syncthreads();
        if (tid < NB) {
            double x = 0.0;
#pragma unroll
            for (int g = 0; g < 16; ++g) x += part[g][tid];
            xk[tid] = x;
            out[k * NB + tid] = x;
        }
        __syncthreads();
        // rows k0 .. k0 + 63 of L, columns base .. k0: thread (column tid & 255 of a pass of 256, rows 16 (tid >> 8) ..)
        const int cg = tid & 255, rg = tid >> 8;
        const double *Ar = A + (k * NB + 16 * rg) * ld + base;
        for (int col = tfirst[k - B0] * NB + cg; col < lk; col += 256) {       // (L has nothing left of the row's first block)
            double u = 0.0;
#pragma unroll
            for (int c = 0; c < 16; ++c) u += Ar[(int64_t)c * ld + col] * xk[16 * rg + c];
            atomicAdd(&bt[col], -u);
        }
        __syncthreads();
    }
}

// b_i -= sum over the rows r behind a level's chains of L_ri x_r, for the unknowns i of the chains (columns from c_base
// on; rows from block B0 on, separators up to block T0).  Workgroup (slab of 128 columns, block row): a separator's rows
// hold entries in the columns of the chains on either side of it only.  rowpos (or null): where in x the solution of
// block row B0 + y stands (a further level behind these chains has reordered those rows).
__global__ void __launch_bounds__(256) k_back_tail_to_chains(const double *__restrict__ A, int64_t ld, const double *__restrict__ x,
                                                             double *__restrict__ b, int64_t c_base, int64_t B0, int64_t T0,
                                                             int64_t bw, int64_t chain_rows, const int *__restrict__ rowpos) {
    __shared__ double xs[NB];
    const int tid = threadIdx.x;
    const int64_t k = B0 + blockIdx.y, col0 = (int64_t)blockIdx.x * 128;       // col0: counted from c_base
    if (k < T0) {
        const int64_t sep = (k - B0) / bw, c_first = col0 / chain_rows, c_last = (col0 + 127) / chain_rows;
        if (c_last < sep || c_first > sep + 1) return;
    }
    const int64_t xrow = rowpos ? rowpos[blockIdx.y] : k;
    if (tid < NB) xs[tid] = x[xrow * NB + tid];
    __syncthreads();
    const int col = tid & 127, half = tid >> 7;
    if (c_base + col0 + col >= B0 * NB) return;
    const double *Ar = A + (k * NB + 32 * half) * ld + c_base + col0 + col;
    double u = 0.0;
#pragma unroll 16
    for (int c = 0; c < 32; ++c) u += Ar[(int64_t)c * ld] * xs[32 * half + c];
    atomicAdd(b + c_base + col0 + col, -u);
}

// Chain blockIdx.x of the chains that start at block `off`: x_l = Linv_l^T b_l for its blocks l = s_blocks - 1 .. 0,
// b_i -= L_li^T x_l for the chain's rows from block first[l] on.  256 threads; dynamic LDS: s_blocks * 64 doubles.
__global__ void __launch_bounds__(256) k_back_chains(const double *__restrict__ A, int64_t ld, const double *__restrict__ Linv,
                                                      const double *__restrict__ b, double *__restrict__ out, int64_t off,
                                                      int s_blocks, const int *__restrict__ first) {
    extern __shared__ double bt[];
    __shared__ double xk[NB], part[4][NB + 1];
    const int tid = threadIdx.x;
    const int64_t base = (off + (int64_t)blockIdx.x * s_blocks) * NB;
    A += base * ld + base;
    Linv += base * NB;
    for (int idx = tid; idx < s_blocks * NB; idx += 256) bt[idx] = b[base + idx];
    const int r = tid & 63, q = tid >> 6;
    // Nothing a step reads from memory depends on the steps before it: the inverse of block l - 1 and the block
    // (l, l - 1) of L - the one the next step's right-hand side waits for - are requested a step ahead, so that a step
    // is LDS traffic and arithmetic only.  Blocks further left (a band wider than one block) are read when needed.
    double lv[16], nx[16];
    {
        const double *Lk = Linv + (int64_t)(s_blocks - 1) * NB * NB;
#pragma unroll
        for (int c = 0; c < 16; ++c) lv[c] = Lk[(16 * q + c) * NB + r];
        if (s_blocks > 1) {
            const double *Ar = A + ((int64_t)(s_blocks - 1) * NB + 16 * q) * ld + (int64_t)(s_blocks - 2) * NB + r;
#pragma unroll
            for (int c = 0; c < 16; ++c) nx[c] = Ar[(int64_t)c * ld];
        }
    }
    __syncthreads();
    for (int l = s_blocks - 1; l >= 0; --l) {
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < 16; ++c) s += lv[c] * bt[l * NB + 16 * q + c];
        part[q][r] = s;
        double cur[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) cur[c] = nx[c];
        if (l > 0) {
            const double *Lk = Linv + (int64_t)(l - 1) * NB * NB;
#pragma unroll
            for (int c = 0; c < 16; ++c) lv[c] = Lk[(16 * q + c) * NB + r];
            if (l > 1) {
                const double *An = A + ((int64_t)(l - 1) * NB + 16 * q) * ld + (int64_t)(l - 2) * NB + r;
#pragma unroll
                for (int c = 0; c < 16; ++c) nx[c] = An[(int64_t)c * ld];
            }
        }
        __syncthreads();
        if (tid < NB) {
            const double x = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
            xk[tid] = x;
            out[base + l * NB + tid] = x;
        }
        __syncthreads();
        if (l > 0 && first[l] < l) {
            double u = 0.0;
#pragma unroll
            for (int c = 0; c < 16; ++c) u += cur[c] * xk[16 * q + c];
            atomicAdd(&bt[(l - 1) * NB + r], -u);
        }
        const double *Ar = A + ((int64_t)l * NB + 16 * q) * ld;
        for (int col = first[l] * NB + r; col < (l - 1) * NB; col += NB) {
            double u = 0.0;
#pragma unroll
            for (int c = 0; c < 16; ++c) u += Ar[(int64_t)c * ld + col] * xk[16 * q + c];
            atomicAdd(&bt[col], -u);
        }
        __syncthreads();
    }
}

}  // namespace ba

using namespace ba;

// ---- the occupied blocks of S as one contiguous buffer (what several GPUs add up instead of the whole matrix) ----
// offsets[ib] = number of 64-blocks in the block rows before ib, block row ib holding the blocks first[ib] .. ib
__global__ void k_envelope_offsets(const int *__restrict__ first, int64_t nb, int64_t *__restrict__ offsets) {
    __shared__ int64_t wsum[16];
    __shared__ int64_t carry_s;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < nb; base += 1024) {
        const int64_t t = base + threadIdx.x;
        const int64_t v = t < nb ? t - first[t] + 1 : 0;
        int64_t inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int64_t u = __shfl_up(inc, off);
            if (lane >= off) inc += u;
        }
        if (lane == 63) wsum[wid] = inc;
        __syncthreads();
        int64_t before = carry_s;
        for (int w = 0; w < wid; ++w) before += wsum[w];
        if (t < nb) offsets[t] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) offsets[nb] = carry_s;
}

// workgroup b: block b of the envelope (block row found by bisection of the offsets) to or from packed[b][64][64]
__global__ void __launch_bounds__(256) k_envelope_copy(double *__restrict__ S, int64_t ld, const int *__restrict__ first,
                                                       const int64_t *__restrict__ offsets, int64_t nb,
                                                       double *__restrict__ packed, int to_packed) {
    const int64_t b = blockIdx.x;
    int64_t lo = 0, hi = nb - 1;
    while (lo < hi) {
        const int64_t mid = (lo + hi + 1) >> 1;
        if (offsets[mid] <= b) lo = mid;
        else hi = mid - 1;
    }
    const int64_t ib = lo, jb = first[ib] + (b - offsets[ib]);
    double *blk = S + ib * NB * ld + jb * NB, *pk = packed + b * NB * NB;
    for (int idx = threadIdx.x; idx < NB * NB; idx += 256) {
        double *e = blk + (int64_t)(idx >> 6) * ld + (idx & 63);
        if (to_packed) pk[idx] = *e;
        else *e = pk[idx];
    }
}

// (the two clearing steps of a call, as kernels like everything else it queues)
__global__ void k_zero_int(int *p) { *p = 0; }
// The queued form assumed the profile `assumed`; the matrix has `now`.  Entries left of the assumed envelope would be
// ignored: the verdict becomes PROFILE_MOVED before anything is factorised (the pivot check only replaces a zero).
constexpr int PROFILE_MOVED = -0x7fffffff - 1;
__global__ void k_profile_guard(const int *__restrict__ now, const int *__restrict__ assumed, int64_t nb, int *__restrict__ info) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nb && now[k] < assumed[k]) *info = PROFILE_MOVED;
}
__global__ void k_publish_info(const int *__restrict__ info, int *__restrict__ host_word) {
    __hip_atomic_store(host_word, *info, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_zero_doubles(double *__restrict__ p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0.0;
}

extern "C" {

int ba_schur_dim(int64_t K, int64_t *n, int64_t *ld) {
    BA_REQUIRE(K > 0 && n && ld, "ba_schur_dim: bad argument");
    *n = 6 * K + 1;
    *ld = ((*n + NB - 1) / NB) * NB;
    return BA_OK;
}

int ba_cholesky_solve(int64_t n, int64_t ld, double *S, double *rhs, int *h_info, void *stream) {
    return ba_cholesky_solve_tracked(n, ld, S, rhs, nullptr, 0, h_info, stream);
}

static int solve_impl(int64_t n, int64_t ld, double *S, double *rhs, const int *block_first, int clear_on_return,
                      int *h_info, void *stream, bool queued);

int ba_cholesky_solve_tracked(int64_t n, int64_t ld, double *S, double *rhs, const int *block_first, int clear_on_return,
                              int *h_info, void *stream) {
    return solve_impl(n, ld, S, rhs, block_first, clear_on_return, h_info, stream, false);
}

int ba_cholesky_solve_queued(int64_t n, int64_t ld, double *S, double *rhs, const int *block_first, int clear_on_return,
                             void *stream) {
    BA_REQUIRE(block_first, "ba_cholesky_solve_queued: block_first is required");
    return solve_impl(n, ld, S, rhs, block_first, clear_on_return, nullptr, stream, true);
}

int ba_schur_envelope(int64_t ld, const int *block_first, int64_t *offsets, int64_t *h_blocks, void *stream) {
    BA_REQUIRE(ld > 0 && ld % NB == 0 && block_first && offsets && h_blocks, "ba_schur_envelope: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int64_t nb = ld / NB;
    hipLaunchKernelGGL(k_envelope_offsets, dim3(1), dim3(1024), 0, st, block_first, nb, offsets);
    BA_HIP(hipGetLastError());
    BA_HIP(hipMemcpyAsync(h_blocks, offsets + nb, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    BA_HIP(hipStreamSynchronize(st));
    return BA_OK;
}

int ba_schur_envelope_copy(int64_t ld, double *S, const int *block_first, const int64_t *offsets, int64_t blocks,
                           double *packed, int to_packed, void *stream) {
    BA_REQUIRE(ld > 0 && ld % NB == 0 && S && block_first && offsets && packed && blocks >= 0,
               "ba_schur_envelope_copy: bad argument");
    if (blocks > 0)
        hipLaunchKernelGGL(k_envelope_copy, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, S, ld, block_first,
                           offsets, ld / NB, packed, to_packed);
    BA_HIP(hipGetLastError());
    return BA_OK;
}

int ba_cholesky_last_plan(int64_t *levels, int capacity, int *n_levels) {
    BA_REQUIRE(n_levels && (levels || capacity == 0), "ba_cholesky_last_plan: null pointer");
    int device = 0;
    BA_HIP(hipGetDevice(&device));
    BA_REQUIRE(device >= 0 && device < 16, "ba_cholesky_solve: device number out of range");
    CholCache &ws = g_chol_cache[device];
    std::lock_guard<std::mutex> lock(ws.mu);
    *n_levels = ws.lists_valid ? (int)(ws.split.size() / 4) : 0;
    for (int k = 0; k < *n_levels && k < capacity; ++k)
        for (int c = 0; c < 4; ++c) levels[4 * k + c] = ws.split[4 * k + c];
    return BA_OK;
}

int ba_cholesky_queued_result(int *h_info) {
    BA_REQUIRE(h_info, "ba_cholesky_queued_result: null pointer");
    int device = 0;
    BA_HIP(hipGetDevice(&device));
    BA_REQUIRE(device >= 0 && device < 16, "ba_cholesky_solve: device number out of range");
    CholCache &ws = g_chol_cache[device];
    std::lock_guard<std::mutex> lock(ws.mu);
    BA_REQUIRE(ws.result_host, "ba_cholesky_queued_result: no ba_cholesky_solve_queued call on this device yet");
    const int v = *(volatile int *)ws.result_host;
    if (v == PROFILE_MOVED) {
        ws.lists_valid = false;     // the next call reads the profile again
        *h_info = -1;
    } else {
        *h_info = v;
    }
    return BA_OK;
}

}  // extern "C"

static int solve_impl(int64_t n, int64_t ld, double *S, double *rhs, const int *block_first, int clear_on_return,
                      int *h_info, void *stream, bool queued) {
    BA_REQUIRE(S && rhs && n > 0 && ld >= n && ld % NB == 0, "ba_cholesky_solve: bad argument (ld must be a multiple of 64)");
    hipStream_t st = (hipStream_t)stream;
    const int64_t nb = ld / NB;
    int device = 0;
    BA_HIP(hipGetDevice(&device));
    BA_REQUIRE(device >= 0 && device < 16, "ba_cholesky_solve: device number out of range");
    CholCache &ws = g_chol_cache[device];
    std::lock_guard<std::mutex> lock(ws.mu);     // one call at a time per device: the workspace is shared
    hipError_t e = ws.grow(ws.linv, sizeof(double) * NB * NB * nb);
    if (e == hipSuccess) e = ws.grow(ws.tmp, sizeof(double) * ld);
    if (e == hipSuccess) e = ws.grow(ws.info, sizeof(int));
    BA_HIP(e);
    double *Linv = (double *)ws.linv.p, *tmp = (double *)ws.tmp.p;   // inverse of every diagonal block; solution vector
    int *info = (int *)ws.info.p;
    const size_t lds_trailing = sizeof(double) * 2 * KC * LDT;
    if (e == hipSuccess && lds_trailing > 64 * 1024)
        e = hipFuncSetAttribute((const void *)k_syrk_trailing, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_trailing);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void *)k_potrf64, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(sizeof(double) * POTRF_LDS_DOUBLES));
    const size_t lds_diag = sizeof(double) * DIAG_LDS_DOUBLES, lds_rows = sizeof(double) * (2 * NB * LP + 4 * 16 * TP);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void *)k_diag_trsm, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_diag);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void *)k_diag_syrk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_diag);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void *)k_panel_rows<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_rows);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void *)k_panel_rows<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_rows);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void *)k_panel_rows<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_rows);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void *)k_panel_rows<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_rows);
    const bool ride_along = ld > n;     // a padding row is there to carry the right-hand side
    // reach[k]: the last block row before the dense border (block rows from `border` on: the last unknown's and the
    // padding) with an entry at or left of column block k; see "Structural zeros" at the top
    int64_t border = (n - 1) / NB;
    std::vector<int> reach((size_t)nb), row_first((size_t)nb, 0);     // row_first[k]: first occupied column block of block row k
    for (int64_t k = 0; k < nb; ++k) reach[k] = (int)(nb - 1);
    std::vector<int> first;
    bool assumed = false;
    if (e == hipSuccess && !getenv("BA_CHOL_DENSE") && border > 1) {
        first.resize((size_t)nb);
        if (queued && ws.lists_valid && ws.guard_valid && ws.n == n && ws.ld == ld && ws.clear == clear_on_return &&
            ws.profile.size() == (size_t)nb) {
            first = ws.profile;     // taken for granted, checked on the device (k_profile_guard): nothing to wait for
            assumed = true;
        } else if (block_first) {      // the caller knows the profile (a lower bound of every row's first block will do)
            e = hipMemcpyAsync(first.data(), block_first, sizeof(int) * nb, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            for (int64_t k = 0; k < nb && e == hipSuccess; ++k)
                if (first[k] < 0 || first[k] > k) {
                    set_error("ba_cholesky_solve_tracked: block_first[k] must lie in 0 .. k");
                    return BA_ERR_ARG;
                }
        } else {
            for (int64_t k = 0; k < nb; ++k) first[k] = (int)k;
            e = ws.grow(ws.first, sizeof(int) * nb);
            int *d_first = (int *)ws.first.p;
            if (e == hipSuccess) e = hipMemcpyAsync(d_first, first.data(), sizeof(int) * nb, hipMemcpyHostToDevice, st);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(k_block_profile, dim3((unsigned)nb, (unsigned)cdiv_c(nb, PROFILE_CHUNK)), dim3(256), 0, st, S,
                                   ld, d_first);
                e = hipMemcpyAsync(first.data(), d_first, sizeof(int) * nb, hipMemcpyDeviceToHost, st);
            }
            if (e == hipSuccess) e = hipStreamSynchronize(st);
        }
        if (e == hipSuccess) {
            // The last block rows before the structural border join it when they reach far further back than the
            // typical row does (a closed camera path couples the last cameras with the first): any choice of the
            // border is correct - its rows are always processed in full - this one keeps them from defining reach.
            row_first = first;
            std::vector<int> width((size_t)border);
            for (int64_t ib = 0; ib < border; ++ib) width[ib] = (int)ib - first[ib];
            std::nth_element(width.begin(), width.begin() + border / 2, width.end());
            const int typical = width[border / 2];
            while (border > 2 && (border - 1) - first[border - 1] > 4 * typical + 4) --border;
        } else {
            first.clear();
        }
    }

    // ---- independent chains (see the top of the file), level by level: P chains of s blocks, a separator of bw blocks
    // behind each, what is left of the band, the border; what lies behind the chains of a level is a banded matrix with
    // a border again (its band: the separators, coupled to their neighbours through the chains between them) and is
    // cut the same way.  Every level's matrix is a suffix of the reordered matrix.
    struct Level {
        int64_t off = 0;            // first block of the level's matrix
        int P = 1;
        int64_t s = 0, bw = 1;      // blocks of a chain, of a separator
        int64_t B0 = 0, T0 = 0;     // counted from off: first block behind the chains, behind the separators
        std::vector<int> chain_first, chain_reach;      // per block of a chain: first column and reach, over all chains
        std::vector<int> newpos;    // position (counted from off) of the level's block ib in the level's order
        size_t moves_at = 0, moves_n = 0, cfirst_at = 0, rowpos_at = 0;
    };
    std::vector<Level> levels;
    std::vector<int4> moves;
    std::vector<int> tail_first;    // profile of what lies behind the last level's chains, counted from its first block
    const int max_levels = getenv("BA_CHOL_LEVELS") ? atoi(getenv("BA_CHOL_LEVELS")) : 3;
    if (!first.empty() && ride_along && !getenv("BA_CHOL_NO_CHAINS")) {
        std::vector<int> cur = first;       // profile of the current level's matrix, blocks counted from its start
        int64_t cur_nb = nb, cur_border = border, cur_off = 0;
        while ((int)levels.size() < max_levels && cur_border > 8) {
            Level L;
            L.off = cur_off;
            for (int64_t ib = 0; ib < cur_border; ++ib) L.bw = std::max<int64_t>(L.bw, ib - cur[ib]);
            // worth it from a third fewer steps on (a further level: from four steps on - it costs the launches of its
            // block moves and of its part of the backward solve)
            int64_t best = levels.empty() ? cur_nb - cur_nb / 3 : cur_nb - 4;
            for (int cand = 2; cand <= 96; ++cand) {
                int64_t sc = (cur_border - (int64_t)cand * L.bw) / cand;
                sc -= sc & 1;       // chains start on even blocks: the 128-row tiles of the update never straddle two chains
                if (sc < 4 * L.bw || sc < 4) break;
                const int64_t steps = sc + (cur_nb - cand * sc);
                if (steps < best) best = steps, L.P = cand, L.s = sc;
            }
            if (L.P == 1) break;
            L.B0 = (int64_t)L.P * L.s;
            L.T0 = L.B0 + (int64_t)L.P * L.bw;
            L.newpos.resize((size_t)cur_nb);
            for (int64_t ib = 0; ib < cur_nb; ++ib) {
                const int64_t c = ib / (L.s + L.bw), l = ib % (L.s + L.bw);
                if (c >= L.P) L.newpos[ib] = (int)ib;                                   // rest of the band, border: where they were
                else if (l < L.s) L.newpos[ib] = (int)(c * L.s + l);
                else L.newpos[ib] = (int)(L.B0 + c * L.bw + (l - L.s));
            }
            // the occupied blocks (from the profile), gathered, cleared, stored at their new places (k_move_blocks, below)
            L.moves_at = moves.size();
            for (int64_t ib = 0; ib < cur_nb; ++ib)
                for (int64_t jb = cur[ib]; jb <= ib; ++jb) {
                    const int from_r = (int)(L.off + ib), from_c = (int)(L.off + jb);
                    const int a = (int)(L.off + L.newpos[ib]), bcol = (int)(L.off + L.newpos[jb]);
                    moves.push_back(a >= bcol ? make_int4(from_r, from_c, a, bcol) : make_int4(from_r, from_c, -bcol - 1, a));
                }
            L.moves_n = moves.size() - L.moves_at;
            // structure of a chain, over all chains: the first column block of local row l and from it the reach
            L.chain_first.assign((size_t)L.s, 0x7fffffff);
            for (int c = 0; c < L.P; ++c)
                for (int64_t l = 0; l < L.s; ++l) {
                    const int64_t old_start = c * (L.s + L.bw);
                    L.chain_first[l] = std::min(L.chain_first[l], (int)std::max<int64_t>(cur[old_start + l] - old_start, 0));
                }
            std::vector<int> last((size_t)L.s, -1);
            for (int64_t l = 0; l < L.s; ++l) last[L.chain_first[l]] = (int)l;
            L.chain_reach.resize((size_t)L.s);
            int far = 0;
            for (int64_t k = 0; k < L.s; ++k) {
                far = std::max(far, std::max(last[k], (int)k));
                L.chain_reach[k] = far;
            }
            // Profile of what lies behind the chains once they are eliminated (blocks counted from B0): the entries
            // the matrix has there, the fill among the rows that hold entries in one and the same chain, and the fill
            // of the factorisation itself (symbolic elimination on the block pattern).
            const int64_t nt = cur_nb - L.B0;
            std::vector<std::vector<char>> pat((size_t)nt, std::vector<char>((size_t)nt, 0)),
                touch((size_t)L.P, std::vector<char>((size_t)nt, 0));
            for (int64_t ib = 0; ib < cur_nb; ++ib) {
                const int64_t a = L.newpos[ib];
                for (int64_t jb = cur[ib]; jb <= ib; ++jb) {
                    const int64_t bcol = L.newpos[jb];
                    if (a >= L.B0 && bcol >= L.B0) pat[std::max(a, bcol) - L.B0][std::min(a, bcol) - L.B0] = 1;
                    else if (a >= L.B0) touch[bcol / L.s][a - L.B0] = 1;
                    else if (bcol >= L.B0) touch[a / L.s][bcol - L.B0] = 1;
                }
            }
            std::vector<int> rows;
            for (int c = 0; c < L.P; ++c) {
                rows.clear();
                for (int64_t t = 0; t < nt; ++t)
                    if (touch[c][t]) rows.push_back((int)t);
                for (size_t x = 0; x < rows.size(); ++x)
                    for (size_t y = 0; y <= x; ++y) pat[rows[x]][rows[y]] = 1;
            }
            for (int64_t k = 0; k < nt; ++k) {
                pat[k][k] = 1;
                rows.clear();
                for (int64_t i = k + 1; i < nt; ++i)
                    if (pat[i][k]) rows.push_back((int)i);
                for (size_t x = 0; x < rows.size(); ++x)
                    for (size_t y = 0; y <= x; ++y) pat[rows[x]][rows[y]] = 1;
            }
            std::vector<int> next((size_t)nt);
            for (int64_t t = 0; t < nt; ++t) {
                int64_t j = 0;
                while (!pat[t][j]) ++j;
                next[t] = (int)j;
            }
            // its border: the level's border, and before it the rows that reach far further back than the typical one
            int64_t next_border = std::max<int64_t>(cur_border - L.B0, 0);
            if (next_border > 2) {
                std::vector<int> width((size_t)next_border);
                for (int64_t ib = 0; ib < next_border; ++ib) width[ib] = (int)ib - next[ib];
                std::nth_element(width.begin(), width.begin() + next_border / 2, width.end());
                const int typical = width[next_border / 2];
                while (next_border > 2 && (next_border - 1) - next[next_border - 1] > 4 * typical + 4) --next_border;
            }
            cur_off += L.B0;
            cur_nb = nt;
            cur_border = next_border;
            cur = next;
            tail_first = next;
            levels.push_back(std::move(L));
        }
    }
    const bool chained = !levels.empty();
    const int64_t tail0 = chained ? levels.back().off + levels.back().B0 : 0;       // first block behind all chains
    if (!chained && !first.empty() && e == hipSuccess) {
        std::vector<int> last((size_t)nb, -1);      // last[c]: the last block row (before the border) that starts at column block c
        for (int64_t ib = 0; ib < border; ++ib) last[first[ib]] = (int)ib;
        int far = 0;
        for (int64_t k = 0; k < nb; ++k) {
            far = std::max(far, std::max(last[k], (int)k));
            reach[k] = far;
        }
    }
    // Where a block ends up: pos is a block position in the order BEFORE level m's reordering.
    auto final_pos = [&](int64_t pos, size_t m) -> int64_t {
        for (; m < levels.size(); ++m)
            if (pos >= levels[m].off) pos = levels[m].off + levels[m].newpos[pos - levels[m].off];
        return pos;
    };
    // Integer lists of the kernels, in one buffer: [position of every block after level 0 | final position of every
    // block | per level: first column of a chain's blocks | per level but the last: final position of the rows behind
    // its chains | profile behind the last chains]
    std::vector<int> ints;
    size_t newpos0_at = 0, finalpos_at = 0, tfirst_at = 0;
    if (chained) {
        newpos0_at = ints.size();
        ints.insert(ints.end(), levels[0].newpos.begin(), levels[0].newpos.end());
        finalpos_at = ints.size();
        for (int64_t ib = 0; ib < nb; ++ib) ints.push_back((int)final_pos(ib, 0));
        for (size_t m = 0; m < levels.size(); ++m) {
            Level &L = levels[m];
            L.cfirst_at = ints.size();
            ints.insert(ints.end(), L.chain_first.begin(), L.chain_first.end());
            L.rowpos_at = ints.size();
            if (m + 1 < levels.size())
                for (int64_t r = L.off + L.B0; r < nb; ++r) ints.push_back((int)final_pos(r, m + 1));
        }
        tfirst_at = ints.size();
        ints.insert(ints.end(), tail_first.begin(), tail_first.end());
    }
    int *d_ints = nullptr;
    int4 *d_moves = nullptr;
    double *T = nullptr;
    if (chained && e == hipSuccess) {
        size_t most = 0;
        for (const Level &L : levels) most = std::max(most, L.moves_n);
        e = ws.grow(ws.moves, sizeof(int4) * moves.size());
        if (e == hipSuccess) e = ws.grow(ws.T, sizeof(double) * NB * NB * most);
        if (e == hipSuccess) e = ws.grow(ws.newpos, sizeof(int) * ints.size());
        d_moves = (int4 *)ws.moves.p, T = (double *)ws.T.p, d_ints = (int *)ws.newpos.p;
    }

    // Rows a launch covers when its panel ends at column block k_last and its rows start at block `start`, in units of
    // `unit` blocks; returns their number.
    //   one chain (no levels): those up to reach[k_last], then the border
    //   chains of level L: k_last and start are counted inside the chain; the blocks up to the chain's reach, the
    //           separators on either side of the chain, everything behind the separators
    //   behind the chains (dense == true): all rows (start: a block of the whole matrix)
    auto select_rows = [&](int64_t k_last, int64_t start, int unit, const Level *L, bool dense, RowSel &sel) -> int64_t {
        sel = RowSel();
        if (dense) return cdiv_c(nb - start, unit);
        const int64_t unit_rows = (int64_t)unit * NB;
        if (L) {
            const int64_t band_end = std::min<int64_t>(L->chain_reach[k_last] + 1, L->s);     // blocks of the chain kept: [start, band_end)
            const int64_t band = std::max<int64_t>(cdiv_c(std::max<int64_t>(band_end - start, 0), unit), 0);
            sel.n1 = (int)band;
            sel.n2 = (int)cdiv_c(2 * L->bw, unit);
            sel.chain_rows = L->s * NB;
            sel.win_rows = (L->B0 - start) * NB;
            sel.win_step = L->bw * NB;
            sel.tail_rows = (L->T0 - start) * NB;
            return band + sel.n2 + cdiv_c(nb - (L->off + L->T0), unit);
        }
        const int64_t total = cdiv_c(nb, unit), border_u = border / unit, reach_u = reach[k_last] / unit, start_u = start / unit;
        if (start_u >= border_u || reach_u + 1 >= border_u) return total - start_u;
        const int64_t band = std::max<int64_t>(reach_u - start_u + 1, 0);
        sel.n1 = (int)band;
        sel.tail_rows = (border_u - start_u) * unit_rows;
        return band + (total - border_u);
    };
    // the blocks cleared on return: every block that can hold anything after the factorisation - the envelope in the
    // order the factorisation worked in, the row that carried the right-hand side
    std::vector<int4> runs;
    int4 *d_runs = nullptr;
    if (clear_on_return && !first.empty() && e == hipSuccess) {
        auto add_run = [&](int64_t row, int64_t c_begin, int64_t c_last) {
            for (int64_t c = c_begin; c <= c_last; c += 32) runs.push_back(make_int4((int)row, (int)c, (int)std::min(c + 31, c_last), 0));
        };
        if (chained) {
            for (const Level &L : levels) {
                for (int c = 0; c < L.P; ++c)
                    for (int64_t l = 0; l < L.s; ++l)
                        add_run(L.off + c * L.s + l, L.off + c * L.s + L.chain_first[l], L.off + c * L.s + l);
                // the rows behind the level's chains, in the columns of its chains: a separator's rows hold entries in
                // the columns of the chains on either side of it only (chain 0's rows kernel also passes over
                // separator 1); the rows behind the separators are dense
                for (int64_t r = L.off + L.B0; r < nb; ++r) {
                    if (r < L.off + L.T0) {
                        const int64_t sep = (r - L.off - L.B0) / L.bw;
                        add_run(r, L.off + (sep > 1 ? sep : 0) * L.s, L.off + std::min<int64_t>((sep + 2) * L.s, L.B0) - 1);
                    } else {
                        add_run(r, L.off, L.off + L.B0 - 1);
                    }
                }
            }
            for (int64_t r = tail0; r < nb; ++r) add_run(r, tail0 + tail_first[r - tail0], r);
        } else {
            for (int64_t r = 0; r + 1 < nb; ++r) add_run(r, row_first[r], r);
            add_run(nb - 1, 0, nb - 1);
        }
        e = ws.grow(ws.runs, sizeof(int4) * runs.size());
        d_runs = (int4 *)ws.runs.p;
    }
    BA_HIP(e);

    const int64_t split_limit = getenv("BA_CHOL_SPLIT_LIMIT") ? atoi(getenv("BA_CHOL_SPLIT_LIMIT")) : 512;     // workgroup slots of the chip (two per CU)
    const bool look_ahead = chained && !getenv("BA_CHOL_NO_LOOKAHEAD");
    if (look_ahead && !ws.aux) {
        BA_HIP(hipStreamCreateWithFlags(&ws.aux, hipStreamNonBlocking));
        BA_HIP(hipStreamCreateWithFlags(&ws.aux2, hipStreamNonBlocking));
        BA_HIP(hipEventCreateWithFlags(&ws.ev_nc_rows, hipEventDisableTiming));
        BA_HIP(hipEventCreateWithFlags(&ws.ev_join2, hipEventDisableTiming));
        BA_HIP(hipEventCreateWithFlags(&ws.ev_diag, hipEventDisableTiming));
        BA_HIP(hipEventCreateWithFlags(&ws.ev_rows, hipEventDisableTiming));
        BA_HIP(hipEventCreateWithFlags(&ws.ev_join, hipEventDisableTiming));
    }
    // Everything the device does for this call, on stream q.
    auto issue = [&](hipStream_t q) -> hipError_t {
        hipError_t e = hipSuccess;
        auto keep = [&](hipError_t x) {     // the first failure of the calls that tie the streams together
            if (e == hipSuccess) e = x;
        };
        hipLaunchKernelGGL(k_zero_int, dim3(1), dim3(1), 0, q, info);
        if (assumed)
            hipLaunchKernelGGL(k_profile_guard, dim3(cdiv_c(nb, 256)), dim3(256), 0, q, block_first, (const int *)ws.guard.p, nb, info);
        // Factorisation of the columns [begin, end) - of every chain of level L at once when L is given (begin: the
        // first chain's first column) - and update of everything behind them.
        auto factor = [&](int64_t begin, int64_t end, const Level *L, bool dense) {
            const int n_chains = L ? L->P : 1;
            const int64_t chain_rows = L ? L->s * NB : 0;
            const int64_t origin = L ? L->off : 0;      // select_rows counts a chain's blocks from the level's first block
            // Look-ahead (chains only).  The next diagonal block of a chain waits for the chain's own rows and their
            // update only; the rows of the separators and of the border - the long strips - and every tile of the update
            // that lies in them are needed when the chains are finished, not before: they run on a second stream, in
            // the order of the panels, beside the chain of diagonal blocks.
            const bool ahead = L && look_ahead;
            // Block-tridiagonal chains (a band one block wide): the diagonal-block kernel itself takes block (k, k - 1)
            // through the inverse before it and updates its own block (with_prev), so the chain of a chain is one launch
            // per block - no launches for the blocks below inside the panel, none for the chain's next rows behind it.
            bool tridiagonal = ahead && L->bw == 1 && !getenv("BA_CHOL_NO_FUSED_DIAG");
            if (tridiagonal)
                for (int64_t l = 0; l < L->s; ++l) tridiagonal = tridiagonal && L->chain_first[l] >= std::max<int64_t>(l - 1, 0);
            // tiles of the region C[c0.., c0 .. c_end): lower triangle of the first tile columns, whole rows below
            // which: 0 every tile; with look-ahead 1 = the tiles of the chain's own rows (caller's stream), 2 = the others
            // (second stream), or those in two launches: 3 = the tiles of separators' and border's rows in the chain's
            // columns (second stream: the next panel's rows kernel reads them), 4 = in their own columns (third stream:
            // nothing reads them before the chains are finished)
            auto trailing = [&](int64_t k_first, int64_t kw, int64_t c0, int64_t c_end, int which) {
                if (c0 >= ld || c_end <= c0) return;
                RowSel sel;
                const int64_t nt = select_rows((k_first + kw - 1) / NB - origin, c0 / NB - origin, TS / NB, L, dense, sel);   // tile rows kept
                int64_t nct = 0;        // tile columns: those of the rows kept that lie before c_end
                while (nct < nt && sel(nct, TS, 0) < c_end - c0) ++nct;
                if (nct == 0) return;
                const int64_t tiles = nct * (nct + 1) / 2 + (nt - nct) * nct;
                auto launch = [&](hipStream_t sq, int64_t t_begin, int64_t count, int64_t bj_lo, int64_t bj_hi, int64_t active) {
                    if (count <= 0 || active <= 0) return;
                    int split = 1;      // workgroups per tile: as long as the launch stays within one round of the chip
                    while (active * n_chains * split * 2 <= split_limit && kw % (split * 2 * NB) == 0) split *= 2;
                    hipLaunchKernelGGL(k_syrk_trailing, dim3((unsigned)count, (unsigned)split, (unsigned)n_chains), dim3(256),
                                       lds_trailing, sq, S, ld, k_first, (int)(kw / split), c0, nct, sel, t_begin, bj_lo, bj_hi);
                };
                const int64_t all = (int64_t)1 << 40;
                if (which == 0) {
                    launch(q, 0, tiles, 0, all, tiles);
                } else {
                    // tiles are counted row by row: those of the chain's own rows come first
                    const int64_t band = std::min<int64_t>(sel.n1, nt);
                    const int64_t own = band <= nct ? band * (band + 1) / 2 : nct * (nct + 1) / 2 + (band - nct) * nct;
                    const int64_t rest_rows = nt - band;     // tile rows of separators and border
                    const int64_t in_band_cols = rest_rows * std::min(band, nct);
                    if (which == 1) launch(q, 0, own, 0, all, own);
                    else if (which == 2) launch(ws.aux, own, tiles - own, 0, all, tiles - own);
                    else if (which == 3) launch(ws.aux, own, tiles - own, 0, band, in_band_cols);               // in the chain's columns
                    else launch(ws.aux2, own, tiles - own, band, all, tiles - own - in_band_cols);              // in their own
                }
            };
            auto rows_launch = [&](hipStream_t sq, int nbk, int64_t p0, const double *Li, const RowSel &sel, int l_begin, int count) {
                if (count <= 0) return;
                const dim3 grid(count, n_chains);
                if (nbk == 4) hipLaunchKernelGGL(k_panel_rows<4>, grid, dim3(256), lds_rows, sq, S, ld, p0, Li, sel, l_begin);
                else if (nbk == 3) hipLaunchKernelGGL(k_panel_rows<3>, grid, dim3(256), lds_rows, sq, S, ld, p0, Li, sel, l_begin);
                else if (nbk == 2) hipLaunchKernelGGL(k_panel_rows<2>, grid, dim3(256), lds_rows, sq, S, ld, p0, Li, sel, l_begin);
                else hipLaunchKernelGGL(k_panel_rows<1>, grid, dim3(256), lds_rows, sq, S, ld, p0, Li, sel, l_begin);
            };
            if (tridiagonal) {
                // With many chains the second stream has more to do per panel than the chain of diagonal blocks: the
                // tiles nothing reads before the chains are finished (separators' and border's rows in their own columns)
                // go to a third stream then.  (With 14 chains that made the chain's kernels slower than it saved:
                // 2.20 -> 2.27 ms per LM iteration at 2 000 cameras; with 48 chains 5.23 -> 5.12 ms at 20 000.)
                const char *sides = getenv("BA_CHOL_SIDE_STREAMS");
                const bool two_side = sides ? atoi(sides) > 1 : L->P > 24;
                int64_t prev_p0 = -1, prev_pw = 0;      // the panel whose update on the second stream is still to be queued
                for (int64_t p0 = begin; p0 < end; p0 += PB) {
                    const int64_t pw = (end - p0 < PB) ? end - p0 : PB;
                    const int nbk = (int)(pw / NB);
                    double *Li = Linv + (p0 / NB) * NB * NB;
                    for (int kb = 0; kb < nbk; ++kb) {
                        hipLaunchKernelGGL(k_potrf64, dim3(1, n_chains), dim3(256), sizeof(double) * POTRF_LDS_DOUBLES, q, S, ld,
                                           p0 + NB * kb, Li + (int64_t)kb * NB * NB, info, chain_rows, p0 + NB * kb > begin ? 1 : 0);
                        if (kb == 0 && prev_p0 >= 0) {
                            // the panel before: its update reads block (this panel's first, that panel's last) of L,
                            // which the launch above has just stored
                            keep(hipEventRecord(ws.ev_rows, q));
                            keep(hipStreamWaitEvent(ws.aux, ws.ev_rows, 0));
                            trailing(prev_p0, prev_pw, prev_p0 + prev_pw, 2 * ld, two_side ? 3 : 2);
                        }
                    }
                    RowSel sel;
                    const int slabs = (int)select_rows((p0 + pw - 1) / NB - origin, (p0 + pw) / NB - origin, 1, L, dense, sel);
                    const int band = std::min(sel.n1, slabs);
                    keep(hipEventRecord(ws.ev_diag, q));
                    keep(hipStreamWaitEvent(ws.aux, ws.ev_diag, 0));
                    rows_launch(ws.aux, nbk, p0, Li, sel, band, slabs - band);
                    if (two_side) {
                        keep(hipEventRecord(ws.ev_nc_rows, ws.aux));
                        keep(hipStreamWaitEvent(ws.aux2, ws.ev_nc_rows, 0));
                        trailing(p0, pw, p0 + pw, 2 * ld, 4);
                    }
                    prev_p0 = p0, prev_pw = pw;
                }
                trailing(prev_p0, prev_pw, prev_p0 + prev_pw, 2 * ld, two_side ? 3 : 2);
                keep(hipEventRecord(ws.ev_join2, ws.aux2));
                keep(hipStreamWaitEvent(q, ws.ev_join2, 0));
            }
            for (int64_t o0 = begin; o0 < end && !tridiagonal; o0 += OB) {
                const int64_t o_end = (end - o0 < OB) ? end : o0 + OB;
                for (int64_t p0 = o0; p0 < o_end; p0 += PB) {
                    const int64_t pw = (o_end - p0 < PB) ? o_end - p0 : PB;
                    const int nbk = (int)(pw / NB);
                    double *Li = Linv + (p0 / NB) * NB * NB;
                    for (int kb = 0; kb < nbk; ++kb) {
                        hipLaunchKernelGGL(k_potrf64, dim3(1, n_chains), dim3(256), sizeof(double) * POTRF_LDS_DOUBLES, q, S, ld,
                                           p0 + NB * kb, Li + (int64_t)kb * NB * NB, info, chain_rows, 0);
                        const int below = nbk - 1 - kb;
                        if (below > 0) {
                            hipLaunchKernelGGL(k_diag_trsm, dim3(below, n_chains), dim3(256), lds_diag, q, S, ld, p0, kb,
                                               Li + (int64_t)kb * NB * NB, chain_rows);
                            hipLaunchKernelGGL(k_diag_syrk, dim3(below * (below + 1) / 2, n_chains), dim3(256), lds_diag, q, S, ld,
                                               p0, kb, chain_rows);
                        }
                    }
                    if (p0 + pw < ld) {
                        RowSel sel;
                        const int slabs = (int)select_rows((p0 + pw - 1) / NB - origin, (p0 + pw) / NB - origin, 1, L, dense, sel);
                        if (ahead) {
                            // the chain's own next rows at once (they start late in the panel: the sub-panel from their first
                            // column block on does); separators and border on the second stream
                            const int band = std::min(sel.n1, slabs);
                            keep(hipEventRecord(ws.ev_diag, q));
                            keep(hipStreamWaitEvent(ws.aux, ws.ev_diag, 0));
                            if (band > 0) {
                                const int64_t panel_first = p0 / NB - origin, next = (p0 + pw) / NB - origin;
                                int fk = nbk - 1;
                                for (int l = 0; l < band; ++l)
                                    fk = std::min<int64_t>(fk, std::max<int64_t>(L->chain_first[next + l] - panel_first, 0));
                                rows_launch(q, nbk - fk, p0 + (int64_t)fk * NB, Li + (int64_t)fk * NB * NB, sel, 0, band);
                            }
                            keep(hipEventRecord(ws.ev_rows, q));
                            rows_launch(ws.aux, nbk, p0, Li, sel, band, slabs - band);
                            keep(hipStreamWaitEvent(ws.aux, ws.ev_rows, 0));
                        } else {
                            rows_launch(q, nbk, p0, Li, sel, 0, slabs);
                        }
                    }
                    // inside the outer panel: only its remaining columns see this 256-wide panel now
                    if (!ahead) {
                        trailing(p0, pw, p0 + pw, o_end, 0);
                    } else {
                        // (the rows of separators and border take every panel at once, in all their columns: their
                        // updates run beside the chain, and nothing long is left for the chain's end)
                        trailing(p0, pw, p0 + pw, o_end, 1);
                        trailing(p0, pw, p0 + pw, 2 * ld, 2);
                    }
                }
                // behind the outer panel: one update with all its columns as the inner dimension
                trailing(o0, o_end - o0, o_end, 2 * ld, ahead ? 1 : 0);
            }
            if (ahead) {
                keep(hipEventRecord(ws.ev_join, ws.aux));
                keep(hipStreamWaitEvent(q, ws.ev_join, 0));
            }
        };
        if (chained) {
            for (size_t m = 0; m < levels.size(); ++m) {
                const Level &L = levels[m];
                for (int mode = 0; mode < 3; ++mode)
                    hipLaunchKernelGGL(k_move_blocks, dim3((unsigned)L.moves_n), dim3(256), 0, q, S, ld, d_moves + L.moves_at, T, mode);
                if (m == 0) hipLaunchKernelGGL(k_rhs_into_row, dim3(cdiv_c(ld, 256)), dim3(256), 0, q, S, ld, n, rhs, d_ints + newpos0_at);
                factor(L.off * NB, (L.off + L.s) * NB, &L, false);      // the level's chains, side by side
            }
            factor(tail0 * NB, ld, nullptr, true);      // what is behind the last chains: dense
        } else {
            if (ride_along) hipLaunchKernelGGL(k_rhs_into_row, dim3(cdiv_c(ld, 256)), dim3(256), 0, q, S, ld, n, rhs, (const int *)nullptr);
            factor(0, ld, nullptr, false);
        }
        // L y = b (unless it rode along), then L^T x = y; the result of every block step lands in tmp
        if (ride_along) {
            hipLaunchKernelGGL(k_row_to_rhs, dim3(cdiv_c(ld, 256)), dim3(256), 0, q, S, ld, n, rhs);
        } else {
            for (int64_t k = 0; k < nb; ++k)
                hipLaunchKernelGGL(k_solve_step, dim3(k + 1 < nb ? cdiv_c(ld - (k + 1) * NB, 256) : 1), dim3(256), 0, q, S,
                                   ld, k * NB, Linv + k * NB * NB, rhs, tmp, 0, (int64_t)0, (int64_t)0);
            e = hipMemcpyAsync(rhs, tmp, sizeof(double) * ld, hipMemcpyDeviceToDevice, q);
        }
        if (chained) {
            // (see "backward solve ... in three launches" above; a launch per block step where a workgroup's LDS
            // does not hold the right-hand side)
            const bool step_by_step = getenv("BA_CHOL_BACK_STEPS") != nullptr;
            if (nb - tail0 <= BACK_TAIL_MAX_BLOCKS && !step_by_step) {
                hipLaunchKernelGGL(k_back_tail, dim3(1), dim3(1024), sizeof(double) * NB * (nb - tail0), q, S, ld, Linv, rhs, tmp,
                                   tail0, nb, d_ints + tfirst_at);
            } else {
                for (int64_t k = nb - 1; k >= tail0; --k) {
                    const int64_t i_begin = (tail0 + tail_first[k - tail0]) * NB;
                    hipLaunchKernelGGL(k_solve_step, dim3(k * NB > i_begin ? cdiv_c(k * NB - i_begin, 256) : 1), dim3(256), 0, q, S,
                                       ld, k * NB, Linv + k * NB * NB, rhs, tmp, 1, i_begin, (int64_t)0);
                }
            }
            for (size_t m = levels.size(); m-- > 0;) {
                const Level &L = levels[m];
                const int64_t B0a = L.off + L.B0;
                hipLaunchKernelGGL(k_back_tail_to_chains, dim3(cdiv_c(L.B0 * NB, 128), (unsigned)(nb - B0a)), dim3(256), 0, q, S, ld,
                                   tmp, rhs, L.off * NB, B0a, L.off + L.T0, L.bw, L.s * NB,
                                   m + 1 < levels.size() ? d_ints + L.rowpos_at : (const int *)nullptr);
                if (L.s <= BACK_CHAIN_MAX_BLOCKS && !step_by_step) {
                    hipLaunchKernelGGL(k_back_chains, dim3(L.P), dim3(256), sizeof(double) * NB * L.s, q, S, ld, Linv, rhs, tmp,
                                       L.off, (int)L.s, d_ints + L.cfirst_at);
                } else {
                    for (int64_t k = L.s - 1; k >= 0; --k) {       // block k of every chain
                        const int64_t k0 = (L.off + k) * NB, i_begin = (L.off + L.chain_first[k]) * NB;
                        hipLaunchKernelGGL(k_solve_step, dim3(k0 > i_begin ? cdiv_c(k0 - i_begin, 256) : 1, L.P), dim3(256), 0, q,
                                           S, ld, k0, Linv + (L.off + k) * NB * NB, rhs, tmp, 1, i_begin, L.s * NB);
                    }
                }
            }
            hipLaunchKernelGGL(k_solution_back, dim3(cdiv_c(ld, 256)), dim3(256), 0, q, tmp, ld, d_ints + finalpos_at, rhs);
        } else {
            for (int64_t k = nb - 1; k >= 0; --k) {
                const int64_t i_begin = (int64_t)row_first[k] * NB;     // L has nothing left of the row's first entry
                hipLaunchKernelGGL(k_solve_step, dim3(k * NB > i_begin ? cdiv_c(k * NB - i_begin, 256) : 1), dim3(256), 0, q, S,
                                   ld, k * NB, Linv + k * NB * NB, rhs, tmp, 1, i_begin, (int64_t)0);
            }
            if (e == hipSuccess) e = hipMemcpyAsync(rhs, tmp, sizeof(double) * ld, hipMemcpyDeviceToDevice, q);
        }
        // The caller keeps S for the next system: what the factor occupied is cleared, at the cost of the entries of L
        // instead of the n^2 of the whole matrix.
        if (clear_on_return && e == hipSuccess) {
            if (first.empty()) hipLaunchKernelGGL(k_zero_doubles, dim3(4096), dim3(256), 0, q, S, ld * ld);
            else hipLaunchKernelGGL(k_clear_runs, dim3((unsigned)runs.size()), dim3(256), 0, q, S, ld, d_runs);
        }
        return e;
    };

    // the lists that steer the moves and the clearing: uploaded when the profile they follow is not the last call's
    // (waited for: the host vectors are this call's)
    // (the layout also depends on the split into chains, which a switch can change under one and the same profile)
    std::vector<int64_t> split;
    for (const Level &L : levels) split.insert(split.end(), {L.off, (int64_t)L.P, L.s, L.bw});
    const bool same = ws.lists_valid && ws.n == n && ws.ld == ld && ws.clear == clear_on_return && ws.profile == first &&
                      ws.split == split;
    if (!same) {
        ws.lists_valid = false;
        if (chained) {
            e = hipMemcpy(d_moves, moves.data(), sizeof(int4) * moves.size(), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(d_ints, ints.data(), sizeof(int) * ints.size(), hipMemcpyHostToDevice);
        }
        if (e == hipSuccess && !runs.empty())
            e = hipMemcpy(d_runs, runs.data(), sizeof(int4) * runs.size(), hipMemcpyHostToDevice);
        BA_HIP(e);
        ws.n = n, ws.ld = ld, ws.clear = clear_on_return, ws.profile = first, ws.lists_valid = true, ws.guard_valid = false;
        ws.split = split;
    }
    if (queued && !first.empty() && !ws.guard_valid) {      // the profile later queued calls are checked against
        const bool valid = ws.lists_valid;
        e = ws.grow(ws.guard, sizeof(int) * nb);        // (a buffer that grows invalidates the lists' flag: not these lists)
        ws.lists_valid = valid;
        if (e == hipSuccess) e = hipMemcpy(ws.guard.p, first.data(), sizeof(int) * nb, hipMemcpyHostToDevice);
        BA_HIP(e);
        ws.guard_valid = true;
    }
    if (queued && !ws.result_host) {
        BA_HIP(hipHostMalloc((void **)&ws.result_host, 64, hipHostMallocMapped));
        BA_HIP(hipHostGetDevicePointer((void **)&ws.result_dev, ws.result_host, 0));
    }
    e = issue(st);
    if (queued && e == hipSuccess) {
        hipLaunchKernelGGL(k_publish_info, dim3(1), dim3(1), 0, st, info, ws.result_dev);
        BA_HIP(hipGetLastError());
        return BA_OK;
    }
    if (e == hipSuccess) e = hipGetLastError();
    int hinfo = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&hinfo, info, sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    BA_HIP(e);
    if (h_info) *h_info = hinfo;
    if (hinfo != 0) {
        set_error("ba_cholesky_solve: matrix is not positive definite (pivot " + std::to_string(hinfo) + ")");
        return BA_ERR_NUMERIC;
    }
    return BA_OK;
}
