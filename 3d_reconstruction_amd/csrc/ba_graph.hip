// Set-up work on the observation list that needs the whole graph (internal; C ABI in include/ba_mi355x.h):
//
//   ba_visibility_components          connected components of the bipartite point / camera visibility graph, the check
//                                     the reference makes on a dense (N + K)^2 adjacency matrix (p2w:519-525)
//   ba_order_points_by_first_camera   the engine's internal point order (ascending first observing camera, ties in the
//                                     caller's order) and the observation list rewritten in it
//
// Both work on the device copy of the COO list; nothing of size |V| or N travels to the host.
#include <algorithm>

#include "ba_common.h"

namespace ba {
namespace {

inline unsigned blocks_of(int64_t n, int per_block) { return (unsigned)std::max<int64_t>(1, (n + per_block - 1) / per_block); }

// ---------------------------------------------------------------------------------------------------------------
// Connected components: nodes 0 .. K-1 are the cameras, K .. K+N-1 the points (so that the smallest label of a
// component with a camera is a camera).  Rounds of "hook the larger root under the smaller one" over the edges and
// full path compression over the nodes, until a round hooks nothing.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int cc_root(const int *parent, int v) {
    int p = __hip_atomic_load(parent + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != v) {
        v = p;
        p = __hip_atomic_load(parent + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return v;
}

__global__ void k_cc_init(int64_t n, int *__restrict__ parent) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n) parent[v] = (int)v;
}

__global__ void k_cc_hook(int64_t nV, int K, const int32_t *__restrict__ oi, const int32_t *__restrict__ oj, int *parent,
                          int *changed) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool hooked = false;
    if (e < nV) {
        const int ra = cc_root(parent, oj[e]), rb = cc_root(parent, K + oi[e]);
        if (ra != rb) {
            // a root that somebody else hooks meanwhile keeps the smaller of the two parents: the edge is looked at
            // again in the next round
            atomicMin(parent + max(ra, rb), min(ra, rb));
            hooked = true;
        }
    }
    if (__ballot(hooked) && (threadIdx.x & 63) == 0) *changed = 1;
}

__global__ void k_cc_compress(int64_t n, int *parent) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n) {
        const int r = cc_root(parent, (int)v);
        __hip_atomic_store(parent + v, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// counts[0]: components that contain a camera, counts[1]: components of points alone; camera j's label
__global__ void k_cc_count(int64_t n, int K, const int *__restrict__ parent, unsigned long long *counts,
                           int32_t *__restrict__ camera_label) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool root = v < n && parent[v] == (int)v;
    if (v < K && camera_label) camera_label[v] = parent[v];
    const unsigned long long cam = __ballot(root && v < K), pts = __ballot(root && v >= K);
    if ((threadIdx.x & 63) == 0) {
        if (cam) atomicAdd(counts, (unsigned long long)__popcll(cam));
        if (pts) atomicAdd(counts + 1, (unsigned long long)__popcll(pts));
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Point order
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_fill_int(int64_t n, int *__restrict__ a, int value) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n) a[v] = value;
}

// The list is sorted by point: the thread of a point's first observation walks the point's observations (a minimum
// by atomics, one per observation, took 0.37 ms at ten million observations).
__global__ void k_first_camera(int64_t nV, const int32_t *__restrict__ oi, const int32_t *__restrict__ oj, int *first,
                               int *flag) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nV) return;
    const int i = oi[e];
    const int before = e > 0 ? oi[e - 1] : -1;
    if (before > i && (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 2) == 0)
        atomicOr(flag, 2);      // the list is not sorted by point (the caller gives up: what is written below is not used)
    if (before == i) return;
    int m = oj[e];
    for (int64_t f = e + 1; f < nV && oi[f] == i; ++f) m = min(m, oj[f]);
    first[i] = m;
}

__global__ void k_iota_and_sorted(int64_t n, const int *__restrict__ key, int *__restrict__ val, int *unsorted) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    val[v] = (int)v;
    // one atomic per wave at most, and none once the flag is up (on a shuffled list every second lane finds a
    // descent: ten million atomics on one address took 1.8 ms)
    const bool descent = v > 0 && key[v] < key[v - 1];
    if (descent && (__hip_atomic_load(unsorted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1) == 0) {
        const unsigned long long m = __ballot(true);
        if ((threadIdx.x & 63) == __ffsll((long long)m) - 1) atomicOr(unsorted, 1);
    }
}

// Stable radix sort of (key, value) pairs by 8-bit digits.  A workgroup is ONE wave that walks its 2048 consecutive
// items 64 at a time, so "earlier in the input" is "earlier chunk, or lower lane": the rank of an item among the
// items of the same digit is a population count over a ballot mask plus a per-digit counter in LDS.
constexpr int RS_CHUNKS = 32, RS_ITEMS = 64 * RS_CHUNKS;

__device__ __forceinline__ unsigned long long same_digit_lanes(int digit, bool active) {
    unsigned long long mask = __ballot(active);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool bit = (digit >> b) & 1;
        const unsigned long long bal = __ballot(active && bit);
        mask &= bit ? bal : ~bal;
    }
    return mask;
}

__global__ void __launch_bounds__(64) k_radix_hist(const int *__restrict__ key, int64_t n, int shift, int *__restrict__ hist,
                                                   int n_blocks) {
    __shared__ int cnt[256];
    const int lane = threadIdx.x;
    for (int d = lane; d < 256; d += 64) cnt[d] = 0;
    __builtin_amdgcn_wave_barrier();
    const int64_t base = (int64_t)blockIdx.x * RS_ITEMS;
    for (int c = 0; c < RS_CHUNKS; ++c) {
        const int64_t idx = base + 64 * c + lane;
        if (idx < n) atomicAdd(&cnt[(key[idx] >> shift) & 255], 1);
    }
    __builtin_amdgcn_wave_barrier();
    for (int d = lane; d < 256; d += 64) hist[(int64_t)d * n_blocks + blockIdx.x] = cnt[d];
}

// Exclusive scan of the digit-major histogram (256 x blocks entries; 1.25 M at ten million items), three launches: a
// scan per 4096 entries, the scan of the block sums by one workgroup (k_scan_int_single), their addition.
constexpr int S32_ITEMS = 16, S32_BLOCK = 256, S32_TILE = S32_ITEMS * S32_BLOCK;

__global__ void __launch_bounds__(S32_BLOCK) k_scan_int_blocks(int *__restrict__ a, int64_t n, int *__restrict__ block_sums) {
    __shared__ int wsum[S32_BLOCK / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * S32_TILE + (int64_t)threadIdx.x * S32_ITEMS;
    int v[S32_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < S32_ITEMS; ++k) {
        v[k] = base + k < n ? a[base + k] : 0;
        s += v[k];
    }
    int inc = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int u = __shfl_up(inc, off);
        if (lane >= off) inc += u;
    }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wid; ++w) before += wsum[w];
    int ex = before + inc - s;
#pragma unroll
    for (int k = 0; k < S32_ITEMS; ++k) {
        if (base + k < n) a[base + k] = ex;
        ex += v[k];
    }
    if (threadIdx.x == S32_BLOCK - 1) block_sums[blockIdx.x] = ex;
}

__global__ void k_scan_int_add(int *__restrict__ a, int64_t n, const int *__restrict__ block_sums) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) a[t] += block_sums[t / S32_TILE];
}

// exclusive scan of an int array by one workgroup
__global__ void __launch_bounds__(1024) k_scan_int_single(int *a, int64_t n) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < n; base += 1024) {
        const int64_t t = base + threadIdx.x;
        const int v = t < n ? a[t] : 0;
        int inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int u = __shfl_up(inc, off);
            if (lane >= off) inc += u;
        }
        if (lane == 63) wsum[wid] = inc;
        __syncthreads();
        int before = carry_s;
        for (int w = 0; w < wid; ++w) before += wsum[w];
        if (t < n) a[t] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = before + inc;
        __syncthreads();
    }
}

__global__ void __launch_bounds__(64) k_radix_scatter(const int *__restrict__ key_in, const int *__restrict__ val_in, int64_t n,
                                                      int shift, const int *__restrict__ offsets, int n_blocks,
                                                      int *__restrict__ key_out, int *__restrict__ val_out) {
    __shared__ int pos[256];
    const int lane = threadIdx.x;
    for (int d = lane; d < 256; d += 64) pos[d] = offsets[(int64_t)d * n_blocks + blockIdx.x];
    __builtin_amdgcn_wave_barrier();
    const int64_t base = (int64_t)blockIdx.x * RS_ITEMS;
    for (int c = 0; c < RS_CHUNKS; ++c) {
        const int64_t idx = base + 64 * c + lane;
        const bool active = idx < n;
        const int k = active ? key_in[idx] : 0, v = active ? val_in[idx] : 0;
        const int d = (k >> shift) & 255;
        const unsigned long long mask = same_digit_lanes(d, active);
        const int rank = __popcll(mask & ((1ull << lane) - 1ull));
        int dst = 0;
        if (active) dst = pos[d] + rank;
        __builtin_amdgcn_wave_barrier();
        if (active && rank == 0) pos[d] += __popcll(mask);
        __builtin_amdgcn_wave_barrier();
        if (active) {
            key_out[dst] = k;
            val_out[dst] = v;
        }
    }
}

// observation range of every point from the sorted point indices (row_ptr[N + 1])
__global__ void k_point_ranges(const int32_t *__restrict__ oi, int64_t nV, int64_t N, int64_t *__restrict__ row_ptr) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e > nV) return;
    const int64_t cur = e < nV ? oi[e] : N, prev = e > 0 ? oi[e - 1] : -1;
    for (int64_t p = prev + 1; p <= cur; ++p) row_ptr[p] = e;
}

// inv[perm[n]] = n and the observation count of the n-th point of the new order
__global__ void k_inverse_and_degree(int64_t N, const int *__restrict__ perm, const int64_t *__restrict__ row_ptr,
                                     int *__restrict__ inv, int64_t *__restrict__ deg_new) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int p = perm[n];
    inv[p] = (int)n;
    deg_new[n] = row_ptr[p + 1] - row_ptr[p];
}

// exclusive scan of int64 values, three launches: per-block scan (256 threads x 16 items), scan of the block sums by
// one workgroup, addition of the block offsets; out[n] = total
constexpr int SC_ITEMS = 16, SC_BLOCK = 256, SC_TILE = SC_ITEMS * SC_BLOCK;

__global__ void __launch_bounds__(SC_BLOCK) k_scan64_blocks(const int64_t *__restrict__ in, int64_t n, int64_t *__restrict__ out,
                                                            int64_t *__restrict__ block_sums) {
    __shared__ int64_t wsum[SC_BLOCK / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * SC_TILE + (int64_t)threadIdx.x * SC_ITEMS;
    int64_t v[SC_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < SC_ITEMS; ++k) {
        v[k] = base + k < n ? in[base + k] : 0;
        s += v[k];
    }
    int64_t inc = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int64_t u = __shfl_up(inc, off);
        if (lane >= off) inc += u;
    }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    int64_t before = 0;
    for (int w = 0; w < wid; ++w) before += wsum[w];
    int64_t ex = before + inc - s;
#pragma unroll
    for (int k = 0; k < SC_ITEMS; ++k) {
        if (base + k < n) out[base + k] = ex;
        ex += v[k];
    }
    if (threadIdx.x == SC_BLOCK - 1) block_sums[blockIdx.x] = ex;
}

__global__ void k_scan64_sums(int64_t *block_sums, int64_t n_blocks) {      // one thread: a few thousand entries
    int64_t acc = 0;
    for (int64_t b = 0; b < n_blocks; ++b) {
        const int64_t v = block_sums[b];
        block_sums[b] = acc;
        acc += v;
    }
    block_sums[n_blocks] = acc;
}

__global__ void k_scan64_add(int64_t *__restrict__ out, int64_t n, const int64_t *__restrict__ block_sums, int64_t n_blocks) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] += block_sums[t / SC_TILE];
    if (t == 0) out[n] = block_sums[n_blocks];
}

__global__ void k_scatter_observations(int64_t nV, const int32_t *__restrict__ oi, const int32_t *__restrict__ oj,
                                       const double2 *__restrict__ ouv, const int64_t *__restrict__ row_ptr,
                                       const int *__restrict__ inv, const int64_t *__restrict__ row_ptr_new,
                                       int32_t *__restrict__ i_out, int32_t *__restrict__ j_out, double2 *__restrict__ uv_out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nV) return;
    const int p = oi[e], n = inv[p];
    const int64_t dst = row_ptr_new[n] + (e - row_ptr[p]);
    i_out[dst] = n;
    j_out[dst] = oj[e];
    uv_out[dst] = ouv[e];
}

__global__ void k_gather_int(int64_t n, const int32_t *__restrict__ src, const int *__restrict__ idx, int *__restrict__ dst) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n) dst[v] = src[idx[v]];
}

__global__ void k_iota_check_order(int64_t n, const int32_t *__restrict__ key, int *__restrict__ key_copy, int *__restrict__ val,
                                   int *unsorted) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    val[v] = (int)v;
    key_copy[v] = key[v];
    // one atomic per wave at most, and none once the flag is up (on a shuffled list every second lane finds a
    // descent: ten million atomics on one address took 1.8 ms)
    const bool descent = v > 0 && key[v] < key[v - 1];
    if (descent && (__hip_atomic_load(unsorted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1) == 0) {
        const unsigned long long m = __ballot(true);
        if ((threadIdx.x & 63) == __ffsll((long long)m) - 1) atomicOr(unsorted, 1);
    }
}

struct DeviceBuffers {      // frees what it holds when the call returns, whichever way
    std::vector<void *> held;
    template <class T>
    hipError_t get(T **p, size_t count) {
        void *q = nullptr;
        const hipError_t e = hipMalloc(&q, sizeof(T) * std::max<size_t>(count, 1));
        if (e == hipSuccess) held.push_back(q);
        *p = (T *)q;
        return e;
    }
    ~DeviceBuffers() {
        for (void *q : held) hipFree(q);
    }
};

// ints of the histogram buffer of a sort of n items: the histogram and the block sums of its scan
size_t radix_hist_ints(int64_t n) {
    const size_t h = (size_t)256 * blocks_of(n, RS_ITEMS);
    return h + blocks_of((int64_t)h, S32_TILE);
}

// Stable sort of the pairs (key[cur], val[cur]) by the low `bits` bits of the keys; returns the index of the buffer
// pair that holds the result.  hist: radix_hist_ints(n) ints.
int radix_sort_pairs(int64_t n, int *key[2], int *val[2], int *hist, int bits, hipStream_t st) {
    const int n_blocks = (int)blocks_of(n, RS_ITEMS);
    const int64_t n_hist = (int64_t)256 * n_blocks, scan_blocks = blocks_of(n_hist, S32_TILE);
    int *block_sums = hist + n_hist;
    int cur = 0;
    for (int shift = 0; shift < bits; shift += 8) {
        hipLaunchKernelGGL(k_radix_hist, dim3(n_blocks), dim3(64), 0, st, key[cur], n, shift, hist, n_blocks);
        if (scan_blocks <= 1) {
            hipLaunchKernelGGL(k_scan_int_single, dim3(1), dim3(1024), 0, st, hist, n_hist);
        } else {
            hipLaunchKernelGGL(k_scan_int_blocks, dim3((unsigned)scan_blocks), dim3(S32_BLOCK), 0, st, hist, n_hist, block_sums);
            hipLaunchKernelGGL(k_scan_int_single, dim3(1), dim3(1024), 0, st, block_sums, scan_blocks);
            hipLaunchKernelGGL(k_scan_int_add, dim3((unsigned)blocks_of(n_hist, 256)), dim3(256), 0, st, hist, n_hist, block_sums);
        }
        hipLaunchKernelGGL(k_radix_scatter, dim3(n_blocks), dim3(64), 0, st, key[cur], val[cur], n, shift, hist, n_blocks,
                           key[cur ^ 1], val[cur ^ 1]);
        cur ^= 1;
    }
    return cur;
}

int bits_for(int64_t max_key) {
    int bits = 1;
    while (((int64_t)1 << bits) <= max_key) ++bits;
    return bits;
}

}  // namespace
}  // namespace ba

using namespace ba;

extern "C" {

int ba_visibility_components(int64_t N, int64_t K, int64_t nV, const int32_t *i, const int32_t *j, int32_t *camera_label,
                             int64_t *h_camera_components, int64_t *h_point_components, void *stream) {
    BA_REQUIRE(N >= 0 && K >= 0 && nV >= 0 && (nV == 0 || (i && j)), "ba_visibility_components: bad argument");
    BA_REQUIRE(N + K < ((int64_t)1 << 31), "ba_visibility_components: more than 2^31 nodes");
    BA_REQUIRE(h_camera_components && h_point_components, "ba_visibility_components: null output");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = N + K;
    if (n == 0) {
        *h_camera_components = *h_point_components = 0;
        return BA_OK;
    }
    DeviceBuffers buf;
    int *parent = nullptr, *changed = nullptr;
    unsigned long long *counts = nullptr;
    BA_HIP(buf.get(&parent, (size_t)n));
    BA_HIP(buf.get(&changed, 1));
    BA_HIP(buf.get(&counts, 2));
    hipLaunchKernelGGL(k_cc_init, dim3(blocks_of(n, 256)), dim3(256), 0, st, n, parent);
    for (int round = 0; nV > 0; ++round) {
        BA_HIP(hipMemsetAsync(changed, 0, sizeof(int), st));
        hipLaunchKernelGGL(k_cc_hook, dim3(blocks_of(nV, 256)), dim3(256), 0, st, nV, (int)K, i, j, parent, changed);
        hipLaunchKernelGGL(k_cc_compress, dim3(blocks_of(n, 256)), dim3(256), 0, st, n, parent);
        int h_changed = 0;
        BA_HIP(hipMemcpyAsync(&h_changed, changed, sizeof(int), hipMemcpyDeviceToHost, st));
        BA_HIP(hipStreamSynchronize(st));
        if (!h_changed) break;
        if (round > 4096) {
            set_error("ba_visibility_components: the labels did not settle");
            return BA_ERR_STATE;
        }
    }
    BA_HIP(hipMemsetAsync(counts, 0, 2 * sizeof(unsigned long long), st));
    hipLaunchKernelGGL(k_cc_count, dim3(blocks_of(n, 256)), dim3(256), 0, st, n, (int)K, parent, counts, camera_label);
    unsigned long long h_counts[2] = {0, 0};
    BA_HIP(hipMemcpyAsync(h_counts, counts, sizeof(h_counts), hipMemcpyDeviceToHost, st));
    BA_HIP(hipStreamSynchronize(st));
    BA_HIP(hipGetLastError());
    *h_camera_components = (int64_t)h_counts[0];
    *h_point_components = (int64_t)h_counts[1];
    return BA_OK;
}

int ba_order_points_by_first_camera(int64_t N, int64_t K, int64_t nV, const int32_t *i, const int32_t *j, const double *uv,
                                    int32_t *perm, int32_t *i_out, int32_t *j_out, double *uv_out, int *h_identity,
                                    void *stream) {
    BA_REQUIRE(N > 0 && K > 0 && nV >= 0 && (nV == 0 || (i && j && uv)), "ba_order_points_by_first_camera: bad argument");
    BA_REQUIRE(perm && i_out && j_out && uv_out && h_identity, "ba_order_points_by_first_camera: null output");
    BA_REQUIRE(N < ((int64_t)1 << 31) && K < ((int64_t)1 << 30), "ba_order_points_by_first_camera: index range");
    hipStream_t st = (hipStream_t)stream;
    DeviceBuffers buf;
    int *key[2] = {nullptr, nullptr}, *val[2] = {nullptr, nullptr}, *flag = nullptr;
    BA_HIP(buf.get(&key[0], (size_t)N));
    BA_HIP(buf.get(&val[0], (size_t)N));
    BA_HIP(buf.get(&flag, 1));
    // first observing camera of every point (K: nobody sees it; such points go last)
    hipLaunchKernelGGL(k_fill_int, dim3(blocks_of(N, 256)), dim3(256), 0, st, N, key[0], (int)K);
    BA_HIP(hipMemsetAsync(flag, 0, sizeof(int), st));
    if (nV > 0) hipLaunchKernelGGL(k_first_camera, dim3(blocks_of(nV, 256)), dim3(256), 0, st, nV, i, j, key[0], flag);
    hipLaunchKernelGGL(k_iota_and_sorted, dim3(blocks_of(N, 256)), dim3(256), 0, st, N, key[0], val[0], flag);
    int unsorted = 0;
    BA_HIP(hipMemcpyAsync(&unsorted, flag, sizeof(int), hipMemcpyDeviceToHost, st));
    BA_HIP(hipStreamSynchronize(st));
    BA_HIP(hipGetLastError());
    if (unsorted & 2) {
        set_error("ba_order_points_by_first_camera: observations must be sorted by point index");
        return BA_ERR_ARG;
    }
    *h_identity = unsorted ? 0 : 1;
    if (!unsorted) return BA_OK;

    // stable sort of the points by that key
    BA_HIP(buf.get(&key[1], (size_t)N));
    BA_HIP(buf.get(&val[1], (size_t)N));
    int *hist = nullptr;
    BA_HIP(buf.get(&hist, radix_hist_ints(N)));
    const int cur = radix_sort_pairs(N, key, val, hist, bits_for(K), st);      // keys 0 .. K
    BA_HIP(hipMemcpyAsync(perm, val[cur], sizeof(int) * N, hipMemcpyDeviceToDevice, st));

    // the observation list in the new order: every point's observations keep their order and move as a block
    int64_t *row_ptr = nullptr, *deg_new = nullptr, *row_ptr_new = nullptr, *block_sums = nullptr;
    int *inv = key[cur];        // the sorted keys are not needed any more
    const int64_t sc_blocks = blocks_of(N, SC_TILE);
    BA_HIP(buf.get(&row_ptr, (size_t)N + 1));
    BA_HIP(buf.get(&deg_new, (size_t)N));
    BA_HIP(buf.get(&row_ptr_new, (size_t)N + 1));
    BA_HIP(buf.get(&block_sums, (size_t)sc_blocks + 1));
    hipLaunchKernelGGL(k_point_ranges, dim3(blocks_of(nV + 1, 256)), dim3(256), 0, st, i, nV, N, row_ptr);
    hipLaunchKernelGGL(k_inverse_and_degree, dim3(blocks_of(N, 256)), dim3(256), 0, st, N, perm, row_ptr, inv, deg_new);
    hipLaunchKernelGGL(k_scan64_blocks, dim3((unsigned)sc_blocks), dim3(SC_BLOCK), 0, st, deg_new, N, row_ptr_new, block_sums);
    hipLaunchKernelGGL(k_scan64_sums, dim3(1), dim3(1), 0, st, block_sums, sc_blocks);
    hipLaunchKernelGGL(k_scan64_add, dim3(blocks_of(N, 256)), dim3(256), 0, st, row_ptr_new, N, block_sums, sc_blocks);
    if (nV > 0)
        hipLaunchKernelGGL(k_scatter_observations, dim3(blocks_of(nV, 256)), dim3(256), 0, st, nV, i, j, (const double2 *)uv,
                           row_ptr, inv, row_ptr_new, i_out, j_out, (double2 *)uv_out);
    BA_HIP(hipGetLastError());
    BA_HIP(hipStreamSynchronize(st));       // the work buffers are freed on return
    return BA_OK;
}

int ba_argsort_by_camera(int64_t nV, int64_t N, int64_t K, const int32_t *i, const int32_t *j, int32_t *order,
                         void *stream) {
    BA_REQUIRE(nV >= 0 && N > 0 && K > 0 && (nV == 0 || (i && j && order)), "ba_argsort_by_camera: bad argument");
    BA_REQUIRE(nV < ((int64_t)1 << 31) && N < ((int64_t)1 << 31) && K < ((int64_t)1 << 31), "ba_argsort_by_camera: index range");
    if (nV == 0) return BA_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceBuffers buf;
    int *key[2] = {nullptr, nullptr}, *val[2] = {nullptr, nullptr}, *hist = nullptr, *flag = nullptr;
    for (int b = 0; b < 2; ++b) {
        BA_HIP(buf.get(&key[b], (size_t)nV));
        BA_HIP(buf.get(&val[b], (size_t)nV));
    }
    BA_HIP(buf.get(&hist, radix_hist_ints(nV)));
    BA_HIP(buf.get(&flag, 1));
    BA_HIP(hipMemsetAsync(flag, 0, sizeof(int), st));
    // is the list already in point order?  then one stable sort by camera gives (camera, point ascending)
    hipLaunchKernelGGL(k_iota_check_order, dim3(blocks_of(nV, 256)), dim3(256), 0, st, nV, i, key[0], val[0], flag);
    int unsorted = 0;
    BA_HIP(hipMemcpyAsync(&unsorted, flag, sizeof(int), hipMemcpyDeviceToHost, st));
    BA_HIP(hipStreamSynchronize(st));
    int cur = 0;
    if (unsorted) cur = radix_sort_pairs(nV, key, val, hist, bits_for(N - 1), st);
    // keys of the second sort: the cameras in the order reached so far
    int *k2[2] = {key[cur], key[cur ^ 1]}, *v2[2] = {val[cur], val[cur ^ 1]};
    hipLaunchKernelGGL(k_gather_int, dim3(blocks_of(nV, 256)), dim3(256), 0, st, nV, j, v2[0], k2[0]);
    const int c2 = radix_sort_pairs(nV, k2, v2, hist, bits_for(K - 1), st);
    BA_HIP(hipMemcpyAsync(order, v2[c2], sizeof(int) * nV, hipMemcpyDeviceToDevice, st));
    BA_HIP(hipGetLastError());
    BA_HIP(hipStreamSynchronize(st));       // the work buffers are freed on return
    return BA_OK;
}

int ba_ranges_of_sorted_keys(int64_t n, int64_t n_keys, const int32_t *keys, int64_t *ranges, void *stream) {
    BA_REQUIRE(n >= 0 && n_keys >= 0 && (n == 0 || keys) && ranges, "ba_ranges_of_sorted_keys: bad argument");
    hipLaunchKernelGGL(k_point_ranges, dim3(blocks_of(n + 1, 256)), dim3(256), 0, (hipStream_t)stream, keys, n, n_keys, ranges);
    BA_HIP(hipGetLastError());
    return BA_OK;
}

}  // extern "C"
