// Sum of a short fp64 vector over the GPUs of one node without a collective library: every GPU
// publishes its vector in a window of its own memory that the other GPUs have mapped (HIP IPC) and
// adds up all windows in rank order (the same order everywhere: all ranks end with bit-identical
// sums).  One kernel and one trip over xGMI; meant for the per-iteration camera-block sum of the fit
// loop (~100 KB), where a ring or tree collective is bound by its latency.
//
// Window: double slots[4][capacity] of "uncached" memory (stores go out to memory, loads of another
// GPU never meet a stale line).  Call c uses slot c % 4.  There are no flags: an element that has not
// been published yet holds EMPTY, a signalling-NaN pattern no arithmetic produces (a value with
// exactly these bits is published as the canonical quiet NaN), and a reader polls the element itself -
// aligned 8-byte stores arrive whole.  Slots are set back to EMPTY two calls after their use: a GPU that
// starts call c has finished call c - 1, for which it has read every rank's elements of call c - 1; those
// ranks had therefore finished call c - 2 (calls of one exchange run one after the other on a GPU), so
// nobody reads the slot of call c - 2 any more and the kernel of call c clears it in the own window, over
// the length that call had.  The slot is used again by call c + 2; a rank that looks at it then has seen this
// GPU's call c + 1, whose kernel started after the kernel of call c - and with it the clearing - was complete.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>

#include "ba_common.h"
#include "ba_mi355x.h"

namespace ba {

constexpr int EX_MAX_RANKS = 16;
constexpr int EX_MAX_BLOCKS = 256;
constexpr int EX_BLOCK = 256;
constexpr int EX_CHUNK = 1024;                    // doubles one workgroup publishes and sums
constexpr int EX_SLOTS = 4;
constexpr uint32_t EX_EMPTY_HALF = 0x7ff4a5a5u;   // both halves of EMPTY (the window is filled 32 bits at a time)
constexpr uint64_t EX_EMPTY = ((uint64_t)EX_EMPTY_HALF << 32) | EX_EMPTY_HALF;
constexpr uint64_t EX_QUIET_NAN = 0x7ff8000000000000ull;

struct ExchangeArgs {
    char *window[EX_MAX_RANKS];                   // window[rank] is this GPU's own
    double *buf;
    int64_t n, n_old, capacity;                   // n_old: length of the call two calls ago (0: none)
    int rank, world;
    uint32_t call;
    long long wait_ticks;                         // upper bound for the wait (wall_clock64 ticks, 100 MHz)
    int *status;                                  // host memory: set to 1 when a wait ran out
};

__global__ void __launch_bounds__(EX_BLOCK) k_exchange_sum(ExchangeArgs a) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t lo = (int64_t)b * EX_CHUNK;
    const int64_t hi = lo + EX_CHUNK < a.n ? lo + EX_CHUNK : a.n;
    const size_t slot_off = sizeof(double) * (size_t)(a.call % EX_SLOTS) * (size_t)a.capacity;
    const size_t old_off = sizeof(double) * (size_t)((a.call + 2) % EX_SLOTS) * (size_t)a.capacity;
    constexpr int PER = EX_CHUNK / EX_BLOCK;                    // elements per thread

    // 0. clear the slot of the call before the previous one (the grid covers the longer of the two lengths)
    {
        uint64_t *old_slot = (uint64_t *)(a.window[a.rank] + old_off);
        const int64_t old_hi = lo + EX_CHUNK < a.n_old ? lo + EX_CHUNK : a.n_old;
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const int64_t i = lo + tid + e * EX_BLOCK;
            if (i < old_hi) __hip_atomic_store(old_slot + i, EX_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (lo >= a.n) return;

    // 1. publish.  Every access to a window is a system-scope atomic on uncached memory (it goes to memory,
    // past both cache levels): no cache has to be written back or invalidated, and no fence is needed - a fence
    // of system scope would write back the whole L2, which is full of the pass's point state at this moment.
    uint64_t *mine = (uint64_t *)(a.window[a.rank] + slot_off);
    uint64_t own[PER];
    {
        double v[PER];
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const int64_t i = lo + tid + e * EX_BLOCK;
            v[e] = i < hi ? a.buf[i] : 0.0;
        }
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const int64_t i = lo + tid + e * EX_BLOCK;
            own[e] = (uint64_t)__double_as_longlong(v[e]);
            if (own[e] == EX_EMPTY) own[e] = EX_QUIET_NAN;
            if (i < hi) __hip_atomic_store(mine + i, own[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }

    // 2. read every rank's elements until none is EMPTY (bounded: a rank that never arrives must not hang the
    // GPU), eight ranks' loads in flight together, and add them up, rank 0 first
    double acc[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) acc[e] = 0.0;
    int failed = 0;
    const long long t0 = wall_clock64();
    for (int r0 = 0; r0 < a.world && !failed; r0 += 8) {
        uint64_t v[8][PER];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr)
#pragma unroll
            for (int e = 0; e < PER; ++e) {
                const bool wanted = r0 + rr < a.world && lo + tid + e * EX_BLOCK < hi;
                v[rr][e] = !wanted ? 0 : r0 + rr == a.rank ? own[e] : EX_EMPTY;       // the own elements come from the registers
            }
        for (;;) {
            // only what is still missing is asked for again: a round of loads is a megabyte over xGMI per GPU
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int r = r0 + rr < a.world ? r0 + rr : a.world - 1;
                const uint64_t *slot = (const uint64_t *)(a.window[r] + slot_off);
#pragma unroll
                for (int e = 0; e < PER; ++e)
                    if (v[rr][e] == EX_EMPTY)
                        v[rr][e] = __hip_atomic_load(slot + (lo + tid + e * EX_BLOCK), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            bool all = true;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr)
#pragma unroll
                for (int e = 0; e < PER; ++e)
                    if (v[rr][e] == EX_EMPTY) all = false;
            if (all) break;
            if (wall_clock64() - t0 > a.wait_ticks) {
                failed = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int rr = 0; rr < 8; ++rr)
#pragma unroll
            for (int e = 0; e < PER; ++e)
                if (r0 + rr < a.world) acc[e] += __longlong_as_double((long long)v[rr][e]);
    }
    if (failed) {            // this workgroup's chunk stays as it was, others may already hold sums: the buffer is
                             // undefined after a failed call (the exchange is dead from here on)
        __hip_atomic_store(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        const int64_t i = lo + tid + e * EX_BLOCK;
        if (i < hi) a.buf[i] = acc[e];
    }
}

}  // namespace ba

struct ba_exchange {
    int device = 0, rank = 0, world = 1;
    int64_t capacity = 0;
    char *window[ba::EX_MAX_RANKS] = {};
    bool opened[ba::EX_MAX_RANKS] = {};
    uint32_t call = 0;
    int64_t n_hist[2] = {0, 0};                   // lengths of the last call and of the one before
    int *h_status = nullptr;                      // pinned, device-visible
    double wait_seconds = 2.0;
};

extern "C" {

int ba_exchange_create(int rank, int world, int64_t capacity_doubles, ba_exchange **out) {
    BA_REQUIRE(out, "ba_exchange_create: null output");
    *out = nullptr;
    BA_REQUIRE(world >= 1 && world <= ba::EX_MAX_RANKS && rank >= 0 && rank < world, "ba_exchange_create: rank / world out of range (at most 16 ranks)");
    BA_REQUIRE(capacity_doubles >= 1 && capacity_doubles <= (int64_t)ba::EX_MAX_BLOCKS * ba::EX_CHUNK,
               "ba_exchange_create: capacity out of range (1 .. 262144 doubles)");
    ba_exchange *x = new ba_exchange;
    x->rank = rank;
    x->world = world;
    x->capacity = capacity_doubles;
    hipError_t e = hipGetDevice(&x->device);
    const size_t bytes = ba::EX_SLOTS * sizeof(double) * (size_t)capacity_doubles;
    void *w = nullptr;
    if (e == hipSuccess) e = hipExtMallocWithFlags(&w, bytes, hipDeviceMallocUncached);
    if (e == hipSuccess) e = hipMemsetD32((hipDeviceptr_t)w, (int)ba::EX_EMPTY_HALF, bytes / sizeof(uint32_t));
    if (e == hipSuccess) e = hipHostMalloc((void **)&x->h_status, sizeof(int), hipHostMallocMapped);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        if (w) (void)hipFree(w);
        if (x->h_status) (void)hipHostFree(x->h_status);
        delete x;
        return ba::hip_fail(e, "ba_exchange_create", __FILE__, __LINE__);
    }
    *x->h_status = 0;
    x->window[rank] = (char *)w;
    *out = x;
    return BA_OK;
}

int ba_exchange_export(ba_exchange *x, void *h_handle64) {
    BA_REQUIRE(x && h_handle64, "ba_exchange_export: null argument");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    hipIpcMemHandle_t h;
    BA_HIP(hipIpcGetMemHandle(&h, x->window[x->rank]));
    memcpy(h_handle64, &h, sizeof h);
    return BA_OK;
}

int ba_exchange_import(ba_exchange *x, int rank, const void *h_handle64) {
    BA_REQUIRE(x && h_handle64, "ba_exchange_import: null argument");
    BA_REQUIRE(rank >= 0 && rank < x->world && rank != x->rank, "ba_exchange_import: rank out of range or own rank");
    BA_REQUIRE(!x->opened[rank], "ba_exchange_import: window of this rank already mapped");
    hipIpcMemHandle_t h;
    memcpy(&h, h_handle64, sizeof h);
    void *p = nullptr;
    BA_HIP(hipSetDevice(x->device));
    BA_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    // a mapping the runtime cannot read through must not reach a kernel: probe it with a copy first
    uint32_t probe = 0;
    hipError_t e = hipMemcpy(&probe, p, sizeof probe, hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        (void)hipIpcCloseMemHandle(p);
        return ba::hip_fail(e, "ba_exchange_import: probe of the mapped window", __FILE__, __LINE__);
    }
    x->window[rank] = (char *)p;
    x->opened[rank] = true;
    return BA_OK;
}

int ba_exchange_sum(ba_exchange *x, double *buf, int64_t n, void *stream) {
    BA_REQUIRE(x && buf, "ba_exchange_sum: null argument");
    BA_REQUIRE(n >= 1 && n <= x->capacity, "ba_exchange_sum: length beyond the capacity of the windows");
    for (int r = 0; r < x->world; ++r)
        if (!x->window[r]) {
            ba::set_error("ba_exchange_sum: window of rank " + std::to_string(r) + " not mapped");
            return BA_ERR_STATE;
        }
    if (*x->h_status != 0) {
        ba::set_error("ba_exchange_sum: an earlier call gave up waiting for another rank");
        return BA_ERR_STATE;
    }
    ba::ExchangeArgs a;
    for (int r = 0; r < ba::EX_MAX_RANKS; ++r) a.window[r] = x->window[r];
    a.buf = buf;
    a.n = n;
    a.n_old = x->n_hist[1];
    x->n_hist[1] = x->n_hist[0];
    x->n_hist[0] = n;
    a.capacity = x->capacity;
    a.rank = x->rank;
    a.world = x->world;
    a.call = ++x->call;
    a.wait_ticks = (long long)(x->wait_seconds * 1e8);
    a.status = x->h_status;
    const int64_t covered = n > a.n_old ? n : a.n_old;
    const int grid = (int)((covered + ba::EX_CHUNK - 1) / ba::EX_CHUNK);
    hipLaunchKernelGGL(ba::k_exchange_sum, dim3(grid), dim3(ba::EX_BLOCK), 0, (hipStream_t)stream, a);
    BA_HIP(hipGetLastError());
    return BA_OK;
}

int ba_exchange_status(ba_exchange *x) {
    if (!x) return BA_ERR_ARG;
    if (*x->h_status != 0) {
        ba::set_error("ba_exchange: a call gave up waiting for another rank; sums since then are not valid");
        return BA_ERR_STATE;
    }
    return BA_OK;
}

int ba_exchange_set_wait(ba_exchange *x, double seconds) {
    BA_REQUIRE(x && seconds > 0.0 && seconds <= 30.0, "ba_exchange_set_wait: 0 < seconds <= 30");
    x->wait_seconds = seconds;
    return BA_OK;
}

void ba_exchange_destroy(ba_exchange *x) {
    if (!x) return;
    (void)hipSetDevice(x->device);
    (void)hipDeviceSynchronize();
    for (int r = 0; r < x->world; ++r)
        if (r != x->rank && x->opened[r]) (void)hipIpcCloseMemHandle(x->window[r]);
    if (x->window[x->rank]) (void)hipFree(x->window[x->rank]);
    if (x->h_status) (void)hipHostFree(x->h_status);
    delete x;
}

}  // extern "C"
