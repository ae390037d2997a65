// Internal helpers shared by the libpgx translation units (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include "../../include/pgx.h"

#include <string>
#include <utility>
#include <algorithm>
#include <cstdlib>
#include <vector>

// Optional per-kernel timing (pgx_profile_*): HIP events recorded on the launch stream
// around each kernel, resolved when the caller reads the slots.
struct ProfSlot {
    std::string name;
    double total_ms = 0.0;
    uint64_t launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

struct pgx_ctx {
    int device_id;
    hipStream_t stream;  // used by the host-pointer entry points
    hipStream_t stream2; // side stream: next sweep's index + table pass overlap the current sweep
    hipEvent_t ev_main = nullptr, ev_side[2] = {nullptr, nullptr};
    hipDeviceProp_t prop;
    bool profiling = false;
    std::vector<ProfSlot> prof;
    std::vector<hipEvent_t> event_pool;
    // grow-only device workspace, reused by successive calls on this context (slot -> buffer)
    std::vector<std::pair<void *, size_t>> arena;
    // grow-only page-locked host staging buffers (slot -> buffer), same idea
    std::vector<std::pair<void *, size_t>> host_arena;
    // grow-only page-locked host scratch (slot -> buffer): the per-sequence arrays of a clustering call
    // (tens of MB) are neither allocated, faulted in nor freed again by every call, and their copies to
    // and from the device are asynchronous (from pageable memory every hipMemcpyAsync blocked the host)
    std::vector<std::pair<void *, size_t>> host_scratch;
    std::vector<hipEvent_t> stage_events;   // pgx_staged_h2d: one per staging buffer (its last DMA)
    // the gene x genome bitmap a pipeline left on the device (pgx_bitmap_from_clusters): its token, shape, buffer
    uint64_t resident_token = 0, resident_next = 1;
    uint32_t resident_genes = 0, resident_genomes = 0;
    // the library's own RCCL communicator (pgx_rccl_comm_create): the record-sharded exchange without a callback
    void *comm = nullptr;
    int comm_rank = 0, comm_world = 0;
};

// Array of n elements of T in the context's host scratch slot `slot` (uninitialised unless `fill` is given).
// A view: the memory belongs to the context and outlives the call. Check ok() after construction.
template <typename T>
struct HostVec {
    T *p = nullptr;
    size_t n = 0;
    HostVec(pgx_ctx *ctx, int slot, size_t count) { bind(ctx, slot, count); }
    HostVec(pgx_ctx *ctx, int slot, size_t count, T fill) {
        bind(ctx, slot, count);
        if (p) std::fill(p, p + n, fill);
    }
    bool ok() const { return p != nullptr || n == 0; }
    T *data() { return p; }
    const T *data() const { return p; }
    size_t size() const { return n; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
    T *begin() { return p; }
    T *end() { return p + n; }

private:
    void bind(pgx_ctx *ctx, int slot, size_t count) {
        if ((int)ctx->host_scratch.size() <= slot) ctx->host_scratch.resize((size_t)slot + 1, {nullptr, 0});
        auto &a = ctx->host_scratch[(size_t)slot];
        const size_t bytes = count * sizeof(T) + 64;
        if (a.second < bytes) {
            if (a.first) (void)hipHostFree(a.first);
            a = {nullptr, 0};
            const size_t want = bytes + bytes / 4;
            void *q = nullptr;
            if (hipHostMalloc(&q, want, hipHostMallocDefault) == hipSuccess) a = {q, want};
            else (void)hipGetLastError();
        }
        p = static_cast<T *>(a.first);
        n = p ? count : 0;
        if (!p && count) n = count;  // ok() reports the failure
    }
};

// RAII bracket: records start/stop events on `stream` when profiling is enabled.
struct ProfScope {
    pgx_ctx *ctx;
    hipStream_t stream;
    int slot = -1;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ProfScope(pgx_ctx *c, const char *name, hipStream_t s);
    ~ProfScope();
};

void pgx_set_error(const char *fmt, ...);

#define PGX_HIP(call)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            pgx_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #call,               \
                          hipGetErrorString(e_));                                          \
            return PGX_ERR_HIP;                                                            \
        }                                                                                  \
    } while (0)

#define PGX_REQUIRE(cond, msg)                                                             \
    do {                                                                                   \
        if (!(cond)) {                                                                     \
            pgx_set_error("%s: %s", __func__, msg);                                        \
            return PGX_ERR_INVALID;                                                        \
        }                                                                                  \
    } while (0)

// Device buffer. Stand-alone (freed on scope exit) or, when bound to a context slot, a view
// of that slot of the context's grow-only workspace: repeated calls then neither allocate
// nor free (hipFree of GB-sized buffers stalls the queue for milliseconds).
struct DevBuf {
    void *p = nullptr;
    pgx_ctx *ctx = nullptr;
    int slot = -1;
    ~DevBuf() {
        if (p && !ctx) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) {
        if (!bytes) bytes = 16;
        if (!ctx) return hipMalloc(&p, bytes);
        if ((int)ctx->arena.size() <= slot) ctx->arena.resize((size_t)slot + 1, {nullptr, 0});
        auto &a = ctx->arena[(size_t)slot];
        if (!a.first || a.second < bytes) {
            if (a.first) (void)hipFree(a.first);
            a = {nullptr, 0};
            const size_t cap = bytes + bytes / 4 + 256;
            hipError_t e = hipMalloc(&a.first, cap);
            if (e != hipSuccess) return e;
            a.second = cap;
        }
        p = a.first;
        return hipSuccess;
    }
    template <typename T>
    T *as() {
        return static_cast<T *>(p);
    }
};

static inline uint32_t ceil_div_u32(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// Host-to-device copy of a large PAGEABLE buffer, enqueued on `stream`: a few threads copy 4 MB chunks into
// page-locked staging buffers of the context (two per thread) and enqueue the DMA of each chunk as soon as it is
// staged, so that the CPU copies of the threads and the DMA engine overlap. hipMemcpyAsync from pageable memory
// stages through one thread: 13.7 GB/s measured for the 370 MB of the benchmark's sequences. Small buffers
// take the plain call. Returns when every chunk has been ENQUEUED (the source may be reused; the copies complete in
// stream order).
int pgx_staged_h2d(pgx_ctx *ctx, void *dst, const void *src, size_t bytes, hipStream_t stream);

// all-gather of `count` uint64 per process with the context's RCCL communicator, enqueued on `stream` (rccl_exchange.hip)
int pgx_rccl_all_gather_u64(pgx_ctx *ctx, const void *send, void *recv, size_t count, hipStream_t stream);
