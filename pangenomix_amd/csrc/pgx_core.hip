// libpgx: context, error reporting, device info.
#include <atomic>
#include <thread>

#include "pgx_internal.h"

static thread_local char g_err[512] = "";

void pgx_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static hipEvent_t prof_event(pgx_ctx *ctx) {
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

ProfScope::ProfScope(pgx_ctx *c, const char *name, hipStream_t s) : ctx(c), stream(s) {
    if (!ctx || !ctx->profiling) return;
    for (size_t i = 0; i < ctx->prof.size(); ++i)
        if (ctx->prof[i].name == name) slot = (int)i;
    if (slot < 0) {
        ctx->prof.emplace_back();
        ctx->prof.back().name = name;
        slot = (int)ctx->prof.size() - 1;
    }
    e0 = prof_event(ctx);
    e1 = prof_event(ctx);
    if (e0) (void)hipEventRecord(e0, stream);
}

ProfScope::~ProfScope() {
    if (slot < 0) return;
    if (e1) (void)hipEventRecord(e1, stream);
    ctx->prof[slot].pending.emplace_back(e0, e1);
    ctx->prof[slot].launches++;
}

static void prof_resolve(pgx_ctx *ctx) {
    for (auto &sl : ctx->prof) {
        for (auto &pr : sl.pending) {
            float ms = 0.f;
            if (pr.first && pr.second && hipEventSynchronize(pr.second) == hipSuccess &&
                hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess)
                sl.total_ms += ms;
            if (pr.first) ctx->event_pool.push_back(pr.first);
            if (pr.second) ctx->event_pool.push_back(pr.second);
        }
        sl.pending.clear();
    }
}

int pgx_staged_h2d(pgx_ctx *ctx, void *dst, const void *src, size_t bytes, hipStream_t stream) {
    constexpr size_t kChunk = 4u << 20;
    constexpr int kStageSlot = 30;           // host scratch slot of the staging buffers
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned T = (unsigned)std::min<size_t>(std::min(4u, std::max(1u, hw / 2)), bytes / (2 * kChunk));
    if (T < 2) { PGX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream)); return PGX_OK; }
    HostVec<uint8_t> stage(ctx, kStageSlot, (size_t)T * 2 * kChunk);
    if (!stage.ok() || !stage.data()) { PGX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream)); return PGX_OK; }
    while (ctx->stage_events.size() < 2 * (size_t)T) {
        hipEvent_t e = nullptr;
        PGX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->stage_events.push_back(e);
    }
    const size_t n_chunks = (bytes + kChunk - 1) / kChunk;
    std::atomic<int> failed{0};
    auto worker = [&](unsigned t) {
        if (hipSetDevice(ctx->device_id) != hipSuccess) { failed = 1; return; }
        unsigned k = 0;
        for (size_t c = t; c < n_chunks && !failed.load(std::memory_order_relaxed); c += T, ++k) {
            const unsigned b = 2 * t + (k & 1u);
            uint8_t *buf = stage.data() + (size_t)b * kChunk;
            // (the buffer's last DMA -- of this call or the one before -- must have read it; a never-recorded event is "done")
            if (hipEventSynchronize(ctx->stage_events[b]) != hipSuccess) { failed = 1; return; }
            const size_t at = c * kChunk, len = std::min(kChunk, bytes - at);
            memcpy(buf, static_cast<const uint8_t *>(src) + at, len);
            if (hipMemcpyAsync(static_cast<uint8_t *>(dst) + at, buf, len, hipMemcpyHostToDevice, stream) != hipSuccess ||
                hipEventRecord(ctx->stage_events[b], stream) != hipSuccess) { failed = 1; return; }
        }
    };
    std::vector<std::thread> pool;
    try {
        for (unsigned t = 1; t < T; ++t) pool.emplace_back(worker, t);
    } catch (...) { failed = 1; }
    if (!failed) worker(0);
    for (auto &th : pool) th.join();
    if (failed) { (void)hipGetLastError(); pgx_set_error("pgx_staged_h2d: a staged copy failed"); return PGX_ERR_HIP; }
    return PGX_OK;
}

extern "C" {

int pgx_profile_enable(pgx_ctx *ctx, int on) {
    PGX_REQUIRE(ctx, "NULL context");
    ctx->profiling = on != 0;
    return PGX_OK;
}

int pgx_profile_reset(pgx_ctx *ctx) {
    PGX_REQUIRE(ctx, "NULL context");
    prof_resolve(ctx);
    ctx->prof.clear();
    return PGX_OK;
}

int pgx_profile_count(pgx_ctx *ctx) {
    if (!ctx) return 0;
    return (int)ctx->prof.size();
}

int pgx_profile_read(pgx_ctx *ctx, int slot, char *name, size_t name_bytes, double *total_ms,
                     uint64_t *launches) {
    PGX_REQUIRE(ctx, "NULL context");
    PGX_REQUIRE(slot >= 0 && slot < (int)ctx->prof.size(), "slot out of range");
    prof_resolve(ctx);
    const ProfSlot &sl = ctx->prof[slot];
    if (name && name_bytes) snprintf(name, name_bytes, "%s", sl.name.c_str());
    if (total_ms) *total_ms = sl.total_ms;
    if (launches) *launches = sl.launches;
    return PGX_OK;
}

int pgx_version(void) { return PGX_VERSION; }

const char *pgx_last_error(void) { return g_err; }

int pgx_ctx_create(int device_id, pgx_ctx **out) {
    PGX_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        pgx_set_error("pgx_ctx_create: no usable HIP device (%s); libpgx has no CPU fallback",
                      e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        return PGX_ERR_NO_DEVICE;
    }
    PGX_REQUIRE(device_id >= 0 && device_id < n, "device_id out of range");
    PGX_HIP(hipSetDevice(device_id));
    pgx_ctx *ctx = new (std::nothrow) pgx_ctx();
    if (!ctx) {
        pgx_set_error("pgx_ctx_create: out of host memory");
        return PGX_ERR_NOMEM;
    }
    ctx->device_id = device_id;
    e = hipGetDeviceProperties(&ctx->prop, device_id);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e == hipSuccess) {  // side stream at the lowest priority: it only runs ahead into idle CUs
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        e = hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, least);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_main, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_side[0], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_side[1], hipEventDisableTiming);
    if (e != hipSuccess) {
        pgx_set_error("pgx_ctx_create: %s", hipGetErrorString(e));
        delete ctx;
        return PGX_ERR_HIP;
    }
    *out = ctx;
    return PGX_OK;
}

int pgx_ctx_create_on(const int *device_ids, int n_devices, pgx_ctx **out) {
    PGX_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    PGX_REQUIRE(device_ids != nullptr && n_devices >= 1, "no device given");
    if (n_devices != 1) {
        pgx_set_error("pgx_ctx_create_on: %d devices asked for; a context owns ONE device -- the multi-GPU mode is one "
                      "process per GPU (pgx_cluster_params.shard_index / shard_count), each with its own context", n_devices);
        return PGX_ERR_INVALID;
    }
    return pgx_ctx_create(device_ids[0], out);
}

void pgx_ctx_destroy(pgx_ctx *ctx) {
    if (!ctx) return;
    (void)pgx_rccl_comm_destroy(ctx);
    (void)hipSetDevice(ctx->device_id);
    // both streams drained before anything they may still read or write is released
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(ctx->stream2);
    prof_resolve(ctx);
    for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->stage_events) (void)hipEventDestroy(e);
    for (auto &a : ctx->arena)
        if (a.first) (void)hipFree(a.first);
    for (auto &a : ctx->host_arena)
        if (a.first) (void)hipHostFree(a.first);
    for (auto &a : ctx->host_scratch)
        if (a.first) (void)hipHostFree(a.first);
    (void)hipStreamDestroy(ctx->stream2);
    (void)hipStreamDestroy(ctx->stream);
    if (ctx->ev_main) (void)hipEventDestroy(ctx->ev_main);
    for (hipEvent_t e : ctx->ev_side) if (e) (void)hipEventDestroy(e);
    delete ctx;
}

int pgx_device_info(pgx_ctx *ctx, pgx_device_info_t *out) {
    PGX_REQUIRE(ctx && out, "NULL argument");
    memset(out, 0, sizeof(*out));
    snprintf(out->name, sizeof(out->name), "%s", ctx->prop.name);
    snprintf(out->arch, sizeof(out->arch), "%s", ctx->prop.gcnArchName);
    out->device_id = ctx->device_id;
    out->compute_units = ctx->prop.multiProcessorCount;
    out->wavefront_size = ctx->prop.warpSize;
    out->lds_bytes_per_block = (int32_t)ctx->prop.sharedMemPerBlock;
    out->hbm_bytes = ctx->prop.totalGlobalMem;
    out->clock_khz = ctx->prop.clockRate;
    return PGX_OK;
}

}  // extern "C"
