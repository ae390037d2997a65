// Random-access floors on gfx950 for the patterns the clustering kernels are made of (development aid, not product):
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/microbench tools/microbench.hip && gpurun_out/microbench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// one random dword atomicMax per item
__global__ void k_atomic4(uint32_t *t, uint32_t n_slots, uint32_t n, uint32_t seed) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t s = mix(i ^ seed) % n_slots;
        atomicMax(&t[s], i);
    }
}
// read, then atomicMax only if larger (the first_open pattern), record of `stride` dwords
__global__ void k_rmw(uint32_t *t, uint32_t n_rec, uint32_t stride, uint32_t n, uint32_t seed, uint32_t *sink) {
    uint32_t acc = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t h = mix(i ^ seed);
        uint32_t *slot = t + (size_t)(h % n_rec) * stride + ((h >> 27) % stride);
        const uint32_t v = *slot;
        if (v < (i | 1u)) atomicMax(slot, i | 1u); else acc += v;
    }
    if (acc == 0x12345u) *sink = acc;
}
// random reads of `bytes` (16/32/64/128) per item, as uint4 loads
template <int NV>
__global__ void k_read(const uint4 *t, uint32_t n_rec, uint32_t n, uint32_t seed, uint32_t *sink) {
    uint32_t acc = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint4 *p = t + (size_t)(mix(i ^ seed) % n_rec) * NV;
#pragma unroll
        for (int j = 0; j < NV; ++j) { const uint4 v = p[j]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345u) *sink = acc;
}
// random dword reads (bit-map probes)
__global__ void k_read4(const uint32_t *t, uint32_t n_slots, uint32_t n, uint32_t seed, uint32_t *sink) {
    uint32_t acc = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc += t[mix(i ^ seed) % n_slots];
    if (acc == 0x12345u) *sink = acc;
}
// streaming read of 6 bytes per word (u32 code + u16 mult) -- the word lists
__global__ void k_stream(const uint32_t *c, const uint16_t *m, size_t n, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += c[i] + m[i];
    if (acc == 0x12345u) *sink = acc;
}

int main() {
    const size_t big = 600ull << 20;
    uint32_t *d = nullptr, *sink = nullptr;
    CK(hipMalloc(&d, big));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(d, 0, big));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint32_t n = 32u << 20;
    auto timeit = [&](const char *what, auto launch, double bytes_per_item) {
        launch(1u); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (uint32_t r = 0; r < 3; ++r) launch(r + 2);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
        printf("%-44s %7.3f ms  %7.2f G items/s  %8.1f GB/s\n", what, ms, n / ms / 1e6, n * bytes_per_item / ms / 1e6);
    };
    const int G = 4096, B = 256;
    for (size_t mb : {1, 4, 16, 64, 512}) {
        char nm[96]; snprintf(nm, sizeof nm, "atomicMax dword, table %zu MB", mb);
        timeit(nm, [&](uint32_t s) { k_atomic4<<<G, B>>>(d, (uint32_t)(mb << 18), n, s); }, 4);
    }
    for (uint32_t stride : {1u, 8u, 16u, 32u}) {
        char nm[96]; snprintf(nm, sizeof nm, "read+atomicMax, 4 M records x %u B", stride * 4);
        timeit(nm, [&](uint32_t s) { k_rmw<<<G, B>>>(d, 4084101u, stride, n, s, sink); }, stride * 4);
    }
    timeit("random 16 B reads, 4 M records (64 MB)", [&](uint32_t s) { k_read<1><<<G, B>>>((uint4 *)d, 4084101u, n, s, sink); }, 16);
    timeit("random 32 B reads, 4 M records (130 MB)", [&](uint32_t s) { k_read<2><<<G, B>>>((uint4 *)d, 4084101u, n, s, sink); }, 32);
    timeit("random 64 B reads, 4 M records (261 MB)", [&](uint32_t s) { k_read<4><<<G, B>>>((uint4 *)d, 4084101u, n, s, sink); }, 64);
    timeit("random 128 B reads, 4 M records (523 MB)", [&](uint32_t s) { k_read<8><<<G, B>>>((uint4 *)d, 4084101u, n, s, sink); }, 128);
    for (size_t kb : {512, 4096, 16384, 65536}) {
        char nm[96]; snprintf(nm, sizeof nm, "random dword reads, table %zu KB", kb);
        timeit(nm, [&](uint32_t s) { k_read4<<<G, B>>>(d, (uint32_t)(kb << 8), n, s, sink); }, 4);
    }
    timeit("streaming u32 + u16 per item", [&](uint32_t) { k_stream<<<G, B>>>(d, (uint16_t *)(d + (64u << 20)), n, sink); }, 6);
    return 0;
}
