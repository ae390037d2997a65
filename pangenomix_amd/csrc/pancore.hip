// K3: presence/absence bitmap + pan/core rarefaction curves on gfx950.
//
// Replaces the Python double loop of pangenome_analysis.py:81-90
//     gene_incidence += gene_data[col,:]
//     pan[i,j]  = (gene_incidence > 0).sum()
//     core[i,j] = (gene_incidence == j+1).sum()
// with, per iteration, a running OR / running AND over bit-packed genome rows taken in
// permuted order and a popcount per step (exact because the matrix is 0/1).
//
// Data layout in HBM
//   bits[genome][stride] uint64, gene g = bit (g&63) of word (g>>6); stride is a multiple
//   of 16 words (128 B). A row is cut into 8, 4 or 2 equal *stripes* of Ls 16-byte lanes (make_geom);
//   stripe x is only ever read by workgroups with blockIdx % stripes == x, i.e. (round-robin
//   dispatch) by one, two or four XCDs, so an XCD's private 4 MiB L2 holds one stripe of every genome
//   and every re-read of a row (n_iter times) is an L2 hit instead of an Infinity-Cache / HBM access.
//
// Work decomposition
//   item (iter, stripe, v): one wave walks all S steps of iteration `iter` over the lanes
//   [v*Lw, (v+1)*Lw) of its stripe, one dwordx4 (128 genes) per lane per step, loads
//   issued PC_UNROLL steps ahead. Per step the lane counts are packed (pan | core<<16),
//   summed over the wave with 6 DPP adds, and the wave total is parked in lane (j & 63);
//   every 64 steps the wave stores 64 totals with one coalesced store into
//   partial[stripe*wps+v][iter][j]. A second, tiny kernel sums the 8*wps partials.
//   No atomics, bit-reproducible.
//
// Roofline: the aggregate L2 (the matrix is L2-resident by design). Algorithmic bytes per iteration =
// S*ceil(G/64)*8 + 2*S*4 (+S*4 for the permutation), SURVEY §8d.
// Tried in round 2 and not kept: staging 256-byte slices of all rows in LDS (100 KB per workgroup, 16
// iterations per workgroup, half-wave per iteration, LDS accumulators): 0.78 ms against this kernel's
// 0.68 ms -- one workgroup per CU leaves two waves per SIMD to hide a chain of two LDS reads per step
// (software-pipelined two deep: no change). DESIGN.md section 6.
#include <algorithm>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "pgx_internal.h"

namespace {

constexpr int PC_XCDS = 8;     // XCDs per MI355X: a row is cut into 8, 4 or 2 stripes, each read by 1, 2 or 4 XCDs only
constexpr int PC_WAVES = 4;    // waves per workgroup
constexpr int PC_UNROLL = 8;   // rows in flight per wave

struct PanCoreGeom {
    uint32_t words;     // ceil(G/64)
    uint32_t stride;    // padded words per genome (multiple of 16)
    uint32_t Ls;        // 16-byte lanes per stripe
    uint32_t wps;       // waves per stripe
    uint32_t Lw;        // lanes per wave
    uint32_t stripes;   // 8, 4 or 2
    uint32_t partials;  // stripes * wps
};

// The kernel is bound by vector instructions per wave and step, whatever the number of live lanes: the stripe count is
// the one that fills the waves best -- 150,000 genes: 8 stripes = 147 lanes = 3 waves of 49 (77 %), 4 stripes = 294 lanes
// = 5 waves of 59 (92 %: 0.57 -> 0.49 ms per 1000 iterations) -- as long as an XCD's share of the matrix (one stripe of
// every genome) stays well inside its 4 MiB L2.
PanCoreGeom make_geom(uint32_t n_genes, uint32_t n_genomes) {
    PanCoreGeom g;
    g.words = (n_genes + 63) / 64;
    g.stride = pgx_bitmap_stride_words(n_genes);
    double best_fill = -1.0;
    for (uint32_t stripes = PC_XCDS; stripes >= 2; stripes /= 2) {
        const uint32_t Ls = g.stride / (2 * stripes);                  // (stride is a multiple of 16 words = 8 lanes)
        if (stripes < PC_XCDS && (uint64_t)n_genomes * Ls * 16 > (3ull << 20)) break;
        const uint32_t w0 = (Ls + 63) / 64;
        const uint32_t Lw = (Ls + w0 - 1) / w0;        // balanced lanes per wave, <= 64
        const uint32_t wps = (Ls + Lw - 1) / Lw;       // so that (wps-1)*Lw < Ls: no empty wave
        const double fill = (double)Ls / (64.0 * wps);
        if (fill > best_fill + 0.04) { best_fill = fill; g.stripes = stripes; g.Ls = Ls; g.Lw = Lw; g.wps = wps; }
    }
    g.partials = g.stripes * g.wps;
    return g;
}

__device__ __forceinline__ uint32_t popc128(const uint4 &v) {
    return __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
}

__global__ __launch_bounds__(PC_WAVES * 64) void pan_core_sweep_kernel(
    const uint4 *__restrict__ bits, uint32_t stride_bytes, const int32_t *__restrict__ perms,
    uint32_t n_iter, uint32_t S, uint32_t Ls, uint32_t wps, uint32_t Lw, uint32_t stripes,
    uint32_t *__restrict__ partial) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t stripe = blockIdx.x & (stripes - 1);      // (block b runs on XCD b % 8: stripe x stays with XCDs x, x + stripes, ...)
    const uint32_t item = __builtin_amdgcn_readfirstlane(
        (blockIdx.x / stripes) * PC_WAVES + (threadIdx.x >> 6));
    if (item >= n_iter * wps) return;  // wave-uniform
    const uint32_t iter = item / wps;
    const uint32_t v = item - iter * wps;
    const uint32_t lane0 = v * Lw;                // < Ls by construction of the geometry
    const uint32_t nl = min(Lw, Ls - lane0);      // >= 1
    const bool active = lane < nl;
    // idle lanes re-read the wave's last live lane (always in bounds) and are masked out
    // of the counts, so the load itself is unconditional and coalesced.
    const uint32_t lane_off = (stripe * Ls + lane0 + (active ? lane : nl - 1u)) * 16u;
    const char *bytes = reinterpret_cast<const char *>(bits);
    const int32_t *prow = perms + (size_t)iter * S;
    uint32_t *out = partial + ((size_t)(stripe * wps + v) * n_iter + iter) * S;

    uint4 acc_or = make_uint4(0u, 0u, 0u, 0u);
    uint4 acc_and = make_uint4(~0u, ~0u, ~0u, ~0u);
    // The wave totals of 64 steps are formed TRANSPOSED: two steps' lane counts are merged into one register
    // (lanes with the level's bit clear keep step a and add the partner lane's step-a count, the others keep
    // step b), so that after levels 1..6 lane l holds the wave total of step l of the block -- 3 instructions
    // per merge, 63 merges per 64 steps, instead of a 6-step DPP reduction + readlane + select per step
    // (35 -> ~29 vector instructions per step; the kernel is VALU-bound).
    auto merge = [&](uint32_t a, uint32_t b, uint32_t bit, auto partner) -> uint32_t {
        const bool hi = (lane & bit) != 0u;
        const uint32_t keep = hi ? b : a, give = hi ? a : b;
        return keep + partner(give);
    };
    auto xor1 = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true); };   // quad_perm [1,0,3,2]
    auto xor2 = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true); };   // quad_perm [2,3,0,1]
    auto xor4 = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (4 << 10) | 0x1F); };
    auto xor8 = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true); };  // row_ror:8
    auto xor16 = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (16 << 10) | 0x1F); };
    auto xor32 = [](uint32_t v) { return (uint32_t)__shfl_xor((int)v, 32); };
    uint32_t p4 = 0, p5 = 0, p6 = 0;      // pending merges of the levels across chunks of 8 steps

    // The permutation is fetched 64 steps at a time with one coalesced vector load and
    // handed to the scalar unit with v_readlane (row base = SGPR pair, lane offset = VGPR).
    auto perm_block = [&](uint32_t b) -> int32_t {
        const uint32_t j = b * 64u + lane;
        return prow[j < S ? j : S - 1u];
    };
    auto row_load = [&](int32_t pv, uint32_t l) -> uint4 {
        const uint32_t r = (uint32_t)__builtin_amdgcn_readlane(pv, l);
        const char *rowp = bytes + (size_t)r * stride_bytes;  // scalar
        return *reinterpret_cast<const uint4 *>(rowp + lane_off);
    };

    const uint32_t n_chunks = ((S + 63u) / 64u) * (64u / PC_UNROLL);
    int32_t pv_cur = perm_block(0), pv_nxt = perm_block(1);
    uint4 cur[PC_UNROLL], nxt[PC_UNROLL];
#pragma unroll
    for (int u = 0; u < PC_UNROLL; ++u) cur[u] = row_load(pv_cur, u);

    for (uint32_t c = 0; c < n_chunks; ++c) {
        // issue the next chunk's row loads before folding this one
        const uint32_t cn = c + 1u;
        const uint32_t sub = (cn & (64u / PC_UNROLL - 1u)) * PC_UNROLL;
        if (sub == 0u) {  // wave-uniform: next chunk starts a new block of 64 steps
            pv_cur = pv_nxt;
            pv_nxt = perm_block(cn / (64u / PC_UNROLL) + 1u);
        }
#pragma unroll
        for (int u = 0; u < PC_UNROLL; ++u) nxt[u] = row_load(pv_cur, sub + u);

        uint32_t cnt[PC_UNROLL];                 // steps >= S only reach lanes that are never stored
#pragma unroll
        for (int u = 0; u < PC_UNROLL; ++u) {
            const uint4 row = cur[u];
            acc_or.x |= row.x; acc_or.y |= row.y; acc_or.z |= row.z; acc_or.w |= row.w;
            acc_and.x &= row.x; acc_and.y &= row.y; acc_and.z &= row.z; acc_and.w &= row.w;
            const uint32_t packed = popc128(acc_or) | (popc128(acc_and) << 16);
            cnt[u] = active ? packed : 0u;
        }
        static_assert(PC_UNROLL == 8, "three merge levels inside a chunk");
        const uint32_t m3 = merge(merge(merge(cnt[0], cnt[1], 1u, xor1), merge(cnt[2], cnt[3], 1u, xor1), 2u, xor2),
                                  merge(merge(cnt[4], cnt[5], 1u, xor1), merge(cnt[6], cnt[7], 1u, xor1), 2u, xor2), 4u, xor4);
        // lane l now holds, summed over its group of 8 lanes, the count of step 8 c + (l & 7)
        if ((c & 1u) == 0u) p4 = m3;
        else {
            const uint32_t m4 = merge(p4, m3, 8u, xor8);
            if ((c & 2u) == 0u) p5 = m4;
            else {
                const uint32_t m5 = merge(p5, m4, 16u, xor16);
                if ((c & 4u) == 0u) p6 = m5;
                else {   // 64 totals, lane l = step l of the block: one coalesced store
                    const uint32_t m6 = merge(p6, m5, 32u, xor32);
                    const uint32_t jb = (c / (64u / PC_UNROLL)) * 64u;
                    if (jb + lane < S) out[jb + lane] = m6;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < PC_UNROLL; ++u) cur[u] = nxt[u];
    }
}


__global__ __launch_bounds__(256) void pan_core_reduce_kernel(const uint32_t *__restrict__ partial,
                                                             uint32_t n_partials, size_t n_out,
                                                             int32_t *__restrict__ out_pan,
                                                             int32_t *__restrict__ out_core) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_out;
         e += (size_t)gridDim.x * blockDim.x) {
        uint32_t pan = 0, core = 0;
        for (uint32_t p = 0; p < n_partials; ++p) {
            const uint32_t x = partial[(size_t)p * n_out + e];
            pan += x & 0xFFFFu;
            core += x >> 16;
        }
        out_pan[e] = (int32_t)pan;
        out_core[e] = (int32_t)core;
    }
}

// The gene x genome bitmap straight from a clustering result: record r of the genome files is an instance of gene
// cluster_of_group[group_of_record[r]] in genome genome_of_file[file_of_record[r]] (records without a sequence or
// outside every cluster set nothing). What build_genetic_feature_tables (pangenome.py:563-680) derives per record.
__global__ __launch_bounds__(256) void bitmap_from_clusters_kernel(const int32_t *__restrict__ cluster_of_group,
                                                                  uint32_t n_groups,
                                                                  const int32_t *__restrict__ group_of_record,
                                                                  const uint32_t *__restrict__ file_of_record,
                                                                  uint64_t n_records,
                                                                  const int32_t *__restrict__ genome_of_file,
                                                                  uint32_t n_files, uint32_t n_genes, uint32_t n_genomes,
                                                                  uint32_t stride, unsigned long long *__restrict__ bits,
                                                                  unsigned long long *__restrict__ counters) {
    uint32_t bad = 0;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_records; r += (uint64_t)gridDim.x * blockDim.x) {
        const int32_t g = group_of_record[r];
        if (g < 0) continue;
        const uint32_t f = file_of_record[r];
        if ((uint32_t)g >= n_groups || f >= n_files) { ++bad; continue; }
        const int32_t gene = cluster_of_group[g], genome = genome_of_file[f];
        if (gene < 0) continue;
        if ((uint32_t)gene >= n_genes || (uint32_t)genome >= n_genomes) { ++bad; continue; }
        const unsigned long long bit = 1ull << ((uint32_t)gene & 63u);
        unsigned long long *w = &bits[(size_t)genome * stride + ((uint32_t)gene >> 6)];
        if (!(*w & bit)) atomicOr(w, bit);
    }
    for (int d = 32; d > 0; d >>= 1) bad += __shfl_xor(bad, d);
    if ((threadIdx.x & 63u) == 0 && bad) atomicAdd(&counters[1], (unsigned long long)bad);
}

// The same sum written as the table estimate_pan_core_size() returns: float64 [n_iter][2 S], pan curves in columns
// 0..S-1, core curves in S..2S-1 (pangenome_analysis.py:93-97) -- the int -> float conversion and the side-by-side
// layout cost the host 1-2 ms per call (np.hstack + astype) and nothing here.
__global__ __launch_bounds__(256) void pan_core_reduce_table_kernel(const uint32_t *__restrict__ partial,
                                                                   uint32_t n_partials, uint32_t n_iter, uint32_t S,
                                                                   double *__restrict__ table) {
    const size_t n_out = (size_t)n_iter * S;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_out; e += (size_t)gridDim.x * blockDim.x) {
        uint32_t pan = 0, core = 0;
        for (uint32_t p = 0; p < n_partials; ++p) {
            const uint32_t x = partial[(size_t)p * n_out + e];
            pan += x & 0xFFFFu;
            core += x >> 16;
        }
        const size_t it = e / S, j = e - it * S;
        table[it * 2 * S + j] = (double)pan;
        table[it * 2 * S + S + j] = (double)core;
    }
}

// counters[0] += records whose bit was already set (duplicate coordinates), counters[1] += records
// with a row or genome index out of range (never written). atomicOr returns the word before the
// update, so a set bit there is a duplicate: the check the reference-side table needs (a 0/1 matrix
// without duplicate coordinates) costs nothing beside the bitmap build itself.
__global__ __launch_bounds__(256) void presence_bitmap_kernel(const int32_t *__restrict__ rows,
                                                             const int32_t *__restrict__ genomes,
                                                             uint64_t n, uint32_t n_rows,
                                                             uint32_t n_genomes, uint32_t stride,
                                                             unsigned long long *__restrict__ bits,
                                                             unsigned long long *__restrict__ counters) {
    uint32_t dup = 0, bad = 0;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n;
         k += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = (uint32_t)rows[k], g = (uint32_t)genomes[k];
        if (r >= n_rows || g >= n_genomes) { ++bad; continue; }  // never write out of bounds
        const unsigned long long bit = 1ull << (r & 63u);
        dup += (atomicOr(&bits[(size_t)g * stride + (r >> 6)], bit) & bit) != 0ull;
    }
    if (counters) {
        for (int d = 32; d > 0; d >>= 1) { dup += __shfl_xor(dup, d); bad += __shfl_xor(bad, d); }
        if ((threadIdx.x & 63u) == 0) {
            if (dup) atomicAdd(&counters[0], (unsigned long long)dup);
            if (bad) atomicAdd(&counters[1], (unsigned long long)bad);
        }
    }
}

// Row occupancy of the bitmap: in how many genomes is gene (or allele) r present? One thread per row; the
// 64 rows of a wave share their word, so every genome costs the wave one load. What the reference's
// downstream consumers compute from the .npz triples with a pandas groupby (core_genome.py:127-155,
// allele_identification.py:129-157) and then threshold (core_genome.py:107-124).
__global__ __launch_bounds__(256) void row_counts_kernel(const unsigned long long *__restrict__ bits, uint32_t stride,
                                                        uint32_t n_genomes, uint32_t n_rows,
                                                        int32_t *__restrict__ counts) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const uint32_t w = r >> 6, b = r & 63u;
    int32_t c = 0;
    for (uint32_t s = 0; s < n_genomes; ++s) c += (int32_t)((bits[(size_t)s * stride + w] >> b) & 1ull);
    counts[r] = c;
}

// K3 device buffers of the host-pointer entry points live in the context's grow-only workspace
// (slots after the clustering's), so repeated calls neither allocate nor free.
enum { PC_SLOT_ROWS = 80, PC_SLOT_GENOMES, PC_SLOT_BITS, PC_SLOT_PERMS, PC_SLOT_PAN, PC_SLOT_CORE, PC_SLOT_WS, PC_SLOT_CNT, PC_SLOT_COUNTS, PC_SLOT_TABLE, PC_SLOT_RESIDENT, PC_SLOT_RES_A, PC_SLOT_RES_B, PC_SLOT_RES_C, PC_SLOT_RES_D };
struct PcBuf : DevBuf {
    PcBuf(pgx_ctx *c, int s) { ctx = c; slot = s; }
};

}  // namespace

extern "C" {

uint32_t pgx_bitmap_stride_words(uint32_t n_genes) {
    const uint32_t words = (n_genes + 63) / 64;
    const uint32_t s = (words + 15u) & ~15u;
    return s ? s : 16u;
}

size_t pgx_pan_core_workspace_bytes(uint32_t n_genes, uint32_t n_genomes, uint32_t n_iter) {
    const PanCoreGeom g = make_geom(n_genes, n_genomes);
    return (size_t)g.partials * n_iter * n_genomes * sizeof(uint32_t);
}

int pgx_presence_bitmap_dev(pgx_ctx *ctx, const int32_t *d_rows, const int32_t *d_genomes,
                            uint64_t n_records, uint32_t n_rows, uint32_t n_genomes,
                            uint64_t *d_out_bits, uint64_t *d_counters, void *stream_) {
    PGX_REQUIRE(ctx && d_out_bits, "NULL argument");
    PGX_REQUIRE(n_records == 0 || (d_rows && d_genomes), "NULL record arrays");
    hipStream_t stream = (hipStream_t)stream_;
    const uint32_t stride = pgx_bitmap_stride_words(n_rows);
    PGX_HIP(hipMemsetAsync(d_out_bits, 0, (size_t)n_genomes * stride * 8, stream));
    if (d_counters) PGX_HIP(hipMemsetAsync(d_counters, 0, 16, stream));
    if (n_records == 0) return PGX_OK;
    const uint64_t want = (n_records + 255) / 256;
    const uint32_t grid = (uint32_t)(want < 4096 ? want : 4096);
    {
        ProfScope prof(ctx, "presence_bitmap_kernel", stream);
        presence_bitmap_kernel<<<grid, 256, 0, stream>>>(d_rows, d_genomes, n_records, n_rows,
                                                         n_genomes, stride,
                                                         (unsigned long long *)d_out_bits,
                                                         (unsigned long long *)d_counters);
    }
    PGX_HIP(hipGetLastError());
    return PGX_OK;
}

// uploads the records, builds the bitmap on the device; counters (device, 2 x u64) are left for the caller
static int upload_and_build_bitmap(pgx_ctx *ctx, const int32_t *rows, const int32_t *genomes, uint64_t n_records,
                                   uint32_t n_rows, uint32_t n_genomes, DevBuf &d_bits, DevBuf &d_cnt) {
    const size_t nbits = (size_t)n_genomes * pgx_bitmap_stride_words(n_rows) * 8;
    PcBuf d_rows(ctx, PC_SLOT_ROWS), d_genomes(ctx, PC_SLOT_GENOMES);
    PGX_HIP(d_rows.alloc(n_records * 4));
    PGX_HIP(d_genomes.alloc(n_records * 4));
    PGX_HIP(d_bits.alloc(nbits));
    PGX_HIP(d_cnt.alloc(16));
    if (n_records) {
        int rc = pgx_staged_h2d(ctx, d_rows.p, rows, n_records * 4, ctx->stream);
        if (rc == PGX_OK) rc = pgx_staged_h2d(ctx, d_genomes.p, genomes, n_records * 4, ctx->stream);
        if (rc != PGX_OK) return rc;
    }
    return pgx_presence_bitmap_dev(ctx, d_rows.as<int32_t>(), d_genomes.as<int32_t>(), n_records, n_rows,
                                   n_genomes, d_bits.as<uint64_t>(), d_cnt.as<uint64_t>(), ctx->stream);
}

int pgx_presence_bitmap(pgx_ctx *ctx, const int32_t *rows, const int32_t *genomes, uint64_t n_records,
                        uint32_t n_rows, uint32_t n_genomes, uint64_t *out_bits, uint64_t *out_duplicates) {
    PGX_REQUIRE(ctx && out_bits, "NULL argument");
    PGX_REQUIRE(n_records == 0 || (rows && genomes), "NULL record arrays");
    PGX_HIP(hipSetDevice(ctx->device_id));
    const size_t nbits = (size_t)n_genomes * pgx_bitmap_stride_words(n_rows) * 8;
    PcBuf d_bits(ctx, PC_SLOT_BITS), d_cnt(ctx, PC_SLOT_CNT);
    int rc = upload_and_build_bitmap(ctx, rows, genomes, n_records, n_rows, n_genomes, d_bits, d_cnt);
    if (rc != PGX_OK) return rc;
    uint64_t cnt[2] = {0, 0};
    if (nbits) PGX_HIP(hipMemcpyAsync(out_bits, d_bits.p, nbits, hipMemcpyDeviceToHost, ctx->stream));
    PGX_HIP(hipMemcpyAsync(cnt, d_cnt.p, 16, hipMemcpyDeviceToHost, ctx->stream));
    PGX_HIP(hipStreamSynchronize(ctx->stream));
    PGX_REQUIRE(cnt[1] == 0, "record with row or genome index out of range");
    if (out_duplicates) *out_duplicates = cnt[0];
    return PGX_OK;
}

int pgx_pan_core_dev(pgx_ctx *ctx, const uint64_t *d_bits, uint32_t n_genes, uint32_t n_genomes,
                     const int32_t *d_perms, uint32_t n_iter, int32_t *d_out_pan, int32_t *d_out_core,
                     void *d_workspace, size_t workspace_bytes, void *stream_) {
    PGX_REQUIRE(ctx, "NULL context");
    if (n_iter == 0 || n_genomes == 0) return PGX_OK;
    PGX_REQUIRE(d_bits && d_perms && d_out_pan && d_out_core && d_workspace, "NULL argument");
    PGX_REQUIRE(workspace_bytes >= pgx_pan_core_workspace_bytes(n_genes, n_genomes, n_iter),
                "workspace too small (see pgx_pan_core_workspace_bytes)");
    PGX_REQUIRE(((uintptr_t)d_bits & 15u) == 0, "bitmap must be 16-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    const PanCoreGeom g = make_geom(n_genes, n_genomes);
    const uint32_t n_partials = g.partials;
    {
        const uint64_t items = (uint64_t)n_iter * g.wps;  // per stripe
        const uint64_t blocks = ((items + PC_WAVES - 1) / PC_WAVES) * g.stripes;
        PGX_REQUIRE(blocks < (1ull << 31), "problem too large for one launch");
        ProfScope prof(ctx, "pan_core_sweep_kernel", stream);
        pan_core_sweep_kernel<<<(uint32_t)blocks, PC_WAVES * 64, 0, stream>>>(
            (const uint4 *)d_bits, g.stride * 8, d_perms, n_iter, n_genomes, g.Ls, g.wps, g.Lw, g.stripes,
            (uint32_t *)d_workspace);
    }
    PGX_HIP(hipGetLastError());
    const size_t n_out = (size_t)n_iter * n_genomes;
    const size_t want = (n_out + 255) / 256;
    {
        ProfScope prof(ctx, "pan_core_reduce_kernel", stream);
        pan_core_reduce_kernel<<<(uint32_t)(want < 2048 ? want : 2048), 256, 0, stream>>>(
            (const uint32_t *)d_workspace, n_partials, n_out, d_out_pan, d_out_core);
    }
    PGX_HIP(hipGetLastError());
    return PGX_OK;
}

// bitmap already on the device (d_bits): permutations up, curves down
static int pan_core_from_device_bitmap(pgx_ctx *ctx, const uint64_t *d_bits, uint32_t n_genes, uint32_t n_genomes,
                                       const int32_t *perms, uint32_t n_iter, int32_t *out_pan, int32_t *out_core) {
    for (size_t k = 0; k < (size_t)n_iter * n_genomes; ++k)
        PGX_REQUIRE((uint32_t)perms[k] < n_genomes, "permutation entry out of range");
    const size_t nperm = (size_t)n_iter * n_genomes * 4;
    const size_t nws = pgx_pan_core_workspace_bytes(n_genes, n_genomes, n_iter);
    PcBuf d_perms(ctx, PC_SLOT_PERMS), d_pan(ctx, PC_SLOT_PAN), d_core(ctx, PC_SLOT_CORE), d_ws(ctx, PC_SLOT_WS);
    PGX_HIP(d_perms.alloc(nperm));
    PGX_HIP(d_pan.alloc(nperm));
    PGX_HIP(d_core.alloc(nperm));
    PGX_HIP(d_ws.alloc(nws));
    PGX_HIP(hipMemcpyAsync(d_perms.p, perms, nperm, hipMemcpyHostToDevice, ctx->stream));
    int rc = pgx_pan_core_dev(ctx, d_bits, n_genes, n_genomes, d_perms.as<int32_t>(), n_iter,
                              d_pan.as<int32_t>(), d_core.as<int32_t>(), d_ws.p, nws, ctx->stream);
    if (rc != PGX_OK) return rc;
    PGX_HIP(hipMemcpyAsync(out_pan, d_pan.p, nperm, hipMemcpyDeviceToHost, ctx->stream));
    PGX_HIP(hipMemcpyAsync(out_core, d_core.p, nperm, hipMemcpyDeviceToHost, ctx->stream));
    return PGX_OK;
}

int pgx_pan_core(pgx_ctx *ctx, const uint64_t *bits, uint32_t n_genes, uint32_t n_genomes,
                 const int32_t *perms, uint32_t n_iter, int32_t *out_pan, int32_t *out_core) {
    PGX_REQUIRE(ctx, "NULL context");
    if (n_iter == 0 || n_genomes == 0) return PGX_OK;
    PGX_REQUIRE(bits && perms && out_pan && out_core, "NULL argument");
    PGX_HIP(hipSetDevice(ctx->device_id));
    const size_t nbits = (size_t)n_genomes * pgx_bitmap_stride_words(n_genes) * 8;
    PcBuf d_bits(ctx, PC_SLOT_BITS);
    PGX_HIP(d_bits.alloc(nbits));
    PGX_HIP(hipMemcpyAsync(d_bits.p, bits, nbits, hipMemcpyHostToDevice, ctx->stream));
    int rc = pan_core_from_device_bitmap(ctx, d_bits.as<uint64_t>(), n_genes, n_genomes, perms, n_iter, out_pan, out_core);
    if (rc != PGX_OK) return rc;
    PGX_HIP(hipStreamSynchronize(ctx->stream));
    return PGX_OK;
}

int pgx_pan_core_coo(pgx_ctx *ctx, const int32_t *rows, const int32_t *genomes, uint64_t n_records,
                     uint32_t n_genes, uint32_t n_genomes, const int32_t *perms, uint32_t n_iter,
                     int32_t *out_pan, int32_t *out_core, uint64_t *out_duplicates) {
    PGX_REQUIRE(ctx, "NULL context");
    PGX_REQUIRE(n_records == 0 || (rows && genomes), "NULL record arrays");
    PGX_REQUIRE(n_iter == 0 || n_genomes == 0 || (perms && out_pan && out_core), "NULL argument");
    PGX_HIP(hipSetDevice(ctx->device_id));
    PcBuf d_bits(ctx, PC_SLOT_BITS), d_cnt(ctx, PC_SLOT_CNT);
    int rc = upload_and_build_bitmap(ctx, rows, genomes, n_records, n_genes, n_genomes, d_bits, d_cnt);
    if (rc != PGX_OK) return rc;
    if (n_iter && n_genomes) {
        rc = pan_core_from_device_bitmap(ctx, d_bits.as<uint64_t>(), n_genes, n_genomes, perms, n_iter, out_pan, out_core);
        if (rc != PGX_OK) return rc;
    }
    uint64_t cnt[2] = {0, 0};
    PGX_HIP(hipMemcpyAsync(cnt, d_cnt.p, 16, hipMemcpyDeviceToHost, ctx->stream));
    PGX_HIP(hipStreamSynchronize(ctx->stream));
    PGX_REQUIRE(cnt[1] == 0, "record with row or genome index out of range");
    if (out_duplicates) *out_duplicates = cnt[0];
    return PGX_OK;
}

// estimate_pan_core_size() in one call: the permutations are drawn from the legacy generator's state on a host
// thread WHILE the coordinates travel to the device and the bitmap is built there (3 ms of draws beside 2-3 ms of
// copies for the 150,000 x 400 table), then they follow and the curves are computed.
int pgx_pan_core_coo_rng(pgx_ctx *ctx, const int32_t *rows, const int32_t *genomes, uint64_t n_records,
                         uint32_t n_genes, uint32_t n_genomes, uint32_t *mt_key, int32_t *mt_pos, uint32_t n_iter,
                         int32_t *out_perms, int32_t *out_pan, int32_t *out_core, uint64_t *out_duplicates) {
    PGX_REQUIRE(ctx, "NULL context");
    PGX_REQUIRE(n_records == 0 || (rows && genomes), "NULL record arrays");
    PGX_REQUIRE(mt_key && mt_pos, "NULL generator state");
    PGX_REQUIRE(n_iter == 0 || n_genomes == 0 || (out_perms && out_pan && out_core), "NULL argument");
    PGX_HIP(hipSetDevice(ctx->device_id));
    int rc_draw = PGX_OK;
    std::string draw_error;
    std::thread draw([&]() {
        rc_draw = pgx_legacy_shuffles(mt_key, mt_pos, n_genomes, n_iter, out_perms);
        if (rc_draw != PGX_OK) draw_error = pgx_last_error();     // (the error text is thread-local)
    });
    PcBuf d_bits(ctx, PC_SLOT_BITS), d_cnt(ctx, PC_SLOT_CNT);
    int rc = upload_and_build_bitmap(ctx, rows, genomes, n_records, n_genes, n_genomes, d_bits, d_cnt);
    draw.join();
    if (rc != PGX_OK) return rc;
    if (rc_draw != PGX_OK) { pgx_set_error("%s", draw_error.c_str()); return rc_draw; }
    if (n_iter && n_genomes) {
        rc = pan_core_from_device_bitmap(ctx, d_bits.as<uint64_t>(), n_genes, n_genomes, out_perms, n_iter, out_pan, out_core);
        if (rc != PGX_OK) return rc;
    }
    uint64_t cnt[2] = {0, 0};
    PGX_HIP(hipMemcpyAsync(cnt, d_cnt.p, 16, hipMemcpyDeviceToHost, ctx->stream));
    PGX_HIP(hipStreamSynchronize(ctx->stream));
    PGX_REQUIRE(cnt[1] == 0, "record with row or genome index out of range");
    if (out_duplicates) *out_duplicates = cnt[0];
    return PGX_OK;
}

// estimate_pan_core_size() from the table's COO arrays to the float64 result in ONE call (pangenome_analysis.py:51-98):
// on host threads, side by side -- the legacy-generator draws, the check that every stored value is 1, the staged
// upload of the coordinates; on the device -- bitmap, curves, the [n_iter][2 S] float64 table; one copy down.
int pgx_pan_core_table(pgx_ctx *ctx, const int32_t *rows, const int32_t *genomes, const int64_t *values,
                       uint64_t n_records, uint32_t n_genes, uint32_t n_genomes, uint32_t *mt_key, int32_t *mt_pos,
                       uint32_t n_iter, int32_t *out_perms, double *out_table, uint64_t *out_duplicates,
                       uint64_t *out_not_one) {
    PGX_REQUIRE(ctx, "NULL context");
    PGX_REQUIRE(n_records == 0 || (rows && genomes), "NULL record arrays");
    PGX_REQUIRE(mt_key && mt_pos, "NULL generator state");
    PGX_REQUIRE(n_iter == 0 || n_genomes == 0 || (out_perms && out_table), "NULL argument");
    PGX_HIP(hipSetDevice(ctx->device_id));
    int rc_draw = PGX_OK;
    std::string draw_error;
    std::thread draw([&]() {
        rc_draw = pgx_legacy_shuffles(mt_key, mt_pos, n_genomes, n_iter, out_perms);
        if (rc_draw != PGX_OK) draw_error = pgx_last_error();     // (the error text is thread-local)
    });
    // every stored value must be 1 (the OR/AND form equals the reference's loop only for a 0/1 table): checked here,
    // on a few threads, instead of by a 3 ms numpy pass before the call
    uint64_t not_one = 0;
    std::vector<std::thread> checkers;
    std::vector<uint64_t> bad(4, 0);
    if (values && n_records) {
        const unsigned T = n_records < (1u << 20) ? 1u : 4u;
        for (unsigned t = 0; t < T; ++t)
            checkers.emplace_back([&, t, T]() {
                const uint64_t a = n_records * t / T, b = n_records * (t + 1) / T;
                uint64_t c = 0;
                for (uint64_t i = a; i < b; ++i) c += values[i] != 1;
                bad[t] = c;
            });
    }
    PcBuf d_bits(ctx, PC_SLOT_BITS), d_cnt(ctx, PC_SLOT_CNT);
    int rc = upload_and_build_bitmap(ctx, rows, genomes, n_records, n_genes, n_genomes, d_bits, d_cnt);
    draw.join();
    for (auto &th : checkers) th.join();
    for (uint64_t c : bad) not_one += c;
    if (out_not_one) *out_not_one = not_one;
    if (rc != PGX_OK) return rc;
    if (rc_draw != PGX_OK) { pgx_set_error("%s", draw_error.c_str()); return rc_draw; }
    if (n_iter && n_genomes && !not_one) {
        const size_t nperm = (size_t)n_iter * n_genomes * 4;
        const size_t nws = pgx_pan_core_workspace_bytes(n_genes, n_genomes, n_iter);
        PcBuf d_perms(ctx, PC_SLOT_PERMS), d_ws(ctx, PC_SLOT_WS), d_table(ctx, PC_SLOT_TABLE);
        PGX_HIP(d_perms.alloc(nperm));
        PGX_HIP(d_ws.alloc(nws));
        PGX_HIP(d_table.alloc(nperm * 4));
        PGX_HIP(hipMemcpyAsync(d_perms.p, out_perms, nperm, hipMemcpyHostToDevice, ctx->stream));
        const PanCoreGeom g = make_geom(n_genes, n_genomes);
        {
            const uint64_t items = (uint64_t)n_iter * g.wps;  // per stripe
            const uint64_t blocks = ((items + PC_WAVES - 1) / PC_WAVES) * g.stripes;
            PGX_REQUIRE(blocks < (1ull << 31), "problem too large for one launch");
            ProfScope prof(ctx, "pan_core_sweep_kernel", ctx->stream);
            pan_core_sweep_kernel<<<(uint32_t)blocks, PC_WAVES * 64, 0, ctx->stream>>>(
                d_bits.as<uint4>(), g.stride * 8, d_perms.as<int32_t>(), n_iter, n_genomes, g.Ls, g.wps, g.Lw, g.stripes,
                d_ws.as<uint32_t>());
        }
        PGX_HIP(hipGetLastError());
        {
            const size_t want = ((size_t)n_iter * n_genomes + 255) / 256;
            ProfScope prof(ctx, "pan_core_reduce_kernel", ctx->stream);
            pan_core_reduce_table_kernel<<<(uint32_t)(want < 2048 ? want : 2048), 256, 0, ctx->stream>>>(
                d_ws.as<uint32_t>(), g.partials, n_iter, n_genomes, d_table.as<double>());
        }
        PGX_HIP(hipGetLastError());
        PGX_HIP(hipMemcpyAsync(out_table, d_table.p, nperm * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    uint64_t cnt[2] = {0, 0};
    PGX_HIP(hipMemcpyAsync(cnt, d_cnt.p, 16, hipMemcpyDeviceToHost, ctx->stream));
    PGX_HIP(hipStreamSynchronize(ctx->stream));
    PGX_REQUIRE(cnt[1] == 0, "record with row or genome index out of range");
    if (out_duplicates) *out_duplicates = cnt[0];
    return PGX_OK;
}

// ---- device-resident hand-off: clustering result -> bitmap (kept in the context) -> pan/core curves ---------------
int pgx_bitmap_from_clusters(pgx_ctx *ctx, const int32_t *cluster_of_group, uint64_t n_groups,
                             const int32_t *group_of_record, const uint32_t *file_of_record, uint64_t n_records,
                             const int32_t *genome_of_file, uint32_t n_files, uint32_t n_genes, uint32_t n_genomes,
                             uint64_t *out_token) {
    PGX_REQUIRE(ctx && out_token, "NULL argument");
    PGX_REQUIRE(n_records == 0 || (cluster_of_group && group_of_record && file_of_record && genome_of_file), "NULL arrays");
    PGX_REQUIRE(n_groups < (1ull << 31), "too many groups");
    PGX_HIP(hipSetDevice(ctx->device_id));
    *out_token = 0;
    ctx->resident_token = 0;                         // (whatever was resident is gone from here on)
    const uint32_t stride = pgx_bitmap_stride_words(n_genes);
    const size_t nbits = (size_t)n_genomes * stride * 8;
    PcBuf d_bits(ctx, PC_SLOT_RESIDENT), d_cl(ctx, PC_SLOT_RES_A), d_grp(ctx, PC_SLOT_RES_B), d_file(ctx, PC_SLOT_RES_C),
        d_gof(ctx, PC_SLOT_RES_D), d_cnt(ctx, PC_SLOT_CNT);
    PGX_HIP(d_bits.alloc(nbits));
    PGX_HIP(d_cl.alloc(n_groups * 4));
    PGX_HIP(d_grp.alloc(n_records * 4));
    PGX_HIP(d_file.alloc(n_records * 4));
    PGX_HIP(d_gof.alloc((size_t)n_files * 4));
    PGX_HIP(d_cnt.alloc(16));
    hipStream_t st = ctx->stream;
    PGX_HIP(hipMemsetAsync(d_bits.p, 0, nbits, st));
    PGX_HIP(hipMemsetAsync(d_cnt.p, 0, 16, st));
    if (n_records) {
        int rc = pgx_staged_h2d(ctx, d_cl.p, cluster_of_group, n_groups * 4, st);
        if (rc == PGX_OK) rc = pgx_staged_h2d(ctx, d_grp.p, group_of_record, n_records * 4, st);
        if (rc == PGX_OK) rc = pgx_staged_h2d(ctx, d_file.p, file_of_record, n_records * 4, st);
        if (rc != PGX_OK) return rc;
        PGX_HIP(hipMemcpyAsync(d_gof.p, genome_of_file, (size_t)n_files * 4, hipMemcpyHostToDevice, st));
        const uint64_t want = (n_records + 255) / 256;
        ProfScope prof(ctx, "presence_bitmap_kernel", st);
        bitmap_from_clusters_kernel<<<(uint32_t)(want < 4096 ? want : 4096), 256, 0, st>>>(
            d_cl.as<int32_t>(), (uint32_t)n_groups, d_grp.as<int32_t>(), d_file.as<uint32_t>(), n_records, d_gof.as<int32_t>(),
            n_files, n_genes, n_genomes, stride, d_bits.as<unsigned long long>(), d_cnt.as<unsigned long long>());
        PGX_HIP(hipGetLastError());
    }
    uint64_t cnt[2] = {0, 0};
    PGX_HIP(hipMemcpyAsync(cnt, d_cnt.p, 16, hipMemcpyDeviceToHost, st));
    PGX_HIP(hipStreamSynchronize(st));
    PGX_REQUIRE(cnt[1] == 0, "group, file, gene or genome index out of range");
    ctx->resident_token = ctx->resident_next++;
    ctx->resident_genes = n_genes; ctx->resident_genomes = n_genomes;
    *out_token = ctx->resident_token;
    return PGX_OK;
}

int pgx_bitmap_resident_read(pgx_ctx *ctx, uint64_t token, uint64_t *out_bits) {
    PGX_REQUIRE(ctx && out_bits, "NULL argument");
    PGX_REQUIRE(token != 0 && token == ctx->resident_token, "the bitmap of that token is not resident any more");
    PGX_HIP(hipSetDevice(ctx->device_id));
    PcBuf d_bits(ctx, PC_SLOT_RESIDENT);
    const size_t nbits = (size_t)ctx->resident_genomes * pgx_bitmap_stride_words(ctx->resident_genes) * 8;
    PGX_HIP(d_bits.alloc(nbits));                     // (a view of the slot: no allocation)
    PGX_HIP(hipMemcpyAsync(out_bits, d_bits.p, nbits, hipMemcpyDeviceToHost, ctx->stream));
    PGX_HIP(hipStreamSynchronize(ctx->stream));
    return PGX_OK;
}

int pgx_pan_core_table_resident(pgx_ctx *ctx, uint64_t token, uint32_t n_genes, uint32_t n_genomes, uint32_t *mt_key,
                                int32_t *mt_pos, uint32_t n_iter, int32_t *out_perms, double *out_table) {
    PGX_REQUIRE(ctx && mt_key && mt_pos, "NULL argument");
    PGX_REQUIRE(token != 0 && token == ctx->resident_token && n_genes == ctx->resident_genes && n_genomes == ctx->resident_genomes,
                "the bitmap of that token is not resident any more");
    PGX_REQUIRE(n_iter == 0 || n_genomes == 0 || (out_perms && out_table), "NULL argument");
    PGX_HIP(hipSetDevice(ctx->device_id));
    int rc = pgx_legacy_shuffles(mt_key, mt_pos, n_genomes, n_iter, out_perms);
    if (rc != PGX_OK) return rc;
    if (n_iter == 0 || n_genomes == 0) return PGX_OK;
    const size_t nperm = (size_t)n_iter * n_genomes * 4;
    const size_t nws = pgx_pan_core_workspace_bytes(n_genes, n_genomes, n_iter);
    PcBuf d_bits(ctx, PC_SLOT_RESIDENT), d_perms(ctx, PC_SLOT_PERMS), d_ws(ctx, PC_SLOT_WS), d_table(ctx, PC_SLOT_TABLE);
    PGX_HIP(d_bits.alloc((size_t)n_genomes * pgx_bitmap_stride_words(n_genes) * 8));   // (a view of the slot)
    PGX_HIP(d_perms.alloc(nperm));
    PGX_HIP(d_ws.alloc(nws));
    PGX_HIP(d_table.alloc(nperm * 4));
    PGX_HIP(hipMemcpyAsync(d_perms.p, out_perms, nperm, hipMemcpyHostToDevice, ctx->stream));
    const PanCoreGeom g = make_geom(n_genes, n_genomes);
    {
        const uint64_t items = (uint64_t)n_iter * g.wps;
        const uint64_t blocks = ((items + PC_WAVES - 1) / PC_WAVES) * g.stripes;
        PGX_REQUIRE(blocks < (1ull << 31), "problem too large for one launch");
        ProfScope prof(ctx, "pan_core_sweep_kernel", ctx->stream);
        pan_core_sweep_kernel<<<(uint32_t)blocks, PC_WAVES * 64, 0, ctx->stream>>>(
            d_bits.as<uint4>(), g.stride * 8, d_perms.as<int32_t>(), n_iter, n_genomes, g.Ls, g.wps, g.Lw, g.stripes,
            d_ws.as<uint32_t>());
    }
    PGX_HIP(hipGetLastError());
    {
        const size_t want = ((size_t)n_iter * n_genomes + 255) / 256;
        ProfScope prof(ctx, "pan_core_reduce_kernel", ctx->stream);
        pan_core_reduce_table_kernel<<<(uint32_t)(want < 2048 ? want : 2048), 256, 0, ctx->stream>>>(
            d_ws.as<uint32_t>(), g.partials, n_iter, n_genomes, d_table.as<double>());
    }
    PGX_HIP(hipGetLastError());
    PGX_HIP(hipMemcpyAsync(out_table, d_table.p, nperm * 4, hipMemcpyDeviceToHost, ctx->stream));
    PGX_HIP(hipStreamSynchronize(ctx->stream));
    return PGX_OK;
}

int pgx_row_counts_dev(pgx_ctx *ctx, const uint64_t *d_bits, uint32_t n_rows, uint32_t n_genomes, int32_t *d_counts,
                       void *stream_) {
    PGX_REQUIRE(ctx, "NULL context");
    if (n_rows == 0) return PGX_OK;
    PGX_REQUIRE(d_bits && d_counts, "NULL argument");
    hipStream_t stream = (hipStream_t)stream_;
    {
        ProfScope prof(ctx, "row_counts_kernel", stream);
        row_counts_kernel<<<(n_rows + 255) / 256, 256, 0, stream>>>((const unsigned long long *)d_bits,
                                                                    pgx_bitmap_stride_words(n_rows), n_genomes, n_rows, d_counts);
    }
    PGX_HIP(hipGetLastError());
    return PGX_OK;
}

int pgx_row_counts(pgx_ctx *ctx, const int32_t *rows, const int32_t *genomes, uint64_t n_records, uint32_t n_rows,
                   uint32_t n_genomes, int32_t *out_counts, uint64_t *out_duplicates) {
    PGX_REQUIRE(ctx, "NULL context");
    PGX_REQUIRE(n_records == 0 || (rows && genomes), "NULL record arrays");
    PGX_REQUIRE(n_rows == 0 || out_counts, "NULL argument");
    PGX_HIP(hipSetDevice(ctx->device_id));
    PcBuf d_bits(ctx, PC_SLOT_BITS), d_cnt(ctx, PC_SLOT_CNT), d_counts(ctx, PC_SLOT_COUNTS);
    int rc = upload_and_build_bitmap(ctx, rows, genomes, n_records, n_rows, n_genomes, d_bits, d_cnt);
    if (rc != PGX_OK) return rc;
    PGX_HIP(d_counts.alloc((size_t)n_rows * 4));
    rc = pgx_row_counts_dev(ctx, d_bits.as<uint64_t>(), n_rows, n_genomes, d_counts.as<int32_t>(), ctx->stream);
    if (rc != PGX_OK) return rc;
    uint64_t cnt[2] = {0, 0};
    if (n_rows) PGX_HIP(hipMemcpyAsync(out_counts, d_counts.p, (size_t)n_rows * 4, hipMemcpyDeviceToHost, ctx->stream));
    PGX_HIP(hipMemcpyAsync(cnt, d_cnt.p, 16, hipMemcpyDeviceToHost, ctx->stream));
    PGX_HIP(hipStreamSynchronize(ctx->stream));
    PGX_REQUIRE(cnt[1] == 0, "record with row or genome index out of range");
    if (out_duplicates) *out_duplicates = cnt[0];
    return PGX_OK;
}

}  // extern "C"
