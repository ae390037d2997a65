// K1/K2: greedy incremental clustering with cd-hit's rules on gfx950.
//
// Replaces the reference's shell-out `cd-hit -i nr.faa -o ... -d 0 -n 5 -c 0.8`
// (pangenome.py:444-450). The sequential rule being reproduced (oracle/cluster_ref.c):
// sequences in stable descending-length order; query q joins the FIRST representative r
// (created before q) that passes  word count >= required_aan  ->  diagonal test >=
// required_aas  ->  banded alignment identity >= c,  candidates ordered by
// (smallest shared word code, representative index); otherwise q becomes a representative.
// Acceptance A(q, r) is a pure function of the pair, so the GPU evaluates pairs in bulk
// and only the final "first accepted in key order" selection follows the greedy order:
// the winner is the 64-bit minimum of  strand << 63 | smallest shared code << 32 | r  over the
// accepted representatives r < q (r = position in the sorted order = creation order).
//
// The word table of the sequential rule is kept on the device for the whole call as an INVERTED
// INDEX, one 256-byte line per word code (list length + first 59 entries; longer lists continue in a
// pool), appended to whenever representatives are confirmed. The short-word filter is QUERY-MAJOR:
// one wave per query walks the lines of the query's words, so a query touches exactly the posting
// entries the sequential rule visits.
//
// One WINDOW handles up to 65536 consecutive queries (nucleotides, both strands: 512, each query with a
// second slot for its reverse complement), bounded also by the word volume of its discovery chunks:
//   phase A   filter over the whole index (representatives of earlier windows) -> candidate pairs
//             -> diag (k-mer diagonal histogram, best band) -> align (banded DP on the anti-diagonal
//             wavefront, four pairs per wave) -> best[q] = min key of the accepted pairs
//   discovery still-open members that cannot have an earlier open candidate (linear-time word test over
//             per-chunk first-open tags) are certain new representatives: appended to the index at once
//             (one pass), and the filter then visits, for all window members, only the entries this round
//             added; twice, all on the device
//   blocks    members still open are resolved <= 4096 at a time, exactly: they are appended to the index
//             tentatively, the filter finds the in-block pairs, the host walks the block in order, the
//             entries of the members that joined a representative are struck out again and every
//             window member is compared with what is left
//   close     winners, new representatives and pair records go to the host in one publish launch;
//             the host's bookkeeping of a window (and its outputs) runs behind the next window's kernels.
// Consecutive windows overlap on two streams: the tail of a window (from its last strike-out on) only reads
// the index, and so does phase A of the next; each has its own counters, best keys, flags and pair records.
// Record-sharded mode (one process per GPU, pgx.h): window member i belongs to process i % W, which
// filters and aligns it against its replica of the index; the members' best keys are all-gathered
// (RCCL, enqueued on the stream) after every evaluation; discovery, index appends and the block
// walk are replicated and deterministic, so every replica of the index stays identical.
//
// HBM layout: residues 1 byte/residue in sorted order (reverse complements appended as virtual
// sequences) + a 5-bit packed copy for the aligner; word lists (u32 code, u16 mult) at the same
// offsets; the index lines and pool; grow-only per-context workspace.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <type_traits>
#include <functional>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "cluster_tables.h"
#include "pgx_internal.h"

namespace {

using namespace pgxc;

constexpr uint32_t kWindowMax = 65536; // largest window (queries); first-open tags keep the member in 16 bits
constexpr int kDiscoveryRounds = 2;    // device-only discovery rounds before the blocks
constexpr uint32_t kBlockCap = 4096;   // most open members resolved together inside a window (512 for nucleotides)
constexpr uint32_t kBlockCapMin = 64;  // what a block shrinks to when its candidate pairs overflow the buffer

constexpr uint32_t kMaxLen = 1u << 22;  // longest supported sequence (cd-hit's own limit is 655,360)
// Up to this length a sequence takes the ordinary paths: its words fit the LDS sorter, and the diagonal histogram packs
// hits and complexity-weighted hits of a diagonal into 16 bits each (hits <= query length; a nucleotide 4-mer weighs
// up to 4, a protein 2-mer up to 2). Longer sequences -- a handful of giant proteins exist -- take a global-memory
// word table and 64-bit histogram cells.
constexpr uint32_t kPackedLenAa = 32767, kPackedLenNt = 16383;
constexpr uint32_t kSentinel = 0xFFFFFFFFu;
constexpr uint32_t kDiagLdsCap = 2048;  // diagonals / query 2-mers kept in LDS by the diag kernel
constexpr uint32_t kDiagLdsSmall = 512; // ... for windows whose queries are that short (most of them)
constexpr int kMaxBand = 64;

__constant__ int8_t kBlosum62_dev[kNAA1 * kNAA1] = PGXC_BLOSUM62_FLAT;

enum : uint32_t { F_DIAG_PASS = 1, F_BAND_OK = 2, F_ACCEPT = 4, F_TOO_BIG = 8, F_EVAL = 16, F_ALIGNED = 32 };
// bits of a window's error word (counter C_ERR). In the record-sharded mode the word travels with every exchange of
// the best keys and is OR-ed over the processes, so that every process sees the same word at the same point of the
// window loop and all of them leave together.
enum : uint32_t { E_TOUCHED = 1,    // an append round set more entries aside than the scratch list holds
                  E_POOL = 2,       // the overflow pool of the word index is full
                  E_TABLE = 4,      // the filter's exact table overflowed at the finest residue class
                  E_BAND = 8,       // a pair that passed the diagonal test has a band wider than kMaxBand
                  E_PAIRS = 16,     // the window's candidate pair buffer overflowed
                  E_WORDMULT = 32 };// a word occurs more than 65535 times in one (giant, degenerate) sequence

struct Pair {           // one (query, representative) candidate
    uint32_t q;         // sorted sequence index of the query
    uint32_t r;         // phase A: representative index; phase B: sorted sequence index
    uint32_t cnt;       // short-word count
    uint32_t minc;      // smallest shared word code (candidate order key)
    int32_t best_sum, band_left, band_center, band_right;
    int32_t iden;
    uint32_t flags;
};

// Which pair records a diag / align launch works on: the device-side range
// [*d_begin, min(*d_end, cap)) or an explicit list; optionally only pairs whose candidate (a
// batch member, local index r - b0) has `only_a` set and/or `only_not_b` clear. Accepted
// pairs can be reported per query in `accepted_out` (local index).
struct PairSel {
    const uint32_t *d_begin, *d_end;
    uint32_t cap;
    const uint32_t *list;
    uint32_t n_list;
    const uint8_t *only_a, *only_not_b;
    uint8_t *accepted_out;
    uint32_t b0;
    uint32_t skip_evaluated;  // leave pairs alone that an earlier round has been through
    uint32_t all_if_le;       // a selection of at most this many pairs ignores only_a / only_not_b: every pair is taken
};
__device__ __forceinline__ uint32_t sel_count(const PairSel &s) {
    if (s.list) return s.n_list;
    uint32_t e = *s.d_end;
    if (e > s.cap) e = s.cap;
    const uint32_t b = s.d_begin ? *s.d_begin : 0u;
    return e > b ? e - b : 0u;
}
__device__ __forceinline__ uint32_t sel_pair(const PairSel &s, uint32_t w) {
    return s.list ? s.list[w] : (s.d_begin ? *s.d_begin : 0u) + w;
}

struct DevSeqs {
    const uint8_t *res;     // residue indices, sorted order, concatenated
    const uint64_t *off;    // [n+1]
    const uint32_t *len;    // [n]
    const uint32_t *wcode;  // distinct word codes at off[k] (no particular order)
    const uint16_t *wmult;  // multiplicities
    const uint32_t *wcnt;   // [n] number of distinct words
    // Nucleotide clustering with both strands stores the reverse complement of sequence k as
    // the virtual sequence n_fwd + k (same length, own residues and word list): comparing the
    // reverse strand of a query is then an ordinary comparison of that virtual sequence.
    // 5-bit packed copy of the residues, six per 32-bit word (bits 5t..5t+4 = residue t); sequence k
    // starts at word pk_off[k]. The alignment kernel stages its operands from this copy with
    // coalesced word loads and unpacks them into LDS.
    const uint32_t *pk;
    const uint32_t *pk_off;
    uint32_t n_fwd;         // number of real sequences
    int32_t base;           // word / k-mer radix: 21 (protein) or 4 (nucleotide)
    int32_t kd;             // k-mer length of the diagonal test: 2 or 4
    int32_t nt;             // nucleotide rules
    // an index entry = sequence index in the low `mshift` bits | min(multiplicity of the word in that sequence,
    // field maximum) above them; the field maximum itself means "look it up in the word list"
    uint32_t mshift;
};
__device__ __forceinline__ uint32_t entry_rmask(const DevSeqs &S) { return (1u << S.mshift) - 1u; }
__device__ __forceinline__ uint32_t entry_fmax(const DevSeqs &S) { return 0xFFFFFFFFu >> S.mshift; }
__device__ __forceinline__ uint32_t real_of(const DevSeqs &S, uint32_t k) { return k >= S.n_fwd ? k - S.n_fwd : k; }

// ----------------------------------------------------------------------------------------
// prep: letter count per input sequence, then gather + encode into sorted order
// ----------------------------------------------------------------------------------------
__constant__ int8_t kAa2Idx_dev[26] = {0, 2, 4, 3, 6, 13, 7, 8, 9, 20, 11, 10, 12,
                                       2, 20, 14, 5, 1, 15, 16, 20, 19, 17, 20, 18, 6};

__device__ __forceinline__ bool is_letter(uint8_t ch) {
    ch &= 0xDF;
    return ch >= 'A' && ch <= 'Z';
}

// one wave per input sequence
__global__ __launch_bounds__(256) void seq_len_kernel(const uint8_t *__restrict__ res,
                                                     const uint64_t *__restrict__ off, uint32_t n, uint64_t total,
                                                     uint32_t *__restrict__ len) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t i = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (i >= n) return;
    const uint64_t b = off[i], e = off[i + 1];
    uint32_t c = 0;
    // four bytes per lane and load (a byte per lane moves 64 bytes per load instruction: the kernel was bound by their
    // number): aligned words from the word that holds byte b on, bytes outside [b, e) masked out
    const uint64_t a0 = b & ~3ull;
    for (uint64_t a = a0 + 4ull * lane; a < e; a += 256) {
        uint32_t word = 0;
        if (a + 4 <= total) word = *reinterpret_cast<const uint32_t *>(res + a);
        else for (uint64_t p = a; p < total; ++p) word |= (uint32_t)res[p] << (8 * (p - a));   // (the buffer's last bytes)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint64_t p = a + t;
            c += (p >= b && p < e) && is_letter((uint8_t)(word >> (8 * t)));
        }
    }
    for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
    if (lane == 0) len[i] = c;
}

// one wave per sorted sequence: copy the letters of input sequence order[k] to out_off[k],
// mapped to residue indices (non-letters are dropped, so the copy compacts with a ballot)
__global__ __launch_bounds__(256) void encode_gather_kernel(const uint8_t *__restrict__ in,
                                                           const uint64_t *__restrict__ in_off,
                                                           const uint32_t *__restrict__ order,
                                                           const uint64_t *__restrict__ out_off,
                                                           uint8_t *__restrict__ out, uint32_t n, int nt) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t k = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (k >= n) return;
    const uint32_t src = order[k];
    const uint64_t b = in_off[src], e = in_off[src + 1];
    uint64_t w = out_off[k];
    for (uint64_t q0 = b; q0 < e; q0 += 256) {           // (four slabs of 64 bytes loaded before any is compacted)
        uint8_t chs[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const uint64_t p = q0 + 64u * j + lane; chs[j] = p < e ? in[p] : (uint8_t)0; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint8_t ch = chs[j];
            const bool ok = is_letter(ch);
            const unsigned long long m = __ballot(ok);
            if (ok) {
                const uint8_t up = ch & 0xDF;
                const uint8_t idx = nt ? (uint8_t)(up == 'A' ? 0 : up == 'C' ? 1 : up == 'G' ? 2 : (up == 'T' || up == 'U') ? 3 : 4)
                                       : (uint8_t)kAa2Idx_dev[up - 'A'];
                out[w + __popcll(m & ((1ull << lane) - 1ull))] = idx;
            }
            w += __popcll(m);
        }
    }
}

// reverse complement of sequence k written as virtual sequence n + k (one wave per sequence)
__global__ __launch_bounds__(256) void revcomp_kernel(uint8_t *__restrict__ res, const uint64_t *__restrict__ off,
                                                     const uint32_t *__restrict__ len, uint32_t n) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t k = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (k >= n) return;
    const uint64_t src = off[k], dst = off[n + k];
    const uint32_t L = len[k];
    for (uint32_t i = lane; i < L; i += 64) {
        const uint8_t b = res[src + L - 1 - i];
        res[dst + i] = b < 4 ? (uint8_t)(3 - b) : b;
    }
}

// 5-bit packing of every (real and virtual) sequence: one wave per sequence, one word per lane
__global__ __launch_bounds__(256) void pack5_kernel(const uint8_t *__restrict__ res, const uint64_t *__restrict__ off,
                                                   const uint32_t *__restrict__ len,
                                                   const uint32_t *__restrict__ pk_off, uint32_t n,
                                                   uint32_t *__restrict__ pk) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t k = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (k >= n) return;
    const uint8_t *s = res + off[k];
    const uint32_t L = len[k], nw = (L + 5u) / 6u;
    uint32_t *dst = pk + pk_off[k];
    for (uint32_t w = lane; w < nw; w += 64) {
        uint32_t x = 0;
        for (uint32_t t = 0; t < 6; ++t)
            if (6 * w + t < L) x |= (uint32_t)s[6 * w + t] << (5u * t);
        dst[w] = x;
    }
}

// ----------------------------------------------------------------------------------------
// words: encode all k-mers of a sequence, sort them in LDS (bitonic), collapse runs
// ----------------------------------------------------------------------------------------
// exclusive prefix of `v` over the workgroup's threads (and the total): a shuffle scan inside every wave, the waves'
// totals through LDS -- one barrier instead of the fourteen of a Hillis-Steele scan over 128 threads in LDS
template <int THREADS>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *__restrict__ wave_tot /* LDS, THREADS / 64 + 1 */,
                                                    uint32_t *total) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t x = v;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d); if ((int)lane >= d) x += y; }
    if (lane == 63u) wave_tot[wave] = x;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; ++w) { const uint32_t t = wave_tot[w]; before += (uint32_t)w < wave ? t : 0u; all += t; }
    *total = all;
    return before + x - v;
}

template <int NCAP, int THREADS>
__global__ __launch_bounds__(THREADS) void words_kernel(const uint8_t *__restrict__ res,
                                                        const uint64_t *__restrict__ off,
                                                        const uint32_t *__restrict__ len, uint32_t k0,
                                                        uint32_t k1, int word_len, int base, int nt,
                                                        uint32_t *__restrict__ wcode,
                                                        uint16_t *__restrict__ wmult,
                                                        uint32_t *__restrict__ wcnt) {
    __shared__ uint32_t keys[NCAP];
    __shared__ uint32_t part[THREADS];
    const uint32_t k = k0 + blockIdx.x;
    if (k >= k1) return;
    const uint32_t tid = threadIdx.x;
    const uint64_t o = off[k];
    const uint8_t *s = res + o;
    const uint32_t nw = len[k] - (uint32_t)word_len + 1u;
    for (uint32_t i = tid; i < NCAP; i += THREADS) {
        uint32_t key = kSentinel;
        if (i < nw) {
            key = 0;
            bool bad = false;
            for (int t = 0; t < word_len; ++t) { key = key * (uint32_t)base + s[i + t]; bad |= s[i + t] >= base; }
            if (nt && bad) key = kSentinel;  // words containing N are skipped (they sort behind all real words)
        }
        keys[i] = key;
    }
    __syncthreads();
    for (uint32_t size = 2; size <= NCAP; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t = tid; t < NCAP / 2; t += THREADS) {
                const uint32_t lo = 2 * t - (t & (stride - 1));
                const uint32_t hi = lo + stride;
                const bool up = (lo & size) == 0;
                const uint32_t a = keys[lo], b = keys[hi];
                if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
            }
            __syncthreads();
        }
    }
    // run heads in this thread's contiguous chunk
    constexpr uint32_t C = NCAP / THREADS;
    const uint32_t beg = tid * C;
    uint32_t heads = 0;
    for (uint32_t i = beg; i < beg + C; ++i)
        heads += (i < nw) && keys[i] != kSentinel && (i == 0 || keys[i] != keys[i - 1]);
    part[tid] = heads;
    __syncthreads();
    for (uint32_t d = 1; d < THREADS; d <<= 1) {  // inclusive Hillis-Steele scan
        const uint32_t v = tid >= d ? part[tid - d] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t idx = part[tid] - heads;
    if (tid == THREADS - 1) wcnt[k] = part[tid];
    for (uint32_t i = beg; i < beg + C; ++i) {
        if ((i < nw) && keys[i] != kSentinel && (i == 0 || keys[i] != keys[i - 1])) {
            uint32_t j = i + 1;
            while (j < nw && keys[j] == keys[i]) ++j;
            wcode[o + idx] = keys[i];
            wmult[o + idx] = (uint16_t)(j - i);
            ++idx;
        }
    }
}

// The same list for sequences of up to 2048 words (nearly all of them) WITHOUT sorting: nothing downstream
// needs the codes in order -- candidates are ordered by an explicit key, the index keeps no order -- only the
// distinct codes and how often each occurs. The words are inserted into an open-addressing table in LDS
// (twice the capacity; compare-and-swap on the code, atomic add on the count) and the occupied slots are
// written out in slot order. Four barriers instead of the bitonic network's 45 stages.
template <int NCAP, int THREADS>
__global__ __launch_bounds__(THREADS) void words_hash_kernel(const uint8_t *__restrict__ res,
                                                             const uint64_t *__restrict__ off,
                                                             const uint32_t *__restrict__ len, uint32_t k0,
                                                             uint32_t k1, int word_len, int base, int nt,
                                                             uint32_t *__restrict__ wcode,
                                                             uint16_t *__restrict__ wmult,
                                                             uint32_t *__restrict__ wcnt) {
    constexpr uint32_t SLOTS = 2 * NCAP;
    __shared__ uint32_t hk[SLOTS];
    __shared__ uint32_t hc[SLOTS];
    __shared__ uint32_t part[THREADS];
    const uint32_t k = k0 + blockIdx.x;
    if (k >= k1) return;
    const uint32_t tid = threadIdx.x;
    const uint64_t o = off[k];
    const uint8_t *s = res + o;
    const uint32_t nw = len[k] - (uint32_t)word_len + 1u;
    for (uint32_t i = tid; i < SLOTS; i += THREADS) { hk[i] = kSentinel; hc[i] = 0u; }
    __syncthreads();
    for (uint32_t i = tid; i < nw; i += THREADS) {
        uint32_t key = 0;
        bool bad = false;
        for (int t = 0; t < word_len; ++t) { key = key * (uint32_t)base + s[i + t]; bad |= s[i + t] >= base; }
        if (nt && bad) continue;                          // words containing N are skipped
        uint32_t slot = (key * 0x9E3779B1u) >> 7 & (SLOTS - 1u);
        for (;;) {                                         // (at most half the slots are ever taken)
            const uint32_t cur = hk[slot];
            if (cur == key) break;
            if (cur == kSentinel) {
                const uint32_t was = atomicCAS(&hk[slot], kSentinel, key);
                if (was == kSentinel || was == key) break;
            }
            slot = (slot + 1u) & (SLOTS - 1u);
        }
        atomicAdd(&hc[slot], 1u);
    }
    __syncthreads();
    constexpr uint32_t C = SLOTS / THREADS;               // slots per thread, contiguous
    const uint32_t beg = tid * C;
    uint32_t mine = 0;
    for (uint32_t i = beg; i < beg + C; ++i) mine += hk[i] != kSentinel;
    uint32_t total;
    uint32_t idx = block_excl_scan<THREADS>(mine, part, &total);
    if (tid == 0) wcnt[k] = total;
    for (uint32_t i = beg; i < beg + C; ++i)
        if (hk[i] != kSentinel) { wcode[o + idx] = hk[i]; wmult[o + idx] = (uint16_t)hc[i]; ++idx; }
}

// Sequences of up to 512 words (most of them), one WAVE per sequence and four sequences per workgroup: no workgroup
// barrier anywhere, 4 KB of LDS per sequence (eight waves per SIMD), one memory latency for the residues.
//   * the residues arrive with ONE 16-byte load per lane (aligned down; up to 537 bytes) and are read from LDS;
//   * code and count share a 32-bit slot (code << 10 | count: a code has 22 bits, a count is at most 512), so an
//     insertion is one compare-and-swap or one add and the table is 1024 x 4 bytes;
//   * the occupied slots are written out in slot order, 64 slots per step, compacted with a ballot.
// SLOTS = 2048 takes the sequences of up to 1023 words the same way (a count still fits ten bits; 8 KB per sequence).
template <uint32_t SLOTS>
__global__ __launch_bounds__(256) void words_wave_kernel(const uint8_t *__restrict__ res, const uint64_t *__restrict__ off,
                                                         const uint32_t *__restrict__ len, uint32_t k0, uint32_t k1,
                                                         int word_len, int base, int nt, uint32_t *__restrict__ wcode,
                                                         uint16_t *__restrict__ wmult, uint32_t *__restrict__ wcnt) {
    constexpr uint32_t kStage = (15 + SLOTS / 2 + 10 + 15) / 16 + 1;   // 16-byte pieces of the longest sequence, shifted
    constexpr uint32_t kLoads = (kStage + 63) / 64;
    __shared__ uint4 table4[4][SLOTS / 4];
    __shared__ uint4 stage[4][kStage];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t k = k0 + blockIdx.x * 4u + wv;
    if (k >= k1) return;                                    // (wave-uniform; the waves of a workgroup share nothing)
    const uint64_t o = off[k];
    const uint32_t L = len[k], nw = L - (uint32_t)word_len + 1u;
    const uint8_t *s = res + o;
    const uint8_t *g = reinterpret_cast<const uint8_t *>(reinterpret_cast<uintptr_t>(s) & ~uintptr_t(15));
    const uint32_t sh = (uint32_t)(s - g), nv = (sh + L + 15u) / 16u;
    uint4 v[kLoads];
#pragma unroll
    for (uint32_t t = 0; t < kLoads; ++t) {
        v[t] = make_uint4(0, 0, 0, 0);
        if (lane + 64u * t < nv && lane + 64u * t < kStage) v[t] = reinterpret_cast<const uint4 *>(g)[lane + 64u * t];
    }
    uint32_t *tb = reinterpret_cast<uint32_t *>(table4[wv]);
#pragma unroll
    for (uint32_t t = 0; t < SLOTS / 4 / 64; ++t) table4[wv][lane + 64u * t] = make_uint4(kSentinel, kSentinel, kSentinel, kSentinel);
#pragma unroll
    for (uint32_t t = 0; t < kLoads; ++t) if (lane + 64u * t < kStage) stage[wv][lane + 64u * t] = v[t];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const uint8_t *l = reinterpret_cast<const uint8_t *>(stage[wv]) + sh;
    for (uint32_t i = lane; i < nw; i += 64u) {
        uint32_t key = 0;
        bool bad = false;
        for (int t = 0; t < word_len; ++t) { const uint32_t r = l[i + t]; key = key * (uint32_t)base + r; bad |= r >= (uint32_t)base; }
        if (nt && bad) continue;                          // words containing N are skipped
        uint32_t slot = (key * 0x9E3779B1u) >> 7 & (SLOTS - 1u);
        for (;;) {                                         // (at most half the slots are ever taken)
            const uint32_t cur = tb[slot];
            if (cur != kSentinel && (cur >> 10) == key) { atomicAdd(&tb[slot], 1u); break; }
            if (cur == kSentinel) {
                const uint32_t was = atomicCAS(&tb[slot], kSentinel, key << 10 | 1u);
                if (was == kSentinel) break;
                if ((was >> 10) == key) { atomicAdd(&tb[slot], 1u); break; }
            }
            slot = (slot + 1u) & (SLOTS - 1u);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    uint32_t n_out = 0;
#pragma unroll 4
    for (uint32_t r = 0; r < SLOTS / 64u; ++r) {
        const uint32_t e = tb[r * 64u + lane];
        const bool occ = e != kSentinel;
        const unsigned long long m = __ballot(occ);
        if (occ) {
            const uint32_t at = n_out + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            wcode[o + at] = e >> 10;
            wmult[o + at] = (uint16_t)(e & 1023u);
        }
        n_out += (uint32_t)__popcll(m);
    }
    if (lane == 0) wcnt[k] = n_out;
}

// Sequences of more than 32768 words (rare: a handful of giant proteins): the same open-addressing table in a GLOBAL
// scratch region of the workgroup (`slots` a power of two >= 2 x words, keys and counts), one workgroup per sequence.
__global__ __launch_bounds__(1024) void words_huge_kernel(const uint8_t *__restrict__ res, const uint64_t *__restrict__ off,
                                                          const uint32_t *__restrict__ len, uint32_t k0, uint32_t k1,
                                                          int word_len, int base, int nt, uint32_t *__restrict__ scratch,
                                                          uint32_t slots, uint32_t *__restrict__ wcode,
                                                          uint16_t *__restrict__ wmult, uint32_t *__restrict__ wcnt,
                                                          uint32_t *__restrict__ err) {
    __shared__ uint32_t part[1024];
    const uint32_t k = k0 + blockIdx.x;
    if (k >= k1) return;
    const uint32_t tid = threadIdx.x;
    uint32_t *hk = scratch + (size_t)blockIdx.x * 2u * slots, *hc = hk + slots;
    const uint64_t o = off[k];
    const uint8_t *s = res + o;
    const uint32_t nw = len[k] - (uint32_t)word_len + 1u;
    for (uint32_t i = tid; i < slots; i += 1024) { hk[i] = kSentinel; hc[i] = 0u; }
    __threadfence();
    __syncthreads();
    for (uint32_t i = tid; i < nw; i += 1024) {
        uint32_t key = 0;
        bool bad = false;
        for (int t = 0; t < word_len; ++t) { key = key * (uint32_t)base + s[i + t]; bad |= s[i + t] >= base; }
        if (nt && bad) continue;
        uint32_t slot = ((key * 0x9E3779B1u) >> 7) & (slots - 1u);
        for (;;) {
            const uint32_t was = atomicCAS(&hk[slot], kSentinel, key);
            if (was == kSentinel || was == key) break;
            slot = (slot + 1u) & (slots - 1u);
        }
        atomicAdd(&hc[slot], 1u);
    }
    __threadfence();
    __syncthreads();
    const uint32_t per = slots / 1024u, beg = tid * per;     // (slots >= 65536)
    uint32_t mine = 0;
    for (uint32_t i = beg; i < beg + per; ++i) mine += __builtin_nontemporal_load(&hk[i]) != kSentinel;
    part[tid] = mine;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const uint32_t v = tid >= d ? part[tid - d] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t idx = part[tid] - mine;
    if (tid == 1023) wcnt[k] = part[tid];
    for (uint32_t i = beg; i < beg + per; ++i) {
        const uint32_t key = __builtin_nontemporal_load(&hk[i]);
        if (key == kSentinel) continue;
        const uint32_t c = __builtin_nontemporal_load(&hc[i]);
        if (c > 65535u) atomicOr(err, (uint32_t)E_WORDMULT);   // (multiplicities are kept in 16 bits)
        wcode[o + idx] = key; wmult[o + idx] = (uint16_t)c; ++idx;
    }
}

// ----------------------------------------------------------------------------------------
// the word index: one 256-byte line per word code, appended to when representatives are created
// ----------------------------------------------------------------------------------------
// The structure the sequential rule itself walks (its "word table": per word the representatives
// that have it), kept for the whole call and grown in place:
//   line[code] = { len, ovf, round, pending, e[59] }   (one 256-byte line in four 64-byte pieces: ONE memory transaction
//                gives a query the list's length and its first 11 entries; the further pieces -- 16 entries each --
//                are read only when a list reaches them. [64-byte lines with 11 entries until round 3, then 128 bytes
//                with 27: a 4000-genome index has 28 entries per list on average, and what is in the overflow pool
//                costs a 6-step search per entry; 128 -> 256 bytes: cfg-4 1117 -> 1079 ms, cfg-3s unchanged.]
//   entry      = sorted sequence index of the representative (low bits) | min(multiplicity of the word in
//                it, field maximum) above them; the field maximum means "look it up in its word list"
//   overflow   = entries 59.. live in pool[ovf + 1 ..], a contiguous array of capacity pool[ovf]
//                that is re-allocated with twice the need when it fills (bump allocation)
//   round      = epoch << 32 | ~len_prev: the latest append round that touched the list and the list's
//                length before that round, so that a pass can visit exactly the entries a round added (new
//                representatives of the window). One 64-bit atomicMax per appended entry maintains it:
//                a newer round beats an older one, and within a round the smallest position wins.
// Entries keep no order (candidates are ordered by an explicit key).
constexpr uint32_t kLineBytes = 256;
constexpr uint32_t kInline = (kLineBytes - 20) / 4;   // entries in the line (59)
constexpr uint32_t kInlineA = 11;    // ... of which in its first 64 bytes
struct __attribute__((aligned(kLineBytes))) IndexLine {
    uint32_t len, ovf;
    unsigned long long round;
    uint32_t pending;
    uint32_t e[kInline];
};
static_assert(sizeof(IndexLine) == kLineBytes && (kInline - kInlineA) % 16 == 0, "one line per word code: 64-byte pieces");
__device__ __forceinline__ uint32_t line_len_prev(const IndexLine &L) { return ~(uint32_t)L.round; }

// Appending the representatives list[*d_lo .. *d_hi) as round `epoch`. 94 % of the lists a protein query meets
// fit their line, so ONE pass over the new representatives' word lists places those entries (position from
// the length counter, round bookkeeping by the atomicMax above, the code marked in the round's bit map);
// entries that fall behind the line are set aside (code, position, entry) and placed by two short passes
// over just those: room first, then the entries. [Three passes over all words -- count, grow, write --
// before: 11.2 -> 7.5 ms per step on cfg-3s.]
// The round's map of touched codes has one bit per code AND SEGMENT of the window (kSegs equal ranges of member
// positions; the kSegs bits of a code are adjacent: eight codes per 32-bit word): bit j of a code is set when one of
// the round's representatives in segments 0..j has the word, and a member of segment j probes bit j. A member only
// ever visits entries of representatives BEFORE it, all of which are in its own segment or an earlier one; with one
// bit per code half of the marked words of a member (a line read each) lead to later representatives only -- with
// kSegs bits an eighth. Setting the bits of segments seg..kSegs-1 is ONE atomicOr, probing one load, as before.
// [filter<new> 31.2 -> 22.1 ms per step on cfg-3s together with filling the exact table directly from every slab.]
constexpr uint32_t kSegs = 4;
__device__ __forceinline__ uint32_t seg_of(uint32_t ql, uint32_t nbq) { return (uint32_t)(((uint64_t)ql * kSegs) / nbq); }
struct Deferred { uint32_t code, pos, entry, pad; };
__global__ __launch_bounds__(256) void index_append_kernel(DevSeqs S, const uint32_t *__restrict__ list,
                                                          const uint32_t *__restrict__ d_lo,
                                                          const uint32_t *__restrict__ d_hi,
                                                          IndexLine *__restrict__ lines, uint32_t epoch,
                                                          uint32_t *__restrict__ newbits, uint32_t b0, uint32_t nbq,
                                                          Deferred *__restrict__ deferred,
                                                          uint32_t *__restrict__ n_deferred, uint32_t deferred_cap,
                                                          uint32_t *__restrict__ err) {
    const uint32_t lane = threadIdx.x & 63u, lo = *d_lo, hi = *d_hi, fmax = entry_fmax(S);
    for (uint32_t w = lo + blockIdx.x * 4 + (threadIdx.x >> 6); w < hi; w += gridDim.x * 4) {
        const uint32_t k = list[w];
        const uint64_t o = S.off[k];
        const uint32_t nw = S.wcnt[k], seg = seg_of(k - b0, nbq);
        for (uint32_t i = lane; i < nw; i += 64) {
            const uint32_t code = S.wcode[o + i], m = S.wmult[o + i];
            IndexLine &L = lines[code];
            const uint32_t pos = atomicAdd(&L.len, 1u);
            atomicMax(&L.round, ((unsigned long long)epoch << 32) | (uint32_t)~pos);
            {
                const uint32_t mask = ((0xFu << seg) & 0xFu) << ((code & 7u) * kSegs);
                uint32_t *bw = newbits + (code >> 3);
                if ((*bw & mask) != mask) atomicOr(bw, mask);
            }
            const uint32_t entry = k | ((m < fmax ? m : fmax) << S.mshift);
            if (pos < kInline) { L.e[pos] = entry; continue; }
            const uint32_t t = atomicAdd(n_deferred, 1u);
            if (t < deferred_cap) deferred[t] = Deferred{code, pos, entry, 0u}; else atomicOr(err, (uint32_t)E_TOUCHED);
        }
    }
}
// room for the entries set aside: the first of a code's entries to arrive sizes the overflow array from
// the list's final length (copying what the list had there before the round)
__global__ __launch_bounds__(256) void index_grow_kernel(IndexLine *__restrict__ lines, uint32_t *__restrict__ pool,
                                                        uint32_t *__restrict__ pool_used, uint32_t pool_cap,
                                                        const Deferred *__restrict__ deferred,
                                                        const uint32_t *__restrict__ n_deferred, uint32_t deferred_cap,
                                                        uint32_t *__restrict__ err) {
    const uint32_t n = min(*n_deferred, deferred_cap);
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        IndexLine &L = lines[deferred[t].code];
        if (atomicExch(&L.pending, 1u) != 0u) continue;
        const uint32_t novf = L.len - kInline, old = L.ovf, cap = old ? pool[old] : 0u;
        if (novf <= cap) continue;
        const uint32_t ncap = max(2u * novf, 32u);
        const uint32_t idx = atomicAdd(pool_used, ncap + 1u);
        if ((uint64_t)idx + ncap + 1u > pool_cap) { atomicOr(err, (uint32_t)E_POOL); continue; }
        pool[idx] = ncap;
        const uint32_t before = line_len_prev(L);
        const uint32_t have = before > kInline ? before - kInline : 0u;
        for (uint32_t i = 0; i < have; ++i) pool[idx + 1u + i] = pool[old + 1u + i];
        L.ovf = idx;
    }
}
// (It also clears the map of touched codes the NEXT append round will use -- there are two, used alternately; the one
// cleared here was last read by the pass over the round before this one, which is complete -- instead of a launch of its own.)
__global__ __launch_bounds__(256) void index_place_kernel(IndexLine *__restrict__ lines, uint32_t *__restrict__ pool,
                                                         const Deferred *__restrict__ deferred,
                                                         const uint32_t *__restrict__ n_deferred, uint32_t deferred_cap,
                                                         uint4 *__restrict__ next_map, uint32_t map_vec4) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < map_vec4; i += gridDim.x * blockDim.x)
        next_map[i] = make_uint4(0u, 0u, 0u, 0u);
    const uint32_t n = min(*n_deferred, deferred_cap);
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        const Deferred d = deferred[t];
        IndexLine &L = lines[d.code];
        pool[L.ovf + 1u + (d.pos - kInline)] = d.entry;
        L.pending = 0u;
    }
}

// ----------------------------------------------------------------------------------------
// filter: the short-word counting filter, query-major over the word index
// ----------------------------------------------------------------------------------------
// One wave per window slot (a query strand). The wave walks the query's distinct words, one word per
// lane and 64-byte line per word, and counts min(mult_q, mult_r) per representative r < q:
//   pass 1  counts into hashed LDS buckets (one LDS atomic per posting entry; 256 to 1024 per wave). A bucket sums the
//           counts of every representative that hashes to it, so it can only over-count: a
//           representative whose bucket stays below the query's threshold is certainly no candidate.
//           Nearly every entry a query meets is a chance hit of a single word and ends here.
//   pass 2  (only when some bucket reached the threshold) walks the lines again -- they are L2
//           hits now -- and accumulates, exactly, count and smallest shared code for the
//           representatives of hot buckets in a small LDS hash table (compare-and-swap insertion).
//           If that table overflows, the representatives are split by residue class of a second
//           hash and the pass is repeated per class, refined until every class fits.
// Candidates reaching the threshold are emitted as pair records, pruned by the query's current
// best key. NEWONLY visits only the entries appended in round `epoch` (the window's new
// representatives; the round's bit map of touched codes saves the line reads elsewhere).
// Every visited entry with r < q counts as a posting visit of the sequential rule, whatever the
// query's state; final queries (`done`) only count.
constexpr int kFB = 1024;          // buckets per wave
constexpr int kFH = 256;           // exact table slots per wave (proteins)
constexpr int kFHNt = 2048;        // ... nucleotides: one shared word makes a candidate, so a query's table holds most representatives it meets
constexpr uint32_t kFProbe = 24;   // probes before the exact table counts as full
constexpr uint32_t kEmpty = 0xFFFFFFFFu;
constexpr int kFWork = 32;         // residue-class work list per wave
constexpr unsigned long long kNoBest = ~0ull;

struct FilterArgs {
    const IndexLine *lines;
    const uint32_t *pool;
    const uint32_t *newbits;                  // NEWONLY: kSegs bits per code (see index_append_kernel)
    const uint32_t *d_round_lo, *d_round_hi;  // NEWONLY: the round's representatives list[lo, hi): nothing to do when empty
    uint32_t epoch;
    uint32_t b0, nbq, ns;
    uint32_t shard_index, shard_count;
    const int32_t *req_aan;
    const unsigned long long *best;
    const uint8_t *done;
    Pair *pairs;
    uint32_t *n_pairs;
    uint32_t pair_cap;
    unsigned long long *visits, *rc_visits;
    uint32_t *err;
    // block mode: the queries are the block's members qlist[0 .. *d_nq) instead of the whole window; a member
    // that gets a candidate is marked; visits are not counted (the entries are tentative)
    const uint32_t *qlist, *d_nq;
    uint8_t *mark;
    uint32_t count_visits;
    uint32_t lean;            // the caller wants no work counters: members that cannot gain from a pass are not walked
};

// multiplicity of `code` in the word list of sequence r (present by construction of the index);
// rare (the word is repeated in the query AND in the representative): a scan, kept out of line
__device__ __noinline__ uint32_t word_mult_of(const uint32_t *__restrict__ wcode, const uint16_t *__restrict__ wmult,
                                              uint64_t o, uint32_t n, uint32_t code) {
    for (uint32_t i = 0; i < n; ++i)          // (word lists keep no order)
        if (wcode[o + i] == code) return wmult[o + i];
    return 1u;
}

__device__ __forceinline__ void wave_lds_sync() {   // LDS writes of this wave's lanes visible to all its lanes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

struct FilterWave {   // a wave's private LDS
    uint32_t *bucket, *hrep, *hcnt, *hminc;
    uint4 *lq;
    uint32_t *pref;
};

// exact table: slot of representative r (inserted if absent), or FH when the table is full
template <int FH>
__device__ __forceinline__ uint32_t exact_slot(uint32_t *hrep, uint32_t r) {
    uint32_t h = ((r * 0xC2B2AE35u) >> 8) & (uint32_t)(FH - 1);
    for (uint32_t probe = 0; probe < kFProbe; ++probe) {
        const uint32_t cur = hrep[h];
        if (cur == r) return h;
        if (cur == kEmpty) {
            const uint32_t was = atomicCAS(&hrep[h], kEmpty, r);
            if (was == kEmpty || was == r) return h;
        }
        h = (h + 1u) & (uint32_t)(FH - 1);
    }
    return (uint32_t)FH;
}

// One walk over the words [0, nw) at `o` of a query strand. PASS2 = false: count into the buckets
// (`visits`, `hot`); true: accumulate the representatives of hot buckets whose residue class is
// (class_k, class_j) in the exact table (`full` when it overflows).
// Development aid (make EXTRA=-DPGX_FTIME): shader-clock time per section of the filter kernel, summed over waves.
#ifdef PGX_FTIME
__device__ unsigned long long g_ftime[16];
struct FTimer {
    unsigned long long t, acc[12];
    __device__ void start() { t = clock64(); }
    __device__ void lap(int i) { const unsigned long long n = clock64(); acc[i] += n - t; t = n; }
};
#define FT_LAP(ft, i) (ft).lap(i)
#else
struct FTimer { __device__ void start() {} };
#define FT_LAP(ft, i) ((void)0)
#endif

// NEWONLY: the query's words whose code the round touched, compacted (one per lane); `complete` when these are all
// of them, so that the exact pass need not walk the word list again
struct Marked { uint32_t code, mult, n; bool complete; };

template <bool NEWONLY, bool PASS2, int FH, int FB>
__device__ __forceinline__ void filter_walk(const DevSeqs &S, const FilterArgs &A, const FilterWave &W, uint32_t lane,
                                            uint64_t o, uint32_t nw, uint32_t q, uint32_t thr, bool count_only,
                                            uint32_t class_k, uint32_t class_j, uint32_t &visits, bool &hot,
                                            bool &full, Marked &M, FTimer &ft, uint32_t seg, bool redo = false) {
    const uint32_t rmask = entry_rmask(S), fmax = entry_fmax(S);
    // NEWONLY, first walk: the representatives met go straight into the exact table -- no bucket pass, no second
    // walk (a round adds few entries to a query's lists; with the segment bits a member of ~1000 residues meets some
    // 140 representatives of a round, which the 256 slots hold). If the table overflows,
    // the walk is redone (`redo`) with the buckets, visits not counted again, and the exact passes per residue
    // class follow as for any other walk. [Direct only for members whose marked words fit ONE slab, 256 slots,
    // before: the quarter of the members beyond that paid a bucket walk and a whole second walk -- a third of the
    // kernel's time.]
    const bool direct = NEWONLY && !PASS2 && !redo;
    auto entry_visit = [&](uint32_t entry, uint32_t code, uint32_t mq) {
        const uint32_t r = entry & rmask;
        if (r >= q) return;                      // only representatives created before the query
        uint32_t c = 1u;
        if (mq > 1u) {
            uint32_t mr = entry >> S.mshift;
            if (mr == fmax) mr = word_mult_of(S.wcode, S.wmult, S.off[r], S.wcnt[r], code);
            c = mr < mq ? mr : mq;
        }
        constexpr int kFBBits = FB == 1024 ? 10 : (FB == 512 ? 9 : 8);
        static_assert(FB == (1 << kFBBits), "bucket count");
        const uint32_t b = (r * 0x9E3779B1u) >> (32 - kFBBits);
        if (!PASS2) {
            if (!redo) ++visits;
            if (count_only) return;
            if (direct) {
                const uint32_t h = exact_slot<FH>(W.hrep, r);
                if (h == (uint32_t)FH) { full = true; return; }
                atomicAdd(&W.hcnt[h], c);
                atomicMin(&W.hminc[h], code);
                return;
            }
            if (thr == 1u) { hot = true; return; }    // one shared word makes a candidate: nothing to bound
            atomicAdd(&W.bucket[b], c);             // (no result needed here: `hot` comes from one look at the buckets
                                                    // after the walk -- a returning atomic per visit made every visit wait
                                                    // for the one before, 130 per lane on a 4000-genome index)
        } else {
            if (thr > 1u && W.bucket[b] < thr) return;
            if (((((r ^ (r >> 15)) * 0x85EBCA6Bu) >> 9) & (class_k - 1u)) != class_j) return;
            const uint32_t h = exact_slot<FH>(W.hrep, r);
            if (h == (uint32_t)FH) { full = true; return; }
            atomicAdd(&W.hcnt[h], c);
            atomicMin(&W.hminc[h], code);
        }
    };
    // One slab: up to 64 words (one per lane), their lines, the entries.
    auto slab = [&](uint32_t code, uint32_t mq, bool live) {
        uint4 la = make_uint4(0u, 0u, 0u, 0u), lb = la, lc = la, ld = la;
        if (live) {
            const uint4 *lp = reinterpret_cast<const uint4 *>(A.lines + code);
            la = lp[0];
            if (!NEWONLY) { lb = lp[1]; lc = lp[2]; ld = lp[3]; }   // (a round's pass reads the few entries it visits where they are)
        }
        // line = { len, ovf, ~len_prev, epoch | pending, e0, e1, e2 | e3..e6 | e7..e10 }
        const uint32_t hi = la.x;
        const uint32_t lo = NEWONLY ? (la.w == A.epoch ? ~la.z : hi) : 0u;
        const uint32_t hi_in = hi < kInlineA ? hi : kInlineA;    // (first half of the line)
        // the lanes start at different entries of their lines: lists keep insertion order, so in a family (or with
        // few codes) entry j of every lane's list is the same representative -- 64 atomics on one LDS address
        const uint32_t n_in = hi_in > lo ? hi_in - lo : 0u;
        if (!NEWONLY) {
            // The pass over the whole index visits every inline entry: entry t of every lane in step t, each straight
            // from its register (picking entry j of a line by ten selects was a third of this pass's instructions,
            // which is what bounds it on a 4000-genome index: 12.8 k vector instructions per member for 17 k visits).
            // The four lane groups take the entries in different rotations, so that a family's representative -- the
            // same entry position in many lanes' lists -- meets at most a quarter of the lanes in one LDS atomic.
            const uint32_t grp = lane & 3u;
            if (__ballot(hi_in != 0u)) {
                const uint32_t ents[kInlineA] = {lb.y, lb.z, lb.w, lc.x, lc.y, lc.z, lc.w, ld.x, ld.y, ld.z, ld.w};
#pragma unroll
                for (uint32_t t = 0; t < kInlineA; ++t) {
                    const uint32_t a = ents[t], b = ents[(t + 3u) % kInlineA], c2 = ents[(t + 6u) % kInlineA], d = ents[(t + 9u) % kInlineA];
                    const uint32_t e = grp == 0u ? a : (grp == 1u ? b : (grp == 2u ? c2 : d));
                    const uint32_t idx = grp == 0u ? t : (grp == 1u ? (t + 3u) % kInlineA : (grp == 2u ? (t + 6u) % kInlineA : (t + 9u) % kInlineA));
                    if (idx < hi_in) entry_visit(e, code, mq);
                }
            }
            // (the further 64-byte pieces of the line, 16 entries each: a piece is read when some lane's list reaches it)
            constexpr uint32_t kB = 16;
            for (uint32_t piece = 0; piece < (kInline - kInlineA) / kB; ++piece) {    // (a loop: three copies of the walk do not pay)
                const uint32_t first = kInlineA + piece * kB;
                if (!__ballot(hi > first)) break;
                uint4 h0 = make_uint4(0u, 0u, 0u, 0u), h1 = h0, h2 = h0, h3 = h0;
                if (hi > first) {
                    const uint4 *lp = reinterpret_cast<const uint4 *>(A.lines + code) + 4u * (piece + 1u);
                    h0 = lp[0]; h1 = lp[1]; h2 = lp[2]; h3 = lp[3];
                }
                const uint32_t hi_b = hi > first ? (hi < first + kB ? hi - first : kB) : 0u;
                const uint32_t entsb[kB] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w, h2.x, h2.y, h2.z, h2.w, h3.x, h3.y, h3.z, h3.w};
#pragma unroll
                for (uint32_t t = 0; t < kB; ++t)     // (no rotation here: what a family shares sits at the front of its lists)
                    if (t < hi_b) entry_visit(entsb[t], code, mq);
            }
        } else {
        // A round's entries: a range of the line that starts anywhere, one or two entries as a rule. They are read
        // where they are (the line's header has just been fetched: L2 hits) instead of holding the line's entries in
        // registers and picking by select chains.
        (void)n_in;
        const uint32_t hi_l = hi < kInline ? hi : kInline;
        const uint32_t *le = reinterpret_cast<const uint32_t *>(A.lines + code) + 5;    // e[0]
        for (uint32_t jb = lo; jb < hi_l; ++jb) entry_visit(le[jb], code, mq);
        }
        // lists longer than the line: their pool parts are flattened into one run of entries that the whole
        // wave walks, one entry per lane and step, whatever the lists' lengths (a list of one new entry
        // and a list of thousands cost what their entries cost)
        const bool longl = hi > kInline && hi > lo;
        bool flatten = __ballot(longl) != 0ull;
        if (NEWONLY && flatten) {
            // a round adds one or two entries to a list, wherever the list ends: when no lane of the slab has more than
            // four behind the line, each lane reads its own in place (on a 4000-genome index nearly every slab has lists
            // that end in the pool, and the flattened walk's set-up -- a prefix scan, LDS records, a search per entry --
            // was paid for a handful of entries)
            const uint32_t from = lo > kInline ? lo : kInline;
            const uint32_t cnt_l = longl ? hi - from : 0u;
            if (!__ballot(cnt_l > 4u)) {
                const uint32_t *pp = A.pool + la.y + 1u + (from - kInline);
                for (uint32_t j2 = 0; j2 < cnt_l; ++j2) entry_visit(pp[j2], code, mq);
                flatten = false;
            }
        }
        if (flatten) {
            const uint32_t from = lo > kInline ? lo : kInline;
            const uint32_t cnt_l = longl ? hi - from : 0u;
            uint32_t incl = cnt_l;
            for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d); if ((int)lane >= d) incl += y; }
            const uint32_t total = __shfl(incl, 63);
            W.lq[lane] = make_uint4(la.y + 1u + (from - kInline), incl - cnt_l, code, mq);
            W.pref[lane] = incl;
            wave_lds_sync();
            // (kPoolAhead entries per lane are located and LOADED before any is visited: a list of a 4000-genome index
            // has 25 entries, 14 of them in the pool, and one dependent pool read per visit made the pass over the
            // whole index half of the run on cfg-4)
            constexpr uint32_t kPoolAhead = NEWONLY ? 1 : 2;   // (the round passes have no registers to spare: one)
            for (uint32_t t0 = lane; t0 < total; t0 += 64u * kPoolAhead) {
                uint32_t ent[kPoolAhead], ecode[kPoolAhead], emq[kPoolAhead];
#pragma unroll
                for (uint32_t u = 0; u < kPoolAhead; ++u) {
                    const uint32_t t = t0 + 64u * u;
                    ent[u] = 0u; ecode[u] = 0u; emq[u] = 0u;     // (mq == 0: no entry)
                    if (t < total) {
                        uint32_t a = 0, b = 63;                // the list that holds entry t: first prefix > t
                        while (a < b) { const uint32_t mid = (a + b) >> 1; if (W.pref[mid] > t) b = mid; else a = mid + 1; }
                        const uint4 it = W.lq[a];
                        ent[u] = A.pool[it.x + (t - it.y)]; ecode[u] = it.z; emq[u] = it.w;
                    }
                }
#pragma unroll
                for (uint32_t u = 0; u < kPoolAhead; ++u)
                    if (emq[u]) entry_visit(ent[u], ecode[u], emq[u]);
            }
            __builtin_amdgcn_wave_barrier();
        }
    };
    auto load_word = [&](uint32_t w0, uint32_t &c, uint32_t &m) {   // (a multiplicity of 0 marks a lane without a word)
        const uint32_t w = w0 + lane;
        c = 0u; m = 0u;
        if (w < nw) { c = S.wcode[o + w]; m = S.wmult[o + w]; }
    };
    if (!NEWONLY) {
        // every word has a list to walk: the word loads run one slab ahead of the slab being processed
        uint32_t c1, m1;
        load_word(0u, c1, m1);
        for (uint32_t w0 = 0; w0 < nw; w0 += 64) {
            const uint32_t code = c1, mq = m1;
            load_word(w0 + 64u, c1, m1);
            slab(code, mq, mq != 0u);
        }
        return;
    }
    // NEWONLY: only the words whose code the round touched (its bit map) have anything to visit -- a few per
    // slab. The waves of this kernel wait for memory nearly all of their time with about one load in flight, so
    // the loads of up to 512 words and then all their bit-map probes are issued together, the marked words are
    // compacted across the slabs in registers (a rotation of the lanes by ds_permute: a slab's marked words go
    // behind the `np` words already pending, the others behind those), and a slab of lines is only walked once
    // 64 marked words are together, and for what is left at the end.
    if ((PASS2 || redo) && M.complete) { slab(M.code, M.mult, lane < M.n); FT_LAP(ft, 6); return; }
    uint32_t pc = 0u, pm = 0u, np = 0u;
    bool flushed = false;
    for (uint32_t base = 0; base < nw; base += 256u) {
        uint32_t c[4], m[4], nbw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) load_word(base + 64u * j, c[j], m[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) { nbw[j] = 0u; if (m[j]) nbw[j] = A.newbits[c[j] >> 3]; }
#ifdef PGX_FTIME
        if (!PASS2) { if (__ballot(nbw[0] == 0x12345u && nbw[3] == 0x54321u) == ~0ull) return; FT_LAP(ft, 1); }   // (waits for the probes)
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (base + 64u * j >= nw) break;
            const bool live = (nbw[j] >> ((c[j] & 7u) * kSegs + seg)) & 1u;
            const unsigned long long mask = __ballot(live);
            if (!mask) continue;
            const uint32_t n = (uint32_t)__popcll(mask);
            const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            const uint32_t tgt = live ? np + below : np + n + (lane - below);       // a bijection of the lanes (mod 64)
            const uint32_t xc = (uint32_t)__builtin_amdgcn_ds_permute((int)(tgt << 2), (int)c[j]);
            const uint32_t xm = (uint32_t)__builtin_amdgcn_ds_permute((int)(tgt << 2), (int)(live ? m[j] : 0u));
            if (np + n >= 64u) {      // lanes [np, 64) complete the pending slab; lanes [0, np + n - 64) hold the rest
                slab(lane >= np ? xc : pc, lane >= np ? xm : pm, true);
                flushed = true;
                np = np + n - 64u;
                pc = xc; pm = lane < np ? xm : 0u;
            } else {
                if (lane >= np && lane < np + n) { pc = xc; pm = xm; }
                np += n;
            }
        }
    }
    if (!PASS2) FT_LAP(ft, 2);
    if (np) slab(pc, pm, lane < np);
    if (!PASS2) { M.code = pc; M.mult = pm; M.n = np; M.complete = !flushed; FT_LAP(ft, 3); }
    else FT_LAP(ft, 6);
}

template <bool NT, bool NEWONLY>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NT ? 1 : 5, 8))) void filter_kernel(DevSeqs S, FilterArgs A) {
    // FIVE waves per SIMD (96 VGPRs, <= 30.7 KB of LDS per workgroup): the waves of these kernels wait for memory 72 % of
    // their cycles and a wave's loads return in order, so occupancy is what hides the waits (four -> five waves: round
    // passes 22.0 -> 18.7 ms per step; six would spill 21 registers). Passes over a round's entries (proteins): the exact
    // table first (256 slots), 256 buckets only for what overflows it (thresholds of long members are far above what
    // chance hits add up to); the pass over the whole index: 512 buckets. (The nucleotide instantiations, whose exact
    // table is 24 KB per wave, run one wave per SIMD.)
    constexpr int FB = NT ? kFB : (NEWONLY ? 256 : 512);
    constexpr int FH = NT ? kFHNt : kFH;
    __shared__ __attribute__((aligned(16))) uint32_t s_bucket[4][FB];
    __shared__ uint32_t s_hrep[4][FH], s_hcnt[4][FH], s_hminc[4][FH];
    __shared__ uint4 s_lq[4][64];
    __shared__ uint32_t s_pref[4][64];
    __shared__ uint2 s_work[4][kFWork];
    __shared__ unsigned long long s_visits;
    __shared__ uint4 s_stage[4][64];
    if (NEWONLY && *A.d_round_lo >= *A.d_round_hi) return;
    if (A.qlist && blockIdx.x * 4u >= (A.ns > A.nbq ? 2u : 1u) * *A.d_nq) return;   // (block mode: few members)
    // (wave-uniform by construction; saying so keeps what depends on it -- the member's slot, its state and offsets -- in
    // scalar registers and scalar loads)
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;
    if (threadIdx.x == 0) s_visits = 0ull;
    __syncthreads();
    const FilterWave W{s_bucket[wave], s_hrep[wave], s_hcnt[wave], s_hminc[wave], s_lq[wave], s_pref[wave]};
    uint2 *work = s_work[wave];
    for (uint32_t i = lane; i < (uint32_t)FB; i += 64) W.bucket[i] = 0u;
    for (uint32_t i = lane; i < (uint32_t)FH; i += 64) { W.hrep[i] = kEmpty; W.hcnt[i] = 0u; W.hminc[i] = kSentinel; }
    wave_lds_sync();

    const uint32_t n_list = A.qlist ? *A.d_nq : 0u;
    const uint32_t n_slots = A.qlist ? (A.ns > A.nbq ? 2u * n_list : n_list) : A.ns;
    unsigned long long wave_visits = 0ull;   // lane 0: posting visits of this wave's members
    uint4 *stage = s_stage[wave];
    uint32_t n_stage = 0;                    // staged pair records of this wave (wave-uniform)
    auto flush_pairs = [&]() {
        if (n_stage == 0u) return;
        wave_lds_sync();
        uint32_t base = 0u;
        if (lane == 0) base = atomicAdd(A.n_pairs, n_stage);
        base = __shfl(base, 0);
        if (lane < n_stage && base + lane < A.pair_cap) {
            const uint4 v = stage[lane];
            Pair p;
            p.q = v.x; p.r = v.y; p.cnt = v.z; p.minc = v.w;
            p.best_sum = 0; p.band_left = p.band_center = p.band_right = 0; p.iden = 0; p.flags = 0;
            A.pairs[base + lane] = p;
        }
        __builtin_amdgcn_wave_barrier();
        n_stage = 0u;
    };
    FTimer ft{};
    ft.start();
    // A member's state and offsets: five independent loads that everything else waits for. The waves of this kernel
    // are chains of dependent memory round trips (state -> words -> map probes -> lines -> best key), four waves per
    // SIMD to hide them behind: the NEXT member's state is fetched while the current member is walked, and the best
    // key (read when candidates are emitted) comes with it.
    struct Member { uint32_t ql, nw; int32_t t0; uint64_t o; unsigned long long best; uint8_t done; bool valid, rstrand; };
    auto load_member = [&](uint32_t s) -> Member {
        Member m{};
        if (s >= n_slots) return m;
        // window slot -> member, strand (block mode: slots enumerate the block list, then its reverse strands)
        const uint32_t half = A.qlist ? n_list : A.nbq;
        m.rstrand = s >= half;
        m.ql = A.qlist ? A.qlist[m.rstrand ? s - half : s] - A.b0 : (m.rstrand ? s - half : s);
        if (A.shard_count > 1 && !A.qlist && m.ql % A.shard_count != A.shard_index) return m;   // another process's member
        const uint32_t q = A.b0 + m.ql, k = m.rstrand ? S.n_fwd + q : q;
        m.done = A.done[m.ql];
        m.t0 = A.req_aan[q];
        m.o = S.off[k];
        m.nw = S.wcnt[k];
        m.best = NEWONLY ? A.best[m.ql] : kNoBest;
        m.valid = m.done != 2;                            // (2: placed by an earlier chunk's sweep, memory-chunked rule)
        return m;
    };
    const uint32_t s_step = gridDim.x * 4;
    Member nxt = load_member(blockIdx.x * 4 + wave);
    for (uint32_t s = blockIdx.x * 4 + wave; s < n_slots; s += s_step) {
        const Member me = nxt;
        nxt = load_member(s + s_step);
        if (!me.valid) continue;
        const bool rstrand = me.rstrand;
        const uint32_t ql = me.ql;
        const uint32_t q = A.b0 + ql;                     // the query (real sequence); candidates are r < q
        const uint32_t k = rstrand ? S.n_fwd + q : q;     // the strand walked
        const bool count_only = NEWONLY && me.done;
        const uint32_t seg = NEWONLY ? seg_of(ql, A.nbq) : 0u;
        const int32_t t0 = me.t0;
        const uint32_t thr = t0 > 1 ? (uint32_t)t0 : 1u;
        const uint64_t o = me.o;
        const uint32_t nw = me.nw;
        if (NEWONLY && A.lean) {
            // Without the counters of the sequential rule a pass over a round's new representatives owes a member
            // only the candidates that can still beat its best key (strand | smallest shared word | representative:
            // what `emit` lets through). A final member has none. A member that holds a best key can only be beaten
            // through a word whose code does not exceed the key's -- in a family that is the member's one or two
            // smallest words -- so its codes are scanned (a stream) and the round's map is asked about those few; only
            // if one of them is marked is the member walked. [More than half of the members have their
            // representative after the pass over the whole index: the round passes walked them all for the counters.]
            if (count_only) continue;
            if (me.best != kNoBest) {
                const bool best_rc = (me.best >> 63) != 0ull;
                if (rstrand && !best_rc) continue;                    // a reverse-strand candidate never beats a forward one
                if (rstrand == best_rc) {
                    const uint32_t mb = (uint32_t)(me.best >> 32) & 0x7FFFFFFFu;
                    bool any = false;
                    for (uint32_t w0 = 0; w0 < nw; w0 += 64) {
                        const uint32_t w = w0 + lane;
                        if (w >= nw) continue;
                        const uint32_t c = S.wcode[o + w];
                        if (c <= mb) any |= ((A.newbits[c >> 3] >> ((c & 7u) * kSegs + seg)) & 1u) != 0u;
                    }
                    if (!__ballot(any)) continue;
                }
            }
        }
        uint32_t visits = 0;
        bool hot = false, full = false;
        Marked marked{0u, 0u, 0u, false};
#ifdef PGX_FTIME
        if (__ballot(nw == 0xFFFFFFFFu && thr == 0u) == ~0ull) return;   // (consumes the prologue loads)
        FT_LAP(ft, 0);
#endif
        filter_walk<NEWONLY, false, FH, FB>(S, A, W, lane, o, nw, q, thr, count_only, 1u, 0u, visits, hot, full, marked, ft, seg);
        wave_lds_sync();
        for (int d = 32; d > 0; d >>= 1) visits += __shfl_xor(visits, d);
        FT_LAP(ft, 4);
        if (visits && lane == 0 && A.count_visits) {
            if (NT && rstrand) atomicAdd(&A.rc_visits[ql], (unsigned long long)visits);
            else wave_visits += visits;     // (one global atomic per member on ONE address was the floor of a pass)
        }
        // the exact table -> pair records (staged per wave, 64 at a time: one atomic on the pair counter per
        // batch, not per member); `over`: the table overflowed, its content is dropped
        auto emit_table = [&](bool over) {
            for (uint32_t h = lane; h < (uint32_t)FH; h += 64) {
                const uint32_t r = W.hrep[h];
                bool emit = r != kEmpty;
                uint32_t c = 0u, mc = 0u;
                if (emit) {
                    c = W.hcnt[h]; mc = W.hminc[h];
                    W.hrep[h] = kEmpty; W.hcnt[h] = 0u; W.hminc[h] = kSentinel;
                    emit = !over && c >= thr;
                }
                if (NEWONLY && emit) {   // only candidates whose key can still beat the member's current best
                    const unsigned long long key = ((unsigned long long)rstrand << 63) | ((unsigned long long)mc << 32) | r;
                    const unsigned long long bo = me.best;
                    emit = bo == kNoBest || key <= bo;
                }
                const unsigned long long em = __ballot(emit);
                if (!em) continue;
                const uint32_t n_em = (uint32_t)__popcll(em);
                if (n_stage + n_em > 64u) flush_pairs();
                if (emit) {
                    if (A.mark) A.mark[ql] = 1;
                    stage[n_stage + (uint32_t)__popcll(em & ((1ull << lane) - 1ull))] = make_uint4(k, r, c, mc);
                }
                n_stage += n_em;
            }
            wave_lds_sync();
        };
        bool direct_done = NEWONLY && !count_only;   // (wave-uniform)
        if (direct_done && visits) {
            if (__ballot(full)) {
#ifdef PGX_FTIME
                if (lane == 0) atomicAdd(&g_ftime[14], 1ull);
#endif
                emit_table(true);
                full = false; hot = false;
                filter_walk<NEWONLY, false, FH, FB>(S, A, W, lane, o, nw, q, thr, count_only, 1u, 0u, visits, hot, full, marked, ft, seg, true);
                wave_lds_sync();
                direct_done = false;
            } else {
                emit_table(false);
                FT_LAP(ft, 8);
#ifdef PGX_FTIME
                if (lane == 0) atomicAdd(&g_ftime[13], 1ull);
#endif
            }
        }
        if (!direct_done && !count_only && visits && thr > 1u) {   // did any bucket reach the threshold?
            uint32_t mx = 0;
            for (uint32_t i = lane * 4; i < (uint32_t)FB; i += 256) {
                const uint4 v = *reinterpret_cast<const uint4 *>(&W.bucket[i]);
                mx = max(max(mx, v.x), max(max(v.y, v.z), v.w));
            }
            hot = mx >= thr;
        }
#ifdef PGX_FTIME
        if (!direct_done && __ballot(hot) && lane == 0) atomicAdd(&g_ftime[15], 1ull);
#endif
        if (!direct_done && __ballot(hot)) {
            // exact pass per residue class of the hot representatives, refined while the table overflows
            uint32_t n_work = 1;
            if (lane == 0) work[0] = make_uint2(1u, 0u);
            wave_lds_sync();
            while (n_work) {
                const uint2 cls = work[n_work - 1];
                --n_work;
                full = false;
                FT_LAP(ft, 5);
                filter_walk<NEWONLY, true, FH, FB>(S, A, W, lane, o, nw, q, thr, false, cls.x, cls.y, visits, hot, full, marked, ft, seg);
                wave_lds_sync();
                FT_LAP(ft, 7);
                const bool over = __ballot(full) != 0ull;
                if (over) {
                    if (cls.x >= (1u << 20) || n_work + 4 > (uint32_t)kFWork) { if (lane == 0) atomicOr(A.err, (uint32_t)E_TABLE); n_work = 0; }
                    else {
                        if (lane < 4) work[n_work + lane] = make_uint2(cls.x * 4u, cls.y + lane * cls.x);
                        n_work += 4;
                    }
                }
                emit_table(over);
                FT_LAP(ft, 8);
            }
        }
        if (visits && !count_only && !direct_done) {
            for (uint32_t i = lane * 4; i < (uint32_t)FB; i += 256) *reinterpret_cast<uint4 *>(&W.bucket[i]) = make_uint4(0u, 0u, 0u, 0u);
            wave_lds_sync();
        }
        FT_LAP(ft, 9);
    }
    flush_pairs();
#ifdef PGX_FTIME
    FT_LAP(ft, 10);
    if (lane == 0 && NEWONLY && !A.qlist) {
        for (int i = 0; i < 12; ++i) atomicAdd(&g_ftime[i], ft.acc[i]);
        atomicAdd(&g_ftime[12], 1ull);
    }
#endif
    if (lane == 0 && wave_visits) atomicAdd(&s_visits, wave_visits);
    __syncthreads();
    if (threadIdx.x == 0 && s_visits) atomicAdd(A.visits, s_visits);
}

// Tentative entries: a block's members are appended to the index as if all of them were
// representatives, so that the filter finds the block's internal candidate pairs the way it finds all
// others; once the block is decided, the entries of the members that joined a representative are
// struck out again (overwritten by an index no query precedes). One wave per member; the entries sit
// in the part of their lists the round added.
__global__ __launch_bounds__(256) void index_strike_kernel(DevSeqs S, const uint32_t *__restrict__ list, uint32_t n,
                                                          IndexLine *__restrict__ lines, uint32_t *__restrict__ pool) {
    const uint32_t lane = threadIdx.x & 63u, rmask = entry_rmask(S);   // tombstone = rmask: an index no query precedes
    for (uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6); w < n; w += gridDim.x * 4) {
        const uint32_t k = list[w];
        const uint64_t o = S.off[k];
        const uint32_t nw = S.wcnt[k];
        for (uint32_t i = lane; i < nw; i += 64) {
            IndexLine &L = lines[S.wcode[o + i]];
            const uint32_t hi = L.len;
            for (uint32_t j = line_len_prev(L); j < hi; ++j) {
                uint32_t *e = j < kInline ? &L.e[j] : &pool[L.ovf + 1u + (j - kInline)];
                if ((*e & rmask) == k) { *e = rmask; break; }
            }
        }
    }
}

// ----------------------------------------------------------------------------------------
// window state: open members, discovery of certain representatives, blocks
// ----------------------------------------------------------------------------------------
// counters: [0] pairs of the window, [1] pairs of the block, [2] begin of the pair range to evaluate,
// [3] block size, [4] open members, [5] new representatives of the window (list length),
// [6] begin of the current round's segment of that list, [7] touched codes of the round,
// [8] error flag, [9] open members of the round
enum { C_NW = 0, C_NK = 1, C_EVAL0 = 2, C_BLK = 3, C_OPEN = 4, C_NEW = 5, C_SEG0 = 6, C_TOUCH = 7, C_ERR = 8, C_ROUND_OPEN = 9, C_ZERO = 10, C_WIDE = 11, C_ROUND_OPEN2 = 12, C_COUNT = 16 };
constexpr int kSelThreads = 1024;
// Pick the next block: the first `block_cap` window members, in order, that are not final (`done`)
// and have no accepted representative yet. One workgroup; thread t looks at members t, t + 1024, ...
// (tiles of 1024: all of a thread's loads are independent and in flight together -- consecutive members per
// thread made this 75 us of dependent loads), ranks come from per-tile, per-wave counts.
// counters[0] = block size, counters[1] = number of such members in total.
__global__ __launch_bounds__(kSelThreads) void select_block_kernel(const unsigned long long *__restrict__ best,
                                                                  uint8_t *__restrict__ done,
                                                                  uint8_t *__restrict__ inblk, uint32_t b0,
                                                                  uint32_t nb, uint32_t block_cap,
                                                                  uint32_t *__restrict__ blk_list,
                                                                  uint32_t *__restrict__ counters,
                                                                  uint32_t *__restrict__ n_k,
                                                                  uint8_t *__restrict__ hascand_accepted,
                                                                  uint32_t window_cap, uint32_t *__restrict__ base_counters) {
    constexpr uint32_t kTiles = kWindowMax / kSelThreads;      // 64
    __shared__ uint32_t s_pre[kTiles][16];                      // open members of tile j in the waves before wave w
    __shared__ uint32_t s_tile[kTiles + 1];                     // ... in the tiles before tile j
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t i = tid; i < 2 * window_cap / 8; i += kSelThreads) reinterpret_cast<unsigned long long *>(hascand_accepted)[i] = 0ull;
    if (tid == 0) { *n_k = 0u; base_counters[C_TOUCH] = 0u; base_counters[C_EVAL0] = base_counters[C_NW]; }   // (the block's tentative append and pairs begin here)
    const uint32_t n_tiles = (nb + kSelThreads - 1) / kSelThreads;
    unsigned long long open = 0ull;                             // bit j: member j * 1024 + tid is open
#pragma unroll 8
    for (uint32_t j = 0; j < n_tiles; ++j) {
        const uint32_t q = j * kSelThreads + tid;
        if (q >= nb) continue;
        const uint8_t ib = inblk[q];
        uint8_t d = done[q];
        const unsigned long long b = best[q];
        if (ib) { done[q] = 1; inblk[q] = 0; d = 1; }           // retire the previous block
        if (!d && b == kNoBest) open |= 1ull << j;
    }
    for (uint32_t j = 0; j < n_tiles; ++j) {
        const unsigned long long bal = __ballot((open >> j) & 1ull);
        if (lane == 0) s_pre[j][wave] = (uint32_t)__popcll(bal);
    }
    __syncthreads();
    {   // thread (j, w): exclusive prefix over the 16 waves of tile j; the tile's total to s_tile
        const uint32_t j = tid >> 4, w = tid & 15u;
        const uint32_t c = j < n_tiles ? s_pre[j][w] : 0u;
        uint32_t incl = c;
        for (int dd = 1; dd < 16; dd <<= 1) { const uint32_t y = __shfl_up(incl, dd, 16); if ((int)w >= dd) incl += y; }
        __syncthreads();
        if (j < n_tiles) s_pre[j][w] = incl - c;
        if (w == 15u) s_tile[j] = incl;
    }
    __syncthreads();
    if (wave == 0) {   // exclusive prefix over the tiles
        const uint32_t c = lane < n_tiles ? s_tile[lane] : 0u;
        uint32_t incl = c;
        for (int dd = 1; dd < 64; dd <<= 1) { const uint32_t y = __shfl_up(incl, dd); if ((int)lane >= dd) incl += y; }
        s_tile[lane] = incl - c;
        if (lane == 63u) { s_tile[kTiles] = incl; counters[0] = incl < block_cap ? incl : block_cap; counters[1] = incl; }
    }
    __syncthreads();
    for (uint32_t j = 0; j < n_tiles; ++j) {
        const bool is_open = (open >> j) & 1ull;
        const unsigned long long bal = __ballot(is_open);
        if (!is_open) continue;
        const uint32_t rank = s_tile[j] + s_pre[j][wave] + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        if (rank < block_cap) { const uint32_t q = j * kSelThreads + tid; blk_list[rank] = b0 + q; inblk[q] = 1; }
    }
}

// Start of a discovery round: the window's still-open members (not final, no accepted
// representative), and where the round's segment of the new-representative list begins.
// (It also opens the round -- the pair range and list segment that the round's kernels work on begin here -- and
// clears the counter the NEXT round's list will use: two counters, used alternately, instead of a launch of its own.)
__global__ __launch_bounds__(256) void list_open_kernel(const unsigned long long *__restrict__ best,
                                                       const uint8_t *__restrict__ done, uint32_t b0, uint32_t nb,
                                                       uint32_t *__restrict__ ulist, uint32_t *__restrict__ c,
                                                       uint32_t which) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q == 0) {
        c[C_EVAL0] = c[C_NW]; c[C_SEG0] = c[C_NEW]; c[C_TOUCH] = 0u;
        c[which ? C_ROUND_OPEN : C_ROUND_OPEN2] = 0u;
    }
    if (q >= nb) return;
    if (!done[q] && best[q] == kNoBest) ulist[atomicAdd(&c[which ? C_ROUND_OPEN2 : C_ROUND_OPEN], 1u)] = b0 + q;
}
// Discovery of certain representatives in time linear in the open members' words.
//
// A member u can only have an earlier open candidate r if it shares at least its word threshold with
// r; the words it shares with r are among its words that ANY open member of r's part of the window
// has. The window is cut into <= 32 consecutive CHUNKS of bounded word volume (so that the words of
// one chunk cover only a fraction of the code space well below the threshold fraction: what unrelated
// members contribute by chance stays below any threshold, however large the window is), and one table
// is filled from the open members' word lists:
//   first[code][c]        per chunk c: the earliest open member of the chunk that has the word, tagged
//                         with the round's epoch (epoch << 16 | 65535 - member: a plain atomicMax keeps the
//                         earliest member of the newest epoch; no per-round reset). The tags of one code
//                         are adjacent (one record of `stride` words, a 128-byte line for 32 chunks), so the
//                         test reads, with one access per word, its own chunk's tag and which earlier chunks
//                         have the word at all (a tag of this epoch). [A separate bit map of chunks per code
//                         cost a second random access per word in both kernels.]
// For member u of chunk c: the multiplicities of its words whose first[.][c] tag is earlier than u, and,
// per earlier chunk t, those of its words that chunk t has. When every one of these sums stays below u's
// threshold (on both strands), no earlier open member can be its candidate: u is a new
// representative for certain -- appended to the window's list of new representatives and made final.
// Typically that is the first member of every family that appears in the window.
constexpr uint32_t kMaxChunks = 32;
struct Chunks { uint32_t n; uint32_t begin[kMaxChunks + 1]; };   // member ql belongs to chunk c: begin[c] <= ql < begin[c+1]
__device__ __forceinline__ uint32_t chunk_of(const Chunks &C, uint32_t ql) {
    uint32_t c = 0;
    for (uint32_t t = 1; t < C.n; ++t) c += ql >= C.begin[t];
    return c;
}
__global__ __launch_bounds__(256) void first_open_kernel(DevSeqs S, const uint32_t *__restrict__ ulist,
                                                        const uint32_t *__restrict__ n_open, uint32_t b0,
                                                        uint32_t epoch, uint32_t *__restrict__ first, uint32_t stride,
                                                        Chunks C, const int32_t *__restrict__ req_aan,
                                                        uint8_t *__restrict__ decided) {
    const uint32_t lane = threadIdx.x & 63u, n = *n_open;
    for (uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6); w < n; w += gridDim.x * 4) {
        const uint32_t k = ulist[w], ql = k - b0, c = chunk_of(C, ql);
        const uint64_t o = S.off[k];
        const uint32_t nw = S.wcnt[k], tag = (epoch << 16) | (65535u - ql);
        // Tags only ever move to earlier members, so the words that carry an earlier member's tag NOW are a
        // lower bound of what certain_kernel will count for this member's own chunk: at the threshold already,
        // the member has an earlier open candidate for sure and that kernel need not look at it (members come
        // roughly in order, so this settles most members of a family but its first)
        uint32_t sum = 0;
        for (uint32_t i = lane; i < nw; i += 64) {   // (members of one family share most words: mostly the reads)
            uint32_t *slot = first + (size_t)S.wcode[o + i] * stride + c;
            const uint32_t v = *slot;
            if (v < tag) atomicMax(slot, tag);
            else if (v > tag && (v >> 16) == epoch) sum += S.wmult[o + i];
        }
        for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d);
        if (lane == 0) decided[ql] = sum != 0u && (int32_t)sum >= req_aan[k];
    }
}
// One wave per open member.
__global__ __launch_bounds__(256) void certain_kernel(DevSeqs S, const uint32_t *__restrict__ ulist,
                                                     const uint32_t *__restrict__ n_open, uint32_t b0,
                                                     uint32_t both, uint32_t epoch,
                                                     const uint32_t *__restrict__ first, uint32_t stride, Chunks C,
                                                     const int32_t *__restrict__ req_aan,
                                                     const uint8_t *__restrict__ decided,
                                                     uint8_t *__restrict__ done, uint32_t *__restrict__ list,
                                                     uint32_t *__restrict__ n_list) {
    __shared__ uint32_t s_cnt[4][kMaxChunks];
    const uint32_t lane = threadIdx.x & 63u, n = *n_open;
    uint32_t *cnt = s_cnt[threadIdx.x >> 6];
    for (uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6); w < n; w += gridDim.x * 4) {
        const uint32_t k = ulist[w], ql = k - b0;
        if (decided[ql]) continue;               // (first_open_kernel: an earlier open candidate for sure)
        const uint32_t c = chunk_of(C, ql);
        const int32_t t0 = req_aan[k];
        bool cand = false;
        for (uint32_t strand = 0; strand < (both ? 2u : 1u) && !cand; ++strand) {
            const uint32_t ks = strand ? S.n_fwd + k : k;
            const uint64_t o = S.off[ks];
            const uint32_t nw = S.wcnt[ks];
            if (lane < kMaxChunks) cnt[lane] = 0u;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            uint32_t sum = 0;
            for (uint32_t i = lane; i < nw; i += 64) {
                const uint32_t m = S.wmult[o + i];
                const uint32_t *rec = first + (size_t)S.wcode[o + i] * stride;
                for (uint32_t t4 = 0; t4 <= c; t4 += 4) {          // (c is the same for the whole wave)
                    uint32_t v[4];
                    if (stride >= 4) { const uint4 x = *reinterpret_cast<const uint4 *>(rec + t4); v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; }
                    else { v[0] = rec[0]; v[1] = stride > 1 ? rec[1] : 0u; v[2] = 0u; v[3] = 0u; }
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) {
                        const uint32_t t = t4 + j;
                        if (t > c || (v[j] >> 16) != epoch) continue;
                        if (t < c) atomicAdd(&cnt[t], m);                     // an earlier chunk has the word
                        else if (65535u - (v[j] & 65535u) < ql) sum += m;    // an earlier member of the own chunk
                    }
                }
            }
            for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            uint32_t mx = lane < c ? cnt[lane] : 0u;
            mx = mx > sum ? mx : sum;
            for (int d = 32; d > 0; d >>= 1) { const uint32_t v = __shfl_xor(mx, d); mx = v > mx ? v : mx; }
            cand = mx != 0u && (int32_t)mx >= t0;
            __builtin_amdgcn_wave_barrier();
        }
        if (!cand && lane == 0) {
            done[ql] = 1;
            list[atomicAdd(n_list, 1u)] = k;
        }
    }
}

// per-sequence thresholds from the per-length tables tab[0..n_len) = aa1, [n_len..2 n_len) = aas, [2 n_len..) = aan
__global__ __launch_bounds__(256) void thresholds_kernel(const uint32_t *__restrict__ len, uint32_t n,
                                                        const int32_t *__restrict__ tab, uint32_t n_len,
                                                        int32_t *__restrict__ aa1, int32_t *__restrict__ aas,
                                                        int32_t *__restrict__ aan) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t L = len[k];
    aa1[k] = tab[L]; aas[k] = tab[n_len + L]; aan[k] = tab[2 * n_len + L];
}

// (hipMemsetAsync between kernels left the stream idle for ~130 us each time -- 55 ms per run in the first timeline of this
// design -- so the per-round clears happen inside kernels of our own: index_place_kernel, list_open_kernel)
// small bookkeeping kernels -----------------------------------------------------------------

// block members are final once the host has walked the block (final = 0: the block is given up, see the host)
// (begin != nullptr: the pass over the block's representatives begins -- no code touched yet, its pairs begin here)
__global__ void retire_block_kernel(uint8_t *__restrict__ done, uint8_t *__restrict__ inblk, uint32_t nb, uint32_t final_,
                                    uint32_t *__restrict__ begin) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q == 0 && begin) { begin[C_TOUCH] = 0u; begin[C_NK] = 0u; begin[C_EVAL0] = begin[C_NW]; }
    if (q < nb && inblk[q]) { done[q] = (uint8_t)final_; inblk[q] = 0; }
}
// Start of a window: counters, best keys, reverse-strand visit counters and member flags in one launch.
__global__ __launch_bounds__(256) void window_init_kernel(uint32_t *__restrict__ counters,
                                                         unsigned long long *__restrict__ best,
                                                         unsigned long long *__restrict__ rc_visits,
                                                         uint8_t *__restrict__ flags, uint32_t window_cap) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;  // grid covers window_cap
    if (q < C_COUNT && q != C_ERR) counters[q] = 0u;
    if (q >= window_cap) return;
    best[q] = kNoBest;
    if (rc_visits) rc_visits[q] = 0ull;
    if (q < 4 * window_cap / 8) reinterpret_cast<unsigned long long *>(flags)[q] = 0ull;
}

// Memory-chunked rule (pgx.h chunk_boundaries): window members that an earlier chunk's sweep has already placed are
// ABSENT from everything that follows -- done = 2 (the filter skips them; they are never open), a best key that is
// not "none". `taken` is the host's per-sequence flag array (page-locked, mapped), offset to the window.
__global__ __launch_bounds__(256) void mark_absent_kernel(const uint8_t *__restrict__ taken, uint32_t nb,
                                                         uint8_t *__restrict__ done,
                                                         unsigned long long *__restrict__ best) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nb && taken[q]) { done[q] = 2; best[q] = 0ull; }
}

// record-sharded mode: element-wise minimum of the processes' best keys (gathered, one row per process)
// (the error words of the processes, one behind each row's keys, are OR-ed into this window's C_ERR)
__global__ __launch_bounds__(256) void min_rows_kernel(const unsigned long long *__restrict__ rows, uint32_t n_rows,
                                                      uint32_t row_stride, uint32_t n,
                                                      unsigned long long *__restrict__ out, uint32_t *__restrict__ err) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        uint32_t e = 0;
        for (uint32_t r = 0; r < n_rows; ++r) e |= (uint32_t)rows[(size_t)r * row_stride + PGX_EXCHANGE_KEYS];
        if (e) atomicOr(err, e);
    }
    if (i >= n) return;
    unsigned long long m = rows[i];
    for (uint32_t r = 1; r < n_rows; ++r) { const unsigned long long v = rows[(size_t)r * row_stride + i]; m = v < m ? v : m; }
    out[i] = m;
}

// before an exchange: this process's error word goes behind its keys
__global__ void exchange_prepare_kernel(const uint32_t *__restrict__ counters, uint32_t pair_cap,
                                        unsigned long long *__restrict__ send, uint32_t inject) {
    send[PGX_EXCHANGE_KEYS] = counters[C_ERR] | (counters[C_NW] > pair_cap ? (uint32_t)E_PAIRS : 0u) | inject;
}

// One round trip's worth of results written straight into page-locked host memory: each
// segment copies min(*count or `fixed`, cap) records of `words` dwords. Replaces a string of
// small device-to-host copies (each one a blit launch of its own) by a single launch, and
// copies exactly the records that exist.
constexpr uint32_t kPairWords = sizeof(Pair) / 4;
static_assert(sizeof(Pair) % 4 == 0, "pair records are published as dwords");
struct PubSeg { const uint32_t *src; uint32_t *dst; const uint32_t *count; uint32_t fixed, words, cap; };
constexpr int kPubSegs = 8;
struct PubArgs { PubSeg seg[kPubSegs]; int n; };
__global__ __launch_bounds__(256) void publish_kernel(PubArgs a) {
    for (int s = 0; s < a.n; ++s) {
        const PubSeg g = a.seg[s];
        uint32_t c = g.count ? *g.count : g.fixed;
        if (c > g.cap) c = g.cap;
        const uint32_t total = c * g.words;
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) g.dst[i] = g.src[i];
    }
}

__global__ __launch_bounds__(256) void gather_pairs_kernel(const Pair *__restrict__ pairs,
                                                          const uint32_t *__restrict__ list, uint32_t n,
                                                          Pair *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = pairs[list[i]];
}


// ----------------------------------------------------------------------------------------
// diag: 2-mer diagonal histogram + best band (one wave per pair)
// ----------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t lane, uint32_t *total) {
    uint32_t x = v;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d);
        if ((int)lane >= d) x += y;
    }
    *total = __shfl(x, 63);
    return x - v;
}

// Band selection of the diagonal test, exactly as the sequential rule states it: the FIRST best
// window of `band_width` diagonals by complexity-weighted hits; centre `imax` = best single
// diagonal among those of the first window and those that entered a window at the moment it
// became the new best (first occurrence of the largest); edges trimmed. `d` packs hits |
// weighted << 16 per diagonal. The rule is a running scan over up to 2 L diagonals; here the
// whole wave evaluates it: every lane slides the window over its own stretch of start positions
// twice (once for the stretch's maximum, once -- knowing the maximum before it -- to find the
// moments the sequential scan would have improved), with wave reductions in between. All 64
// lanes must call; the results are valid on every lane.
template <typename DiagPtr>
__device__ void band_from_histogram(DiagPtr d, int len1, int len2, int band_width, int required_aa1,
                                    double cluster_thd, int *best_sum, int *bl, int *bc, int *br) {
    const int lane = threadIdx.x & 63;
    const int nall = len1 + len2 - 1;
    const int band_b = required_aa1 - 1 >= 0 ? required_aa1 - 1 : 0;
    const int band_e = nall - band_b;
    const int band_m = band_b + band_width - 1 < band_e ? band_b + band_width - 1 : band_e;
    const int w = band_m - band_b + 1;                         // diagonals in a window (<= 64; <= 0: none)
    const int T = band_e - band_m - 1 > 0 && w > 0 ? band_e - band_m - 1 : 0;   // windows after the first: starts band_b + 1 .. band_b + T
    using Cell = std::remove_cv_t<std::remove_reference_t<decltype(d[0])>>;   // 16 + 16 bits, or 32 + 32 for giant queries
    constexpr int kHalf = (int)sizeof(Cell) * 4;
    auto hits = [&](int i) { return (int)(d[i] & ((Cell(1) << kHalf) - 1)); };
    auto weighted = [&](int i) { return (int)(d[i] >> kHalf); };
    auto wave_max = [&](int v) { for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o)); return v; };
    auto wave_min = [&](int v) { for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o)); return v; };
    auto wave_sum = [&](int v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; };
    // ---- first window: sums, and the first largest weighted diagonal in it (0 if all are zero) ----
    const int x0 = lane < w ? weighted(band_b + lane) : 0;
    const int score2_first = wave_sum(x0);
    int max_diag2 = wave_max(x0);
    int imax = max_diag2 > 0 ? wave_min(x0 == max_diag2 && lane < w ? band_b + lane : INT32_MAX) : 0;
    // ---- later windows, start position band_b + t for t = 1..T: lane's stretch [t_lo, t_hi) ----
    const int per = (T + 63) / 64;
    const int t_lo = 1 + lane * per, t_hi = min(T + 1, t_lo + per);
    auto window_at = [&](int t) {   // weighted sum of the window that starts at band_b + t
        int sum = 0;
        for (int i = 0; i < w; ++i) sum += weighted(band_b + t + i);
        return sum;
    };
    int first = 0, local_max = INT32_MIN;
    if (t_lo < t_hi) {
        first = window_at(t_lo);
        int sc = first;
        local_max = sc;
        for (int t = t_lo + 1; t < t_hi; ++t) {
            sc += weighted(band_m + t) - weighted(band_b + t - 1);
            local_max = max(local_max, sc);
        }
    }
    // largest window score before this lane's stretch (the first window included)
    int before = local_max;
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(before, o);
        if (lane >= o) before = max(before, up);
    }
    before = __shfl_up(before, 1);
    before = max(lane == 0 ? INT32_MIN : before, score2_first);
    // second slide: the moments the running scan improves (strictly), its last one, and the best
    // diagonal entering at such a moment (first occurrence of the largest)
    int last_t = 0, cand_val = 0, cand_t = INT32_MAX;
    if (t_lo < t_hi) {
        int sc = first, run = before;
        for (int t = t_lo; t < t_hi; ++t) {
            if (t > t_lo) sc += weighted(band_m + t) - weighted(band_b + t - 1);
            if (sc > run) {
                run = sc; last_t = t;
                const int v = weighted(band_m + t);
                if (v > cand_val) { cand_val = v; cand_t = t; }
            }
        }
    }
    const int best_t = wave_max(last_t);                         // 0 = the first window stayed the best
    const int cand_best = wave_max(cand_val);
    if (cand_best > max_diag2) {                                 // (a tie keeps the earlier diagonal)
        max_diag2 = cand_best;
        imax = band_m + wave_min(cand_val == cand_best ? cand_t : INT32_MAX);
    }
    int from = band_b + best_t, end = band_m + best_t;
    int best_score = wave_sum(lane < w ? hits(from + lane) : 0);
    // ---- trimming (at most a window's width of steps) ----
    int mlen = imax;
    if (imax > len1) mlen = nall - imax;
    const int emax = (int)((1.0 - cluster_thd) * mlen) + 1;
    // (a window without a single hit leaves imax = 0 and trims below band_b: those diagonals
    // are not stored; such a pair fails the test whatever they hold because required_aas >= 1)
    for (int j = from; j < imax; ++j) {
        const int s1 = hits(j);
        if ((imax - j) > emax || s1 < 1) { best_score -= s1; from++; } else break;
    }
    for (int j = end; j > imax; --j) {
        const int s1 = j >= band_b ? hits(j) : 0;
        if ((j - imax) > emax || s1 < 1) { best_score -= s1; end--; } else break;
    }
    *bl = from - len1 + 1; *br = end - len1 + 1; *bc = imax - len1 + 1; *best_sum = best_score;
}

// code of the diagonal-test k-mer at s[j] (2-mer base 21 or 4-mer base 4), -1 if a nucleotide
// k-mer contains N; *cpx = 1 + number of adjacent unequal residues inside it
__device__ __forceinline__ int kd_code(const DevSeqs &S, const uint8_t *__restrict__ s, int j, int *cpx) {
    int code = 0, c = 1;
    bool bad = false;
    for (int t = 0; t < S.kd; ++t) {
        const int r = s[j + t];
        bad |= r >= S.base;
        code = code * S.base + r;
        if (t) c += r != s[j + t - 1];
    }
    *cpx = c;
    return (S.nt && bad) ? -1 : code;
}

// rep_seq == nullptr: p.r is already a sequence index (phase B)
// CAP: diagonals / query 2-mers kept in LDS (larger pairs use the global scratch): 512 gives six waves per SIMD
// where 2048 gives three -- a pair of 340-residue sequences at 0.8 identity needs 140 diagonals and 340 positions
// Cell: one diagonal's hits | complexity-weighted hits: 16 + 16 bits, or -- queries beyond kPackedLen*, always in the
// global scratch -- 32 + 32.
template <uint32_t CAP, typename Cell>
__global__ __launch_bounds__(64) void diag_kernel(DevSeqs S, const uint32_t *__restrict__ rep_seq,
                                                 Pair *__restrict__ pairs, PairSel sel,
                                                 const int32_t *__restrict__ req_aa1,
                                                 const int32_t *__restrict__ req_aas, int band_width,
                                                 double cluster_thd, uint32_t *__restrict__ gscratch,
                                                 uint32_t gscratch_stride, uint32_t gscratch_cells,
                                                 uint32_t *__restrict__ n_wide, uint32_t *__restrict__ err) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_wide = 0u;   // (pairs that the 16-lane aligner, which runs next, leaves to align_kernel)
    constexpr bool kWideCell = sizeof(Cell) == 8;
    constexpr int kHalf = (int)sizeof(Cell) * 4;
    __shared__ Cell diag[CAP];
    __shared__ uint32_t taap[kNAA1 * kNAA1 + 7];
    __shared__ uint32_t abeg[kNAA1 * kNAA1 + 7];
    __shared__ uint16_t alist[CAP];
    // the pair's residues, staged with 16-byte loads (one instruction per sequence up to 1 KB): read byte by byte from
    // global memory, every one of the passes below waited for its own loads -- four to five memory latencies per pair,
    // which is what a pair cost
    constexpr uint32_t kRepCap = CAP * 3 / 2;
    __shared__ uint4 stage1[kWideCell ? 1 : CAP / 16 + 2];
    __shared__ uint4 stage2[kWideCell ? 1 : kRepCap / 16 + 2];
    const uint32_t lane = threadIdx.x;
    const uint32_t np = sel_count(sel);
    const int N2 = S.nt ? 256 : kNAA1 * kNAA1;
    for (uint32_t w = blockIdx.x; w < np; w += gridDim.x) {
        const uint32_t p = sel_pair(sel, w);
        const Pair pr = pairs[p];
        if (np > sel.all_if_le) {   // (a few hundred pairs cost one dependent alignment chain whether all or some are taken)
            if (sel.only_a && !sel.only_a[pr.r - sel.b0]) continue;  // block-uniform
            if (sel.only_not_b && sel.only_not_b[pr.r - sel.b0]) continue;
        }
        if (sel.skip_evaluated && (pr.flags & F_EVAL)) continue;
        const uint32_t k1 = pr.q, k2 = rep_seq ? rep_seq[pr.r] : pr.r;
        const uint32_t k1r = real_of(S, k1);  // thresholds are the query's, whichever strand
        const int len1 = (int)S.len[k1], len2 = (int)S.len[k2];
        const uint8_t *s1 = S.res + S.off[k1];
        const uint8_t *s2 = S.res + S.off[k2];
        const int nall = len1 + len2 - 1;
        // only diagonals with an overlap of at least required_aa1 residues are ever read back
        // (band_b .. band_e of the sequential rule), so only that window is histogrammed
        const int d_lo = req_aa1[k1r] - 1 >= 0 ? req_aa1[k1r] - 1 : 0;
        const int d_hi = nall - d_lo;
        const int n_d = d_hi >= d_lo ? d_hi - d_lo + 1 : 0;
        const bool big = kWideCell || (uint32_t)n_d > CAP || (uint32_t)len1 > CAP;
        Cell *dg = big ? reinterpret_cast<Cell *>(gscratch + (size_t)blockIdx.x * gscratch_stride) : diag;
        // the query's 2-mer position lists (global scratch tail for oversized queries: behind 2 x cells diagonals)
        uint32_t *al_big = big ? reinterpret_cast<uint32_t *>(dg + 2 * (size_t)gscratch_cells) : nullptr;
        const int last1 = len1 - S.kd, last2 = len2 - S.kd;
        const bool in_lds = !kWideCell && !big && (uint32_t)len2 <= kRepCap;
        if (in_lds) {
            // ---- the common case, all in LDS: the query's k-mer positions as chains (head per code, next per position:
            // one pass, no counting), the representative's k-mers walk them ----
            const uint8_t *g1 = reinterpret_cast<const uint8_t *>(reinterpret_cast<uintptr_t>(s1) & ~uintptr_t(15));
            const uint8_t *g2 = reinterpret_cast<const uint8_t *>(reinterpret_cast<uintptr_t>(s2) & ~uintptr_t(15));
            const uint32_t sh1 = (uint32_t)(s1 - g1), sh2 = (uint32_t)(s2 - g2);
            const uint32_t nv1 = (sh1 + (uint32_t)len1 + 15u) / 16u, nv2 = (sh2 + (uint32_t)len2 + 15u) / 16u;
            uint4 v1[(CAP / 16 + 2 + 63) / 64], v2[(kRepCap / 16 + 2 + 63) / 64];
#pragma unroll
            for (uint32_t t = 0; t < sizeof(v1) / sizeof(uint4); ++t) if (lane + 64 * t < nv1) v1[t] = reinterpret_cast<const uint4 *>(g1)[lane + 64 * t];
#pragma unroll
            for (uint32_t t = 0; t < sizeof(v2) / sizeof(uint4); ++t) if (lane + 64 * t < nv2) v2[t] = reinterpret_cast<const uint4 *>(g2)[lane + 64 * t];
            constexpr uint32_t kEnd = 0xFFFFu;
            uint32_t *head = taap;
            uint16_t *next = alist;
            for (int i = lane; i < n_d; i += 64) diag[i] = Cell(0);
            for (int c = lane; c < N2; c += 64) head[c] = kEnd;
#pragma unroll
            for (uint32_t t = 0; t < sizeof(v1) / sizeof(uint4); ++t) if (lane + 64 * t < nv1) stage1[lane + 64 * t] = v1[t];
#pragma unroll
            for (uint32_t t = 0; t < sizeof(v2) / sizeof(uint4); ++t) if (lane + 64 * t < nv2) stage2[lane + 64 * t] = v2[t];
            __syncthreads();
            const uint8_t *l1 = reinterpret_cast<const uint8_t *>(stage1) + sh1;
            const uint8_t *l2 = reinterpret_cast<const uint8_t *>(stage2) + sh2;
            for (int j = lane; j <= last1; j += 64) {
                int cpx;
                const int c = kd_code(S, l1, j, &cpx);
                if (c >= 0) next[j] = (uint16_t)atomicExch(&head[c], (uint32_t)j);
            }
            __syncthreads();
            for (int i = lane; i <= last2; i += 64) {
                int cpx;
                const int c = kd_code(S, l2, i, &cpx);
                if (c < 0) continue;
                const Cell inc = Cell(1) | (Cell((uint32_t)cpx) << kHalf);
                for (uint32_t j = head[c]; j != kEnd; j = next[j]) {
                    const int d = len1 - 1 + i - (int)j;
                    if (d >= d_lo && d <= d_hi) atomicAdd(&diag[d - d_lo], inc);
                }
            }
            __syncthreads();
        } else {
        for (int i = lane; i < n_d; i += 64) dg[i] = Cell(0);
        for (int c = lane; c < N2; c += 64) taap[c] = 0u;
        __syncthreads();
        for (int j = lane; j <= last1; j += 64) {
            int cpx;
            const int c = kd_code(S, s1, j, &cpx);
            if (c >= 0) atomicAdd(&taap[c], 1u);
        }
        __syncthreads();
        {   // exclusive scan of the 441 bucket sizes, 7 per lane
            uint32_t loc[7], sum = 0;
            for (int t = 0; t < 7; ++t) { const int c = lane * 7 + t; loc[t] = c < N2 ? taap[c] : 0u; sum += loc[t]; }
            uint32_t total, base = wave_excl_scan(sum, lane, &total);
            for (int t = 0; t < 7; ++t) { const int c = lane * 7 + t; if (c < N2) { abeg[c] = base; taap[c] = 0u; } base += loc[t]; }
        }
        __syncthreads();
        for (int j = lane; j <= last1; j += 64) {
            int cpx;
            const int c = kd_code(S, s1, j, &cpx);
            if (c < 0) continue;
            const uint32_t pos = abeg[c] + atomicAdd(&taap[c], 1u);
            if (big) al_big[pos] = (uint32_t)j; else alist[pos] = (uint16_t)j;
        }
        __syncthreads();
        for (int i = lane; i <= last2; i += 64) {
            int cpx;
            const int c = kd_code(S, s2, i, &cpx);
            if (c < 0) continue;
            const Cell inc = Cell(1) | (Cell((uint32_t)cpx) << kHalf);
            const uint32_t b = abeg[c], e = b + taap[c];
            for (uint32_t t = b; t < e; ++t) {
                const int j = big ? (int)al_big[t] : (int)alist[t];
                const int d = len1 - 1 + i - j;
                if (d >= d_lo && d <= d_hi) atomicAdd(&dg[d - d_lo], inc);
            }
        }
        __syncthreads();
        }
        int best_sum, bl, bc, br;
        {
            const int bw = band_width < len1 + len2 - 2 ? band_width : len1 + len2 - 2;
            if (in_lds) band_from_histogram(diag - d_lo, len1, len2, bw, req_aa1[k1r], cluster_thd, &best_sum, &bl, &bc, &br);
            else band_from_histogram(dg - d_lo, len1, len2, bw, req_aa1[k1r], cluster_thd, &best_sum, &bl, &bc, &br);
        }
        if (lane == 0) {
            uint32_t fl = F_EVAL;
            if (best_sum >= req_aas[k1r]) fl |= F_DIAG_PASS;
            if (!(br >= len2 || bl <= -len1 || bl > br)) fl |= F_BAND_OK;
            if (br - bl + 1 > kMaxBand) { fl |= F_TOO_BIG; if (fl & F_DIAG_PASS) atomicOr(err, (uint32_t)E_BAND); }
            pairs[p].best_sum = best_sum; pairs[p].band_left = bl; pairs[p].band_center = bc;
            pairs[p].band_right = br; pairs[p].flags = fl;
        }
        __syncthreads();
    }
}

// ----------------------------------------------------------------------------------------
// align: banded DP, one band column per lane, anti-diagonal wavefront (one wave per pair)
// ----------------------------------------------------------------------------------------
enum { BK_NONE = 0, BK_DIAG = 1, BK_LEFT = 2, BK_TOP = 3 };

__device__ __forceinline__ int64_t shfl_i64(int64_t v, int src_lane) {
    const int lo = __shfl((int)(uint32_t)v, src_lane);
    const int hi = __shfl((int)(v >> 32), src_lane);
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

// Banded alignment of query s1 (len1) against representative s2 (len2) over diagonals
// [bl, br] (bc = centre). Cell (i, j1), j = j1 + i + bl, is computed by lane j1 at step
// t = 2 i + j1; its neighbours (i, j1-1) and (i-1, j1+1) were computed one step earlier by
// lanes j1-1 and j1+1, (i-1, j1) two steps earlier by the lane itself. Border cells carry
// score = ext * distance and never count as gap continuations. Returns the number of
// identical pairs on the best path into the end cell (wave-uniform).
__device__ int band_align_wave(const uint8_t *__restrict__ s1, const uint8_t *__restrict__ s2, int len1,
                               int len2, int bl, int bc, int br, const int8_t *__restrict__ sub /*LDS 21x21*/,
                               int gap_open, int gap_ext) {
    const int lane = threadIdx.x & 63;
    const int bw = br - bl + 1;
    const int j1 = lane;
    const bool col = j1 < bw;
    const int i_start = max(1, 1 - bl - j1);
    const int i_end = min(len1, len2 - bl - j1);
    const int i_start_left = max(1, 2 - bl - j1);
    const int maxd = bc - bl;
    const int dist = j1 > maxd ? j1 - maxd : maxd - j1;
    const int64_t bonus = 4 - (dist & 3);
    const int64_t gap = kScoreScale * gap_open, ext = kScoreScale * gap_ext;
    int64_t sc = 0;
    int bk = BK_NONE, id = 0;
    const int t_last = 2 * len1 + bw - 1;
    for (int t = 2; t <= t_last; ++t) {
        const int64_t l_sc_n = shfl_i64(sc, (lane + 63) & 63);
        const int l_meta = __shfl((id << 2) | bk, (lane + 63) & 63);
        const int64_t r_sc_n = shfl_i64(sc, (lane + 1) & 63);
        const int r_meta = __shfl((id << 2) | bk, (lane + 1) & 63);
        const int d = t - j1;
        const int i = d >> 1;
        if (col && !(d & 1) && i >= i_start && i <= i_end) {
            const int j = j1 + i + bl;
            const int ci = s1[i - 1], cj = s2[j - 1];
            int64_t sij = kScoreScale * (int64_t)sub[ci * kNAA1 + cj];
            if (sij > 0) sij += bonus;
            int64_t p_sc; int p_id;
            if (i - 1 >= i_start) { p_sc = sc; p_id = id; }
            else { p_sc = ext * (int64_t)((i - 1 == 0) ? (j - 1) : (i - 1)); p_id = 0; }
            int64_t best = p_sc + sij;
            int back = BK_DIAG, nid = p_id + (ci == cj);
            const int64_t gap0 = (i == len1 || j == len2) ? ext : gap;
            if (j1 > 0) {
                int64_t l_sc; int l_bk, l_id;
                if (i >= i_start_left) { l_sc = l_sc_n; l_bk = l_meta & 3; l_id = l_meta >> 2; }
                else { l_sc = ext * (int64_t)i; l_bk = BK_NONE; l_id = 0; }
                const int64_t s = l_sc + (l_bk == BK_LEFT ? ext : gap0);
                if (s > best) { best = s; back = BK_LEFT; nid = l_id; }
            }
            if (j1 + 1 < bw) {
                int64_t r_sc; int r_bk, r_id;
                if (i - 1 >= 1) { r_sc = r_sc_n; r_bk = r_meta & 3; r_id = r_meta >> 2; }
                else { r_sc = ext * (int64_t)j; r_bk = BK_NONE; r_id = 0; }
                const int64_t s = r_sc + (r_bk == BK_TOP ? ext : gap0);
                if (s > best) { best = s; back = BK_TOP; nid = r_id; }
            }
            sc = best; bk = back; id = nid;
        }
    }
    int end_lane;
    if (len2 - bl < len1) end_lane = 0;
    else if (len1 + br < len2) end_lane = bw - 1;
    else end_lane = len2 - len1 - bl;
    return __shfl(id, end_lane);
}

// ----------------------------------------------------------------------------------------
// align16: the same recurrence for the common case (band <= 32 diagonals, len1 + len2 <=
// slot - 96), four pairs per wave.
//
//   * one pair per DPP row of 16 lanes; lane g owns band columns 2g and 2g+1, so every lane
//     computes one cell on every step of the anti-diagonal wavefront (no idle parity) and
//     needs ONE neighbour exchange per step: row_shr:1 before the even column (left cell,
//     from lane g-1's odd column), row_shl:1 before the odd column (top cell, from lane
//     g+1's even column); everything else is the lane's own previous cells.
//   * scores are int32: S * 2^14 + E with S the BLOSUM/gap sum and E the centre-diagonal
//     bonus total (E <= 4 * min(len) < 2^14), which orders and ties exactly like the
//     reference's 655360 * S + E.
//   * both sequences are staged in LDS (aligned dword copies); substitution score + bonus
//     come from four LDS tables (bonus 1..4), looked up one row ahead of use.
//   * border cells (row 0 / column 0 of the DP matrix) are produced by the same lanes as
//     forced values, so interior cells never special-case their neighbours.
// ----------------------------------------------------------------------------------------
// LDS bytes per pair: a template parameter (1024 / 1536 / 2048 / 3072 / 4096), chosen per window from its longest query. The
// kernel is bound by each pair's dependent chain, so what counts is waves per SIMD: 30 KB of LDS per workgroup
// with 1 KB slots = five, 78 KB with 4 KB slots = two (17.8 -> 8.8 ms per step for the windows of cfg-3s that fit).
// len1 + len2 <= slot - 96 is handled here (both sequences, 6-padded, fit the slot); the rest, and bands over 32
// diagonals, are counted in *n_wide and left to align_kernel.
constexpr int kA16MaxSlot = 4096;
constexpr int kA16BigSlot = 8192;   // second pass over what the 4 KB slots left (one workgroup per CU: only for those pairs)
constexpr int kScaleShift = 14;

__device__ __forceinline__ int dpp_row_shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true); }
__device__ __forceinline__ int dpp_row_shl1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x101, 0xF, 0xF, true); }

// lane-wise select under a lane mask, pinned to one v_cndmask (the compiler otherwise turns
// chains of selects in the alignment's inner loop into branches)
__device__ __forceinline__ uint32_t lane_select(uint64_t mask, uint32_t if_set, uint32_t if_clear) {
    uint32_t d;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(if_clear), "v"(if_set), "s"(mask));
    return d;
}
__device__ __forceinline__ uint64_t lanes_equal(uint32_t a, uint32_t b) { return __builtin_amdgcn_uicmp(a, b, 32 /* ICMP_EQ */); }

__device__ __forceinline__ bool pair_is_wide(int len1, int len2, int bl, int br, int max_sum) {
    return len1 + len2 > max_sum || br - bl + 1 > 32;
}

template <int kA16Slot>
__global__ __launch_bounds__(256) void align16_kernel(DevSeqs S, const uint32_t *__restrict__ rep_seq,
                                                     Pair *__restrict__ pairs, PairSel sel,
                                                     const int32_t *__restrict__ req_aa1, double cluster_thd,
                                                     uint32_t b0, unsigned long long *__restrict__ best,
                                                     uint32_t key_flag, uint32_t *__restrict__ n_wide, uint32_t second_pass) {
    // substitution score + centre bonus in key form (value * 4 + 2, see the interior rows) and the
    // identity bit, one 8-byte entry per residue pair and bonus class
    __shared__ int2 tab[4][kNAA1 * kNAA1];
    __shared__ uint32_t seqbuf[16 * (kA16Slot / 4) + 32];  // + padding: lanes outside a band may read a few bytes past the last slot
    // one wave per SIMD, bound by its own dependent chain: issue ahead of the side stream's table pass
    __builtin_amdgcn_s_setprio(3);
    const uint32_t n = sel_count(sel);
    if (blockIdx.x * 16u >= n) return;   // (a round with few pairs: most workgroups have none; 50 us of table set-up otherwise)
    // second_pass: the 8 KB-slot instantiation, launched behind the window's own for the pairs that one left (windows of
    // sequences beyond 2,000 residues); it counts nothing and leaves at once when nothing was left
    if (second_pass && *n_wide == 0u) return;
    for (int c = threadIdx.x; c < 4 * kNAA1 * kNAA1; c += 256) {
        const int cc = c % (kNAA1 * kNAA1);
        const int s = S.nt ? (cc / kNAA1 == cc % kNAA1 ? 2 : -2) : (int)kBlosum62_dev[cc];
        const int v = s * (1 << kScaleShift) + (s > 0 ? c / (kNAA1 * kNAA1) + 1 : 0);
        tab[c / (kNAA1 * kNAA1)][cc] = make_int2(v * 4 + 2, cc / kNAA1 == cc % kNAA1);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, gl = lane & 15;
    const int slot = (threadIdx.x >> 6) * 4 + (lane >> 4);
    const uint8_t *sb = reinterpret_cast<const uint8_t *>(seqbuf + slot * (kA16Slot / 4));
    const int gap = (S.nt ? -6 : kGapOpen) * (1 << kScaleShift), ext = (S.nt ? -1 : kGapExt) * (1 << kScaleShift);
    constexpr int kNever = INT32_MIN / 2;

    for (uint32_t base = blockIdx.x * 16u; base < n; base += gridDim.x * 16u) {
        const uint32_t w = base + (uint32_t)slot;
        uint32_t p = 0;
        Pair pr{};
        bool fast = false;
        int len1 = 1, len2 = 1;
        uint32_t k1 = 0, k2 = 0;
        if (w < n) {
            p = sel_pair(sel, w);
            pr = pairs[p];
            k1 = pr.q; k2 = rep_seq ? rep_seq[pr.r] : pr.r;
            len1 = (int)S.len[k1]; len2 = (int)S.len[k2];
            fast = (pr.flags & (F_DIAG_PASS | F_BAND_OK | F_TOO_BIG | F_ALIGNED)) == (F_DIAG_PASS | F_BAND_OK);
            if (fast && pair_is_wide(len1, len2, pr.band_left, pr.band_right, kA16Slot - 96)) {
                fast = false;
                if (gl == 0 && !second_pass) atomicAdd(n_wide, 1u);      // left to the second pass / align_kernel
            }
        }
        const int bl = pr.band_left, bw = fast ? pr.band_right - pr.band_left + 1 : 0;
        // ---- stage both sequences: coalesced loads of the 5-bit packed words, unpacked into LDS bytes ----
        int o1 = 0, o2 = 0;
        if (fast) {
            const uint32_t *p1 = S.pk + S.pk_off[k1], *p2 = S.pk + S.pk_off[k2];
            const int w1 = (len1 + 5) / 6, w2 = (len2 + 5) / 6;
            uint8_t *sbw = reinterpret_cast<uint8_t *>(seqbuf + slot * (kA16Slot / 4));
            o2 = (6 * w1 + 7) & ~7;
            for (int w = gl; w < w1; w += 16) {
                const uint32_t x = p1[w];
                for (int t = 0; t < 6; ++t) sbw[6 * w + t] = (uint8_t)((x >> (5 * t)) & 31u);
            }
            for (int w = gl; w < w2; w += 16) {
                const uint32_t x = p2[w];
                for (int t = 0; t < 6; ++t) sbw[o2 + 6 * w + t] = (uint8_t)(((x >> (5 * t)) & 31u) << 3);  // * 8: a table offset
            }
        }
        // ---- per-lane geometry of its two band columns -------------------------------------
        const int c0 = 2 * gl, c1 = c0 + 1;
        const bool v0 = c0 < bw, v1 = c1 < bw;
        const int if0 = max(0, -bl - c0), if1 = max(0, -bl - c1);          // border row of the column
        const int ie0 = min(len1, len2 - bl - c0), ie1 = min(len1, len2 - bl - c1);
        const int bs0 = ext * (if0 > 0 ? if0 : c0 + bl), bs1 = ext * (if1 > 0 ? if1 : c1 + bl);
        const int maxd = pr.band_center - bl;
        const int d0 = c0 > maxd ? c0 - maxd : maxd - c0, d1 = c1 > maxd ? c1 - maxd : maxd - c1;
        const int2 *t0 = tab[3 - (d0 & 3)], *t1 = tab[3 - (d1 & 3)];   // bonus = 4 - (dist & 3)
        const bool left0 = c0 > 0, top0 = c0 + 1 < bw, top1 = c1 + 1 < bw;
        const int trip = fast ? len1 + ((bw + 1) >> 1) : 0;                // rows 0..len1 for lanes 0..nl-1

        // residues / table entries are fetched one row ahead
        auto res1 = [&](int i) { return (int)sb[o1 + max(i - 1, 0)]; };       // query residue of row i
        auto res2 = [&](int j) { return (int)sb[o2 + min(max(j - 1, 0), len2 - 1)]; };  // representative residue of column j (* 8)
        auto entry = [](const int2 *t, int c_query, int c_rep8) {  // table entry of a residue pair
            return *reinterpret_cast<const int2 *>(__builtin_assume_aligned(
                reinterpret_cast<const char *>(t) + c_query * (kNAA1 * 8) + c_rep8, 8));
        };
        int i = -gl;                                 // row of this lane in iteration tau: i = tau - gl
        int ci = res1(i), cje = res2(c0 + i + bl), cjo = res2(c1 + i + bl);
        int2 te = entry(t0, ci, cje), to = entry(t1, ci, cjo);   // {key-form score, match} of the row's two cells
        int sc0 = 0, m0 = 0, sc1 = 0, m1 = 0;       // last cell of the even / odd column: score, iden << 2 | back

        // One BORDER-AWARE row of the wavefront for this lane's two columns: first / last rows of
        // a column, forced border cells, end gaps. Used before and after the interior rows.
        auto row = [&](int tau) {
            const int ci_n = res1(i + 1), cjo_n = res2(c1 + i + 1 + bl);   // operands of the next row
            const int cje_n = cjo;                                         // j_even(tau + 1) == j_odd(tau)
            const int se = te.x >> 2, so = to.x >> 2, me = te.y, mo = to.y;
            const int l_sc = dpp_row_shr1(sc1), l_m = dpp_row_shr1(m1);
            {   // ---- even column: cell (i, c0) ----
                const int g0 = i == ie0 ? ext : gap;
                int bst = sc0 + se, bm = ((m0 >> 2) + me) << 2 | BK_DIAG;
                const int ls = left0 ? l_sc + ((l_m & 3) == BK_LEFT ? ext : g0) : kNever;
                if (ls > bst) { bst = ls; bm = (l_m & ~3) | BK_LEFT; }
                const int ts = top0 ? sc1 + ((m1 & 3) == BK_TOP ? ext : g0) : kNever;
                if (ts > bst) { bst = ts; bm = (m1 & ~3) | BK_TOP; }
                if (i == if0) { bst = bs0; bm = BK_NONE; }
                if (v0 && tau < trip && i >= if0 && i <= ie0) { sc0 = bst; m0 = bm; }
            }
            const int r_sc = dpp_row_shl1(sc0), r_m = dpp_row_shl1(m0);
            {   // ---- odd column: cell (i, c1) ----
                const int g0 = i == ie1 ? ext : gap;
                int bst = sc1 + so, bm = ((m1 >> 2) + mo) << 2 | BK_DIAG;
                const int ls = sc0 + ((m0 & 3) == BK_LEFT ? ext : g0);
                if (ls > bst) { bst = ls; bm = (m0 & ~3) | BK_LEFT; }
                const int ts = top1 ? r_sc + ((r_m & 3) == BK_TOP ? ext : g0) : kNever;
                if (ts > bst) { bst = ts; bm = (r_m & ~3) | BK_TOP; }
                if (i == if1) { bst = bs1; bm = BK_NONE; }
                if (v1 && tau < trip && i >= if1 && i <= ie1) { sc1 = bst; m1 = bm; }
            }
            // keep the next row's table look-ups (which wait for the residues read above) behind
            // the DP arithmetic, so the LDS latency overlaps it
            __builtin_amdgcn_sched_barrier(0);
            cje = cje_n; cjo = cjo_n; ci = ci_n;
            te = entry(t0, ci, cje); to = entry(t1, ci, cjo);
            ++i;
        };
        // interior range of the wave: [lo, hi) = rows where every live lane has both cells interior
        // (no border cell, no last row / column, nothing starts or ends)
        int lo = 0, hi = INT32_MAX, wtrip = trip;
        if (fast && v0) {
            lo = max(if0, v1 ? if1 : 0) + gl + 1;
            hi = min(ie0, v1 ? ie1 : ie0) + gl;
        }
        for (int d = 32; d > 0; d >>= 1) {
            lo = max(lo, __shfl_xor(lo, d)); hi = min(hi, __shfl_xor(hi, d)); wtrip = max(wtrip, __shfl_xor(wtrip, d));
        }
        // (the three bounds are wave-uniform by construction: keep them, and the loops, scalar)
        wtrip = __builtin_amdgcn_readfirstlane(wtrip);
        lo = __builtin_amdgcn_readfirstlane(min(lo, wtrip)); hi = __builtin_amdgcn_readfirstlane(min(hi, wtrip));
        int tau = 0;
        for (; tau < lo; ++tau) row(tau);
        if (tau < hi) {
            // ---- interior rows in KEY FORM ----------------------------------------------------
            // Nearly all rows. A cell is K = score * 4 + code with code 2 = diagonal, 1 = left,
            // 0 = top, so ONE max3 over the three candidate keys picks the winner with the
            // reference's tie order (diagonal > left > top). Every cell also publishes the keys it
            // offers as a left source (XL: + extension if it came from the left itself, else + open,
            // code 1) and as a top source (XT, code 0), so a consumer spends no instruction on its
            // neighbours' back pointers. Columns outside the band offer kNeverK for ever, which
            // replaces all per-cell validity tests; identities travel beside the keys.
            // Keys are kept biased by 2^31 and compared unsigned, so that 0 is "never": what a DPP
            // read past the pair's 16 lanes delivers by itself (bound_ctrl), and what `& live` leaves.
            constexpr uint32_t kBias = 0x80000000u;
            uint32_t G1 = (uint32_t)(gap * 4 + 1), E1 = (uint32_t)(ext * 4 + 1), G0 = (uint32_t)(gap * 4), E0 = (uint32_t)(ext * 4);
            // (held in vector registers: a select between two scalar constants under a scalar mask
            // would have to re-materialise one of them on every row)
            asm volatile("" : "+v"(G1), "+v"(E1), "+v"(G0), "+v"(E0));
            const uint32_t live0 = v0 ? ~0u : 0u, live1 = v1 ? ~0u : 0u;
            auto to_key = [&](int sc, int m, uint32_t live, uint32_t &K, uint32_t &N, uint32_t &XL, uint32_t &XT) {
                const uint32_t Kc = (uint32_t)(sc * 4) + kBias;
                const int b = m & 3;
                K = Kc | (b == BK_LEFT ? 1u : (b == BK_TOP ? 0u : 2u));
                N = (uint32_t)(m >> 2);
                XL = (Kc + (b == BK_LEFT ? E1 : G1)) & live;
                XT = (Kc + (b == BK_TOP ? E0 : G0)) & live;
            };
            uint32_t K0, N0, XL0, XT0, K1, N1, XL1, XT1;
            to_key(sc0, m0, live0, K0, N0, XL0, XT0);
            to_key(sc1, m1, live1, K1, N1, XL1, XT1);
            // LDS byte addresses of the NEXT row's residues, advanced by one per row (no clamping: live
            // cells are inside both sequences on interior rows, other lanes read padding or neighbours)
            const uint8_t *a1 = sb + o1 + i, *a2 = sb + o2 + c1 + i + bl;
            auto step = [&]() {
                const int ci_n = *a1, cjo_n = *a2;
                const int cje_n = cjo;
                ++a1; ++a2;
                {   // ---- even column: left = lane g-1's odd cell, top = own odd cell (both of the step before) ----
                    const uint32_t lXL = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)XL1, 0x111, 0xF, 0xF, true);
                    const uint32_t lN = (uint32_t)dpp_row_shr1((int)N1);
                    const uint32_t K = max(max((K0 & ~3u) + (uint32_t)te.x, lXL), XT1);
                    const uint32_t b = K & 3u, Kc = K & ~3u;
                    const uint64_t isL = lanes_equal(b, 1u), isT = lanes_equal(b, 0u);
                    N0 = lane_select(isL, lN, lane_select(isT, N1, N0 + (uint32_t)te.y));
                    K0 = K;
                    XL0 = (Kc + lane_select(isL, E1, G1)) & live0; XT0 = (Kc + lane_select(isT, E0, G0)) & live0;
                }
                {   // ---- odd column: left = own even cell, top = lane g+1's even cell (both of this step) ----
                    const uint32_t rXT = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)XT0, 0x101, 0xF, 0xF, true);
                    const uint32_t rN = (uint32_t)dpp_row_shl1((int)N0);
                    const uint32_t K = max(max((K1 & ~3u) + (uint32_t)to.x, XL0), rXT);
                    const uint32_t b = K & 3u, Kc = K & ~3u;
                    const uint64_t isL = lanes_equal(b, 1u), isT = lanes_equal(b, 0u);
                    N1 = lane_select(isL, N0, lane_select(isT, rN, N1 + (uint32_t)to.y));
                    K1 = K;
                    XL1 = (Kc + lane_select(isL, E1, G1)) & live1; XT1 = (Kc + lane_select(isT, E0, G0)) & live1;
                }
                __builtin_amdgcn_sched_barrier(0);
                cje = cje_n; cjo = cjo_n; ci = ci_n;
                te = entry(t0, ci, cje); to = entry(t1, ci, cjo);
                ++i;
            };
            for (; tau < hi; ++tau) step();  // (unrolling makes the compiler turn the selects into branches)
            // back to score / (identities << 2 | back pointer) for the closing rows (from the keys
            // alone: no comparison result is carried out of the loop)
            asm volatile("" : "+v"(K0), "+v"(K1));
            auto from_key = [&](uint32_t K, uint32_t N, int &sc, int &m) {
                sc = (int)(K - kBias) >> 2;
                m = (int)(N << 2) | ((K & 3u) == 1u ? BK_LEFT : ((K & 3u) == 0u ? BK_TOP : BK_DIAG));
            };
            if (v0) from_key(K0, N0, sc0, m0);
            if (v1) from_key(K1, N1, sc1, m1);
        }
        for (; tau < wtrip; ++tau) row(tau);
        // ---- end cell = last cell of its column (see band_align_wave) ----------------------
        int ce;
        if (len2 - bl < len1) ce = 0;
        else if (len1 + pr.band_right < len2) ce = bw - 1;
        else ce = len2 - len1 - bl;
        ce = max(ce, 0);
        const int src = (lane & 48) + (ce >> 1);
        const int e0 = __shfl(m0, src), e1 = __shfl(m1, src);
        const int iden = ((ce & 1) ? e1 : e0) >> 2;
        if (fast && gl == 0) {
            const uint32_t k1r = real_of(S, k1);
            bool ok = iden >= req_aa1[k1r];
            if (ok) {
                const float pc = (float)iden / (float)len1;
                ok = !((double)pc < cluster_thd);
            }
            pairs[p].iden = iden;
            pairs[p].flags = pr.flags | F_ALIGNED | (ok ? F_ACCEPT : 0u);
            if (ok && sel.accepted_out) sel.accepted_out[k1r - b0] = 1;
            if (ok && best)
                atomicMin(&best[k1r - b0], ((unsigned long long)(k1 != k1r) << 63) | ((unsigned long long)pr.minc << 32) | key_flag | pr.r);
        }
    }
}

// Aligns pairs [*d_begin, *d_npairs) that passed the diagonal test. With `best` given, an
// accepted pair is folded into best[q - b0] = min(minc << 32 | key_flag | p.r): the
// 64-bit minimum is the first accepted candidate in the sequential order.
__global__ __launch_bounds__(256) void align_kernel(DevSeqs S, const uint32_t *__restrict__ rep_seq,
                                                   Pair *__restrict__ pairs, PairSel sel,
                                                   const int32_t *__restrict__ req_aa1, double cluster_thd,
                                                   uint32_t b0, unsigned long long *__restrict__ best,
                                                   uint32_t key_flag, int wide_only, int max_sum,
                                                   const uint32_t *__restrict__ n_wide) {
    __shared__ int8_t sub[kNAA1 * kNAA1];
    if (wide_only && *n_wide == 0u) return;      // (nearly always: align16_kernel has met no pair it cannot take)
    for (int c = threadIdx.x; c < kNAA1 * kNAA1; c += 256)
        sub[c] = S.nt ? (int8_t)(c / kNAA1 == c % kNAA1 ? 2 : -2) : kBlosum62_dev[c];
    __syncthreads();
    const int gap_open = S.nt ? -6 : kGapOpen, gap_ext = S.nt ? -1 : kGapExt;
    const uint32_t n = sel_count(sel);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (uint32_t w = blockIdx.x * 4 + wave; w < n; w += gridDim.x * 4) {
        const uint32_t p = sel_pair(sel, w);
        const Pair pr = pairs[p];
        if ((pr.flags & (F_DIAG_PASS | F_BAND_OK | F_TOO_BIG | F_ALIGNED)) != (F_DIAG_PASS | F_BAND_OK)) continue;
        const uint32_t k1 = pr.q, k2 = rep_seq ? rep_seq[pr.r] : pr.r;
        const int len1 = (int)S.len[k1], len2 = (int)S.len[k2];
        if (wide_only && !pair_is_wide(len1, len2, pr.band_left, pr.band_right, max_sum)) continue;
        const int iden = band_align_wave(S.res + S.off[k1], S.res + S.off[k2], len1, len2, pr.band_left,
                                         pr.band_center, pr.band_right, sub, gap_open, gap_ext);
        const uint32_t k1r = real_of(S, k1);
        bool ok = iden >= req_aa1[k1r];
        if (ok) {
            const float pc = (float)iden / (float)len1;
            ok = !((double)pc < cluster_thd);
        }
        if (lane == 0) {
            pairs[p].iden = iden;
            pairs[p].flags = pr.flags | F_ALIGNED | (ok ? F_ACCEPT : 0u);
            if (ok && sel.accepted_out) sel.accepted_out[k1r - b0] = 1;
            if (ok && best)
                atomicMin(&best[k1r - b0], ((unsigned long long)(k1 != k1r) << 63) | ((unsigned long long)pr.minc << 32) | key_flag | pr.r);
        }
    }
}

}  // namespace

// ========================================================================================
// host side: one call = upload, word lists, windows, in-order resolution, download
// ========================================================================================
namespace {

// Kernels write results straight into these buffers (publish_kernel) and read lists from them, and
// the host polls for completion instead of making a synchronising call: the memory must be
// COHERENT (fine-grained, never cached by the GPU), or the device could serve a re-used list from
// its L2 and leave published records there.
constexpr unsigned kPinnedFlags = hipHostMallocCoherent | hipHostMallocMapped;
template <typename T>
struct Pinned {  // page-locked host staging buffer; bound to a context slot it outlives the call
    T *p = nullptr;
    size_t cap = 0;
    pgx_ctx *ctx = nullptr;
    int slot = -1;
    ~Pinned() { if (p && !ctx) (void)hipHostFree(p); }
    void bind(pgx_ctx *c, int s) { ctx = c; slot = s; }
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        if (ctx) {
            if ((int)ctx->host_arena.size() <= slot) ctx->host_arena.resize((size_t)slot + 1, {nullptr, 0});
            auto &a = ctx->host_arena[(size_t)slot];
            const size_t bytes = n * sizeof(T);
            if (!a.first || a.second < bytes) {
                if (a.first) (void)hipHostFree(a.first);
                a = {nullptr, 0};
                const size_t want = bytes + bytes / 2 + 4096;
                hipError_t e = hipHostMalloc(&a.first, want, kPinnedFlags);
                if (e != hipSuccess) return e;
                a.second = want;
            }
            p = static_cast<T *>(a.first);
            cap = a.second / sizeof(T);
            return hipSuccess;
        }
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = n + n / 2 + 1024;
        return hipHostMalloc((void **)&p, cap * sizeof(T), kPinnedFlags);
    }
};

enum : uint8_t { ST_OPEN = 0, ST_MEMBER = 1, ST_REP = 2,
                 ST_ABSENT = 3,    // memory-chunked rule: placed by an earlier chunk's sweep, not a member of this window any more
                 ST_NONE = 4 };    // ... a sweep found no representative for it: it stays in the game

#define LAUNCH_CHECK() PGX_HIP(hipGetLastError())

// The window loop waits on the stream a few times per window for tens of microseconds;
// polling returns as soon as the queue drains instead of after an interrupt wake-up.
static double g_wait_s = 0.0;  // host time spent waiting for the stream (PGX_TRACE)
static inline hipError_t spin_sync(hipStream_t st) {
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e;
    while ((e = hipStreamQuery(st)) == hipErrorNotReady) {
    }
    g_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return e;
}

template <int NCAP, int THREADS>
int launch_words_hash(pgx_ctx *ctx, hipStream_t st, const uint8_t *res, const uint64_t *off, const uint32_t *len,
                      uint32_t k0, uint32_t k1, int word_len, int base, int nt, uint32_t *wcode, uint16_t *wmult,
                      uint32_t *wcnt) {
    if (k1 <= k0) return PGX_OK;
    ProfScope prof(ctx, "words_kernel", st);
    words_hash_kernel<NCAP, THREADS><<<k1 - k0, THREADS, 0, st>>>(res, off, len, k0, k1, word_len, base, nt, wcode, wmult,
                                                                  wcnt);
    LAUNCH_CHECK();
    return PGX_OK;
}

template <int NCAP, int THREADS>
int launch_words(pgx_ctx *ctx, hipStream_t st, const uint8_t *res, const uint64_t *off, const uint32_t *len,
                 uint32_t k0, uint32_t k1, int word_len, int base, int nt, uint32_t *wcode, uint16_t *wmult,
                 uint32_t *wcnt) {
    if (k1 <= k0) return PGX_OK;
    ProfScope prof(ctx, "words_kernel", st);
    words_kernel<NCAP, THREADS><<<k1 - k0, THREADS, 0, st>>>(res, off, len, k0, k1, word_len, base, nt, wcode, wmult,
                                                             wcnt);
    LAUNCH_CHECK();
    return PGX_OK;
}

}  // namespace

static int cluster_greedy_impl(pgx_ctx *ctx, const uint8_t *d_residues, const uint64_t *d_offsets,
                               uint32_t n_in, uint64_t total_in, const pgx_cluster_params *P,
                               int32_t *out_cluster, int32_t *out_member, float *out_identity,
                               uint8_t *out_strand, uint32_t *out_n_clusters,
                               pgx_cluster_stats *stats, void *stream_);

// The per-sequence host loops before and after the window loop (order, offsets, thresholds, outputs) run on a few
// threads: with a million sequences they were a tenth of the call. fn(t, begin, end) for contiguous shares.
// A few worker threads, started at the first large call and parked between regions.
namespace {
class HostPool {
public:
    static HostPool &get() { static HostPool p; return p; }
    unsigned size() const { return (unsigned)workers_.size() + 1; }
    // job(t) for t = 0 .. size()-1, t = 0 on the calling thread; returns when all are done
    void run(const std::function<void(unsigned)> &job) {
        std::lock_guard<std::mutex> one(run_m_);   // (callers on several threads, e.g. one context each, take turns)
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = &job; pending_ = (unsigned)workers_.size(); ++gen_;
        }
        cv_.notify_all();
        job(0u);
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&] { return pending_ == 0; });
        job_ = nullptr;
    }
private:
    HostPool() {
        const unsigned hw = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
        for (unsigned t = 1; t < hw; ++t) workers_.emplace_back([this, t] { loop(t); });
    }
    ~HostPool() {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; ++gen_; }
        cv_.notify_all();
        for (auto &w : workers_) w.join();
    }
    void loop(unsigned t) {
        unsigned long long seen = 0;
        for (;;) {
            const std::function<void(unsigned)> *job;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                job = job_;
            }
            (*job)(t);
            std::lock_guard<std::mutex> lk(m_);
            if (--pending_ == 0) done_.notify_one();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_, run_m_;
    std::condition_variable cv_, done_;
    const std::function<void(unsigned)> *job_ = nullptr;
    unsigned pending_ = 0;
    unsigned long long gen_ = 0;
    bool stop_ = false;
};
}  // namespace
static unsigned host_threads(size_t n) { return n < (1u << 17) ? 1u : HostPool::get().size(); }
template <class F>
static void parallel_for(size_t n, unsigned nt, F fn) {
    if (nt <= 1) { fn(0u, (size_t)0, n); return; }
    const size_t per = (n + nt - 1) / nt;
    HostPool::get().run([&](unsigned t) { fn(t, std::min(n, t * per), std::min(n, (t + 1) * per)); });
}

extern "C" uint32_t pgx_cluster_window_cap(const pgx_cluster_params *P) {
    if (!P) return 0;
    const bool both = P->alphabet == 1 && P->both_strands != 0;
    // nucleotide rules at the reference's -n 5 -c 0.8 pass any pair that shares one word: small windows
    uint32_t w = P->alphabet == 1 ? (both ? 512u : 1024u) : 65536u;     // (nucleotides, 400-genome set: 512 -> 147 ms, 1024 -> 168, 2048 -> 231, 256 -> 173; proteins: the chunk volume bounds the windows first)
    if (P->batch_size > 0) w = (uint32_t)P->batch_size;
    if (const char *e = std::getenv("PGX_WINDOW")) { const long v = std::atol(e); if (v > 0) w = (uint32_t)v; }
    w = std::max(w, 64u);
    w = std::min(w, kWindowMax);
    return (w + 63u) & ~63u;
}

extern "C" int pgx_cluster_greedy_dev(pgx_ctx *ctx, const uint8_t *d_residues, const uint64_t *d_offsets,
                                      uint32_t n_in, uint64_t total_in, const pgx_cluster_params *P,
                                      int32_t *out_cluster, int32_t *out_member, float *out_identity,
                                      uint8_t *out_strand, uint32_t *out_n_clusters,
                                      pgx_cluster_stats *stats, void *stream_) {
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = cluster_greedy_impl(ctx, d_residues, d_offsets, n_in, total_in, P, out_cluster, out_member,
                                       out_identity, out_strand, out_n_clusters, stats, stream_);
    if (std::getenv("PGX_TRACE"))
        fprintf(stderr, "[pgx] call total (incl. clean-up)    %8.2f ms\n",
                1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    if (rc != PGX_OK && ctx) {
        // a failed call may leave kernels in flight that use the context's workspace:
        // drain them, so that the context stays usable (the error text is the caller's to read)
        (void)hipStreamSynchronize(static_cast<hipStream_t>(stream_));
        (void)hipStreamSynchronize(ctx->stream2);
        (void)hipGetLastError();
    }
    return rc;
}

static int cluster_greedy_impl(pgx_ctx *ctx, const uint8_t *d_residues, const uint64_t *d_offsets,
                               uint32_t n_in, uint64_t total_in, const pgx_cluster_params *P,
                               int32_t *out_cluster, int32_t *out_member, float *out_identity,
                               uint8_t *out_strand, uint32_t *out_n_clusters,
                               pgx_cluster_stats *stats, void *stream_) {
    PGX_REQUIRE(ctx && P, "NULL argument");
    PGX_REQUIRE(n_in == 0 || (d_residues && d_offsets), "NULL sequence arrays");
    PGX_REQUIRE(out_cluster && out_member && out_identity, "NULL output arrays");
    PGX_REQUIRE(P->alphabet == 0 || P->alphabet == 1, "alphabet must be 0 (protein) or 1 (nucleotide)");
    const bool nt = P->alphabet == 1;
    const bool both = nt && P->both_strands != 0;
    PGX_REQUIRE(P->word_len >= 2 && P->word_len <= (nt ? 11 : kMaxWordLen), "word_len must be 2..5 (protein) or 2..11 (nucleotide)");
    PGX_REQUIRE(P->identity >= 0.4 && P->identity <= 1.0, "identity must be 0.4..1.0");
    PGX_REQUIRE(P->band_width >= 1 && P->band_width <= kMaxBand, "band_width must be 1..64");
    PGX_REQUIRE(P->min_length >= P->word_len - 1, "min_length must be at least word_len - 1");
    PGX_REQUIRE(P->batch_size >= 0, "batch_size must not be negative");
    PGX_REQUIRE(P->shard_count >= 0 && (P->shard_count == 0 ? P->shard_index == 0 : (P->shard_index >= 0 && P->shard_index < P->shard_count)),
                "shard_index must be in [0, shard_count)");
    // the exchange of the record-sharded mode: the caller's callback, or the context's own RCCL communicator
    const bool native_exchange = !P->exchange && P->shard_count >= 1 && ctx->comm;
    const bool exchanging = P->exchange || native_exchange;
    PGX_REQUIRE(P->shard_count <= 1 || exchanging, "shard_count > 1 needs an exchange callback or a communicator (pgx_rccl_comm_create)");
    PGX_REQUIRE(!native_exchange || (ctx->comm_world == P->shard_count && ctx->comm_rank == P->shard_index),
                "shard_index / shard_count differ from the rank / size of the context's communicator");
    PGX_REQUIRE(!P->exchange || (P->exchange_send && P->exchange_recv),
                "the exchange callback needs exchange_send / exchange_recv (2 slots of PGX_EXCHANGE_WORDS uint64 per process, device memory)");
    PGX_HIP(hipSetDevice(ctx->device_id));
    hipStream_t st = (hipStream_t)stream_;
    const auto t_call0 = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t) {
        return 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count();
    };
    const bool trace_phases = std::getenv("PGX_TRACE") != nullptr;
    double t_mark = 0.0;
    auto phase = [&](const char *what) {   // PGX_TRACE: host wall time of the phases outside the window loop
        if (!trace_phases) return;
        const double now = ms_since(t_call0);
        fprintf(stderr, "[pgx] %-28s %8.2f ms\n", what, now - t_mark);
        t_mark = now;
    };
    pgx_cluster_stats S{};
    S.n_input = n_in;
    if (out_n_clusters) *out_n_clusters = 0;
    const unsigned nth = host_threads(n_in);
    parallel_for(n_in, nth, [&](unsigned, size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            out_cluster[i] = -1; out_member[i] = -1; out_identity[i] = 0.f;
            if (out_strand) out_strand[i] = 0;
        }
    });
    if (n_in == 0) { if (stats) *stats = S; return PGX_OK; }

    // ---- A.2 / A.3: letters -> indices, stable descending-length order ---------------------
    // The GPU counts letters and later gathers/encodes the residues; the O(n) bookkeeping
    // (counting sort by length, offsets, thresholds) is done on the host from the lengths.
    DevBuf d_in_len;
    d_in_len.ctx = ctx; d_in_len.slot = 0;
    PGX_HIP(d_in_len.alloc((size_t)n_in * 4));
    {
        ProfScope prof(ctx, "seq_len_kernel", st);
        seq_len_kernel<<<(n_in + 3) / 4, 256, 0, st>>>(d_residues, d_offsets, n_in, total_in, d_in_len.as<uint32_t>());
    }
    LAUNCH_CHECK();
    // (the per-sequence host arrays live in the context's scratch: no allocation, page faults or frees per call)
    HostVec<uint32_t> in_len(ctx, 0, n_in);
    PGX_REQUIRE(in_len.ok(), "out of host memory");
    PGX_HIP(hipMemcpyAsync(in_len.data(), d_in_len.p, (size_t)n_in * 4, hipMemcpyDeviceToHost, st));
    PGX_HIP(hipStreamSynchronize(st));
    phase("  seq_len kernel + copy");
    uint32_t max_len = 0;
    {
        std::vector<uint32_t> mx(nth, 0u);
        parallel_for(n_in, nth, [&](unsigned t, size_t b, size_t e) {
            uint32_t m = 0;
            for (size_t i = b; i < e; ++i) m = std::max(m, in_len[i]);
            mx[t] = m;
        });
        for (uint32_t m : mx) max_len = std::max(max_len, m);
    }
    if (max_len > kMaxLen) {
        pgx_set_error("pgx_cluster_greedy: sequence of %u residues exceeds the supported maximum %u", max_len, kMaxLen);
        return PGX_ERR_CAPACITY;
    }
    // stable counting sort by descending length: a histogram per thread over its share of the input, then
    // every thread places its share behind the shares before it
    const size_t n_bkt = (size_t)max_len + 2;
    std::vector<uint32_t> bucket(n_bkt * nth, 0);   // [t][l]
    parallel_for(n_in, nth, [&](unsigned t, size_t b, size_t e) {
        uint32_t *h = bucket.data() + n_bkt * t;
        for (size_t i = b; i < e; ++i)
            if ((int)in_len[i] > P->min_length) h[max_len - in_len[i]]++;
    });
    uint32_t n = 0;
    for (size_t l = 0; l < n_bkt; ++l)
        for (unsigned t = 0; t < nth; ++t) { const uint32_t c = bucket[n_bkt * t + l]; bucket[n_bkt * t + l] = n; n += c; }
    HostVec<uint32_t> order(ctx, 1, n);
    PGX_REQUIRE(order.ok(), "out of host memory");
    parallel_for(n_in, nth, [&](unsigned t, size_t b, size_t e) {
        uint32_t *h = bucket.data() + n_bkt * t;
        for (size_t i = b; i < e; ++i)
            if ((int)in_len[i] > P->min_length) order[h[max_len - in_len[i]]++] = (uint32_t)i;
    });
    if (n == 0) { if (stats) *stats = S; return PGX_OK; }
    PGX_REQUIRE(n < (1u << 30), "too many sequences for the index entries (30 bits)");
    uint32_t mshift = 8;                         // smallest field that holds n (and the strike-out value above it)
    while ((1u << mshift) <= n) ++mshift;

    phase("  counting sort");
    // sequences n .. 2n-1 are the reverse complements (nucleotides, both strands)
    const uint32_t nv = both ? 2 * n : n;
    HostVec<uint64_t> h_off(ctx, 2, (size_t)nv + 1);
    HostVec<uint32_t> h_len(ctx, 3, nv);
    PGX_REQUIRE(h_off.ok() && h_len.ok(), "out of host memory");
    HostVec<uint32_t> h_pkoff(ctx, 4, (size_t)nv + 1);
    PGX_REQUIRE(h_pkoff.ok(), "out of host memory");
    // lengths in sorted order; residue and packed-word offsets = prefix sums (per-share sums first, then the shares)
    std::vector<uint64_t> sum_len(nth + 1, 0), sum_pk(nth + 1, 0);
    parallel_for(nv, nth, [&](unsigned t, size_t b, size_t e) {
        uint64_t sl = 0, sp = 0;
        for (size_t k = b; k < e; ++k) {
            const uint32_t L = in_len[order[k < n ? k : k - n]];
            h_len[k] = L; sl += L; sp += (L + 5) / 6;
        }
        sum_len[t + 1] = sl; sum_pk[t + 1] = sp;
    });
    for (unsigned t = 0; t < nth; ++t) { sum_len[t + 1] += sum_len[t]; sum_pk[t + 1] += sum_pk[t]; }
    const uint64_t total = sum_len[nth];
    if (sum_pk[nth] > 0xFFFFFFF0ull) { pgx_set_error("pgx_cluster_greedy: too many residues for 32-bit packed offsets"); return PGX_ERR_CAPACITY; }
    h_off[0] = 0; h_pkoff[0] = 0;
    parallel_for(nv, nth, [&](unsigned t, size_t b, size_t e) {
        uint64_t so = sum_len[t], sp = sum_pk[t];
        for (size_t k = b; k < e; ++k) {
            so += h_len[k]; sp += (h_len[k] + 5) / 6;
            h_off[k + 1] = so; h_pkoff[k + 1] = (uint32_t)sp;
        }
    });
    phase("lengths + order");
    // per-query thresholds in double, exactly as the sequential rule computes them
    // They depend on the length only: one small table per threshold (host, in double), expanded per sequence on
    // the device (thresholds_kernel) instead of three host passes and three uploads of n integers.
    HostVec<int32_t> h_thr(ctx, 5, 3 * ((size_t)max_len + 1));
    PGX_REQUIRE(h_thr.ok(), "out of host memory");
    {
        int32_t *t_aa1 = h_thr.data(), *t_aas = t_aa1 + max_len + 1, *t_aan = t_aas + max_len + 1;
        for (uint32_t L = 0; L <= max_len; ++L) {
            const int len = (int)L;
            const int aa1 = (int)(P->identity * (double)len);
            t_aa1[L] = aa1;
            if (P->identity > 0.95) {
                t_aas[L] = len - (nt ? 4 : 2) + 1 - (len - aa1) * (nt ? 4 : 2);
                t_aan[L] = len - P->word_len + 1 - (len - aa1) * P->word_len;
            } else {
                t_aas[L] = (int)(P->aas_cutoff * (double)len);
                t_aan[L] = (int)(P->aan_cutoff * (double)len);
            }
        }
        std::vector<uint8_t> bad(nth, 0);
        parallel_for(n, nth, [&](unsigned t, size_t b, size_t e) {   // (the lengths that occur: run boundaries of the sorted list)
            for (size_t k = b; k < e; ++k)
                if ((k == b || h_len[k] != h_len[k - 1]) && t_aas[h_len[k]] < 1) bad[t] = 1;
        });
        for (uint8_t b : bad) PGX_REQUIRE(!b, "aas_cutoff too small: every sequence needs required_aas >= 1");
        S.sum_len_queries = h_off[n];
    }
    S.n_clustered = n;

    // ---- device buffers ------------------------------------------------------------------
    uint32_t n_codes = 1;
    for (int t = 0; t < P->word_len; ++t) n_codes *= nt ? 4u : (uint32_t)kNAA1;
    const uint32_t window_cap = pgx_cluster_window_cap(P);           // queries per window
    PGX_REQUIRE(!exchanging || window_cap <= PGX_EXCHANGE_KEYS, "window larger than PGX_EXCHANGE_KEYS");
    uint32_t pair_cap = 4u << 20;  // grows per window for nucleotides, whose word filter passes almost every pair
    // blocks: nucleotide rules pass nearly every pair, so 512 members (all their pairs fit); proteins 4096
    uint32_t block_cap = std::min(nt ? 512u : kBlockCap, window_cap);
    const uint32_t pair_cap_k = nt ? 512u * 512u + 16u : (1u << 20);
    // A window = consecutive members: at most window_cap of them, in at most kMaxChunks chunks of bounded word
    // volume (see certain_kernel: the words of one chunk must cover a fraction of the code space well below
    // the word threshold fraction, or unrelated members alone would defeat the discovery test).
    const double thr_frac = P->identity > 0.95 ? std::max(0.0, 1.0 - (1.0 - P->identity) * P->word_len) : P->aan_cutoff;
    // (a code space too small for that -- nucleotide 5-mers, tiny thresholds -- gets one chunk per window: the
    // discovery test then certifies little and the exact block resolution does the work)
    // (0.8 of the threshold fraction: unrelated members alone then share 17 % of their words with a chunk by chance,
    // against a threshold of 23 %. Measured, cfg-3s / cfg-4 / 300,000 unrelated random proteins: 0.4 -> 101 ms / 1.57 s /
    // 49 ms, 0.8 -> 94 ms / 1.3 s / 64 ms, 1.0 -> 93 ms / 1.26 s / 82 ms -- families fill far less of the code space than
    // their word count, so larger chunks mean fewer, larger windows; without families the test starts to fail.)
    static const double chunk_frac = std::getenv("PGX_CHUNK_FRAC") ? std::atof(std::getenv("PGX_CHUNK_FRAC")) : 0.8;
    // The fraction ADAPTS: a window that needed three or more blocks (the discovery test certified too little: members
    // without families) halves it for the windows after, a window with at most one block raises it again.
    double cur_frac = chunk_frac;
    uint64_t chunk_words = (uint64_t)(cur_frac * thr_frac * (double)n_codes);
    const bool chunking = chunk_words >= 16384;
    // cd-hit's memory-chunked rule (pgx.h): flush positions in the sorted list; a window never crosses one
    std::vector<uint32_t> flush_at(P->chunk_boundaries, P->chunk_boundaries + (P->chunk_boundaries ? P->n_chunk_boundaries : 0u));
    for (size_t i = 0; i < flush_at.size(); ++i)
        PGX_REQUIRE(flush_at[i] > 0 && flush_at[i] < n && (i == 0 || flush_at[i] > flush_at[i - 1]),
                    "chunk_boundaries must be strictly increasing positions in (0, number of clustered sequences)");
    auto form_window = [&](uint32_t b0, Chunks &C, uint32_t limit) -> uint32_t {
        C.n = 1; C.begin[0] = 0;
        uint64_t in_chunk = 0;
        uint32_t q = 0;
        for (; b0 + q < limit && q < window_cap; ++q) {
            const uint64_t w = h_len[b0 + q];
            if (chunking && in_chunk && in_chunk + w > chunk_words) {
                if (C.n == kMaxChunks) break;
                C.begin[C.n++] = q;
                in_chunk = 0;
            }
            in_chunk += w;
        }
        C.begin[C.n] = q;
        return q;
    };
    uint64_t max_window_words = 0;
    uint32_t max_chunks = 1;   // the tag records are only as long as some window has chunks
    {
        Chunks C;
        size_t nf = 0;
        for (uint32_t b0 = 0, nbw; b0 < n; b0 += nbw) {
            while (nf < flush_at.size() && flush_at[nf] <= b0) ++nf;
            nbw = form_window(b0, C, nf < flush_at.size() ? flush_at[nf] : n);
            max_window_words = std::max<uint64_t>(max_window_words, h_off[b0 + nbw] - h_off[b0]);
            max_chunks = std::max(max_chunks, C.n);
        }
        if (!flush_at.empty())   // (the sweeps' windows start anywhere: no window holds more than the window_cap longest sequences)
            max_window_words = std::max<uint64_t>(max_window_words, h_off[std::min(n, window_cap)]);
    }
    // The chunk volume adapts (cur_frac below), so later windows are not the ones of this scan: bound a window's words
    // independently of the partition -- no more than the window_cap longest sequences, no more than kMaxChunks
    // chunks of the largest volume (a chunk closes before it would exceed that, or holds one sequence).
    if (chunking && flush_at.empty())
        max_window_words = std::max<uint64_t>(max_window_words,
            std::min<uint64_t>(h_off[std::min(n, window_cap)], (uint64_t)kMaxChunks * std::max<uint64_t>(chunk_words, max_len)));
    PGX_REQUIRE(max_window_words < 0xFFFFFFF0ull, "window too large");
    const bool need_gscratch = (uint64_t)max_len * 2 > kDiagLdsSmall;
    const uint32_t packed_len = nt ? kPackedLenNt : kPackedLenAa;   // queries beyond this: 64-bit histogram cells
    const uint32_t gs_cells = (max_len + 64u) & ~63u;
    // per workgroup: 2 L diagonals (one or two words each) + L query positions
    const uint32_t gs_stride = (max_len > packed_len ? 5u : 3u) * gs_cells;
    // diag: one wave per pair, ~11 workgroups fit a CU; giant sequences: fewer workgroups, at most ~4 GB of scratch
    const uint32_t diag_grid = (uint32_t)std::max<uint64_t>(64, std::min<uint64_t>(8192, (4ull << 30) / ((uint64_t)gs_stride * 4)));
    const uint32_t align_grid = 1024;

    DevBuf d_res, d_off, d_len, d_wcode, d_wmult, d_wcnt, d_aa1, d_aas, d_aan, d_lines, d_pool[2], d_idx,
        d_newbits, d_touched, d_first, d_best_own, d_rcvis, d_counters, d_visits, d_pairsW, d_pairsK, d_blk_list,
        d_ulist, d_new_list, d_flags, d_gscratch, d_order, d_list, d_gather, d_pk, d_pkoff,
        d_counters2, d_best2, d_flags2, d_pairsW2, d_gscratch2,   // second set of a window's own state (see `overlap`)
        d_thr, d_hugewords, d_xsend, d_xrecv;
    {   // all of them live in the context's workspace (slots 1..)
        DevBuf *all[] = {&d_res, &d_off, &d_len, &d_wcode, &d_wmult, &d_wcnt, &d_aa1, &d_aas, &d_aan, &d_lines,
                         &d_pool[0], &d_pool[1], &d_idx, &d_newbits, &d_touched, &d_first, &d_best_own, &d_rcvis,
                         &d_counters, &d_visits, &d_pairsW, &d_pairsK, &d_blk_list, &d_ulist, &d_new_list, &d_flags,
                         &d_gscratch, &d_order, &d_list, &d_gather, &d_pk, &d_pkoff,
                         &d_counters2, &d_best2, &d_flags2, &d_pairsW2, &d_gscratch2, &d_thr, &d_hugewords,
                         &d_xsend, &d_xrecv};
        int sl = 1;
        for (DevBuf *b : all) { b->ctx = ctx; b->slot = sl++; }
    }
    phase("thresholds + offsets");
    PGX_HIP(d_res.alloc(total + 16));
    PGX_HIP(d_off.alloc(((size_t)nv + 1) * 8));
    PGX_HIP(d_len.alloc((size_t)nv * 4));
    PGX_HIP(d_wcode.alloc((total + 16) * 4));
    PGX_HIP(d_wmult.alloc((total + 16) * 2));
    PGX_HIP(d_wcnt.alloc((size_t)nv * 4));
    PGX_HIP(d_pk.alloc(((size_t)h_pkoff[nv] + 16) * 4));
    PGX_HIP(d_pkoff.alloc(((size_t)nv + 1) * 4));
    PGX_HIP(hipMemcpyAsync(d_pkoff.p, h_pkoff.data(), ((size_t)nv + 1) * 4, hipMemcpyHostToDevice, st));
    PGX_HIP(d_aa1.alloc((size_t)n * 4));
    PGX_HIP(d_aas.alloc((size_t)n * 4));
    PGX_HIP(d_aan.alloc((size_t)n * 4));
    // the word index: all lines empty (one memset of n_codes * 128 B per call, e.g. 523 MB for protein 5-mers)
    PGX_HIP(d_lines.alloc((size_t)n_codes * sizeof(IndexLine)));
    PGX_HIP(hipMemsetAsync(d_lines.p, 0, (size_t)n_codes * sizeof(IndexLine), st));
    PGX_HIP(d_idx.alloc(16));
    static_assert(kSegs == 4, "eight codes per word of the round's map");
    const uint32_t bm_words = (n_codes / 8 + 2 + 3) & ~3u;   // words of the round's map of touched codes
    PGX_HIP(d_newbits.alloc(2 * (size_t)bm_words * 4 + 16));          // two maps, used by alternate append rounds
    PGX_HIP(hipMemsetAsync(d_newbits.p, 0, 2 * (size_t)bm_words * 4, st));
    PGX_HIP(d_touched.alloc((max_window_words + 16) * sizeof(Deferred)));   // entries set aside by an append round
    if (chunking && n > window_cap / 2) max_chunks = kMaxChunks;   // (smaller chunks later may need all of them)
    uint32_t tag_stride = 1;   // first-open tags: one record of tag_stride >= max_chunks words per code
    while (tag_stride < max_chunks) tag_stride *= 2;
    PGX_HIP(d_first.alloc((size_t)tag_stride * n_codes * 4 + 16));
    PGX_HIP(d_best_own.alloc((size_t)window_cap * 8));
    PGX_HIP(d_rcvis.alloc((size_t)window_cap * 8));
    PGX_HIP(d_blk_list.alloc((size_t)kBlockCap * 4));
    PGX_HIP(d_ulist.alloc((size_t)window_cap * 4));
    PGX_HIP(d_new_list.alloc((size_t)window_cap * 4));
    PGX_HIP(d_flags.alloc(4 * (size_t)window_cap));  // done, in-block, has-candidate, accepted
    PGX_HIP(d_list.alloc((size_t)pair_cap_k * 4));
    PGX_HIP(d_gather.alloc((size_t)pair_cap_k * sizeof(Pair)));
    PGX_HIP(d_counters.alloc(C_COUNT * 4));
    PGX_HIP(hipMemsetAsync(d_counters.p, 0, C_COUNT * 4, st));
    PGX_HIP(d_visits.alloc(8));
    PGX_HIP(d_pairsW.alloc((size_t)pair_cap * sizeof(Pair)));
    PGX_HIP(d_pairsK.alloc((size_t)pair_cap_k * sizeof(Pair)));
    if (need_gscratch) PGX_HIP(d_gscratch.alloc((size_t)diag_grid * gs_stride * 4));
    // Consecutive windows overlap on two streams (proteins, one process): the tail of window w -- the pass over
    // all members after its last block, that pass's diagonal tests and alignments, the close -- only READS the
    // index, and so does phase A of window w+1. Each window therefore has its own counters, best keys, member flags,
    // pair records and host copies (two sets, used alternately); window w+1 starts when the index of window w is
    // final (event after its last strike-out) and appends to the index only after window w's close (second event).
    // A few hundred pairs per evaluation keep a 60-us dependent chain per alignment on the path otherwise.
    // (Record-sharded mode too: the exchange buffers have one slot per window in flight, and every process enqueues its
    // collectives in the same order because the host logic is the same function of replicated results.)
    const bool overlap = !nt && !std::getenv("PGX_NO_OVERLAP");
    if (overlap) {
        PGX_HIP(d_counters2.alloc(C_COUNT * 4));
        PGX_HIP(hipMemsetAsync(d_counters2.p, 0, C_COUNT * 4, st));
        if (!exchanging) PGX_HIP(d_best2.alloc((size_t)window_cap * 8));
        PGX_HIP(d_flags2.alloc(4 * (size_t)window_cap));
        PGX_HIP(d_pairsW2.alloc((size_t)pair_cap * sizeof(Pair)));
        if (need_gscratch) PGX_HIP(d_gscratch2.alloc((size_t)diag_grid * gs_stride * 4));
    }
    // overflow pool of the index: sized before every window from the entries that exist and the most the
    // window can add (every array is re-allocated with twice the need: <= 8 words per entry in all)
    int pool_cur = 0;
    uint64_t pool_cap = 0;
    auto ensure_pool = [&](uint64_t entries, hipStream_t on) -> int {
        const uint64_t need = 8 * entries + 4096;
        if (need <= pool_cap) return PGX_OK;
        if (need > 0xFFFFFFF0ull) { pgx_set_error("pgx_cluster_greedy: word index too large for 32-bit pool offsets"); return PGX_ERR_CAPACITY; }
        const uint64_t want = std::min<uint64_t>(0xFFFFFFF0ull, need + need / 2);
        DevBuf &nb_ = d_pool[pool_cur ^ 1];
        PGX_HIP(nb_.alloc(want * 4));
        if (pool_cap) PGX_HIP(hipMemcpyAsync(nb_.p, d_pool[pool_cur].p, pool_cap * 4, hipMemcpyDeviceToDevice, on));
        pool_cur ^= 1;
        pool_cap = want;
        return PGX_OK;
    };
    {
        const uint32_t one = 1u;   // pool word 0 is never an array: ovf == 0 means "no overflow array"
        PGX_HIP(hipMemcpyAsync(d_idx.p, &one, 4, hipMemcpyHostToDevice, st));
    }

    PGX_HIP(d_order.alloc((size_t)n * 4));
    PGX_HIP(hipMemcpyAsync(d_order.p, order.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
    PGX_HIP(hipMemcpyAsync(d_off.p, h_off.data(), ((size_t)nv + 1) * 8, hipMemcpyHostToDevice, st));
    PGX_HIP(hipMemcpyAsync(d_len.p, h_len.data(), (size_t)nv * 4, hipMemcpyHostToDevice, st));
    {
        ProfScope prof(ctx, "encode_gather_kernel", st);
        encode_gather_kernel<<<(n + 3) / 4, 256, 0, st>>>(d_residues, d_offsets, d_order.as<uint32_t>(),
                                                          d_off.as<uint64_t>(), d_res.as<uint8_t>(), n, nt ? 1 : 0);
        if (both)
            revcomp_kernel<<<(n + 3) / 4, 256, 0, st>>>(d_res.as<uint8_t>(), d_off.as<uint64_t>(),
                                                        d_len.as<uint32_t>(), n);
        pack5_kernel<<<(nv + 3) / 4, 256, 0, st>>>(d_res.as<uint8_t>(), d_off.as<uint64_t>(), d_len.as<uint32_t>(),
                                                   d_pkoff.as<uint32_t>(), nv, d_pk.as<uint32_t>());
    }
    LAUNCH_CHECK();
    {   // thresholds per sequence from the per-length tables
        PGX_HIP(d_thr.alloc(3 * ((size_t)max_len + 1) * 4));
        PGX_HIP(hipMemcpyAsync(d_thr.p, h_thr.data(), 3 * ((size_t)max_len + 1) * 4, hipMemcpyHostToDevice, st));
        thresholds_kernel<<<(n + 255) / 256, 256, 0, st>>>(d_len.as<uint32_t>(), n, d_thr.as<int32_t>(), max_len + 1,
                                                          d_aa1.as<int32_t>(), d_aas.as<int32_t>(), d_aan.as<int32_t>());
        LAUNCH_CHECK();
    }
    PGX_HIP(hipMemsetAsync(d_visits.p, 0, 8, st));

    phase("alloc + upload + encode (enq)");
    // ---- word lists: size classes are contiguous because the order is by length ----------
    {
        const int wl = P->word_len;
        auto first_with_words_le = [&](uint32_t cap) {  // first k whose word count fits `cap`
            return (uint32_t)(std::partition_point(h_len.begin(), h_len.begin() + n,
                                                   [&](uint32_t L) { return L - wl + 1 > cap; }) - h_len.begin());
        };
        const uint32_t k32 = first_with_words_le(32768), k8 = first_with_words_le(8192),
                       k2 = first_with_words_le(2048), k1k = first_with_words_le(1023), k5 = first_with_words_le(512);
        int rc;
        uint8_t *r8 = d_res.as<uint8_t>(); uint64_t *o64 = d_off.as<uint64_t>(); uint32_t *l32 = d_len.as<uint32_t>();
        uint32_t *wc = d_wcode.as<uint32_t>(); uint16_t *wm = d_wmult.as<uint16_t>(); uint32_t *wn = d_wcnt.as<uint32_t>();
        const int base = nt ? 4 : kNAA1;
        for (uint32_t half = 0; half < (both ? 2u : 1u); ++half) {  // the reverse complements have the same lengths
            const uint32_t o = half * n;
            if (k32) {   // giant sequences (more than 32768 words): a table in global memory, a few at a time
                uint32_t slots = 65536;
                while (slots < 2u * max_len) slots *= 2;
                const uint32_t per_launch = std::min(k32, std::max(1u, (uint32_t)((256ull << 20) / ((uint64_t)slots * 8))));
                PGX_HIP(d_hugewords.alloc((size_t)per_launch * slots * 8));
                ProfScope prof(ctx, "words_kernel", st);
                for (uint32_t a = 0; a < k32; a += per_launch)
                    words_huge_kernel<<<std::min(per_launch, k32 - a), 1024, 0, st>>>(r8, o64, l32, o + a, o + std::min(k32, a + per_launch), wl, base,
                                                                                       nt, d_hugewords.as<uint32_t>(), slots, wc, wm, wn,
                                                                                       d_counters.as<uint32_t>() + C_ERR);
                LAUNCH_CHECK();
            }
            if ((rc = launch_words<32768, 1024>(ctx, st, r8, o64, l32, o + k32, o + k8, wl, base, nt, wc, wm, wn))) return rc;
            if ((rc = launch_words<8192, 1024>(ctx, st, r8, o64, l32, o + k8, o + k2, wl, base, nt, wc, wm, wn))) return rc;
            if ((rc = launch_words_hash<2048, 256>(ctx, st, r8, o64, l32, o + k2, o + k1k, wl, base, nt, wc, wm, wn))) return rc;
            if (k5 > k1k) {
                ProfScope prof(ctx, "words_kernel", st);
                words_wave_kernel<2048><<<(k5 - k1k + 3) / 4, 256, 0, st>>>(r8, o64, l32, o + k1k, o + k5, wl, base, nt, wc, wm, wn);
                LAUNCH_CHECK();
            }
            if (n > k5) {
                ProfScope prof(ctx, "words_kernel", st);
                words_wave_kernel<1024><<<(n - k5 + 3) / 4, 256, 0, st>>>(r8, o64, l32, o + k5, o + n, wl, base, nt, wc, wm, wn);
                LAUNCH_CHECK();
            }
        }
    }
    DevSeqs DS{d_res.as<uint8_t>(), d_off.as<uint64_t>(), d_len.as<uint32_t>(),
               d_wcode.as<uint32_t>(), d_wmult.as<uint16_t>(), d_wcnt.as<uint32_t>(),
               d_pk.as<uint32_t>(), d_pkoff.as<uint32_t>(), n, nt ? 4 : kNAA1, nt ? 4 : 2, nt ? 1 : 0, mshift};
    HostVec<uint32_t> h_wcnt(ctx, 8, n);
    PGX_REQUIRE(h_wcnt.ok(), "out of host memory");
    PGX_HIP(hipMemcpyAsync(h_wcnt.data(), d_wcnt.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));

    phase("word lists (enqueue)");
    // ---- windows --------------------------------------------------------------------------
    std::vector<uint32_t> rep_seq;            // representative index -> sorted sequence index
    HostVec<int32_t> cluster_of(ctx, 9, n, -1);   // sorted sequence index -> cluster
    HostVec<int32_t> iden_of(ctx, 10, n, -1);     // identical residues against the representative, -1 = is one
    PGX_REQUIRE(cluster_of.ok() && iden_of.ok(), "out of host memory");
    Pinned<Pair> hW, hK;
    Pinned<unsigned long long> h_best, h_rcvis;
    Pinned<uint32_t> h_cnt, h_blk, h_new, h_list, h_push;
    Pinned<Pair> h_gather;
    hW.bind(ctx, 0); hK.bind(ctx, 1); h_best.bind(ctx, 2); h_cnt.bind(ctx, 3); h_blk.bind(ctx, 4);
    h_new.bind(ctx, 5); h_rcvis.bind(ctx, 6); h_list.bind(ctx, 7); h_gather.bind(ctx, 8); h_push.bind(ctx, 9);
    constexpr uint32_t kPrefix = 65536;  // initial capacity (records) of the host copy of the window's pairs
    PGX_HIP(h_best.reserve(window_cap)); PGX_HIP(h_cnt.reserve(C_COUNT)); PGX_HIP(h_blk.reserve(kBlockCap));
    PGX_HIP(h_new.reserve(window_cap)); PGX_HIP(hK.reserve(pair_cap_k)); PGX_HIP(hW.reserve(kPrefix));
    PGX_HIP(h_rcvis.reserve(window_cap)); PGX_HIP(h_push.reserve(2 * (size_t)window_cap + 4 * kBlockCap));
    PGX_HIP(h_list.reserve(pair_cap_k)); PGX_HIP(h_gather.reserve(pair_cap_k));
    // what a window's CLOSE publishes has its own host buffers, one set per window in flight
    Pinned<Pair> hW2;
    Pinned<unsigned long long> h_best2;
    Pinned<uint32_t> h_ccnt, h_ccnt2;
    hW2.bind(ctx, 10); h_best2.bind(ctx, 11); h_ccnt.bind(ctx, 12); h_ccnt2.bind(ctx, 13);
    PGX_HIP(h_ccnt.reserve(C_COUNT));
    if (overlap) { PGX_HIP(hW2.reserve(kPrefix)); PGX_HIP(h_best2.reserve(window_cap)); PGX_HIP(h_ccnt2.reserve(C_COUNT)); }
    struct Events {   // (destroyed on every way out)
        hipEvent_t strike[2] = {nullptr, nullptr}, post[2] = {nullptr, nullptr};
        ~Events() { for (hipEvent_t e : strike) if (e) (void)hipEventDestroy(e); for (hipEvent_t e : post) if (e) (void)hipEventDestroy(e); }
    } ev;
    for (int i = 0; i < 2; ++i) {
        PGX_HIP(hipEventCreateWithFlags(&ev.strike[i], hipEventDisableTiming));
        PGX_HIP(hipEventCreateWithFlags(&ev.post[i], hipEventDisableTiming));
    }
    const hipStream_t st_main = st;
    uint32_t *const dc_set[2] = {d_counters.as<uint32_t>(), d_counters2.as<uint32_t>()};
    unsigned long long *d_rcvis_p = d_rcvis.as<unsigned long long>();
    // record-sharded mode (see pgx.h): this process filters and aligns the window members ql % shard_count ==
    // shard_index; the window's best keys live in the caller's exchange buffer and are all-gathered after
    // every evaluation
    const uint32_t shard_count = P->shard_count > 0 ? (uint32_t)P->shard_count : 1u;
    if (native_exchange) {   // the library's own exchange buffers, laid out as pgx.h asks of a caller's
        PGX_HIP(d_xsend.alloc((size_t)PGX_EXCHANGE_SLOTS * PGX_EXCHANGE_WORDS * 8));
        PGX_HIP(d_xrecv.alloc((size_t)PGX_EXCHANGE_SLOTS * shard_count * PGX_EXCHANGE_WORDS * 8));
    }
    unsigned long long *const x_send = native_exchange ? d_xsend.as<unsigned long long>() : static_cast<unsigned long long *>(P->exchange_send);
    const unsigned long long *const x_recv = native_exchange ? d_xrecv.as<unsigned long long>() : static_cast<const unsigned long long *>(P->exchange_recv);
    unsigned long long *d_best = exchanging ? x_send : d_best_own.as<unsigned long long>();
    const uint32_t shard_index = P->shard_count > 0 ? (uint32_t)P->shard_index : 0u;
    const bool count_replicated = shard_index == 0;  // work every process repeats is counted by the first one only
    // test hook (tests/test_gpu_cluster_sharded.py): PGX_INJECT_ERROR="<rank>:<n>" makes that process report a pair
    // buffer overflow with its n-th exchange, to check that every process then leaves at the same point
    int inject_rank = -1;
    uint64_t inject_at = 0, n_exchanges = 0;
    if (const char *e = std::getenv("PGX_INJECT_ERROR")) { unsigned long long at = 0; if (sscanf(e, "%d:%llu", &inject_rank, &at) == 2) inject_at = at; else inject_rank = -1; }
    uint64_t visits_rc = 0;
    unsigned long long *const d_best_set[2] = {d_best, exchanging ? d_best + PGX_EXCHANGE_WORDS : d_best2.as<unsigned long long>()};
    uint8_t *const d_flags_set[2] = {d_flags.as<uint8_t>(), d_flags2.as<uint8_t>()};
    std::vector<uint8_t> status(window_cap);
    HostVec<uint8_t> strand_of(ctx, 11, n, (uint8_t)0);
    PGX_REQUIRE(strand_of.ok(), "out of host memory");
    std::vector<unsigned long long> winner_key(window_cap);  // strand<<63 | minc<<32 | sequence index of the winner
    std::vector<uint32_t> new_reps, order_k, rank_of(window_cap), bucket_k(kBlockCap + 2), fill_k(kBlockCap + 2), flight;
    auto real = [&](uint32_t k) { return k >= n ? k - n : k; };
    auto pair_key = [&](const Pair &pp) {
        return ((unsigned long long)(pp.q >= n) << 63) | ((unsigned long long)pp.minc << 32) | pp.r;
    };
    uint64_t n_rounds = 0;

    uint64_t gpu_pairs = 0, gpu_aligned = 0, gpu_aligned_bytes = 0, filter_walk_words = 0;  // actual device work (reserved stats slots)
    auto account = [&](const Pair *pp, uint32_t cnt) {
        for (uint32_t i = 0; i < cnt; ++i) {
            ++gpu_pairs;
            if ((pp[i].flags & (F_DIAG_PASS | F_BAND_OK)) != (F_DIAG_PASS | F_BAND_OK)) continue;
            ++gpu_aligned;
            gpu_aligned_bytes += h_len[pp[i].q] + h_len[pp[i].r];
        }
    };
    auto window_error = [&](uint32_t e, uint32_t at) {   // (the word is the OR over all processes in the record-sharded mode)
        pgx_set_error("pgx_cluster_greedy: capacity failure in the window at %u%s:%s%s%s%s%s%s", at,
                      exchanging ? " (on this or another process)" : "",
                      e & E_TOUCHED ? " append scratch list full;" : "", e & E_POOL ? " overflow pool of the word index full;" : "",
                      e & E_TABLE ? " exact table of the filter overflowed;" : "",
                      e & E_BAND ? " alignment band wider than 64 diagonals;" : "", e & E_PAIRS ? " candidate pair buffer overflow;" : "",
                      e & E_WORDMULT ? " a word occurs more than 65535 times in one sequence;" : "");
    };
    const bool trace = std::getenv("PGX_TRACE") != nullptr;
    const uint32_t filter_grid = 4096u;
    uint32_t epoch_idx = 0;      // append rounds of the index (line.epoch); 0 = never
    uint32_t epoch_tag = 0;      // discovery rounds (first-open tags, 16 bits)
    PGX_HIP(hipMemsetAsync(d_first.p, 0, (size_t)tag_stride * n_codes * 4, st));
    phase("window set-up");
    const auto t_loop0 = std::chrono::steady_clock::now();
    double t_resolve = 0.0, t_close = 0.0;
    uint64_t n_blocks = 0;
    g_wait_s = 0.0;
    std::function<int()> deferred;  // bookkeeping of the window before, see the close of a window
    std::vector<uint32_t> members;  // cluster -> members numbered so far
    uint64_t pending_words = 0;     // ... and its words
    Chunks chunks;
    // Memory-chunked rule: at a flush position every sequence not yet placed is compared with the index as it is
    // (a SWEEP: windows over the rest of the list that run phase A only; a member with an accepted representative is
    // placed there and then, the others stay in the game), then the index is emptied and clustering goes on.
    Pinned<uint8_t> h_taken;        // per sorted sequence: placed by a sweep (the device reads it through the mapping)
    h_taken.bind(ctx, 14);
    if (!flush_at.empty()) { PGX_HIP(h_taken.reserve(n)); std::fill(h_taken.p, h_taken.p + n, (uint8_t)0); }
    auto drain = [&]() -> int {     // nothing in flight, no bookkeeping pending
        if (deferred) { int rc = deferred(); deferred = nullptr; if (rc) return rc; }
        PGX_HIP(hipStreamSynchronize(st_main));
        PGX_HIP(hipStreamSynchronize(ctx->stream2));
        return PGX_OK;
    };
    size_t next_flush = 0;
    bool sweeping = false;
    uint32_t sweep_at = 0;
    for (uint32_t cursor = 0, nb = 0; cursor < n;) {
        if (!sweeping && next_flush < flush_at.size() && cursor == flush_at[next_flush]) {
            ++next_flush;
            sweeping = true; sweep_at = cursor;
            int rc = drain(); if (rc) return rc;
        }
        const bool sweep = sweeping;
        const uint32_t b0 = sweep ? sweep_at : cursor;
        nb = form_window(b0, chunks, sweep || next_flush >= flush_at.size() ? n : flush_at[next_flush]);   // queries of this window
        // this window's own state and stream (see `overlap`); the names below shadow the first set's
        const int set = overlap ? (int)(S.sweeps & 1u) : 0;
        const hipStream_t st = set ? ctx->stream2 : st_main;
        uint32_t *const dc = dc_set[set];
        unsigned long long *const d_best = d_best_set[set];
        uint8_t *const d_done = d_flags_set[set], *const d_inblk = d_done + window_cap;
        uint8_t *const d_hascand = d_done + 2 * (size_t)window_cap, *const d_accepted = d_done + 3 * (size_t)window_cap;
        Pinned<Pair> &hW_w = set ? hW2 : hW;
        Pinned<unsigned long long> &h_best_w = set ? h_best2 : h_best;
        Pinned<uint32_t> &h_ccnt_w = set ? h_ccnt2 : h_ccnt;
        uint32_t *const d_gs = (set ? d_gscratch2 : d_gscratch).as<uint32_t>();
        if (overlap && S.sweeps > 0)   // the index of the window before is final (its last strike-out is done)
            PGX_HIP(hipStreamWaitEvent(st, ev.strike[set ^ 1], 0));
        bool strike_recorded = false;
        const uint32_t ns = both ? 2 * nb : nb;            // window slots: + one per reverse complement
        const uint32_t n_reps = (uint32_t)rep_seq.size();
        const uint64_t window_words = h_off[b0 + nb] - h_off[b0];
        // (the bookkeeping of the window before may still be pending: its members count as representatives)
        { int rc = ensure_pool(S.rep_words + pending_words + window_words, st); if (rc) return rc; }
        uint32_t *d_poolp = d_pool[pool_cur].as<uint32_t>();
        if (nt) {  // at cd-hit-est's -n 5 -c 0.8 one shared word is enough: size the pair buffer for all pairs
            const uint64_t need = (uint64_t)ns * ((uint64_t)n_reps + nb) + 1024;
            if (need > (400ull << 20)) {
                pgx_set_error("pgx_cluster_greedy: %llu candidate pairs in one window exceed the supported maximum",
                              (unsigned long long)need);
                return PGX_ERR_CAPACITY;
            }
            if (need > pair_cap) {
                PGX_HIP(spin_sync(st));
                if (deferred) { int rc = deferred(); deferred = nullptr; if (rc) return rc; }
                pair_cap = (uint32_t)(need + need / 2);
                PGX_HIP(d_pairsW.alloc((size_t)pair_cap * sizeof(Pair)));
            }
        }
        S.sweeps++;
        const auto t_sweep0 = std::chrono::steady_clock::now();
        const uint64_t blocks_before = n_blocks;
        // the general (int64, one pair per wave) aligner is only needed when some pair of this
        // window cannot use the 16-lane fast path: query length + longest sequence, or the band
        // LDS slot per pair of the 16-lane aligner: sized for two sequences as long as the window's longest query
        const int two_len = 2 * (int)h_len[b0];
        const int a16_slot = two_len <= 1024 - 96 ? 1024 : (two_len <= 1536 - 96 ? 1536 : (two_len <= 2048 - 96 ? 2048 : (two_len <= 3072 - 96 ? 3072 : kA16MaxSlot)));
        Pair *pairsW = (set ? d_pairsW2 : d_pairsW).as<Pair>();
        FilterArgs FA{};
        FA.lines = d_lines.as<IndexLine>(); FA.pool = d_poolp; FA.newbits = d_newbits.as<uint32_t>();
        FA.d_round_lo = dc + C_SEG0; FA.d_round_hi = dc + C_NEW; FA.epoch = 0;
        FA.b0 = b0; FA.nbq = nb; FA.ns = ns; FA.shard_index = shard_index; FA.shard_count = shard_count;
        FA.req_aan = d_aan.as<int32_t>(); FA.best = d_best; FA.done = d_done;
        FA.pairs = pairsW; FA.n_pairs = dc + C_NW; FA.pair_cap = pair_cap;
        FA.visits = d_visits.as<unsigned long long>(); FA.rc_visits = d_rcvis_p; FA.err = dc + C_ERR;
        FA.qlist = nullptr; FA.d_nq = nullptr; FA.mark = nullptr; FA.count_visits = 1u;
        FA.lean = stats == nullptr && !std::getenv("PGX_NO_LEAN") ? 1u : 0u;

        window_init_kernel<<<(window_cap + 255) / 256, 256, 0, st>>>(dc, d_best, both ? d_rcvis_p : nullptr, d_done, window_cap);
        if (!flush_at.empty() && next_flush > 0)
            mark_absent_kernel<<<(nb + 255) / 256, 256, 0, st>>>(h_taken.p + b0, nb, d_done, d_best);
        LAUNCH_CHECK();
        // the host side of a window starts with the bookkeeping of the window before (deferred behind this window's
        // first kernels), then the member states
        auto begin_host_side = [&]() -> int {
            if (deferred) { int rc = deferred(); deferred = nullptr; if (rc) return rc; }
            for (uint32_t q = 0; q < nb; ++q) status[q] = cluster_of[b0 + q] >= 0 ? ST_ABSENT : ST_OPEN;
            return PGX_OK;
        };
        // Can a pair of the evaluation at hand fall outside the 16-lane aligner's slots? Its band cannot when the band
        // width is at most 32; its lengths can when the candidates are OLDER representatives (the pass over the whole
        // index: any length) or when the window's own longest pair exceeds the largest slot. Every other evaluation
        // pairs members of this window with each other -- two sequences no longer than the window's first, which is
        // what the slot was sized for -- and does not launch the general aligner at all (an empty launch is ≈ 11 us of
        // stream time, 105 of them per step on cfg-3s).
        bool older_reps = false;
        // diag + align of a selection of pair records, enqueued on the stream
        auto evaluate = [&](Pair *pairs, const PairSel &sel, unsigned long long *best_arr, uint32_t grid_hint) -> int {
            const uint32_t dg = grid_hint ? std::min(diag_grid, grid_hint) : diag_grid;
            const uint32_t ag = grid_hint ? std::min(align_grid, (grid_hint + 15) / 16) : align_grid;
            {
                ProfScope prof(ctx, "diag_kernel", st);
                // (LDS by the window's longest query: 512 positions give six waves per SIMD, 1024 four, 2048 two and a half)
                auto kern = h_len[b0] > packed_len ? diag_kernel<64, unsigned long long>
                            : (h_len[b0] <= kDiagLdsSmall ? diag_kernel<kDiagLdsSmall, uint32_t>
                               : (h_len[b0] <= 2 * kDiagLdsSmall ? diag_kernel<2 * kDiagLdsSmall, uint32_t> : diag_kernel<kDiagLdsCap, uint32_t>));
                kern<<<dg, 64, 0, st>>>(DS, nullptr, pairs, sel, d_aa1.as<int32_t>(), d_aas.as<int32_t>(),
                                        P->band_width, P->identity, d_gs, gs_stride, gs_cells, dc + C_WIDE, dc + C_ERR);
            }
            LAUNCH_CHECK();
            {
                ProfScope prof(ctx, "align_kernel", st);
                auto kern = a16_slot == 1024 ? align16_kernel<1024> : (a16_slot == 1536 ? align16_kernel<1536> : (a16_slot == 2048 ? align16_kernel<2048>
                            : (a16_slot == 3072 ? align16_kernel<3072> : align16_kernel<kA16MaxSlot>)));
                kern<<<ag, 256, 0, st>>>(DS, nullptr, pairs, sel, d_aa1.as<int32_t>(), P->identity, b0, best_arr, 0u, dc + C_WIDE, 0u);
                // a window whose longest query does not fit the 4 KB slots twice over: the pairs left get 8 KB slots (the
                // general aligner computes a cell per lane every OTHER step, one pair per wave: 1.6 ms for the first
                // window of cfg-3s, whose longest sequences are 2,101 residues)
                const bool big_pass = a16_slot == kA16MaxSlot && two_len > kA16MaxSlot - 96;
                const bool may_be_wide = older_reps || two_len > a16_slot - 96 || P->band_width > 32;
                if (big_pass)
                    align16_kernel<kA16BigSlot><<<ag, 256, 0, st>>>(DS, nullptr, pairs, sel, d_aa1.as<int32_t>(), P->identity, b0, best_arr,
                                                                    0u, dc + C_WIDE, 1u);
                if (may_be_wide)
                    align_kernel<<<grid_hint ? std::min(align_grid, (grid_hint + 3) / 4) : align_grid, 256, 0, st>>>(
                        DS, nullptr, pairs, sel, d_aa1.as<int32_t>(), P->identity, b0, best_arr, 0u, 1, (big_pass ? kA16BigSlot : a16_slot) - 96, dc + C_WIDE);
            }
            LAUNCH_CHECK();
            return PGX_OK;
        };
        // record-sharded mode: every process learns every member's best key (one all-gather per evaluation,
        // enqueued on the stream by the caller's collective library; no host synchronisation)
        auto exchange_best = [&]() -> int {
            if (!exchanging) return PGX_OK;
            ++n_exchanges;
            exchange_prepare_kernel<<<1, 1, 0, st>>>(dc, pair_cap, d_best,
                                                     inject_rank == (int)shard_index && inject_at == n_exchanges ? (uint32_t)E_PAIRS : 0u);
            if (native_exchange) {
                const int rc = pgx_rccl_all_gather_u64(ctx, x_send + (size_t)set * PGX_EXCHANGE_WORDS,
                                                       const_cast<unsigned long long *>(x_recv) + (size_t)set * shard_count * PGX_EXCHANGE_WORDS,
                                                       PGX_EXCHANGE_WORDS, st);
                if (rc) return rc;
            } else if (P->exchange(P->exchange_user, (void *)st, set) != 0) {
                pgx_set_error("pgx_cluster_greedy: the exchange callback failed in the window at %u", b0);
                return PGX_ERR_INTERNAL;
            }
            min_rows_kernel<<<(nb + 255) / 256, 256, 0, st>>>(
                x_recv + (size_t)set * shard_count * PGX_EXCHANGE_WORDS,
                shard_count, PGX_EXCHANGE_WORDS, nb, d_best, dc + C_ERR);
            LAUNCH_CHECK();
            return PGX_OK;
        };
        // the window's pairs found since the last round_begin: evaluated, winners folded into best[]
        auto evaluate_round = [&]() -> int {
            const PairSel sel{dc + C_EVAL0, dc + C_NW, pair_cap, nullptr, 0, nullptr, nullptr, nullptr, b0, 0, 0};
            int rc = evaluate(pairsW, sel, d_best, 0);
            if (rc) return rc;
            return exchange_best();
        };
        // phase A: against the representatives that exist already
        if (b0 > 0) {
            {
                ProfScope prof(ctx, "filter_kernel<all>", st);
                auto kern = nt ? filter_kernel<true, false> : filter_kernel<false, false>;
                kern<<<std::min(filter_grid, (ns + 3) / 4), 256, 0, st>>>(DS, FA);
            }
            LAUNCH_CHECK();
            filter_walk_words += window_words * (both ? 2 : 1) / shard_count;
            older_reps = true;
            int rc = evaluate_round();
            older_reps = false;
            if (rc) return rc;
        }
        // append list[*lo, *hi) to the index as round `epoch_idx` (the bit map of touched codes is the round's)
        auto index_append = [&](const uint32_t *list, const uint32_t *d_lo, const uint32_t *d_hi) -> int {
            ++epoch_idx;
            uint32_t *const map = d_newbits.as<uint32_t>() + (size_t)(epoch_idx & 1u) * bm_words;   // (clear: index_place_kernel of the round before)
            uint32_t *const next_map = d_newbits.as<uint32_t>() + (size_t)((epoch_idx + 1u) & 1u) * bm_words;
            ProfScope prof(ctx, "index_append", st);
            index_append_kernel<<<512, 256, 0, st>>>(DS, list, d_lo, d_hi, d_lines.as<IndexLine>(), epoch_idx,
                                                     map, b0, nb, d_touched.as<Deferred>(),
                                                     dc + C_TOUCH, (uint32_t)max_window_words, dc + C_ERR);
            index_grow_kernel<<<256, 256, 0, st>>>(d_lines.as<IndexLine>(), d_poolp, d_idx.as<uint32_t>(),
                                                   (uint32_t)pool_cap, d_touched.as<Deferred>(), dc + C_TOUCH,
                                                   (uint32_t)max_window_words, dc + C_ERR);
            index_place_kernel<<<256, 256, 0, st>>>(d_lines.as<IndexLine>(), d_poolp, d_touched.as<Deferred>(),
                                                    dc + C_TOUCH, (uint32_t)max_window_words, reinterpret_cast<uint4 *>(next_map),
                                                    bm_words / 4);
            LAUNCH_CHECK();
            return PGX_OK;
        };
        // every window member against the entries round `epoch_idx` added (the representatives list[*lo, *hi))
        auto filter_new_and_evaluate = [&](const uint32_t *d_lo, const uint32_t *d_hi) -> int {
            FilterArgs F = FA;
            F.epoch = epoch_idx; F.d_round_lo = d_lo; F.d_round_hi = d_hi;
            F.newbits = d_newbits.as<uint32_t>() + (size_t)(epoch_idx & 1u) * bm_words;
            {
                ProfScope prof(ctx, "filter_kernel<new>", st);
                auto kern = nt ? filter_kernel<true, true> : filter_kernel<false, true>;
                kern<<<std::min(filter_grid, (ns + 3) / 4), 256, 0, st>>>(DS, F);
            }
            LAUNCH_CHECK();
            filter_walk_words += window_words * (both ? 2 : 1) / shard_count;
            return evaluate_round();
        };
        // Discovery rounds: still-open members that cannot have an earlier open candidate are certain new
        // representatives (first_open_kernel / certain_kernel). They are confirmed at once, appended to the
        // index, and the pass against them assigns most of the remaining members; a second round confirms
        // the members the first round's representatives rejected (the outliers of their families). All on
        // the device: the host learns the outcome with the first block's results.
        uint32_t n_listed = 0;
        if (sweep) { int rc = begin_host_side(); if (rc) return rc; }
        if (!sweep) {   // ---- discovery rounds and blocks (a sweep window has neither: nothing is appended) ----
        if (overlap && S.sweeps > 1)   // from here on the index is written: the window before must have closed
            PGX_HIP(hipStreamWaitEvent(st, ev.post[set ^ 1], 0));
        static const int n_disc = std::getenv("PGX_ROUNDS") ? std::atoi(std::getenv("PGX_ROUNDS")) : kDiscoveryRounds;
        for (int round = 0; round < n_disc; ++round) {
            if (++epoch_tag == 0xFFFFu) { PGX_HIP(hipMemsetAsync(d_first.p, 0, (size_t)tag_stride * n_codes * 4, st)); epoch_tag = 1; }
            const uint32_t which = (uint32_t)round & 1u;
            uint32_t *const d_n_open = dc + (which ? C_ROUND_OPEN2 : C_ROUND_OPEN);
            list_open_kernel<<<(nb + 255) / 256, 256, 0, st>>>(d_best, d_done, b0, nb, d_ulist.as<uint32_t>(), dc, which);
            LAUNCH_CHECK();
            {
                ProfScope prof(ctx, "discover_kernels", st);
                first_open_kernel<<<std::min(2048u, (nb + 3) / 4), 256, 0, st>>>(DS, d_ulist.as<uint32_t>(), d_n_open, b0,
                                                                                 epoch_tag, d_first.as<uint32_t>(), tag_stride,
                                                                                 chunks, d_aan.as<int32_t>(), d_accepted);
                certain_kernel<<<std::min(2048u, (nb + 3) / 4), 256, 0, st>>>(DS, d_ulist.as<uint32_t>(), d_n_open, b0,
                                                                              both ? 1u : 0u, epoch_tag, d_first.as<uint32_t>(), tag_stride,
                                                                              chunks,
                                                                              d_aan.as<int32_t>(), d_accepted, d_done, d_new_list.as<uint32_t>(),
                                                                              dc + C_NEW);
            }
            LAUNCH_CHECK();
            int rc = index_append(d_new_list.as<uint32_t>(), dc + C_SEG0, dc + C_NEW);
            if (rc) return rc;
            rc = filter_new_and_evaluate(dc + C_SEG0, dc + C_NEW);
            if (rc) return rc;
        }
        // Blocks: members still without a representative, <= block_cap at a time, in order. A block is
        // resolved exactly: its members are appended to the index TENTATIVELY, as if all were
        // representatives, so the filter finds the block's internal candidate pairs like any others;
        // they are evaluated, the host walks the block in order, the entries of the members that joined a
        // representative are struck out again, and every other window member is compared with what is
        // left: the block's new representatives. Pair work stays close to what the one-by-one pass does.
        bool first_block = true;
        n_listed = 0;              // discovery representatives the host has seen
        uint32_t n_struck = 0;     // members struck out of the index so far (offsets into the pinned list)
        for (;;) {
            select_block_kernel<<<1, kSelThreads, 0, st>>>(d_best, d_done, d_inblk, b0, nb, block_cap,
                                                           d_blk_list.as<uint32_t>(), dc + C_BLK, dc + C_NK,
                                                           d_hascand, window_cap, dc);  // + has_cand, accepted = 0
            LAUNCH_CHECK();
            {
                int rc = index_append(d_blk_list.as<uint32_t>(), dc + C_ZERO, dc + C_BLK);
                if (rc) return rc;
                FilterArgs F = FA;   // the block's members against the block's (tentative) entries
                F.epoch = epoch_idx; F.d_round_lo = dc + C_ZERO; F.d_round_hi = dc + C_BLK;
                F.newbits = d_newbits.as<uint32_t>() + (size_t)(epoch_idx & 1u) * bm_words;
                F.qlist = d_blk_list.as<uint32_t>(); F.d_nq = dc + C_BLK; F.mark = d_hascand; F.count_visits = 0u;
                F.pairs = d_pairsK.as<Pair>(); F.n_pairs = dc + C_NK; F.pair_cap = pair_cap_k;
                F.shard_count = 1; F.shard_index = 0;   // (replicated on every process)
                ProfScope prof(ctx, "filter_kernel<block>", st);
                auto kern = nt ? filter_kernel<true, true> : filter_kernel<false, true>;
                kern<<<std::min(filter_grid, (block_cap * (both ? 2u : 1u) + 3) / 4), 256, 0, st>>>(DS, F);
            }
            LAUNCH_CHECK();
            // A block member without an earlier in-block candidate (has_cand clear) is certainly a
            // new representative; only pairs against those are evaluated up front. Second round on
            // the device: a member that has candidates but was accepted by none of the certain
            // representatives is LIKELY a representative itself (an outlier of its family), so the
            // pairs against those follow. Whatever the in-order walk on the host still needs
            // afterwards goes through follow-up rounds.
            {
                const PairSel selK{nullptr, dc + C_NK, pair_cap_k, nullptr, 0, nullptr, d_hascand, d_accepted, b0, 0, 2048};   // (a small block: all its pairs at once, the second round then finds nothing left)
                int rc = evaluate(d_pairsK.as<Pair>(), selK, nullptr, 0);
                if (rc) return rc;
                const PairSel selK2{nullptr, dc + C_NK, pair_cap_k, nullptr, 0, d_hascand, d_accepted, nullptr, b0, 1, 0};
                rc = evaluate(d_pairsK.as<Pair>(), selK2, nullptr, 0);
                if (rc) return rc;
            }
            // one round trip: counters, the block list, the in-block pairs (and the representatives the
            // discovery rounds confirmed), written to host memory by one kernel
            {
                PubArgs pa{};
                pa.seg[0] = {dc, h_cnt.p, nullptr, C_COUNT, 1, C_COUNT};
                pa.seg[1] = {d_blk_list.as<uint32_t>(), h_blk.p, dc + C_BLK, 0, 1, block_cap};
                pa.seg[2] = {d_pairsK.as<uint32_t>(), reinterpret_cast<uint32_t *>(hK.p), dc + C_NK, 0, kPairWords, pair_cap_k};
                pa.n = 3;
                if (first_block) pa.seg[pa.n++] = {d_new_list.as<uint32_t>(), h_new.p, dc + C_NEW, 0, 1, window_cap};
                publish_kernel<<<64, 256, 0, st>>>(pa);
                LAUNCH_CHECK();
            }
            if (first_block) {
                // everything up to here was enqueued without looking at results: the previous window's
                // bookkeeping runs now, behind that work, and only then is the member state reset
                int rc = begin_host_side(); if (rc) return rc;
            }
            PGX_HIP(spin_sync(st));
            if (h_cnt.p[C_ERR]) { window_error(h_cnt.p[C_ERR], b0); return PGX_ERR_CAPACITY; }
            if (first_block) {
                n_listed = h_cnt.p[C_NEW];
                for (uint32_t i = 0; i < n_listed; ++i) status[h_new.p[i] - b0] = ST_REP;
                first_block = false;
            }
            const uint32_t n_blk = h_cnt.p[C_BLK], n_open = h_cnt.p[C_OPEN], nK = h_cnt.p[C_NK];
            if (h_cnt.p[C_NW] > pair_cap) {
                pgx_set_error("pgx_cluster_greedy: candidate pair buffer overflow (%u) in the window at %u", h_cnt.p[C_NW], b0);
                return PGX_ERR_CAPACITY;
            }
            if (nK > pair_cap_k) {
                // more in-block candidate pairs than the buffer holds: give the block up (all its tentative
                // entries are struck out again) and come back with a quarter of the members
                if (block_cap <= kBlockCapMin) {
                    pgx_set_error("pgx_cluster_greedy: block pair buffer overflow (%u) in the window at %u", nK, b0);
                    return PGX_ERR_CAPACITY;
                }
                uint32_t *list = h_push.p + n_struck;
                std::copy(h_blk.p, h_blk.p + n_blk, list);
                n_struck += n_blk;
                index_strike_kernel<<<std::min(1024u, (n_blk + 3) / 4), 256, 0, st>>>(DS, list, n_blk, d_lines.as<IndexLine>(), d_poolp);
                retire_block_kernel<<<(nb + 255) / 256, 256, 0, st>>>(d_done, d_inblk, nb, 0u, nullptr);
                LAUNCH_CHECK();
                block_cap = std::max(kBlockCapMin, block_cap / 4);
                continue;
            }
            if (n_blk == 0) {
                if (overlap) { PGX_HIP(hipEventRecord(ev.strike[set], st)); strike_recorded = true; }   // nothing tentative is left
                break;
            }
            // resolve the block in order: first accepted in-block representative by (minc, index)
            const auto t_r0 = std::chrono::steady_clock::now();
            ++n_blocks;
            // bucket the in-block pairs by query (counting sort on the query's rank in the block)
            for (uint32_t t = 0; t < n_blk; ++t) { rank_of[h_blk.p[t] - b0] = t; bucket_k[t] = 0; }
            bucket_k[n_blk] = 0;
            for (uint32_t i = 0; i < nK; ++i) bucket_k[rank_of[real(hK.p[i].q) - b0] + 1]++;
            for (uint32_t t = 0; t < n_blk; ++t) bucket_k[t + 1] += bucket_k[t];   // bucket t = [bucket_k[t], bucket_k[t+1])
            for (uint32_t t = 0; t < n_blk; ++t) fill_k[t] = bucket_k[t];
            order_k.resize(nK);
            for (uint32_t i = 0; i < nK; ++i) order_k[fill_k[rank_of[real(hK.p[i].q) - b0]]++] = i;
            // In-order walk. For member q the sequential rule takes the accepted representative
            // with the smallest (minc, index) among its in-block candidates that ARE
            // representatives. Pairs that neither device round evaluated and that could precede
            // the winner are collected for ALL still-open members in one pass (candidates that
            // are themselves undecided count as possible representatives) and evaluated in a
            // follow-up round; the walk repeats until every member is decided.
            uint32_t t_first = 0;
            for (;;) {
                flight.clear();
                bool any_open = false;
                for (uint32_t t = t_first; t < n_blk; ++t) {
                    const uint32_t k = h_blk.p[t], q = k - b0;
                    if (status[q] != ST_OPEN) continue;
                    const uint32_t lo = bucket_k[t], hi = bucket_k[t + 1];
                    unsigned long long win = kNoBest, open_acc = kNoBest;
                    int32_t win_iden = 0;
                    for (uint32_t e = lo; e < hi; ++e) {
                        const Pair &pr = hK.p[order_k[e]];
                        const uint8_t su = status[pr.r - b0];
                        if (su == ST_MEMBER || !(pr.flags & F_EVAL)) continue;
                        if ((pr.flags & F_TOO_BIG) && (pr.flags & F_DIAG_PASS)) {
                            pgx_set_error("pgx_cluster_greedy: alignment band wider than %d diagonals", kMaxBand);
                            return PGX_ERR_CAPACITY;
                        }
                        if (!(pr.flags & F_ACCEPT)) continue;
                        const unsigned long long key = pair_key(pr);
                        if (su == ST_REP) { if (key < win) { win = key; win_iden = pr.iden; } }
                        else if (key < open_acc) open_acc = key;  // wins if that member turns out a representative
                    }
                    bool needs = false;
                    for (uint32_t e = lo; e < hi; ++e) {
                        const Pair &pr = hK.p[order_k[e]];
                        if (status[pr.r - b0] == ST_MEMBER || (pr.flags & F_EVAL)) continue;
                        if (pair_key(pr) < win) { flight.push_back(order_k[e]); needs = true; }
                    }
                    if (needs || open_acc < win) { any_open = true; continue; }
                    if (win != kNoBest) {
                        status[q] = ST_MEMBER;
                        winner_key[q] = win; iden_of[k] = win_iden; strand_of[k] = (uint8_t)(win >> 63);
                    } else {
                        status[q] = ST_REP;
                    }
                }
                if (!any_open) break;
                while (t_first < n_blk && status[h_blk.p[t_first] - b0] != ST_OPEN) ++t_first;
                if (flight.empty()) { pgx_set_error("pgx_cluster_greedy: block resolution made no progress"); return PGX_ERR_INTERNAL; }
                // follow-up round: evaluate the listed pairs and fetch their records back
                const uint32_t nl = (uint32_t)flight.size();
                ++n_rounds;
                std::copy(flight.begin(), flight.end(), h_list.p);
                PGX_HIP(hipMemcpyAsync(d_list.p, h_list.p, (size_t)nl * 4, hipMemcpyHostToDevice, st));
                {
                    const PairSel selL{nullptr, nullptr, 0, d_list.as<uint32_t>(), nl, nullptr, nullptr, nullptr, b0, 0, 0};
                    int rc = evaluate(d_pairsK.as<Pair>(), selL, nullptr, nl);
                    if (rc) return rc;
                }
                gather_pairs_kernel<<<(nl + 255) / 256, 256, 0, st>>>(d_pairsK.as<Pair>(), d_list.as<uint32_t>(), nl,
                                                                      d_gather.as<Pair>());
                LAUNCH_CHECK();
                PGX_HIP(hipMemcpyAsync(h_gather.p, d_gather.p, (size_t)nl * sizeof(Pair), hipMemcpyDeviceToHost, st));
                PGX_HIP(spin_sync(st));
                for (uint32_t w = 0; w < nl; ++w) hK.p[flight[w]] = h_gather.p[w];
            }
            // block decided: new representatives in order + the candidates the one-by-one pass examines
            new_reps.clear();
            for (uint32_t t = 0; t < n_blk; ++t) {
                const uint32_t k = h_blk.p[t], q = k - b0;
                if (status[q] == ST_REP) new_reps.push_back(k);
                const unsigned long long win = status[q] == ST_MEMBER ? winner_key[q] : kNoBest;
                for (uint32_t e = bucket_k[t]; e < bucket_k[t + 1]; ++e) {
                    const Pair &pr = hK.p[order_k[e]];
                    if (status[pr.r - b0] != ST_REP) continue;
                    if (pair_key(pr) > win) continue;
                    if (!count_replicated) continue;
                    S.filter_pairs++;
                    if ((pr.flags & (F_DIAG_PASS | F_BAND_OK)) == (F_DIAG_PASS | F_BAND_OK)) {
                        S.aligned_pairs++;
                        S.aligned_rep_len += h_len[pr.r];
                        S.dp_cells += (uint64_t)h_len[k] * (uint64_t)(pr.band_right - pr.band_left + 1);
                    }
                }
            }
            if (count_replicated) account(hK.p, nK);
            t_resolve += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_r0).count();
            // the members that joined a representative leave the index again; every window member is then
            // compared with the block's representatives (the block's own members only count their visits)
            {
                uint32_t *list = h_push.p + n_struck;   // read from page-locked host memory; every block has its own range
                uint32_t n_out = 0;
                for (uint32_t t = 0; t < n_blk; ++t)
                    if (status[h_blk.p[t] - b0] != ST_REP) list[n_out++] = h_blk.p[t];
                n_struck += n_out;
                if (n_out)
                    index_strike_kernel<<<std::min(1024u, (n_out + 3) / 4), 256, 0, st>>>(DS, list, n_out, d_lines.as<IndexLine>(), d_poolp);
                retire_block_kernel<<<(nb + 255) / 256, 256, 0, st>>>(d_done, d_inblk, nb, 1u, dc);
                LAUNCH_CHECK();
                if (overlap && n_open == n_blk) {   // the last block: from here on this window only reads the index
                    PGX_HIP(hipEventRecord(ev.strike[set], st));
                    strike_recorded = true;
                }
                int rc = filter_new_and_evaluate(dc + C_ZERO, dc + C_BLK);
                if (rc) return rc;
            }
            if (n_open == n_blk) break;  // that was the last block
        }
        }   // (!sweep)
        if (chunking && !sweep) {   // the discovery chunks of the windows to come (see chunk_frac)
            const uint64_t blk = n_blocks - blocks_before;
            if (blk >= 3) cur_frac = std::max(0.1, cur_frac * 0.5);
            else if (blk <= 1) cur_frac = std::min(chunk_frac, cur_frac * 1.5);
            chunk_words = std::max<uint64_t>(16384, (uint64_t)(cur_frac * thr_frac * (double)n_codes));
        }
        // ---- close the window ---------------------------------------------------------------
        {   // counters, winners, and exactly the pair records that exist, in one launch
            PubArgs pa{};
            pa.seg[0] = {dc, h_ccnt_w.p, nullptr, C_COUNT, 1, C_COUNT};
            pa.seg[1] = {reinterpret_cast<const uint32_t *>(d_best), reinterpret_cast<uint32_t *>(h_best_w.p), nullptr, nb, 2, nb};
            pa.seg[2] = {reinterpret_cast<const uint32_t *>(pairsW), reinterpret_cast<uint32_t *>(hW_w.p), dc + C_NW, 0, kPairWords,
                         (uint32_t)std::min<size_t>(hW_w.cap, pair_cap)};
            pa.n = 3;
            if (both) pa.seg[pa.n++] = {reinterpret_cast<const uint32_t *>(d_rcvis_p), reinterpret_cast<uint32_t *>(h_rcvis.p), nullptr, nb, 2, nb};
            if (overlap && !strike_recorded) PGX_HIP(hipEventRecord(ev.strike[set], st));
            publish_kernel<<<64, 256, 0, st>>>(pa);
            LAUNCH_CHECK();
            if (overlap) PGX_HIP(hipEventRecord(ev.post[set], st));
        }
        // What the close published is looked at by the window's bookkeeping, which waits for it first: with
        // overlapping windows that is when the next window's first kernels have been enqueued, without right here.
        auto closed = [&, b0, st, pairsW, ev_post = ev.post[set], p_cnt = &h_ccnt_w, p_hW = &hW_w](uint32_t &nW) -> int {
            if (overlap) {
                const auto t0 = std::chrono::steady_clock::now();
                hipError_t e;
                while ((e = hipEventQuery(ev_post)) == hipErrorNotReady) {
                }
                g_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                PGX_HIP(e);
            } else {
                PGX_HIP(spin_sync(st));
            }
            nW = p_cnt->p[C_NW];
            if (p_cnt->p[C_ERR]) { window_error(p_cnt->p[C_ERR], b0); return PGX_ERR_CAPACITY; }
            if (nW > pair_cap) {
                pgx_set_error("pgx_cluster_greedy: candidate pair buffer overflow (%u > %u) in the window at %u", nW, pair_cap, b0);
                return PGX_ERR_CAPACITY;
            }
            if (nW > p_hW->cap) {  // the host buffer was too small: grow it and fetch again (the window's stream is idle by now)
                PGX_HIP(p_hW->reserve(nW));  // (reserve keeps nothing: copy the whole range again)
                PGX_HIP(hipMemcpyAsync(p_hW->p, pairsW, (size_t)nW * sizeof(Pair), hipMemcpyDeviceToHost, st));
                PGX_HIP(spin_sync(st));
            }
            return PGX_OK;
        };
        uint32_t nW = 0;
        if (!overlap) { int rc = closed(nW); if (rc) return rc; }
        // The window's bookkeeping -- winners, numbering of the new representatives, identities, the candidates
        // the one-by-one pass would have examined -- works on the host copies only: it is deferred until the
        // next window's first kernels are enqueued (the device starts the next window right away).
        const auto t_c0 = std::chrono::steady_clock::now();
        {
            pending_words = window_words;
            deferred = [&, b0, nb, nW, sweep, closed, p_best = &h_best_w, p_hW = &hW_w]() mutable -> int {
                if (overlap) { int rc = closed(nW); if (rc) return rc; }
                const Pair *pW = p_hW->p;
                pending_words = 0;
                // members that were never in a block: their winner is the 64-bit minimum in best[]
                for (uint32_t q = 0; q < nb; ++q) {
                    if (status[q] != ST_OPEN) continue;
                    const unsigned long long key = p_best->p[q];
                    if (key == kNoBest) {
                        if (sweep) { status[q] = ST_NONE; continue; }   // no representative in the table being dropped
                        pgx_set_error("pgx_cluster_greedy: unresolved member after the last block"); return PGX_ERR_INTERNAL;
                    }
                    status[q] = ST_MEMBER;
                    winner_key[q] = key;
                    strand_of[b0 + q] = (uint8_t)(key >> 63);
                    if (sweep) h_taken.p[b0 + q] = 1;
                }
                if (both)  // reverse-strand word walks happen only for queries the forward strand did not place
                    for (uint32_t q = 0; q < nb; ++q)
                        if (status[q] == ST_REP || status[q] == ST_NONE || (status[q] == ST_MEMBER && (winner_key[q] >> 63))) visits_rc += h_rcvis.p[q];
                // new representatives are numbered in sequence order
                for (uint32_t q = 0; q < nb; ++q)
                    if (status[q] == ST_REP) {
                        cluster_of[b0 + q] = (int32_t)rep_seq.size();
                        rep_seq.push_back(b0 + q);
                        S.sum_len_reps += h_len[b0 + q];
                        S.rep_words += h_wcnt[b0 + q];
                    }
                for (uint32_t q = 0; q < nb; ++q)
                    if (status[q] == ST_MEMBER) cluster_of[b0 + q] = cluster_of[(uint32_t)winner_key[q] & 0x7FFFFFFFu];
                account(pW, nW);
                bool fits = true;
                for (uint32_t i = 0; i < nW; ++i) {
                    const Pair &p = pW[i];
                    const uint32_t k = real(p.q), q = k - b0;
                    if ((p.flags & F_TOO_BIG) && (p.flags & F_DIAG_PASS)) { fits = false; continue; }
                    const unsigned long long key = pair_key(p);
                    // the one-by-one pass examines candidates in key order up to and including the winner
                    if (status[q] == ST_REP || status[q] == ST_NONE || key <= winner_key[q]) {
                        S.filter_pairs++;
                        if ((p.flags & (F_DIAG_PASS | F_BAND_OK)) == (F_DIAG_PASS | F_BAND_OK)) {
                            S.aligned_pairs++;
                            S.aligned_rep_len += h_len[p.r];
                            S.dp_cells += (uint64_t)h_len[k] * (uint64_t)(p.band_right - p.band_left + 1);
                        }
                        if ((p.flags & F_ACCEPT) && status[q] == ST_MEMBER && key == winner_key[q]) iden_of[k] = p.iden;
                    }
                }
                if (!fits) {
                    pgx_set_error("pgx_cluster_greedy: alignment band wider than %d diagonals", kMaxBand);
                    return PGX_ERR_CAPACITY;
                }
                // the window's outputs, in the caller's order; member numbers follow the sorted order (A.3)
                members.resize(rep_seq.size(), 0u);
                for (uint32_t q = 0; q < nb; ++q) {
                    if (status[q] == ST_ABSENT || status[q] == ST_NONE) continue;   // (written when placed / not placed yet)
                    const uint32_t k = b0 + q, o = order[k];
                    const int32_t c = cluster_of[k];
                    out_cluster[o] = c;
                    out_member[o] = (int32_t)members[(size_t)c]++;
                    out_identity[o] = iden_of[k] >= 0 ? (float)iden_of[k] / (float)h_len[k] : 0.f;
                    if (out_strand) out_strand[o] = strand_of[k];   // (set for members only)
                }
                return PGX_OK;
            };
        }
        const bool trace2 = trace && std::getenv("PGX_TRACE")[0] == '2';
        if (nt || trace2) {   // (nucleotide windows size their pair buffer from the number of representatives)
            int rc = deferred(); deferred = nullptr; if (rc) return rc;
        }
        t_close += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_c0).count();
        if (trace2)
            fprintf(stderr, "[pgx] window %4llu b0 %8u nb %6u len %5u..%5u blocks %2llu reps +%5zu (discovery %5u) (total %7zu) pairs %7u  %.2f ms\n",
                    (unsigned long long)S.sweeps, b0, nb, h_len[b0], h_len[b0 + nb - 1],
                    (unsigned long long)(n_blocks - blocks_before), rep_seq.size() - n_reps, n_listed, rep_seq.size(), nW,
                    1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_sweep0).count());
        if (!sweep) { cursor += nb; continue; }
        sweep_at += nb;
        if (sweep_at >= n) {   // the sweep is complete: the index is emptied, clustering goes on with what is left
            int rc = drain(); if (rc) return rc;
            PGX_HIP(hipMemsetAsync(d_lines.p, 0, (size_t)n_codes * sizeof(IndexLine), st_main));
            const uint32_t one = 1u;
            PGX_HIP(hipMemcpyAsync(d_idx.p, &one, 4, hipMemcpyHostToDevice, st_main));
            PGX_HIP(hipStreamSynchronize(st_main));
            sweeping = false;
        }
    }
    if (deferred) { int rc = deferred(); deferred = nullptr; if (rc) return rc; }
    if (trace)
        fprintf(stderr, "[pgx] windows %llu blocks %llu (+%llu follow-up rounds): loop %.1f ms = wait %.1f + block resolve %.1f + window close %.1f + "
                "enqueue/other %.1f\n", (unsigned long long)S.sweeps, (unsigned long long)n_blocks, (unsigned long long)n_rounds,
                1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_loop0).count(), 1e3 * g_wait_s,
                1e3 * t_resolve, 1e3 * t_close,
                1e3 * (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_loop0).count() - g_wait_s -
                       t_resolve - t_close));

#ifdef PGX_FTIME
    {
        unsigned long long h[16], z[16] = {0};
        PGX_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ftime), sizeof h));
        PGX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_ftime), z, sizeof z));
        static const char *nm[12] = {"prologue", "words+probes", "compaction", "last slab", "sync+reduce", "pass2 begin",
                                     "pass2 walk", "pass2 sync", "emit", "bucket clear", "kernel end", "-"};
        double tot = 0;
        for (int i = 0; i < 12; ++i) tot += (double)h[i];
        for (int i = 0; i < 11; ++i) fprintf(stderr, "[ftime] %-14s %6.2f %%\n", nm[i], 100.0 * (double)h[i] / tot);
        fprintf(stderr, "[ftime] %llu waves, %.0f clocks per wave; direct %llu, fallbacks %llu, classic %llu\n", h[12], tot / (double)h[12], h[13], h[14], h[15]);
    }
#endif
    if (overlap) PGX_HIP(hipStreamSynchronize(ctx->stream2));
    unsigned long long visits_table = 0;
    PGX_HIP(hipMemcpyAsync(&visits_table, d_visits.p, 8, hipMemcpyDeviceToHost, st));
    PGX_HIP(hipStreamSynchronize(st));
    S.posting_visits = visits_table + visits_rc;
    S.reserved[0] = gpu_pairs; S.reserved[1] = gpu_aligned; S.reserved[2] = gpu_aligned_bytes;
    S.reserved[3] = filter_walk_words;
    S.n_clusters = rep_seq.size();

    phase("window loop");
    // (the outputs were written window by window, with each window's bookkeeping)
    if (out_n_clusters) *out_n_clusters = (uint32_t)rep_seq.size();
    if (stats) *stats = S;
    return PGX_OK;
}

extern "C" int pgx_cluster_greedy(pgx_ctx *ctx, const uint8_t *residues, const uint64_t *offsets,
                                  uint32_t n_in, const pgx_cluster_params *P, int32_t *out_cluster,
                                  int32_t *out_member, float *out_identity, uint8_t *out_strand,
                                  uint32_t *out_n_clusters, pgx_cluster_stats *stats) {
    PGX_REQUIRE(ctx && P, "NULL argument");
    PGX_REQUIRE(n_in == 0 || (residues && offsets), "NULL sequence arrays");
    PGX_HIP(hipSetDevice(ctx->device_id));
    for (uint32_t i = 0; i < n_in; ++i) PGX_REQUIRE(offsets[i + 1] >= offsets[i], "offsets must be non-decreasing");
    const uint64_t total_in = n_in ? offsets[n_in] : 0;
    DevBuf d_res, d_off;
    d_res.ctx = ctx; d_res.slot = 70;      // the upload buffers live in the context's workspace too
    d_off.ctx = ctx; d_off.slot = 71;
    PGX_HIP(d_res.alloc(total_in + 16));
    PGX_HIP(d_off.alloc(((size_t)n_in + 1) * 8));
    if (n_in) {
        { int rc = pgx_staged_h2d(ctx, d_res.p, residues, total_in, ctx->stream); if (rc) return rc; }   // (pageable: staged by a few threads)
        PGX_HIP(hipMemcpyAsync(d_off.p, offsets, ((size_t)n_in + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    return pgx_cluster_greedy_dev(ctx, d_res.as<uint8_t>(), d_off.as<uint64_t>(), n_in, total_in, P, out_cluster,
                                  out_member, out_identity, out_strand, out_n_clusters, stats, ctx->stream);
}
