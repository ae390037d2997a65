// libpgx host side: multi-threaded FASTA ingest, exact de-duplication and the text outputs around the
// clustering call. No GPU work here; plain C++ behind the same C ABI (include/pgx.h).
//
// What it replaces in the reference (AnnaLew/pangenomix, pure Python, one line at a time):
//   pangenome.py:336-405   consolidate_seqs(): merge genome FASTA files, first-seen exact de-duplication
//                          by sha256 of the joined sequence lines, nr FASTA + redundant/missing header files
//   pangenome.py:425-450   the FASTA -> arrays step of the clustering boundary and cd-hit's .clstr writer
//   pangenome.py:453-560   rename_genes_and_alleles(): <name>_C#A# names, the name table, the renamed nr FASTA
// The reading rules are the reference's own (a record starts at a line whose first character is '>';
// header = first whitespace token minus '>'; sequence lines are stripped and joined; a record without
// sequence is "missing"). Inputs the line-by-line Python semantics treat specially -- carriage returns,
// non-ASCII or unusual control bytes, one header naming two different sequences -- are REPORTED (info.simple = 0), not guessed at: the Python layer then
// takes its own slow path, which mirrors the reference statement by statement. Two oddities ARE handled,
// because the reference's own fixtures hold them: sequence text before the first header and a record whose
// header is empty both count as one record without a name (skipped by consolidate_seqs, :384; reported as
// 'MISSING: ' by build_genetic_feature_tables, :652): group -2.
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <memory>
#include <new>
#include <stdexcept>
#include <thread>
#include <vector>

#include "../../include/pgx.h"

void pgx_set_error(const char *fmt, ...);

namespace {

// ---- SHA-256 (FIPS 180-4), the reference's de-duplication key (pangenome.py:2057-2059) ----------------
const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
void sha256_block_plain(uint32_t st[8], const uint8_t *p) {
    uint32_t w[64];
    for (int i = 0; i < 16; ++i) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
    for (int i = 16; i < 64; ++i) {
        const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
        const uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
    for (int i = 0; i < 64; ++i) {
        const uint32_t t1 = h + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K256[i] + w[i];
        const uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}
#if defined(__x86_64__)
// the same compression function on the SHA extensions (x86 SHA-NI), used when the CPU has them
__attribute__((target("sha,sse4.1,ssse3"))) void sha256_block_ni(uint32_t st[8], const uint8_t *p) {
    const __m128i mask = _mm_set_epi64x(0x0c0d0e0f08090a0bLL, 0x0405060700010203LL);
    __m128i tmp = _mm_loadu_si128((const __m128i *)&st[0]);
    __m128i s1 = _mm_loadu_si128((const __m128i *)&st[4]);
    tmp = _mm_shuffle_epi32(tmp, 0xB1);            // CDAB
    s1 = _mm_shuffle_epi32(s1, 0x1B);              // EFGH
    __m128i s0 = _mm_alignr_epi8(tmp, s1, 8);      // ABEF
    s1 = _mm_blend_epi16(s1, tmp, 0xF0);           // CDGH
    const __m128i save0 = s0, save1 = s1;
    __m128i m[4];
    for (int i = 0; i < 16; ++i) {
        if (i < 4) m[i] = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i *)(p + 16 * i)), mask);
        else {
            __m128i x = _mm_sha256msg1_epu32(m[i & 3], m[(i + 1) & 3]);                 // W[t-16..] + s0(W[t-15..])
            x = _mm_add_epi32(x, _mm_alignr_epi8(m[(i + 3) & 3], m[(i + 2) & 3], 4));   // + W[t-7..]
            m[i & 3] = _mm_sha256msg2_epu32(x, m[(i + 3) & 3]);                         // + s1(W[t-2..])
        }
        __m128i msg = _mm_add_epi32(m[i & 3], _mm_loadu_si128((const __m128i *)&K256[4 * i]));
        s1 = _mm_sha256rnds2_epu32(s1, s0, msg);
        msg = _mm_shuffle_epi32(msg, 0x0E);
        s0 = _mm_sha256rnds2_epu32(s0, s1, msg);
    }
    s0 = _mm_add_epi32(s0, save0);
    s1 = _mm_add_epi32(s1, save1);
    tmp = _mm_shuffle_epi32(s0, 0x1B);             // FEBA
    s1 = _mm_shuffle_epi32(s1, 0xB1);              // DCHG
    s0 = _mm_blend_epi16(tmp, s1, 0xF0);           // DCBA
    s1 = _mm_alignr_epi8(s1, tmp, 8);              // HGFE
    _mm_storeu_si128((__m128i *)&st[0], s0);
    _mm_storeu_si128((__m128i *)&st[4], s1);
}
const bool kHaveShaNi = __builtin_cpu_supports("sha") && __builtin_cpu_supports("sse4.1");
#else
inline void sha256_block_ni(uint32_t *, const uint8_t *) {}
const bool kHaveShaNi = false;
#endif
void sha256(const uint8_t *data, size_t n, uint8_t out[32]) {
    uint32_t st[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    const auto sha256_block = kHaveShaNi ? sha256_block_ni : sha256_block_plain;
    size_t i = 0;
    for (; i + 64 <= n; i += 64) sha256_block(st, data + i);
    uint8_t tail[128] = {0};
    const size_t rem = n - i;
    memcpy(tail, data + i, rem);
    tail[rem] = 0x80;
    const size_t tl = rem + 1 + 8 <= 64 ? 64 : 128;
    const uint64_t bits = (uint64_t)n * 8;
    for (int k = 0; k < 8; ++k) tail[tl - 1 - k] = (uint8_t)(bits >> (8 * k));
    sha256_block(st, tail);
    if (tl == 128) sha256_block(st, tail + 64);
    for (int k = 0; k < 8; ++k) { out[4 * k] = st[k] >> 24; out[4 * k + 1] = st[k] >> 16; out[4 * k + 2] = st[k] >> 8; out[4 * k + 3] = st[k]; }
}

struct Rec {
    uint32_t file;
    uint32_t hdr_len;
    uint64_t hdr_off;      // in the file: header token (after '>')
    uint64_t body_off;     // in the file: the lines after the header line
    uint64_t body_len;     // raw span up to the next record (valid when raw_ok)
    uint64_t seq_off;      // in the file's sequence arena: stripped lines joined
    uint32_t seq_len;
    uint32_t letters;      // [A-Za-z] in the sequence: the length cd-hit's reader keeps
    int32_t body_alt;      // >= 0: index of the normalised body ('\n'.join(stripped lines) + '\n') in the file's alt list
};

struct FileData {
    const char *map = nullptr;
    size_t size = 0;
    std::vector<Rec> recs;
    std::vector<uint8_t> seq;          // sequences of this file's records, concatenated
    std::vector<std::string> alt;      // normalised bodies of the records whose raw text differs
    std::vector<uint8_t> digest;       // 32 bytes per record
    bool simple = true;
    std::string why;
};

inline bool is_blank(char c) { return c == ' ' || c == '\t'; }

void parse_file(FileData &F, uint32_t file_id) {
    const char *p = F.map;
    const size_t n = F.size;
    for (size_t i = 0; i < n; ++i) {
        const unsigned char c = (unsigned char)p[i];
        if (c == '\r' || c >= 0x80 || (c < 0x20 && c != '\n' && c != '\t')) {
            F.simple = false;
            F.why = c == '\r' ? "carriage returns" : "non-ASCII or control bytes";
            return;
        }
    }
    F.seq.reserve(n);
    size_t pos = 0;
    Rec cur{};          // (the lines before the first header form a record without a name)
    cur.file = file_id; cur.body_alt = -1;
    bool raw_ok = true;
    std::string alt;
    auto close_record = [&](size_t end) {
        cur.body_len = end - cur.body_off;
        cur.seq_len = (uint32_t)(F.seq.size() - cur.seq_off);
        if (cur.hdr_len == 0 && cur.seq_len == 0) return;         // nameless and empty: nobody ever sees it
        if (cur.body_len && p[end - 1] != '\n') raw_ok = false;   // last line of the file without a newline
        if (!raw_ok) { cur.body_alt = (int32_t)F.alt.size(); F.alt.push_back(alt); }
        F.recs.push_back(cur);
    };
    while (pos < n) {
        const char *nl = (const char *)memchr(p + pos, '\n', n - pos);
        const size_t eol = nl ? (size_t)(nl - p) : n;      // line = [pos, eol)
        const size_t next = nl ? eol + 1 : n;
        if (p[pos] == '>') {
            close_record(pos);
            size_t e = pos + 1;
            while (e < eol && !is_blank(p[e])) ++e;
            cur = Rec{};
            cur.file = file_id; cur.hdr_off = pos + 1; cur.hdr_len = (uint32_t)(e - pos - 1);
            cur.body_off = next; cur.seq_off = F.seq.size(); cur.body_alt = -1;
            raw_ok = true; alt.clear();
        } else {
            size_t a = pos, b = eol;
            while (a < b && is_blank(p[a])) ++a;
            while (b > a && is_blank(p[b - 1])) --b;
            if (a != pos || b != eol) raw_ok = false;
            alt.append(p + a, b - a); alt.push_back('\n');
            for (size_t i = a; i < b; ++i) cur.letters += (unsigned)((p[i] | 0x20) - 'a') < 26u;
            F.seq.insert(F.seq.end(), (const uint8_t *)p + a, (const uint8_t *)p + b);
        }
        pos = next;
    }
    close_record(n);
    F.digest.resize(F.recs.size() * 32);
    for (size_t r = 0; r < F.recs.size(); ++r)
        if (F.recs[r].seq_len) sha256(F.seq.data() + F.recs[r].seq_off, F.recs[r].seq_len, &F.digest[r * 32]);
}

// Cores this process may really use: the smallest of the hardware's count, the affinity mask and the cgroup's CPU
// quota (containers: 256 hardware threads visible, 16 allowed -- more runnable threads than the quota only get the
// whole group throttled).
unsigned usable_cpus() {
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min<unsigned>(n, (unsigned)std::max(1, CPU_COUNT(&set)));
    auto quota = [&](const char *path, bool v2) {
        FILE *f = fopen(path, "r");
        if (!f) return;
        char a[64] = "", b[64] = "";
        if (v2) {
            if (fscanf(f, "%63s %63s", a, b) == 2 && strcmp(a, "max") != 0) {
                const double q = atof(a), per = atof(b);
                if (q > 0 && per > 0) n = std::min<unsigned>(n, (unsigned)std::max(1.0, q / per + 0.5));
            }
        } else if (fscanf(f, "%63s", a) == 1) {
            const double q = atof(a);
            FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
            double per = 100000;
            if (g) { if (fscanf(g, "%63s", b) == 1 && atof(b) > 0) per = atof(b); fclose(g); }
            if (q > 0) n = std::min<unsigned>(n, (unsigned)std::max(1.0, q / per + 0.5));
        }
        fclose(f);
    };
    quota("/sys/fs/cgroup/cpu.max", true);
    quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", false);
    return n;
}

// An exception in a worker (std::bad_alloc from the vectors and strings the items grow) must not reach
// std::terminate: the workers catch, the first failure stops the hand-out of items, and the CALLING thread throws
// once everybody has joined -- from there the guards of the C ABI (`guarded`) turn it into a status code.
template <typename F>
void parallel_for(size_t n, int threads, F f) {
    std::atomic<size_t> next{0};
    std::atomic<int> failed{0};                 // 1 = out of memory, 2 = anything else
    auto work = [&]() {
        try {
            for (size_t i; !failed.load(std::memory_order_relaxed) && (i = next.fetch_add(1)) < n;) f(i);
        } catch (const std::bad_alloc &) { failed = 1; }
        catch (...) { int none = 0; failed.compare_exchange_strong(none, 2); }
    };
    std::vector<std::thread> pool;
    const int t = std::max(1, std::min<int>(threads, (int)n));
    try {
        for (int k = 1; k < t; ++k) pool.emplace_back(work);
    } catch (...) { failed = 2; }               // (no more threads to be had: what was started finishes the items)
    work();
    for (auto &th : pool) th.join();
    if (failed == 1) throw std::bad_alloc();
    if (failed) throw std::runtime_error("a worker thread failed");
}

// Every entry point of the C ABI that allocates runs inside this guard: no exception crosses the boundary (pgx.h).
template <typename F>
int guarded(const char *fn, F body) {
    try { return body(); }
    catch (const std::bad_alloc &) { pgx_set_error("%s: out of host memory", fn); return PGX_ERR_NOMEM; }
    catch (const std::exception &e) { pgx_set_error("%s: %s", fn, e.what()); return PGX_ERR_INTERNAL; }
    catch (...) { pgx_set_error("%s: unexpected exception", fn); return PGX_ERR_INTERNAL; }
}

struct Out {   // buffered file writer
    FILE *f = nullptr;
    std::vector<char> buf;
    explicit Out(const char *path) { f = fopen(path, "w"); buf.reserve(1 << 22); }
    ~Out() { close(); }
    bool ok() const { return f != nullptr; }
    void put(const char *p, size_t n) {
        if (buf.size() + n > buf.capacity()) flush();
        if (n > buf.capacity()) { fwrite(p, 1, n, f); return; }
        buf.insert(buf.end(), p, p + n);
    }
    void put(const std::string &s) { put(s.data(), s.size()); }
    void put(char c) { put(&c, 1); }
    void flush() { if (f && !buf.empty()) { fwrite(buf.data(), 1, buf.size(), f); buf.clear(); } }
    bool close() { if (!f) return true; flush(); const bool good = fclose(f) == 0; f = nullptr; return good; }
};

// A text file whose items can be formatted independently: the items are cut into chunks, the chunks are
// formatted by all cores into strings, and the strings land in the file with positioned writes.
template <typename F>
bool parallel_write(const char *path, uint64_t n_items, int threads, F format /* (begin, end, std::string &out) */) {
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return false;
    const size_t n_chunks = (size_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)threads * 4, (n_items + 4095) / 4096));
    std::vector<std::string> parts(n_chunks);
    parallel_for(n_chunks, threads, [&](size_t c) {
        format(n_items * c / n_chunks, n_items * (c + 1) / n_chunks, parts[c]);
    });
    std::vector<uint64_t> at(n_chunks + 1, 0);
    for (size_t c = 0; c < n_chunks; ++c) at[c + 1] = at[c] + parts[c].size();
    std::atomic<bool> good{ftruncate(fd, (off_t)at[n_chunks]) == 0};
    // (Landing the chunks through a shared mapping of the file instead -- stores need no inode lock -- was three times
    // slower on the benchmark's 370 MB FASTA: 350 ms against 108 ms, a page fault per 4 KB.)
    parallel_for(n_chunks, threads, [&](size_t c) {
        const char *p = parts[c].data();
        size_t left = parts[c].size();
        uint64_t off = at[c];
        while (left) {
            const ssize_t w = pwrite(fd, p, left, (off_t)off);
            if (w <= 0) { good = false; return; }
            p += w; left -= (size_t)w; off += (uint64_t)w;
        }
    });
    return close(fd) == 0 && good;
}

}  // namespace

struct pgx_fasta_set {
    std::vector<std::string> paths;
    std::vector<FileData> files;
    std::vector<uint64_t> first_rec;          // per file: index of its first record in the global order
    uint64_t n_records = 0, n_missing = 0, n_groups = 0;
    std::vector<int32_t> group_of;            // per record, -1 = no sequence
    std::vector<uint32_t> file_of;            // per record
    std::vector<uint64_t> rep_of_group;       // first-seen record of the group
    std::vector<uint64_t> hdr_off;            // header blob offsets, n_records + 1
    std::string hdr_blob;
    std::unique_ptr<uint8_t[]> residues;      // groups' sequences, first-seen order (offsets[n_groups] bytes + 16 of
                                              // padding; allocated uninitialised: the threads that fill it fault its pages in)
    std::vector<uint64_t> offsets;            // n_groups + 1
    std::vector<uint32_t> letters;            // per group
    std::vector<uint8_t> digests;             // per group, 32 bytes
    std::vector<uint64_t> members_off;        // per group: its records, encounter order (CSR)
    std::vector<uint64_t> members;
    bool simple = true;
    std::string why;
    int threads = 1;
    const Rec &rec(uint64_t r) const {
        const size_t f = file_of[r];
        return files[f].recs[r - first_rec[f]];
    }
    ~pgx_fasta_set() {
        for (auto &F : files) if (F.map && F.size) munmap((void *)F.map, F.size);
    }
};

extern "C" {

static int pgx_fasta_open_impl(const char *const *paths, uint32_t n_paths, int n_threads, pgx_fasta_set **out) {
    if (!out || (n_paths && !paths)) { pgx_set_error("pgx_fasta_open: NULL argument"); return PGX_ERR_INVALID; }
    *out = nullptr;
    std::unique_ptr<pgx_fasta_set> holder(new pgx_fasta_set());   // (released to the caller at the end; freed on every other way out)
    pgx_fasta_set *S = holder.get();
    // (the writers format in memory: one thread per usable core; the readers here wait for files and fresh pages, and
    // twice as many keep the cores busy: 155 ms against 200 ms for the benchmark's 400 files on a 16-core quota)
    const bool automatic = n_threads <= 0;
    if (automatic) n_threads = (int)std::max(1u, std::min(32u, usable_cpus()));
    S->threads = n_threads;
    if (automatic) n_threads = std::min(32, 2 * n_threads);
    S->files.resize(n_paths);
    for (uint32_t i = 0; i < n_paths; ++i) S->paths.emplace_back(paths[i]);
    const bool trace = std::getenv("PGX_TRACE") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[pgx] ingest: %-28s %8.2f ms\n", what, 1e3 * std::chrono::duration<double>(now - t_prev).count());
        t_prev = now;
    };
    std::atomic<int> failed{-1};
    parallel_for(n_paths, n_threads, [&](size_t i) {
        FileData &F = S->files[i];
        const int fd = open(S->paths[i].c_str(), O_RDONLY);
        struct stat sb;
        if (fd < 0 || fstat(fd, &sb) != 0) { failed = (int)i; if (fd >= 0) close(fd); return; }
        F.size = (size_t)sb.st_size;
        if (F.size) {
            void *m = mmap(nullptr, F.size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { failed = (int)i; F.size = 0; close(fd); return; }
            F.map = (const char *)m;
            madvise(m, F.size, MADV_SEQUENTIAL);
        }
        close(fd);
        parse_file(F, (uint32_t)i);
    });
    if (failed >= 0) {
        pgx_set_error("pgx_fasta_open: cannot read %s", S->paths[(size_t)failed].c_str());
        return PGX_ERR_INVALID;
    }
    lap("read + parse + sha256");
    S->first_rec.resize(n_paths + 1, 0);
    for (uint32_t i = 0; i < n_paths; ++i) {
        if (!S->files[i].simple && S->simple) { S->simple = false; S->why = S->paths[i] + ": " + S->files[i].why; }
        S->first_rec[i + 1] = S->first_rec[i] + S->files[i].recs.size();
    }
    S->n_records = S->first_rec[n_paths];
    if (!S->simple) { *out = holder.release(); return PGX_OK; }   // the caller takes its own path; nothing else is built
    const uint64_t R = S->n_records;
    S->group_of.assign(R, -1);
    S->file_of.resize(R);
    S->hdr_off.resize(R + 1);
    uint64_t hb = 0;
    for (uint32_t i = 0; i < n_paths; ++i)
        for (size_t r = 0; r < S->files[i].recs.size(); ++r) {
            const uint64_t g = S->first_rec[i] + r;
            S->file_of[g] = i;
            S->hdr_off[g] = hb;
            hb += S->files[i].recs[r].hdr_len;
        }
    S->hdr_off[R] = hb;
    S->hdr_blob.resize(hb);
    parallel_for(n_paths, n_threads, [&](size_t i) {
        const FileData &F = S->files[i];
        for (size_t r = 0; r < F.recs.size(); ++r)
            memcpy(&S->hdr_blob[S->hdr_off[S->first_rec[i] + r]], F.map + F.recs[r].hdr_off, F.recs[r].hdr_len);
    });
    lap("headers");
    // first-seen exact de-duplication, in the order given (files, then records): open addressing on the
    // digest's first 8 bytes, full 32-byte compare
    // Partitioned by digest: thread p owns the records whose digest falls into its share of the hash space and
    // finds, for each of them in record order, the first record with the same digest (its own table: no
    // sharing); the groups are then numbered in one sequential pass over the records, which needs no hashing.
    const unsigned n_part = (unsigned)std::max(1, std::min(n_threads, 16));
    auto digest_of = [&](uint64_t g) -> const uint8_t * {
        const Rec &rc = S->rec(g);
        return &S->files[rc.file].digest[(g - S->first_rec[rc.file]) * 32];
    };
    std::vector<uint64_t> first_same(R);   // record -> first record with the same sequence
    parallel_for(n_part, (int)n_part, [&](size_t part) {
        size_t cap = 16;
        while (cap < 2 * ((size_t)R / n_part + 1) + 16) cap <<= 1;
        std::vector<int64_t> table(cap, -1);   // -> record
        for (uint64_t g = 0; g < R; ++g) {
            const Rec &rc = S->rec(g);
            if (!rc.hdr_len || !rc.seq_len) continue;
            const uint8_t *d = digest_of(g);
            uint64_t h;
            memcpy(&h, d, 8);
            h *= 0x9E3779B97F4A7C15ull;
            if ((unsigned)((h >> 40) % n_part) != part) continue;
            size_t slot = (size_t)h & (cap - 1);
            for (size_t probes = 0;; ++probes) {
                const int64_t t = table[slot];
                if (t < 0) { table[slot] = (int64_t)g; first_same[g] = g; break; }
                if (memcmp(digest_of((uint64_t)t), d, 32) == 0) { first_same[g] = (uint64_t)t; break; }
                slot = (slot + 1) & (cap - 1);
                if (probes > cap / 2) {   // (a skewed partition: grow and re-insert)
                    std::vector<int64_t> bigger(cap * 2, -1);
                    for (int64_t e : table) if (e >= 0) {
                        uint64_t hh; memcpy(&hh, digest_of((uint64_t)e), 8); hh *= 0x9E3779B97F4A7C15ull;
                        size_t sl = (size_t)hh & (cap * 2 - 1);
                        while (bigger[sl] >= 0) sl = (sl + 1) & (cap * 2 - 1);
                        bigger[sl] = e;
                    }
                    table.swap(bigger); cap *= 2; slot = (size_t)h & (cap - 1); probes = 0;
                }
            }
        }
    });
    for (uint64_t g = 0; g < R; ++g) {
        const Rec &rc = S->rec(g);
        if (!rc.hdr_len) { S->group_of[g] = -2; continue; }      // a sequence without a name
        if (!rc.seq_len) { ++S->n_missing; continue; }
        if (first_same[g] == g) {
            S->group_of[g] = (int32_t)S->rep_of_group.size();
            S->rep_of_group.push_back(g);
        } else {
            S->group_of[g] = S->group_of[first_same[g]];
        }
    }
    S->n_groups = S->rep_of_group.size();
    const uint64_t G = S->n_groups;
    S->digests.resize(G * 32);
    parallel_for((size_t)((G + 4095) / 4096), n_threads, [&](size_t c) {
        for (uint64_t k = c * 4096; k < std::min<uint64_t>(G, (c + 1) * 4096); ++k)
            memcpy(&S->digests[k * 32], digest_of(S->rep_of_group[k]), 32);
    });
    lap("de-duplication");
    // the groups' members in encounter order (CSR) and the sequences handed to the clustering call
    S->members_off.assign(G + 1, 0);
    for (uint64_t g = 0; g < R; ++g) if (S->group_of[g] >= 0) S->members_off[(size_t)S->group_of[g] + 1]++;
    for (uint64_t k = 0; k < G; ++k) S->members_off[k + 1] += S->members_off[k];
    S->members.resize(S->members_off[G]);
    {
        std::vector<uint64_t> fill(S->members_off.begin(), S->members_off.end() - 1);
        for (uint64_t g = 0; g < R; ++g) if (S->group_of[g] >= 0) S->members[fill[(size_t)S->group_of[g]]++] = g;
    }
    S->offsets.resize(G + 1);
    S->letters.resize(G);
    uint64_t tot = 0;
    for (uint64_t k = 0; k < G; ++k) {
        const Rec &rc = S->rec(S->rep_of_group[k]);
        S->offsets[k] = tot; tot += rc.seq_len; S->letters[k] = rc.letters;
    }
    S->offsets[G] = tot;
    S->residues.reset(new uint8_t[tot + 16]);
    memset(S->residues.get() + tot, 0, 16);
    parallel_for((size_t)((G + 4095) / 4096), n_threads, [&](size_t c) {
        for (uint64_t k = c * 4096; k < std::min<uint64_t>(G, (c + 1) * 4096); ++k) {
            const Rec &rc = S->rec(S->rep_of_group[k]);
            memcpy(&S->residues[S->offsets[k]], S->files[rc.file].seq.data() + rc.seq_off, rc.seq_len);
        }
    });
    lap("groups + sequences");
    // one header, one group: the reference maps records to alleles through a dictionary keyed by the header
    // string (pangenome.py:505-521, :649); a header that names two different sequences is resolved there by
    // insertion order. Such inputs are left to the Python path.
    {
        std::vector<uint64_t> hhash(R);
        parallel_for((size_t)((R + 65535) / 65536), n_threads, [&](size_t c) {
            for (uint64_t g = c * 65536; g < std::min<uint64_t>(R, (c + 1) * 65536); ++g) {
                const char *h = &S->hdr_blob[S->hdr_off[g]];
                const size_t hl = (size_t)(S->hdr_off[g + 1] - S->hdr_off[g]);
                uint64_t x = 1469598103934665603ull;
                for (size_t i = 0; i < hl; ++i) x = (x ^ (unsigned char)h[i]) * 1099511628211ull;
                hhash[g] = x * 0x9E3779B97F4A7C15ull;
            }
        });
        std::vector<int64_t> clash(n_part, -1);   // per partition: the first record whose header names another sequence too
        parallel_for(n_part, (int)n_part, [&](size_t part) {
            size_t hc = 16;
            while (hc < 4 * ((size_t)R / n_part + 1) + 16) hc <<= 1;
            std::vector<int64_t> ht(hc, -1);   // -> a record with that header
            for (uint64_t g = 0; g < R; ++g) {
                if (S->group_of[g] < 0 || (unsigned)((hhash[g] >> 40) % n_part) != part) continue;
                const char *h = &S->hdr_blob[S->hdr_off[g]];
                const size_t hl = (size_t)(S->hdr_off[g + 1] - S->hdr_off[g]);
                size_t slot = (size_t)hhash[g] & (hc - 1);
                for (size_t probes = 0;; ++probes) {
                    const int64_t t = ht[slot];
                    if (t < 0) { ht[slot] = (int64_t)g; break; }
                    const size_t tl = (size_t)(S->hdr_off[t + 1] - S->hdr_off[t]);
                    if (tl == hl && memcmp(&S->hdr_blob[S->hdr_off[t]], h, hl) == 0) {
                        if (S->group_of[t] != S->group_of[g] && clash[part] < 0) clash[part] = (int64_t)g;
                        break;
                    }
                    slot = (slot + 1) & (hc - 1);
                    if (probes > hc / 2) { clash[part] = (int64_t)g; return; }   // (hopelessly skewed: leave it to the Python path)
                }
                if (clash[part] >= 0) return;
            }
        });
        int64_t first_clash = -1;
        for (int64_t c : clash) if (c >= 0 && (first_clash < 0 || c < first_clash)) first_clash = c;
        if (first_clash >= 0) {
            const char *h = &S->hdr_blob[S->hdr_off[first_clash]];
            S->simple = false;
            S->why = "header '" + std::string(h, (size_t)(S->hdr_off[first_clash + 1] - S->hdr_off[first_clash])) + "' names two different sequences";
        }
    }
    lap("header check");
    *out = holder.release();
    return PGX_OK;
}

void pgx_fasta_close(pgx_fasta_set *S) { delete S; }

int pgx_fasta_info(const pgx_fasta_set *S, pgx_fasta_info_t *out) {
    if (!S || !out) { pgx_set_error("pgx_fasta_info: NULL argument"); return PGX_ERR_INVALID; }
    memset(out, 0, sizeof(*out));
    out->n_records = S->n_records; out->n_missing = S->n_missing; out->n_groups = S->n_groups;
    out->n_residue_bytes = S->offsets.empty() ? 0 : S->offsets.back();
    out->n_header_bytes = S->hdr_blob.size();
    out->simple = S->simple ? 1 : 0;
    snprintf(out->why, sizeof(out->why), "%s", S->why.c_str());
    return PGX_OK;
}

const int32_t *pgx_fasta_group_of_record(const pgx_fasta_set *S) { return S && S->simple ? S->group_of.data() : nullptr; }
const uint32_t *pgx_fasta_file_of_record(const pgx_fasta_set *S) { return S && S->simple ? S->file_of.data() : nullptr; }
const uint64_t *pgx_fasta_rep_of_group(const pgx_fasta_set *S) { return S && S->simple ? S->rep_of_group.data() : nullptr; }
const uint8_t *pgx_fasta_residues(const pgx_fasta_set *S) { return S && S->simple ? S->residues.get() : nullptr; }
const uint64_t *pgx_fasta_offsets(const pgx_fasta_set *S) { return S && S->simple ? S->offsets.data() : nullptr; }
const uint32_t *pgx_fasta_letters(const pgx_fasta_set *S) { return S && S->simple ? S->letters.data() : nullptr; }
const uint8_t *pgx_fasta_digests(const pgx_fasta_set *S) { return S && S->simple ? S->digests.data() : nullptr; }
const char *pgx_fasta_header_blob(const pgx_fasta_set *S) { return S && S->simple ? S->hdr_blob.data() : nullptr; }
const uint64_t *pgx_fasta_header_offsets(const pgx_fasta_set *S) { return S && S->simple ? S->hdr_off.data() : nullptr; }

static void add_body(std::string &o, const pgx_fasta_set *S, const Rec &rc) {
    const FileData &F = S->files[rc.file];
    if (rc.body_alt >= 0) o += F.alt[(size_t)rc.body_alt];
    else o.append(F.map + rc.body_off, rc.body_len);
}
static void add_header(std::string &o, const pgx_fasta_set *S, uint64_t r) {
    o.append(&S->hdr_blob[S->hdr_off[r]], (size_t)(S->hdr_off[r + 1] - S->hdr_off[r]));
}
static void put_header(Out &o, const pgx_fasta_set *S, uint64_t r) {
    o.put(&S->hdr_blob[S->hdr_off[r]], (size_t)(S->hdr_off[r + 1] - S->hdr_off[r]));
}

// consolidate_seqs()'s three files (pangenome.py:374-403): the non-redundant FASTA (first-seen records, the
// stripped sequence lines as they were wrapped), the groups with more than one header (encounter order,
// tab-separated), the headers without sequence.
static int pgx_fasta_write_consolidated_impl(const pgx_fasta_set *S, const char *nr_path, const char *shared_path,
                                 const char *missing_path) {
    if (!S || !S->simple || !shared_path) { pgx_set_error("pgx_fasta_write_consolidated: invalid argument"); return PGX_ERR_INVALID; }
    if (nr_path) {
        const bool ok = parallel_write(nr_path, S->n_groups, S->threads, [&](uint64_t b, uint64_t e, std::string &o) {
            uint64_t bytes = 0;
            for (uint64_t k = b; k < e; ++k) bytes += S->rec(S->rep_of_group[k]).body_len + 64;
            o.reserve(bytes);
            for (uint64_t k = b; k < e; ++k) {
                const uint64_t r = S->rep_of_group[k];
                o.push_back('>'); add_header(o, S, r); o.push_back('\n');
                add_body(o, S, S->rec(r));
            }
        });
        if (!ok) { pgx_set_error("cannot write %s", nr_path); return PGX_ERR_INVALID; }
    }
    {
        Out o(shared_path);
        if (!o.ok()) { pgx_set_error("cannot write %s", shared_path); return PGX_ERR_INVALID; }
        for (uint64_t k = 0; k < S->n_groups; ++k) {
            const uint64_t a = S->members_off[k], b = S->members_off[k + 1];
            if (b - a < 2) continue;
            for (uint64_t i = a; i < b; ++i) { if (i > a) o.put('\t'); put_header(o, S, S->members[i]); }
            o.put('\n');
        }
        if (!o.close()) { pgx_set_error("write to %s failed", shared_path); return PGX_ERR_INVALID; }
    }
    if (missing_path) {
        Out o(missing_path);
        if (!o.ok()) { pgx_set_error("cannot write %s", missing_path); return PGX_ERR_INVALID; }
        for (uint64_t r = 0; r < S->n_records; ++r)
            if (S->group_of[r] == -1) { put_header(o, S, r); o.put('\n'); }
        if (!o.close()) { pgx_set_error("write to %s failed", missing_path); return PGX_ERR_INVALID; }
    }
    return PGX_OK;
}

// What follows the clustering call, as text (pangenome.py:444-450 -> cd-hit's .clstr; :505-544): for the
// groups' sequences in first-seen order, cluster[k] / member[k] / identity[k] / strand[k] as pgx_cluster_greedy
// returns them.
//   clstr_path   cd-hit's grammar: clusters in creation order, members in member order
//   names_path   <prefix><cluster><variant><member> \t header \t synonyms...   in .clstr order
//   nr_out_path  the non-redundant FASTA again with the allele names as headers; unclustered records dropped
// NULL paths are skipped.
static int pgx_fasta_write_clustered_impl(const pgx_fasta_set *S, const int32_t *cluster, const int32_t *member,
                              const float *identity, const uint8_t *strand, int nucleotide, const char *prefix,
                              const char *variant, const char *clstr_path, const char *names_path,
                              const char *nr_out_path) {
    if (!S || !S->simple || !cluster || !member || !identity || !prefix || !variant) {
        pgx_set_error("pgx_fasta_write_clustered: invalid argument");
        return PGX_ERR_INVALID;
    }
    const uint64_t G = S->n_groups;
    const bool trace = std::getenv("PGX_TRACE") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[pgx] outputs: %-27s %8.2f ms\n", what, 1e3 * std::chrono::duration<double>(now - t_prev).count());
        t_prev = now;
    };
    // the clustered groups by (cluster, member number, group): a clustering numbers the members of a cluster 0, 1, 2 ...
    // so every group has its place at once (start of its cluster + member number); anything else is sorted
    std::vector<uint64_t> order;
    {
        int32_t cmax = -1;
        uint64_t n_in = 0;
        for (uint64_t k = 0; k < G; ++k) if (cluster[k] >= 0) { ++n_in; cmax = std::max(cmax, cluster[k]); }
        bool placed = cmax >= 0 && (uint64_t)cmax < 4 * n_in + 1024;
        if (placed) {
            std::vector<uint64_t> start((size_t)cmax + 2, 0);
            for (uint64_t k = 0; k < G; ++k) if (cluster[k] >= 0) ++start[(size_t)cluster[k] + 1];
            for (size_t c = 0; c <= (size_t)cmax; ++c) start[c + 1] += start[c];
            constexpr uint64_t kFree = ~0ull;
            order.assign(n_in, kFree);
            for (uint64_t k = 0; k < G && placed; ++k) {
                if (cluster[k] < 0) continue;
                const uint64_t lo = start[(size_t)cluster[k]], size = start[(size_t)cluster[k] + 1] - lo;
                if (member[k] < 0 || (uint64_t)member[k] >= size || order[lo + (uint64_t)member[k]] != kFree) placed = false;
                else order[lo + (uint64_t)member[k]] = k;
            }
        }
        if (!placed) {
            order.clear();
            order.reserve(n_in);
            for (uint64_t k = 0; k < G; ++k) if (cluster[k] >= 0) order.push_back(k);
            std::sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) {
                return cluster[a] != cluster[b] ? cluster[a] < cluster[b] : (member[a] != member[b] ? member[a] < member[b] : a < b);
            });
        }
    }
    lap("order");
    const int T = S->threads;
    auto name_of = [&](std::string &o, uint64_t k) {
        char num[64];
        o.append(prefix);
        o.append(num, (size_t)snprintf(num, sizeof num, "%d%s%d", cluster[k], variant, member[k]));
    };
    // the three files are independent of one another and their items of each other: every file is formatted
    // in chunks by all cores and written with positioned writes
    // ... and the files beside one another (each has serial stretches -- open, truncate, the chunk table, close)
    bool ok_file[3] = {true, true, true};
    parallel_for(3, 3, [&](size_t which) {
    if (which == 0 && clstr_path) {
        const bool ok = parallel_write(clstr_path, order.size(), T, [&](uint64_t b, uint64_t e, std::string &o) {
            char num[96];
            o.reserve((e - b) * 64);
            for (uint64_t i = b; i < e; ++i) {
                const uint64_t k = order[i];
                if (i == 0 || cluster[order[i - 1]] != cluster[k]) o.append(num, (size_t)snprintf(num, sizeof num, ">Cluster %d\n", cluster[k]));
                o.append(num, (size_t)snprintf(num, sizeof num, "%d\t%u%s, >", member[k], S->letters[k], nucleotide ? "nt" : "aa"));
                add_header(o, S, S->rep_of_group[k]);
                if (member[k] == 0) o.append("... *\n", 6);
                else {
                    const float pct = identity[k] * 100.0f;
                    if (nucleotide) o.append(num, (size_t)snprintf(num, sizeof num, "... at %c/%.2f%%\n", strand && strand[k] ? '-' : '+', (double)pct));
                    else o.append(num, (size_t)snprintf(num, sizeof num, "... at %.2f%%\n", (double)pct));
                }
            }
        });
        ok_file[0] = ok;
    }
    if (which == 1 && names_path) {
        const bool ok = parallel_write(names_path, order.size(), T, [&](uint64_t b, uint64_t e, std::string &o) {
            o.reserve((e - b) * 64);
            for (uint64_t i = b; i < e; ++i) {
                const uint64_t k = order[i];
                name_of(o, k);
                for (uint64_t m = S->members_off[k]; m < S->members_off[k + 1]; ++m) { o.push_back('\t'); add_header(o, S, S->members[m]); }
                o.push_back('\n');
            }
        });
        ok_file[1] = ok;
    }
    if (which == 2 && nr_out_path) {
        const bool ok = parallel_write(nr_out_path, G, T, [&](uint64_t b, uint64_t e, std::string &o) {
            uint64_t bytes = 0;
            for (uint64_t k = b; k < e; ++k) if (cluster[k] >= 0) bytes += S->rec(S->rep_of_group[k]).body_len + 40;
            o.reserve(bytes);
            for (uint64_t k = b; k < e; ++k) {
                if (cluster[k] < 0) continue;
                o.push_back('>'); name_of(o, k); o.push_back('\n');
                add_body(o, S, S->rec(S->rep_of_group[k]));
            }
        });
        ok_file[2] = ok;
    }
    });
    lap(".clstr, allele names, nr FASTA");
    const char *const file_path[3] = {clstr_path, names_path, nr_out_path};
    for (int f = 0; f < 3; ++f) if (!ok_file[f]) { pgx_set_error("cannot write %s", file_path[f]); return PGX_ERR_INVALID; }
    return PGX_OK;
}

// The permutations estimate_pan_core_size() consumes (pangenome_analysis.py:84-85): per iteration
// `p = np.arange(S); np.random.shuffle(p)` on numpy's GLOBAL legacy generator. To keep a seeded run identical
// to the reference's, the same stream has to be consumed in the same way: MT19937 (the legacy RandomState's
// generator), numpy's random_interval (smallest bit mask >= i, 32-bit draws, rejection) and the
// Fisher-Yates order of RandomState.shuffle (i = n-1 .. 1, swap x[i] <-> x[j]). The caller passes the
// generator's state (np.random.get_state(): key[624], pos) and stores the advanced state back.
// The loop runs over the DRAWS, not over the positions: every draw is masked, compared and applied as a swap
// that is a no-op when the draw is rejected (x[i] <-> x[i]), and the position only advances on acceptance -- no
// unpredictable branch per draw (the rejection loop mispredicts on a third of the draws). The generator's 624
// outputs of a block are tempered in one vectorisable pass.
static void mt_block(uint32_t *__restrict__ mt, uint32_t *__restrict__ out) {
    // (plain counted loops over disjoint ranges, written so that the compiler vectorises them: an element depends on
    // elements 1 and 397 places ahead or 227 behind, never on a neighbour computed in the same vector step)
    constexpr uint32_t kUp = 0x80000000u, kLo = 0x7fffffffu, kA = 0x9908b0dfu;
    for (int k = 0; k < 227; ++k) {
        const uint32_t y = (mt[k] & kUp) | (mt[k + 1] & kLo);
        mt[k] = mt[k + 397] ^ (y >> 1) ^ ((0u - (y & 1u)) & kA);
    }
    for (int k = 227; k < 454; ++k) {          // (two stretches of 227 and 169: each reads what the stretch before wrote)
        const uint32_t y = (mt[k] & kUp) | (mt[k + 1] & kLo);
        mt[k] = mt[k - 227] ^ (y >> 1) ^ ((0u - (y & 1u)) & kA);
    }
    for (int k = 454; k < 623; ++k) {
        const uint32_t y = (mt[k] & kUp) | (mt[k + 1] & kLo);
        mt[k] = mt[k - 227] ^ (y >> 1) ^ ((0u - (y & 1u)) & kA);
    }
    {
        const uint32_t y = (mt[623] & kUp) | (mt[0] & kLo);
        mt[623] = mt[396] ^ (y >> 1) ^ ((0u - (y & 1u)) & kA);
    }
    for (int k = 0; k < 624; ++k) {
        uint32_t y = mt[k];
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        out[k] = y;
    }
}
// Two phases, because the position of an iteration's first draw depends on every rejection before it but the
// PERMUTATIONS do not depend on each other: (1) one serial walk over the stream that only counts -- per draw a mask,
// a compare and a decrement, no memory traffic beyond the draws themselves -- and notes where each iteration starts;
// (2) the Fisher-Yates swaps of the iterations, each from its own start, on all cores. [One serial loop doing both:
// 3.0 ms per 1000 x 400 on the benchmark host, the critical path of estimate_pan_core_size() beside the upload.]
static int pgx_legacy_shuffles_impl(uint32_t *key, int32_t *pos, uint32_t n, uint32_t n_iter, int32_t *out_perms) {
    if (!key || !pos || (n && n_iter && !out_perms) || *pos < 0 || *pos > 624) {
        pgx_set_error("pgx_legacy_shuffles: invalid argument");
        return PGX_ERR_INVALID;
    }
    if (n < 2 || n_iter == 0) {      // nothing is drawn
        for (uint32_t it = 0; it < n_iter; ++it) for (uint32_t i = 0; i < n; ++i) out_perms[(size_t)it * n + i] = (int32_t)i;
        return PGX_OK;
    }
    // The stream of tempered outputs from the current position on, generated block by block as the count needs them.
    // Never re-allocated while the workers read it: room for four draws per position (every draw is accepted with
    // probability > 1/2, so a seeded generator cannot come near that; running into it is reported, not survived).
    const bool trace_ = std::getenv("PGX_TRACE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto lap_ = [&](const char *what) { if (trace_) fprintf(stderr, "[pgx] shuffles: %-18s at %7.3f ms\n", what, 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count()); };
    const size_t cap = 4 * (size_t)n_iter * n + 2 * 624;
    std::unique_ptr<uint32_t[]> raw_mem(new uint32_t[cap]);
    uint32_t *raw = raw_mem.get();
    size_t raw_n = 624 - (size_t)*pos;
    for (int k = *pos; k < 624; ++k) {     // what is left of the current block
        uint32_t y = key[k];
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        raw[(size_t)(k - *pos)] = y;
    }
    size_t last_block_at = 0;              // where the block generated last starts in `raw`
    bool fresh = false;                    // a new block has been generated
    std::vector<uint64_t> start((size_t)n_iter + 1, 0);
    bool overflow = false;
    size_t cur = 0;
    for (uint32_t it = 0; it < n_iter && !overflow; ++it) {
        start[it] = cur;
        uint32_t i = n - 1;
        while (i >= 1) {
            if (cur == raw_n) {
                if (raw_n + 624 > cap) { overflow = true; break; }
                mt_block(key, raw + raw_n);
                last_block_at = raw_n; fresh = true;
                raw_n += 624;
            }
            // draws of the stretch where the mask (smallest 2^k - 1 >= i) stays the same: i down to the next power of two
            const uint32_t mask = 0xFFFFFFFFu >> __builtin_clz(i);
            const uint32_t floor_i = mask >> 1;
            size_t k = cur;
            for (; k < raw_n && i > floor_i; ++k) i -= (raw[k] & mask) <= i;
            cur = k;
        }
    }
    if (overflow) { pgx_set_error("pgx_legacy_shuffles: generator stream longer than four draws per position"); return PGX_ERR_INTERNAL; }
    lap_("count done");
    // (2) the swaps, iterations spread over a few cores (helpers that SPIN beside the count, waiting for batches, cost
    // more than they gain on hosts with a CPU quota)
    const unsigned hw = usable_cpus();
    const int threads = (size_t)n_iter * n < (1u << 16) ? 1 : (int)std::min(8u, hw);
    constexpr size_t kBatch = 8;
    parallel_for(((size_t)n_iter + kBatch - 1) / kBatch, threads, [&](size_t w) {
        for (size_t it = w * kBatch; it < std::min<size_t>(n_iter, (w + 1) * kBatch); ++it) {
            int32_t *x = out_perms + it * n;
            for (uint32_t i = 0; i < n; ++i) x[i] = (int32_t)i;
            uint32_t i = n - 1;
            for (size_t k = start[it]; i >= 1; ++k) {   // a rejected draw swaps a position with itself: no branch per draw
                const uint32_t mask = 0xFFFFFFFFu >> __builtin_clz(i);
                const uint32_t j = raw[k] & mask;
                const uint32_t acc = j <= i;
                const uint32_t jj = acc ? j : i;
                const int32_t t = x[i]; x[i] = x[jj]; x[jj] = t;
                i -= acc;
            }
        }
    });
    lap_("swaps done");
    *pos = fresh ? (int32_t)(cur - last_block_at) : (int32_t)(*pos + (int32_t)cur);
    return PGX_OK;
}
int pgx_legacy_shuffles(uint32_t *key, int32_t *pos, uint32_t n, uint32_t n_iter, int32_t *out_perms) {
    return guarded("pgx_legacy_shuffles", [&] { return pgx_legacy_shuffles_impl(key, pos, n, n_iter, out_perms); });
}

// Feature names as fixed-width, zero-padded ASCII records (numpy dtype 'S<width>'):
// <prefix><cluster[i]>            (variant == NULL: gene names)
// <prefix><cluster[i]><variant><member[i]>   (allele names), reference pangenome.py:1944-1969.
int pgx_format_labels(const char *prefix, const char *variant, const int32_t *cluster, const int32_t *member,
                      uint64_t n, uint32_t width, char *out) {
    if (!prefix || !cluster || (variant && !member) || (n && !out)) { pgx_set_error("pgx_format_labels: NULL argument"); return PGX_ERR_INVALID; }
    const size_t pl = strlen(prefix);
    char num[64];
    for (uint64_t i = 0; i < n; ++i) {
        const int k = variant ? snprintf(num, sizeof num, "%d%s%d", cluster[i], variant, member[i]) : snprintf(num, sizeof num, "%d", cluster[i]);
        if (pl + (size_t)k > width) { pgx_set_error("pgx_format_labels: width %u too small", width); return PGX_ERR_INVALID; }
        char *o = out + i * width;
        memcpy(o, prefix, pl);
        memcpy(o + pl, num, (size_t)k);
        memset(o + pl + k, 0, width - pl - (size_t)k);
    }
    return PGX_OK;
}

/* the same names as numpy 'U<width>' records (UCS-4 code points, zero padded), written by several threads;
 * prefix and variant must be ASCII */
static int pgx_format_labels_ucs4_impl(const char *prefix, const char *variant, const int32_t *cluster, const int32_t *member,
                           uint64_t n, uint32_t width, uint32_t *out) {
    if (!prefix || !cluster || (variant && !member) || (n && !out)) { pgx_set_error("pgx_format_labels_ucs4: NULL argument"); return PGX_ERR_INVALID; }
    const size_t pl = strlen(prefix), vl = variant ? strlen(variant) : 0;
    for (size_t i = 0; i < pl; ++i) if ((unsigned char)prefix[i] >= 128) { pgx_set_error("pgx_format_labels_ucs4: prefix is not ASCII"); return PGX_ERR_INVALID; }
    for (size_t i = 0; i < vl; ++i) if ((unsigned char)variant[i] >= 128) { pgx_set_error("pgx_format_labels_ucs4: variant is not ASCII"); return PGX_ERR_INVALID; }
    std::atomic<int> bad{0};
    const size_t chunk = 1u << 15;
    parallel_for((size_t)((n + chunk - 1) / chunk), (int)std::max(1u, std::min(16u, usable_cpus())), [&](size_t c) {
        auto put_int = [](uint32_t *o, int32_t v) -> size_t {     // decimal digits of v >= 0 (names never hold negatives)
            char tmp[12];
            size_t k = 0;
            uint32_t u = v < 0 ? 0u : (uint32_t)v;
            do { tmp[k++] = (char)('0' + u % 10u); u /= 10u; } while (u);
            for (size_t i = 0; i < k; ++i) o[i] = (uint32_t)tmp[k - 1 - i];
            return k;
        };
        for (uint64_t i = c * chunk; i < std::min<uint64_t>(n, (c + 1) * chunk); ++i) {
            uint32_t *o = out + i * width, rec[64];
            size_t k = 0;
            if (pl + vl + 24 > 64 + (size_t)0 || width > 64) { bad = 1; return; }
            for (size_t t = 0; t < pl; ++t) rec[k++] = (uint32_t)(unsigned char)prefix[t];
            k += put_int(rec + k, cluster[i]);
            if (variant) {
                for (size_t t = 0; t < vl; ++t) rec[k++] = (uint32_t)(unsigned char)variant[t];
                k += put_int(rec + k, member[i]);
            }
            if (k > width) { bad = 2; return; }
            for (size_t t = 0; t < width; ++t) o[t] = t < k ? rec[t] : 0u;
        }
    });
    if (bad) { pgx_set_error("pgx_format_labels_ucs4: %s", bad == 2 ? "width too small" : "labels longer than 64 characters"); return PGX_ERR_INVALID; }
    return PGX_OK;
}

// ---- the feature tables' two orderings (pangenome.py:563-680) ---------------------------------------------------
namespace {
// Key under which non-negative integers order like their decimal strings inside an allele name: digits left-aligned to
// `width` digits; where the name ends after the number the shorter string sorts first (1 < 10 < 100), where a letter
// follows (it sorts after every digit) the longer one does (100A < 10A < 1A); the digit count breaks the ties of the padding.
inline uint32_t digits_of(uint64_t v) { uint32_t d = 1; while (v >= 10) { v /= 10; ++d; } return d; }
inline uint64_t name_key(uint64_t v, uint32_t width, bool shorter_first) {
    const uint32_t nd = digits_of(v);
    uint64_t scale = 1;
    for (uint32_t i = nd; i < width; ++i) scale *= 10;
    return shorter_first ? v * scale * 32 + nd : ((v + 1) * scale - 1) * 32 + (31 - nd);
}
// stable LSD radix sort of (key, index) pairs, 11 bits per pass, passes in which all keys agree skipped
void radix_by_key(std::vector<uint64_t> &key, std::vector<uint32_t> &idx, std::vector<uint64_t> &key2, std::vector<uint32_t> &idx2) {
    const size_t n = key.size();
    constexpr int kBits = 11, kPasses = (64 + kBits - 1) / kBits;
    std::vector<uint32_t> hist((size_t)kPasses << kBits, 0u);
    for (size_t i = 0; i < n; ++i)
        for (int p = 0; p < kPasses; ++p) ++hist[((size_t)p << kBits) + ((key[i] >> (p * kBits)) & ((1u << kBits) - 1))];
    for (int p = 0; p < kPasses; ++p) {
        uint32_t *h = &hist[(size_t)p << kBits];
        bool one = false;
        for (uint32_t b = 0; b < (1u << kBits) && !one; ++b) one = h[b] == n;
        if (one) continue;
        uint32_t run = 0;
        for (uint32_t b = 0; b < (1u << kBits); ++b) { const uint32_t c = h[b]; h[b] = run; run += c; }
        for (size_t i = 0; i < n; ++i) {
            const uint32_t at = h[(key[i] >> (p * kBits)) & ((1u << kBits) - 1)]++;
            key2[at] = key[i]; idx2[at] = idx[i];
        }
        key.swap(key2); idx.swap(idx2);
    }
}
}  // namespace

/* Row order of the allele table: out_order[i] = position (in the given arrays) of the i-th allele when the names
 * <prefix><cluster><letter><member> are sorted as strings (reference pangenome.py:615 sorts the names themselves).
 * Stable; cluster[i], member[i] >= 0; n < 2^32. */
static int pgx_allele_order_impl(const int32_t *cluster, const int32_t *member, uint64_t n, int64_t *out_order) {
    if ((n && (!cluster || !member || !out_order)) || n >= (1ull << 32)) { pgx_set_error("pgx_allele_order: bad argument"); return PGX_ERR_INVALID; }
    int32_t cmax = 0, mmax = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if (cluster[i] < 0 || member[i] < 0) { pgx_set_error("pgx_allele_order: negative cluster or member number"); return PGX_ERR_INVALID; }
        cmax = std::max(cmax, cluster[i]); mmax = std::max(mmax, member[i]);
    }
    const uint32_t cw = digits_of((uint64_t)cmax), mw = digits_of((uint64_t)mmax);
    std::vector<uint64_t> key(n), key2(n);
    std::vector<uint32_t> idx(n), idx2(n);
    for (uint64_t i = 0; i < n; ++i) { key[i] = name_key((uint64_t)member[i], mw, true); idx[i] = (uint32_t)i; }
    radix_by_key(key, idx, key2, idx2);                 // by member first ...
    for (uint64_t i = 0; i < n; ++i) key[i] = name_key((uint64_t)cluster[idx[i]], cw, false);
    radix_by_key(key, idx, key2, idx2);                 // ... then, stably, by cluster
    for (uint64_t i = 0; i < n; ++i) out_order[i] = (int64_t)idx[i];
    return PGX_OK;
}

/* The triples a dictionary-of-keys matrix keeps when (rows[i], cols[i]) are inserted one after the other (reference
 * pangenome.py:649-650, scipy dok_matrix): out_first = the ascending positions i whose pair occurs there for the first
 * time. rows, cols >= 0; rows[i] * n_cols + cols[i] must fit 63 bits. */
static int pgx_first_insertions_impl(const int64_t *rows, const int64_t *cols, uint64_t n, uint64_t n_cols, int64_t *out_first,
                                     uint64_t *out_count) {
    if ((n && (!rows || !cols || !out_first)) || !out_count) { pgx_set_error("pgx_first_insertions: NULL argument"); return PGX_ERR_INVALID; }
    if (!n_cols) n_cols = 1;
    size_t cap = 16;
    while (cap < 2 * n + 2) cap <<= 1;
    const int shift = 64 - __builtin_ctzll(cap);
    std::vector<uint64_t> slot(cap, 0ull);              // key + 1, 0 = empty
    uint64_t m = 0;
    const uint64_t row_limit = 0x7ffffffffffffffeull / n_cols - 1;
    constexpr uint64_t kAhead = 24;                     // the table is far larger than the caches: its lines are asked for early
    auto home = [&](uint64_t i) { return (size_t)((((uint64_t)rows[i] * n_cols + (uint64_t)cols[i] + 1) * 0x9E3779B97F4A7C15ull) >> shift); };
    for (uint64_t i = 0; i < n; ++i) {
        if (rows[i] < 0 || cols[i] < 0 || (uint64_t)cols[i] >= n_cols || (uint64_t)rows[i] > row_limit) {
            pgx_set_error("pgx_first_insertions: coordinate out of range at %llu", (unsigned long long)i);
            return PGX_ERR_INVALID;
        }
        if (i + kAhead < n) __builtin_prefetch(&slot[home(i + kAhead)], 1);
        const uint64_t k = (uint64_t)rows[i] * n_cols + (uint64_t)cols[i] + 1;
        size_t h = (size_t)((k * 0x9E3779B97F4A7C15ull) >> shift);
        for (;;) {
            const uint64_t v = slot[h];
            if (v == k) break;
            if (!v) { slot[h] = k; out_first[m++] = (int64_t)i; break; }
            h = (h + 1) & (cap - 1);
        }
    }
    *out_count = m;
    return PGX_OK;
}

/* The coordinates of both feature tables straight from the parsed files and the clustering result
 * (build_genetic_feature_tables, pangenome.py:563-680): see pgx.h. Files are independent -- every file is one genome, so a
 * pair can only repeat inside a file -- and are worked on side by side, each with a small table of its own. */
static int pgx_fasta_feature_coo_impl(const pgx_fasta_set *S, const int32_t *cluster, const int32_t *member,
                                      const int32_t *file_order, const int32_t *genome_of_file, int64_t *allele_groups,
                                      int32_t *gene_of_allele, uint64_t *n_alleles, uint64_t *n_genes, int32_t *a_row,
                                      int32_t *a_col, uint64_t *a_nnz, int32_t *g_row, int32_t *g_col, uint64_t *g_nnz,
                                      int64_t *lost_records, uint64_t *n_lost) {
    if (!S || !S->simple || !cluster || !member || !file_order || !genome_of_file || !allele_groups || !gene_of_allele ||
        !n_alleles || !n_genes || !a_row || !a_col || !a_nnz || !g_row || !g_col || !g_nnz || !lost_records || !n_lost) {
        pgx_set_error("pgx_fasta_feature_coo: invalid argument");
        return PGX_ERR_INVALID;
    }
    const uint64_t G = S->n_groups, NF = S->files.size();
    {   // every file once in file_order, every genome once
        std::vector<uint8_t> seen_file(NF, 0), seen_genome(NF, 0);
        for (uint64_t i = 0; i < NF; ++i) {
            const int32_t f = file_order[i], g = f >= 0 && (uint64_t)f < NF ? genome_of_file[f] : -1;
            if (g < 0 || (uint64_t)g >= NF || seen_file[(size_t)f]++ || seen_genome[(size_t)g]++) {
                pgx_set_error("pgx_fasta_feature_coo: file_order / genome_of_file must be permutations of the files");
                return PGX_ERR_INVALID;
            }
        }
    }
    // rows of the allele table: the clustered groups in the order of their names; genes: runs of one cluster
    std::vector<int32_t> cl_c, mem_c;
    std::vector<uint32_t> group_c;
    for (uint64_t k = 0; k < G; ++k) if (cluster[k] >= 0) { cl_c.push_back(cluster[k]); mem_c.push_back(member[k]); group_c.push_back((uint32_t)k); }
    const uint64_t A = cl_c.size();
    std::vector<int64_t> order(A);
    const int rc = pgx_allele_order_impl(cl_c.data(), mem_c.data(), A, order.data());
    if (rc) return rc;
    std::vector<int32_t> row_of_group(G, -1);
    int32_t gene = -1;
    for (uint64_t i = 0; i < A; ++i) {
        const uint32_t k = group_c[(size_t)order[i]];
        allele_groups[i] = (int64_t)k;
        row_of_group[k] = (int32_t)i;
        if (i == 0 || cluster[k] != cluster[(size_t)allele_groups[i - 1]]) ++gene;
        gene_of_allele[i] = gene;
    }
    *n_alleles = A; *n_genes = (uint64_t)(gene + 1);
    // per file: its first insertions (allele rows, gene rows) and the records that have no row, in record order
    struct PerFile { std::vector<int32_t> a, g; std::vector<int64_t> lost; };
    std::vector<PerFile> per(NF);
    parallel_for(NF, S->threads, [&](size_t f) {
        const uint64_t r0 = S->first_rec[f], r1 = S->first_rec[f + 1];
        size_t cap = 16;
        while (cap < 2 * (r1 - r0) + 2) cap <<= 1;
        std::vector<int32_t> ta(cap, -1), tg(cap, -1);
        PerFile &P = per[f];
        auto insert = [&](std::vector<int32_t> &t, int32_t v) {     // true: v is new
            size_t h = ((size_t)(uint32_t)v * 0x9E3779B1u) & (cap - 1);
            for (;; h = (h + 1) & (cap - 1)) {
                if (t[h] == v) return false;
                if (t[h] < 0) { t[h] = v; return true; }
            }
        };
        for (uint64_t r = r0; r < r1; ++r) {
            const int32_t grp = S->group_of[r];
            if (grp == -1) continue;                                  // a header without a sequence
            const int32_t row = grp >= 0 ? row_of_group[(size_t)grp] : -1;
            if (row < 0) { P.lost.push_back((int64_t)r); continue; }   // no name, or a sequence the clustering discarded
            if (insert(ta, row)) P.a.push_back(row);
            if (insert(tg, gene_of_allele[row])) P.g.push_back(gene_of_allele[row]);
        }
    });
    uint64_t na = 0, ng = 0, nl = 0;
    std::vector<uint64_t> at_a(NF), at_g(NF);
    for (uint64_t i = 0; i < NF; ++i) {
        const size_t f = (size_t)file_order[i];
        at_a[f] = na; na += per[f].a.size();
        at_g[f] = ng; ng += per[f].g.size();
        for (int64_t r : per[f].lost) lost_records[nl++] = r;
    }
    parallel_for(NF, S->threads, [&](size_t f) {
        const int32_t col = genome_of_file[f];
        std::copy(per[f].a.begin(), per[f].a.end(), a_row + at_a[f]);
        std::fill(a_col + at_a[f], a_col + at_a[f] + per[f].a.size(), col);
        std::copy(per[f].g.begin(), per[f].g.end(), g_row + at_g[f]);
        std::fill(g_col + at_g[f], g_col + at_g[f] + per[f].g.size(), col);
    });
    *a_nnz = na; *g_nnz = ng; *n_lost = nl;
    return PGX_OK;
}

// the allocating entry points behind their exception guards
int pgx_fasta_feature_coo(const pgx_fasta_set *S, const int32_t *cluster, const int32_t *member, const int32_t *file_order,
                          const int32_t *genome_of_file, int64_t *allele_groups, int32_t *gene_of_allele, uint64_t *n_alleles,
                          uint64_t *n_genes, int32_t *a_row, int32_t *a_col, uint64_t *a_nnz, int32_t *g_row, int32_t *g_col,
                          uint64_t *g_nnz, int64_t *lost_records, uint64_t *n_lost) {
    return guarded("pgx_fasta_feature_coo", [&] {
        return pgx_fasta_feature_coo_impl(S, cluster, member, file_order, genome_of_file, allele_groups, gene_of_allele, n_alleles,
                                          n_genes, a_row, a_col, a_nnz, g_row, g_col, g_nnz, lost_records, n_lost);
    });
}
int pgx_allele_order(const int32_t *cluster, const int32_t *member, uint64_t n, int64_t *out_order) {
    return guarded("pgx_allele_order", [&] { return pgx_allele_order_impl(cluster, member, n, out_order); });
}
int pgx_first_insertions(const int64_t *rows, const int64_t *cols, uint64_t n, uint64_t n_cols, int64_t *out_first,
                         uint64_t *out_count) {
    return guarded("pgx_first_insertions", [&] { return pgx_first_insertions_impl(rows, cols, n, n_cols, out_first, out_count); });
}
int pgx_fasta_open(const char *const *paths, uint32_t n_paths, int n_threads, pgx_fasta_set **out) {
    return guarded("pgx_fasta_open", [&] { return pgx_fasta_open_impl(paths, n_paths, n_threads, out); });
}
int pgx_fasta_write_consolidated(const pgx_fasta_set *S, const char *nr_path, const char *shared_path,
                                 const char *missing_path) {
    return guarded("pgx_fasta_write_consolidated", [&] { return pgx_fasta_write_consolidated_impl(S, nr_path, shared_path, missing_path); });
}
int pgx_fasta_write_clustered(const pgx_fasta_set *S, const int32_t *cluster, const int32_t *member,
                              const float *identity, const uint8_t *strand, int nucleotide, const char *prefix,
                              const char *variant, const char *clstr_path, const char *names_path,
                              const char *nr_out_path) {
    return guarded("pgx_fasta_write_clustered", [&] {
        return pgx_fasta_write_clustered_impl(S, cluster, member, identity, strand, nucleotide, prefix, variant, clstr_path,
                                              names_path, nr_out_path);
    });
}
int pgx_format_labels_ucs4(const char *prefix, const char *variant, const int32_t *cluster, const int32_t *member,
                           uint64_t n, uint32_t width, uint32_t *out) {
    return guarded("pgx_format_labels_ucs4", [&] { return pgx_format_labels_ucs4_impl(prefix, variant, cluster, member, n, width, out); });
}

}  // extern "C"
