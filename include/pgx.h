/*
 * pgx.h -- C ABI of libpgx.so, the MI355X (gfx950) hot path of pangenomix.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch types.
 * It replaces two things in the reference (AnnaLew/pangenomix, a pure-Python
 * package that has no native layer of its own):
 *
 *   pangenome.py:425-450 + :2061-2068   cluster_with_cdhit(): the shell pipe to the
 *                                       external `cd-hit` / `cd-hit-est` programs
 *                                       -> pgx_cluster_greedy*
 *   pangenome_analysis.py:72-98         estimate_pan_core_size(): the Python double loop
 *                                       over scipy CSR rows -> pgx_presence_bitmap* +
 *                                       pgx_pan_core*
 *
 * The reference-side binding is a ctypes stub. Conventions:
 *   - every function returns 0 on success and a negative pgx_status on error;
 *     pgx_last_error() gives the thread-local message; no exception crosses the ABI;
 *   - the caller owns every buffer; the library keeps no caller pointer after return;
 *   - pgx_ctx owns device state (stream, scratch); one context per thread, calls on
 *     one context are not re-entrant;
 *   - *_dev variants take DEVICE pointers and a hipStream_t (as void*); they only
 *     enqueue work on that stream (graph-capturable: no allocation, no sync) and are
 *     what bench.py times with inputs resident in HBM. The plain variants take HOST
 *     pointers and do the copies themselves.
 *   - there is no CPU fallback: without a usable GPU every compute entry point fails
 *     with PGX_ERR_NO_DEVICE.
 */
#ifndef PGX_H
#define PGX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PGX_VERSION 300 /* 0.3.0 */
#define PGX_EXCHANGE_KEYS 65536u /* keys per process and exchange = the largest window */
#define PGX_EXCHANGE_WORDS (PGX_EXCHANGE_KEYS + 8u) /* uint64 per process and exchange: the keys + the error word (+ padding) */
#define PGX_EXCHANGE_SLOTS 2u    /* windows in flight, each with its own send / receive buffers */

typedef enum pgx_status {
    PGX_OK = 0,
    PGX_ERR_INVALID = -1,   /* bad argument */
    PGX_ERR_NO_DEVICE = -2, /* no usable HIP device */
    PGX_ERR_HIP = -3,       /* a HIP runtime call failed */
    PGX_ERR_NOMEM = -4,
    PGX_ERR_CAPACITY = -5,  /* an internal fixed-capacity buffer overflowed */
    PGX_ERR_INTERNAL = -6
} pgx_status;

typedef struct pgx_ctx pgx_ctx;

typedef struct pgx_device_info_t {
    char name[64];
    char arch[32];          /* gcnArchName, e.g. "gfx950:sramecc+:xnack-" */
    int32_t device_id;
    int32_t compute_units;
    int32_t wavefront_size;
    int32_t lds_bytes_per_block;
    uint64_t hbm_bytes;
    int32_t clock_khz;
    int32_t reserved;
} pgx_device_info_t;

int pgx_version(void);
const char *pgx_last_error(void);
/* One context = ONE device. SURVEY 8b sketched `pgx_ctx_create(const int *device_ids, int n_devices, ...)`; the
 * build deliberately deviates: the multi-GPU mode is one PROCESS per GPU (torch.distributed / RCCL, see
 * pgx_cluster_params.shard_*), each process with its own single-device context, so a context never spans
 * devices. pgx_ctx_create_on() takes the surveyed argument shape and accepts exactly one device id. */
int pgx_ctx_create(int device_id, pgx_ctx **out);
int pgx_ctx_create_on(const int *device_ids, int n_devices, pgx_ctx **out);
void pgx_ctx_destroy(pgx_ctx *ctx);
int pgx_device_info(pgx_ctx *ctx, pgx_device_info_t *out);

/* Optional per-kernel timing for bench.py's roofline line: when enabled, every kernel the
 * library launches is bracketed by HIP events on the stream it is launched on; slots
 * accumulate by kernel name. Reading a slot waits for its pending events. */
int pgx_profile_enable(pgx_ctx *ctx, int on);
int pgx_profile_reset(pgx_ctx *ctx);
int pgx_profile_count(pgx_ctx *ctx);
int pgx_profile_read(pgx_ctx *ctx, int slot, char *name, size_t name_bytes, double *total_ms,
                     uint64_t *launches);

/* ------------------------------------------------------------------------------------
 * K3: presence/absence bitmap and pan/core rarefaction curves
 * (pangenome_analysis.py:72-98).
 *
 * Bitmap layout: genome-major; genome s owns `stride_words` 64-bit words, gene g is
 * bit (g & 63) of word (g >> 6). stride_words = pgx_bitmap_stride_words(n_genes)
 * (rows padded to a multiple of 16 words = 128 B so that every lane's 16-byte load is
 * aligned and in bounds; pad bits are zero).
 * ---------------------------------------------------------------------------------- */
uint32_t pgx_bitmap_stride_words(uint32_t n_genes);

/* bits[genome][word] |= 1 for every (row_of_record[k], genome_of_record[k]).
 * out_bits must hold n_genomes * stride_words words; it is zeroed first.
 * Duplicates: the number of records whose bit was already set, i.e. repeated (row, genome)
 * coordinates (counted on the device while the bitmap is built: the read-modify-write returns the
 * word before the update). The reference's loop (pangenome_analysis.py:88-90) equals the OR/AND
 * form only for a 0/1 table without duplicates, so the Python layer refuses a table with any.
 * Host variant: out_duplicates may be NULL; a record out of range fails with PGX_ERR_INVALID.
 * Device variant: d_counters = 2 x uint64 in device memory {duplicates, records out of range
 * (skipped)}, zeroed first; may be NULL. */
int pgx_presence_bitmap(pgx_ctx *ctx, const int32_t *row_of_record, const int32_t *genome_of_record,
                        uint64_t n_records, uint32_t n_rows, uint32_t n_genomes, uint64_t *out_bits,
                        uint64_t *out_duplicates);
int pgx_presence_bitmap_dev(pgx_ctx *ctx, const int32_t *d_row_of_record,
                            const int32_t *d_genome_of_record, uint64_t n_records, uint32_t n_rows,
                            uint32_t n_genomes, uint64_t *d_out_bits, uint64_t *d_counters, void *stream);

/* For each iteration i and step j (genomes taken in the order perms[i][0..j]):
 *   out_pan[i][j]  = number of genes present in at least one of the first j+1 genomes
 *   out_core[i][j] = number of genes present in all of the first j+1 genomes
 * perms: [n_iter][n_genomes] int32, each row a permutation of 0..n_genomes-1 (generated by
 * the caller with the legacy numpy RNG, pangenome_analysis.py:84-85). Outputs [n_iter][n_genomes]. */
int pgx_pan_core(pgx_ctx *ctx, const uint64_t *bits, uint32_t n_genes, uint32_t n_genomes,
                 const int32_t *perms, uint32_t n_iter, int32_t *out_pan, int32_t *out_core);
/* The whole of estimate_pan_core_size()'s device work in one call (pangenome_analysis.py:72-98):
 * COO coordinates of the binary gene x genome table and the permutations go up once, the bitmap
 * is built and consumed on the device (no bitmap round trip), the two curves come back. Device
 * buffers live in the context's workspace: repeated calls do not allocate. */
int pgx_pan_core_coo(pgx_ctx *ctx, const int32_t *row_of_record, const int32_t *genome_of_record,
                     uint64_t n_records, uint32_t n_genes, uint32_t n_genomes, const int32_t *perms,
                     uint32_t n_iter, int32_t *out_pan, int32_t *out_core, uint64_t *out_duplicates);
/* Row occupancy: out_counts[r] = number of genomes that hold row r (gene or allele) = the popcount of
 * the row across the bitmap. What core_genome.py:127-155 (count_gene_occurence) and
 * allele_identification.py:129-157 (count_allele_occurence) obtain from the .npz triples with a pandas
 * groupby; equal to their counts for a table without duplicate coordinates (duplicates reported). */
int pgx_row_counts(pgx_ctx *ctx, const int32_t *row_of_record, const int32_t *genome_of_record,
                   uint64_t n_records, uint32_t n_rows, uint32_t n_genomes, int32_t *out_counts,
                   uint64_t *out_duplicates);
int pgx_row_counts_dev(pgx_ctx *ctx, const uint64_t *d_bits, uint32_t n_rows, uint32_t n_genomes,
                       int32_t *d_counts, void *stream);
size_t pgx_pan_core_workspace_bytes(uint32_t n_genes, uint32_t n_genomes, uint32_t n_iter);
int pgx_pan_core_dev(pgx_ctx *ctx, const uint64_t *d_bits, uint32_t n_genes, uint32_t n_genomes,
                     const int32_t *d_perms, uint32_t n_iter, int32_t *d_out_pan,
                     int32_t *d_out_core, void *d_workspace, size_t workspace_bytes, void *stream);

/* Heaps-law fits of the pan curves (pangenome_analysis.py:24-48, fit_heaps_by_iteration): per
 * iteration i the least-squares (alpha, kappa) of  pan[i][j-1] = kappa * j^alpha,  j = 1..n_genomes, from
 * the reference's start point (0.5, min of the row). Floating point: equal to scipy's curve_fit to a
 * tolerance (tests: rtol 1e-5; scipy's default stopping rule leaves it ~3e-6 off the minimum). The device variant reads pgx_pan_core_dev's int32 output in place. */
int pgx_heaps_fit(pgx_ctx *ctx, const double *pan, uint32_t n_iter, uint32_t n_genomes, double *out_alpha,
                  double *out_kappa);
int pgx_heaps_fit_dev(pgx_ctx *ctx, const int32_t *d_pan, uint32_t n_iter, uint32_t n_genomes,
                      double *d_alpha, double *d_kappa, void *stream);

/* ------------------------------------------------------------------------------------
 * K1/K2: greedy incremental clustering with cd-hit's rules.
 * ---------------------------------------------------------------------------------- */
typedef struct pgx_cluster_params {
    int32_t alphabet;       /* 0 = protein (cd-hit), 1 = nucleotide (cd-hit-est) */
    int32_t word_len;       /* -n, 2..5 protein; nucleotide words use the same value */
    int32_t band_width;     /* -b, default 20 */
    int32_t min_length;     /* -l, default 10: sequences with length <= this are discarded */
    int32_t both_strands;   /* -r, nucleotide only, default 1 */
    int32_t batch_size;     /* queries per window (greedy sweep), at most PGX_EXCHANGE_KEYS; 0 = library default
                             * (pgx_cluster_window_cap). Any value gives the same clusters. */
    double identity;        /* -c, global identity threshold (double, as cd-hit parses it) */
    /* short-word filter cut-offs as fractions of the query length. cd-hit takes
     * max(analytic bound, naa_stat[tolerance-1][100c-40][...]/100); its table is not
     * available offline, so the caller passes the final values (see
     * pangenomix_amd/cluster.py:filter_cutoffs and DESIGN.md "filter table"). */
    double aan_cutoff;      /* shared word_len-mers   */
    double aas_cutoff;      /* shared 2-mers (4-mers for nucleotides) on the best band */
    /* Record-sharded multi-GPU mode, one process per GPU, every process called with the same
     * sequences (all 0 / NULL = single GPU). The sorted list is cut into windows; member i of a
     * window belongs to process i % shard_count, which runs the short-word filter and the
     * alignments of its members against its replica of the representative index. After every
     * evaluation the window's best keys (one uint64 per member; the minimum over the accepted
     * representatives) are ALL-GATHERED: the library keeps them in exchange_send, calls
     * `exchange`, and folds the rows of exchange_recv. New representatives follow from the
     * gathered keys by a deterministic rule every process repeats, so all replicas of the index
     * stay identical and all processes return the same clusters. Work counters and identities are
     * partial per process: sum the counters / take the maximum of out_identity over the processes
     * (pangenomix_amd/cluster.py does). */
    int32_t shard_index, shard_count;
    /* All-gather of PGX_EXCHANGE_WORDS uint64 per process: slot `slot` of exchange_send of every process p into
     * exchange_recv[slot][p][...] of all. Two consecutive windows are in flight on two streams, each with its own
     * slot (send: [PGX_EXCHANGE_SLOTS][PGX_EXCHANGE_WORDS], recv: [PGX_EXCHANGE_SLOTS][shard_count][PGX_EXCHANGE_WORDS]).
     * The callback ENQUEUES the collective on `stream` (the stream that window works on) and returns without
     * waiting: e.g. torch.distributed.all_gather_into_tensor under torch.cuda.ExternalStream(stream) = RCCL over
     * xGMI with the nccl backend. Every process issues its calls in the same order. 0 = success. A few calls per
     * window. The word behind the keys carries the process's error state: a capacity failure on one process makes
     * every process return PGX_ERR_CAPACITY at the same point (nobody is left waiting in a collective). */
    /* (exchange == NULL with shard_count >= 1: the context's own communicator, pgx_rccl_comm_create below; the
     * exchange_* fields are then unused) */
    int (*exchange)(void *user, void *stream, int slot);
    void *exchange_user;
    void *exchange_send;    /* device (the caller's allocation, so that its collective library can address it) */
    void *exchange_recv;    /* device */
    /* cd-hit's MEMORY-CHUNKED rule, optional emulation. cd-hit bounds its word table by what is left
     * of `-M` (default 800 MB; the reference's call passes no -M, pangenome.py:444-447): when the table is full,
     * every sequence not yet clustered is compared with the current table at once (and joins the first
     * representative that accepts it), the table is emptied, and clustering goes on with what is left. A
     * sequence therefore joins a representative of the EARLIEST chunk that accepts it, which can differ from the
     * unchunked winner. Where cd-hit places the boundaries depends on its memory accounting, which cannot be
     * reproduced offline, so they are an input: positions in the length-sorted list of the clustered sequences
     * (those longer than min_length), strictly increasing, each in (0, n_clustered); position b means "the table
     * is flushed before the b-th sequence of that list is processed". NULL / 0 = the unchunked rule (-M 0). */
    const uint32_t *chunk_boundaries;
    uint32_t n_chunk_boundaries;
    uint32_t reserved0;
} pgx_cluster_params;

/* Instrumentation that defines the roofline denominator. All are
 * properties of the sequential algorithm (what a one-by-one greedy pass visits), so the
 * oracle and the GPU path must agree on every field except `sweeps`. */
typedef struct pgx_cluster_stats {
    uint64_t n_input;          /* sequences handed in */
    uint64_t n_clustered;      /* N: sequences longer than min_length */
    uint64_t n_clusters;
    uint64_t sum_len_queries;  /* sum of L over clustered sequences */
    uint64_t sum_len_reps;     /* R */
    uint64_t rep_words;        /* R_words: distinct-word list entries written for representatives */
    uint64_t posting_visits;   /* P: posting entries a query's distinct words meet in the table */
    uint64_t filter_pairs;     /* (query, rep) pairs examined whose word count reached required_aan */
    uint64_t aligned_pairs;    /* of those, pairs that passed the diagonal test and were aligned */
    uint64_t aligned_rep_len;  /* A: sum of L_rep over aligned_pairs */
    uint64_t dp_cells;         /* sum over aligned_pairs of L_query * (band_right-band_left+1) */
    uint64_t sweeps;           /* windows (GPU path; 0 in the oracle) */
    /* GPU path only, actual device work incl. speculative pairs (0 in the oracle):
     * [0] candidate pairs through the diagonal test, [1] pairs aligned, [2] residue bytes of the
     * aligned pairs (len_query + len_rep, 1 byte per residue), [3] query words walked by the filter
     * passes (an upper bound: residues of the window per pass; each costs one bit-map probe and, for
     * the pass over the whole index, one 64-byte line read) */
    uint64_t reserved[4];
} pgx_cluster_stats;

/* residues: concatenated ASCII letters (already validated / upper-cased by the caller),
 * offsets: n+1 entries. Sequences are processed in stable descending-length order.
 *   out_cluster[i]  cluster number in order of representative creation, -1 = discarded
 *   out_member[i]   index inside the cluster = .clstr member number (0 = representative)
 *   out_identity[i] matches / query length as float (0 for representatives)
 *   out_strand[i]   0 '+', 1 '-' (nucleotide, both_strands; may be NULL)
 *   stats           may be NULL. The counters are those of the SEQUENTIAL rule, which looks up every word of every
 *                   sequence in the whole table; with stats the library does all of those look-ups so that its
 *                   counters equal that rule's. Without, the passes over a window's new representatives leave out the
 *                   members that cannot gain from them (final ones; those whose best candidate no new representative
 *                   can precede). The clusters, member numbers, identities and strands are the same either way. */
int pgx_cluster_greedy(pgx_ctx *ctx, const uint8_t *residues, const uint64_t *offsets, uint32_t n,
                       const pgx_cluster_params *params, int32_t *out_cluster, int32_t *out_member,
                       float *out_identity, uint8_t *out_strand, uint32_t *out_n_clusters,
                       pgx_cluster_stats *stats);
/* Same, with the sequences already resident in HBM: d_residues / d_offsets are DEVICE
 * pointers (total_bytes = offsets[n]); the out_* arrays and stats are HOST memory. The inputs are
 * read on `stream` (work already enqueued there is waited for); consecutive windows then alternate
 * between `stream` and a second, lower-priority stream of the context, ordered by events. The call
 * returns after the last window has been resolved and both streams have drained. */
/* queries per window the library will use for these parameters (env PGX_WINDOW overrides) */
uint32_t pgx_cluster_window_cap(const pgx_cluster_params *params);
int pgx_cluster_greedy_dev(pgx_ctx *ctx, const uint8_t *d_residues, const uint64_t *d_offsets, uint32_t n,
                           uint64_t total_bytes, const pgx_cluster_params *params, int32_t *out_cluster,
                           int32_t *out_member, float *out_identity, uint8_t *out_strand,
                           uint32_t *out_n_clusters, pgx_cluster_stats *stats, void *stream);

/* The record-sharded mode without a callback: the library's own RCCL communicator (optional). With shard_count >= 1,
 * exchange == NULL and a communicator on the context, pgx_cluster_greedy[_dev] keeps the exchange buffers itself and
 * enqueues ncclAllGather on the window's stream (RCCL over xGMI; nothing of the host language in the loop).
 *   pgx_rccl_load         dlopen of librccl.so (path NULL = the loader's search path; a PyTorch-ROCm process passes
 *                         torch/lib/librccl.so so that the copy PyTorch uses is shared). libpgx does not link RCCL.
 *   pgx_rccl_unique_id    128 bytes made by ONE process and handed to the others by the caller's own means
 *                         (a file, MPI, torch.distributed.broadcast_object_list ...)
 *   pgx_rccl_comm_create  collective: every process calls it with the same id, its rank and the world size
 *   pgx_rccl_comm_destroy (pgx_ctx_destroy does it too)
 * The reference has no counterpart (its clustering is one cd-hit process, pangenome.py:444-447). */
int pgx_rccl_load(const char *librccl_path);
int pgx_rccl_unique_id(uint8_t *out_id128);
int pgx_rccl_comm_create(pgx_ctx *ctx, const uint8_t *id128, int rank, int world);
int pgx_rccl_comm_destroy(pgx_ctx *ctx);

/* ------------------------------------------------------------------------------------
 * Host side of the pipeline around the clustering call: multi-threaded FASTA
 * ingest, first-seen exact de-duplication and the text outputs. No GPU involved.
 *
 *   pangenome.py:336-405   consolidate_seqs()             -> pgx_fasta_open + pgx_fasta_write_consolidated
 *   pangenome.py:425-450   FASTA -> sequences, .clstr out -> pgx_fasta_residues/offsets, pgx_fasta_write_clustered
 *   pangenome.py:453-560   rename_genes_and_alleles()     -> pgx_fasta_write_clustered
 *
 * Reading rules are the reference's (record = line starting with '>'; header = first whitespace
 * token minus '>'; sequence = stripped lines joined; no sequence = "missing"). Inputs its
 * line-by-line Python semantics treat specially (carriage returns, non-ASCII or control bytes,
 * one header naming two different sequences) are reported through info.simple = 0 with the reason in info.why: the array
 * accessors then return NULL and the Python layer takes its own statement-by-statement path.
 * Pointers returned by the accessors stay valid until pgx_fasta_close.
 * ---------------------------------------------------------------------------------- */
typedef struct pgx_fasta_set pgx_fasta_set;
typedef struct pgx_fasta_info_t {
    uint64_t n_records;        /* records, all files, in the order given */
    uint64_t n_missing;        /* of those, without sequence */
    uint64_t n_groups;         /* distinct sequences, first-seen order (the non-redundant set) */
    uint64_t n_residue_bytes;  /* sequence bytes of the non-redundant set */
    uint64_t n_header_bytes;
    uint32_t simple;           /* 1: everything below is available */
    uint32_t reserved;
    char why[256];             /* simple == 0: what was found */
} pgx_fasta_info_t;

int pgx_fasta_open(const char *const *paths, uint32_t n_paths, int n_threads /* 0 = all cores (<= 32) */,
                   pgx_fasta_set **out);
void pgx_fasta_close(pgx_fasta_set *fs);
int pgx_fasta_info(const pgx_fasta_set *fs, pgx_fasta_info_t *out);
const int32_t *pgx_fasta_group_of_record(const pgx_fasta_set *fs);   /* [n_records], -1 = no sequence, -2 = a
                                                                      * sequence without a header (nameless) */
const uint32_t *pgx_fasta_file_of_record(const pgx_fasta_set *fs);   /* [n_records] index into paths */
const uint64_t *pgx_fasta_rep_of_group(const pgx_fasta_set *fs);     /* [n_groups] first-seen record */
const uint8_t *pgx_fasta_residues(const pgx_fasta_set *fs);          /* the groups' sequences, concatenated: */
const uint64_t *pgx_fasta_offsets(const pgx_fasta_set *fs);          /* [n_groups + 1]; pgx_cluster_greedy's input */
const uint32_t *pgx_fasta_letters(const pgx_fasta_set *fs);          /* [n_groups] letters per sequence (.clstr length) */
const uint8_t *pgx_fasta_digests(const pgx_fasta_set *fs);           /* [n_groups][32] sha256 of the sequence */
const char *pgx_fasta_header_blob(const pgx_fasta_set *fs);          /* the records' headers, concatenated: */
const uint64_t *pgx_fasta_header_offsets(const pgx_fasta_set *fs);   /* [n_records + 1] */
/* consolidate_seqs()'s files: non-redundant FASTA, groups with several headers, headers without sequence */
int pgx_fasta_write_consolidated(const pgx_fasta_set *fs, const char *nr_path /* may be NULL */,
                                 const char *shared_path, const char *missing_path /* may be NULL */);
/* after pgx_cluster_greedy on (residues, offsets): cd-hit's .clstr, the allele name table
 * (<prefix><cluster><variant><member> TAB header TAB synonyms) and the non-redundant FASTA with the
 * allele names as headers (unclustered records dropped). NULL paths are skipped. */
int pgx_fasta_write_clustered(const pgx_fasta_set *fs, const int32_t *cluster, const int32_t *member,
                              const float *identity, const uint8_t *strand /* may be NULL */, int nucleotide,
                              const char *prefix, const char *variant, const char *clstr_path,
                              const char *names_path, const char *nr_out_path);

/* n_iter permutations of 0..n-1 drawn exactly as `p = np.arange(n); np.random.shuffle(p)` draws them from
 * numpy's legacy global generator (pangenome_analysis.py:84-85): MT19937 state key[624] / pos in, advanced
 * state out (np.random.get_state() / set_state()). out_perms: [n_iter][n] int32. */
int pgx_legacy_shuffles(uint32_t *key, int32_t *pos, uint32_t n, uint32_t n_iter, int32_t *out_perms);

/* pgx_legacy_shuffles + pgx_pan_core_coo in one call (pangenome_analysis.py:51-98 from the binary table's
 * coordinates and the generator state): the draws are made on a host thread while the coordinates are uploaded and
 * the bitmap is built. out_perms: [n_iter][n_genomes], the permutations that were used. */
int pgx_pan_core_coo_rng(pgx_ctx *ctx, const int32_t *rows, const int32_t *genomes, uint64_t n_records,
                         uint32_t n_genes, uint32_t n_genomes, uint32_t *mt_key, int32_t *mt_pos, uint32_t n_iter,
                         int32_t *out_perms, int32_t *out_pan, int32_t *out_core, uint64_t *out_duplicates);

/* The whole of estimate_pan_core_size() (pangenome_analysis.py:51-98) from the gene x genome table's COO arrays:
 * `values` (the table's stored values, int64; may be NULL) are checked to be all 1 (out_not_one = how many are
 * not; the curves are then NOT computed), the permutations are drawn from the generator state, and the result is
 * written as the reference returns it: out_table float64 [n_iter][2 * n_genomes], pan curves in columns
 * 0..n_genomes-1, core curves behind them. out_perms [n_iter][n_genomes] receives the permutations used. */
int pgx_pan_core_table(pgx_ctx *ctx, const int32_t *rows, const int32_t *genomes, const int64_t *values,
                       uint64_t n_records, uint32_t n_genes, uint32_t n_genomes, uint32_t *mt_key, int32_t *mt_pos,
                       uint32_t n_iter, int32_t *out_perms, double *out_table, uint64_t *out_duplicates,
                       uint64_t *out_not_one);

/* Device-resident hand-off (SURVEY build plan step 5; north_star "emitting the gene x genome presence/absence bitmap"):
 * the bitmap of a pangenome is built ON THE DEVICE straight from the clustering result -- record r of the genome files is
 * an instance of gene cluster_of_group[group_of_record[r]] in genome genome_of_file[file_of_record[r]] (negative group or
 * cluster: nothing), rows = cluster numbers -- and stays in the context until the next pgx_bitmap_from_clusters on it.
 * out_token names it; pgx_pan_core_table_resident computes estimate_pan_core_size()'s table from it with no upload of
 * the table at all (the curves do not depend on the order of the rows); pgx_bitmap_resident_read copies it out
 * (n_genomes x pgx_bitmap_stride_words(n_genes) words). A stale token fails with PGX_ERR_INVALID. */
int pgx_bitmap_from_clusters(pgx_ctx *ctx, const int32_t *cluster_of_group, uint64_t n_groups,
                             const int32_t *group_of_record, const uint32_t *file_of_record, uint64_t n_records,
                             const int32_t *genome_of_file, uint32_t n_files, uint32_t n_genes, uint32_t n_genomes,
                             uint64_t *out_token);
int pgx_bitmap_resident_read(pgx_ctx *ctx, uint64_t token, uint64_t *out_bits);
int pgx_pan_core_table_resident(pgx_ctx *ctx, uint64_t token, uint32_t n_genes, uint32_t n_genomes, uint32_t *mt_key,
                                int32_t *mt_pos, uint32_t n_iter, int32_t *out_perms, double *out_table);

/* feature names (pangenome.py:1944-1969) as fixed-width zero-padded ASCII records (numpy 'S<width>'):
 * <prefix><cluster>[<variant><member>]; variant NULL = gene names */
int pgx_format_labels(const char *prefix, const char *variant, const int32_t *cluster, const int32_t *member,
                      uint64_t n, uint32_t width, char *out);
/* the same names as numpy 'U<width>' records (UCS-4 code points, zero padded; prefix and variant ASCII, numbers >= 0,
 * width <= 64), written by several threads */
int pgx_format_labels_ucs4(const char *prefix, const char *variant, const int32_t *cluster, const int32_t *member,
                           uint64_t n, uint32_t width, uint32_t *out);

/* The two orderings of the feature tables (build_genetic_feature_tables, pangenome.py:563-680), host code:
 *   pgx_allele_order      out_order[i] = position of the i-th allele when the names <prefix><cluster><letter><member>
 *                         are sorted as strings (:615 sorts the names; here two stable radix passes over keys that
 *                         order integers like their decimal strings). cluster, member >= 0, n < 2^32.
 *   pgx_first_insertions  the triples a dictionary-of-keys matrix keeps when (rows[i], cols[i]) are set one after the
 *                         other (:649-650): out_first[0..*out_count) = ascending positions whose pair occurs there for
 *                         the first time (room for n entries). 0 <= cols[i] < n_cols, rows[i] >= 0. */
int pgx_allele_order(const int32_t *cluster, const int32_t *member, uint64_t n, int64_t *out_order);
/* Both tables' coordinates in one pass over a parsed set (the loops of pangenome.py:598-650 for inputs where every file is
 * one genome): cluster / member per non-redundant sequence as for pgx_fasta_write_clustered; file_order = the files in
 * the order their records are inserted (the reference walks sorted(paths)), genome_of_file = each file's column, both
 * permutations of 0..n_files-1. Out: allele_groups[i] = the sequence of allele row i (rows in pgx_allele_order's order),
 * gene_of_allele[i] = its gene row (room for n_groups each), *n_alleles, *n_genes; the COO triples of the allele and the
 * gene table in first-insertion order (a_row/a_col, g_row/g_col: room for n_records each; all values are 1);
 * lost_records = records that have no row (a sequence without a name, or one the clustering discarded), in
 * insertion order (room for n_records). */
int pgx_fasta_feature_coo(const pgx_fasta_set *fs, const int32_t *cluster, const int32_t *member, const int32_t *file_order,
                          const int32_t *genome_of_file, int64_t *allele_groups, int32_t *gene_of_allele, uint64_t *n_alleles,
                          uint64_t *n_genes, int32_t *a_row, int32_t *a_col, uint64_t *a_nnz, int32_t *g_row, int32_t *g_col,
                          uint64_t *g_nnz, int64_t *lost_records, uint64_t *n_lost);
int pgx_first_insertions(const int64_t *rows, const int64_t *cols, uint64_t n, uint64_t n_cols, int64_t *out_first,
                         uint64_t *out_count);

#ifdef __cplusplus
}
#endif
#endif /* PGX_H */
