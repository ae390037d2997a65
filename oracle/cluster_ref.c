/*
 * oracle/cluster_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Sequential CPU restatement of the greedy incremental clustering that the reference
 * obtains by shelling out to the third-party program cd-hit:
 *
 *   /root/reference/pangenomix/pangenome.py:444-450  `cd-hit -i X -o Y -d 0 -n 5 -c 0.8`
 *   /root/reference/pangenomix/pangenome.py:505-513  the only consumer of its .clstr output
 *
 * PARITY UNPINNED. The algorithm lives in cd-hit (github.com/weizhongli/cdhit; the
 * reference pins no version, README.md:22-26 says `conda install -c bioconda cd-hit`;
 * current bioconda release 4.8.1). Neither its source nor a binary is in this container
 * and the reference holds no test, fixture or golden .clstr for this boundary, so this
 * file restates cd-hit's published algorithm as described in SURVEY.md Appendix A
 * (A.2 encoding, A.3 order, A.4 per-query work, A.5 candidate order, A.7 .clstr) and in
 * the cd-hit papers (Li, Jaroszewski & Godzik 2001/2002; Li & Godzik 2006; Fu et al. 2012):
 * function names in comments ("CountWords", "diag_test_aapn", "local_band_align", ...)
 * refer to cd-hit 4.8.1 cdhit-common.c++ by recollection only. Known open points:
 *   - cd-hit's statistical filter table naa_stat[5][61][4] is not recoverable offline;
 *     the two cut-offs it feeds are therefore INPUTS (params->aan_cutoff/aas_cutoff);
 *   - cd-hit's default `-M 800` flushes the word table when memory fills (A.6). Where it does so
 *     depends on its memory accounting, which cannot be restated offline: the flush positions are an
 *     INPUT (params->chunk_boundaries, positions in the sorted list); without them this is the
 *     memory-independent, unchunked rule (`-M 0`).
 * What it pins instead: the HIP path must reproduce this file bit-for-bit (cluster ids,
 * member numbers, float identities, counters), and tests/test_cluster_oracle.py holds
 * known-answer cases (identical sequences, threshold straddling, discard length, ordering).
 *
 * Nucleotide rules (alphabet = 1, the reference's `cd-hit-est -n 5 -c 0.8` call for files that
 * end in .fna, pangenome.py:444) follow SURVEY.md A.1/A.2/A.4: A C G T/U -> 0..3, anything else
 * 4 (N); words are base-4 k-mers, words containing N are skipped; the diagonal test counts
 * shared 4-mers; scoring +2 / -2, gap open -6, extension -1; with both_strands the reverse
 * complement of the query is tried when the forward strand finds no representative.
 * Equally unpinned, for the same reason.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/pgx.h"

#define NAA1 21            /* protein alphabet size incl. X (A.2) */
#define MAX_NAA 5
#define SCORE_SCALE 655360 /* cd-hit MAX_SEQ: scales substitution scores above the tie-break bonus */

enum { BACK_NONE = 0, BACK_LEFT_TOP = 1, BACK_LEFT = 2, BACK_TOP = 3 };

/* A.2: letter -> residue index; index order A R N D C Q E G H I L K M F P S T W Y V, then 20
 * for everything ambiguous (B->N=2 and Z->E=6 are folded onto their amide/acid partner;
 * J O U X -> 20). Table as given in SURVEY.md A.2 for A..Z. */
static const int8_t AA2IDX[26] = {0, 2, 4, 3, 6, 13, 7, 8, 9, 20, 11, 10, 12,
                                  2, 20, 14, 5, 1, 15, 16, 20, 19, 17, 20, 18, 6};

/* BLOSUM62, lower triangle, residue order A R N D C Q E G H I L K M F P S T W Y V B Z X.
 * The matrix has 23 rows while AA2IDX only produces 0..20, so index 20 (J/O/U/X) reads
 * the row labelled B: a quirk of cd-hit as recollected, kept (rows 21, 22 are never read). */
static const int8_t BLOSUM62_TRI[] = {
    4,
    -1, 5,
    -2, 0, 6,
    -2, -2, 1, 6,
    0, -3, -3, -3, 9,
    -1, 1, 0, 0, -3, 5,
    -1, 0, 0, 2, -4, 2, 5,
    0, -2, 0, -1, -3, -2, -2, 6,
    -2, 0, 1, -1, -3, 0, 0, -2, 8,
    -1, -3, -3, -3, -1, -3, -3, -4, -3, 4,
    -1, -2, -3, -4, -1, -2, -3, -4, -3, 2, 4,
    -1, 2, 0, -1, -3, 1, 1, -2, -1, -3, -2, 5,
    -1, -1, -2, -3, -1, 0, -2, -3, -2, 1, 2, -1, 5,
    -2, -3, -3, -3, -2, -3, -3, -3, -1, 0, 0, -3, 0, 6,
    -1, -2, -2, -1, -3, -1, -1, -2, -2, -3, -3, -1, -2, -4, 7,
    1, -1, 1, 0, -1, 0, 0, 0, -1, -2, -2, 0, -1, -2, -1, 4,
    0, -1, 0, -1, -1, -1, -1, -2, -2, -1, -1, -1, -1, -2, -1, 1, 5,
    -3, -3, -4, -4, -2, -2, -3, -2, -2, -3, -2, -3, -1, 1, -4, -3, -2, 11,
    -2, -2, -2, -3, -2, -1, -2, -3, 2, -1, -1, -2, -1, 3, -3, -2, -2, 2, 7,
    0, -3, -3, -3, -1, -2, -2, -3, -3, 3, 1, -2, 1, -1, -2, -2, 0, -3, -1, 4,
    -2, -1, 3, 4, -3, 0, 1, -1, 0, -3, -4, 0, -3, -3, -2, 0, -1, -4, -3, -3, 4,
    -1, 0, 0, 1, -3, 3, 4, -2, 0, -3, -3, 1, -1, -3, -1, 0, -1, -3, -2, -2, 1, 4,
    0, -1, -1, -1, -2, -1, -1, -1, -1, -1, -1, -1, -1, -1, -2, 0, 0, -2, -1, -1, -1, -1, -1};

#define GAP_OPEN (-11)
#define GAP_EXT (-1)

/* exported so that tests and the GPU build can cross-check their own copy */
void pgxo_protein_tables(int8_t aa2idx[26], int8_t blosum[23][23]) {
    memcpy(aa2idx, AA2IDX, 26);
    int k = 0;
    for (int i = 0; i < 23; ++i)
        for (int j = 0; j <= i; ++j) blosum[i][j] = blosum[j][i] = BLOSUM62_TRI[k++];
}

typedef struct { uint32_t rep; uint32_t count; } post_t;
typedef struct { post_t *items; uint32_t size, cap; } postlist_t;

typedef struct {
    /* sequences in sorted order */
    uint32_t n;            /* clustered sequences (length > min_length) */
    const uint8_t **seq;   /* residue indices */
    uint32_t *len;
    uint32_t *orig;        /* original input index */
    /* word table ("WordTable") */
    uint32_t n_codes;
    postlist_t *table;
    uint32_t n_reps;
    uint32_t *rep_seq;     /* rep index -> sorted sequence index */
    /* per-query buffers ("WorkingBuffer") */
    int32_t *word_codes;   /* sorted */
    uint32_t *word_mult;   /* multiplicity on the first of a run, 0 on the others */
    post_t *look;          /* candidates in first-encounter order ("lookCounts") */
    uint32_t n_look;
    uint32_t *index_map;   /* rep -> position+1 in look ("indexMapping") */
    int32_t *taap, *aap_begin, *aap_list; /* 2-mer positions of the query ("ComputeAAP") */
    int32_t *diag_score, *diag_score2;
    int64_t *score_mat; uint8_t *back_mat; size_t mat_cells;
    int64_t sub[23][23];
    pgx_cluster_stats st;
    int word_len;
    int nt;            /* nucleotide rules */
    int base;          /* word radix: 21 or 4 */
    int kd;            /* k-mer length of the diagonal test: 2 or 4 */
    int nkd;           /* number of diagonal-test k-mer codes: 441 or 256 */
    int64_t gap_open, gap_ext;
    uint8_t *rc_buf;   /* reverse complement of the current query */
} state_t;

static int cmp_i32(const void *a, const void *b) {
    int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return (x > y) - (x < y);
}

/* "EncodeWords": all L-k+1 words, first residue most significant; sort ascending; collapse
 * runs to (code, multiplicity) with the multiplicity on the first element of the run. */
static uint32_t encode_words(state_t *S, const uint8_t *seq, uint32_t len) {
    const int k = S->word_len;
    const uint32_t npos = len - k + 1;
    uint32_t nw = 0;
    for (uint32_t j = 0; j < npos; ++j) {
        int32_t code = 0;
        int bad = 0;
        for (int t = 0; t < k; ++t) { code = code * S->base + seq[j + t]; bad |= seq[j + t] >= S->base; }
        if (S->nt && bad) continue; /* nucleotide words containing N are skipped */
        S->word_codes[nw++] = code;
    }
    qsort(S->word_codes, nw, sizeof(int32_t), cmp_i32);
    for (uint32_t j = 0; j < nw; ++j) S->word_mult[j] = 1;
    for (uint32_t j = nw; j-- > 1;)
        if (S->word_codes[j] == S->word_codes[j - 1]) {
            S->word_mult[j - 1] += S->word_mult[j];
            S->word_mult[j] = 0;
        }
    return nw;
}

/* "CountWords": walk the posting list of every distinct query word in ascending code
 * order; candidates are recorded in order of first encounter (A.4 step 2, A.5). */
static void count_words(state_t *S, uint32_t nw, int min_count) {
    for (uint32_t i = 0; i < S->n_look; ++i) S->index_map[S->look[i].rep] = 0;
    S->n_look = 0;
    for (uint32_t j0 = 0; j0 < nw; ++j0) {
        const uint32_t m = S->word_mult[j0];
        if (m == 0) continue;
        const postlist_t *pl = &S->table[S->word_codes[j0]];
        const int rest = (int)(nw - j0) + 1;
        S->st.posting_visits += pl->size;
        for (uint32_t k = 0; k < pl->size; ++k) {
            const uint32_t c = pl->items[k].count < m ? pl->items[k].count : m;
            uint32_t *slot = &S->index_map[pl->items[k].rep];
            if (*slot == 0) {
                if (rest < min_count) continue; /* cannot reach the threshold any more */
                S->look[S->n_look].rep = pl->items[k].rep;
                S->look[S->n_look].count = c;
                *slot = ++S->n_look;
            } else {
                S->look[*slot - 1].count += c;
            }
        }
    }
}

/* "AddWordCounts": append the new representative's (code, multiplicity) list. */
static int add_word_counts(state_t *S, uint32_t nw, uint32_t rep) {
    for (uint32_t j = 0; j < nw; ++j) {
        if (!S->word_mult[j]) continue;
        postlist_t *pl = &S->table[S->word_codes[j]];
        if (pl->size == pl->cap) {
            uint32_t cap = pl->cap ? pl->cap * 2 : 2;
            post_t *p = (post_t *)realloc(pl->items, (size_t)cap * sizeof(post_t));
            if (!p) return -1;
            pl->items = p; pl->cap = cap;
        }
        pl->items[pl->size].rep = rep;
        pl->items[pl->size].count = S->word_mult[j];
        pl->size++;
        S->st.rep_words++;
    }
    return 0;
}

/* code of the diagonal-test k-mer starting at seq[j] (2-mer base 21, or 4-mer base 4), -1 when a
 * nucleotide k-mer contains N; *cpx = 1 + number of adjacent unequal residues inside it */
static int kd_code(const state_t *S, const uint8_t *seq, int j, int *cpx) {
    int code = 0, c = 1;
    for (int t = 0; t < S->kd; ++t) {
        if (S->nt && seq[j + t] >= 4) return -1;
        code = code * S->base + seq[j + t];
        if (t) c += seq[j + t] != seq[j + t - 1];
    }
    if (cpx) *cpx = c;
    return code;
}

/* "ComputeAAP": positions of every diagonal-test k-mer of the query, bucketed by code. */
static void compute_aap(state_t *S, const uint8_t *seq, uint32_t len) {
    const int n2 = S->nkd, last = (int)len - S->kd;
    memset(S->taap, 0, n2 * sizeof(int32_t));
    for (int j = 0; j <= last; ++j) { const int c = kd_code(S, seq, j, NULL); if (c >= 0) S->taap[c]++; }
    int32_t mm = 0;
    for (int c = 0; c < n2; ++c) { S->aap_begin[c] = mm; mm += S->taap[c]; S->taap[c] = 0; }
    for (int j = 0; j <= last; ++j) {
        const int c = kd_code(S, seq, j, NULL);
        if (c >= 0) S->aap_list[S->aap_begin[c] + S->taap[c]++] = (int32_t)j;
    }
}

/* "diag_test_aapn" / "_est" (A.4 step 4): histogram of shared k-mers per diagonal, best window
 * of `band_width` diagonals, centre = best single diagonal, edges trimmed. */
static void diag_test(state_t *S, const uint8_t *seq2, int len1, int len2, int band_width,
                      int required_aa1, double cluster_thd, int *best_sum, int *band_left,
                      int *band_center, int *band_right) {
    const int nall = len1 + len2 - 1;
    int32_t *ds = S->diag_score, *ds2 = S->diag_score2;
    memset(ds, 0, (size_t)nall * sizeof(int32_t));
    memset(ds2, 0, (size_t)nall * sizeof(int32_t));
    int i1 = len1 - 1;
    for (int i = 0; i <= len2 - S->kd; ++i, ++i1) {
        int cpx;
        const int c22 = kd_code(S, seq2, i, &cpx);
        if (c22 < 0) continue;
        const int cnt = S->taap[c22];
        if (!cnt) continue;
        const int32_t *pos = S->aap_list + S->aap_begin[c22];
        for (int k = 0; k < cnt; ++k) { ds[i1 - pos[k]]++; ds2[i1 - pos[k]] += cpx; }
    }
    const int band_b = required_aa1 - 1 >= 0 ? required_aa1 - 1 : 0;
    const int band_e = nall - band_b;
    const int band_m = band_b + band_width - 1 < band_e ? band_b + band_width - 1 : band_e;
    int best_score = 0, best_score2 = 0, max_diag2 = 0, imax_diag = 0;
    for (int i = band_b; i <= band_m; ++i) {
        best_score += ds[i]; best_score2 += ds2[i];
        if (ds2[i] > max_diag2) { max_diag2 = ds2[i]; imax_diag = i; }
    }
    int from = band_b, end = band_m, score = best_score, score2 = best_score2;
    for (int k = from, j = band_m + 1; j < band_e; ++j, ++k) {
        score -= ds[k]; score += ds[j];
        score2 -= ds2[k]; score2 += ds2[j];
        if (score2 > best_score2) {
            from = k + 1; end = j; best_score = score; best_score2 = score2;
            if (ds2[j] > max_diag2) { max_diag2 = ds2[j]; imax_diag = j; }
        }
    }
    int mlen = imax_diag;
    if (imax_diag > len1) mlen = nall - imax_diag;
    const int emax = (int)((1.0 - cluster_thd) * mlen) + 1;
    for (int j = from; j < imax_diag; ++j) {
        if ((imax_diag - j) > emax || ds[j] < 1) { best_score -= ds[j]; from++; } else break;
    }
    for (int j = end; j > imax_diag; --j) {
        if ((j - imax_diag) > emax || ds[j] < 1) { best_score -= ds[j]; end--; } else break;
    }
    *band_left = from - len1 + 1;
    *band_right = end - len1 + 1;
    *band_center = imax_diag - len1 + 1;
    *best_sum = best_score;
}

/* "local_band_align" (A.4 step 5): banded global-style DP with one score matrix and a
 * back-pointer matrix; gap extension is recognised from the neighbour's back pointer;
 * end gaps cost the extension penalty; ties prefer diagonal, then left, then top.
 * Returns 0 and the number of identical aligned pairs on the traced-back path, or -1. */
static int band_align(state_t *S, const uint8_t *seq1, const uint8_t *seq2, int len1, int len2,
                      int band_left, int band_center, int band_right, int *iden_no) {
    *iden_no = 0;
    if (band_right >= len2 || band_left <= -len1 || band_left > band_right) return -1;
    const int bw = band_right - band_left + 1;
    const int bw1 = bw + 1;
    const size_t need = (size_t)(len1 + 1) * bw1;
    if (need > S->mat_cells) {
        free(S->score_mat); free(S->back_mat);
        S->score_mat = (int64_t *)malloc(need * sizeof(int64_t));
        S->back_mat = (uint8_t *)malloc(need);
        S->mat_cells = need;
        if (!S->score_mat || !S->back_mat) return -2;
    }
    int64_t *sm = S->score_mat; uint8_t *bm = S->back_mat;
#define SM(i, j1) sm[(size_t)(i) * bw1 + (j1)]
#define BM(i, j1) bm[(size_t)(i) * bw1 + (j1)]
    const int64_t gap = S->gap_open, ext = S->gap_ext;
    if (band_left < 0) { /* left border: leading query residues hang over */
        const int tband = band_right < 0 ? band_right : 0;
        for (int k = band_left; k <= tband; ++k) {
            const int i = -k, j1 = k - band_left;
            SM(i, j1) = ext * i; BM(i, j1) = BACK_TOP;
        }
        BM(-tband, tband - band_left) = BACK_NONE;
    }
    if (band_right >= 0) { /* top border: leading representative residues hang over */
        const int tband = band_left > 0 ? band_left : 0;
        for (int j = tband; j <= band_right; ++j) {
            const int j1 = j - band_left;
            SM(0, j1) = ext * j; BM(0, j1) = BACK_LEFT;
        }
        BM(0, tband - band_left) = BACK_NONE;
    }
    const int max_diag = band_center - band_left;
    static const int extra_score[4] = {4, 3, 2, 1};
    for (int i = 1; i <= len1; ++i) {
        int J0 = 1 - band_left - i, J1 = len2 - band_left - i;
        if (J0 < 0) J0 = 0;
        if (J1 >= bw) J1 = bw - 1; /* cd-hit also fills column bw, which nothing reads */
        const int ci = seq1[i - 1];
        for (int j1 = J0; j1 <= J1; ++j1) {
            const int j = j1 + i + band_left;
            int64_t sij = S->sub[ci][seq2[j - 1]];
            int d = j1 - max_diag; if (d < 0) d = -d;
            if (sij > 0) sij += extra_score[d & 3];
            int back = BACK_LEFT_TOP;
            int64_t best = SM(i - 1, j1) + sij;
            const int64_t gap0 = (i == len1 || j == len2) ? ext : gap;
            if (j1 > 0) {
                const int64_t g = BM(i, j1 - 1) == BACK_LEFT ? ext : gap0;
                const int64_t sc = SM(i, j1 - 1) + g;
                if (sc > best) { back = BACK_LEFT; best = sc; }
            }
            if (j1 + 1 < bw) {
                const int64_t g = BM(i - 1, j1 + 1) == BACK_TOP ? ext : gap0;
                const int64_t sc = SM(i - 1, j1 + 1) + g;
                if (sc > best) { back = BACK_TOP; best = sc; }
            }
            SM(i, j1) = best; BM(i, j1) = (uint8_t)back;
        }
    }
    int i, j;
    if (len2 - band_left < len1) { i = len2 - band_left; j = len2; }
    else if (len1 + band_right < len2) { i = len1; j = len1 + band_right; }
    else { i = len1; j = len2; }
    int j1 = j - i - band_left;
    int back = BM(i, j1), matches = 0;
    while (back != BACK_NONE) {
        if (back == BACK_TOP) { i -= 1; j1 += 1; }
        else if (back == BACK_LEFT) { j1 -= 1; j -= 1; }
        else { i -= 1; j -= 1; matches += seq1[i] == seq2[j]; }
        back = BM(i, j1);
    }
#undef SM
#undef BM
    *iden_no = matches; /* global identity: all matches on the path ("count3") */
    return 0;
}

/* One strand of "CheckOneAA" / "CheckOneEST": `seq` is the query as given or its reverse
 * complement. Returns 1 and fills rep/identity when it joins a representative. */
static int check_strand(state_t *S, const uint8_t *seq, int len, const pgx_cluster_params *P,
                        uint32_t *nw_out, uint32_t *hit_rep, float *hit_iden) {
    const int required_aa1 = (int)(P->identity * (double)len);
    int required_aas, required_aan;
    if (P->identity > 0.95) {
        required_aas = len - S->kd + 1 - (len - required_aa1) * S->kd;
        required_aan = len - S->word_len + 1 - (len - required_aa1) * S->word_len;
    } else {
        required_aas = (int)(P->aas_cutoff * (double)len);
        required_aan = (int)(P->aan_cutoff * (double)len);
    }
    const uint32_t nw = encode_words(S, seq, (uint32_t)len);
    *nw_out = nw;
    count_words(S, nw, required_aan);
    int has_aap = 0;
    for (uint32_t c = 0; c < S->n_look; ++c) {
        if ((int)S->look[c].count < required_aan) continue;
        S->st.filter_pairs++;
        const uint32_t rep = S->look[c].rep;
        const uint32_t r = S->rep_seq[rep];
        const uint8_t *seq2 = S->seq[r];
        const int len2 = (int)S->len[r];
        if (!has_aap) { compute_aap(S, seq, (uint32_t)len); has_aap = 1; }
        const int bw = P->band_width < len + len2 - 2 ? P->band_width : len + len2 - 2;
        int best_sum, band_left, band_center, band_right, iden;
        diag_test(S, seq2, len, len2, bw, required_aa1, P->identity, &best_sum, &band_left,
                  &band_center, &band_right);
        if (best_sum < required_aas) continue;
        const int rc = band_align(S, seq, seq2, len, len2, band_left, band_center, band_right, &iden);
        if (rc == -2) return -1;
        if (rc != 0) continue;
        S->st.aligned_pairs++;
        S->st.aligned_rep_len += (uint64_t)len2;
        S->st.dp_cells += (uint64_t)len * (uint64_t)(band_right - band_left + 1);
        if (iden < required_aa1) continue;
        const float pc = iden / (float)len;
        if (pc < P->identity) continue; /* float compared with the double threshold */
        *hit_rep = rep; *hit_iden = pc;
        return 1;
    }
    return 0;
}

/* forward strand first; for nucleotides with both_strands the reverse complement second */
static int check_one(state_t *S, uint32_t q, const pgx_cluster_params *P, uint32_t *nw_out,
                     uint32_t *hit_rep, float *hit_iden, int *hit_strand) {
    const uint8_t *seq = S->seq[q];
    const int len = (int)S->len[q];
    *hit_strand = 0;
    int hit = check_strand(S, seq, len, P, nw_out, hit_rep, hit_iden);
    if (hit != 0 || !S->nt || !P->both_strands) return hit;
    for (int i = 0; i < len; ++i) {
        const uint8_t b = seq[len - 1 - i];
        S->rc_buf[i] = b < 4 ? (uint8_t)(3 - b) : b;
    }
    uint32_t nw_rc;
    hit = check_strand(S, S->rc_buf, len, P, &nw_rc, hit_rep, hit_iden);
    if (hit == 1) { *hit_strand = 1; return 1; }
    if (hit < 0) return hit;
    *nw_out = encode_words(S, seq, (uint32_t)len); /* a new representative stores its forward words */
    return 0;
}

int pgxo_cluster_greedy(const uint8_t *residues, const uint64_t *offsets, uint32_t n,
                        const pgx_cluster_params *P, int32_t *out_cluster, int32_t *out_member,
                        float *out_identity, uint8_t *out_strand, uint32_t *out_n_clusters,
                        pgx_cluster_stats *stats) {
    if (!P || (P->alphabet != 0 && P->alphabet != 1)) return PGX_ERR_INVALID;
    if (P->word_len < 2 || P->word_len > (P->alphabet ? 11 : MAX_NAA)) return PGX_ERR_INVALID;
    state_t S; memset(&S, 0, sizeof(S));
    S.word_len = P->word_len;
    S.nt = P->alphabet == 1;
    S.base = S.nt ? 4 : NAA1;
    S.kd = S.nt ? 4 : 2;
    S.nkd = S.nt ? 256 : NAA1 * NAA1;
    S.gap_open = (int64_t)SCORE_SCALE * (S.nt ? -6 : GAP_OPEN);
    S.gap_ext = (int64_t)SCORE_SCALE * (S.nt ? -1 : GAP_EXT);
    S.st.n_input = n;
    int rc = PGX_ERR_NOMEM;

    /* A.2: letters -> indices; anything that is not a letter is dropped */
    const uint64_t total = n ? offsets[n] : 0;
    uint8_t *enc = (uint8_t *)malloc(total ? total : 1);
    uint64_t *eoff = (uint64_t *)malloc(((size_t)n + 1) * sizeof(uint64_t));
    uint32_t *order = (uint32_t *)malloc(((size_t)n ? n : 1) * sizeof(uint32_t));
    if (!enc || !eoff || !order) goto done0;
    {
        uint64_t w = 0; uint32_t max_len = 0;
        for (uint32_t i = 0; i < n; ++i) {
            eoff[i] = w;
            for (uint64_t p = offsets[i]; p < offsets[i + 1]; ++p) {
                uint8_t ch = residues[p];
                if (ch >= 'a' && ch <= 'z') ch -= 32;
                if (ch >= 'A' && ch <= 'Z')
                    enc[w++] = S.nt ? (uint8_t)(ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : (ch == 'T' || ch == 'U') ? 3 : 4)
                                    : (uint8_t)AA2IDX[ch - 'A'];
            }
            const uint64_t L = w - eoff[i];
            if (L > max_len) max_len = (uint32_t)L;
            if (out_cluster) out_cluster[i] = -1;
            if (out_member) out_member[i] = -1;
            if (out_identity) out_identity[i] = 0.0f;
            if (out_strand) out_strand[i] = 0;
        }
        eoff[n] = w;
        /* A.3: stable counting sort by descending length; length <= min_length discarded */
        uint32_t *cnt = (uint32_t *)calloc((size_t)max_len + 2, sizeof(uint32_t));
        if (!cnt) goto done0;
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t L = (uint32_t)(eoff[i + 1] - eoff[i]);
            if ((int)L > P->min_length) cnt[max_len - L + 1]++;
        }
        for (uint32_t l = 0; l <= max_len; ++l) cnt[l + 1] += cnt[l];
        S.n = cnt[max_len + 1];
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t L = (uint32_t)(eoff[i + 1] - eoff[i]);
            if ((int)L > P->min_length) order[cnt[max_len - L]++] = i;
        }
        free(cnt);
        S.seq = (const uint8_t **)malloc(((size_t)S.n ? S.n : 1) * sizeof(uint8_t *));
        S.len = (uint32_t *)malloc(((size_t)S.n ? S.n : 1) * sizeof(uint32_t));
        S.orig = order;
        if (!S.seq || !S.len) goto done;
        for (uint32_t k = 0; k < S.n; ++k) {
            S.seq[k] = enc + eoff[order[k]];
            S.len[k] = (uint32_t)(eoff[order[k] + 1] - eoff[order[k]]);
            if ((int)S.len[k] < S.word_len) { rc = PGX_ERR_INVALID; goto done; } /* needs min_length >= word_len-1 */
        }
        S.n_codes = 1;
        for (int t = 0; t < S.word_len; ++t) S.n_codes *= (uint32_t)S.base;
        S.table = (postlist_t *)calloc(S.n_codes, sizeof(postlist_t));
        S.rep_seq = (uint32_t *)malloc(((size_t)S.n ? S.n : 1) * sizeof(uint32_t));
        S.word_codes = (int32_t *)malloc(((size_t)max_len + 1) * sizeof(int32_t));
        S.word_mult = (uint32_t *)malloc(((size_t)max_len + 1) * sizeof(uint32_t));
        S.look = (post_t *)malloc(((size_t)S.n + 1) * sizeof(post_t));
        S.index_map = (uint32_t *)calloc((size_t)S.n + 1, sizeof(uint32_t));
        S.taap = (int32_t *)malloc(NAA1 * NAA1 * sizeof(int32_t));
        S.aap_begin = (int32_t *)malloc(NAA1 * NAA1 * sizeof(int32_t));
        S.aap_list = (int32_t *)malloc(((size_t)max_len + 1) * sizeof(int32_t));
        S.diag_score = (int32_t *)malloc(((size_t)max_len * 2 + 2) * sizeof(int32_t));
        S.diag_score2 = (int32_t *)malloc(((size_t)max_len * 2 + 2) * sizeof(int32_t));
        S.rc_buf = (uint8_t *)malloc((size_t)max_len + 1);
        if (!S.table || !S.rep_seq || !S.word_codes || !S.word_mult || !S.look || !S.index_map ||
            !S.taap || !S.aap_begin || !S.aap_list || !S.diag_score || !S.diag_score2 || !S.rc_buf) goto done;
    }
    {
        int k = 0;
        for (int i = 0; i < 23; ++i)
            for (int j = 0; j <= i; ++j)
                S.sub[i][j] = S.sub[j][i] = (int64_t)SCORE_SCALE * BLOSUM62_TRI[k++];
        if (S.nt) /* A.1: match +2, mismatch -2 over the indices 0..4 */
            for (int i = 0; i < 23; ++i)
                for (int j = 0; j < 23; ++j) S.sub[i][j] = (int64_t)SCORE_SCALE * (i == j ? 2 : -2);
    }
    uint32_t *member_count = (uint32_t *)calloc((size_t)S.n + 1, sizeof(uint32_t));
    if (!member_count) goto done;

    /* main greedy pass ("DoClustering" / "ClusterOne"). A.6, the memory-chunked rule: at a chunk boundary every
     * sequence not yet clustered is checked against the current table ("CheckOne": it joins the first
     * representative that accepts it and is out of the game), then the table is emptied. */
    uint8_t *taken = (uint8_t *)calloc((size_t)S.n + 1, 1);
    if (!taken) { free(member_count); goto done; }
    uint32_t next_bd = 0;
    for (uint32_t i = 0; i < P->n_chunk_boundaries; ++i) {
        const uint32_t b = P->chunk_boundaries[i];
        if (b == 0 || b >= S.n || (i && b <= P->chunk_boundaries[i - 1])) {
            rc = PGX_ERR_INVALID; free(member_count); free(taken); goto done;
        }
    }
    for (uint32_t q = 0; q < S.n; ++q) {
        uint32_t nw = 0, hit_rep = 0; float hit_iden = 0.0f; int hit_strand = 0;
        if (next_bd < P->n_chunk_boundaries && q == P->chunk_boundaries[next_bd]) {
            ++next_bd;
            for (uint32_t k = q; k < S.n; ++k) {
                if (taken[k]) continue;
                const int hit = check_one(&S, k, P, &nw, &hit_rep, &hit_iden, &hit_strand);
                if (hit < 0) { free(member_count); free(taken); goto done; }
                if (!hit) continue;
                taken[k] = 1;
                const uint32_t o = S.orig[k];
                if (out_cluster) out_cluster[o] = (int32_t)hit_rep;
                if (out_member) out_member[o] = (int32_t)member_count[hit_rep];
                if (out_identity) out_identity[o] = hit_iden;
                if (out_strand) out_strand[o] = (uint8_t)hit_strand;
                member_count[hit_rep]++;
            }
            for (uint32_t c = 0; c < S.n_codes; ++c) S.table[c].size = 0;   /* "word_table.Clear()" */
        }
        S.st.n_clustered++;
        S.st.sum_len_queries += S.len[q];
        if (taken[q]) continue;
        const int hit = check_one(&S, q, P, &nw, &hit_rep, &hit_iden, &hit_strand);
        if (hit < 0) { free(member_count); free(taken); goto done; }
        uint32_t cluster;
        if (hit) {
            cluster = hit_rep;
        } else { /* new representative: its own words were just encoded */
            cluster = S.n_reps;
            if (add_word_counts(&S, nw, S.n_reps) != 0) { free(member_count); goto done; }
            S.rep_seq[S.n_reps++] = q;
            S.st.sum_len_reps += S.len[q];
        }
        const uint32_t o = S.orig[q];
        if (out_cluster) out_cluster[o] = (int32_t)cluster;
        if (out_member) out_member[o] = (int32_t)member_count[cluster];
        if (out_identity) out_identity[o] = hit ? hit_iden : 0.0f;
        if (out_strand) out_strand[o] = (uint8_t)(hit ? hit_strand : 0);
        member_count[cluster]++;
    }
    free(member_count);
    free(taken);
    S.st.n_clusters = S.n_reps;
    if (out_n_clusters) *out_n_clusters = S.n_reps;
    if (stats) *stats = S.st;
    rc = PGX_OK;
done:
    if (S.table) { for (uint32_t c = 0; c < S.n_codes; ++c) free(S.table[c].items); }
    free(S.table); free(S.rep_seq); free(S.word_codes); free(S.word_mult); free(S.look);
    free(S.index_map); free(S.taap); free(S.aap_begin); free(S.aap_list); free(S.diag_score);
    free(S.diag_score2); free(S.rc_buf); free(S.score_mat); free(S.back_mat); free((void *)S.seq); free(S.len);
done0:
    free(enc); free(eoff); free(order);
    return rc;
}
