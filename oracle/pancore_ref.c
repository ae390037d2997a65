/*
 * oracle/pancore_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's pan/core rarefaction loop,
 * /root/reference/pangenomix/pangenome_analysis.py:72-98, kept in the reference's own
 * (dense incidence count) form so that it is an independent check of the bit-packed HIP
 * kernel:
 *
 *   :74-75  gene_data = df_genes.data.T.tocsr()           -> CSR genome -> genes below
 *   :86     gene_incidence = zeros(num_genes, int)
 *   :88     gene_incidence += gene_data[shuffle_col,:]    (duplicates in COO sum up, as in scipy)
 *   :89     pan[i,j]  = (gene_incidence > 0).sum()
 *   :90     core[i,j] = (gene_incidence == j+1).sum()
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * Pinned against tests/golden/pancore (npz files), which were produced by running the reference
 * function itself (tests/golden/make_golden.py).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* COO (row = gene, col = genome, val) -> pan/core tables. perms: [n_iter][n_genomes].
 * Returns 0, or -1 on allocation failure / bad index. */
int pgxo_pan_core(const int32_t *row, const int32_t *col, const int64_t *val, uint64_t nnz,
                  uint32_t n_genes, uint32_t n_genomes, const int32_t *perms, uint32_t n_iter,
                  int64_t *out_pan, int64_t *out_core) {
    /* CSR by genome (the .T.tocsr() of :75) */
    uint64_t *ptr = (uint64_t *)calloc((size_t)n_genomes + 1, sizeof(uint64_t));
    int32_t *idx = (int32_t *)malloc((nnz ? nnz : 1) * sizeof(int32_t));
    int64_t *dat = (int64_t *)malloc((nnz ? nnz : 1) * sizeof(int64_t));
    int64_t *inc = (int64_t *)malloc(((size_t)n_genes ? n_genes : 1) * sizeof(int64_t));
    if (!ptr || !idx || !dat || !inc) { free(ptr); free(idx); free(dat); free(inc); return -1; }
    for (uint64_t k = 0; k < nnz; ++k) {
        if ((uint32_t)row[k] >= n_genes || (uint32_t)col[k] >= n_genomes) {
            free(ptr); free(idx); free(dat); free(inc); return -1;
        }
        ptr[col[k] + 1]++;
    }
    for (uint32_t s = 0; s < n_genomes; ++s) ptr[s + 1] += ptr[s];
    uint64_t *fill = (uint64_t *)malloc(((size_t)n_genomes ? n_genomes : 1) * sizeof(uint64_t));
    if (!fill) { free(ptr); free(idx); free(dat); free(inc); return -1; }
    memcpy(fill, ptr, (size_t)n_genomes * sizeof(uint64_t));
    for (uint64_t k = 0; k < nnz; ++k) {
        uint64_t p = fill[col[k]]++;
        idx[p] = row[k];
        dat[p] = val ? val[k] : 1;
    }
    free(fill);

    for (uint32_t i = 0; i < n_iter; ++i) {
        memset(inc, 0, (size_t)n_genes * sizeof(int64_t));
        const int32_t *perm = perms + (size_t)i * n_genomes;
        for (uint32_t j = 0; j < n_genomes; ++j) {
            const uint32_t s = (uint32_t)perm[j];
            for (uint64_t p = ptr[s]; p < ptr[s + 1]; ++p) inc[idx[p]] += dat[p];
            int64_t pan = 0, core = 0;
            for (uint32_t g = 0; g < n_genes; ++g) {
                pan += inc[g] > 0;
                core += inc[g] == (int64_t)j + 1;
            }
            out_pan[(size_t)i * n_genomes + j] = pan;
            out_core[(size_t)i * n_genomes + j] = core;
        }
    }
    free(ptr); free(idx); free(dat); free(inc);
    return 0;
}
