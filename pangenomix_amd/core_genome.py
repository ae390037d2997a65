"""Downstream consumers of the gene table on MI355X: how many genomes hold each gene,
and which genes are core. Mirrors of the reference's core_genome.count_gene_occurence (:127-155) and
find_core_genes (:107-124) -- same arguments, same DataFrames -- with the pandas groupby over the .npz
triples replaced by the row popcount of the device bitmap (libpgx, csrc/pancore.hip). The FASTA
extraction helpers of that module (Biopython) are out of scope."""
from __future__ import print_function

import numpy as np
import pandas as pd

from . import _native


def _row_occurrence(npz_file, index_name, ctx=None):
    """[index_name int32, count int64] for every row with at least one entry, ascending."""
    with np.load(npz_file) as data:
        rows, cols = data['row'], data['col']
        shape = tuple(int(x) for x in data['shape']) if 'shape' in data.files else None
    n_rows = shape[0] if shape else (int(rows.max()) + 1 if rows.size else 0)
    n_cols = shape[1] if shape else (int(cols.max()) + 1 if cols.size else 0)
    ctx = ctx or _native.default_context()
    counts, dup = ctx.row_counts(rows, cols, n_rows, n_cols)
    if dup:   # the reference counts triples, duplicates included; the bitmap counts genomes
        counts = np.bincount(rows, minlength=n_rows).astype(np.int32)
    present = np.flatnonzero(counts > 0)
    return pd.DataFrame({index_name: present.astype(rows.dtype), 'count': counts[present].astype(np.int64)})


def count_gene_occurence(gene_npz_file, ctx=None):
    """Occurrence of each gene over all genomes (reference core_genome.py:127-155)."""
    df = _row_occurrence(gene_npz_file, 'gene_index', ctx)
    print("\nCounted gene occurence")
    return df


def find_core_genes(gene_occurrence_count, genomes_num):
    """Genes present in at least `genomes_num` genomes (reference core_genome.py:107-124: columns
    gene_index / highest_expression; an empty frame without columns when there is none)."""
    hit = gene_occurrence_count[gene_occurrence_count['count'] >= genomes_num]
    if hit.empty:
        return pd.DataFrame([])
    return pd.DataFrame({'gene_index': hit['gene_index'].values.astype(np.int64),
                         'highest_expression': hit['count'].values.astype(np.int64)})
