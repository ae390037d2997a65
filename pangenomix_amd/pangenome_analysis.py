"""
Pan/core genome rarefaction curves on MI355X.

Drop-in mirror of `estimate_pan_core_size` in the reference
(pangenomix/pangenome_analysis.py:51-98): same signature, same DataFrame out
(index Iter1..IterN, columns Pan1..PanS + Core1..CoreS, float64), and the same
consumption of the global legacy numpy RNG -- exactly one
`np.random.shuffle(np.arange(S))` per iteration (:84-85) -- so a seeded run gives
the table the reference gives. The Python double loop over CSR rows (:81-90) is
replaced by libpgx: presence bitmap -> bit-packed permuted cumulative OR/AND +
popcount (pangenomix_amd/csrc/pancore.hip).
"""

from __future__ import print_function

import numpy as np
import pandas as pd

from . import _native


def _binary_coo(df_genes):
    """(row, col) of the stored entries. The OR/AND restatement is exact only for a
    0/1 matrix without duplicate coordinates, which is what
    build_genetic_feature_tables emits; anything else is rejected rather than
    silently computed differently from the reference."""
    coo = df_genes.data.tocoo()
    data = np.asarray(coo.data)
    if data.size and not np.all(data == 1):
        raise ValueError('estimate_pan_core_size needs a binary (0/1) gene x genome table')
    row = np.asarray(coo.row, dtype=np.int64)
    col = np.asarray(coo.col, dtype=np.int64)
    if row.size:
        flat = row * coo.shape[1] + col
        if np.unique(flat).size != flat.size:
            raise ValueError('estimate_pan_core_size needs a table without duplicate entries')
    return row.astype(np.int32), col.astype(np.int32)


def draw_permutations(num_strains, num_iter):
    """One legacy-RNG shuffle of arange(S) per iteration, in iteration order
    (reference :84-85). Kept on the host so seeded runs match the reference."""
    perms = np.empty((num_iter, num_strains), dtype=np.int32)
    for i in range(num_iter):
        p = np.arange(num_strains)
        np.random.shuffle(p)
        perms[i] = p
    return perms


def estimate_pan_core_size(df_genes, num_iter, log_batch=-1, ctx=None):
    """Pan/core genome size curves for `num_iter` random genome orders.

    df_genes : LightSparseDataFrame, binary gene x genome table
    num_iter : number of randomisations
    log_batch: accepted for compatibility (the GPU computes all iterations in one
               launch; a line is printed per batch boundary as the reference does)
    ctx      : optional pangenomix_amd._native.Context (default: process-wide)
    """
    num_genes, num_strains = df_genes.shape
    print('Converting DataFrame to matrix...')
    row, col = _binary_coo(df_genes)
    ctx = ctx or _native.default_context()
    bits = ctx.presence_bitmap(row, col, num_genes, num_strains)

    print('Generating pan/core curves from shuffled strains')
    perms = draw_permutations(num_strains, num_iter)
    if log_batch > 0:
        for i in range(log_batch, num_iter + 1, log_batch):
            print('\tIteration', i, 'of', num_iter)
    if num_iter > 0 and num_strains > 0:
        pan, core = ctx.pan_core(bits, num_genes, perms)
    else:
        pan = np.zeros((num_iter, num_strains), dtype=np.int32)
        core = np.zeros((num_iter, num_strains), dtype=np.int32)

    iter_index = ['Iter' + str(x) for x in range(1, num_iter + 1)]
    pan_cols = ['Pan' + str(x) for x in range(1, num_strains + 1)]
    core_cols = ['Core' + str(x) for x in range(1, num_strains + 1)]
    return pd.DataFrame(index=iter_index, columns=pan_cols + core_cols,
                        data=np.hstack([pan, core]).astype(np.float64))
