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
    """(row, col) int32 of the stored entries. The OR/AND restatement is exact only for a 0/1
    matrix without duplicate coordinates, which is what
    build_genetic_feature_tables emits; anything else is rejected rather than silently computed
    differently from the reference. Values are checked here; duplicate coordinates are detected
    on the device while the bitmap is built (libpgx counts the bits that were already set)."""
    coo = df_genes.data if df_genes.data.format == 'coo' else df_genes.data.tocoo()
    data = np.asarray(coo.data)
    if data.size and not np.all(data == 1):
        raise ValueError('estimate_pan_core_size needs a binary (0/1) gene x genome table')
    return np.asarray(coo.row, dtype=np.int32), np.asarray(coo.col, dtype=np.int32)


_DUPLICATES = 'estimate_pan_core_size needs a table without duplicate entries'


def draw_permutations(num_strains, num_iter):
    """One legacy-RNG shuffle of arange(S) per iteration, in iteration order
    (reference :84-85). Kept on the host so seeded runs match the reference."""
    perms = np.empty((num_iter, num_strains), dtype=np.int32)
    for i in range(num_iter):
        p = np.arange(num_strains)
        np.random.shuffle(p)
        perms[i] = p
    return perms


def shard_bounds(num_iter, rank, world):
    """Contiguous slice [lo, hi) of the iterations that `rank` of `world` computes.
    Iterations are independent given their permutation: no data-path collective."""
    base, extra = divmod(num_iter, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _pan_core_sharded(ctx, bits, num_genes, perms, group, compute=None):
    """Iterations sharded over the ranks of `group`; results gathered on every rank. The
    bitmap is replicated (<= 10 MB at 400 genomes). `compute(bits, n_genes, perms)` defaults
    to the GPU kernel; tests inject the oracle to rehearse the sharding on CPU (gloo)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n_iter, S = perms.shape
    lo, hi = shard_bounds(n_iter, rank, world)
    compute = compute or ctx.pan_core
    if hi > lo:
        pan, core = compute(bits, num_genes, perms[lo:hi])
    else:
        pan = core = np.zeros((0, S), dtype=np.int32)
    width = (n_iter + world - 1) // world                      # equal-sized slots for all_gather
    mine = np.zeros((2, width, S), dtype=np.int32)
    mine[0, :hi - lo], mine[1, :hi - lo] = pan, core
    backend = dist.get_backend(group)
    dev = torch.device('cuda', ctx.device_info()['device_id']) if backend == 'nccl' else torch.device('cpu')
    t = torch.from_numpy(mine).to(dev)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    pan_all = np.empty((n_iter, S), dtype=np.int32)
    core_all = np.empty((n_iter, S), dtype=np.int32)
    for r in range(world):
        a, b = shard_bounds(n_iter, r, world)
        part = out[r].cpu().numpy()
        pan_all[a:b], core_all[a:b] = part[0, :b - a], part[1, :b - a]
    return pan_all, core_all


def estimate_pan_core_size(df_genes, num_iter, log_batch=-1, ctx=None, group=None):
    """Pan/core genome size curves for `num_iter` random genome orders.

    df_genes : LightSparseDataFrame, binary gene x genome table
    num_iter : number of randomisations
    log_batch: accepted for compatibility (the GPU computes all iterations in one
               launch; a line is printed per batch boundary as the reference does)
    ctx      : optional pangenomix_amd._native.Context (default: process-wide)
    group    : optional torch.distributed process group (one process per GPU). Every rank
               draws the SAME permutations (same RNG state required, e.g. np.random.seed(k)
               on every rank), computes its contiguous share of the iterations on its own
               GPU, and the shares are all-gathered; every rank returns the full table.
    """
    num_genes, num_strains = df_genes.shape
    print('Converting DataFrame to matrix...')
    row, col = _binary_coo(df_genes)
    ctx = ctx or _native.default_context()

    print('Generating pan/core curves from shuffled strains')
    perms = draw_permutations(num_strains, num_iter)
    if log_batch > 0:
        for i in range(log_batch, num_iter + 1, log_batch):
            print('\tIteration', i, 'of', num_iter)
    if group is not None and num_iter > 0 and num_strains > 0:
        bits, dup = ctx.presence_bitmap(row, col, num_genes, num_strains, return_duplicates=True)
        if dup:
            raise ValueError(_DUPLICATES)
        pan, core = _pan_core_sharded(ctx, bits, num_genes, perms, group)
    else:   # one call: coordinates and permutations up, bitmap built and consumed on the device, curves down
        pan, core, dup = ctx.pan_core_coo(row, col, num_genes, num_strains, perms)
        if dup:
            raise ValueError(_DUPLICATES)

    iter_index = ['Iter' + str(x) for x in range(1, num_iter + 1)]
    pan_cols = ['Pan' + str(x) for x in range(1, num_strains + 1)]
    core_cols = ['Core' + str(x) for x in range(1, num_strains + 1)]
    return pd.DataFrame(index=iter_index, columns=pan_cols + core_cols,
                        data=np.hstack([pan, core]).astype(np.float64))
