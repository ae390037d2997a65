"""Multi-process path: pan/core iterations are sharded over ranks with no
data-path collective; the shares are all-gathered. Rehearsed here with 2 (and 3) gloo ranks
on CPU, the oracle standing in for the kernel; on GPUs the same code runs over RCCL."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import oracle
    from pangenomix_amd import pangenome_analysis as pa
    rng = np.random.default_rng(5)
    G, S, n_iter = 700, 23, 11
    dense = rng.random((G, S)) < 0.3
    row, col = (a.astype(np.int32) for a in np.nonzero(dense))
    np.random.seed(3)                               # same stream on every rank
    perms = pa.draw_permutations(S, n_iter)

    def compute(bits, n_genes, p):                  # the oracle in place of the HIP kernel
        pan, core = oracle.pan_core(row, col, None, n_genes, S, p)
        return pan.astype(np.int32), core.astype(np.int32)
    pan, core = pa._pan_core_sharded(None, None, G, perms, dist.group.WORLD, compute=compute)
    np.savez(os.path.join(out_dir, 'r%d.npz' % rank), pan=pan, core=core, perms=perms)
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_pan_core_equals_single_process(world, tmp_path):
    port = 29500 + os.getpid() % 2000 + world
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    import oracle
    outs = [np.load(str(tmp_path / ('r%d.npz' % r))) for r in range(world)]
    rng = np.random.default_rng(5)
    dense = rng.random((700, 23)) < 0.3
    row, col = (a.astype(np.int32) for a in np.nonzero(dense))
    pan, core = oracle.pan_core(row, col, None, 700, 23, outs[0]['perms'])
    for o in outs:                                   # every rank holds the full, identical table
        assert np.array_equal(o['pan'], pan) and np.array_equal(o['core'], core)


def test_shard_bounds_cover_everything():
    from pangenomix_amd.pangenome_analysis import shard_bounds
    for n in (0, 1, 7, 1000):
        for w in (1, 2, 3, 8):
            cuts = [shard_bounds(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in cuts) - min(b - a for a, b in cuts) <= 1
