"""Multi-process paths, rehearsed with 2 and 3 gloo ranks on CPU, the oracle standing in
for the kernels; on GPUs the same host code runs over RCCL. Pan/core: iterations sharded, no data-path
collective, shares all-gathered. Clustering: records sharded, best keys all-gathered between the steps
of a window (tests/sharded_model.py restates the protocol the HIP library runs)."""
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


# ---- clustering, record-sharded mode: the partition protocol over real collectives ----------
def _model_inputs():
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from test_cluster_oracle import mutate, params, rand_seq
    rng = np.random.default_rng(3)
    fams = [rand_seq(rng, int(n)) for n in rng.integers(40, 200, 14)]
    seqs = [mutate(rng, fams[i % 14], int(rng.integers(0, len(fams[i % 14]) * 0.3))) for i in range(120)]
    return seqs + ['MKV', 'ACDEFGHIKLM'], params()


def _cluster_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import torch
    import oracle
    import sharded_model
    from pangenomix_amd import cluster
    from test_cluster_oracle import pack
    seqs, p = _model_inputs()
    # (1) the window protocol with this rank evaluating only its own members; the exchange is the
    #     product's all-gather wrapper over the gloo group, the fold and the partition rule are the product's
    got, evaluated = sharded_model.run(seqs, p, oracle, pack, rank, world, cluster.group_all_gather(dist.group.WORLD),
                                       cluster.fold_best_keys, cluster.owner_of, window=16)
    # (2) the C-ABI callback as the library calls it: enqueue-only all-gather of the send buffer
    send, recv = cluster.exchange_buffers(world, 'cpu')
    assert tuple(send.shape) == (2, cluster.EXCHANGE_KEYS + 8)         # two windows in flight; keys + the error word
    send[1].fill_(-1)
    send[1, rank::world] = 1000 + rank                                # this rank's members (slot 1)
    send[1, 7] = -(2 ** 63) + rank                                    # top bit set: unsigned order matters
    send[1, cluster.EXCHANGE_KEYS] = 16 if rank == world - 1 else 0   # the last rank reports a failure
    sp, keep = cluster.shard_params(p, rank, world, send, recv, cluster.group_all_gather(dist.group.WORLD))
    assert (sp.shard_index, sp.shard_count) == (rank, world) and sp.identity == p.identity
    assert sp.exchange(None, None, 1) == 0
    assert not recv[0].any()                                          # the other slot is untouched
    folded = cluster.fold_best_keys(recv[1].numpy()[:, :cluster.EXCHANGE_KEYS])
    assert int(np.bitwise_or.reduce(recv[1].numpy()[:, cluster.EXCHANGE_KEYS])) == 16   # every rank sees the failure

    # (3) the fold of the per-rank partial outputs
    def host_reduce(op):
        def f(a):
            t = torch.from_numpy(np.ascontiguousarray(a).astype(np.int64 if a.dtype != np.float32 else np.float32))
            dist.all_reduce(t, op=op)
            return t.numpy().astype(a.dtype)
        return f
    n = 10
    ident = np.zeros(n, dtype=np.float32); ident[rank::world] = 0.8 + 0.01 * rank   # known to one rank each
    stats = {k: 10 * (rank + 1) for k in cluster.PARTIAL_STATS}
    stats.update(n_input=n, n_clusters=3, gpu={k: rank + 1 for k in cluster.GPU_STATS})
    merged = cluster.merge_shard_results((np.arange(n), np.arange(n), ident, np.zeros(n, np.uint8), 3, stats),
                                         host_reduce(dist.ReduceOp.SUM), host_reduce(dist.ReduceOp.MAX))
    np.savez(os.path.join(out_dir, 'c%d.npz' % rank), clusters=got, evaluated=evaluated, folded=folded,
             ident=merged[2],
             stats=np.array([merged[5][k] for k in cluster.PARTIAL_STATS] + [merged[5]['gpu']['pairs'], merged[5]['n_input']]))
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_record_sharded_protocol_over_gloo(world, tmp_path):
    port = 31500 + os.getpid() % 2000 + world
    mp.spawn(_cluster_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    outs = [np.load(str(tmp_path / ('c%d.npz' % r))) for r in range(world)]
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import oracle
    from test_cluster_oracle import pack
    seqs, p = _model_inputs()
    res, off = pack(seqs)
    want = oracle.cluster_greedy(res, off, p)
    evaluated = [int(o['evaluated']) for o in outs]
    for o in outs:
        assert np.array_equal(o['clusters'], want[0])             # every rank: the sequential result
        # the gathered keys folded by unsigned minimum: each member's key comes from its owner
        f = o['folded']
        assert all(int(f[i]) == 1000 + i % world for i in range(20) if i != 7) and int(f[7]) == 2 ** 63
        assert np.array_equal(o['ident'], outs[0]['ident']) and (o['ident'] > 0.79).all()
        tri = 10 * world * (world + 1) // 2
        assert o['stats'].tolist() == [tri] * 5 + [world * (world + 1) // 2, 10]   # partial counters add, replicated stay
    assert max(evaluated) < 0.8 * sum(evaluated)                  # the pair work really was split over the ranks


# ---- rank 0 drives the reference-style entry point, the other ranks serve its clustering calls -----
def _serve_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import torch
    import oracle
    from pangenomix_amd import cluster
    calls = []

    def fake_cluster_sequences(residues, offsets, params, ctx=None, group=None):
        # the oracle in place of the HIP library; one collective proves every rank takes part in the call
        t = torch.ones(1)
        dist.all_reduce(t, group=group)
        calls.append((int(t.item()), int(offsets.size - 1)))
        return oracle.cluster_greedy(residues, offsets, params)
    cluster.cluster_sequences = fake_cluster_sequences
    cluster.set_process_group(dist.group.WORLD)
    fasta = os.path.join(out_dir, 'in.faa')
    if rank == 0:
        rng = np.random.default_rng(2)
        aa = np.array(list('ACDEFGHIKLMNPQRSTVWY'))
        a = ''.join(rng.choice(aa, 150))
        with open(fasta, 'w') as f:
            f.write('>s1 first\n%s\n>s2\n%s\n>s3\n%s\n' % (a, a[:140], ''.join(rng.choice(aa, 90))))
        for out in ('o1', 'o2'):
            cluster.cluster_fasta_to_clstr(fasta, os.path.join(out_dir, out), {'-n': 5, '-c': 0.8})
        cluster.stop_workers()
        served = -1
    else:
        served = cluster.serve()
    np.save(os.path.join(out_dir, 's%d.npy' % rank), np.array([served, len(calls)] + [c[0] for c in calls]))
    dist.destroy_process_group()


def test_rank0_drives_and_workers_serve(tmp_path):
    world, port = 3, 33500 + os.getpid() % 2000
    mp.spawn(_serve_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    outs = [np.load(str(tmp_path / ('s%d.npy' % r))) for r in range(world)]
    assert outs[0].tolist() == [-1, 2, world, world]            # rank 0 made two calls, each with all ranks in
    for o in outs[1:]:
        assert o.tolist() == [2, 2, world, world]               # every worker served both, then was released
    clstr = open(str(tmp_path / 'o1.clstr')).read()
    assert clstr.count('>Cluster') == 2 and 'at 100.00%' in clstr
    assert open(str(tmp_path / 'o2.clstr')).read() == clstr
