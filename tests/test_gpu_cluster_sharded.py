"""Table-sharded multi-GPU clustering, rehearsed on ONE GPU: W virtual ranks run as
threads of this process, each with its own libpgx context, each streaming only its share of the
representatives in phase A; the per-sweep exchange of winner keys goes through the same C-ABI
callback the torch.distributed host uses, here backed by a thread barrier. The folded result must
equal the oracle bit for bit, counters included, and every rank must return the same clusters."""
import threading

import numpy as np
import pytest

import oracle
from pangenomix_amd import _native, cluster, synth
from test_cluster_oracle import params
from test_gpu_cluster import assert_same, assert_same_nt, nt_params

pytestmark = pytest.mark.gpu


def run_virtual_ranks(res, off, p, world):
    import torch
    torch.cuda.init()
    dev = torch.device('cuda', 0)
    keys = [torch.empty(cluster.EXCHANGE_KEYS, dtype=torch.int64, device=dev) for _ in range(world)]
    barrier = threading.Barrier(world)
    results, errors = [None] * world, []

    def all_reduce_min_for(rank):
        def f(t):                                   # t = keys[rank][:n], already bit-flipped
            n = t.numel()
            torch.cuda.synchronize()
            barrier.wait()
            m = keys[0][:n]
            for k in keys[1:]:
                m = torch.minimum(m, k[:n])
            torch.cuda.synchronize()
            barrier.wait()                          # every rank has read all inputs
            t.copy_(m)
            torch.cuda.synchronize()
        return f

    def worker(rank):
        ctx = _native.Context(0)
        try:
            sp, keep = cluster.shard_params(p, rank, world, keys[rank], all_reduce_min_for(rank))
            results[rank] = ctx.cluster_greedy(res, off, sp)
            del keep
        except Exception as exc:                    # a failing rank must not leave the others at the barrier
            errors.append((rank, exc))
            barrier.abort()
        finally:
            ctx.close()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    return results


def fold(results):
    """What merge_shard_results does over a process group, on the collected per-rank results."""
    ident = np.maximum.reduce([r[2] for r in results])
    strand = np.maximum.reduce([r[3] for r in results])
    vecs = []
    for r in results:
        st = r[5]
        vecs.append(np.array([st[k] for k in cluster.PARTIAL_STATS] +
                             [st['gpu'][k] for k in ('pairs', 'aligned', 'aligned_bytes', 'table_stream_words')],
                             dtype=np.int64))
    summed = np.sum(vecs, axis=0)
    return cluster.merge_shard_results(results[0], lambda a: summed,
                                       lambda a: ident if a.dtype == np.float32 else strand)


@pytest.mark.parametrize('world', [2, 3])
def test_protein_virtual_ranks_match_oracle(world):
    ps = synth.ProteinSet(30, 500, 800, 150, 77)            # > 2 sweeps: later sweeps meet a sharded table
    res, off, _ = ps.nr_arrays()
    p = params()
    results = run_virtual_ranks(res, off, p, world)
    for r in results[1:]:                                   # replicated outputs agree on every rank
        assert np.array_equal(r[0], results[0][0]) and np.array_equal(r[1], results[0][1]) and r[4] == results[0][4]
    # phase A really was split: no rank streamed the whole table
    words = [r[5]['gpu']['table_stream_words'] for r in results]
    single = _native.Context(0)
    try:
        whole = single.cluster_greedy(res, off, p)
    finally:
        single.close()
    assert sum(words) == whole[5]['gpu']['table_stream_words'] and max(words) < 0.7 * sum(words)
    assert_same(fold(results), oracle.cluster_greedy(res, off, p))


def test_nucleotide_both_strands_virtual_ranks_match_oracle():
    res, off, _ = synth.noncoding_set(n_genomes=120, seed=9)    # > 2 sweeps of 2048 queries
    p = nt_params()
    results = run_virtual_ranks(res, off, p, 2)
    assert_same_nt(fold(results), oracle.cluster_greedy(res, off, p))


def test_single_rank_group_is_the_plain_path(gpu_ctx):
    """world = 1 through the exchange callback (identity exchange): same result as without it."""
    import torch
    res, off, _ = synth.protein_set('small').nr_arrays()
    p = params()
    keys = torch.empty(cluster.EXCHANGE_KEYS, dtype=torch.int64, device='cuda:0')
    calls = []
    sp, keep = cluster.shard_params(p, 0, 1, keys, lambda t: calls.append(t.numel()))
    got = gpu_ctx.cluster_greedy(res, off, sp)
    assert calls and all(c == cluster.EXCHANGE_KEYS for c in calls) and len(calls) == got[5]['sweeps']
    assert_same(got, oracle.cluster_greedy(res, off, p))


def test_bad_shard_arguments_are_rejected(gpu_ctx):
    res, off, _ = synth.protein_set('tiny').nr_arrays()
    p = params()
    p.shard_count, p.shard_index = 2, 0                      # no exchange callback
    with pytest.raises(_native.PgxError):
        gpu_ctx.cluster_greedy(res, off, p)
    p.shard_count, p.shard_index = 2, 2
    with pytest.raises(_native.PgxError):
        gpu_ctx.cluster_greedy(res, off, p)
