"""Record-sharded multi-GPU clustering, rehearsed on ONE GPU: W virtual ranks run as
threads of this process, each with its own libpgx context and its own replica of the representative
index; window member i is filtered and aligned by rank i % W only, and the members' best keys are
all-gathered after every evaluation through the same C-ABI callback the torch.distributed host uses,
here backed by a thread barrier and device copies. The folded result must equal the oracle bit for
bit, counters included, and every rank must return the same clusters."""
import threading

import numpy as np
import pytest

import oracle
from pangenomix_amd import _native, cluster, synth
from test_cluster_oracle import params
from test_gpu_cluster import assert_same, assert_same_nt, nt_params

pytestmark = pytest.mark.gpu


def run_virtual_ranks(res, off, p, world, expect_errors=False, want_stats=True):
    import torch
    torch.cuda.init()
    dev = torch.device('cuda', 0)
    bufs = [cluster.exchange_buffers(world, dev) for _ in range(world)]
    sends, recvs = [b[0] for b in bufs], [b[1] for b in bufs]
    barrier = threading.Barrier(world)
    results, errors = [None] * world, []

    def all_gather(recv, send, stream):             # (recv / send: one slot of this rank's buffers)
        slot = next(s_ for s_ in range(cluster.EXCHANGE_SLOTS) if any(send.data_ptr() == b[s_].data_ptr() for b in sends))
        torch.cuda.synchronize()                    # this rank's keys are complete
        barrier.wait()
        for r in range(world):
            recv[r].copy_(sends[r][slot])
        torch.cuda.synchronize()
        barrier.wait()                              # every rank has read all contributions

    def worker(rank):
        ctx = _native.Context(0)
        try:
            sp, keep = cluster.shard_params(p, rank, world, sends[rank], recvs[rank], all_gather)
            results[rank] = ctx.cluster_greedy(res, off, sp, want_stats=want_stats)
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
    if expect_errors:
        return errors
    assert not errors, errors
    return results


def fold(results):
    """What merge_shard_results does over a process group, on the collected per-rank results."""
    ident = np.maximum.reduce([r[2] for r in results])
    vecs = []
    for r in results:
        st = r[5]
        vecs.append(np.array([st[k] for k in cluster.PARTIAL_STATS] + [st['gpu'][k] for k in cluster.GPU_STATS],
                             dtype=np.int64))
    summed = np.sum(vecs, axis=0)
    return cluster.merge_shard_results(results[0], lambda a: summed, lambda a: ident)


def assert_replicated(results):
    for r in results[1:]:                                   # replicated outputs agree on every rank
        assert np.array_equal(r[0], results[0][0]) and np.array_equal(r[1], results[0][1]) and r[4] == results[0][4]
        assert np.array_equal(r[3], results[0][3])


@pytest.mark.parametrize('world', [2, 3])
def test_protein_virtual_ranks_match_oracle(world, monkeypatch):
    monkeypatch.setenv('PGX_WINDOW', '4096')                # several windows: later ones meet an index built by all ranks
    ps = synth.ProteinSet(30, 500, 800, 150, 77)
    res, off, _ = ps.nr_arrays()
    p = params()
    results = run_virtual_ranks(res, off, p, world)
    assert results[0][5]['sweeps'] >= 3
    assert_replicated(results)
    want = oracle.cluster_greedy(res, off, p)
    # the filter work really was split by record: the ranks' posting visits add up to the sequential count
    visits = [r[5]['posting_visits'] for r in results]
    assert sum(visits) == want[5]['posting_visits'] and max(visits) < 0.7 * sum(visits)
    assert_same(fold(results), want)


def test_nucleotide_both_strands_virtual_ranks_match_oracle():
    res, off, _ = synth.noncoding_set(n_genomes=120, seed=9)    # > 2 windows of 2048 queries
    p = nt_params()
    results = run_virtual_ranks(res, off, p, 2)
    assert_replicated(results)
    assert_same_nt(fold(results), oracle.cluster_greedy(res, off, p))


@pytest.mark.slow
@pytest.mark.parametrize('world', [2, 3])
def test_cfg3s_virtual_ranks_match_single_process(world, gpu_ctx):
    """The benchmark workload (1.14 M proteins) split over 2 and 3 virtual ranks: clusters, members,
    identities and every counter equal the single-process result (which test_cfg3s_full_size_parity
    pins to the oracle)."""
    res, off, _ = synth.protein_set('cfg-3s').nr_arrays()
    p = params()
    whole = gpu_ctx.cluster_greedy(res, off, p)
    results = run_virtual_ranks(res, off, p, world)
    assert_replicated(results)
    assert_same(fold(results), whole)


def test_single_rank_group_is_the_plain_path(gpu_ctx):
    """world = 1 through the exchange callback (identity all-gather): same result as without it."""
    import torch
    res, off, _ = synth.protein_set('small').nr_arrays()
    p = params()
    send, recv = cluster.exchange_buffers(1, 'cuda:0')
    calls = []

    def all_gather(r, s_, stream):
        calls.append(stream)
        with torch.cuda.stream(torch.cuda.ExternalStream(stream)):
            r[0].copy_(s_)
    sp, keep = cluster.shard_params(p, 0, 1, send, recv, all_gather)
    got = gpu_ctx.cluster_greedy(res, off, sp)
    assert len(calls) >= got[5]['sweeps'] and all(calls)
    assert_same(got, oracle.cluster_greedy(res, off, p))


def test_single_rank_native_rccl_exchange_is_the_plain_path():
    """world = 1 with the library's own communicator (pgx_rccl_*): ncclAllGather enqueued by libpgx on the window's
    stream instead of the callback. Same result as the oracle; a second call reuses the communicator; a context whose
    communicator has another rank / size than the parameters is refused. (More than one rank needs one GPU each:
    RCCL refuses two ranks on one card, so the multi-rank logic is covered by the virtual ranks above and the
    collective itself here.)"""
    res, off, _ = synth.protein_set('small').nr_arrays()
    p = params()
    want = oracle.cluster_greedy(res, off, p)
    with _native.Context(0) as ctx:
        ctx.comm_create(_native.rccl_unique_id(), 0, 1)
        sp = cluster.native_shard_params(p, 0, 1)
        assert not sp.exchange and sp.shard_count == 1
        for _ in range(2):
            assert_same(ctx.cluster_greedy(res, off, sp), want)
        bad = cluster.native_shard_params(p, 0, 2)
        with pytest.raises(_native.PgxError, match='communicator'):
            ctx.cluster_greedy(res, off, bad)
        with pytest.raises(_native.PgxError, match='already'):
            ctx.comm_create(_native.rccl_unique_id(), 0, 1)
        ctx.comm_destroy()
        assert_same(ctx.cluster_greedy(res, off, p), want)      # and the context is an ordinary one again


def test_bad_shard_arguments_are_rejected(gpu_ctx):
    res, off, _ = synth.protein_set('tiny').nr_arrays()
    p = params()
    p.shard_count, p.shard_index = 2, 0                      # no exchange callback
    with pytest.raises(_native.PgxError):
        gpu_ctx.cluster_greedy(res, off, p)
    p.shard_count, p.shard_index = 2, 2
    with pytest.raises(_native.PgxError):
        gpu_ctx.cluster_greedy(res, off, p)


@pytest.mark.parametrize('seed', range(6))
def test_randomized_sets_on_virtual_ranks_match_oracle(seed):
    """The randomized sets of test_gpu_cluster.py, split over 2 or 3 virtual ranks with small windows."""
    from test_cluster_oracle import AA, pack
    from test_gpu_cluster import _random_families
    rng = np.random.default_rng(3000 + seed)
    nucleotide = seed % 3 == 2
    if nucleotide:
        seqs = _random_families(rng, 'ACGT', int(rng.integers(4, 20)), int(rng.integers(1, 20)), 30, 300)
        p = nt_params(**{'-c': float(rng.choice([0.8, 0.9])), '-n': int(rng.choice([5, 8]))})
    else:
        seqs = _random_families(rng, AA, int(rng.integers(10, 60)), int(rng.integers(1, 30)), 20, int(rng.choice([200, 600])))
        p = params(**{'-c': float(rng.choice([0.7, 0.8, 0.9])), '-n': int(rng.choice([5, 4]))})
    p.batch_size = int(rng.choice([64, 256, 1024]))
    res, off = pack(seqs)
    results = run_virtual_ranks(res, off, p, 2 + seed % 2)
    assert_replicated(results)
    (assert_same_nt if nucleotide else assert_same)(fold(results), oracle.cluster_greedy(res, off, p))


@pytest.mark.slow
def test_cfg4_on_two_virtual_ranks_matches_single_process(gpu_ctx):
    """BASELINE config 4 (4000 synthetic genomes, 7.3 M non-redundant proteins) record-sharded over 2 virtual ranks,
    windows overlapping on two streams with their own exchange slots: clusters, members, identities and every
    counter equal the single-process result, which test_cfg4_shape_properties_and_prefix_parity ties to the
    size-independent properties and to the oracle on a prefix."""
    res, off, _ = synth.protein_set('cfg-4').nr_arrays()
    p = params()
    whole = gpu_ctx.cluster_greedy(res, off, p)
    results = run_virtual_ranks(res, off, p, 2)
    assert_replicated(results)
    assert_same(fold(results), whole)


def test_one_ranks_capacity_failure_ends_every_rank(monkeypatch):
    """A capacity failure on ONE process (here injected: rank 1 reports a pair buffer overflow with its 3rd exchange)
    travels with the exchanged keys: every rank returns the error from the same point of the window loop, none is
    left waiting in a collective (this test would hang otherwise)."""
    monkeypatch.setenv('PGX_WINDOW', '2048')
    monkeypatch.setenv('PGX_INJECT_ERROR', '1:3')
    ps = synth.ProteinSet(30, 500, 800, 150, 77)
    res, off, _ = ps.nr_arrays()
    errors = run_virtual_ranks(res, off, params(), 2, expect_errors=True)
    assert sorted(r for r, _ in errors) == [0, 1]
    assert all('candidate pair buffer overflow' in str(e) for _, e in errors), errors


def test_overlapped_windows_under_exchange_match_the_serial_loop(monkeypatch):
    """Two windows in flight with the exchange callback (slots 0 and 1 both used) against PGX_NO_OVERLAP=1."""
    monkeypatch.setenv('PGX_WINDOW', '1024')
    ps = synth.ProteinSet(30, 500, 800, 150, 77)
    res, off, _ = ps.nr_arrays()
    p = params()
    both = run_virtual_ranks(res, off, p, 2)
    monkeypatch.setenv('PGX_NO_OVERLAP', '1')
    serial = run_virtual_ranks(res, off, p, 2)
    assert_same(fold(both), fold(serial))
    assert_same(fold(both), oracle.cluster_greedy(res, off, p))


def test_edge_cases_on_virtual_ranks():
    """What a record-sharded call must survive without leaving a rank behind: nothing to cluster, fewer sequences than
    ranks, nucleotide rules with flush positions, a giant sequence."""
    from test_cluster_oracle import pack, rand_seq, mutate, rand_nt
    rng = np.random.default_rng(91)
    # nothing to cluster / everything too short
    for seqs in ([], ['MKV', 'ACD']):
        res, off = pack(seqs)
        results = run_virtual_ranks(res, off, params(), 2)
        assert all(r[4] == 0 for r in results) and all((r[0] < 0).all() for r in results)
    # fewer sequences than ranks
    a = rand_seq(rng, 120)
    res, off = pack([a, mutate(rng, a, 5)])
    results = run_virtual_ranks(res, off, params(), 3)
    assert_replicated(results)
    assert_same(fold(results), oracle.cluster_greedy(res, off, params()))
    # nucleotide rules, both strands, with flush positions
    res, off, _ = synth.noncoding_set(n_genomes=40, seed=11)
    p = nt_params()
    n = oracle.cluster_greedy(res, off, p)[5]['n_clustered']
    pc, keep = cluster.with_chunk_boundaries(p, [n // 3, n // 2])
    results = run_virtual_ranks(res, off, pc, 2)
    assert_replicated(results)
    assert_same_nt(fold(results), oracle.cluster_greedy(res, off, pc))
    # a giant protein and its variant among ordinary sequences
    g = rand_seq(rng, 34000)
    seqs = [g, mutate(rng, g, 900)] + [rand_seq(rng, 250) for _ in range(20)]
    seqs += [mutate(rng, s_, 8) for s_ in seqs[2:12]]
    res, off = pack(seqs)
    results = run_virtual_ranks(res, off, params(), 2)
    assert_replicated(results)
    assert_same(fold(results), oracle.cluster_greedy(res, off, params()))


def test_without_counters_on_virtual_ranks_clusters_the_same(gpu_ctx):
    """stats = NULL in the record-sharded mode: every rank leaves out the members that cannot gain from a pass over the
    new representatives (a function of the replicated best keys), the exchanges stay in step, and every rank returns
    the clusters of the single-process call with counters."""
    res, off, _ = synth.protein_set('small').nr_arrays()
    p = params()
    whole = gpu_ctx.cluster_greedy(res, off, p)
    results = run_virtual_ranks(res, off, p, 2, want_stats=False)
    for r in results:
        assert r[5] is None and r[4] == whole[4]
        np.testing.assert_array_equal(r[0], whole[0])
        np.testing.assert_array_equal(r[1], whole[1])
    np.testing.assert_array_equal(np.maximum(results[0][2], results[1][2]), whole[2])    # identities are partial per rank
