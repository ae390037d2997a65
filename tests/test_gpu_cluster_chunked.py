"""cd-hit's MEMORY-CHUNKED rule (pgx.h `chunk_boundaries`) on the GPU against the oracle: at a flush
position every sequence not yet clustered is compared with the index as it is, then the index is emptied. Where the
real program flushes cannot be restated offline, so positions are inputs; parity (like all of K1/K2) is GPU == own
restatement, bit for bit, counters included."""
import numpy as np
import pytest

import oracle
from pangenomix_amd import cluster, synth
from test_cluster_oracle import AA, _chunk_case, pack, params, run
from test_gpu_cluster import _random_families, assert_same, assert_same_nt, nt_params

pytestmark = pytest.mark.gpu


def both(ctx, res, off, p, boundaries):
    pc, keep = cluster.with_chunk_boundaries(p, boundaries)
    return ctx.cluster_greedy(res, off, pc), oracle.cluster_greedy(res, off, pc)


def test_known_answer_membership_change(gpu_ctx):
    seqs = next(s for s in (_chunk_case(seed) for seed in range(40)) if run(s)[0].tolist() == [0, 1, 1])
    res, off = pack(seqs)
    got, want = both(gpu_ctx, res, off, params(), [1])
    assert got[0].tolist() == [0, 1, 0]
    assert_same(got, want)
    for bd in ([2], [1, 2]):
        assert_same(*both(gpu_ctx, res, off, params(), bd))


@pytest.mark.parametrize('window', [0, 256])
def test_synthetic_set_with_flushes_matches_oracle(window, gpu_ctx):
    """Several windows per chunk and several sweep windows per flush (window 256), and the default window."""
    ps = synth.ProteinSet(30, 500, 800, 150, 77)
    res, off, _ = ps.nr_arrays()
    p = params()
    p.batch_size = window
    n = oracle.cluster_greedy(res, off, p)[5]['n_clustered']
    unchunked = gpu_ctx.cluster_greedy(res, off, p)
    for bd in ([n // 3], [n // 10, n // 4, n // 2, n - 5], [1], [n - 1]):
        got, want = both(gpu_ctx, res, off, p, bd)
        assert_same(got, want)
    assert got[4] >= unchunked[4]


def test_nucleotide_both_strands_with_flushes(gpu_ctx):
    res, off, _ = synth.noncoding_set(n_genomes=60, seed=9)
    p = nt_params()
    n = oracle.cluster_greedy(res, off, p)[5]['n_clustered']
    got, want = both(gpu_ctx, res, off, p, [n // 4, n // 2])
    assert_same_nt(got, want)


@pytest.mark.parametrize('seed', range(8))
def test_randomized_sets_with_random_flush_positions(seed, gpu_ctx):
    rng = np.random.default_rng(7000 + seed)
    nucleotide = seed % 4 == 3
    if nucleotide:
        seqs = _random_families(rng, 'ACGT', int(rng.integers(4, 20)), int(rng.integers(1, 20)), 30, 300)
        p = nt_params(**{'-c': float(rng.choice([0.8, 0.9]))})
    else:
        seqs = _random_families(rng, AA, int(rng.integers(10, 60)), int(rng.integers(1, 30)), 20, int(rng.choice([200, 600])))
        p = params(**{'-c': float(rng.choice([0.7, 0.8, 0.9]))})
    p.batch_size = int(rng.choice([0, 64, 512]))
    res, off = pack(seqs)
    n = oracle.cluster_greedy(res, off, p)[5]['n_clustered']
    bd = sorted(set(int(x) for x in rng.integers(1, n, size=int(rng.integers(1, 5)))))
    got, want = both(gpu_ctx, res, off, p, bd)
    (assert_same_nt if nucleotide else assert_same)(got, want)


def test_bad_positions_are_refused(gpu_ctx):
    from pangenomix_amd import _native
    res, off = pack(_chunk_case(0))
    for bad in ([0], [3], [2, 1]):
        b = np.array(bad, dtype=np.uint32)
        import ctypes as C
        p = params()
        p.chunk_boundaries = b.ctypes.data_as(C.POINTER(C.c_uint32))
        p.n_chunk_boundaries = b.size
        with pytest.raises(_native.PgxError):
            gpu_ctx.cluster_greedy(res, off, p)


def test_flushes_on_virtual_ranks(monkeypatch):
    """The record-sharded mode runs the sweeps as well (phase A of a sweep window is split by record like any other)."""
    from test_gpu_cluster_sharded import assert_replicated, fold, run_virtual_ranks
    monkeypatch.setenv('PGX_WINDOW', '2048')
    ps = synth.ProteinSet(20, 400, 600, 120, 78)
    res, off, _ = ps.nr_arrays()
    p = params()
    n = oracle.cluster_greedy(res, off, p)[5]['n_clustered']
    pc, keep = cluster.with_chunk_boundaries(p, [n // 3, 2 * n // 3])
    results = run_virtual_ranks(res, off, pc, 2)
    assert_replicated(results)
    assert_same(fold(results), oracle.cluster_greedy(res, off, pc))


@pytest.mark.parametrize('window', [0, 256])
def test_flushes_without_counters_cluster_the_same(window, gpu_ctx):
    """stats = NULL together with flush positions: the sweeps run no pass over new representatives, the windows between
    them do and leave out the members that cannot gain from it -- the same clusters as the call with counters."""
    ps = synth.ProteinSet(30, 500, 800, 150, 77)
    res, off, _ = ps.nr_arrays()
    p = params()
    p.batch_size = window
    n = off.size - 1
    for bd in ([n // 3], [n // 10, n // 4, n // 2]):
        pc, keep = cluster.with_chunk_boundaries(p, bd)
        full = gpu_ctx.cluster_greedy(res, off, pc)
        lean = gpu_ctx.cluster_greedy(res, off, pc, want_stats=False)
        assert lean[5] is None and lean[4] == full[4]
        for i in range(4):
            np.testing.assert_array_equal(lean[i], full[i])
