"""K3 parity on the GPU: libpgx (HIP, through the C ABI) against the reference's own
tables (tests/golden/pancore), against the CPU oracle on seeded inputs, and -- at
BASELINE's full size -- through size-independent properties. Integer work: bit-exact."""
import glob
import os

import numpy as np
import pytest
import scipy.sparse

import oracle
from pangenomix_amd import pangenome_analysis as pa
from pangenomix_amd import sparse_utils as su
from pangenomix_amd import synth

pytestmark = pytest.mark.gpu
CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'golden', 'pancore', '*.npz')))


def bitmap_reference(row, col, G, S, stride):
    bits = np.zeros((S, stride), dtype=np.uint64)
    np.bitwise_or.at(bits, (col, row >> 6), np.uint64(1) << (row & 63).astype(np.uint64))
    return bits


@pytest.mark.parametrize('path', CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_entry_point_reproduces_reference_table(path, gpu_ctx):
    z = np.load(path)
    G, S = (int(x) for x in z['shape'])
    coo = scipy.sparse.coo_matrix((np.ones(z['row'].size, dtype=np.int64), (z['row'], z['col'])), shape=(G, S))
    lsdf = su.LightSparseDataFrame(['g%d' % i for i in range(G)], ['s%d' % i for i in range(S)], coo)
    np.random.seed(int(z['seed']))
    df = pa.estimate_pan_core_size(lsdf, z['perms'].shape[0], ctx=gpu_ctx)
    assert df.values.dtype == np.float64
    assert np.array_equal(df.values, z['expected'])
    assert list(df.index) == list(z['index']) and list(df.columns) == list(z['columns'])


@pytest.mark.parametrize('G,S,density,n_iter', [
    (1, 1, 1.0, 1), (63, 5, 0.5, 3), (64, 64, 0.3, 4), (65, 65, 0.3, 4), (1000, 129, 0.1, 5),
    (8200, 200, 0.02, 9), (70001, 31, 0.5, 6), (300000, 16, 0.3, 3)])
def test_matches_oracle_on_seeded_inputs(G, S, density, n_iter, gpu_ctx):
    rng = np.random.default_rng(G * 7 + S)
    dense = rng.random((G, S)) < density
    row, col = (a.astype(np.int32) for a in np.nonzero(dense))
    perms = np.array([rng.permutation(S) for _ in range(n_iter)], dtype=np.int32)
    bits = gpu_ctx.presence_bitmap(row, col, G, S)
    assert np.array_equal(bits, bitmap_reference(row.astype(np.int64), col, G, S, bits.shape[1]))
    pan, core = gpu_ctx.pan_core(bits, G, perms)
    opan, ocore = oracle.pan_core(row, col, None, G, S, perms)
    assert np.array_equal(pan, opan) and np.array_equal(core, ocore)
    # the fused entry (coordinates up, bitmap built and consumed on the device) gives the same curves
    pan2, core2, dup = gpu_ctx.pan_core_coo(row, col, G, S, perms)
    assert dup == 0 and np.array_equal(pan2, opan) and np.array_equal(core2, ocore)


def test_empty_inputs(gpu_ctx):
    bits = gpu_ctx.presence_bitmap(np.zeros(0, np.int32), np.zeros(0, np.int32), 10, 3)
    assert bits.shape == (3, 16) and not bits.any()
    pan, core = gpu_ctx.pan_core(bits, 10, np.array([[2, 0, 1]], dtype=np.int32))
    assert pan.tolist() == [[0, 0, 0]] and core.tolist() == [[0, 0, 0]]


def test_duplicate_coordinates_are_counted_on_the_device_and_refused(gpu_ctx):
    """The OR/AND form equals the reference's loop only for a 0/1 table without duplicate
    coordinates (pangenome_analysis.py:88-90 sums duplicates to 2): the bitmap build counts the
    bits that were already set, and the entry point refuses such a table."""
    rng = np.random.default_rng(5)
    G, S = 5000, 37
    dense = rng.random((G, S)) < 0.2
    row, col = (a.astype(np.int32) for a in np.nonzero(dense))
    extra = rng.choice(row.size, size=123, replace=False)
    row2, col2 = np.concatenate([row, row[extra]]), np.concatenate([col, col[extra]])
    order = rng.permutation(row2.size)
    bits, dup = gpu_ctx.presence_bitmap(row2[order], col2[order], G, S, return_duplicates=True)
    assert dup == 123
    assert np.array_equal(bits, gpu_ctx.presence_bitmap(row, col, G, S))
    assert gpu_ctx.presence_bitmap(row, col, G, S, return_duplicates=True)[1] == 0
    coo = scipy.sparse.coo_matrix((np.ones(row2.size, dtype=np.int64), (row2, col2)), shape=(G, S))
    lsdf = su.LightSparseDataFrame(['g%d' % i for i in range(G)], ['s%d' % i for i in range(S)], coo)
    with pytest.raises(ValueError, match='duplicate'):
        pa.estimate_pan_core_size(lsdf, 2, ctx=gpu_ctx)
    two = scipy.sparse.coo_matrix((np.full(row.size, 2, dtype=np.int64), (row, col)), shape=(G, S))
    with pytest.raises(ValueError, match='binary'):
        pa.estimate_pan_core_size(su.LightSparseDataFrame(lsdf.index, lsdf.columns, two), 2, ctx=gpu_ctx)


def test_rejects_bad_arguments(gpu_ctx):
    from pangenomix_amd._native import PgxError
    with pytest.raises(PgxError, match='out of range'):
        gpu_ctx.presence_bitmap(np.array([10], np.int32), np.array([0], np.int32), 10, 3)
    bits = gpu_ctx.presence_bitmap(np.array([1], np.int32), np.array([0], np.int32), 10, 3)
    with pytest.raises(PgxError, match='permutation'):
        gpu_ctx.pan_core(bits, 10, np.array([[0, 1, 3]], dtype=np.int32))


def test_full_size_properties(gpu_ctx):
    """BASELINE config 3 size (150k genes x 400 genomes, 1000 iterations): properties that
    hold for every permutation, plus the oracle on a sample of the iterations."""
    row, col, G = synth.pancore_matrix()
    S, n_iter = 400, 1000
    rng = np.random.default_rng(0)
    perms = np.array([rng.permutation(S) for _ in range(n_iter)], dtype=np.int32)
    bits = gpu_ctx.presence_bitmap(row, col, G, S)
    pan, core = gpu_ctx.pan_core(bits, G, perms)
    per_genome = np.bincount(col, minlength=S)
    per_gene = np.bincount(row, minlength=G)
    assert np.array_equal(pan[:, 0], per_genome[perms[:, 0]])       # first step = that genome's gene count
    assert np.array_equal(core[:, 0], pan[:, 0])
    assert (np.diff(pan, axis=1) >= 0).all() and (np.diff(core, axis=1) <= 0).all()
    assert (pan[:, -1] == G).all()                                    # no empty rows -> union is everything
    assert (core[:, -1] == (per_gene == S).sum()).all()               # intersection is order independent
    assert (core <= pan).all()
    sample = [0, 1, 499, 998, 999]
    opan, ocore = oracle.pan_core(row, col, None, G, S, perms[sample])
    assert np.array_equal(pan[sample], opan) and np.array_equal(core[sample], ocore)
    # identical permutations give identical rows; reversing a permutation keeps the end points
    pan2, core2 = gpu_ctx.pan_core(bits, G, perms[::-1].copy())
    assert np.array_equal(pan2[::-1], pan) and np.array_equal(core2[::-1], core)


def test_device_resident_handoff_from_the_pipeline(tmp_path, gpu_ctx):
    """SURVEY build plan step 5: build_cds_pangenome() leaves the gene x genome bitmap ON THE DEVICE, built there from the
    clustering result (rows = cluster numbers), and estimate_pan_core_size(df_genes) on the returned table consumes it
    without uploading the table: same curves, same generator state as the path that uploads the coordinates; the bitmap
    equals the one built from the table's own coordinates; a later pipeline makes the token stale and the call falls back."""
    import copy
    from pangenomix_amd import _native, pangenome, sparse_utils, synth
    from pangenomix_amd import pangenome_analysis as pa
    ctx = _native.default_context()
    paths = synth.ProteinSet(9, 300, 400, 90, 5).write_faa(str(tmp_path / 'g1'))
    (tmp_path / 'o1').mkdir()
    dfa, dfg = pangenome.build_cds_pangenome(paths, str(tmp_path / 'o1'), name='R')
    res = dfg._pgx_resident
    assert res['token'] and res['shape'] == dfg.shape
    # the resident bitmap against the table's own coordinates (row = cluster number of the gene name)
    G, S = dfg.shape
    cluster_of_row = np.array([int(str(x).rsplit('_C', 1)[1]) for x in dfg.index], dtype=np.int32)
    coo = dfg.data.tocoo()
    want_bits, dup = ctx.presence_bitmap(cluster_of_row[coo.row], coo.col.astype(np.int32), G, S, return_duplicates=True)
    assert dup == 0 and np.array_equal(ctx.bitmap_resident_read(res['token'], G, S), want_bits)
    plain = sparse_utils.LightSparseDataFrame(list(dfg.index), list(dfg.columns), dfg.data.copy())   # no hand-off
    np.random.seed(3)
    a = pa.estimate_pan_core_size(dfg, 50)
    state_a = np.random.get_state()
    np.random.seed(3)
    b = pa.estimate_pan_core_size(plain, 50)
    state_b = np.random.get_state()
    assert a.equals(b) and state_a[2] == state_b[2] and np.array_equal(state_a[1], state_b[1])
    assert a.values[:, S - 1].max() == G                      # all genomes: every gene
    # another pipeline replaces the resident bitmap: the first table's token is stale, its call uploads instead
    paths2 = synth.ProteinSet(5, 200, 300, 60, 6).write_faa(str(tmp_path / 'g2'))
    (tmp_path / 'o2').mkdir()
    pangenome.build_cds_pangenome(paths2, str(tmp_path / 'o2'), name='R2')
    with pytest.raises(_native.PgxError):
        ctx.bitmap_resident_read(res['token'], G, S)
    np.random.seed(3)
    c = pa.estimate_pan_core_size(dfg, 50)
    assert c.equals(b)
