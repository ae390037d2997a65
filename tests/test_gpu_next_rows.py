"""SURVEY 8f-2 / 8f-3 on the GPU against fixtures produced by the reference itself
(tests/golden/next, tests/golden/make_golden_next.py) and against the CPU oracle.
  fit_heaps_by_iteration   floating point: rtol 1e-5 against the reference's curve_fit results. (curve_fit stops
                           at its default ftol = xtol = 1e-8, which leaves its parameters within ~3e-6 of the
                           least-squares minimum -- re-running it with tolerances of 1e-15 moves alpha by 1-3e-6 on
                           these tables; the device fit iterates to the minimum itself.)
  count_*_occurence / find_core_genes   integer work: exact, dtypes included"""
import glob
import json
import os

import numpy as np
import pandas as pd
import pytest

from pangenomix_amd import allele_identification, core_genome, plot, synth
from pangenomix_amd import pangenome_analysis as pa

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
HEAPS = sorted(glob.glob(os.path.join(HERE, 'golden', 'next', 'heaps_*.npz')))
RTOL = 1e-5


@pytest.mark.parametrize('path', HEAPS, ids=[os.path.basename(p)[6:-4] for p in HEAPS])
def test_heaps_fit_reproduces_reference(path, gpu_ctx):
    want = np.load(path)
    z = np.load(os.path.join(HERE, 'golden', 'pancore', os.path.basename(path)[6:]))
    df = pd.DataFrame(z['expected'], index=[str(x) for x in z['index']], columns=[str(x) for x in z['columns']])
    fit = pa.fit_heaps_by_iteration(df, ctx=gpu_ctx)
    assert list(fit.columns) == [str(x) for x in want['columns']] == ['alpha', 'kappa']
    assert list(fit.index) == [str(x) for x in want['index']]
    np.testing.assert_allclose(fit["alpha"].values, want["alpha"], rtol=RTOL, atol=1e-7)   # (a flat curve: alpha = 0 up to the optimisers' stopping tolerance)
    np.testing.assert_allclose(fit['kappa'].values, want['kappa'], rtol=RTOL)
    np.testing.assert_array_equal(plot.calculate_mean(df).values[0], want['mean'])


def test_heaps_fit_full_size_against_oracle(gpu_ctx):
    """1000 iterations x 400 genomes (the benchmark's pan table): device fits against scipy's on a sample."""
    from oracle import heaps_ref
    row, col, G = synth.pancore_matrix()
    S, n_iter = 400, 1000
    rng = np.random.default_rng(0)
    perms = np.array([rng.permutation(S) for _ in range(n_iter)], dtype=np.int32)
    pan, core, dup = gpu_ctx.pan_core_coo(row, col, G, S, perms)
    alpha, kappa = gpu_ctx.heaps_fit(pan)
    sample = [0, 1, 499, 999]
    oa, ok = heaps_ref.fit_rows(pan[sample])
    np.testing.assert_allclose(alpha[sample], oa, rtol=RTOL)
    np.testing.assert_allclose(kappa[sample], ok, rtol=RTOL)
    assert (alpha > 0).all() and (alpha < 1).all() and (kappa > 0).all()


def test_occurrence_counts_reproduce_reference(gpu_ctx, golden_dir, capsys):
    want = json.load(open(os.path.join(golden_dir, 'next', 'occurrence.json')))
    exp = os.path.join(golden_dir, 'cds', 'expected')
    g = core_genome.count_gene_occurence(os.path.join(exp, 'T_strain_by_gene.npz'), ctx=gpu_ctx)
    a = allele_identification.count_allele_occurence(os.path.join(exp, 'T_strain_by_allele.npz'), ctx=gpu_ctx)
    for df, w in ((g, want['gene']), (a, want['allele'])):
        assert list(df.columns) == w['columns'] and [str(t) for t in df.dtypes] == w['dtypes']
        assert df.values.tolist() == w['values']
        assert list(df.index) == list(range(len(df)))
    for k, w in want['core'].items():
        c = core_genome.find_core_genes(g, int(k))
        assert list(c.columns) == w['columns'] and [str(t) for t in c.dtypes] == w['dtypes']
        assert c.values.tolist() == w['values']
    assert 'Counted gene occurence' in capsys.readouterr().out


def test_occurrence_counts_full_size(gpu_ctx, tmp_path):
    """150,000 genes x 400 genomes: equal to numpy's bincount; duplicates fall back to triple counts."""
    import scipy.sparse
    row, col, G = synth.pancore_matrix()
    path = str(tmp_path / 'genes.npz')
    scipy.sparse.save_npz(path, scipy.sparse.coo_matrix((np.ones(row.size, np.int64), (row, col)), shape=(G, 400)))
    df = core_genome.count_gene_occurence(path, ctx=gpu_ctx)
    counts = np.bincount(row, minlength=G)
    assert np.array_equal(df['gene_index'].values, np.flatnonzero(counts)) and np.array_equal(df['count'].values, counts[counts > 0])
    core = core_genome.find_core_genes(df, 400)
    assert np.array_equal(core['gene_index'].values, np.flatnonzero(counts == 400))
    row2, col2 = np.concatenate([row, row[:5]]), np.concatenate([col, col[:5]])
    scipy.sparse.save_npz(path, scipy.sparse.coo_matrix((np.ones(row2.size, np.int64), (row2, col2)), shape=(G, 400)))
    df2 = core_genome.count_gene_occurence(path, ctx=gpu_ctx)
    assert np.array_equal(df2['count'].values, np.bincount(row2, minlength=G)[counts > 0])


def test_heaps_fit_of_two_genomes_is_the_exact_two_point_fit(gpu_ctx):
    """scipy's leastsq refuses only fewer points than parameters: the reference fits a 2-genome table exactly
    (fixture heaps_two_genomes, produced by the reference); one genome raises TypeError there and here."""
    z = np.load(os.path.join(HERE, 'golden', 'pancore', 'two_genomes.npz'))
    df = pd.DataFrame(z['expected'], index=[str(x) for x in z['index']], columns=[str(x) for x in z['columns']])
    fit = pa.fit_heaps_by_iteration(df, ctx=gpu_ctx)
    pan = df.values[:, :2]
    np.testing.assert_allclose(fit['kappa'].values, pan[:, 0], rtol=1e-9)
    np.testing.assert_allclose(fit['alpha'].values, np.log2(pan[:, 1] / pan[:, 0]), rtol=1e-9)
    with pytest.raises(TypeError):
        pa.fit_heaps_by_iteration(df.iloc[:, [0, 2]], ctx=gpu_ctx)
