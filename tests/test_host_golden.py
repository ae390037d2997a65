"""Host plumbing against fixtures produced by the reference itself
(tests/golden/make_golden.py). Byte-for-byte on every text file, array-for-array
(values, dtypes, order) on the .npz members."""
import filecmp
import json
import os
import shutil

import numpy as np
import pytest

from pangenomix_amd import pangenome as pg
from pangenomix_amd import sparse_utils as su

GENOMES = ['gB', 'gA', 'g10', 'g2', 'gC.v1', 'gD']   # the (unsorted) order the fixture was built with


@pytest.fixture()
def cds(tmp_path, golden_dir):
    src = os.path.join(golden_dir, 'cds')
    work = tmp_path / 'cds'
    shutil.copytree(os.path.join(src, 'in'), work)
    paths = [str(work / (g + '.faa')) for g in GENOMES]
    return str(work), paths, os.path.join(src, 'expected')


def same_file(a, b):
    assert filecmp.cmp(a, b, shallow=False), 'files differ: %s vs %s' % (a, b)


def test_consolidate_seqs_matches_reference(cds, capsys):
    work, paths, exp = cds
    nr, shared, missing = (os.path.join(work, n) for n in
                           ('T_nr.faa', 'T_redundant_headers.tsv', 'T_missing_headers.txt'))
    groups, miss = pg.consolidate_seqs(paths, nr, shared, missing)
    same_file(nr, os.path.join(exp, 'T_nr.consolidated.faa'))
    same_file(shared, os.path.join(exp, 'T_redundant_headers.tsv'))
    same_file(missing, os.path.join(exp, 'T_missing_headers.txt'))
    ret = json.load(open(os.path.join(exp, 'consolidate_return.json')))
    assert [[k.hex(), v] for k, v in groups.items()] == ret['groups']
    assert miss == ret['missing']
    out = json.load(open(os.path.join(exp, 'stdout.json')))
    assert capsys.readouterr().out == out['consolidate']


def run_rename(work, exp):
    shutil.copy(os.path.join(exp, 'T_nr.consolidated.faa'), os.path.join(work, 'T_nr.faa'))
    shutil.copy(os.path.join(exp, 'T_redundant_headers.tsv'), work)
    nr = os.path.join(work, 'T_nr.faa')
    return pg.rename_genes_and_alleles(
        os.path.join(work, 'T_nr.faa.cdhit.clstr'), nr, nr, os.path.join(work, 'T_allele_names.tsv'),
        name='T', cluster_type='cds', shared_headers_file=os.path.join(work, 'T_redundant_headers.tsv'))


def test_rename_genes_and_alleles_matches_reference(cds, capsys):
    work, paths, exp = cds
    h2a = run_rename(work, exp)
    same_file(os.path.join(work, 'T_allele_names.tsv'), os.path.join(exp, 'T_allele_names.tsv'))
    same_file(os.path.join(work, 'T_nr.faa'), os.path.join(exp, 'T_nr.faa'))
    assert h2a == json.load(open(os.path.join(exp, 'header_to_allele.json')))
    assert capsys.readouterr().out == json.load(open(os.path.join(exp, 'stdout.json')))['rename']
    assert not os.path.exists(os.path.join(work, 'T_nr.faa.tmp'))


def npz_members(path):
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


def assert_same_npz(a, b):
    ma, mb = npz_members(a), npz_members(b)
    assert sorted(ma) == sorted(mb) == ['col', 'data', 'format', 'row', 'shape']
    for k in ma:
        assert ma[k].dtype == mb[k].dtype, k
        assert ma[k].shape == mb[k].shape, k
        assert np.array_equal(ma[k], mb[k]), k


def test_feature_tables_and_npz_match_reference(cds, capsys):
    work, paths, exp = cds
    h2a = json.load(open(os.path.join(exp, 'header_to_allele.json')))
    capsys.readouterr()
    dfa, dfg = pg.build_genetic_feature_tables(os.path.join(work, 'T_nr.faa.cdhit.clstr'), paths, 'T',
                                               cluster_type='cds', header_to_allele=h2a)
    assert capsys.readouterr().out == json.load(open(os.path.join(exp, 'stdout.json')))['tables']
    for df, stem in ((dfa, 'T_strain_by_allele'), (dfg, 'T_strain_by_gene')):
        out = os.path.join(work, stem + '.npz')
        df.to_npz(out)
        same_file(out + '.labels.txt', os.path.join(exp, stem + '.npz.labels.txt'))
        assert_same_npz(out, os.path.join(exp, stem + '.npz'))
        back = su.read_lsdf(out)
        assert back.shape == df.shape
        assert list(back.index) == list(df.index) and list(back.columns) == list(df.columns)
        assert (back.data != df.data).nnz == 0
    # lexicographic row order: C100 < C10 < C11 < C1 < C2
    idx = list(dfg.index)
    assert idx.index('T_C100') < idx.index('T_C10') < idx.index('T_C11') < idx.index('T_C1') < idx.index('T_C2')
    assert list(dfg.columns) == sorted(GENOMES)


def test_gene_row_is_or_of_allele_rows(cds):
    """The reference validators' invariant (pangenome.py:1299-1330)."""
    work, paths, exp = cds
    dfa = su.read_lsdf(os.path.join(exp, 'T_strain_by_allele.npz'))
    dfg = su.read_lsdf(os.path.join(exp, 'T_strain_by_gene.npz'))
    genes = np.array([pg.__get_gene_from_allele__(a) for a in dfa.index])
    A = dfa.data.toarray() > 0
    G = dfg.data.toarray() > 0
    for gi, gene in enumerate(dfg.index):
        assert np.array_equal(A[genes == gene].any(axis=0), G[gi])


def test_load_header_to_allele_from_clstr(cds):
    work, paths, exp = cds
    full = pg.load_header_to_allele(os.path.join(work, 'T_nr.faa.cdhit.clstr'),
                                    os.path.join(exp, 'T_redundant_headers.tsv'), None, 'T', 'cds')
    assert full == json.load(open(os.path.join(exp, 'header_to_allele.json')))


@pytest.mark.parametrize('flank,tag', [((0, 0), ''), ((7, 12), '_f7_12')])
def test_extract_noncoding_matches_reference(tmp_path, golden_dir, flank, tag):
    src = os.path.join(golden_dir, 'noncoding')
    for g in ('n1', 'n2'):
        out = str(tmp_path / (g + '.fna'))
        pg.extract_noncoding(os.path.join(src, 'in', g + '.gff'), os.path.join(src, 'in', g + '.fna'),
                             out, flanking=flank)
        same_file(out, os.path.join(src, 'expected', g + '_noncoding' + tag + '.fna'))


def test_small_helpers():
    assert pg.create_feature_name('X', 'cds', 12, 'allele', 3) == 'X_C12A3'
    assert pg.create_feature_name('X', 'noncoding', '7') == 'X_T7'
    assert pg.__get_gene_from_allele__('Eco_C10A12') == 'Eco_C10'
    assert pg.__get_gene_from_allele__('ACME_C1A0') == 'ACME_C1'
    assert pg.__get_genome_from_filename__('/a/b/gC.v1.faa') == 'gC.v1'
    assert pg.__get_header_from_fasta_line__('>fig|1.peg.2   desc\n') == 'fig|1.peg.2'
    assert pg.reverse_complement('ACGTNacgtRYKM') == 'KMRYacgtNACGT'
    with pytest.raises(KeyError):
        pg.reverse_complement('ACGX')


def test_label_maps_are_the_construction_time_maps():
    """SURVEY App. B.5 (reference sparse_utils.py:199-200, pangenome.py:292-293): index_map / column_map are filled
    when the frame is built and never refreshed, so after `.columns` is re-assigned labelslice() still resolves the
    OLD names; the maps are plain attributes a caller may replace."""
    import scipy.sparse
    df = su.LightSparseDataFrame(['g0', 'g1'], ['a_noncoding', 'b_noncoding'],
                                           scipy.sparse.coo_matrix(np.array([[1, 0], [1, 1]])))
    df.columns = np.array(['a', 'b'])
    assert df.column_map == {'a_noncoding': 0, 'b_noncoding': 1}
    assert df.labelslice(columns=['b_noncoding']).values.tolist() == [[0], [1]]
    with pytest.raises(KeyError):
        df.labelslice(columns=['b'])
    df.column_map = {'a': 0, 'b': 1}
    assert df.labelslice(columns=['b']).values.tolist() == [[0], [1]]
    assert df.index_map == {'g0': 0, 'g1': 1}


def test_npz_members_deflated_in_pieces_read_back_whole(tmp_path):
    """to_npz() deflates a member in 1 MB pieces on several threads and puts the pieces end to end
    (sparse_utils.write_npz_parallel): the standard readers must see one intact member -- compressible and
    incompressible data, sizes around the piece boundary, a 0-d and an empty member, a transposed view."""
    import zipfile
    rng = np.random.default_rng(3)
    piece = 1 << 14
    members = [('noise', rng.integers(0, 1 << 62, 5 * piece // 8 + 3)), ('ones', np.ones(3 * piece // 8, dtype=np.int64)),
               ('edge', np.arange((piece - 128) // 4, dtype=np.int32)), ('edge1', np.arange((piece - 128) // 4 + 1, dtype=np.int32)),
               ('format', np.array(b'coo')), ('empty', np.zeros(0, dtype=np.int32)), ('view', np.arange(12).reshape(3, 4).T)]
    path = str(tmp_path / 'm.npz')
    su.write_npz_parallel(path, members, chunk=piece)
    with zipfile.ZipFile(path) as z:
        assert z.testzip() is None
        assert [i.filename for i in z.infolist()] == [n + '.npy' for n, _ in members]
    with np.load(path) as got:
        for name, arr in members:
            assert got[name].dtype == arr.dtype and got[name].shape == arr.shape and (got[name] == arr).all(), name
