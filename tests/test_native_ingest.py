"""libpgx's host side (csrc/ingest.cpp) against (a) the fixtures produced by the reference
itself (tests/golden/cds: byte for byte on every text file, array for array on the .npz members) and
(b) the step-by-step Python functions -- themselves pinned by tests/test_host_golden.py -- on randomly
generated FASTA sets with everything the reader has rules for. No GPU needed: the clustering call inside
the pipeline is replaced by a reader of the fixture's .clstr or by the CPU oracle."""
import filecmp
import json
import os
import shutil

import numpy as np
import pytest

from pangenomix_amd import _native, cluster
from pangenomix_amd import pangenome as pg
from test_host_golden import GENOMES, assert_same_npz, same_file


@pytest.fixture()
def cds(tmp_path, golden_dir):
    src = os.path.join(golden_dir, 'cds')
    work = tmp_path / 'cds'
    shutil.copytree(os.path.join(src, 'in'), work)
    return str(work), [str(work / (g + '.faa')) for g in GENOMES], os.path.join(src, 'expected')


def clstr_reader(clstr_file, headers_of_groups):
    """A stand-in for the clustering call: cluster / member / identity per non-redundant sequence as a
    given .clstr states them (sequences it does not list stay unclustered)."""
    found = {}
    with open(clstr_file) as f:
        for line in f:
            if line[0] == '>':
                c = int(line.split()[-1])
            else:
                tok = line.split()
                pct = 0.0 if tok[-1] == '*' else float(tok[-1].rstrip('%').lstrip('+-/'))
                found[tok[2][1:-3]] = (c, int(tok[0]), pct / 100.0)

    def fn(residues, offsets, params):
        heads = headers_of_groups()
        cl = np.array([found.get(h, (-1, -1, 0))[0] for h in heads], dtype=np.int32)
        mem = np.array([found.get(h, (-1, -1, 0))[1] for h in heads], dtype=np.int32)
        iden = np.array([found.get(h, (-1, -1, 0))[2] for h in heads], dtype=np.float32)
        return cl, mem, iden, np.zeros(len(heads), np.uint8), int(cl.max()) + 1
    return fn


def test_fasta_set_reproduces_consolidate_seqs(cds):
    work, paths, exp = cds
    fs = _native.FastaSet(paths, threads=3)
    assert fs.simple, fs.why
    nr, shared, missing = (os.path.join(work, n) for n in ('T_nr.faa', 'T_redundant_headers.tsv', 'T_missing_headers.txt'))
    fs.write_consolidated(nr, shared, missing)
    same_file(nr, os.path.join(exp, 'T_nr.consolidated.faa'))
    same_file(shared, os.path.join(exp, 'T_redundant_headers.tsv'))
    same_file(missing, os.path.join(exp, 'T_missing_headers.txt'))
    ret = json.load(open(os.path.join(exp, 'consolidate_return.json')))
    heads, grp = fs.headers(), fs.group_of_record
    digests = fs.digests.reshape(-1, 32)
    groups = [[bytes(digests[g]).hex(), [heads[r] for r in np.flatnonzero(grp == g)]] for g in range(fs.n_groups)]
    assert groups == ret['groups']                                   # sha256 keys, first-seen order, encounter order
    assert [heads[r] for r in np.flatnonzero(grp == -1)] == ret["missing"]
    # the sequences handed to the clustering call = what the FASTA reader makes of the nr file
    h2, res, off, _ = cluster.read_fasta_for_clustering(nr)
    raw = [bytes(fs.residues[fs.offsets[g]:fs.offsets[g + 1]]).decode() for g in range(fs.n_groups)]
    cleaned = [''.join(ch for ch in s if ch.isalpha()).upper() for s in raw]
    assert cleaned == [bytes(res[off[i]:off[i + 1]]).decode() for i in range(len(h2))]
    assert fs.letters.tolist() == [len(s) for s in cleaned]
    assert h2 == fs.headers(fs.rep_of_group)
    fs.close()


def test_native_pipeline_matches_reference_outputs(cds, capsys):
    work, paths, exp = cds
    nr = os.path.join(work, 'T_nr.faa')
    holder = {}

    def heads():
        fs = _native.FastaSet(paths)
        try:
            return fs.headers(fs.rep_of_group)
        finally:
            fs.close()
    shutil.copy(os.path.join(work, 'T_nr.faa.cdhit.clstr'), os.path.join(work, 'given.clstr'))
    out = pg._native_pipeline(paths, nr, os.path.join(work, 'T_redundant_headers.tsv'),
                              os.path.join(work, 'T_missing_headers.txt'), os.path.join(work, 'T_allele_names.tsv'),
                              'T', 'cds', {'-n': 5, '-c': 0.8}, None,
                              cluster_fn=clstr_reader(os.path.join(work, 'given.clstr'), heads))
    assert out is not None
    dfa, dfg = out
    printed = capsys.readouterr().out
    for f in ('T_nr.faa', 'T_redundant_headers.tsv', 'T_missing_headers.txt', 'T_allele_names.tsv'):
        same_file(os.path.join(work, f), os.path.join(exp, f))
    for df, stem in ((dfa, 'T_strain_by_allele'), (dfg, 'T_strain_by_gene')):
        path = os.path.join(work, stem + '.npz')
        df.to_npz(path)
        same_file(path + '.labels.txt', os.path.join(exp, stem + '.npz.labels.txt'))
        assert_same_npz(path, os.path.join(exp, stem + '.npz'))
    # the .clstr it wrote says what the given one says (cluster, member, header per line)
    assert list(pg._parse_clstr(os.path.join(work, 'T_nr.faa.cdhit.clstr'))) == list(pg._parse_clstr(os.path.join(work, 'given.clstr')))
    # the same records are reported missing as by the reference's two loops
    ref_out = json.load(open(os.path.join(exp, 'stdout.json')))
    want_missing = [ln for ln in (ref_out['rename'] + ref_out['tables']).splitlines() if ln.startswith('MISSING:')]
    assert [ln for ln in printed.splitlines() if ln.startswith('MISSING:')] == want_missing
    assert not os.path.exists(nr + '.tmp')


def random_fasta_set(rng, directory, n_files=5, quirks=True):
    aa = np.array(list('ACDEFGHIKLMNPQRSTVWY'))
    fams = [''.join(rng.choice(aa, int(n))) for n in rng.integers(15, 300, 25)]
    paths = []
    for g in range(n_files):
        lines = []
        for k in range(int(rng.integers(20, 60))):
            s = fams[int(rng.integers(0, len(fams)))]
            if rng.random() < 0.4:                            # a point mutant: another sequence
                p = int(rng.integers(0, len(s)))
                s = s[:p] + str(rng.choice(aa)) + s[p + 1:]
            kind = rng.random()
            head = '>g%d|p%d' % (g, k) + ('  some text here' if rng.random() < 0.5 else '')
            wrap = int(rng.integers(10, 80))
            body = [s[i:i + wrap] for i in range(0, len(s), wrap)]
            if quirks:
                if kind < 0.05:
                    body = []                                 # no sequence at all
                elif kind < 0.10:
                    body = ['', '  ']                         # blank lines only
                elif kind < 0.20:
                    body = [b + '  ' for b in body]           # trailing blanks
                elif kind < 0.25:
                    body = ['\t' + b for b in body] + ['']    # leading tabs, a blank line
                elif kind < 0.30:
                    body = [b.lower() for b in body]
                elif kind < 0.35:
                    body[-1] = body[-1] + '*'
                elif kind < 0.38:
                    body = ['MKV']                            # too short for the clusterer
                elif kind < 0.41:
                    head = '>' if rng.random() < 0.5 else '>  only a description'   # a record without a name
            lines.append(head)
            lines.extend(body)
        text = '\n'.join(lines) + ('\n' if rng.random() < 0.7 or not quirks else '')
        if quirks and g == 1:
            text = '\n\n' + text                              # blank lines before the first header
        if quirks and g == 3:
            text = 'ACDEFGHIKLMNP\n' + text                    # sequence text before the first header
        path = os.path.join(directory, 'genome_%d%s.faa' % (g, '.v2' if g == 2 else ''))
        with open(path, 'w') as f:
            f.write(text)
        paths.append(path)
    order = rng.permutation(len(paths))
    return [paths[i] for i in order]


@pytest.mark.parametrize('seed', range(6))
def test_native_pipeline_equals_step_by_step_functions(tmp_path, seed, capsys):
    """Everything build_cds_pangenome() writes, by both routes, with the CPU oracle as the clusterer."""
    import oracle
    rng = np.random.default_rng(seed)
    src = tmp_path / 'genomes'
    src.mkdir()
    paths = random_fasta_set(rng, str(src))
    args = {'-n': 5, '-c': 0.8}
    a, b = tmp_path / 'native', tmp_path / 'python'
    a.mkdir(), b.mkdir()
    names = ('X_nr.faa', 'X_redundant_headers.tsv', 'X_missing_headers.txt', 'X_allele_names.tsv')
    fa = [str(a / n) for n in names]
    fb = [str(b / n) for n in names]
    out = pg._native_pipeline(paths, fa[0], fa[1], fa[2], fa[3], 'X', 'cds', args, None,
                              cluster_fn=lambda r, o, p: oracle.cluster_greedy(r, o, p))
    assert out is not None
    printed_native = capsys.readouterr().out
    # the step-by-step route, the oracle behind cluster_with_cdhit's file interface
    pg.consolidate_seqs(paths, fb[0], fb[1], fb[2])
    headers, res, off, records = cluster.read_fasta_for_clustering(fb[0])
    cl, mem, iden, strand, nc, _ = oracle.cluster_greedy(res, off, cluster.params_from_cdhit_args(args))
    cluster.write_clstr(fb[0] + '.cdhit.clstr', headers, np.diff(off.astype(np.int64)), cl, mem, iden, strand, False)
    h2a = pg.rename_genes_and_alleles(fb[0] + '.cdhit.clstr', fb[0], fb[0], fb[3], name='X', cluster_type='cds',
                                      shared_headers_file=fb[1])
    dfa, dfg = pg.build_genetic_feature_tables(fb[0] + '.cdhit.clstr', paths, 'X', cluster_type='cds', header_to_allele=h2a)
    printed_python = capsys.readouterr().out
    for x, y in zip(fa, fb):
        assert filecmp.cmp(x, y, shallow=False), os.path.basename(x)
    assert filecmp.cmp(fa[0] + '.cdhit.clstr', fb[0] + '.cdhit.clstr', shallow=False)
    for got, want, stem in ((out[0], dfa, 'alleles'), (out[1], dfg, 'genes')):
        got.to_npz(str(a / (stem + '.npz')))
        want.to_npz(str(b / (stem + '.npz')))
        assert filecmp.cmp(str(a / (stem + '.npz.labels.txt')), str(b / (stem + '.npz.labels.txt')), shallow=False)
        assert_same_npz(str(a / (stem + '.npz')), str(b / (stem + '.npz')))
    missing = lambda text: [ln for ln in text.splitlines() if ln.startswith('MISSING:')]   # noqa: E731
    assert missing(printed_native) == missing(printed_python)


@pytest.mark.parametrize('text,why', [
    ('>a\r\nMKV\r\n', 'carriage'), ('>a\nMK\xc3\xa9V\n', 'non-ASCII'),
    ('>a x\nMKVLLA\n>a y\nMKVLLC\n', 'two different sequences'), ('>a\nMK\x0bV\n', 'control')])
def test_special_inputs_are_left_to_the_python_path(tmp_path, text, why, capsys):
    p = tmp_path / 'odd.faa'
    p.write_bytes(text.encode('latin-1'))
    other = tmp_path / 'fine.faa'
    other.write_text('>z\nMKVLLAAAAAAAAAAAAAAA\n')
    fs = _native.FastaSet([str(other), str(p)])
    assert not fs.simple and why in fs.why
    assert fs.group_of_record.size == 0 and fs.residues.size == 0
    fs.close()
    assert pg._native_pipeline([str(other), str(p)], str(tmp_path / 'nr.faa'), str(tmp_path / 's.tsv'), None,
                               str(tmp_path / 'n.tsv'), 'X', 'cds', {'-n': 5, '-c': 0.8}, None) is None
    assert 'step-by-step' in capsys.readouterr().out


def test_duplicate_genome_names_and_missing_files(tmp_path):
    d1, d2 = tmp_path / 'a', tmp_path / 'b'
    d1.mkdir(), d2.mkdir()
    for d in (d1, d2):
        (d / 'g.faa').write_text('>x\nMKVLLAAAAAAAAAAAAA\n')
    assert pg._native_pipeline([str(d1 / 'g.faa'), str(d2 / 'g.faa')], str(tmp_path / 'nr.faa'), str(tmp_path / 's.tsv'),
                               None, str(tmp_path / 'n.tsv'), 'X', 'cds', {'-n': 5, '-c': 0.8}, None) is None
    with pytest.raises(_native.PgxError, match='cannot read'):
        _native.FastaSet([str(tmp_path / 'nope.faa')])


def test_lex_key_orders_like_the_names():
    rng = np.random.default_rng(0)
    c = np.concatenate([rng.integers(0, 3000, 500), [0, 1, 9, 10, 11, 99, 100, 101, 999, 1000, 123456789]])
    m = np.concatenate([rng.integers(0, 300, 500), [0, 1, 9, 10, 11, 99, 100, 101, 7, 3, 2]])
    names = ['N_C%dA%d' % cm for cm in zip(c, m)]
    order = np.lexsort((pg._lex_key(m, True), pg._lex_key(c, False)))
    assert [names[i] for i in order] == sorted(names)


def test_native_allele_order_sorts_like_the_names():
    """pgx_allele_order: the row order of the allele table = the names sorted as strings (pangenome.py:615), stable for
    equal pairs, over ranges where the digit counts differ; and it equals the numpy restatement above."""
    from pangenomix_amd import _native
    rng = np.random.default_rng(1)
    for cmax, mmax, n in ((3000, 300, 5000), (12, 4, 200), (2_000_000_000, 150_000, 4000), (1, 1, 3)):
        c = np.concatenate([rng.integers(0, cmax, n), [0, 1, 9, 10, 11, 99, 100, 101, 999, 1000, cmax]]).astype(np.int32)
        m = np.concatenate([rng.integers(0, mmax, n), [0, 1, 9, 10, 11, 99, 100, 101, 7, 3, mmax]]).astype(np.int32)
        names = ['N_C%dA%d' % cm for cm in zip(c.tolist(), m.tolist())]
        order = _native.allele_order(c, m)
        assert sorted(order.tolist()) == list(range(c.size))
        assert [names[i] for i in order] == sorted(names)
        assert order.tolist() == np.lexsort((pg._lex_key(m, True), pg._lex_key(c, False))).tolist()   # (both stable)
    assert _native.allele_order(np.zeros(0, np.int32), np.zeros(0, np.int32)).size == 0
    with pytest.raises(_native.PgxError, match='negative'):
        _native.allele_order(np.array([1, -1], np.int32), np.array([0, 0], np.int32))


def test_native_first_insertions_keep_the_dictionary_order():
    """pgx_first_insertions against a Python dict filled pair by pair (what scipy's dok_matrix is, pangenome.py:649-650)."""
    from pangenomix_amd import _native
    rng = np.random.default_rng(2)
    for n, n_rows, n_cols in ((20000, 300, 7), (5000, 1 << 40, 3), (10, 1, 1), (0, 5, 5)):
        rows = rng.integers(0, n_rows, n)
        cols = rng.integers(0, n_cols, n)
        seen, want = {}, []
        for i, rc in enumerate(zip(rows.tolist(), cols.tolist())):
            if rc not in seen:
                seen[rc] = 1
                want.append(i)
        assert _native.first_insertions(rows, cols, n_cols).tolist() == want
    with pytest.raises(_native.PgxError, match='out of range'):
        _native.first_insertions(np.array([0, 1]), np.array([0, 5]), 5)
    with pytest.raises(_native.PgxError, match='out of range'):
        _native.first_insertions(np.array([-1]), np.array([0]), 5)


def test_format_labels_matches_python_formatting():
    """Feature names from the library (numpy 'U' records written by several threads; the 'S' path for a
    prefix that is not ASCII) equal the reference's string formatting (pangenome.py:1944-1969)."""
    from pangenomix_amd import _native
    rng = np.random.default_rng(4)
    c = np.concatenate([rng.integers(0, 2_000_000, 70000), [0, 9, 10, 99, 100, 1999999]]).astype(np.int32)
    m = np.concatenate([rng.integers(0, 500, 70000), [0, 9, 10, 99, 100, 499]]).astype(np.int32)
    for prefix in ('Test_C', 'Étude_C'):
        alleles = _native.format_labels(prefix, c, m, 'A')
        genes = _native.format_labels(prefix, c)
        assert alleles.dtype.kind == 'U' and genes.dtype.kind == 'U'
        assert alleles.tolist() == ['%s%dA%d' % (prefix, a, b) for a, b in zip(c.tolist(), m.tolist())]
        assert genes.tolist() == ['%s%d' % (prefix, a) for a in c.tolist()]
    assert _native.format_labels('X', np.zeros(0, dtype=np.int32)).shape == (0,)


def test_clustered_outputs_with_member_numbers_that_are_not_consecutive(tmp_path):
    """pgx_fasta_write_clustered places every sequence at (start of its cluster + member number) when a cluster's members
    are numbered 0, 1, 2, ... and sorts otherwise: numbers with gaps, a repeated number, cluster numbers far apart and
    unclustered sequences give the same .clstr as the Python writer (cluster, then member number, then input order)."""
    from pangenomix_amd import _native
    rng = np.random.default_rng(5)
    src = tmp_path / 'g'
    src.mkdir()
    paths = random_fasta_set(rng, str(src), quirks=False)
    with _native.FastaSet(paths) as fs:
        n = fs.n_groups
        headers = fs.headers(fs.rep_of_group)
        lengths = np.diff(fs.offsets.astype(np.int64))
        iden = rng.uniform(0.8, 1.0, n).astype(np.float32)
        cases = {'consecutive': (np.arange(n) // 4, np.arange(n) % 4),
                 'gaps': (np.arange(n) // 4, 2 * (np.arange(n) % 4)),
                 'repeated': (np.arange(n) // 4, np.minimum(np.arange(n) % 4, 2)),
                 'sparse clusters': (1000003 * (np.arange(n) // 4), np.arange(n) % 4),
                 'some unclustered': (np.where(np.arange(n) % 7 == 3, -1, np.arange(n) // 5), np.arange(n) % 5)}
        for what, (cl, mem) in cases.items():
            perm = rng.permutation(n)                      # (the sequences are not in cluster order)
            cl, mem = cl[perm].astype(np.int32), mem[perm].astype(np.int32)
            got, want = str(tmp_path / 'got.clstr'), str(tmp_path / 'want.clstr')
            fs.write_clustered(cl, mem, iden, None, False, 'X_C', 'A', clstr_path=got, names_path=str(tmp_path / 'names.tsv'),
                               nr_out_path=str(tmp_path / 'nr.faa'))
            cluster.write_clstr(want, headers, lengths, cl, mem, iden, np.zeros(n, np.uint8), False)
            assert filecmp.cmp(got, want, shallow=False), what
            names = [ln.split('\t')[0] for ln in open(str(tmp_path / 'names.tsv'))]
            keep = np.flatnonzero(cl >= 0)
            order = keep[np.lexsort((keep, mem[keep], cl[keep]))]
            assert names == ['X_C%dA%d' % (cl[i], mem[i]) for i in order], what
