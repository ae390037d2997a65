"""K1 parity on the GPU: libpgx greedy clustering (HIP, through the C ABI) must equal the
CPU oracle bit-for-bit -- cluster numbers, member numbers, float identities and every
instrumentation counter -- on seeded inputs, edge cases, and through the file-level entry
point. (The oracle restates cd-hit; parity with the real program is unpinned, DESIGN.md.)"""
import numpy as np
import pytest

import oracle
from pangenomix_amd import pangenome, sparse_utils, synth
from test_cluster_oracle import AA as AA_LETTERS, mutate, pack, params, rand_seq

pytestmark = pytest.mark.gpu


def assert_same(got, want):
    for g, w, name in zip(got[:3], want[:3], ('cluster', 'member', 'identity')):
        assert np.array_equal(g, w), '%s differs at %s' % (name, np.flatnonzero(g != w)[:10])
    assert got[4] == want[4]
    gs, ws = dict(got[5]), dict(want[5])
    gs.pop('sweeps'), ws.pop('sweeps'), gs.pop('gpu'), ws.pop('gpu')
    assert gs == ws


@pytest.mark.parametrize('name', ['tiny', 'small'])
def test_matches_oracle_on_synthetic_sets(name, gpu_ctx):
    res, off, _ = synth.protein_set(name).nr_arrays()
    p = params()
    assert_same(gpu_ctx.cluster_greedy(res, off, p), oracle.cluster_greedy(res, off, p))


@pytest.mark.parametrize('args', [{'-c': 0.9}, {'-c': 0.7, '-n': 4}, {'-c': 0.97}, {'-c': 0.8, '-n': 3},
                                  {'-c': 0.8, '-b': 5}, {'-c': 0.8, '-l': 30}])
def test_matches_oracle_for_other_thresholds(args, gpu_ctx):
    res, off, _ = synth.protein_set('tiny').nr_arrays()
    p = params(**args)
    assert_same(gpu_ctx.cluster_greedy(res, off, p), oracle.cluster_greedy(res, off, p))


def test_edge_cases(gpu_ctx):
    rng = np.random.default_rng(11)
    s = rand_seq(rng, 300)
    cases = {
        'empty': [],
        'all short': ['MKV', 'ACDEFGHIKL'],
        'one': [s],
        'identical': [s, s, s.lower()],
        'indels': [s, s[:100] + s[103:], s[:50] + 'WWW' + s[50:], s[5:], s[:-7]],
        'low complexity': ['A' * 200, 'A' * 150 + 'C' * 30, 'AC' * 90, rand_seq(rng, 40) + 'A' * 100],
        'ragged': [rand_seq(rng, n) for n in (11, 12, 500, 40, 41, 2000, 11)],
        'ambiguous': [s, s.replace('A', 'X').replace('C', 'B'), s.replace('L', 'J')],
        'long': [rand_seq(rng, 5000)] * 2 + [mutate(rng, rand_seq(rng, 5000), 3)],
        # largest scores the 32-bit key form of the fast alignment path can meet (len1 + len2 = 4000, all W-W = 11)
        'extreme scores': ['W' * 2000, 'W' * 2000, 'W' * 1999 + 'A', 'W' * 1000 + 'C' * 200 + 'W' * 799],
        'fast path limit': [rand_seq(rng, 2001)] * 2 + [rand_seq(rng, 1999)] * 2,
    }
    p = params()
    for name, seqs in cases.items():
        res, off = pack(seqs)
        try:
            assert_same(gpu_ctx.cluster_greedy(res, off, p), oracle.cluster_greedy(res, off, p))
        except AssertionError as e:
            raise AssertionError('%s: %s' % (name, e))


def test_giant_sequences(gpu_ctx):
    """Sequences beyond 32,767 residues (16,383 nucleotides) -- a handful of giant proteins exist, and cd-hit accepts
    them -- take a word table in global memory and 64-bit cells in the diagonal histogram; everything else is the
    ordinary path. Against the oracle, together with ordinary sequences. Only a word that occurs more than 65,535
    times in ONE sequence is refused (multiplicities are kept in 16 bits)."""
    from pangenomix_amd._native import PgxError
    rng = np.random.default_rng(41)
    g = rand_seq(rng, 41000)
    seqs = [g, mutate(rng, g, 1200), mutate(rng, g[:36000], 9000), rand_seq(rng, 33500), g[:33000]]
    seqs += [rand_seq(rng, 300) for _ in range(5)] + [mutate(rng, seqs[-1], 10)]
    res, off = pack(seqs)
    p = params()
    got = gpu_ctx.cluster_greedy(res, off, p)
    assert_same(got, oracle.cluster_greedy(res, off, p))
    assert got[0][0] == got[0][1] == got[0][4] and got[0][3] != got[0][0]
    from test_cluster_oracle import nt_params, rand_nt, revcomp

    def mutate_nt(seq, k):
        s_ = list(seq)
        for pos in rng.choice(len(s_), size=k, replace=False):
            s_[pos] = 'ACGT'[('ACGT'.index(s_[pos]) + 1 + int(rng.integers(0, 3))) % 4]
        return ''.join(s_)
    nt = rand_nt(rng, 21000)
    nts = [nt, mutate_nt(nt, 500), revcomp(nt[:18000]), rand_nt(rng, 17000), rand_nt(rng, 200)]
    res, off = pack(nts)
    assert_same_nt(gpu_ctx.cluster_greedy(res, off, nt_params()), oracle.cluster_greedy(res, off, nt_params()))
    res, off = pack(['A' * 70000, rand_seq(rng, 100)])
    with pytest.raises(PgxError, match='more than 65535 times'):
        gpu_ctx.cluster_greedy(res, off, p)


def test_family_resolved_inside_one_sweep(gpu_ctx):
    """Many near-identical sequences of one length in one sweep: the in-batch resolution
    must still pick the first accepted representative in order."""
    rng = np.random.default_rng(12)
    base = [rand_seq(rng, 180) for _ in range(6)]
    seqs = [mutate(rng, base[k % 6], int(rng.integers(0, 45))) for k in range(900)]   # 0..25 % substituted
    p = params()
    res, off = pack(seqs)
    assert_same(gpu_ctx.cluster_greedy(res, off, p), oracle.cluster_greedy(res, off, p))


@pytest.mark.parametrize('window', [64, 1024, 4096, 0])
def test_several_windows_and_any_window_size(window, gpu_ctx):
    """More sequences than one window holds: representatives created in one window are in the index
    the next one is filtered against. The window size (pgx_cluster_params.batch_size; 0 = default)
    only changes how the work is cut, never the clusters."""
    ps = synth.ProteinSet(30, 500, 800, 150, 77)
    res, off, _ = ps.nr_arrays()
    assert off.size - 1 > 2 * 4096
    p = params()
    p.batch_size = window
    got = gpu_ctx.cluster_greedy(res, off, p)
    if window:
        assert got[5]['sweeps'] >= got[5]['n_clustered'] // window
    assert_same(got, oracle.cluster_greedy(res, off, p))


def test_long_posting_lists_and_repeated_words(gpu_ctx):
    """Index lines hold 11 entries; lists beyond that continue in the overflow pool, re-allocated as
    they grow. Hundreds of representatives that all carry the same words (a shared 60-residue
    domain in otherwise unrelated sequences), and low-complexity members whose words repeat inside the
    query AND inside the representative (multiplicity > 1 on both sides)."""
    rng = np.random.default_rng(21)
    domain = rand_seq(rng, 60)
    seqs = [rand_seq(rng, 150) + domain + rand_seq(rng, int(rng.integers(100, 200))) for _ in range(700)]
    rep = 'ACDEFGHIKLMNPQRSTVWY'
    seqs += [(rep * 12)[:int(n)] for n in rng.integers(150, 240, 40)]                  # every word several times
    seqs += [mutate(rng, (rep * 12)[:200], int(k)) for k in rng.integers(5, 60, 40)]
    seqs += [rand_seq(rng, 80) + 'Q' * int(n) + rand_seq(rng, 80) for n in rng.integers(20, 90, 60)]
    order = rng.permutation(len(seqs))
    res, off = pack([seqs[i] for i in order])
    for window in (256, 0):
        p = params()
        p.batch_size = window
        assert_same(gpu_ctx.cluster_greedy(res, off, p), oracle.cluster_greedy(res, off, p))


def test_short_members_meeting_hundreds_of_new_representatives(gpu_ctx):
    """A member whose marked words fit one 64-word slab accumulates the representatives it meets straight
    into the 256-slot exact table (no bucket pass); with several hundred mutually unrelated sequences
    that share one 22-residue motif -- every one a candidate of every other by word count, none by
    identity -- that table overflows and the walk falls back to the bounded two-pass scheme with residue
    classes. Sequences of 58..64 residues: at most 60 words each."""
    rng = np.random.default_rng(23)
    motif = rand_seq(rng, 22)
    seqs = []
    for _ in range(900):
        n = int(rng.integers(58, 65))
        a = int(rng.integers(0, n - 22 + 1))
        seqs.append(rand_seq(rng, a) + motif + rand_seq(rng, n - 22 - a))
    seqs += [mutate(rng, seqs[int(i)], int(rng.integers(1, 8))) for i in rng.integers(0, 900, 300)]   # some do join
    order = rng.permutation(len(seqs))
    res, off = pack([seqs[i] for i in order])
    for window in (0, 512):
        p = params()
        p.batch_size = window
        got = gpu_ctx.cluster_greedy(res, off, p)
        assert_same(got, oracle.cluster_greedy(res, off, p))
    assert got[4] > 600                                          # (the motif carriers stay apart)


def test_dependency_chains_inside_one_window(gpu_ctx):
    """Members that are each other's candidates by word count but not by identity (a chain of ~77 %
    identical variants): none of them is 'certain' before its predecessors are decided, so the
    discovery rounds leave them to the exact in-order block resolution."""
    rng = np.random.default_rng(22)
    seqs = []
    for fam in range(12):
        cur = rand_seq(rng, 260)
        for step in range(40):
            seqs.append(cur)
            cur = mutate(rng, cur, 26 if step % 3 else 60)     # ~10 % / ~23 % steps: neighbours share words
    res, off = pack(seqs)
    p = params()
    got = gpu_ctx.cluster_greedy(res, off, p)
    assert_same(got, oracle.cluster_greedy(res, off, p))


def test_build_cds_pangenome_end_to_end(tmp_path, gpu_ctx):
    """The reference's entry point, file names and .npz contract, with the clustering step
    on the GPU; cluster counts are cross-checked against the oracle on the same sequences."""
    ps = synth.protein_set('tiny')
    paths = ps.write_faa(str(tmp_path / 'genomes'))
    out = tmp_path / 'out'
    out.mkdir()
    dfa, dfg = pangenome.build_cds_pangenome(paths, str(out), name='Syn')
    for f in ('Syn_nr.faa', 'Syn_nr.faa.cdhit.clstr', 'Syn_allele_names.tsv', 'Syn_redundant_headers.tsv',
              'Syn_missing_headers.txt', 'Syn_strain_by_allele.npz', 'Syn_strain_by_allele.npz.labels.txt',
              'Syn_strain_by_gene.npz', 'Syn_strain_by_gene.npz.labels.txt'):
        assert (out / f).exists(), f
    assert not (out / 'Syn_nr.faa.cdhit').exists()
    back = sparse_utils.read_lsdf(str(out / 'Syn_strain_by_gene.npz'))
    assert back.shape == dfg.shape == (len(dfg.index), 6)
    genes = np.array([pangenome.__get_gene_from_allele__(a) for a in dfa.index])
    A, G = dfa.data.toarray() > 0, dfg.data.toarray() > 0
    for gi, gene in enumerate(dfg.index):      # gene row = OR of its allele rows (pangenome.py:1299-1330)
        assert np.array_equal(A[genes == gene].any(axis=0), G[gi])
    res, off, _ = ps.nr_arrays()
    want = oracle.cluster_greedy(res, off, params())
    assert len(dfg.index) == want[4]
    assert len(dfa.index) == int((want[0] >= 0).sum())


def test_whole_output_equals_the_step_by_step_path_fed_by_the_oracle(tmp_path, gpu_ctx, monkeypatch):
    """Everything at once, byte for byte: build_cds_pangenome() as shipped -- native ingest + de-duplication, clustering
    on the GPU, native .clstr / names / FASTA writers, array tables, .npz -- against the statement-by-statement Python
    path (consolidate_seqs -> cluster_with_cdhit -> rename_genes_and_alleles -> build_genetic_feature_tables, each
    pinned to the reference's own output by tests/test_host_golden.py) with the CPU ORACLE as its clustering step.
    Every file of the output directory must be identical, the .npz files member for member."""
    import filecmp
    import os
    from pangenomix_amd import cluster
    from test_host_golden import assert_same_npz
    ps = synth.ProteinSet(14, 350, 500, 120, 31)
    paths = ps.write_faa(str(tmp_path / 'genomes'))
    out_gpu, out_ref = tmp_path / 'gpu', tmp_path / 'ref'
    out_gpu.mkdir(), out_ref.mkdir()
    pangenome.build_cds_pangenome(paths, str(out_gpu), name='E')

    def oracle_cdhit(fasta_file, cdhit_out, cdhit_args={'-n': 5, '-c': 0.8}):
        headers, residues, offsets, records = cluster.read_fasta_for_clustering(fasta_file)
        cl, mem, iden, strand, n_clusters, _ = oracle.cluster_greedy(residues, offsets, cluster.params_from_cdhit_args(cdhit_args))
        cluster.write_clstr(cdhit_out + '.clstr', headers, np.diff(offsets.astype(np.int64)), cl, mem, iden, strand)
        with open(cdhit_out, 'w') as f:
            for i in np.flatnonzero(mem == 0):
                f.write(records[i])
    monkeypatch.setattr(pangenome, '_native_pipeline', lambda *a, **k: None)      # the step-by-step path
    monkeypatch.setattr(pangenome, 'cluster_with_cdhit', oracle_cdhit)
    pangenome.build_cds_pangenome(paths, str(out_ref), name='E')
    names = sorted(os.listdir(out_ref))
    assert names == sorted(os.listdir(out_gpu)) and len(names) == 9
    for f in names:
        if f.endswith('.npz'):
            assert_same_npz(str(out_gpu / f), str(out_ref / f))
        else:
            assert filecmp.cmp(str(out_gpu / f), str(out_ref / f), shallow=False), f


# ---- K2: nucleotide rules (cd-hit-est), both strands ---------------------------------------------
from test_cluster_oracle import nt_params, rand_nt, revcomp   # noqa: E402


def test_nt_matches_oracle_on_synthetic_noncoding_set(gpu_ctx):
    res, off, n_raw = synth.noncoding_set(n_genomes=40, seed=5)
    for args in ({}, {'-r': 0}, {'-c': 0.9, '-n': 8}, {'-c': 0.97, '-n': 10}):
        p = nt_params(**args)
        assert_same_nt(gpu_ctx.cluster_greedy(res, off, p), oracle.cluster_greedy(res, off, p), args)


def assert_same_nt(got, want, tag=''):
    for g, w, name in zip(got[:4], want[:4], ('cluster', 'member', 'identity', 'strand')):
        assert np.array_equal(g, w), '%s %s differs at %s' % (tag, name, np.flatnonzero(g != w)[:10])
    assert got[4] == want[4]
    gs, ws = dict(got[5]), dict(want[5])
    for d in (gs, ws):
        d.pop('sweeps'), d.pop('gpu')
    assert gs == ws, tag


def test_nt_edge_cases(gpu_ctx):
    rng = np.random.default_rng(31)
    a, b = rand_nt(rng, 400), rand_nt(rng, 1500)
    cases = {
        'strands': [a, revcomp(a[:380]), b, revcomp(b)[3:], b[:1400], revcomp(a)[5:250]],
        'with N': [a, a[:100] + 'NNN' + a[103:390], 'N' * 50 + a[50:300], revcomp(a[:200] + 'N' + a[201:350])],
        'low complexity': ['A' * 200, 'AC' * 100, 'T' * 180, 'ACGT' * 60, 'GT' * 90],
        'short and ragged': [rand_nt(rng, n) for n in (11, 12, 74, 90, 10, 3000, 11)],
        'lower case / U': [a, a.lower(), a.replace('T', 'U')],
        'palindrome': [a[:60] + revcomp(a[:60]), revcomp(a[:60] + revcomp(a[:60]))],
    }
    p = nt_params()
    for name, seqs in cases.items():
        res, off = pack(seqs)
        assert_same_nt(gpu_ctx.cluster_greedy(res, off, p), oracle.cluster_greedy(res, off, p), name)


def test_nt_more_queries_than_one_window(gpu_ctx):
    res, off, _ = synth.noncoding_set(n_genomes=120, seed=9)
    assert off.size - 1 > 2 * 2048          # nucleotide windows hold 2048 queries (two slots each: both strands)
    p = nt_params()
    got = gpu_ctx.cluster_greedy(res, off, p)
    assert got[5]['sweeps'] >= 3
    assert_same_nt(got, oracle.cluster_greedy(res, off, p))


def test_nt_config5_size_400_genomes(gpu_ctx):
    """BASELINE configs[4] at its stated size: the non-coding features of 400 genomes (31 k raw
    records -> ~19 k non-redundant nucleotide sequences), cd-hit-est rules, both strands, against the
    oracle: clusters, members, identities, strands and every counter."""
    res, off, n_raw = synth.noncoding_set(n_genomes=400, seed=5)
    assert n_raw > 30000 and off.size - 1 > 15000
    p = nt_params()
    assert_same_nt(gpu_ctx.cluster_greedy(res, off, p), oracle.cluster_greedy(res, off, p))


def test_fasta_file_entry_point_matches_oracle(tmp_path, gpu_ctx):
    """cluster_with_cdhit() itself, file in -> .clstr out, on a FASTA with what cd-hit's reader
    cleans up: inner and trailing '*', gaps, digits, blanks, lower case, a record
    without sequence, a too-short one. The .clstr must equal the one written from the oracle's
    result on the same file (same reader, same writer: only the clustering differs)."""
    from pangenomix_amd import cluster
    rng = np.random.default_rng(41)
    fams = [rand_seq(rng, int(n)) for n in rng.integers(60, 400, 12)]
    recs = []
    for i in range(150):
        s = mutate(rng, fams[i % 12], int(rng.integers(0, 50)))
        kind = i % 7
        if kind == 0:
            s = s[:30] + '*' + s[30:] + '*'
        elif kind == 1:
            s = s[:10] + '-' * 3 + s[10:50] + '12' + s[50:]
        elif kind == 2:
            s = s.lower()
        elif kind == 3:
            s = s[:20] + ' ' + s[20:] + ' \t'
        wrap = int(rng.integers(40, 90))
        recs.append('>seq%d some text\n%s\n' % (i, '\n'.join(s[j:j + wrap] for j in range(0, len(s), wrap))))
    recs.insert(5, '>empty\n')
    recs.insert(9, '>short\nMKV*LL\n')
    recs.insert(11, '>gappy\n--**--\n')
    fasta = tmp_path / 'in.faa'
    fasta.write_text(''.join(recs))
    pangenome.cluster_with_cdhit(str(fasta), str(fasta) + '.cdhit', {'-n': 5, '-c': 0.8})
    headers, res, off, records = cluster.read_fasta_for_clustering(str(fasta))
    p = params()
    cl, mem, iden, strand, nc, _ = oracle.cluster_greedy(res, off, p)
    want = tmp_path / 'want.clstr'
    cluster.write_clstr(str(want), headers, np.diff(off.astype(np.int64)), cl, mem, iden, strand, False)
    got = (tmp_path / 'in.faa.cdhit.clstr').read_text()
    assert got == want.read_text()
    assert got.count('>Cluster') == nc and 'seq0...' in got and '>empty' not in got and '>short' not in got
    reps = (tmp_path / 'in.faa.cdhit').read_text()
    assert reps.count('>') == nc


def test_build_noncoding_pangenome_end_to_end(tmp_path, gpu_ctx, golden_dir):
    """GFF+FNA -> derived/*_noncoding.fna -> nucleotide clustering on the GPU -> tables; file
    names per the reference (pangenome.py:259-307); extraction checked against its fixture."""
    import filecmp
    import os
    import shutil
    src = os.path.join(golden_dir, 'noncoding', 'in')
    gdir = tmp_path / 'genomes'
    shutil.copytree(src, gdir)
    pairs = sorted(pangenome.find_matching_genome_files(str(gdir), str(gdir)))
    assert [os.path.basename(g) for g, f in pairs] == ['n1.gff', 'n2.gff']
    out = tmp_path / 'out'
    out.mkdir()
    dfa, dfg = pangenome.build_noncoding_pangenome(pairs, str(out), name='NC')
    for g in ('n1', 'n2'):
        assert filecmp.cmp(str(gdir / 'derived' / (g + '_noncoding.fna')),
                           os.path.join(golden_dir, 'noncoding', 'expected', g + '_noncoding.fna'), shallow=False)
    for f in ('NC_noncoding_nr.fna', 'NC_noncoding_nr.fna.cdhit.clstr', 'NC_noncoding_allele_names.tsv',
              'NC_noncoding_redundant_headers.tsv', 'NC_noncoding_missing_headers.txt',
              'NC_strain_by_noncoding_allele.npz', 'NC_strain_by_noncoding_gene.npz',
              'NC_strain_by_noncoding_gene.npz.labels.txt'):
        assert (out / f).exists(), f
    assert list(dfg.columns) == ['n1', 'n2']            # '_noncoding' stripped from the column labels (:292-293)
    assert all(x.startswith('NC_T') for x in dfg.index)
    assert 'nt, >' in open(out / 'NC_noncoding_nr.fna.cdhit.clstr').read()


# ---- BASELINE-size checks ---------------------------------------------------------------------
def test_cfg2s_full_parity_and_idempotence(gpu_ctx):
    """BASELINE configs[1] size (50 genomes x 4,500 CDS synthetic, ~165k non-redundant proteins):
    full bit-exact comparison with the oracle, plus the size-independent property that the
    representatives alone re-cluster into singletons in the same order (idempotence)."""
    res, off, _ = synth.protein_set('cfg-2s').nr_arrays()
    p = params()
    got = gpu_ctx.cluster_greedy(res, off, p)
    assert_same(got, oracle.cluster_greedy(res, off, p))
    cl, mem, iden = got[0], got[1], got[2]
    lens = np.diff(off.astype(np.int64))
    assert (iden[mem > 0] >= np.float32(0.8)).all() and (iden[mem == 0] == 0).all()
    assert ((lens <= 10) == (cl < 0)).all()
    reps = np.flatnonzero(mem == 0)
    reps = reps[np.argsort(cl[reps])]                      # creation order = descending length, stable
    rres = np.concatenate([res[off[i]:off[i + 1]] for i in reps])
    roff = np.zeros(reps.size + 1, dtype=np.uint64)
    np.cumsum(lens[reps].astype(np.uint64), out=roff[1:])
    again = gpu_ctx.cluster_greedy(rres, roff, p)
    assert again[4] == reps.size and (again[1] == 0).all()
    assert np.array_equal(again[0], np.arange(reps.size, dtype=np.int32))


@pytest.mark.slow
def test_cfg3s_full_size_parity(gpu_ctx):
    """The benchmark workload itself (BASELINE configs[2] shape: 400 genomes x 4,500 CDS synthetic,
    1.14 M non-redundant proteins, 276 sweeps): every cluster number, member number, identity and
    counter against the oracle. About 70 s of single-core oracle time. (This size found a pruning
    rule that compared only the upper half of the candidate key: three pairs the one-by-one pass
    examines were never evaluated -- harmless there, they were rejections -- which no smaller set
    exposed.)"""
    res, off, _ = synth.protein_set('cfg-3s').nr_arrays()
    p = params()
    got = gpu_ctx.cluster_greedy(res, off, p)
    assert got[5]['sweeps'] >= 15
    assert_same(got, oracle.cluster_greedy(res, off, p))


@pytest.mark.slow
def test_cfg4_shape_properties_and_prefix_parity(gpu_ctx):
    """BASELINE configs[3] shape on ONE GPU (the 8-GPU run is the driver's): 4000 synthetic genomes x
    3000 CDS -> 7.3 M non-redundant proteins. Full oracle parity would take ~15 min of CPU time, so:
    the size-independent properties used for cfg-2s, plus bit-exact parity with the oracle on the set
    of the first 400 of its genomes."""
    ps = synth.protein_set('cfg-4')
    sub = synth.ProteinSet(400, ps.cds, ps.F, ps.C, ps.seed)
    res, off, _ = sub.nr_arrays()
    p = params()
    assert_same(gpu_ctx.cluster_greedy(res, off, p), oracle.cluster_greedy(res, off, p))
    res, off, n_raw = ps.nr_arrays()
    assert n_raw == 12000000
    got = gpu_ctx.cluster_greedy(res, off, p)
    cl, mem, iden = got[0], got[1], got[2]
    lens = np.diff(off.astype(np.int64))
    assert (iden[mem > 0] >= np.float32(0.8)).all() and (iden[mem == 0] == 0).all()
    assert ((lens <= 10) == (cl < 0)).all()
    reps = np.flatnonzero(mem == 0)
    reps = reps[np.argsort(cl[reps])]                      # creation order = descending length, stable
    assert (np.diff(lens[reps]) <= 0).all()
    assert np.array_equal(np.sort(cl[cl >= 0])[[0, -1]], [0, got[4] - 1])
    rres = np.concatenate([res[off[i]:off[i + 1]] for i in reps])
    roff = np.zeros(reps.size + 1, dtype=np.uint64)
    np.cumsum(lens[reps].astype(np.uint64), out=roff[1:])
    again = gpu_ctx.cluster_greedy(rres, roff, p)          # representatives alone re-cluster into singletons
    assert again[4] == reps.size and (again[1] == 0).all()
    assert np.array_equal(again[0], np.arange(reps.size, dtype=np.int32))


def test_errors_are_reported_and_leave_the_context_usable(gpu_ctx):
    from pangenomix_amd._native import PgxError
    rng = np.random.default_rng(3)
    good = pack([rand_seq(rng, 200) for _ in range(50)])
    p = params()
    want = oracle.cluster_greedy(good[0], good[1], p)
    # a sequence beyond the supported length (4 M residues; giant proteins of tens of thousands are fine: test_giant_sequences)
    res, off = pack([rand_seq(rng, (1 << 22) + 1), rand_seq(rng, 100)])
    with pytest.raises(PgxError, match='exceeds the supported maximum'):
        gpu_ctx.cluster_greedy(res, off, p)
    assert_same(gpu_ctx.cluster_greedy(*good, p), want)
    # parameters out of range
    for field, value, text in (('word_len', 6, 'word_len'), ('identity', 0.2, 'identity'), ('band_width', 100, 'band_width'),
                               ('alphabet', 3, 'alphabet')):
        q = params()
        setattr(q, field, value)
        with pytest.raises(PgxError, match=text):
            gpu_ctx.cluster_greedy(*good, q)
    # a failing exchange callback in the record-sharded mode, in the middle of a run
    import torch
    from pangenomix_amd import cluster
    res, off, _ = synth.ProteinSet(30, 500, 800, 150, 77).nr_arrays()
    send, recv = cluster.exchange_buffers(1, 'cuda:0')
    calls = []

    def failing(r, s_, stream):
        calls.append(1)
        if len(calls) == 2:
            raise RuntimeError('link down')
        with torch.cuda.stream(torch.cuda.ExternalStream(stream)):
            r[0].copy_(s_)
    sp, keep = cluster.shard_params(p, 0, 1, send, recv, failing)
    with pytest.raises(PgxError, match='exchange callback failed'):
        gpu_ctx.cluster_greedy(res, off, sp)
    assert len(calls) == 2
    assert_same(gpu_ctx.cluster_greedy(*good, p), want)


# ---- randomized sweep: family structure, parameters and window size drawn at random -----------------
def _random_families(rng, alphabet, n_fam, max_members, lo, hi):
    letters = list(alphabet)
    seqs = []
    for _ in range(n_fam):
        L = int(rng.integers(lo, hi))
        root = ''.join(rng.choice(letters, size=L))
        seqs.append(root)
        cur = root
        for _ in range(int(rng.integers(0, max_members))):
            src = list(cur if rng.random() < 0.4 else root)          # chains of variants and stars around the root
            for p in rng.choice(len(src), size=int(rng.integers(0, max(1, len(src) * 35 // 100))), replace=False):
                src[p] = letters[int(rng.integers(0, len(letters)))]
            a, b = int(rng.integers(0, 6)), int(rng.integers(0, 6))   # ragged ends
            cur = ''.join(src)[a:len(src) - b]
            if len(cur) < 6:
                cur = root
            seqs.append(cur if alphabet != 'ACGT' or rng.random() < 0.7 else revcomp(cur))
    seqs += [seqs[int(i)] for i in rng.integers(0, len(seqs), len(seqs) // 10)]   # exact duplicates
    if alphabet == 'ACGT':
        seqs += ['ACGTN' * 8, 'N' * 40]
    else:
        seqs += ['A' * 30, 'ACDEF', '']
    order = rng.permutation(len(seqs))
    return [seqs[i] for i in order]


@pytest.mark.parametrize('seed', range(48))
def test_randomized_protein_sets_match_oracle(seed, gpu_ctx):
    rng = np.random.default_rng(1000 + seed)
    c = float(rng.choice([0.7, 0.75, 0.8, 0.85, 0.9, 0.95, 0.97, 1.0]))
    n = int(rng.choice([5, 5, 5, 4, 3, 2])) if c < 0.97 else 5
    extra = {}
    if rng.random() < 0.3:
        extra['-b'] = int(rng.choice([5, 10, 32]))
    if rng.random() < 0.3:
        extra['-l'] = int(rng.choice([4, 20, 60]))
    seqs = _random_families(rng, AA_LETTERS, int(rng.integers(5, 70)), int(rng.integers(1, 40)),
                            int(rng.choice([12, 40, 120])), int(rng.choice([130, 400, 900])))
    res, off = pack(seqs)
    p = params(**{'-c': c, '-n': n}, **extra)
    p.batch_size = int(rng.choice([0, 64, 128, 1024]))
    assert_same(gpu_ctx.cluster_greedy(res, off, p), oracle.cluster_greedy(res, off, p))


@pytest.mark.parametrize('seed', range(24))
def test_randomized_nucleotide_sets_match_oracle(seed, gpu_ctx):
    rng = np.random.default_rng(2000 + seed)
    c = float(rng.choice([0.8, 0.85, 0.9, 0.95, 1.0]))
    n = int(rng.choice([5, 6, 8, 10])) if c >= 0.9 else int(rng.choice([5, 6, 7]))
    seqs = _random_families(rng, 'ACGT', int(rng.integers(4, 30)), int(rng.integers(1, 25)),
                            int(rng.choice([20, 60])), int(rng.choice([150, 500])))
    res, off = pack(seqs)
    p = nt_params(**{'-c': c, '-n': n, '-r': int(rng.integers(0, 2))})
    p.batch_size = int(rng.choice([0, 64, 256]))
    assert_same_nt(gpu_ctx.cluster_greedy(res, off, p), oracle.cluster_greedy(res, off, p), 'seed %d' % seed)


@pytest.mark.parametrize('window', [64, 1024])
def test_overlapped_windows_match_the_serial_loop(window, gpu_ctx, monkeypatch):
    """Consecutive windows overlap on two streams (each with its own counters, best keys, flags and pair records);
    the result must not depend on it: the same call with PGX_NO_OVERLAP=1 and the oracle give the same everything,
    run after run, with hundreds of small windows in flight."""
    res, off, _ = synth.ProteinSet(40, 1500, 4000, 600, 91).nr_arrays()      # 41 k sequences, 5.5 k clusters
    p = params()
    p.batch_size = window
    want = oracle.cluster_greedy(res, off, p)
    monkeypatch.setenv('PGX_NO_OVERLAP', '1')
    assert_same(gpu_ctx.cluster_greedy(res, off, p), want)
    monkeypatch.delenv('PGX_NO_OVERLAP')
    for _ in range(3):
        assert_same(gpu_ctx.cluster_greedy(res, off, p), want)


def test_whole_noncoding_output_equals_the_step_by_step_path_fed_by_the_oracle(tmp_path, gpu_ctx, monkeypatch, golden_dir):
    """The same for build_noncoding_pangenome() (nucleotide rules, both strands): every output file of the shipped
    path equals the statement-by-statement path with the CPU oracle as its clustering step."""
    import filecmp
    import os
    import shutil
    from pangenomix_amd import cluster
    from test_host_golden import assert_same_npz
    outs = {}
    for tag in ('gpu', 'ref'):
        gdir = tmp_path / tag / 'genomes'
        shutil.copytree(os.path.join(golden_dir, 'noncoding', 'in'), gdir)
        pairs = sorted(pangenome.find_matching_genome_files(str(gdir), str(gdir)))
        out = tmp_path / tag / 'out'
        out.mkdir()
        if tag == 'ref':
            def oracle_cdhit(fasta_file, cdhit_out, cdhit_args={'-n': 5, '-c': 0.8}):
                nt = fasta_file[-4:].lower() == '.fna'
                headers, residues, offsets, records = cluster.read_fasta_for_clustering(fasta_file)
                p = cluster.params_from_cdhit_args(cdhit_args, 'nt' if nt else 'aa')
                cl, mem, iden, strand, n_clusters, _ = oracle.cluster_greedy(residues, offsets, p)
                cluster.write_clstr(cdhit_out + '.clstr', headers, np.diff(offsets.astype(np.int64)), cl, mem, iden, strand, nt)
                with open(cdhit_out, 'w') as f:
                    for i in np.flatnonzero(mem == 0):
                        f.write(records[i])
            monkeypatch.setattr(pangenome, '_native_pipeline', lambda *a, **k: None)
            monkeypatch.setattr(pangenome, 'cluster_with_cdhit', oracle_cdhit)
        pangenome.build_noncoding_pangenome(pairs, str(out), name='NC')
        outs[tag] = out
    names = sorted(os.listdir(outs['ref']))
    assert names == sorted(os.listdir(outs['gpu']))
    for f in names:
        if f.endswith('.npz'):
            assert_same_npz(str(outs['gpu'] / f), str(outs['ref'] / f))
        else:
            assert filecmp.cmp(str(outs['gpu'] / f), str(outs['ref'] / f), shallow=False), f


def assert_same_outputs(lean, full, tag=''):
    """A call without work counters against one with them: the four output arrays and the cluster count."""
    assert lean[5] is None and full[5] is not None
    assert lean[4] == full[4], tag
    for i, what in enumerate(('cluster', 'member', 'identity', 'strand')):
        np.testing.assert_array_equal(lean[i], full[i], err_msg='%s %s' % (what, tag))


@pytest.mark.parametrize('seed', range(24))
def test_without_counters_randomized_sets_cluster_the_same(seed, gpu_ctx):
    """stats = NULL (want_stats=False): the passes over a window's new representatives leave out the members that
    cannot gain from them -- final members, members whose best key no new representative can precede (pgx.h). The
    result must not change: the randomized protein and nucleotide sets of the oracle tests above, several window
    sizes, both strands."""
    rng = np.random.default_rng(7000 + seed)
    if seed % 3 < 2:
        c = float(rng.choice([0.7, 0.8, 0.9, 0.95, 1.0]))
        seqs = _random_families(rng, AA_LETTERS, int(rng.integers(5, 70)), int(rng.integers(1, 40)),
                                int(rng.choice([12, 40, 120])), int(rng.choice([130, 400, 900])))
        p = params(**{'-c': c, '-n': int(rng.choice([5, 4, 3])) if c < 0.97 else 5})
        p.batch_size = int(rng.choice([0, 64, 128, 1024]))
    else:
        c = float(rng.choice([0.8, 0.9, 0.95]))
        seqs = _random_families(rng, 'ACGT', int(rng.integers(4, 30)), int(rng.integers(1, 25)),
                                int(rng.choice([20, 60])), int(rng.choice([150, 500])))
        p = nt_params(**{'-c': c, '-n': int(rng.choice([8, 10])) if c >= 0.9 else 6, '-r': int(seed % 2)})
        p.batch_size = int(rng.choice([0, 64, 256]))
    res, off = pack(seqs)
    full = gpu_ctx.cluster_greedy(res, off, p)
    assert_same_outputs(gpu_ctx.cluster_greedy(res, off, p, want_stats=False), full, 'seed %d' % seed)
    if seed % 3 < 2:
        assert_same(full, oracle.cluster_greedy(res, off, p))


@pytest.mark.parametrize('window', [64, 1024, 0])
def test_without_counters_synthetic_genomes_cluster_the_same(window, gpu_ctx):
    ps = synth.ProteinSet(30, 500, 800, 150, 77)
    res, off, _ = ps.nr_arrays()
    p = params()
    p.batch_size = window
    assert_same_outputs(gpu_ctx.cluster_greedy(res, off, p, want_stats=False), gpu_ctx.cluster_greedy(res, off, p))
    res, off, _ = synth.noncoding_set(n_genomes=120, seed=9)
    p = nt_params()
    p.batch_size = window if window != 1024 else 256
    assert_same_outputs(gpu_ctx.cluster_greedy(res, off, p, want_stats=False), gpu_ctx.cluster_greedy(res, off, p), 'nt')


@pytest.mark.slow
def test_without_counters_cfg3s_clusters_the_same(gpu_ctx):
    """The benchmark workload: the call bench.py times (no counters) against the instrumented call, which
    test_cfg3s_full_size_parity pins to the oracle."""
    res, off, _ = synth.protein_set('cfg-3s').nr_arrays()
    p = params()
    assert_same_outputs(gpu_ctx.cluster_greedy(res, off, p, want_stats=False), gpu_ctx.cluster_greedy(res, off, p))


def test_pairs_beyond_the_aligners_largest_slot(gpu_ctx):
    """Pairs of two long sequences: up to 4,000 residues together they fit the 16-lane aligner's 4 KB slots, up to
    8,096 the second pass with 8 KB slots (a window whose longest query exceeds 2,000 residues launches it), beyond
    that the general one-pair-per-wave aligner -- and a short query against a much longer, older representative takes
    the general aligner from any window. Families at each of these sizes, real members and near misses, against the
    oracle (clusters, identities, counters)."""
    rng = np.random.default_rng(77)
    seqs = []
    for length, n_members in ((1900, 4), (2050, 5), (2600, 5), (3900, 4), (4100, 4), (5200, 3)):
        base = rand_seq(rng, length)
        seqs.append(base)
        for m in range(n_members):
            frac = (0.05, 0.12, 0.19, 0.24, 0.30)[m % 5]                 # around the 0.8 threshold
            member = mutate(rng, base, int(frac * length))
            seqs.append(member[:length - int(rng.integers(0, length // 10))])
        seqs.append(base[:400])                                          # a fragment: short query, long representative
        seqs.append(mutate(rng, base[200:900], 60))
    seqs += [rand_seq(rng, int(n)) for n in rng.integers(60, 900, 40)]
    order = rng.permutation(len(seqs))
    res, off = pack([seqs[i] for i in order])
    for window in (0, 64):
        p = params()
        p.batch_size = window
        got = gpu_ctx.cluster_greedy(res, off, p)
        assert_same(got, oracle.cluster_greedy(res, off, p))
        lean = gpu_ctx.cluster_greedy(res, off, p, want_stats=False)
        for i in range(4):
            np.testing.assert_array_equal(lean[i], got[i])
    assert got[5]['aligned_pairs'] > 30


def test_word_list_kernels_at_their_class_boundaries(gpu_ctx):
    """The word lists come from four kernels by word count (<= 512 and <= 1023: one wave per sequence with code and
    count in one 32-bit slot; <= 2048: a workgroup per sequence; beyond: sorted). Sequences with exactly 511 .. 514,
    1022 .. 1025 and 2047 .. 2050 words, and low-complexity ones whose single word reaches a count of 512, 1023 and
    1024 -- the largest a ten-bit count holds and the first that must not take that path -- each with a near copy
    that has to find it."""
    rng = np.random.default_rng(5)
    seqs = []
    for words in (511, 512, 513, 514, 1022, 1023, 1024, 1025, 2047, 2048, 2049, 2050):
        s = rand_seq(rng, words + 4)
        seqs += [s, mutate(rng, s, max(3, words // 12))]
    for words in (512, 1023, 1024):
        seqs += ['A' * (words + 4), 'A' * (words + 2) + 'C' + 'A', 'AC' * ((words + 4) // 2)]
    order = rng.permutation(len(seqs))
    res, off = pack([seqs[i] for i in order])
    p = params()
    got = gpu_ctx.cluster_greedy(res, off, p)
    assert_same(got, oracle.cluster_greedy(res, off, p))
    assert got[4] < len(seqs)                       # the near copies joined
