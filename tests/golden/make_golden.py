#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (needs /root/reference; it never travels to the GPU box):

    python tests/golden/make_golden.py

What is committed is data: the synthetic inputs written here and the outputs the
reference's own functions produced for them. Reference functions exercised
(file:line in /root/reference/pangenomix):

  pangenome.consolidate_seqs              pangenome.py:336-405   (H1)
  pangenome.rename_genes_and_alleles      pangenome.py:453-560   (H2)
  pangenome.build_genetic_feature_tables  pangenome.py:563-680   (H4)
  sparse_utils.LightSparseDataFrame.to_npz sparse_utils.py:295-314 (H5)
  pangenome.extract_noncoding             pangenome.py:1187-1243 (H6)
  pangenome_analysis.estimate_pan_core_size pangenome_analysis.py:51-98 (K3)

The clustering step itself (cd-hit, pangenome.py:425-450) cannot be run here -- the
program is not installed -- so the .clstr files used below are written by this script
(family tag in the header -> cluster), in cd-hit's grammar.

`pangenome_analysis` imports `statsmodels.stats` at module level (:18) and uses it only at
:380, off the hot path; statsmodels is not installed, so an EMPTY placeholder module object
is registered under that name for the import to succeed. Nothing of it is ever called.
"""
import contextlib
import io
import json
import os
import shutil
import sys
import types

import numpy as np
import scipy.sparse

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, '/root/reference')
for _name in ('statsmodels', 'statsmodels.stats'):
    sys.modules.setdefault(_name, types.ModuleType(_name))
sys.modules['statsmodels'].stats = sys.modules['statsmodels.stats']

import pangenomix.pangenome as ref_pg            # noqa: E402
import pangenomix.pangenome_analysis as ref_pa   # noqa: E402
import pangenomix.sparse_utils as ref_su         # noqa: E402

AA = 'ACDEFGHIKLMNPQRSTVWY'


def quiet(fn, *a, **k):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = fn(*a, **k)
    return out, buf.getvalue()


def reset(path):
    if os.path.exists(path):
        shutil.rmtree(path)
    os.makedirs(path)


# ---------------------------------------------------------------------------------------
# CDS fixture: 6 genomes, 130 families, every quirk of SURVEY §8c G-H1/H2/H4/H5
# ---------------------------------------------------------------------------------------
def wrap(seq, width):
    return '\n'.join(seq[i:i + width] for i in range(0, len(seq), width))


def make_cds():
    root = os.path.join(HERE, 'cds')
    reset(root)
    din, dexp = os.path.join(root, 'in'), os.path.join(root, 'expected')
    os.makedirs(din), os.makedirs(dexp)
    rng = np.random.default_rng(7)
    fam_seq = [''.join(rng.choice(list(AA), size=int(rng.integers(30, 90)))) for _ in range(130)]
    genome_names = ['gB', 'gA', 'g10', 'g2', 'gC.v1', 'gD']   # unsorted on purpose; one with a dot
    paths = []
    for gi, gname in enumerate(genome_names):
        recs = []
        fams = rng.choice(130, size=45, replace=False)
        for k, f in enumerate(fams):
            s = fam_seq[f]
            variant = int(rng.integers(0, 3))
            if variant:  # allele: substitute a few sites deterministically per (family, variant)
                r2 = np.random.default_rng(1000 * f + variant)
                s = list(s)
                for p in r2.choice(len(s), size=3, replace=False):
                    s[p] = AA[int(r2.integers(0, 20))]
                s = ''.join(s)
            width = [60, 70, 80, 1000][int(rng.integers(0, 4))]  # same sequence, different wrapping
            hdr = 'fig|%s.peg.%d|fam%d' % (gname, k, f)
            desc = '   hypothetical protein' if k % 3 == 0 else ''
            recs.append('>' + hdr + desc + '\n' + wrap(s, width) + '\n')
        # quirks
        recs.insert(5, '>fig|%s.peg.dup|fam%d\n%s\n' % (gname, fams[0], wrap(fam_seq[fams[0]], 50)))  # within-genome duplicate (maybe)
        recs.insert(9, '>fig|%s.peg.empty|famX some description\n' % gname)                           # header without sequence
        recs.insert(12, '>fig|%s.peg.short|famS\nMKV%s\n' % (gname, AA[gi]))                           # <= 10 aa: absent from .clstr
        recs.insert(20, '>fig|%s.peg.blank|fam%d\n%s\n\n%s\n' % (gname, fams[1], fam_seq[fams[1]][:20], fam_seq[fams[1]][20:]))  # blank line inside
        text = ''.join(recs)
        if gi == 3:
            text = text.rstrip('\n')       # file without trailing newline
        if gi == 4:
            text = 'ACDEFGHIKL\n' + text   # sequence before any header
        p = os.path.join(din, gname + '.faa')
        with open(p, 'w') as f:
            f.write(text)
        paths.append(p)

    nr = os.path.join(dexp, 'T_nr.faa')
    shared = os.path.join(dexp, 'T_redundant_headers.tsv')
    missing = os.path.join(dexp, 'T_missing_headers.txt')
    (groups, miss), out1 = quiet(ref_pg.consolidate_seqs, paths, nr, shared, missing)
    shutil.copy(nr, os.path.join(dexp, 'T_nr.consolidated.faa'))   # before renaming rewrites it
    with open(os.path.join(dexp, 'consolidate_return.json'), 'w') as f:
        json.dump({'groups': [[k.hex(), v] for k, v in groups.items()], 'missing': miss}, f, indent=0)

    # hand-made .clstr over the nr headers: cluster per family tag, in order of first
    # appearance shifted so that numbers >= 10 and >= 100 occur; 'famS' (short) is left out
    nr_records = []
    with open(nr) as f:
        hdr, seq = None, []
        for line in f:
            if line[0] == '>':
                if hdr is not None:
                    nr_records.append((hdr, ''.join(seq)))
                hdr, seq = line[1:].strip(), []
            else:
                seq.append(line.strip())
        nr_records.append((hdr, ''.join(seq)))
    clusters = {}
    for hdr, seq in nr_records:
        fam = hdr.split('|')[-1]
        if fam == 'famS':
            continue
        clusters.setdefault(fam, []).append((hdr, len(seq)))
    clstr = os.path.join(din, 'T_nr.faa.cdhit.clstr')
    with open(clstr, 'w') as f:
        for c, (fam, members) in enumerate(clusters.items()):
            f.write('>Cluster %d\n' % c)
            for m, (hdr, n) in enumerate(members):
                f.write('%d\t%daa, >%s... %s\n' % (m, n, hdr, '*' if m == 0 else 'at 97.50%'))
    assert len(clusters) > 101

    names = os.path.join(dexp, 'T_allele_names.tsv')
    h2a, out2 = quiet(ref_pg.rename_genes_and_alleles, clstr, nr, nr, names, name='T', cluster_type='cds',
                      shared_headers_file=shared, fastasort_path=None)
    with open(os.path.join(dexp, 'header_to_allele.json'), 'w') as f:
        json.dump(h2a, f, indent=0, sort_keys=True)
    (dfa, dfg), out3 = quiet(ref_pg.build_genetic_feature_tables, clstr, paths, 'T', cluster_type='cds',
                             output_format='lsdf', header_to_allele=h2a)
    dfa.to_npz(os.path.join(dexp, 'T_strain_by_allele.npz'))
    dfg.to_npz(os.path.join(dexp, 'T_strain_by_gene.npz'))
    with open(os.path.join(dexp, 'stdout.json'), 'w') as f:
        json.dump({'consolidate': out1, 'rename': out2, 'tables': out3}, f, indent=0)
    print('cds: %d genomes, %d nr records, %d clusters, allele table %s nnz %d' % (
        len(paths), len(nr_records), len(clusters), dfa.shape, dfa.data.nnz))


# ---------------------------------------------------------------------------------------
# non-coding fixture (G-H6)
# ---------------------------------------------------------------------------------------
def make_noncoding():
    root = os.path.join(HERE, 'noncoding')
    reset(root)
    din, dexp = os.path.join(root, 'in'), os.path.join(root, 'expected')
    os.makedirs(din), os.makedirs(dexp)
    rng = np.random.default_rng(11)
    for gname in ('n1', 'n2'):
        contigs = {'ctgA': ''.join(rng.choice(list('ACGT'), size=900)),
                   'ctgB': ''.join(rng.choice(list('ACGTN'), size=400)).lower()}
        with open(os.path.join(din, gname + '.fna'), 'w') as f:
            for c, s in contigs.items():
                f.write('>%s   [%s | some organism]\n%s\n' % (c, gname, wrap(s, 80)))
        rows = [
            '##gff-version 3', '',
            'accn|ctgA\tPATRIC\ttRNA\t11\t85\t.\t+\t0\tID=fig|%s.rna.1;product=tRNA-Ala' % gname,
            'accn|ctgA\tPATRIC\trRNA\t200\t520\t.\t-\t0\tID=fig|%s.rna.2;product=16S' % gname,
            'accn|ctgA\tPATRIC\tCDS\t530\t700\t.\t+\t0\tID=fig|%s.peg.1;product=skipped' % gname,
            'accn|ctgB\tPATRIC\tmisc_binding\t1\t60\t.\t-\t0\tID=fig|%s.rna.3;product=riboswitch' % gname,
            'accn|ctgZ\tPATRIC\ttRNA\t5\t80\t.\t+\t0\tID=fig|%s.rna.4;product=contig missing' % gname,
            'accn|ctgB\tPATRIC\ttranscript\t350\t400\t.\t+\t0\tID=fig|%s.rna.5;product=at the end' % gname,
            'accn|ctgA\tPATRIC\trepeat_region\t1\t50\t.\t+\t0\tID=fig|%s.rep.1' % gname,
        ]
        with open(os.path.join(din, gname + '.gff'), 'w') as f:
            f.write('\n'.join(rows) + '\n')
        for flank, tag in (((0, 0), ''), ((7, 12), '_f7_12')):
            out = os.path.join(dexp, gname + '_noncoding' + tag + '.fna')
            quiet(ref_pg.extract_noncoding, os.path.join(din, gname + '.gff'),
                  os.path.join(din, gname + '.fna'), out, flanking=flank)
    print('noncoding: 2 genomes x 2 flanking settings')


# ---------------------------------------------------------------------------------------
# pan/core fixtures (G-K3)
# ---------------------------------------------------------------------------------------
def make_pancore():
    root = os.path.join(HERE, 'pancore')
    reset(root)
    cases = {
        'basic': dict(G=300, S=12, density=0.3, seed=1, iters=7),
        'one_genome': dict(G=70, S=1, density=0.5, seed=2, iters=3),
        'two_genomes': dict(G=90, S=2, density=0.6, seed=8, iters=4),            # the smallest table a Heaps fit accepts
        'all_ones': dict(G=64, S=9, density=1.1, seed=3, iters=4),
        'zero_row_odd_words': dict(G=131, S=17, density=0.2, seed=4, iters=5),   # G not a multiple of 64
        'wide': dict(G=5000, S=70, density=0.05, seed=5, iters=6),              # > 64 steps: parked-lane wrap
        'core_heavy': dict(G=1030, S=33, density=0.97, seed=6, iters=5),
        'many_words': dict(G=70000, S=8, density=0.4, seed=7, iters=3),          # several waves per stripe
    }
    for name, c in cases.items():
        rng = np.random.default_rng(c['seed'])
        dense = (rng.random((c['G'], c['S'])) < c['density']).astype(np.int64)
        if name == 'zero_row_odd_words':
            dense[5, :] = 0
            dense[130, :] = 1
        coo = scipy.sparse.coo_matrix(dense)
        lsdf = ref_su.LightSparseDataFrame(['g%d' % i for i in range(c['G'])],
                                           ['s%d' % i for i in range(c['S'])], coo)
        np.random.seed(c['seed'])
        df, _ = quiet(ref_pa.estimate_pan_core_size, lsdf, c['iters'])
        # the permutations the reference consumed, regenerated from the same seed
        np.random.seed(c['seed'])
        perms = []
        for _ in range(c['iters']):
            p = np.arange(c['S'])
            np.random.shuffle(p)
            perms.append(p)
        np.savez_compressed(os.path.join(root, name + '.npz'),
                            row=coo.row.astype(np.int32), col=coo.col.astype(np.int32),
                            shape=np.array([c['G'], c['S']], dtype=np.int64), seed=np.int64(c['seed']),
                            perms=np.array(perms, dtype=np.int32).reshape(c['iters'], c['S']),
                            expected=df.values, index=np.array(df.index.tolist()),
                            columns=np.array(df.columns.tolist()))
    print('pancore: %d cases' % len(cases))


if __name__ == '__main__':
    make_cds()
    make_noncoding()
    make_pancore()
