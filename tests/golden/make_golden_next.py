#!/usr/bin/env python
"""Golden fixtures for the SURVEY 8f-2 / 8f-3 rows, produced by RUNNING THE REFERENCE in the build
container (needs /root/reference; it never travels to the GPU box):

    python tests/golden/make_golden_next.py

  pangenome_analysis.fit_heaps_by_iteration   pangenome_analysis.py:24-48   on the pan/core tables already
                                              held by tests/golden/pancore (themselves reference output)
  plot.calculate_mean                         plot.py:5-43 is matplotlib-bound; only its first statement
                                              (df.mean()) is data, reproduced by pandas directly
  core_genome.count_gene_occurence            core_genome.py:127-155   on tests/golden/cds/expected/*.npz
  core_genome.find_core_genes                 core_genome.py:107-124
  allele_identification.count_allele_occurence allele_identification.py:129-157
  pangenome.build_upstream_pangenome / build_downstream_pangenome (-> build_proximal_pangenome,
  extract_proximal_sequences, consolidate_proximal)  pangenome.py:743-1184   on small GFF + FNA genomes
                                              written here (tests/golden/proximal)

`pangenome_analysis` imports statsmodels.stats and the two consumer modules import Bio (Biopython) at
module level; neither is installed and neither is used by the functions above, so EMPTY placeholder
module objects are registered under those names for the imports to succeed (the technique
tests/golden/make_golden.py already uses). Nothing of them is ever called.
"""
import contextlib
import glob
import io
import json
import os
import sys
import types

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, '/root/reference/pangenomix')
for _name in ('statsmodels', 'statsmodels.stats', 'Bio', 'Bio.SeqIO'):
    sys.modules.setdefault(_name, types.ModuleType(_name))
sys.modules['statsmodels'].stats = sys.modules['statsmodels.stats']
sys.modules['Bio'].SeqIO = sys.modules['Bio.SeqIO']

import pangenomix.pangenome as ref_pg                     # noqa: E402
import pangenomix.pangenome_analysis as ref_pa            # noqa: E402
import pangenomix.core_genome as ref_cg                   # noqa: E402
import pangenomix.allele_identification as ref_ai         # noqa: E402


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def main():
    out = os.path.join(HERE, 'next')
    os.makedirs(out, exist_ok=True)
    n = 0
    for path in sorted(glob.glob(os.path.join(HERE, 'pancore', '*.npz'))):
        z = np.load(path)
        if int(z['shape'][1]) < 2:
            continue                                        # curve_fit (leastsq) refuses fewer points than parameters
        df = pd.DataFrame(z['expected'], index=[str(x) for x in z['index']], columns=[str(x) for x in z['columns']])
        fit = ref_pa.fit_heaps_by_iteration(df)
        np.savez_compressed(os.path.join(out, 'heaps_' + os.path.basename(path)), alpha=fit['alpha'].values,
                            kappa=fit['kappa'].values, index=np.array(fit.index.tolist()),
                            columns=np.array(fit.columns.tolist()), mean=df.mean().values)
        n += 1
    occ = {}
    exp = os.path.join(HERE, 'cds', 'expected')
    g = quiet(ref_cg.count_gene_occurence, os.path.join(exp, 'T_strain_by_gene.npz'))
    a = quiet(ref_ai.count_allele_occurence, os.path.join(exp, 'T_strain_by_allele.npz'))
    occ['gene'] = {'columns': list(g.columns), 'dtypes': [str(t) for t in g.dtypes], 'values': g.values.tolist()}
    occ['allele'] = {'columns': list(a.columns), 'dtypes': [str(t) for t in a.dtypes], 'values': a.values.tolist()}
    occ['core'] = {}
    for k in (1, 3, 5, 6, 7):
        c = ref_cg.find_core_genes(g, k)
        occ['core'][str(k)] = {'columns': list(c.columns), 'dtypes': [str(t) for t in c.dtypes], 'values': c.values.tolist()}
    json.dump(occ, open(os.path.join(out, 'occurrence.json'), 'w'), indent=0)
    print('next: %d heaps tables, occurrence counts of %d genes / %d alleles' % (n, len(g), len(a)))


def make_proximal():
    import shutil
    root = os.path.join(HERE, 'proximal')
    if os.path.exists(root):
        shutil.rmtree(root)
    din, dexp = os.path.join(root, 'in'), os.path.join(root, 'expected')
    os.makedirs(din)
    os.makedirs(dexp)
    rng = np.random.default_rng(77)
    nt = np.array(list('ACGT'))
    base = {c: ''.join(rng.choice(nt, n)) for c, n in (('c1', 900), ('c2', 520))}
    # features: (contig, type, start, stop, strand, peg number); 1-based inclusive coordinates
    feats = [('c1', 'CDS', 20, 140, '+', 1),      # upstream region cut off by the contig start
             ('c1', 'CDS', 200, 320, '+', 2),
             ('c1', 'CDS', 326, 450, '+', 3),     # 5 nt after peg.2: overlap truncation
             ('c1', 'tRNA', 460, 530, '+', None),
             ('c1', 'CDS', 560, 700, '-', 4),
             ('c1', 'CDS', 760, 880, '-', 5),     # downstream of a minus-strand gene near... upstream cut by the contig end
             ('c2', 'CDS', 60, 200, '-', 6),
             ('c2', 'CDS', 260, 400, '+', 7),
             ('c9', 'CDS', 10, 100, '+', 8)]      # contig missing from the FNA
    genomes = ['p1', 'p2', 'p10']
    names = {}
    for gi, g in enumerate(genomes):
        seqs = {c: list(s) for c, s in base.items()}
        if gi >= 1:                                # variants: a change upstream of peg.2 and downstream of peg.7
            seqs['c1'][170] = 'A' if seqs['c1'][170] != 'A' else 'C'
            seqs['c2'][420] = 'G' if seqs['c2'][420] != 'G' else 'T'
        if gi == 2:                                # and a third variant upstream of peg.2
            seqs['c1'][180] = 'T' if seqs['c1'][180] != 'T' else 'G'
        with open(os.path.join(din, g + '.fna'), 'w') as f:
            for c, s_ in seqs.items():
                s_ = ''.join(s_)
                f.write('>%s   contig of %s\n' % (c, g))
                f.write('\n'.join(s_[i:i + 70] for i in range(0, len(s_), 70)) + '\n')
        with open(os.path.join(din, g + '.gff'), 'w') as f:
            f.write('##gff-version 3\n\n')
            for c, t, a, b, st, k in feats:
                fid = 'fig|%s.peg.%d' % (g, k) if k else 'fig|%s.rna.1' % g
                f.write('\t'.join(['accn|' + c, 'PATRIC', t, str(a), str(b), '.', st, '0', 'ID=%s;product=x y' % fid]) + '\n')
                if k and not (g == 'p10' and k == 5):          # p10's peg.5 is not in the name table
                    names.setdefault(k, []).append('%s|fam%d' % (fid, k))
    with open(os.path.join(din, 'T_allele_names.tsv'), 'w') as f:
        for k, heads in sorted(names.items()):
            f.write('T_C%dA0\t%s\n' % (k + 8, '\t'.join(heads[:2])))      # C9.. C16: lexicographic order matters
            if len(heads) > 2:
                f.write('T_C%dA1\t%s\n' % (k + 8, heads[2]))
    pairs = [(os.path.join(din, g + '.gff'), os.path.join(din, g + '.fna')) for g in genomes]
    runs = {'upstream': dict(fn=ref_pg.build_upstream_pangenome, kw={}),
            'downstream': dict(fn=ref_pg.build_downstream_pangenome, kw={}),
            'upstream_ov5': dict(fn=ref_pg.build_upstream_pangenome,
                                 kw=dict(max_overlap=5, include_fragments=True, fna_output_footer='_ov5', name='O'))}
    printed = {}
    for tag, r in runs.items():
        out = os.path.join(dexp, tag)
        os.makedirs(out)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            r['fn'](pairs, os.path.join(din, 'T_allele_names.tsv'), out, **r['kw'])
        printed[tag] = buf.getvalue().replace(din, '<in>').replace(out, '<out>')
        side = 'upstream' if 'up' in tag else 'downstream'
        for g in genomes:                          # the per-genome extracts land next to the inputs: move them
            src = os.path.join(din, 'derived', '%s_%s%s.fna' % (g, side, r['kw'].get('fna_output_footer', '')))
            shutil.move(src, os.path.join(out, os.path.basename(src)))
    shutil.rmtree(os.path.join(din, 'derived'))
    json.dump(printed, open(os.path.join(dexp, 'stdout.json'), 'w'), indent=0)
    print('proximal: %d runs on %d genomes' % (len(runs), len(genomes)))


if __name__ == '__main__':
    main()
    make_proximal()
