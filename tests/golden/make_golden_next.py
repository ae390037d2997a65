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
        if int(z['shape'][1]) < 3:
            continue                                        # curve_fit needs more points than parameters
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


if __name__ == '__main__':
    main()
