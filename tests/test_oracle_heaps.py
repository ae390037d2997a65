"""The Heaps-fit oracle (oracle/heaps_ref.py) against the tables the reference function itself produced
(tests/golden/next), and the host-side pieces of the 8f-2 / 8f-3 rows that need no GPU."""
import glob
import os

import numpy as np
import pandas as pd

from oracle import heaps_ref
from pangenomix_amd import core_genome, plot

HERE = os.path.dirname(os.path.abspath(__file__))


def test_oracle_reproduces_reference_fits():
    paths = sorted(glob.glob(os.path.join(HERE, 'golden', 'next', 'heaps_*.npz')))
    assert len(paths) >= 5
    for path in paths:
        want = np.load(path)
        z = np.load(os.path.join(HERE, 'golden', 'pancore', os.path.basename(path)[6:]))
        half = z['expected'].shape[1] // 2
        alpha, kappa = heaps_ref.fit_rows(z['expected'][:, :half])
        np.testing.assert_allclose(alpha, want['alpha'], rtol=1e-12)
        np.testing.assert_allclose(kappa, want['kappa'], rtol=1e-12)


def test_find_core_genes_and_mean():
    occ = pd.DataFrame({'gene_index': np.array([0, 2, 5], np.int32), 'count': np.array([3, 6, 6], np.int64)})
    core = core_genome.find_core_genes(occ, 6)
    assert core.values.tolist() == [[2, 6], [5, 6]] and list(core.columns) == ['gene_index', 'highest_expression']
    assert core_genome.find_core_genes(occ, 7).shape == (0, 0)
    df = pd.DataFrame([[1.0, 3.0, 1.0, 0.0], [3.0, 5.0, 1.0, 1.0]], columns=['Pan1', 'Pan2', 'Core1', 'Core2'])
    assert plot.calculate_mean(df).values.tolist() == [[2.0, 4.0, 1.0, 0.5]]
