"""oracle/heaps_ref.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the reference's Heaps-law fit (pangenomix/pangenome_analysis.py:24-48):
per iteration of the pan table, scipy.optimize.curve_fit of  kappa * x**alpha  over x = 1..S from
p0 = [0.5, min(y)]. scipy is the reference's own dependency and is installed, so the restatement
calls the same routine; pinned by tests/golden/next/heaps_*.npz, which the reference function itself
produced (tests/golden/make_golden_next.py). Only tests/ may import this.
"""
import numpy as np
import scipy.optimize


def fit_single(y):
    y = np.asarray(y, dtype=np.float64)
    popt, _ = scipy.optimize.curve_fit(lambda x, alpha, kappa: kappa * np.power(x, alpha),
                                       np.arange(1, y.size + 1), y, p0=[0.5, float(y.min())])
    return popt


def fit_rows(pan):
    """(alpha, kappa) per row of a [n_iter, S] pan table."""
    out = np.array([fit_single(row) for row in np.asarray(pan, dtype=np.float64)]).reshape(-1, 2)
    return out[:, 0], out[:, 1]
