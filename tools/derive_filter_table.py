#!/usr/bin/env python
"""Derive the short-word filter table used by pangenomix_amd/cluster.py.

cd-hit takes its filter cut-offs from a hard-coded empirical table
(naa_stat[5][61][4]: per tolerance sheet, per identity 40..100 %, the percentage of the
sequence length that must be covered by shared 5/4/3/2-mers so that `cover` of all pairs
at that identity pass). That table is not available offline, so the
build derives its own from the simplest defensible model: substitutions placed uniformly
at random on a pair of length L=100 with exactly int(c*L) identical positions; the entry
is floor(100 * (1-cover) quantile of shared k-mers / L). Uniform placement minimises the
expected number of intact k-mers for a given identity, so real (clustered) substitution
patterns share at least as many words: the derived cut-offs err on the side of aligning
more pairs, not fewer. Output: the literal pasted into cluster.py (FILTER_TABLE_T2).
"""
import numpy as np


def quantile_pct(rng, c, k, L=100, cover=0.95, trials=20000):
    nm = int(c * L + 1e-9)
    order = np.argsort(rng.random((trials, L)), axis=1)
    match = np.zeros((trials, L), bool)
    np.put_along_axis(match, order[:, :nm], True, axis=1)
    win = np.ones((trials, L - k + 1), bool)
    for t in range(k):
        win &= match[:, t:L - k + 1 + t]
    frac = win.sum(1) / float(L)
    return int(np.floor(100 * np.quantile(frac, 1 - cover, method='lower') + 1e-9))


def main():
    rng = np.random.default_rng(20250117)
    print('FILTER_TABLE_T2 = {  # identity % -> (N=5, N=4, N=3, N=2), cover 0.95')
    for pct in range(40, 101):
        row = tuple(quantile_pct(rng, pct / 100.0, k) for k in (5, 4, 3, 2))
        print('    %d: (%d, %d, %d, %d),' % ((pct,) + row))
    print('}')


if __name__ == '__main__':
    main()
