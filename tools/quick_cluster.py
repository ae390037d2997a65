"""GPU box: cluster a named synthetic set, compare with the oracle, print timings and the first
differences (development aid). usage: quick_cluster.py NAME [repeat] [--no-oracle]"""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import oracle
from pangenomix_amd import _native, cluster, synth

name = sys.argv[1]
repeat = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 1
ctx = _native.Context(0)
if name == 'nt':
    res, off, _ = synth.noncoding_set(n_genomes=120, seed=9)
    p = cluster.params_from_cdhit_args({'-n': 5, '-c': 0.8}, 'nt')
else:
    res, off, _ = synth.protein_set(name).nr_arrays()
    p = cluster.params_from_cdhit_args({'-n': 5, '-c': 0.8})
print(name, off.size - 1, 'sequences', res.size, 'residues', flush=True)
for _ in range(repeat):
    t = time.perf_counter()
    got = ctx.cluster_greedy(res, off, p)
    print('gpu %.1f ms, %d clusters, windows %d' % ((time.perf_counter() - t) * 1e3, got[4], got[5]['sweeps']), flush=True)
if '--no-oracle' not in sys.argv:
    t = time.perf_counter()
    want = oracle.cluster_greedy(res, off, p)
    print('oracle %.1f s, %d clusters' % (time.perf_counter() - t, want[4]))
    ok = True
    for g, w, nm in zip(got[:4], want[:4], ('cluster', 'member', 'identity', 'strand')):
        bad = np.flatnonzero(g != w)
        if bad.size:
            ok = False
            print('DIFF', nm, bad.size, 'first', bad[:8], g[bad[:8]], w[bad[:8]])
    gs, ws = dict(got[5]), dict(want[5])
    for k in ws:
        if k in ('sweeps', 'gpu'):
            continue
        if gs[k] != ws[k]:
            ok = False
            print('STAT', k, gs[k], ws[k])
    print('gpu stats', got[5]['gpu'])
    print('PARITY OK' if ok else 'PARITY FAILED')
