"""One-off large parity check: GPU clustering against the CPU oracle on a synthetic set between the test
sizes and the benchmark size. Usage: python tools/parity_check.py [n_genomes] (default 120; the oracle is
single-threaded and its per-sequence cost grows with the table)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np  # noqa: E402
import oracle  # noqa: E402
from pangenomix_amd import _native, cluster, synth  # noqa: E402


def main():
    n_genomes = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    ps = synth.ProteinSet(n_genomes, 4500, 30000, 2000, n_genomes)
    res, off, n_raw = ps.nr_arrays()
    p = cluster.params_from_cdhit_args({'-n': 5, '-c': 0.8})
    print('%d genomes: %d raw, %d non-redundant, %.1f M residues' % (n_genomes, n_raw, off.size - 1, res.size / 1e6), flush=True)
    ctx = _native.Context(0)
    t = time.perf_counter()
    got = ctx.cluster_greedy(res, off, p)
    print('GPU: %.2f s, %d clusters, %d sweeps' % (time.perf_counter() - t, got[4], got[5]['sweeps']), flush=True)
    t = time.perf_counter()
    want = oracle.cluster_greedy(res, off, p)
    print('oracle: %.1f s, %d clusters' % (time.perf_counter() - t, want[4]), flush=True)
    ok = True
    for g, w, name in zip(got[:3], want[:3], ('cluster', 'member', 'identity')):
        same = np.array_equal(g, w)
        ok &= same
        print('%-9s %s' % (name, 'identical' if same else 'DIFFERS at %s' % np.flatnonzero(g != w)[:10]))
    gs, ws = dict(got[5]), dict(want[5])
    for d in (gs, ws):
        d.pop('sweeps'), d.pop('gpu')
    print('counters  %s' % ('identical' if gs == ws else 'DIFFER: %s vs %s' % (gs, ws)))
    ok &= gs == ws and got[4] == want[4]
    ctx.close()
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
