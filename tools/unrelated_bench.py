"""Worst case for the discovery rounds: N unrelated random proteins (no families at all), every one its own cluster.
Usage: python tools/unrelated_bench.py [n]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np  # noqa: E402
from pangenomix_amd import _native, cluster  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    rng = np.random.default_rng(7)
    lens = rng.integers(60, 500, n).astype(np.uint64)
    off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    letters = np.frombuffer(b'ACDEFGHIKLMNPQRSTVWY', dtype=np.uint8)
    res = letters[rng.integers(0, 20, int(off[-1]))]
    p = cluster.params_from_cdhit_args({'-n': 5, '-c': 0.8})
    ctx = _native.Context(0)
    for rep in range(2):
        t = time.perf_counter()
        out = ctx.cluster_greedy(res, off, p)
        print('run %d: %.1f ms, %d sequences -> %d clusters, %d windows' % (rep, (time.perf_counter() - t) * 1e3, n, out[4], out[5]['sweeps']),
              flush=True)
    ctx.close()


if __name__ == '__main__':
    main()
