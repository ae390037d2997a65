"""Timing of the nucleotide (cd-hit-est rules, both strands) clustering on the synthetic non-coding
set of SURVEY 8d config 5 (400 genomes x ~90 features). Usage: python tools/nt_bench.py [n_genomes]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np  # noqa: E402
from pangenomix_amd import _native, cluster, synth  # noqa: E402


def main():
    n_genomes = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    res, off, n_raw = synth.noncoding_set(n_genomes=n_genomes, seed=5)
    p = cluster.params_from_cdhit_args({'-n': 5, '-c': 0.8}, 'nt')
    ctx = _native.Context(0)
    print('non-coding set: %d genomes, %d raw, %d non-redundant features, %.2f M nt'
          % (n_genomes, n_raw, off.size - 1, res.size / 1e6), flush=True)
    for rep in range(3):
        ctx.profile(rep == 2)
        ctx.profile_reset()
        t = time.perf_counter()
        out = ctx.cluster_greedy(res, off, p)
        dt = time.perf_counter() - t
        st = out[5]
        print('run %d: %.1f ms, %d clusters, %d sweeps, filter pairs %d, aligned %d (gpu pairs %d, aligned %d)'
              % (rep, dt * 1e3, out[4], st['sweeps'], st['filter_pairs'], st['aligned_pairs'], st['gpu']['pairs'],
                 st['gpu']['aligned']), flush=True)
    for k, v in sorted(ctx.profile_read().items(), key=lambda kv: -kv[1][0]):
        print('  %-24s %9.3f ms %6d launches' % (k, v[0], v[1]))
    ctx.close()


if __name__ == '__main__':
    main()
