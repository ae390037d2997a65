"""One-off checks at benchmark scale: (1) nucleotide rules, 400 genomes, GPU vs oracle; (2) the table-sharded
mode with 2 and 3 virtual ranks on the cfg-3s set vs the single-process GPU result (itself checked against the
oracle by tools/parity_check.py 400)."""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
torch.cuda.init()
import oracle  # noqa: E402
from pangenomix_amd import _native, cluster, synth  # noqa: E402
from test_gpu_cluster_sharded import fold, run_virtual_ranks  # noqa: E402


def same(a, b, what):
    ok = all(np.array_equal(x, y) for x, y in zip(a[:4], b[:4])) and a[4] == b[4]
    sa, sb = dict(a[5]), dict(b[5])
    for d in (sa, sb):
        d.pop('sweeps'), d.pop('gpu')
    print('%-44s %s' % (what, 'identical (clusters, members, identities, strands, counters)' if ok and sa == sb
                        else 'DIFFERS %s %s' % (sa, sb)), flush=True)
    return ok and sa == sb


def main():
    ok = True
    ctx = _native.Context(0)
    res, off, n_raw = synth.noncoding_set(n_genomes=400, seed=5)
    p = cluster.params_from_cdhit_args({'-n': 5, '-c': 0.8}, 'nt')
    ok &= same(ctx.cluster_greedy(res, off, p), oracle.cluster_greedy(res, off, p), 'nucleotide, 400 genomes, GPU vs oracle')
    res, off, _ = synth.protein_set('cfg-3s').nr_arrays()
    p = cluster.params_from_cdhit_args({'-n': 5, '-c': 0.8})
    single = ctx.cluster_greedy(res, off, p)
    ctx.close()
    for world in (2, 3):
        ok &= same(fold(run_virtual_ranks(res, off, p, world)), single, 'cfg-3s, %d virtual ranks vs one process' % world)
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
