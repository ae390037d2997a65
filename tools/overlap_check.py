"""Determinism of the overlapped window loop: the same set clustered several times with the windows overlapping on two
streams and once without (PGX_NO_OVERLAP=1) must give identical clusters, members, identities and counters.
Usage: python tools/overlap_check.py [workload] [repeats] [window]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np  # noqa: E402
from pangenomix_amd import _native, cluster, synth  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else 'cfg-3s'
    repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    res, off, _ = synth.protein_set(workload).nr_arrays()
    p = cluster.params_from_cdhit_args({'-n': 5, '-c': 0.8})
    if len(sys.argv) > 3:
        p.batch_size = int(sys.argv[3])     # window size (many small windows stress the hand-over between the streams)
    ctx = _native.Context(0)
    os.environ['PGX_NO_OVERLAP'] = '1'
    ref = ctx.cluster_greedy(res, off, p)
    del os.environ['PGX_NO_OVERLAP']
    rs = dict(ref[5]); rs.pop('gpu')
    for i in range(repeats):
        got = ctx.cluster_greedy(res, off, p)
        gs = dict(got[5]); gs.pop('gpu')
        assert all(np.array_equal(a, b) for a, b in zip(got[:4], ref[:4])) and got[4] == ref[4], 'run %d differs' % i
        assert gs == rs, (i, gs, rs)
        print('run %d identical: %d clusters, %d posting visits' % (i, got[4], gs['posting_visits']), flush=True)
    ctx.close()


if __name__ == '__main__':
    main()
