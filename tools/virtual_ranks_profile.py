"""GPU box: the record-sharded clustering on W virtual ranks of ONE GPU (threads, each with its own context and replica of
the index; the exchange is a barrier + device copies) with per-kernel HIP-event times per rank: which part of the device
time is split by record and which is replicated on every rank. The ranks share one GPU, so the wall
time says nothing about scaling; the per-kernel sums do. usage: virtual_ranks_profile.py [workload] [world]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import threading  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402
from pangenomix_amd import _native, cluster, synth  # noqa: E402

SHARDED = ('filter_kernel<all>', 'filter_kernel<new>', 'diag_kernel', 'align_kernel')


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else 'cfg-3s'
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    res, off, _ = synth.protein_set(workload).nr_arrays()
    p = cluster.params_from_cdhit_args({'-n': 5, '-c': 0.8})
    torch.cuda.init()
    dev = torch.device('cuda', 0)
    bufs = [cluster.exchange_buffers(world, dev) for _ in range(world)]
    sends, recvs = [b[0] for b in bufs], [b[1] for b in bufs]
    barrier = threading.Barrier(world)
    tables, walls = [None] * world, [0.0] * world

    def all_gather(recv, send, stream):
        slot = next(s_ for s_ in range(cluster.EXCHANGE_SLOTS) if any(send.data_ptr() == b[s_].data_ptr() for b in sends))
        torch.cuda.synchronize()
        barrier.wait()
        for r in range(world):
            recv[r].copy_(sends[r][slot])
        torch.cuda.synchronize()
        barrier.wait()

    def worker(rank):
        ctx = _native.Context(0)
        sp, keep = cluster.shard_params(p, rank, world, sends[rank], recvs[rank], all_gather)
        ctx.cluster_greedy(res, off, sp)                      # sizes the workspace
        ctx.profile(True)
        ctx.profile_reset()
        t = time.perf_counter()
        ctx.cluster_greedy(res, off, sp)
        walls[rank] = time.perf_counter() - t
        tables[rank] = ctx.profile_read()
        ctx.close()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    out = {'workload': workload, 'world': world, 'ranks': []}
    for r, tab in enumerate(tables):
        sharded = sum(v[0] for k, v in tab.items() if k in SHARDED)
        total = sum(v[0] for v in tab.values())
        out['ranks'].append({'rank': r, 'device_ms': total, 'split_by_record_ms': sharded, 'replicated_ms': total - sharded,
                             'replicated_fraction': (total - sharded) / total,
                             'kernels_ms': {k: round(v[0], 3) for k, v in sorted(tab.items(), key=lambda kv: -kv[1][0])}})
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
