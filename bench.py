#!/usr/bin/env python
"""bench.py -- headline benchmark of the pangenomix hot path on MI355X.

One "step" = one pass of the hot path over the 400-genome workload (BASELINE.json
configs[2], synthetic stand-in `cfg-3s`):
  (a) greedy clustering at 0.8 identity of the non-redundant protein set   -> proteins/s
  (b) 1000 pan/core rarefaction iterations on a 150,000 x 400 presence matrix -> iters/s
with inputs resident in HBM when the timed region starts.

  python bench.py [--gpus N --steps K --warmup W] [--workload cfg-3s|cfg-2s|small] [--skip-cluster]

N > 1 is launched by torch.distributed.run (one rank per GPU, RCCL). Units are independent
(pan/core iterations; for clustering each rank runs the whole set: "replicas only" until the
sharded sweep of DESIGN.md lands), so scaling is "weak" and there is no data-path collective.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # gfx950 spec, /opt/skills/guides/MI355X_MICROARCH.md


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default='cfg-3s')
    ap.add_argument('--pancore-genes', type=int, default=150000)
    ap.add_argument('--pancore-iters', type=int, default=1000)
    ap.add_argument('--skip-cluster', action='store_true')
    ap.add_argument('--skip-cpu', action='store_true')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from pangenomix_amd import _native, cluster, synth

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (no CPU fallback)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        dist.init_process_group('nccl', device_id=dev)
    ctx = _native.Context(local_rank)
    info = ctx.device_info()
    if rank == 0:
        log('device:', info)

    # ---- inputs (synthetic, deterministic) -------------------------------------------
    S = 400
    row, col, G = synth.pancore_matrix(args.pancore_genes, S, seed=1)
    n_iter = args.pancore_iters
    np.random.seed(0)
    from pangenomix_amd import pangenome_analysis as pa
    perms = pa.draw_permutations(S, n_iter)
    stride = _native.lib().pgx_bitmap_stride_words(G)
    d_row = torch.from_numpy(row).to(dev)
    d_col = torch.from_numpy(col).to(dev)
    d_bits = torch.zeros((S, stride), dtype=torch.int64, device=dev)
    d_perms = torch.from_numpy(perms).to(dev)
    d_pan = torch.empty((n_iter, S), dtype=torch.int32, device=dev)
    d_core = torch.empty((n_iter, S), dtype=torch.int32, device=dev)
    ws_bytes = _native.lib().pgx_pan_core_workspace_bytes(G, S, n_iter)
    d_ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def pancore_step():
        ctx.presence_bitmap_dev(d_row.data_ptr(), d_col.data_ptr(), row.size, G, S, d_bits.data_ptr(), stream)
        ctx.pan_core_dev(d_bits.data_ptr(), G, S, d_perms.data_ptr(), n_iter, d_pan.data_ptr(),
                         d_core.data_ptr(), d_ws.data_ptr(), ws_bytes, stream)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        pancore_step()
    barrier()
    ctx.profile(True)
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pancore_step()
    barrier()
    dt = time.perf_counter() - t0
    prof = ctx.profile_read()
    ctx.profile(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # parity spot-check of what was just timed (oracle as the checker only)
    pan = d_pan.cpu().numpy()
    core = d_core.cpu().numpy()

    if rank == 0:
        import oracle
        sample = [0, n_iter - 1]
        opan, ocore = oracle.pan_core(row, col, None, G, S, perms[sample])
        assert np.array_equal(pan[sample], opan) and np.array_equal(core[sample], ocore), 'pan/core parity'
        ms_step = dt / args.steps * 1e3
        sweep_ms, sweep_n = prof['pan_core_sweep_kernel']
        sweep_avg = sweep_ms / sweep_n
        words = (G + 63) // 64
        alg_bytes = n_iter * (S * words * 8 + 2 * S * 4) + n_iter * S * 4
        achieved = alg_bytes / (sweep_avg * 1e-3) / 1e9
        cpu = None
        if not args.skip_cpu:
            k = 20
            t1 = time.perf_counter()
            oracle.pan_core(row, col, None, G, S, perms[:k])
            cdt = time.perf_counter() - t1
            cpu = {'value': k / cdt, 'unit': 'pan/core iters/s', 'cores': 1, 'kind': 'port',
                   'sample': '%d of %d iterations, oracle/pancore_ref.c (dense incidence loop of '
                             'pangenome_analysis.py:81-90), same matrix' % (k, n_iter)}
        line = {
            'metric': 'pan/core iters/sec, 400-genome set',
            'value': world * n_iter * args.steps / dt, 'unit': 'iters/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'u64',
            'data': 'synthetic',
            'config': {'workload': 'pan/core 1000 iterations on synthetic %d x %d presence matrix '
                                   '(SURVEY 8d, seed 1)' % (G, S)},
            'roofline': {'bound': 'hbm', 'kernel': 'pan_core_sweep_kernel', 'achieved': achieved,
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': None, 'avg_kernel_ms': sweep_avg, 'algorithmic_bytes': alg_bytes},
            'cpu_baseline': cpu,
            'kernels_ms': {k_: v[0] / max(v[1], 1) for k_, v in prof.items()},
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    ctx.close()


if __name__ == '__main__':
    main()
