#!/usr/bin/env python
"""bench.py -- headline benchmark of the pangenomix hot path on MI355X.

One "step" = one pass of the hot path over one workload, inputs resident in HBM when the timed
region starts:
  (a) K1  greedy clustering at 0.8 identity of the non-redundant protein set  -> proteins/s  (M1)
  (b) K3  1000 pan/core rarefaction iterations on a 150,000 x 400 presence matrix -> iters/s (M2)

  python bench.py [--gpus N --steps K --warmup W] [--workload cfg-3s|cfg-4|cfg-2s|small|tiny]

N = 1: the 400-genome workload the metric is quoted on (BASELINE.json configs[2]; the Bacteroides
files are not available offline, so the deterministic synthetic stand-in `cfg-3s` of SURVEY.md 8d:
400 genomes x 4,500 CDS). N > 1, one rank per GPU over RCCL: ONE clustering job of the 4000-genome
shape `cfg-4` (configs[3]) split over the ranks by record -- window member i is filtered and aligned
by rank i % N, best keys all-gathered over xGMI -- and the pan/core iterations split by iteration:
"strong" scaling; rank 0 also runs the same job unsharded once, so the line carries `one_gpu_ms` and
`speedup_vs_one_gpu` measured in the same run. `--workload cfg-4` at N = 1 gives the series' first
point with the same workload. `--shard replicas` (explicit) runs N independent copies instead.

Launch: `python bench.py --gpus N` starts the N ranks itself (a child `python -m
torch.distributed.run --nproc-per-node N ... bench.py ...`, before this process touches a GPU, and
relays its JSON line and exit code); under a launcher (RANK / WORLD_SIZE in the environment) it is
one of the ranks and WORLD_SIZE must equal --gpus.

`value` is proteins/s of (a): sequences handed to the clustering call / its time. Beside the timed
steps, rank 0 measures once each (reported, never part of `value`): the host-pointer entry point
(H2D included), M2 through estimate_pan_core_size() (permutations, upload, DataFrame included), the
end-to-end build_cds_pangenome() wall time, and the CPU oracle on a bounded sample.
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # gfx950 spec, /opt/skills/guides/MI355X_MICROARCH.md
L2_PEAK_GBS = 34500.0   # aggregate L2, same guide
PMC_SUMMARY = os.path.join('profiles', 'r03_pmc_summary_cfg3s.json')


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cluster_terms(st, bits=5):
    """The terms of B_cluster from the sequential-rule counters (b = 5 bits per residue, the
    packed layout the alignment kernel reads)."""
    b = bits
    return {'queries': (b * st['sum_len_queries'] + 7) // 8, 'postings': 4 * st['posting_visits'],
            'aligned_reps': (b * st['aligned_rep_len'] + 7) // 8, 'rep_residues': (b * st['sum_len_reps'] + 7) // 8,
            'rep_words': 4 * st['rep_words'], 'outputs': 12 * st['n_clustered']}


def cdhit_reference(sres, soff, got_cluster, got_member, workdir):
    """if a real cd-hit is on PATH, run it on the CPU sample exactly as the reference does
    (pangenome.py:444-447: `cd-hit -i nr.faa -o nr.faa.cdhit -d 0 -n 5 -c 0.8`; one thread, default -M 800) and once
    more with `-T 0 -M 0` (all cores, unchunked), and compare the membership with the GPU result on the same
    sequences. Returns None when the program is absent (the usual case: it cannot be installed offline)."""
    import subprocess
    exe = shutil.which('cd-hit')
    if exe is None:
        return None
    n = soff.size - 1
    faa = os.path.join(workdir, 'sample_nr.faa')
    text = sres.tobytes().decode('ascii')
    with open(faa, 'w') as f:
        for i in range(n):
            f.write('>s%d\n%s\n' % (i, text[int(soff[i]):int(soff[i + 1])]))
    rep_got = np.full(n, -1, dtype=np.int64)                 # per sequence: the representative of its cluster
    reps = np.flatnonzero(got_member == 0)
    by_cluster = np.full(int(got_cluster.max()) + 2, -1, dtype=np.int64)
    by_cluster[got_cluster[reps]] = reps
    rep_got[got_cluster >= 0] = by_cluster[got_cluster[got_cluster >= 0]]
    out = {'path': exe}
    for tag, more in (('as_reference', []), ('all_cores_unchunked', ['-T', '0', '-M', '0'])):
        o = os.path.join(workdir, 'sample_nr.faa.cdhit.' + tag)
        t = time.perf_counter()
        rc = subprocess.call([exe, '-i', faa, '-o', o, '-d', '0', '-n', '5', '-c', '0.8'] + more,
                             stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dt = time.perf_counter() - t
        if rc != 0 or not os.path.exists(o + '.clstr'):
            out[tag] = {'error': 'cd-hit exited with %d' % rc}
            continue
        rep_ref = np.full(n, -1, dtype=np.int64)
        members, rep = [], -1
        with open(o + '.clstr') as f:
            for line in f:
                if line.startswith('>'):
                    for m_ in members:
                        rep_ref[m_] = rep
                    members, rep = [], -1
                    continue
                tok = line.split()
                idx = int(tok[2][2:].rstrip('.'))              # '>s<i>...'
                members.append(idx)
                if tok[-1] == '*':
                    rep = idx
        for m_ in members:
            rep_ref[m_] = rep
        differ = int((rep_ref != rep_got).sum())
        out[tag] = {'value': n / dt, 'unit': 'proteins/s', 'seconds': dt, 'clusters': int((rep_ref == np.arange(n)).sum()),
                    'membership_equal': differ == 0, 'sequences_in_a_different_cluster': differ}
    return out


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD process (never exec: this
    process must not have touched a GPU, and it has not) and relay its output and exit code."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(('127.0.0.1', 0))
        port = s_.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log('bench.py: starting %d ranks: %s' % (n, ' '.join(cmd)))
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default=None, help='default: cfg-3s on one GPU, cfg-4 on several')
    ap.add_argument('--pancore-genes', type=int, default=150000)
    ap.add_argument('--pancore-iters', type=int, default=1000)
    ap.add_argument('--cpu-sample-genomes', type=int, default=100,
                    help='CPU baseline: the oracle on the non-redundant set of the first K genomes (bounded sample)')
    ap.add_argument('--skip-cpu', action='store_true')
    ap.add_argument('--skip-e2e', action='store_true', help='skip the end-to-end build_cds_pangenome() measurement')
    ap.add_argument('--skip-cfg4', action='store_true',
                    help='N = 1 only: skip the one-GPU run of the 4000-genome shape (the reference point of the N > 1 series)')
    ap.add_argument('--e2e-genomes', type=int, default=0, help='genomes of the end-to-end run (0 = the whole workload)')
    ap.add_argument('--shard', choices=['records', 'replicas'], default='records',
                    help='N > 1: ONE clustering job sharded by record over the ranks (default, strong scaling) or N '
                         'independent replicas (weak scaling)')
    ap.add_argument('--exchange', choices=['torch', 'native'], default='torch',
                    help='the all-gather of the record-sharded mode: torch.distributed (nccl backend) through the '
                         'callback, or the library\'s own ncclAllGather (pgx_rccl_*); RCCL over xGMI either way')
    ap.add_argument('--only', choices=['all', 'cluster', 'pancore'], default='all',
                    help='restrict the step (used for rocprofv3 counter passes); the JSON line needs all')
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit('--gpus must be at least 1')
    if 'WORLD_SIZE' not in os.environ and 'RANK' not in os.environ:
        if args.gpus > 1:
            raise SystemExit(launch_ranks(args.gpus))
    elif int(os.environ.get('WORLD_SIZE', '1')) != args.gpus:
        raise SystemExit('bench.py: launched with WORLD_SIZE=%s but --gpus %d: they must agree (the line reports '
                         'n_gpus = the ranks that really ran)' % (os.environ.get('WORLD_SIZE'), args.gpus))

    import torch
    import torch.distributed as dist
    from pangenomix_amd import _native, cluster, synth
    from pangenomix_amd import pangenome_analysis as pa

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if torch.cuda.device_count() < max(world, local_rank + 1):   # (counting devices does not initialise one)
        raise SystemExit('bench.py: %d rank(s) asked for, %d GPU(s) visible: one GPU per rank is needed (no CPU '
                         'fallback, no two ranks on one card)' % (world, torch.cuda.device_count()))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (there is no CPU fallback)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    use_dist = world > 1 or 'RANK' in os.environ   # (torchrun with one rank rehearses RCCL)
    if use_dist:
        # (librccl prints a version banner on STDOUT when it initialises: it goes to stderr, stdout stays the one JSON line)
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group('nccl', device_id=dev)
            t_ = torch.zeros(1, device=dev)
            dist.all_reduce(t_)
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    sharded = use_dist and args.shard == 'records'
    workload = args.workload or ('cfg-4' if world > 1 else 'cfg-3s')
    ctx = _native.Context(local_rank)
    if rank == 0:
        log('device:', ctx.device_info())
    stream = torch.cuda.current_stream().cuda_stream

    # ---- inputs (synthetic, deterministic), made resident in HBM ---------------------------
    t0 = time.perf_counter()
    pset = synth.protein_set(workload)
    res, off, n_raw = pset.nr_arrays(progress=200 if rank == 0 else None)
    n_nr = off.size - 1
    params = base_params = cluster.params_from_cdhit_args({'-n': 5, '-c': 0.8})
    if sharded:   # every rank holds the same sequences; window member i belongs to rank i % world
        if args.exchange == 'native':
            cluster.native_comm(ctx, dist.group.WORLD)
            params = cluster.native_shard_params(params, rank, world)
        else:
            send, recv = cluster.exchange_buffers(world, dev)
            params, _keep = cluster.shard_params(params, rank, world, send, recv, cluster.group_all_gather(dist.group.WORLD))
    d_res = torch.from_numpy(res.copy()).to(dev)
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    S = 400
    row, col, G = synth.pancore_matrix(args.pancore_genes, S, seed=1)
    n_iter = args.pancore_iters
    np.random.seed(0)
    perms = pa.draw_permutations(S, n_iter)
    it_lo, it_hi = pa.shard_bounds(n_iter, rank, world) if sharded else (0, n_iter)
    my_iter = it_hi - it_lo
    stride = _native.lib().pgx_bitmap_stride_words(G)
    d_row, d_col = torch.from_numpy(row).to(dev), torch.from_numpy(col).to(dev)
    d_bits = torch.zeros((S, stride), dtype=torch.int64, device=dev)
    d_perms = torch.from_numpy(perms[it_lo:it_hi].copy()).to(dev)
    d_pan = torch.empty((max(my_iter, 1), S), dtype=torch.int32, device=dev)
    d_core = torch.empty((max(my_iter, 1), S), dtype=torch.int32, device=dev)
    ws_bytes = _native.lib().pgx_pan_core_workspace_bytes(G, S, max(my_iter, 1))
    d_ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    if rank == 0:
        log('inputs: %s -> %d raw records, %d non-redundant, %.1f M residues; pan/core %d x %d; %.1f s'
            % (workload, n_raw, n_nr, res.size / 1e6, G, S, time.perf_counter() - t0))

    last = {}
    wake = torch.empty(256, device=dev)

    def step(profile_cluster=False):
        t = time.perf_counter()
        ctx.profile(profile_cluster)   # per-kernel events on ~10^3 launches would perturb the timed run
        if args.only != 'pancore':
            last['cluster'] = ctx.cluster_greedy_dev(d_res.data_ptr(), d_off.data_ptr(), n_nr, res.size, params,
                                                     stream)
        # The call returns with its results on the host: nothing of it is in flight any more. Its last
        # milliseconds are host-only (outputs), the GPU drops into an idle state meanwhile, and the first
        # kernel after it can need 10-25 ms to start. One trivial kernel + synchronize bring the GPU back;
        # their time is charged to the clustering (conservative), not to whatever runs next.
        wake.zero_()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ctx.profile(True)              # three launches: the events bracket the pan/core kernels live
        if args.only != 'cluster' and my_iter:
            ctx.presence_bitmap_dev(d_row.data_ptr(), d_col.data_ptr(), row.size, G, S, d_bits.data_ptr(), stream)
            ctx.pan_core_dev(d_bits.data_ptr(), G, S, d_perms.data_ptr(), my_iter, d_pan.data_ptr(),
                             d_core.data_ptr(), d_ws.data_ptr(), ws_bytes, stream)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ctx.profile(False)
        return t1 - t, t2 - t1

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ctx.profile_reset()
    t_cluster = t_pancore = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        a, b = step()
        t_cluster += a
        t_pancore += b
    barrier()
    dt = time.perf_counter() - t0
    prof = ctx.profile_read()
    # one more, untimed, step with events on every clustering kernel: the per-kernel table
    ctx.profile_reset()
    step(profile_cluster=True)
    prof_all = ctx.profile_read()
    ctx.profile(False)
    if use_dist:
        t = torch.tensor([dt, t_cluster, t_pancore], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, t_cluster, t_pancore = (float(x) for x in t.tolist())
    if sharded and args.only != 'pancore':   # work counters and identities are partial per rank

        def host_reduce(op):
            def f(a):
                tt = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
                dist.all_reduce(tt, op=op)
                return tt.cpu().numpy().astype(a.dtype)
            return f
        last['cluster'] = cluster.merge_shard_results(last['cluster'], host_reduce(dist.ReduceOp.SUM),
                                                      host_reduce(dist.ReduceOp.MAX))

    if args.only != 'all':
        if rank == 0:
            log('--only %s: %d step(s) done in %.3f s (no JSON line)' % (args.only, args.steps, dt))
            for k_, v in sorted(prof_all.items(), key=lambda kv: -kv[1][0]):
                log('  %-24s %9.3f ms %6d launches' % (k_, v[0], v[1]))
        if use_dist:
            dist.destroy_process_group()
        ctx.close()
        return

    # ---- once-only measurements beside the timed steps (rank 0; the other ranks wait at the barrier) ------
    extra = {}
    if rank == 0:
        import oracle
        import scipy.sparse
        from pangenomix_amd import pangenome, sparse_utils
        steps = args.steps
        cl, mem, iden, _, n_clusters, st = last['cluster']
        # parity spot checks of what was just timed (the oracle is the checker only)
        if my_iter:
            pan, core = d_pan.cpu().numpy(), d_core.cpu().numpy()
            sample = [0, my_iter - 1]
            opan, ocore = oracle.pan_core(row, col, None, G, S, perms[it_lo:it_hi][sample])
            assert np.array_equal(pan[sample], opan) and np.array_equal(core[sample], ocore), 'pan/core parity'
        # clustering: size-independent properties of the result that was timed, and oracle parity on a prefix
        lens = np.diff(off.astype(np.int64))
        assert ((lens <= 10) == (cl < 0)).all(), 'discard rule'
        assert (iden[mem > 0] >= np.float32(0.8)).all() and (iden[mem == 0] == 0).all(), 'identities'
        assert int(cl.max()) + 1 == n_clusters == int((mem == 0).sum()), 'cluster numbering'
        rep_of = np.full(n_clusters, -1, dtype=np.int64)
        rep_of[cl[mem == 0]] = np.flatnonzero(mem == 0)
        assert (lens[rep_of[cl[cl >= 0]]] >= lens[cl >= 0]).all(), 'a representative is the longest member'

        if not sharded:   # (the multi-rank run keeps rank 0's extra work short: the others wait)
            t = time.perf_counter()
            hp = ctx.cluster_greedy(res, off, base_params)
            extra['host_pointer_ms'] = (time.perf_counter() - t) * 1e3
            assert np.array_equal(hp[0], cl) and np.array_equal(hp[1], mem), 'host-pointer entry point differs'
            # the call as the pipeline makes it, without the work counters (stats = NULL: the library leaves out the
            # look-ups only they need; `value` is the instrumented call, whose counters are the roofline's terms)
            lean_best = None
            for _ in range(3):
                t = time.perf_counter()
                lean = ctx.cluster_greedy_dev(d_res.data_ptr(), d_off.data_ptr(), n_nr, res.size, base_params, stream,
                                              want_stats=False)
                wake.zero_()
                torch.cuda.synchronize()
                e = (time.perf_counter() - t) * 1e3
                lean_best = e if lean_best is None else min(lean_best, e)
            assert lean[5] is None and np.array_equal(lean[0], cl) and np.array_equal(lean[1], mem) \
                and np.array_equal(lean[2], iden), 'the call without counters clusters differently'
            extra['without_counters_ms'] = lean_best
            coo = scipy.sparse.coo_matrix((np.ones(row.size, dtype=np.int64), (row, col)), shape=(G, S))
            lsdf = sparse_utils.LightSparseDataFrame(['g%d' % i for i in range(G)], ['s%d' % i for i in range(S)], coo)
            best = None
            with open(os.devnull, 'w') as null:
                for _ in range(3):
                    so, sys.stdout = sys.stdout, null
                    try:
                        t = time.perf_counter()
                        np.random.seed(0)
                        df = pa.estimate_pan_core_size(lsdf, n_iter, ctx=ctx)
                        e = time.perf_counter() - t
                    finally:
                        sys.stdout = so
                    best = e if best is None else min(best, e)
            assert np.array_equal(df.values[[0, n_iter - 1], :S], d_pan.cpu().numpy()[[0, n_iter - 1]]), 'entry point differs'
            extra['pan_core_entry_point'] = {'value': n_iter / best, 'unit': 'iters/s', 'ms': best * 1e3,
                                             'what': 'estimate_pan_core_size(df_genes, %d): np.random.seed(0), value check, %d '
                                                     'legacy-RNG shuffles, upload, bitmap + curves on the device, DataFrame'
                                                     % (n_iter, n_iter)}
        if not args.skip_e2e and not sharded:
            sub = pset if not args.e2e_genomes else synth.ProteinSet(args.e2e_genomes, pset.cds, pset.F, pset.C, pset.seed)
            tmp = tempfile.mkdtemp(prefix='pgx_e2e_')
            try:
                paths = sub.write_faa(os.path.join(tmp, 'genomes'))
                os.mkdir(os.path.join(tmp, 'out'))
                with open(os.devnull, 'w') as null:
                    so, sys.stdout = sys.stdout, null
                    try:
                        t = time.perf_counter()
                        dfa, dfg = pangenome.build_cds_pangenome(paths, os.path.join(tmp, 'out'), name='Bench')
                        e2e = time.perf_counter() - t
                    finally:
                        sys.stdout = so
                # ... and the pan/core curves of the table it returned, from the bitmap the pipeline left on the device
                with open(os.devnull, 'w') as null:
                    so, sys.stdout = sys.stdout, null
                    try:
                        handoff = None
                        for _ in range(3):
                            np.random.seed(0)
                            t = time.perf_counter()
                            pa.estimate_pan_core_size(dfg, n_iter)
                            e = time.perf_counter() - t
                            handoff = e if handoff is None else min(handoff, e)
                    finally:
                        sys.stdout = so
                raw = sub.n_genomes * sub.cds
                extra['end_to_end'] = {'value': raw / e2e, 'unit': 'raw records/s', 'seconds': e2e, 'genomes': sub.n_genomes,
                                       'pan_core_from_resident_bitmap_ms': handoff * 1e3,
                                       'pan_core_handoff': 'estimate_pan_core_size(df_genes, %d) on the table build_cds_pangenome() '
                                                           'returned: %d genes x %d genomes, bitmap built on the device from the '
                                                           'clustering result and still resident (no upload of the table)'
                                                           % (n_iter, int(dfg.shape[0]), int(dfg.shape[1])),
                                       'raw_records': raw, 'genes': int(dfg.shape[0]), 'alleles': int(dfa.shape[0]),
                                       'what': 'build_cds_pangenome(): FASTA files in, dedupe, clustering, naming, tables, '
                                               '.npz out (page cache warm)'}
                if sub is pset:
                    assert dfg.shape[0] == n_clusters, 'end-to-end gene count differs from the clustering'
            finally:
                shutil.rmtree(tmp, ignore_errors=True)

        if sharded and args.only == 'all':
            # The same job, unsharded, on rank 0's GPU in the same run (the other ranks wait at the barrier):
            # the reference point of this line's value, and a parity check of the folded multi-rank result.
            ctx.profile(False)
            best1, out1 = None, None
            for _ in range(2):                               # (the first run sizes the second window set)
                torch.cuda.synchronize()
                t = time.perf_counter()
                out1 = ctx.cluster_greedy_dev(d_res.data_ptr(), d_off.data_ptr(), n_nr, res.size, base_params, stream)
                torch.cuda.synchronize()
                e = time.perf_counter() - t
                best1 = e if best1 is None else min(best1, e)
            assert np.array_equal(out1[0], cl) and np.array_equal(out1[1], mem) and np.array_equal(out1[2], iden), \
                'record-sharded result differs from the one-GPU result'
            s1, sN = dict(out1[5]), dict(st)
            for d_ in (s1, sN):
                d_.pop('sweeps'), d_.pop('gpu')
            assert s1 == sN, 'record-sharded counters differ from the one-GPU counters'
            extra['one_gpu_ms'] = best1 * 1e3
        if world == 1 and not use_dist and workload == 'cfg-3s' and not args.skip_cfg4 and args.only == 'all':
            # The N > 1 series splits ONE job of the 4000-genome shape (cfg-4) over the ranks: its one-GPU point,
            # measured here so that the series has its reference in the same record (inputs resident, as above).
            t = time.perf_counter()
            p4 = synth.protein_set('cfg-4')
            r4, o4, raw4 = p4.nr_arrays(progress=1000)
            log('inputs: cfg-4 -> %d non-redundant of %d raw records, %.1f s' % (o4.size - 1, raw4, time.perf_counter() - t))
            d_r4 = torch.from_numpy(r4.copy()).to(dev)
            d_o4 = torch.from_numpy(o4.view(np.int64)).to(dev)
            ctx.profile(False)
            best4, out4 = None, None
            for _ in range(3):                               # the first run sizes the workspace
                torch.cuda.synchronize()
                t = time.perf_counter()
                out4 = ctx.cluster_greedy_dev(d_r4.data_ptr(), d_o4.data_ptr(), o4.size - 1, r4.size, base_params, stream)
                torch.cuda.synchronize()
                e = time.perf_counter() - t
                best4 = e if best4 is None else min(best4, e)
            l4 = np.diff(o4.astype(np.int64))
            assert ((l4 <= 10) == (out4[0] < 0)).all() and (out4[2][out4[1] > 0] >= np.float32(0.8)).all(), 'cfg-4 properties'
            extra['cfg4_one_gpu'] = {'value': (o4.size - 1) / best4, 'unit': 'proteins/s', 'ms': best4 * 1e3,
                                     'proteins': int(o4.size - 1), 'clusters': int(out4[4]), 'windows': int(out4[5]['sweeps']),
                                     'what': 'the 4000-genome synthetic shape (configs[3]) clustered on ONE GPU, inputs in HBM, '
                                             'best of 3: the N = 1 point of the record-sharded series bench.py --gpus N runs'}
            del d_r4, d_o4, r4, o4
            torch.cuda.empty_cache()

        kern = {k: (v[0], v[1]) for k, v in prof_all.items()}   # one profiled step: (ms, launches)
        words = (G + 63) // 64
        pc_bytes = my_iter * (S * words * 8 + 2 * S * 4) + my_iter * S * 4
        sweep_ms = prof['pan_core_sweep_kernel'][0] / prof['pan_core_sweep_kernel'][1] if my_iter else float('nan')
        pc_gbs = pc_bytes / (sweep_ms * 1e-3) / 1e9
        terms = cluster_terms(st)
        cl_bytes = sum(terms.values())
        cl_gbs = cl_bytes / (t_cluster / steps) / 1e9
        pmc, pmc_src = {}, None
        if os.path.exists(os.path.join(ROOT, PMC_SUMMARY)):
            pmc = json.load(open(os.path.join(ROOT, PMC_SUMMARY))).get('kernels', {})
            pmc_src = 'static: %s (rocprofv3 --pmc passes of this command; not measured in this run)' % PMC_SUMMARY

        def traffic(*names):
            vals = [pmc[n_]['hbm_bytes_per_launch'] * pmc[n_].get('launches', 1) for n_ in names if n_ in pmc]
            cnt = sum(pmc[n_].get('launches', 1) for n_ in names if n_ in pmc)
            return sum(vals) / cnt if cnt else None
        # kernel families of the clustering by accumulated device time (one profiled step)
        # (the block pass is the <true> instantiation of the filter too: rocprofv3 lists it with the passes over new entries)
        fam = {'filter_kernel': ('filter_kernel<all>', 'filter_kernel<new>', 'filter_kernel<block>'), 'align_kernel': ('align_kernel',),
               'diag_kernel': ('diag_kernel',)}
        fam_ms = {f: sum(kern.get(k_, (0, 0))[0] for k_ in ks) for f, ks in fam.items()}
        fam_n = {f: sum(kern.get(k_, (0, 0))[1] for k_ in ks) for f, ks in fam.items()}
        dom = max(fam_ms, key=fam_ms.get)
        # algorithmic bytes of that family: SURVEY 8d's own terms
        alg = {'filter_kernel': terms['postings'],           # 4 * P: one 4-byte posting entry per visit
               'align_kernel': terms['aligned_reps'],        # ceil(5 * A / 8): the aligned representatives' residues
               'diag_kernel': terms['aligned_reps']}[dom]    # (the diagonal test reads the same representatives)
        dom_ms, dom_n = fam_ms[dom], max(fam_n[dom], 1)
        dom_gbs = alg / (dom_ms * 1e-3) / 1e9
        roofline = {'bound': 'hbm', 'kernel': dom, 'achieved': dom_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': dom_gbs / HBM_PEAK_GBS, 'traffic': traffic(*fam[dom]), 'traffic_source': pmc_src,
                    'avg_kernel_ms': dom_ms / dom_n, 'launches_per_step': dom_n,
                    'algorithmic_bytes_per_launch': alg / dom_n,
                    'note': 'kernel durations as they are in the timed configuration: consecutive windows overlap on two '
                            'streams, so a launch shares the device with the other window\'s (PGX_NO_OVERLAP=1 gives '
                            'exclusive durations: DESIGN.md section 6)',
                    'model': {'filter_kernel': '4 B x P posting visits of the sequential rule (SURVEY 8d)',
                              'align_kernel': 'ceil(5 A / 8): residues of the aligned representatives (SURVEY 8d)',
                              'diag_kernel': 'ceil(5 A / 8) (SURVEY 8d has no term of its own for the diagonal test)'}[dom],
                    'implementation_traffic_model': {
                        'filter line reads (the first 64-byte piece of a code\'s 256-byte line per query word in the pass over '
                        'the whole index; the further pieces only for lists of more than 11, 27, 43 entries)':
                            64 * st['sum_len_queries'], 'note': 'what the layout moves, not algorithmic work'}}
        roofline_pc = {'bound': 'l2', 'kernel': 'pan_core_sweep_kernel', 'achieved': pc_gbs, 'peak': L2_PEAK_GBS,
                       'unit': 'GB/s', 'frac': pc_gbs / L2_PEAK_GBS, 'frac_of_hbm_peak': pc_gbs / HBM_PEAK_GBS,
                       'traffic': traffic('pan_core_sweep_kernel'), 'traffic_source': pmc_src,
                       'avg_kernel_ms': sweep_ms, 'algorithmic_bytes_per_launch': pc_bytes,
                       'note': 'the 7.5 MB matrix is L2-resident by design (XCD striping), so the roof is the aggregate L2 '
                               '(34.5 TB/s); the same bytes against the 8 TB/s HBM peak are given beside it'}
        extra.update(kern=kern, terms=terms, cl_bytes=cl_bytes, cl_gbs=cl_gbs, roofline=roofline, roofline_pc=roofline_pc)

        cpu = None
        if not args.skip_cpu and world == 1:   # (the CPU baseline is taken at N = 1 only)
            k_g = min(args.cpu_sample_genomes, pset.n_genomes)
            sub = synth.ProteinSet(k_g, pset.cds, pset.F, pset.C, pset.seed)
            sres, soff, _ = sub.nr_arrays()
            t1 = time.perf_counter()
            ocl = oracle.cluster_greedy(sres, soff, base_params)
            cdt = time.perf_counter() - t1
            # the sample doubles as a parity check of the library on the same sequences
            gcl = ctx.cluster_greedy(sres, soff, base_params)
            assert np.array_equal(gcl[0], ocl[0]) and np.array_equal(gcl[1], ocl[1]) and np.array_equal(gcl[2], ocl[2]), \
                'clustering parity on the CPU sample'
            k = 20
            t1 = time.perf_counter()
            oracle.pan_core(row, col, None, G, S, perms[:k])
            pdt = time.perf_counter() - t1
            tmpc = tempfile.mkdtemp(prefix='pgx_cdhit_')
            try:
                ref = cdhit_reference(sres, soff, gcl[0], gcl[1], tmpc)
            finally:
                shutil.rmtree(tmpc, ignore_errors=True)
            cpu = {'value': (soff.size - 1) / cdt, 'unit': 'proteins/s', 'cores': 1, 'kind': 'port',
                   'host_cores_available': os.cpu_count(),
                   'cd_hit': ref if ref is not None else 'probed (shutil.which("cd-hit")): absent on this host',
                   'sample': 'own restatement (oracle/cluster_ref.c, sequential) on the '
                             'non-redundant set of the first %d of %d genomes: %d sequences, %d clusters, %.1f s on one core. '
                             'The rule is sequential, so there is no all-cores figure; the per-sequence cost grows with '
                             'the table: the full 400-genome set runs at about 16 k proteins/s (DESIGN.md).'
                             % (k_g, pset.n_genomes, soff.size - 1, ocl[4], cdt),
                   'pan_core': {'value': k / pdt, 'unit': 'iters/s', 'cores': 1,
                                'sample': '%d of %d iterations, oracle/pancore_ref.c (dense incidence loop of '
                                          'pangenome_analysis.py:81-90)' % (k, n_iter)}}
            if ref is not None and 'value' in ref.get('as_reference', {}):   # the real program ran: it is the baseline
                cpu.update(kind='reference', value=ref['as_reference']['value'], port_value=(soff.size - 1) / cdt)
        extra['cpu'] = cpu
    barrier()
    if rank == 0:
        steps = args.steps
        cl, mem, iden, _, n_clusters, st = last['cluster']
        jobs = 1 if (sharded or world == 1) else world   # independent jobs running side by side
        line = {
            'metric': 'proteins/sec clustered at 0.8 identity + pan/core iters/sec, 400-genome set',
            'value': jobs * n_nr * steps / t_cluster, 'unit': 'proteins/s', 'n_gpus': world, 'steps': steps,
            'warmup': args.warmup, 'ms_per_step': dt / steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak' if jobs > 1 else 'strong',
            'vs_baseline': None, 'dtype': 'i32', 'data': 'synthetic',
            'config': {'workload': '%s: %d genomes x %d CDS synthetic (SURVEY 8d), %d raw records -> %d '
                                   'non-redundant proteins, %d clusters at -c 0.8 -n 5; pan/core %d iterations '
                                   'on synthetic %d x %d matrix' % (workload, pset.n_genomes, pset.cds, n_raw,
                                                                    n_nr, n_clusters, n_iter, G, S),
                       'parallelism': ('record-sharded x%d: one job, window member i on rank i %% %d, all-gather of the '
                                       'best keys per evaluation (%s); pan/core iterations split'
                                       % (world, world, 'ncclAllGather enqueued by libpgx' if args.exchange == 'native'
                                          else 'torch.distributed, nccl backend')) if sharded
                       else ('replicas x%d' % world if world > 1 else 'one GPU')},
            'pan_core': {'value': jobs * n_iter * steps / t_pancore, 'unit': 'iters/s',
                         'ms': t_pancore / steps * 1e3, 'roofline': extra['roofline_pc'],
                         'entry_point': extra.get('pan_core_entry_point')},
            'cluster': {'ms': t_cluster / steps * 1e3, 'raw_records_per_s': jobs * n_raw * steps / t_cluster,
                        'host_pointer_ms': extra.get('host_pointer_ms'),
                        'without_counters_ms': extra.get('without_counters_ms'),
                        'algorithmic_bytes': extra['cl_bytes'], 'algorithmic_terms': extra['terms'],
                        'achieved_GBs': extra['cl_gbs'], 'frac_hbm': extra['cl_gbs'] / HBM_PEAK_GBS,
                        'dp_cells_per_s': st['dp_cells'] / (t_cluster / steps), 'stats': st},
            'one_gpu_ms': extra.get('one_gpu_ms'),
            'speedup_vs_one_gpu': (extra['one_gpu_ms'] / (t_cluster / steps * 1e3)) if extra.get('one_gpu_ms') else None,
            'end_to_end': extra.get('end_to_end'),
            'cfg4_one_gpu': extra.get('cfg4_one_gpu'),
            'roofline': extra['roofline'],
            'cpu_baseline': extra['cpu'],
            'kernels_ms_per_step': {k_: {'ms': round(v[0], 4), 'launches': v[1]} for k_, v in sorted(extra['kern'].items())},
        }
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()
    ctx.close()


if __name__ == '__main__':
    main()
