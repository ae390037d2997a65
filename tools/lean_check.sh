#!/bin/bash
# GPU box (development aid): the call without work counters against the call with them -- its parity tests, then both
# timed on cfg-3s in one process, then the per-kernel table of the call without counters.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_cluster.py -x -q -m gpu -k "without_counters" > gpurun_out/lean_tests.log 2>&1; rc=$?
tail -5 gpurun_out/lean_tests.log
[ $rc -ne 0 ] && exit $rc
python - <<'PY'
import time, numpy as np, torch
from pangenomix_amd import _native, cluster, synth
ctx = _native.Context(0)
res, off, _ = synth.protein_set('cfg-3s').nr_arrays()
p = cluster.params_from_cdhit_args({'-n': 5, '-c': 0.8})
d_res = torch.from_numpy(res.copy()).cuda(); d_off = torch.from_numpy(off.view(np.int64)).cuda()
st = torch.cuda.current_stream().cuda_stream
for want in (True, False, True, False):
    ts = []
    for _ in range(4):
        t = time.perf_counter(); r = ctx.cluster_greedy_dev(d_res.data_ptr(), d_off.data_ptr(), off.size - 1, res.size, p, st, want_stats=want); ts.append(time.perf_counter() - t)
    print('want_stats', want, ['%.1f' % (x * 1e3) for x in ts], r[4])
ctx.profile(True)
ctx.cluster_greedy_dev(d_res.data_ptr(), d_off.data_ptr(), off.size - 1, res.size, p, st, want_stats=False)
for name, (ms, n) in sorted(ctx.profile_read().items(), key=lambda kv: -kv[1][0])[:8]: print('  %-26s %8.3f ms %5d' % (name, ms, n))
PY
