#!/bin/bash
# One build-measure iteration on the GPU box (development aid): the quick clustering parity tests, then a short
# bench of the clustering step with the per-kernel table.  usage: tools/gpu_iter.sh LABEL [pytest -k expression]
set -o pipefail
L=${1:-iter}
K=${2:-}
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_cluster.py tests/test_gpu_cluster_chunked.py -x -q -m "gpu and not slow" ${K:+-k "$K"} > gpurun_out/${L}_tests.log 2>&1
rc=$?
tail -4 gpurun_out/${L}_tests.log
[ $rc -ne 0 ] && exit $rc
PGX_TRACE=1 timeout -k 10 500 python bench.py --skip-cpu --skip-e2e --skip-cfg4 --steps 5 > gpurun_out/${L}_bench.json 2> gpurun_out/${L}_bench.err || { tail -20 gpurun_out/${L}_bench.err; exit 1; }
python - <<PY
import json
d = json.load(open('gpurun_out/${L}_bench.json'))
print('value %.3f M proteins/s, cluster %.2f ms, step %.2f ms' % (d['value'] / 1e6, d['cluster']['ms'], d['ms_per_step']))
for k, v in sorted(d['kernels_ms_per_step'].items(), key=lambda kv: -kv[1]['ms']):
    print('  %-26s %8.3f ms %5d' % (k, v['ms'], v['launches']))
print('gpu stats', d['cluster']['stats']['gpu'])
PY
grep "windows" gpurun_out/${L}_bench.err | tail -2
if [ -n "$CFG4" ]; then
  timeout -k 10 600 python bench.py --workload cfg-4 --skip-cpu --skip-e2e --steps 2 --warmup 1 > gpurun_out/${L}_bench4.json 2> gpurun_out/${L}_bench4.err || { tail -5 gpurun_out/${L}_bench4.err; exit 1; }
  python - <<PY
import json
d = json.load(open('gpurun_out/${L}_bench4.json'))
print('cfg-4: value %.3f M proteins/s, cluster %.1f ms' % (d['value'] / 1e6, d['cluster']['ms']))
for k, v in sorted(d['kernels_ms_per_step'].items(), key=lambda kv: -kv[1]['ms'])[:8]:
    print('  %-26s %8.2f ms %5d' % (k, v['ms'], v['launches']))
PY
fi
