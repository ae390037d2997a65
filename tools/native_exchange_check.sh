#!/bin/bash
# GPU box (development aid): the record-sharded tests, then cfg-3s through RCCL at world size 1 with both exchanges and
# unsharded, for comparison.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_cluster_sharded.py -x -q -m "gpu and not slow" > gpurun_out/nat_tests.log 2>&1; rc=$?
tail -5 gpurun_out/nat_tests.log
[ $rc -ne 0 ] && exit $rc
for X in torch native; do
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --exchange $X --workload cfg-3s --skip-cpu --skip-e2e --skip-cfg4 --steps 5 > gpurun_out/nat_bench_$X.json 2> gpurun_out/nat_bench_$X.err || { tail -20 gpurun_out/nat_bench_$X.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/nat_bench_$X.json')); print('$X', d['cluster']['ms'], d['value'], d['config']['parallelism'])"
done
timeout -k 10 400 python bench.py --skip-cpu --skip-e2e --skip-cfg4 --steps 5 > gpurun_out/nat_bench_plain.json 2> gpurun_out/nat_bench_plain.err && python -c "
import json; d=json.load(open('gpurun_out/nat_bench_plain.json')); print('plain', d['cluster']['ms'], d['value'])"
