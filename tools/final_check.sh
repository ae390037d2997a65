#!/bin/bash
# GPU box (development aid): smoke(), `bench.py --gpus 2` on a one-GPU box (must fail loudly), and cfg-4 through RCCL at
# world size 1 with both exchanges (torch.distributed's all-gather, the library's own ncclAllGather).
set -o pipefail
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final_smoke.log 2>&1 || { tail -20 gpurun_out/final_smoke.log; exit 1; }
tail -2 gpurun_out/final_smoke.log
python bench.py --gpus 2 > gpurun_out/final_gpus2.log 2>&1; echo "bench --gpus 2 on one GPU: rc=$?"; tail -3 gpurun_out/final_gpus2.log
for X in torch native; do
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --exchange $X --workload cfg-4 --skip-cpu --skip-e2e --steps 2 --warmup 1 > gpurun_out/final_cfg4_$X.json 2> gpurun_out/final_cfg4_$X.err || { tail -20 gpurun_out/final_cfg4_$X.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/final_cfg4_$X.json')); print('$X', d['cluster']['ms'], d['value'], d.get('one_gpu_ms'), d.get('speedup_vs_one_gpu'))"
done
