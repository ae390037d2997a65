#!/bin/bash
# GPU box: the records DESIGN.md section 6 and bench.py's `roofline.traffic` cite, written under gpurun_out/<TAG>_*
# (copy what is to be kept into profiles/).  usage: tools/collect_profiles.sh TAG
# rocprofv3 gets the program itself after `--` (python3 bench.py ...), never a wrapper.
set -o pipefail
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SHORT="--steps 2 --warmup 1 --skip-cpu --skip-e2e --skip-cfg4"
echo "== bench (default run)"; (cd $R && timeout -k 10 900 python3 bench.py > $OUT/${TAG}_bench_cfg3s.json 2> $OUT/${TAG}_bench_cfg3s.err) || { tail -5 $OUT/${TAG}_bench_cfg3s.err; exit 1; }
echo "== kernel trace + stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_kt -- python3 $R/bench.py $SHORT > $OUT/${TAG}_kt.log 2>&1 || exit 1
f=$(find $OUT/${TAG}_kt -name "*kernel_stats.csv" | head -1); cp $f $OUT/${TAG}_kernel_stats_cfg3s.csv
f=$(find $OUT/${TAG}_kt -name "*kernel_trace.csv" | head -1); (cd $R && python3 tools/timeline.py $f auto 10 > $OUT/${TAG}_timeline_cfg3s.txt 2>&1)
rm -rf $OUT/${TAG}_kt
echo "== FETCH_SIZE"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pf -- python3 $R/bench.py --steps 1 --warmup 0 --skip-cpu --skip-e2e --skip-cfg4 > $OUT/${TAG}_pf.log 2>&1 || exit 1
echo "== WRITE_SIZE"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pw -- python3 $R/bench.py --steps 1 --warmup 0 --skip-cpu --skip-e2e --skip-cfg4 > $OUT/${TAG}_pw.log 2>&1 || exit 1
(cd $R && python3 tools/pmc_summary.py $OUT/${TAG}_pf $OUT/${TAG}_pw > $OUT/${TAG}_pmc_summary_cfg3s.json)
rm -rf $OUT/${TAG}_pf $OUT/${TAG}_pw
echo "== SQ counters"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/${TAG}_sq -- python3 $R/bench.py --only cluster --steps 1 --warmup 0 > $OUT/${TAG}_sq.log 2>&1 || exit 1
(cd $R && python3 tools/pmc_sq_summary.py $OUT/${TAG}_sq > $OUT/${TAG}_sq_counters_cfg3s.txt 2>&1)
rm -rf $OUT/${TAG}_sq
echo "== done"; ls -la $OUT | grep ${TAG}_
