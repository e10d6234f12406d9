#!/bin/bash
# Round-3 evidence in one GPU call: kernel traces (--stats), PMC passes (their own runs, never combined with tracing),
# and the un-profiled bench lines.  Output under gpurun_out/r03/; the summaries are copied to profiles/ by hand.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "== kernel trace: inference headline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/infer -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/infer.log 2>&1 || echo "infer trace failed"
echo "== kernel trace: training step, every launch on one stream"
WG_TRAIN_SERIAL=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_serial -- python3 $ROOT/tools/bench_train.py --adam --steps 2 --warmup 1 > $OUT/train_serial.log 2>&1 || echo "train serial trace failed"
echo "== kernel trace: training step, shipped stream configuration"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_streams -- python3 $ROOT/tools/bench_train.py --adam --steps 2 --warmup 1 > $OUT/train_streams.log 2>&1 || echo "train streams trace failed"
echo "== PMC: inference"
bash $ROOT/tools/profile_pmc.sh r03 > $OUT/pmc_infer.log 2>&1
echo "== PMC: training (serial)"
WG_TRAIN_SERIAL=1 bash $ROOT/tools/profile_pmc_train.sh r03t > $OUT/pmc_train.log 2>&1
echo "== un-profiled bench lines"
cd $ROOT
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout -k 10 300 python3 bench.py --workload train --steps 10 --warmup 3 > $OUT/bench_train.json 2> $OUT/bench_train.err
timeout -k 10 200 python3 tools/bench_latency.py > $OUT/latency.json 2> $OUT/latency.err
find $OUT -name "*kernel_stats.csv" | head
tail -c 600 $OUT/bench_train.json
