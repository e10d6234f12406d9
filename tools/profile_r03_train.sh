set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
WG_TRAIN_SERIAL=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_serial -- python3 $ROOT/tools/bench_train.py --adam --steps 2 --warmup 1 > $OUT/train_serial.log 2>&1 || echo "train serial trace failed"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_streams -- python3 $ROOT/tools/bench_train.py --adam --steps 2 --warmup 1 > $OUT/train_streams.log 2>&1 || echo "train streams trace failed"
WG_TRAIN_SERIAL=1 bash $ROOT/tools/profile_pmc_train.sh r03u > $OUT/pmc_train.log 2>&1
cd $ROOT
timeout -k 10 300 python3 bench.py --workload train --steps 10 --warmup 3 > $OUT/bench_train.json 2> $OUT/bench_train.err
tail -c 700 $OUT/bench_train.json
