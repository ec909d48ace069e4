#!/bin/bash
# Runs on the GPU box (gpurun): the measurements profiles/ is built from.  Output: gpurun_out/<tag>_profiles/.
# usage: bash tools/collect_profiles.sh [tag]
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/${1:-r03}_profiles
mkdir -p $O
echo "== bench config 3 (driver line)"; timeout -k 10 500 python bench.py --steps 5 --warmup 2 > $O/bench_config3.json 2> $O/bench_config3.err || exit 1
echo "== bench config 1"; timeout -k 10 300 python bench.py --config 1 --steps 5 --warmup 2 --cpu-clips 64 > $O/bench_config1.json 2> $O/bench_config1.err || exit 1
echo "== bench config 2"; timeout -k 10 500 python bench.py --config 2 --steps 3 --warmup 1 --cpu-clips 32 > $O/bench_config2.json 2> $O/bench_config2.err || exit 1
echo "== kernel trace of the driver command"; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o b -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hard-workload > $O/bench_traced.json 2> $O/bench_traced.err || exit 1
rm -f $O/trace/*.db $O/trace/*/*.db
echo "== done"; ls $O
