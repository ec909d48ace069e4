set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
timeout -k 10 200 python tools/kmeans_small.py 2>&1 | grep rows/rank
timeout -k 10 200 python tools/train_fixed_cost.py 2>&1 | grep -v WARN | head -4
timeout -k 10 100 python tools/logmel_only.py
timeout -k 10 400 python bench.py > gpurun_out/bench_n1.json
cut -c1-600 gpurun_out/bench_n1.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o b -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_bench.json 2> gpurun_out/prof.log
python tools/profile_summary.py gpurun_out/prof gpurun_out/prof_bench.json > /dev/null
cp profiles/r01_bench_kernel_stats.csv profiles/r01_bench_kernel_stats_top.txt gpurun_out/
rm -rf gpurun_out/prof
