set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
for s in 0 1 0 1; do
echo speculate=$s
AT_KMEANS_SPECULATE=$s timeout -k 10 200 python tools/kmeans_small.py 2>&1 | grep rows/rank
done
AT_KMEANS_SPECULATE=0 timeout -k 10 300 python bench.py --no-cpu-baseline | cut -c1-900
AT_KMEANS_SPECULATE=1 timeout -k 10 300 python bench.py --no-cpu-baseline | cut -c1-900
