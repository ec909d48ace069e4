#!/bin/bash
# Runs on the GPU box (gpurun): the measurements profiles/ is built from.  Output: gpurun_out/r02_profiles/.
# usage: bash tools/collect_profiles.sh [tag]
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/${1:-r02}_profiles
mkdir -p $O
echo "== bench config 3 (driver line)"; timeout -k 10 500 python bench.py --steps 5 --warmup 2 > $O/bench_config3.json 2> $O/bench_config3.err || exit 1
echo "== bench config 1"; timeout -k 10 300 python bench.py --config 1 --steps 5 --warmup 2 --cpu-clips 64 > $O/bench_config1.json 2> $O/bench_config1.err || exit 1
echo "== bench config 2"; timeout -k 10 500 python bench.py --config 2 --steps 3 --warmup 1 --cpu-clips 32 > $O/bench_config2.json 2> $O/bench_config2.err || exit 1
echo "== kernel trace of the driver command"; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o b -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_traced.json 2> $O/bench_traced.err || exit 1
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE TCC_HIT_sum" ; do
  n=$(echo $c | cut -d" " -f1)
  echo "== pmc $n (Lloyd sweeps of tools/kmeans_small.py 1)"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_filter/$n -o p -- python3 tools/kmeans_small.py 1 > $O/pmc_filter_$n.log 2>&1 || echo "   (pass $n failed)"
done
python tools/pmc_kernel.py "f16filter_kernel<64, 2, false, true, 3" $O/pmc_filter/* > $O/pmc_filter.json
echo "== done"; ls $O
