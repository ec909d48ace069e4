#!/bin/bash
# PMC counters of the exact filter sweep on bench-like rows (tools/walk_ablate.py), old and new kernel.
# usage (GPU box): bash tools/pmc_sweep.sh <tag> <name=value of a debug switch>
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/${1:-pmc}_sweep
W=${2:-filter_nb=0}
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
         "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM" \
         "SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
         "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum" ; do
  n=$(echo $c | cut -d" " -f1)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$n -o p -- python3 tools/walk_ablate.py $W > $O/$n.log 2>&1 || echo "   (pass $n failed)"
done
python tools/pmc_kernel.py "f16filter" $O/* > $O/pmc.json
cat $O/pmc.json
