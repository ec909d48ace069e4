#!/bin/bash
# PMC counters of the 512-point log-mel kernel (tools/logmel_only.py: 2000 ten-second clips, n_mels = 64, frame-major unit rows).
# usage (GPU box): bash tools/pmc_logmel.sh <tag>     -> gpurun_out/<tag>_lm_pmc/<first counter>/...
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/${1:-r03}_lm_pmc
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
         "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" ; do
  n=$(echo $c | cut -d" " -f1)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$n -o p -- python3 tools/logmel_only.py 64 > $O/$n.log 2>&1 || echo "   (pass $n failed)"
done
ls $O
