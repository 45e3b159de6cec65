#!/bin/bash
# PMC passes for the dominant kernel (run on the GPU box through gpurun); counters in separate passes as the
# microarch guide prescribes.  Output: gpurun_out/pmc/*.db -> summarise with tools/pmc_to_json.py
set -e
L=${1:-32}
mkdir -p /root/repo/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d" " -f1)
  rocprofv3 --pmc $c --kernel-trace -d /root/repo/gpurun_out/pmc -o $n -- python3 /root/repo/tools/bench_dominant.py 20 $L > /dev/null 2>&1
done
ls /root/repo/gpurun_out/pmc
