#!/bin/bash
# PMC passes + a plain kernel trace for the persistent GEMM on the network's shapes (on the GPU box, inside one gpurun call):
#   bash tools/profile_gemm_pmc.sh [shape ...]      -> gpurun_out/pmc_gemm_<shape>/{*.db, report.txt}
# One rocprofv3 process per counter group and nothing but --kernel-trace next to --pmc (MI355X_MICROARCH.md, HBM section).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for shape in ${@:-residual conv1 lin}; do
  OUT=$R/gpurun_out/pmc_gemm_$shape
  rm -rf $OUT && mkdir -p $OUT
  rocprofv3 --kernel-trace -d $OUT -o trace -- python3 $R/tools/pmc_gemm.py run $shape 30 > /dev/null 2>&1
  for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
    n=$(echo $c | cut -d" " -f1)
    rocprofv3 --pmc $c --kernel-trace -d $OUT -o $n -- python3 $R/tools/pmc_gemm.py run $shape 10 > /dev/null 2>&1
  done
  # rocprofv3 puts the databases in a per-host subdirectory: flatten
  find $OUT -name "*.db" -exec mv {} $OUT/ \; 2>/dev/null
  python3 $R/tools/pmc_gemm.py report $shape > $OUT/report.txt 2>&1
  cat $OUT/report.txt
done
