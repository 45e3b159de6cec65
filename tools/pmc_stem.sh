#!/bin/bash
# PMC passes over the fused stem (tools/bench_stem.py 256): where do its cycles go?   -> gpurun_out/pmc_stem.txt
# (no HBM-byte pass here: FETCH_SIZE / WRITE_SIZE together with TCC_HIT / TCC_MISS in ONE pass did not finish in 14 minutes on this pool --
#  collect them as tools/profile_round.sh does, one counter per pass, each under its own `timeout`)
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/pmc_stem
rm -rf $D && mkdir -p $D
cd /tmp && export TMPDIR=/tmp
: > $R/gpurun_out/pmc_stem.txt
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
         "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_IFETCH"; do
  n=$(echo $c | cut -d" " -f1)
  timeout 120 rocprofv3 --pmc $c --kernel-trace -d $D -o $n -- python3 $R/tools/bench_stem.py 256 > $D/$n.log 2>&1
  python3 $R/tools/rocpd_pmc.py $(find $D -name "${n}*.db" | head -1) stem_x3 >> $R/gpurun_out/pmc_stem.txt 2>&1
done
cat $R/gpurun_out/pmc_stem.txt
