#!/bin/bash
# PMC passes over tools/bench_wino_x3.py (the bf16x3 Winograd kernels next to the fp32-pipe ones; the TCP_* / TA_* groups hang rocprofv3 on this pool: left out): one counter group per run,
# nothing but --kernel-trace beside --pmc.   tools/pmc_wino_x3.sh [crops]   -> gpurun_out/pmc_wx3/summary.txt
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
L=${1:-256}
O=$R/gpurun_out/pmc_wx3
mkdir -p $O && rm -rf $O/*
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL" \
         "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL" \
         "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace -d $O -o g$i -- python3 $R/tools/bench_wino_x3.py $L > $O/g$i.log 2>&1
done
python3 - <<PY > $O/summary.txt
import os, sqlite3, collections
O = "$O"
vals = collections.defaultdict(dict)
for root, _, files in os.walk(O):
    for f in files:
        if not f.endswith(".db"): continue
        c = sqlite3.connect(os.path.join(root, f))
        try:
            cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
        except Exception as e:
            continue
        if not cols: continue
        namec = "kernel_name" if "kernel_name" in cols else "name"
        for name, cnt, n, avg in c.execute(f"select {namec}, counter_name, count(*), avg(value) from counters_collection group by {namec}, counter_name"):
            if "wino3x3" in name: vals[name.replace(" ", "")[:60]][cnt] = avg
for k in sorted(vals):
    print("==", k)
    for cnt in sorted(vals[k]): print("   %-40s %.4g" % (cnt, vals[k][cnt]))
PY
cat $O/summary.txt
