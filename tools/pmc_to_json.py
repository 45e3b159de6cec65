"""gpurun_out/pmc/*.db (tools/profile_round.sh) -> profiles/pmc_dominant_conv.{json,txt}"""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
# the fused Residual tail in Winograd form as the network launches it: bf16 pipe with 3-way split operands (default), or the fp32 pipe
X3 = os.environ.get("SUO_WINO_BF16X3", "1") not in ("0", "")
F16 = X3 and os.environ.get("SUO_F16X2", "1") not in ("0", "")          # two fp16 planes (csrc/f16x2.h), the default
KERNEL = ("wino3x3_x3_kernel<true,false,true,4,2,false,false>" if F16 else "wino3x3_x3_kernel<true,false,true,4,3,false,false>") if X3 else "wino3x3_kernel<true,4,false>"
vals = {}
dur_ns = {}                                     # counter -> average duration of the kernel in the pass that collected it
for f in os.listdir(os.path.join(ROOT, "gpurun_out", "pmc")):
    if not f.endswith(".db"):
        continue
    c = sqlite3.connect(os.path.join(ROOT, "gpurun_out", "pmc", f))
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    namec = "kernel_name" if "kernel_name" in cols else "name"
    seen = []
    for name, cnt, n, avg in c.execute(f"select {namec}, counter_name, count(*), avg(value) from counters_collection group by {namec}, counter_name"):
        if KERNEL in name.replace(" ", ""):
            vals[cnt] = avg
            seen.append(cnt)
    try:
        d = [r[1] for r in c.execute("select name, avg(end - start) from kernels group by name") if KERNEL in r[0].replace(" ", "")]
        for cnt in seen:
            dur_ns[cnt] = d[0]
    except Exception:
        pass
fetch = vals["FETCH_SIZE"] * 1024 * 2          # KB -> B; gfx950: FETCH_SIZE reads 1/2 of wide coalesced reads (microarch guide, HBM section)
write = vals["WRITE_SIZE"] * 1024
cycles = vals["GRBM_GUI_ACTIVE"] / 8            # summed over the 8 XCDs
rec = {"kernel": KERNEL, "crops_per_launch": L, "hbm_bytes_per_launch": round(fetch + write),
       "fetch_bytes_corrected": round(fetch), "write_bytes": round(write),
       "algorithmic_bytes": L * 64 * 64 * (128 + 256 + 256) * 4 + (128 * 128 * 16 + 128 * 256) * (4 if F16 or not X3 else 6),      # in 128 ch + skip 256 ch + out 256 ch + weights
       "mfma_busy_cycles": vals["SQ_VALU_MFMA_BUSY_CYCLES"], "kernel_cycles": cycles,
       "mfma_util": vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cycles),      # busy SIMD-cycles / (256 CUs x 4 SIMDs x kernel cycles)
       # shader clock under THIS kernel's load = GRBM_GUI_ACTIVE / 8 XCDs / the launch's duration in the same (profiled) pass
       "pmc_pass_avg_launch_us": round(dur_ns["GRBM_GUI_ACTIVE"] / 1e3, 2) if "GRBM_GUI_ACTIVE" in dur_ns else None,
       "shader_clock_ghz": round(cycles / dur_ns["GRBM_GUI_ACTIVE"], 4) if "GRBM_GUI_ACTIVE" in dur_ns else None,
       "l2_hit_rate": vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]),
       "lds_bank_conflict_frac": vals["SQ_LDS_BANK_CONFLICT"] / vals["SQ_LDS_IDX_ACTIVE"], "raw": vals}
json.dump(rec, open(os.path.join(ROOT, "profiles", "pmc_dominant_conv.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in rec.items() if k != "raw"}, indent=1))
