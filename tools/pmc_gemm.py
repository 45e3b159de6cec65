"""HBM traffic and MFMA utilisation of the persistent 1x1 GEMM on one of the network's shapes (rocprofv3 PMC passes).

Two roles:
  python3 tools/pmc_gemm.py run <shape> [iters]     -- launch the shape `iters` times (the target of each rocprofv3 pass)
  python3 tools/pmc_gemm.py report <shape>          -- gpurun_out/pmc_gemm_<shape>/*.db -> text summary on stdout
Shapes: residual  (K 128 -> N 256 + identity skip, M = 524288: Residual.conv3 at 64x64, 128 crops)
        conv1     (K 256 -> N 128, BN-ReLU prologue + ReLU, M = 524288: Residual.conv1)
        lin       (K 256 -> N 256 + ReLU, M = 524288: the lin_ head)
Driver: tools/profile_gemm_pmc.sh (one rocprofv3 process per counter group, no other tracing, as the micro-architecture
guide prescribes)."""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M = int(os.environ.get("SUO_PMC_GEMM_M", 524288))          # 1048576 = the bench's launch shape (256 crops @64x64)
SHAPES = {   # K1, N, kwargs, algorithmic bytes (activations + output (+ residual) + weights), flop
    "residual": (128, 256, dict(res=True), 4 * (M * 128 + 2 * M * 256 + 128 * 256)),
    "conv1": (256, 128, dict(pro=True, relu=True), 4 * (M * 256 + M * 128 + 256 * 128)),
    "lin": (256, 256, dict(relu=True), 4 * (M * 256 + M * 256 + 256 * 256)),
}


def run(shape, iters):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_ops as bo
    import torch
    K, N, kw, _ = SHAPES[shape]
    bo.timeit.__defaults__ = (iters,)
    bo.gemm(M, K, N, **kw)
    torch.cuda.synchronize()


def report(shape):
    K, N, kw, alg = SHAPES[shape]
    d = os.path.join(ROOT, "gpurun_out", f"pmc_gemm_{shape}")
    vals, dur = {}, None
    for f in sorted(os.listdir(d)):
        if not f.endswith(".db"):
            continue
        c = sqlite3.connect(os.path.join(d, f))
        tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
        if "counters_collection" in tabs:
            cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
            namec = "kernel_name" if "kernel_name" in cols else "name"
            for name, cnt, n, avg in c.execute(f"select {namec}, counter_name, count(*), avg(value) from counters_collection group by {namec}, counter_name"):
                if "gemm_persist_kernel" in name:
                    vals[cnt] = avg
        if f.startswith("trace") and "kernels" in tabs:
            rows = c.execute("select avg(end-start), min(end-start), count(*) from kernels where name like '%gemm_persist_kernel%'").fetchall()
            dur = rows[0]
    fetch = vals["FETCH_SIZE"] * 1024 * 2          # KB -> B; gfx950 reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM section)
    write = vals["WRITE_SIZE"] * 1024
    cycles = vals["GRBM_GUI_ACTIVE"] / 8           # summed over the 8 XCDs
    flop = 2.0 * M * K * N
    rec = {"kernel": "gemm_persist_kernel<2,2,2,2,%s>" % ("true" if kw.get("res") else "false"), "shape": f"M={M} K={K} N={N} {kw}",
           "avg_launch_us": round(dur[0] / 1e3, 2), "min_launch_us": round(dur[1] / 1e3, 2), "launches_timed": dur[2],
           "tflops": round(flop / dur[0] / 1e3, 1),
           "hbm_bytes_per_launch": round(fetch + write), "fetch_bytes_corrected": round(fetch), "write_bytes": round(write),
           "algorithmic_bytes": alg, "traffic_over_algorithmic": round((fetch + write) / alg, 3),
           "hbm_GBps": round((fetch + write) / dur[0], 1), "hbm_frac_of_8TBps": round((fetch + write) / dur[0] / 8000, 3),
           "mfma_util": round(vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cycles), 3),
           "l2_hit_rate": round(vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]), 3)}
    print(json.dumps(rec, indent=1))
    if shape == "conv1" and M == 1048576:                 # bench.py's roofline_all.largest_gemm reads `traffic` from here
        out = {"kernel": "gemm_persist_kernel", "crops_per_launch": M // 4096, "hbm_bytes_per_launch": rec["hbm_bytes_per_launch"], "detail": rec}
        json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_gemm.json"), "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 20)
    else:
        report(sys.argv[2])
