"""HBM traffic and MFMA utilisation of the persistent 1x1 GEMM on one of the network's shapes (rocprofv3 PMC passes).

Two roles:
  python3 tools/pmc_gemm.py run <shape> [iters]     -- launch the shape `iters` times (the target of each rocprofv3 pass)
  python3 tools/pmc_gemm.py report <shape>          -- gpurun_out/pmc_gemm_<shape>/*.db -> text summary on stdout
Shapes: residual  (K 128 -> N 256 + identity skip, M = 524288: Residual.conv3 at 64x64, 128 crops)
        conv1     (K 256 -> N 128, BN-ReLU prologue + ReLU, M = 524288: Residual.conv1)
        lin       (K 256 -> N 256 + ReLU, M = 524288: the lin_ head)
        conv1_x3  (conv1 through gemm_bf16x3_kernel with three bf16 planes: csrc/gemm_bf16x3.hip, SUO_F16X2=0)
        conv1_f16 (the same kernel on two fp16 planes, csrc/f16x2.h: the form the network launches by default)
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
    "conv1_x3": (256, 128, dict(pro=True, relu=True, x3=True), 4 * (M * 256 + M * 128) + 6 * 256 * 128),
    "conv1_f16": (256, 128, dict(pro=True, relu=True, x3=True, f16=True), 4 * (M * 256 + M * 128) + 4 * 256 * 128),
}


def run(shape, iters):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_ops as bo
    import torch
    K, N, kw, _ = SHAPES[shape]
    bo.timeit.__defaults__ = (iters,)
    if kw.get("x3"):
        import ctypes as C
        import numpy as np
        from suo_slam_amd import _lib
        lib = _lib.lib()
        rng = np.random.default_rng(0)
        w = (rng.standard_normal((N, K)) / 16).astype(np.float32)
        w3 = np.empty(3 * N * K, np.uint16)
        _lib.check(lib.suo_pack_gemm_weight_bf16x3(w.ctypes.data, N, K, w3.ctypes.data))
        w3d = torch.from_numpy(w3.view(np.int16)).cuda()
        w16, osc = np.empty(2 * N * K, np.uint16), np.empty(N, np.float32)
        _lib.check(lib.suo_pack_gemm_weight_f16x2(w.ctypes.data, N, K, w16.ctypes.data, osc.ctypes.data))
        w16d, oscd, flag = torch.from_numpy(w16.view(np.int16)).cuda(), torch.from_numpy(osc).cuda(), torch.zeros(1, dtype=torch.int32, device="cuda")
        rot = 8                                                # distinct buffers: inputs are not L2-resident between launches
        a = [torch.randn((M, K), device="cuda") for _ in range(rot)]
        o = [torch.empty((M, N), device="cuda") for _ in range(rot)]
        sc, sh, b = torch.ones(K, device="cuda"), torch.zeros(K, device="cuda"), torch.zeros(N, device="cuda")
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        i = [0]

        def fn():
            k = i[0] % rot
            i[0] += 1
            if kw.get("f16"):
                _lib.check(lib.suo_conv1x1_f16x2_ex(P(a[k]), K, K, P(sc), P(sh), None, 0, 0, P(w16d), P(oscd), P(b), None, 0, P(o[k]), N, M, N, 1, P(flag), s))
            else:
                _lib.check(lib.suo_conv1x1_bf16x3(P(a[k]), K, K, P(sc), P(sh), P(w3d), P(b), P(o[k]), N, M, N, 1, s))
        us = bo.timeit(fn)
        print(f"gemm_{'f16x2' if kw.get('f16') else 'bf16x3'} M={M} K={K} N={N}: {us:8.2f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF (fp32-equivalent)")
    else:
        bo.gemm(M, K, N, **{k: v for k, v in kw.items() if k != "x3"})
    torch.cuda.synchronize()


def report(shape):
    K, N, kw, alg = SHAPES[shape]
    kname = "gemm_bf16x3_kernel" if kw.get("x3") else "gemm_persist_kernel"
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
                if kname in name:
                    vals[cnt] = avg
        if f.startswith("trace") and "kernels" in tabs:
            rows = c.execute(f"select avg(end-start), min(end-start), count(*) from kernels where name like '%{kname}%'").fetchall()
            dur = rows[0]
    fetch = vals["FETCH_SIZE"] * 1024 * 2          # KB -> B; gfx950 reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM section)
    write = vals["WRITE_SIZE"] * 1024
    cycles = vals["GRBM_GUI_ACTIVE"] / 8           # summed over the 8 XCDs
    flop = 2.0 * M * K * N
    rec = {"kernel": ("gemm_bf16x3_kernel<true,...,NP=2> (f16x2)" if kw.get("f16") else "gemm_bf16x3_kernel<true,...,NP=3>") if kw.get("x3") else "gemm_persist_kernel<2,2,2,2,%s>" % ("true" if kw.get("res") else "false"),
           "shape": f"M={M} K={K} N={N} {kw}",
           "avg_launch_us": round(dur[0] / 1e3, 2), "min_launch_us": round(dur[1] / 1e3, 2), "launches_timed": dur[2],
           "tflops": round(flop / dur[0] / 1e3, 1), "mfma_flop_per_busy_cycle": 1024 if kw.get("x3") else 64,
           "hbm_bytes_per_launch": round(fetch + write), "fetch_bytes_corrected": round(fetch), "write_bytes": round(write),
           "algorithmic_bytes": alg, "traffic_over_algorithmic": round((fetch + write) / alg, 3),
           "hbm_GBps": round((fetch + write) / dur[0], 1), "hbm_frac_of_8TBps": round((fetch + write) / dur[0] / 8000, 3),
           "mfma_util": round(vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cycles), 3),
           "l2_hit_rate": round(vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]), 3)}
    print(json.dumps(rec, indent=1))
    want = "conv1" if os.environ.get("SUO_WINO_BF16X3", "1") in ("0", "") else ("conv1_x3" if os.environ.get("SUO_F16X2", "1") in ("0", "") else "conv1_f16")
    if shape == want and M == 1048576:                    # bench.py's roofline_all.largest_gemm reads `traffic` from here
        out = {"kernel": kname, "crops_per_launch": M // 4096, "hbm_bytes_per_launch": rec["hbm_bytes_per_launch"], "detail": rec}
        json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_gemm.json"), "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 20)
    else:
        report(sys.argv[2])
