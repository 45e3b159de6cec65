"""Build libsuo_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsuo_hip.so")
SOURCES = ["capi.hip", "net.hip", "conv.hip", "conv_wino.hip", "conv_wino_x3.hip", "conv_small.hip", "stem_x3.hip", "res_small.hip", "res_small_x3.hip", "gemm_persist.hip", "gemm_bf16x3.hip", "misc.hip", "pnp.hip", "lm.hip", "lm_big.hip", "lm_cam.hip", "lm_cam2.hip", "lm_frame.hip", "lm_frame2.hip", "lm_dist.hip", "geom_api.hip", "frame_geom.hip", "eval.hip", "slam_score.hip", "slam_vote.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off"]
# -ffp-contract=off: the fp64 geometry kernels (pnp.hip / lm.hip) must round like the gcc-built
# oracle; the fp32 CNN kernels use explicit fmaf()/MFMA where fusion is intended.


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "suo_hip.h"))
    objs = []
    procs = []
    for src in srcs:
        obj = src[:-4] + ".o"
        objs.append(obj)
        deps = [src] + hdrs
        if src.endswith("lm_big.hip"):
            deps.append(os.path.join(CSRC, "lm.hip"))        # lm_big.hip re-compiles lm.hip with 1024 threads
        if force or _stale(obj, deps):
            cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    if force or procs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
