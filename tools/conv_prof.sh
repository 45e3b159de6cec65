#!/bin/bash
# Phase timeline of the dominant 3x3 convolution (tools/micro/conv_prof.hip), plus timing-only experiments:
#   SUO_CONV_EXP=1 weights never re-fetched, =2 activations staged once, =3 both (results are wrong, times are not).
#   [EXPS="0 1 2 3"] [EXTRA="-DSUO_CONV_BRING3=6 ..."] bash tools/conv_prof.sh [crops]
set -u
cd "$(dirname "$0")/.."
for exp in ${EXPS:-0 1 2 3}; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DSUO_CONV_PROFILE -DSUO_CONV_EXP=$exp ${EXTRA:-} -I suo_slam_amd/csrc -I include \
        tools/micro/conv_prof.hip -o /tmp/conv_prof_$exp 2>/dev/null || { echo "build failed (exp $exp)"; exit 1; }
    echo "== SUO_CONV_EXP=$exp ${EXTRA:-}"
    timeout 60 /tmp/conv_prof_$exp "${1:-128}" | head -${LINES_SHOWN:-10}
done
