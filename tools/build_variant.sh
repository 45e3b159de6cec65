#!/bin/bash
# Build a tuning variant of libsuo_hip.so: tools/build_variant.sh <name> [-DFLAG=VALUE ...]   (select it with SUO_HIP_LIB=<path>).  Variants are built with -DSUO_TUNING:
# every SUO_TUNE knob of csrc/tune.h then follows the environment (the product library compiles them to their defaults).  "tools/build_variant.sh tuning" = just that.
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/suo_slam_amd/csrc
V=$ROOT/suo_slam_amd/variants
mkdir -p $V
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DSUO_TUNING"     # the tuning knobs of csrc/tune.h read the environment in these builds only
VAR="conv conv_wino conv_wino_x3 gemm_persist gemm_bf16x3 conv_small res_small res_small_x3 stem_x3 lm_grid lm_frame lm_frame2 lm_dist geom_api net misc"
for f in $VAR; do /opt/rocm/bin/hipcc $FL "$@" -c $C/$f.hip -o $V/${f}_$NAME.o & done
wait
OBJS=""
for f in capi pnp lm lm_big lm_cam lm_cam2 frame_geom eval slam_score slam_vote; do OBJS="$OBJS $C/$f.o"; done
for f in $VAR; do OBJS="$OBJS $V/${f}_$NAME.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libsuo_hip_$NAME.so $OBJS
rm -f $V/*_$NAME.o
echo $V/libsuo_hip_$NAME.so
