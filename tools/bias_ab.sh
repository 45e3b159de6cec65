#!/bin/bash
# The two operand splits side by side on one box: round-to-nearest (the product) and truncation (rounds 1-3; tools/build_variant.sh trunc -DSUO_S3_TRUNC),
# per-element error tables of tests/test_gpu_x3_accuracy.py.   gpurun -- bash tools/bias_ab.sh   -> gpurun_out/bias_ab.txt
cd $GRAFT_REPO_ROOT
{
echo "==== round-to-nearest split (libsuo_hip.so) ===="
python3 -m pytest tests/test_gpu_x3_accuracy.py -q -s -m gpu 2>&1 | grep -A6 "case, max\|fused tail, one"
echo "==== truncating split (variants/libsuo_hip_trunc.so, -DSUO_S3_TRUNC) ===="
SUO_HIP_LIB=$GRAFT_REPO_ROOT/suo_slam_amd/variants/libsuo_hip_trunc.so python3 -m pytest tests/test_gpu_x3_accuracy.py -q -s -m gpu 2>&1 | grep -A6 "case, max\|fused tail, one"
} > gpurun_out/bias_ab.txt
