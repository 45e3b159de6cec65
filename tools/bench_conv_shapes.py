"""Time the network's 3x3 convolution shapes through the C ABI (SUO_CONV3_CFG=0/1/2/3 forces a tile configuration, one
process per setting): python tools/bench_conv_shapes.py [L]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_ops as bo  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 128
print({k: v for k, v in os.environ.items() if k.startswith("SUO_")}, "L =", L)
bo.conv3(L, 128, 64, 64)        # r1 at 128x128
for H in (64, 32, 16, 8):
    bo.conv3(L, H, 128, 128)    # hourglass levels
