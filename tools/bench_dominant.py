"""Run only the dominant kernel (3x3 conv 128->128 @64x64, L=8) N times: target for rocprofv3 --pmc passes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
L = int(sys.argv[2]) if len(sys.argv) > 2 else 32
r = bench.conv_roofline(L, iters=n)
print(r)
