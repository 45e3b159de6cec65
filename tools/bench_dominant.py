"""Run only the dominant kernel (the fused Residual tail in Winograd form, 3x3 128->128 + 1x1 128->256 + skip @64x64) and, beside it,
the Winograd 3x3 alone and the two direct-form kernels of the same tile shape, N times each: target for rocprofv3 --kernel-trace / --pmc."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
L = int(sys.argv[2]) if len(sys.argv) > 2 else 32
r = bench.conv_roofline(L, iters=n)
print(r)
