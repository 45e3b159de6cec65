"""In-kernel phase times of csrc/res_small.hip (variant built with -DSUO_RS_PROF):
   SUO_HIP_LIB=suo_slam_amd/variants/libsuo_hip_rsprof.so python tools/prof_res_block.py [crops]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from tests import hipops as ops
from tests.test_gpu_res_block import _block_weights
L = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = _block_weights(np.random.default_rng(1))
for H in (32, 16, 8, 4):
    x = torch.rand((L, H, H, 256), device="cuda") - 0.5
    for rep in range(3):
        ops.res_block(x, B["pro"], B["w1"], B["b1"], B["w2"], B["b2"], B["w3"], B["b3"])
    for rep in range(3):
        ops.res_block_x3(x, B["pro"], B["w1"], B["b1"], B["w2"], B["b2"], B["w3"], B["b3"])
