#!/usr/bin/env python3
"""Golden draw sequences of the REFERENCE's RANSAC sampler: thirdparty/lambdatwist/utils/random.h compiled from where it lies by `make -C oracle ref`
(oracle/ref_random_shim.cpp -> oracle/_ref/librandom_ref.so; get4RandomInRange0 of pnp_ransac.cpp:161-183 restated over its mlib::randui there).
One process-global stream from a fresh seed: first raw randui draws for several ranges, then -- continuing the SAME stream, as consecutive pnp calls do --
blocks of 4-point samples for the point counts the keypoint configurations produce.  Data only.
Run in the build container:  python tests/golden/make_sampler_golden.py"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all", "ref"])
from oracle import geometry as G  # noqa: E402

R = G.ref_random()
R.ref_rng_reset()
out = {}
ranges = np.array([4, 5, 7, 8, 10, 22, 41, 64, 1000, 65536, 2 ** 30], np.int64)
raw = np.zeros((len(ranges), 400), np.int32)
for i, m in enumerate(ranges):
    R.ref_randui(int(m), raw.shape[1], raw[i])
out["randui_ranges"], out["randui_draws"] = ranges, raw
counts = np.array([8, 4, 41, 10, 5, 22, 6, 12, 4, 18], np.int32)             # consecutive get4 blocks on the continuing stream
blocks = np.zeros((len(counts), 1000, 4), np.int32)
for i, n in enumerate(counts):
    R.ref_get4(int(n), blocks.shape[1], blocks[i].reshape(-1))
out["get4_counts"], out["get4_blocks"] = counts, blocks
np.savez_compressed(os.path.join(HERE, "sampler_golden.npz"), **out)
print("wrote sampler_golden.npz:", {k: v.shape for k, v in out.items()})
