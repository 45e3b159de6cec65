#!/usr/bin/env python3
"""Golden vectors for P3P/P4P produced by the REFERENCE's own code: thirdparty/lambdatwist/p4p.cpp
compiled from where it lies by `make -C oracle ref` (oracle/_ref/libp4p_ref.so).  Also stores the
known-answer vector the reference ships in thirdparty/lambdatwist/test_pnp.py:5-14 (data only).
Run in the build container:  python tests/golden/make_pnp_golden.py"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all", "ref"])
from oracle import geometry as G  # noqa: E402

R = G.ref()
rng = np.random.Generator(np.random.PCG64(4242))
N = 400
xs = np.zeros((N, 4, 3)); ys = np.zeros((N, 4, 2)); T4 = np.zeros((N, 4, 4)); nvalid = np.zeros(N, np.int32)
Rs = np.zeros((N, 4, 9)); Ts = np.zeros((N, 4, 3))
for i in range(N):
    x = rng.uniform(-100, 100, (4, 3))
    A = rng.standard_normal((3, 3)); Q, _ = np.linalg.qr(A)
    if np.linalg.det(Q) < 0:
        Q[:, 0] *= -1
    t = np.array([rng.uniform(-80, 80), rng.uniform(-80, 80), rng.uniform(300, 1500)])
    X = x @ Q.T + t
    y = X[:, :2] / X[:, 2:3]
    if i % 5 == 0:
        y += rng.normal(0, 2e-3, y.shape)          # noisy
    if i % 37 == 0:
        x[2] = 0.5 * (x[0] + x[1])                   # collinear / degenerate triple
    xs[i], ys[i] = x, y
    T4[i] = G.ref_p4p(x, y, [0, 1, 2, 3])
    yh = np.ascontiguousarray(np.c_[y, np.ones(4)])
    r = np.zeros(36); tt = np.zeros(12)
    nvalid[i] = R.ref_p3p(yh[0].copy(), yh[1].copy(), yh[2].copy(), x[0].copy(), x[1].copy(), x[2].copy(), r, tt)
    Rs[i] = r.reshape(4, 9); Ts[i] = tt.reshape(4, 3)
    Rs[i, nvalid[i]:] = 0; Ts[i, nvalid[i]:] = 0

kat_xs = np.array([[-17.8431, 0.570044, 11.1874], [-80.6362, -23.8517, 21.0087], [-68.0126, 9.19776, 20.6913], [-8.31825, -13.5394, 23.8776], [-32.3177, 30.9775, 35.0005], [-60.5264, 3.64722, 62.0491], [-13.8288, -0.638686, 30.1851], [-25.1182, 35.7954, 81.3263], [0.841874, -20.8397, 42.3626], [-2.04336, 0.61477, 0.620302]])
kat_ys = np.array([[-0.083742, 0.314872], [-0.516025, 0.0535602], [-0.392733, 0.51515], [0.400942, -0.423236], [0.371449, 0.98387], [0.123111, 0.257844], [0.481032, 0.102744], [0.850471, 0.608635], [0.846186, -0.652791], [0.154041, 0.784826]])
kat_pose = np.array([[0.621007, 0.253154, 0.741798, 0.947568], [-0.336352, 0.940907, -0.039522, 0.258716], [-0.707968, -0.224961, 0.669458, 0.187565], [0, 0, 0, 1]])
np.savez_compressed(os.path.join(HERE, "pnp_golden.npz"), xs=xs, ys=ys, p4p_T=T4, p3p_valid=nvalid, p3p_R=Rs, p3p_t=Ts,
                    kat_xs=kat_xs, kat_ys=kat_ys, kat_pose=kat_pose)
print("wrote pnp_golden.npz", N, "problems; valid counts", np.bincount(nvalid))
