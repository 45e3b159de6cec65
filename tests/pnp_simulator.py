"""The point-cloud simulator of the reference's PnP benchmark (/root/reference/thirdparty/lambdatwist/simulator.h:25-94 and
utils/random_vectors.h:26-43), restated in numpy so the statistical contract of thirdparty/lambdatwist/test_pnp.cpp:68-147
(4 noise levels x 1000 problems x 250 points, 50 % outliers, < 5 % failures at the DEFAULT threshold 0.001) can be run on
the oracle and on the HIP kernel.  Test data only."""
import numpy as np


def unit_vector(rng, n):
    """getRandomUnitVector: normalised standard normal."""
    while True:
        v = rng.standard_normal(n)
        if np.abs(v).sum() >= 1e-10:
            return v / np.linalg.norm(v)


def rotation_from_quaternion(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def point_cloud_with_noisy_measurements(rng, n=250, pixel_sigma=0.0, outlier_ratio=0.5):
    """PointCloudWithNoisyMeasurements (simulator.h:46-94): returns xs [n,3], yns [n,2], Pcw [4,4].
    Pose = random rotation + unit translation; points = random rays in [-1,1]^2 at depths U(0.1,100) in the camera frame;
    noise = a random 2-D unit vector times sigma*0.001 (so every inlier sits exactly sigma/1000 from its projection);
    outliers: n*ratio draws of an index (with repetition), moved at least 0.002 + 3 sigma/1000 away."""
    R = rotation_from_quaternion(unit_vector(rng, 4))
    t = unit_vector(rng, 3)
    Pcw = np.eye(4)
    Pcw[:3, :3], Pcw[:3, 3] = R, t
    yn = rng.uniform(-1, 1, (n, 2))
    dist = rng.uniform(0.1, 100, n)
    xc = np.c_[yn, np.ones(n)] * dist[:, None]
    xs = (xc - t) @ R                                             # Pwc * xc = R^T (xc - t)
    pc = xs @ R.T + t
    yns_gt = pc[:, :2] / pc[:, 2:3]
    sigma = pixel_sigma * 0.001
    yns = yns_gt + np.stack([unit_vector(rng, 2) for _ in range(n)]) * sigma
    for _ in range(int(n * outlier_ratio)):
        i = int(rng.integers(0, n))
        y = yns[i].copy()
        if rng.uniform(0, 1) > 0.5:
            y = y + unit_vector(rng, 2)
        while np.linalg.norm(yns_gt[i] - y) < 0.002 + pixel_sigma * 0.001 * 3:
            y = y + unit_vector(rng, 2) * 0.1 * rng.uniform(3, 10)
        yns[i] = y
    return xs, yns, Pcw


def pose_error(P, Pcw):
    """test_pnp.cpp:98-99: I = P * Pcw^-1; error = |angle(I)| + |translation(I)|."""
    I = P @ np.linalg.inv(Pcw)
    ang = np.arccos(np.clip((np.trace(I[:3, :3]) - 1) / 2, -1, 1))
    return abs(ang) + np.linalg.norm(I[:3, 3])
