"""Pin the BOP results CSV (the hand-off to bop_toolkit's eval_siso.py, evaluate.py:276-282,323-336) against the consumer:
writes a CSV with THIS repository's ``evaluator.bop_csv_line`` for seeded poses, parses it with the REFERENCE's vendored
``bop_toolkit_lib.inout.load_bop_results`` (imported from /root/reference in the build container) and stores what the toolkit
read (tests/golden/bop_results_golden.npz) together with the CSV text.

    python tests/golden/make_bop_results_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/thirdparty/bop_toolkit")
np.float = float                                  # the vendored toolkit predates numpy 1.24
import types  # noqa: E402
for _absent in ("imageio", "png"):                # image codecs imported at the top of inout.py; the CSV reader does not touch them
    sys.modules.setdefault(_absent, types.ModuleType(_absent))

from bop_toolkit_lib import inout  # noqa: E402

from suo_slam_amd import evaluator  # noqa: E402
from suo_slam_amd import synthetic as S  # noqa: E402


def make_rows(seed=7, n=24):
    rng = np.random.default_rng(seed)
    rows = []
    for i in range(n):
        T = np.eye(4)
        T[:3, :3] = S.random_rotation(rng)
        T[:3, 3] = [rng.uniform(-300, 300), rng.uniform(-200, 200), rng.uniform(400, 1500)]
        if i % 5 == 0:
            T[:3, 3] = np.round(T[:3, 3])                                  # integral floats print as "123.0"
        if i % 7 == 0:
            T[:3, :3] = np.eye(3)                                          # exact ones and zeros
        if i % 11 == 0:
            T[:3, 3] *= 1e-5                                               # exponent notation
        rows.append((int(rng.integers(1, 21)), int(rng.integers(0, 500)), int(rng.integers(1, 31)), int(rng.integers(1, 60)), T))
    return rows


if __name__ == "__main__":
    rows = make_rows()
    text = "".join(evaluator.bop_csv_line(s, v, o, score, T) for s, v, o, score, T in rows)
    path = "/tmp/bop_results_golden.csv"
    with open(path, "w") as f:
        f.write(text)
    res = inout.load_bop_results(path)
    assert len(res) == len(rows)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "bop_results_golden.npz"),
                        csv=np.frombuffer(text.encode(), np.uint8),
                        ids=np.array([[r["scene_id"], r["im_id"], r["obj_id"]] for r in res], np.int64),
                        score=np.array([r["score"] for r in res]), time=np.array([r["time"] for r in res]),
                        R=np.stack([r["R"] for r in res]), t=np.stack([r["t"] for r in res]))
    print("toolkit parsed", len(res), "poses")
