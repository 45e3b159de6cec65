"""Record-and-replay harness: every call the host driver (ObjectSLAM / Evaluator) makes into libsuo_hip.so during a run
-- network forward, keypoint masks, batched PnP, LM / bundle adjustment -- is recorded with its inputs and outputs and
then replayed through the CPU oracle on the SAME inputs.  This checks a whole configuration (thresholds, modes, box
formats) end to end without needing the two sides to agree on anything upstream of each call."""
import contextlib

import numpy as np

STRIDE = 0x9E3779B97F4A7C15


class Recording:
    def __init__(self):
        self.forward, self.masks, self.pnp, self.ba = [], [], [], []


@contextlib.contextmanager
def record():
    """Patch the product's call sites (suo_slam_amd.object_slam / pkpnet) with recording wrappers."""
    from suo_slam_amd import object_slam as OS
    from suo_slam_amd import pkpnet as PN
    rec = Recording()
    orig_pnp, orig_ba, orig_fwd, orig_masks = OS._lt.pnp_batch, OS._ba.optimize_batch, PN.PkpNet.forward, PN.keypoint_masks

    def pnp_batch(xs, ys, thr=0.001, seed=0, **kw):
        out = orig_pnp(xs, ys, thr, seed=seed, **kw)
        rec.pnp.append({"xs": [np.array(x) for x in xs], "ys": [np.array(y) for y in ys], "thr": thr, "seed": seed,
                        "T": np.array(out[0]), "status": np.array(out[1])})
        return out

    def optimize_batch(problems):
        before = [{k: np.array(getattr(p, k)) for k in ("cam_T", "cam_fixed", "obj_T", "obj_fixed", "edge_cam", "edge_obj", "edge_camk", "edge_p",
                                                        "edge_uv", "edge_info", "inlier")} for p in problems]
        out = orig_ba(problems)
        for b, p in zip(before, problems):
            b.update(its=p.its, init_with_outliers=p.init_with_outliers,
                     out={"cam_T": np.array(p.cam_T), "obj_T": np.array(p.obj_T), "inlier": np.array(p.inlier), "chi2": np.array(p.chi2),
                          "stats": np.array(p.stats)})
            rec.ba.append(b)
        return out

    def forward(self, images, boxes, prior_kp=None, **kw):
        out = orig_fwd(self, images, boxes, prior_kp, **kw)
        rec.forward.append({"image": np.array(images), "boxes": np.asarray(boxes[0].cpu() if hasattr(boxes[0], "cpu") else boxes[0], dtype=np.float32), "prior_uv": kw.get("prior_uv"),
                            "prior_mask": kw.get("prior_mask"),
                            "out": {k: out[k].cpu().numpy() for k in ("uv", "cov", "kp_mask", "prob_logits")}})
        return out

    def keypoint_masks(uv, cov, kp, mm, bt=0.9, vt=0.2):
        out = orig_masks(uv, cov, kp, mm, bt, vt)
        rec.masks.append({"uv": uv.cpu().numpy(), "cov": cov.cpu().numpy(), "kp": kp.cpu().numpy(), "mm": None if mm is None else np.array(mm),
                          "bt": bt, "vt": vt, "out": out.cpu().numpy().astype(bool)})
        return out
    OS._lt.pnp_batch, OS._ba.optimize_batch, PN.PkpNet.forward, PN.PkpNet.__call__, PN.keypoint_masks = pnp_batch, optimize_batch, forward, forward, keypoint_masks
    try:
        yield rec
    finally:
        OS._lt.pnp_batch, OS._ba.optimize_batch, PN.PkpNet.forward, PN.PkpNet.__call__, PN.keypoint_masks = orig_pnp, orig_ba, orig_fwd, orig_fwd, orig_masks


def _rot_close(a, b, tol_R, tol_t):
    a, b = np.asarray(a).reshape(3, 4), np.asarray(b).reshape(3, 4)
    return np.linalg.norm(a[:, :3] - b[:, :3]) < tol_R and np.linalg.norm(a[:, 3] - b[:, 3]) < tol_t * max(1.0, np.linalg.norm(b[:, 3]))


def check_pnp(rec, tol=1e-8):
    """Every recorded suo_pnp_batch call vs the C oracle, object by object (same counter-based sampler seeds)."""
    from oracle import geometry as G
    n = 0
    for c in rec.pnp:
        for j, (x, y) in enumerate(zip(c["xs"], c["ys"])):
            T, best, its = G.pnp(x, y, c["thr"], seed=(c["seed"] + j * STRIDE) % 2 ** 64)
            assert np.abs(T - c["T"][j]).max() < tol * max(1.0, np.abs(T).max()), (j, np.abs(T - c["T"][j]).max())
            assert int(np.allclose(T, np.eye(4))) == int(c["status"][j])
            n += 1
    return n


def check_ba(rec, tol_R=1e-6, tol_t=1e-6):
    """Every recorded suo_optimize call vs the dense-Cholesky C oracle: inlier flags and round counts exact, poses close."""
    from oracle import geometry as G
    n = 0
    for b in rec.ba:
        ref = G.optimize(b["cam_T"].reshape(-1, 3, 4), b["cam_fixed"], b["obj_T"].reshape(-1, 3, 4), b["obj_fixed"], b["edge_cam"], b["edge_obj"],
                         b["edge_camk"], b["edge_p"], b["edge_uv"], b["edge_info"], b["inlier"], its=b["its"],
                         init_with_outliers=b["init_with_outliers"])
        out = b["out"]
        assert np.array_equal(out["inlier"], ref[2]), (n, int((out["inlier"] != ref[2]).sum()))
        assert out["stats"][0] == ref[4][0] and out["stats"][3] == ref[4][3], (out["stats"], ref[4])
        for got, want in ((out["cam_T"].reshape(-1, 3, 4), ref[0]), (out["obj_T"].reshape(-1, 3, 4), ref[1])):
            for a, w in zip(got, want):
                assert _rot_close(a, w, tol_R, tol_t), (n, np.abs(a - w).max())
        n += 1
    return n


def check_network(rec, sd, max_calls=None, logit_tol=2e-4, uv_tol=2e-4):
    """Recorded forwards (no priors) vs the torch-CPU oracle on the same pixels and boxes; recorded mask calls vs the oracle's
    mask logic on the SAME uv / cov / kp_mask (bit-exact: identical float32 inputs, float32 comparisons)."""
    from oracle import cnn_oracle as O
    P = O.to_torch(sd)
    n = 0
    for c in rec.forward[:max_calls]:
        if c["prior_uv"] is not None:
            continue
        ref = O.pkpnet_forward(c["image"], c["boxes"], None, sd, P)
        lr = ref["prob_logits"].numpy()
        assert np.abs(c["out"]["prob_logits"] - lr).max() < logit_tol * np.abs(lr).max()
        assert np.abs(c["out"]["uv"] - ref["uv"].numpy()).max() < uv_tol
        assert np.abs(c["out"]["cov"] - ref["cov"].numpy()).max() < uv_tol
        assert np.abs(c["out"]["kp_mask"] - ref["kp_mask"].numpy()).max() < uv_tol
        n += 1
    for m in rec.masks:
        want = O.keypoint_masks(m["uv"], m["cov"], m["kp"], m["mm"], m["bt"], m["vt"])
        assert np.array_equal(m["out"], want)
    return n
