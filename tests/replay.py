"""Record-and-replay harness: every call the host driver (ObjectSLAM / Evaluator) makes into libsuo_hip.so during a run
-- network forward, keypoint masks, batched PnP, LM / bundle adjustment -- is recorded with its inputs and outputs and
then replayed through the CPU oracle on the SAME inputs.  This checks a whole configuration (thresholds, modes, box
formats) end to end without needing the two sides to agree on anything upstream of each call."""
import contextlib

import numpy as np

STRIDE = 0x9E3779B97F4A7C15


class Recording:
    def __init__(self):
        self.forward, self.masks, self.pnp, self.ba, self.chain = [], [], [], [], []


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
        # (since round 4 ObjectSLAM uploads the frame once per view and hands the DEVICE tensor to both network passes)
        rec.forward.append({"image": images.cpu().numpy() if hasattr(images, "cpu") else np.array(images), "boxes": np.asarray(boxes[0].cpu() if hasattr(boxes[0], "cpu") else boxes[0], dtype=np.float32), "prior_uv": _host(kw.get("prior_uv")),
                            "prior_mask": _host(kw.get("prior_mask")),
                            "out": {k: out[k].cpu().numpy() for k in ("uv", "cov", "kp_mask", "prob_logits")}})
        return out

    def _host(t):
        return t.cpu().numpy() if hasattr(t, "cpu") else t

    def keypoint_masks(uv, cov, kp, mm, bt=0.9, vt=0.2):
        out = orig_masks(uv, cov, kp, mm, bt, vt)
        rec.masks.append({"uv": uv.cpu().numpy(), "cov": cov.cpu().numpy(), "kp": kp.cpu().numpy(), "mm": None if mm is None else (mm.cpu().numpy() if hasattr(mm, "cpu") else np.array(mm)),
                          "bt": bt, "vt": vt, "out": out.cpu().numpy().astype(bool)})
        return out
    # the device-resident frame chain (suo_frame_geom_*): what it was launched on and what it read back
    from suo_slam_amd import frame_geom as FG
    orig_launch, orig_fetch = FG.FrameGeometry.launch, FG.FrameGeometry.fetch

    def launch(self, frame_first, uv_dev, cov_dev, mask_dev, model_kps_dev, kinv, camk, min_depth, seed=0, use_cov=True, do_lm=True,
               its=(10, 10, 40, 40), pnp_threshold=1e-3, stream=None, seed_dev=None):
        # (seed_dev: the device-resident running key of chains that enqueue several launches before reading any back -- its value when this launch's PnP
        #  runs is what everything enqueued before it left there: read it now, the stream is in order)
        run = int(seed_dev.cpu().item()) if seed_dev is not None else 0
        self._rec = {"frame_first": np.array(frame_first), "uv": uv_dev.cpu().numpy(), "cov": cov_dev.cpu().numpy(),
                     "mask": mask_dev.cpu().numpy().astype(bool), "model_kps": model_kps_dev.cpu().numpy(), "kinv": np.array(kinv),
                     "camk": np.array(camk), "min_depth": np.array(min_depth), "seed": int(seed) + run, "use_cov": bool(use_cov), "do_lm": bool(do_lm),
                     "its": tuple(its), "thr": pnp_threshold}
        return orig_launch(self, frame_first, uv_dev, cov_dev, mask_dev, model_kps_dev, kinv, camk, min_depth, seed=seed, use_cov=use_cov,
                           do_lm=do_lm, its=its, pnp_threshold=pnp_threshold, stream=stream, seed_dev=seed_dev)

    def fetch(self, copy=True):
        out = orig_fetch(self, True)
        c = dict(self._rec)
        c["out"] = out
        rec.chain.append(c)
        return out
    OS._lt.pnp_batch, OS._ba.optimize_batch, PN.PkpNet.forward, PN.PkpNet.__call__, PN.keypoint_masks = pnp_batch, optimize_batch, forward, forward, keypoint_masks
    FG.FrameGeometry.launch, FG.FrameGeometry.fetch = launch, fetch
    try:
        yield rec
    finally:
        OS._lt.pnp_batch, OS._ba.optimize_batch, PN.PkpNet.forward, PN.PkpNet.__call__, PN.keypoint_masks = orig_pnp, orig_ba, orig_fwd, orig_fwd, orig_masks
        FG.FrameGeometry.launch, FG.FrameGeometry.fetch = orig_launch, orig_fetch


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


def check_chain(rec, tol_pnp=1e-8, tol_R=1e-6, tol_t=1e-6):
    """Every recorded device-chain launch (csrc/frame_geom.hip) vs the CPU oracles on the SAME device inputs: compaction in mask
    order, K^-T normalisation, PnP per crop with the chain's sampler keys, acceptance, the frame's graph (fp64 inverse of the float32
    covariance as information) through the dense-Cholesky LM oracle.  Returns (#PnP problems, #frames with an LM) checked."""
    from oracle import geometry as G
    n_pnp = n_lm = 0
    for c in rec.chain:
        out, ff = c["out"], c["frame_first"]
        assert np.array_equal(out["mask"], c["mask"]) and np.array_equal(out["uv"], c["uv"]) and np.array_equal(out["cov"], c["cov"])
        before = 0
        for f in range(len(ff) - 1):
            rank, objs, init = 0, [], []
            for g in range(ff[f], ff[f + 1]):
                m = c["mask"][g]
                n = int(m.sum())
                assert out["n_kp"][g] == n
                if n < 4:
                    assert out["pnp_status"][g] == 1 and not out["accepted"][g]
                    continue
                xs = c["model_kps"][g][m].astype(np.float64)
                uv = c["uv"][g][m].astype(np.float64)
                k = c["kinv"][g]
                ys = np.stack([(uv[:, 0] * k[0] + uv[:, 1] * k[1]) + k[2], (uv[:, 0] * k[3] + uv[:, 1] * k[4]) + k[5]], 1)
                T, best, its = G.pnp(xs, ys, c["thr"], seed=(c["seed"] + before + rank * STRIDE) % 2 ** 64)
                rank += 1
                assert np.abs(T - out["T_pnp"][g]).max() < tol_pnp * max(1.0, np.abs(T).max()), (g, np.abs(T - out["T_pnp"][g]).max())
                ident = bool(np.allclose(T, np.eye(4)))
                assert int(ident) == int(out["pnp_status"][g])
                acc = (not ident) and T[2, 3] > c["min_depth"][g]
                assert bool(out["accepted"][g]) == acc
                n_pnp += 1
                if acc:
                    objs.append(g)
                    init.append(T[:3])
            before += rank
            if not c["do_lm"] or not objs:
                continue
            cnt = [int(c["mask"][g].sum()) for g in objs]
            e_obj = np.concatenate([np.full(n, j, np.int32) for j, n in enumerate(cnt)])
            camk = np.concatenate([np.tile(c["camk"][g], (n, 1)) for g, n in zip(objs, cnt)])
            p = np.concatenate([c["model_kps"][g][c["mask"][g]].astype(np.float64) for g in objs])
            uv = np.concatenate([c["uv"][g][c["mask"][g]].astype(np.float64) for g in objs])
            if c["use_cov"]:
                cv = np.concatenate([c["cov"][g][c["mask"][g]].astype(np.float64) for g in objs])
                det = cv[:, 0, 0] * cv[:, 1, 1] - cv[:, 0, 1] * cv[:, 1, 0]
                info = np.stack([cv[:, 1, 1] / det, 0.5 * (-cv[:, 0, 1] / det + -cv[:, 1, 0] / det), cv[:, 0, 0] / det], 1)
            else:
                info = np.tile([1.0, 0.0, 1.0], (len(p), 1))
            ref = G.optimize(np.eye(4)[None, :3], np.array([1], np.uint8), np.array(init), np.zeros(len(objs), np.uint8), np.zeros(len(p), np.int32),
                             e_obj, camk, p, uv, info, np.ones(len(p), np.uint8), its=c["its"])
            k0 = 0
            for j, g in enumerate(objs):
                assert _rot_close(out["T_opt"][g], ref[1][j], tol_R, tol_t), (g, np.abs(out["T_opt"][g] - ref[1][j]).max())
                assert np.array_equal(out["inlier"][g, :cnt[j]], ref[2][k0:k0 + cnt[j]].astype(bool)), g
                k0 += cnt[j]
            assert out["lm_stats"][f][0] == ref[4][0] and out["lm_stats"][f][3] == ref[4][3]
            n_lm += 1
    return n_pnp, n_lm
