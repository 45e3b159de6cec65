"""One STEP of bench.py's timed region: the per-frame product path through the C ABI (include/suo_hip.h), `depth` steps in flight."""
import ctypes as C
import os

import numpy as np

from .common import BBOX_THRESH, KP_VAR_THRESH, confident_state_dict


class FramePipeline:
    """The per-frame product path through the C ABI.  One STEP on a slot's stream, no host wait anywhere inside:
        host prep of the step's small arrays (K_bbox terms) -> async H2D of the F frames (pinned) + boxes + model keypoints / masks ->
        suo_net_forward_frames -> suo_keypoint_masks -> suo_frame_geom_launch (compaction, PnP, acceptance, graph, LM, one D2H).
    `depth` steps are in flight: step i runs on slot i % depth (own network workspace, hipGraph, stream, geometry context,
    device + pinned staging); before a slot is reused the host fetches the results of the step that used it.  Every call
    processes exactly the frames that are counted; nothing is cached across steps."""

    def __init__(self, L, pool, F, use_graph=True, only="all", depth=2, state_dict=None, gt_keypoints=False, resident=False):
        import torch
        from suo_slam_amd import _lib
        from suo_slam_amd.frame_geom import FrameGeometry
        from suo_slam_amd.pkpnet import PkpNet
        self.torch, self.lib, self._lib = torch, _lib.lib(), _lib
        self.only, self.L, self.F, self.depth, self.gt = only, L, F, depth, gt_keypoints
        sd = state_dict if state_dict is not None else confident_state_dict()
        self.pool = pool
        assert len(pool) % F == 0, "--pool must be a multiple of --frames-per-step"
        self.n_groups = len(pool) // F
        LF = L * F
        # the dataset side: frames in pinned host memory (what a loader thread would hand over)
        self.h_imgs = torch.from_numpy(np.stack([fr["image"] for fr in pool])).pin_memory()
        self.H, self.W = int(self.h_imgs.shape[1]), int(self.h_imgs.shape[2])      # 480 x 640 (YCB-V) or 540 x 720 (T-LESS)
        dev = "cuda"
        # resident: the pool's frames already in HBM when a step starts (the frames_resident_in_hbm leg); the network reads them where they lie
        self.d_imgs = self.h_imgs.to(dev) if resident else None
        self.slots = []
        for _ in range(depth):
            net = PkpNet(state_dict=sd, max_crops=LF)
            net.set_graph(use_graph)
            ts = torch.cuda.Stream()      # a real (non-NULL) stream: hipGraph replay is then fully asynchronous
            S = {"net": net, "tstream": ts, "stream": C.c_void_p(ts.cuda_stream), "busy": None, "fg": FrameGeometry(LF, F),
                 "imgs": torch.empty((F, self.H, self.W, 3), dtype=torch.uint8, device=dev),
                 "uv": torch.empty((LF, 41, 2), device=dev), "cov": torch.empty((LF, 41, 2, 2), device=dev),
                 "kp": torch.empty((LF, 41), device=dev), "mask": torch.empty((LF, 41), dtype=torch.uint8, device=dev),
                 "boxes": torch.empty((LF, 4), device=dev), "box_img": torch.arange(F, dtype=torch.int32, device=dev).repeat_interleave(L),
                 "mm": torch.empty((LF, 41), dtype=torch.uint8, device=dev), "kps": torch.empty((LF, 41, 3), device=dev),
                 "h_boxes": torch.empty((LF, 4)).pin_memory(), "h_mm": torch.empty((LF, 41), dtype=torch.uint8).pin_memory(),
                 "h_kps": torch.empty((LF, 41, 3)).pin_memory()}
            self.slots.append(S)
        self.first = np.arange(F + 1, dtype=np.int32) * L
        self.reset_metrics()

    def reset_metrics(self):
        self.n_frames = self.n_crops = self.n_kp = self.n_pose = self.n_inl = self.n_trials = 0
        self.pose_err, self.n_pose_gt = 0.0, 0

    def step(self, i):
        from suo_slam_amd import geometry as geo
        from suo_slam_amd.frame_geom import kbbox_terms
        torch = self.torch
        S = self.slots[i % self.depth]
        self.retire(S)
        g = i % self.n_groups
        frames = self.pool[g * self.F:(g + 1) * self.F]
        L, LF = self.L, self.L * self.F
        P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        # ---- host side of the step (lib/object_slam.py:1082-1098): per-box intrinsics in the reference's float32 container
        boxes = np.concatenate([fr["boxes"] for fr in frames]).astype(np.float32)
        K_bbox = np.concatenate([geo.fix_K_for_bbox_ndc_many(fr["K"], fr["boxes"].astype(np.float64)) for fr in frames]).astype(np.float32)
        kinv, camk = kbbox_terms(K_bbox)
        min_depth = 0.5 * np.concatenate([fr["diameter"] for fr in frames])
        S["h_boxes"].numpy()[:] = boxes
        S["h_mm"].numpy()[:] = np.concatenate([fr["model_kps_masks"] for fr in frames])
        S["h_kps"].numpy()[:] = np.concatenate([fr["model_kps"] for fr in frames])
        # the frames' H2D (0.92 MB each) and the small per-crop arrays, from pinned memory, stream-ordered (suo_upload: a copy kernel --
        # an asynchronous hipMemcpy in front of the network makes the next host-side wait on this stack take 10-20 ms)
        src = self.h_imgs[g * self.F:(g + 1) * self.F]
        imgs = S["imgs"] if self.d_imgs is None else self.d_imgs[g * self.F:(g + 1) * self.F]
        for dst, h in ((S["imgs"], src), (S["boxes"], S["h_boxes"]), (S["mm"], S["h_mm"]), (S["kps"], S["h_kps"])):
            if dst is S["imgs"] and self.d_imgs is not None:
                continue
            self._lib.check(self.lib.suo_upload(P(dst), C.c_void_p(h.data_ptr()), dst.numel() * dst.element_size(), S["stream"]), "suo_upload")
        self._lib.check(self.lib.suo_net_forward_frames(S["net"]._h, P(imgs), 0, self.H, self.W, P(S["boxes"]), P(S["box_img"]), LF, None,
                                                        P(S["uv"]), P(S["cov"]), P(S["kp"]), None, None, S["stream"]), "suo_net_forward_frames")
        self._lib.check(self.lib.suo_keypoint_masks(P(S["uv"]), P(S["cov"]), P(S["kp"]), P(S["mm"]), LF, BBOX_THRESH, KP_VAR_THRESH, P(S["mask"]),
                                                    S["stream"]), "suo_keypoint_masks")
        if self.gt:
            # pose check only: overwrite what the network said with the frames' projected ground-truth keypoints + noise
            with torch.cuda.stream(S["tstream"]):
                S["uv"].copy_(torch.from_numpy(np.concatenate([fr["uv"] for fr in frames])), non_blocking=False)
                S["cov"].copy_(torch.from_numpy(np.concatenate([fr["cov"] for fr in frames])), non_blocking=False)
                S["mask"].copy_(S["mm"])
        if self.only != "cnn":
            gs = S["tstream"]
            mode = os.environ.get("SUO_BENCH_GEOM_STREAM", "0")
            if mode != "0":
                if not hasattr(self, "gstreams"):
                    n = {"1": 1, "2": 2}.get(mode, 1)
                    self.gstreams = [torch.cuda.Stream(priority=-1) for _ in range(n)]
                if "nev" not in S:
                    S["nev"] = torch.cuda.Event()
                gs = self.gstreams[(i % self.depth) % len(self.gstreams)]
                S["nev"].record(S["tstream"])
                gs.wait_event(S["nev"])
            S["fg"].launch(self.first, S["uv"], S["cov"], S["mask"], S["kps"], kinv, camk, min_depth, seed=i, use_cov=True, do_lm=True,
                           its=(10, 10, 40, 40), stream=gs.cuda_stream)
        else:
            S["ev"] = torch.cuda.Event()
            S["ev"].record(S["tstream"])
        S["busy"] = (g, i)

    def retire(self, S):
        """Fetch the results of the step that last used this slot (the ONE read-back of the step) and account for them."""
        if S["busy"] is None:
            return None
        (g, i), S["busy"] = S["busy"], None
        self.n_frames += self.F
        self.n_crops += self.L * self.F
        if self.only == "cnn":
            S["ev"].synchronize()
            self.check_range(S)
            return None
        r = S["fg"].fetch(copy=False)
        self.check_range(S)
        assert np.isfinite(r["uv"]).all() and np.isfinite(r["T_opt"][r["accepted"]]).all()
        self.n_kp += int(r["n_kp"].sum())
        self.n_pose += int(r["accepted"].sum())
        self.n_inl += int(np.count_nonzero(r["inlier"][r["accepted"]] & (np.arange(41)[None, :] < r["n_kp"][r["accepted"], None])))
        self.n_trials += int(r["lm_stats"][:, 2].sum())
        if self.gt:
            frames = self.pool[g * self.F:(g + 1) * self.F]
            gt = np.concatenate([fr["T_OtoC"] for fr in frames])
            ok = r["accepted"]
            d = np.linalg.norm(r["T_opt"][:, :, 3] - gt[:, :3, 3], axis=1) / gt[:, 2, 3]
            self.pose_err += float(d[ok].sum())
            self.n_pose_gt += int(ok.sum())
        return r

    @staticmethod
    def check_range(S):
        """The fp16 form's contract (include/suo_hip.h: suo_net_range_exceeded): a step whose activations left fp16's range has invalid outputs and would have
        to be re-issued on bf16x3.  The synthetic weights sit 200x inside the range (profiles/r05_activation_range.txt): if this fires the measurement is void."""
        if S["net"].range_exceeded():
            raise RuntimeError("an activation left the fp16 range inside the timed region: the line would not be a measurement of the fp16 form")

    def drain(self, next_step):
        """Retire every step still in flight, oldest first."""
        for k in range(self.depth):
            self.retire(self.slots[(next_step + k) % self.depth])
