"""Host-side mirror of the reference's ``PkpNet`` (/root/reference/lib/models/pkpnet.py:65-119).

Same constructor / ``forward(images, boxes, prior_kp)`` / ``load_state_dict`` surface, but every
operator runs in libsuo_hip.so (hand-written gfx950 kernels).  PyTorch is used only to own device
memory and the current HIP stream.  There is no CPU fallback: a missing extension or GPU raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .weights import NUM_KP

HEAT = 64


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(_lib.current_stream_ptr())


class PkpNet:
    """Keypoint / uncertainty network (eval mode only)."""

    def __init__(self, input_res=(256, 256), calc_cov=True, state_dict=None, max_crops=16, device="cuda:0"):
        assert tuple(input_res) == (256, 256), "the HIP path is built for 256x256 crops (lib/datasets/bop.py:21)"
        assert calc_cov, "covariance is always computed on the HIP path"
        self.input_res = tuple(input_res)
        self.calc_cov = True
        self.num_kp = NUM_KP
        self.max_crops = int(max_crops)
        self.device = torch.device(device)
        self._h = None
        if state_dict is not None:
            self.load_state_dict(state_dict)

    # -- reference surface -------------------------------------------------------------------------
    def load_state_dict(self, state_dict, strict=True):
        """Accepts ``checkpoint['model']`` of the reference (torch tensors or numpy arrays)."""
        lib = _lib.lib()
        _lib.require_gpu()
        torch.cuda.set_device(self.device)
        arrs = {}
        for k, v in state_dict.items():
            if k.endswith("num_batches_tracked"):
                continue
            a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            arrs[k] = np.ascontiguousarray(a, dtype=np.float32)
        n = len(arrs)
        names = (C.c_char_p * n)(*[k.encode() for k in arrs])
        data = (C.c_void_p * n)(*[a.ctypes.data for a in arrs.values()])
        shapes_keep = [(C.c_int64 * max(a.ndim, 1))(*(a.shape if a.ndim else (1,))) for a in arrs.values()]
        shapes = (C.c_void_p * n)(*[C.cast(s, C.c_void_p).value for s in shapes_keep])
        ndims = (C.c_int * n)(*[max(a.ndim, 1) for a in arrs.values()])
        h = C.c_void_p()
        _lib.check(lib.suo_net_create(n, names, data, shapes, ndims, self.max_crops, C.byref(h)), "suo_net_create")
        self.close()
        self._h = h
        return self

    def eval(self):
        return self

    def cuda(self):
        return self

    def close(self):
        if self._h is not None:
            _lib.lib().suo_net_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_graph(self, enable: bool):
        _lib.check(_lib.lib().suo_net_set_graph(self._h, int(enable)), "suo_net_set_graph")

    def prepare(self, crop_counts=None, with_priors=(False, True), fallback_pipe=True):
        """Capture the backbone graphs for these crop counts (default: 1..max_crops) before the first frame arrives, so that no
        frame with a not-yet-seen number of detections pays a graph capture (suo_net_prepare).  fallback_pipe: on the fp16 form also capture the graphs of the
        bf16x3 form the network moves to when an activation leaves fp16's range -- the first such view then pays a re-issued forward, not a capture."""
        counts = list(range(1, self.max_crops + 1) if crop_counts is None else crop_counts)
        pipes = [self.pipe()] + ([1] if (fallback_pipe and self.pipe() == 2) else [])
        for p in reversed(pipes):                          # (the form in use last: the network is left on it)
            if p != self.pipe():
                self.set_pipe(p)
            for L in counts:
                for wp in with_priors:
                    _lib.check(_lib.lib().suo_net_prepare(self._h, int(L), int(bool(wp)), _stream()), "suo_net_prepare")
        if self.pipe() != pipes[0]:
            self.set_pipe(pipes[0])

    def workspace_bytes(self):
        return int(_lib.lib().suo_net_workspace_bytes(self._h))

    SCHEDULE_KINDS = ("staging_stem", "conv3x3_and_fused_tails", "gemm_1x1", "one_launch_blocks", "pool_upsample", "decode_classifier")

    def schedule_bytes(self, n_crops, n_frames=1, H=480, W=640, with_priors=False):
        """Algorithmic HBM bytes of ONE call of n_crops crops (suo_net_schedule_bytes: a dry run of the launch schedule on the current pipe, nothing runs):
        {kind: bytes, ..., "total": bytes, "launches": n}."""
        import ctypes as C
        b, n = (C.c_double * 6)(), C.c_int(0)
        _lib.check(_lib.lib().suo_net_schedule_bytes(self._h, int(n_crops), int(n_frames), int(H), int(W), int(bool(with_priors)), b, C.byref(n)), "suo_net_schedule_bytes")
        out = {k: float(v) for k, v in zip(self.SCHEDULE_KINDS, b)}
        out["total"] = float(sum(b))
        out["launches"] = int(n.value)
        return out

    # -- matrix pipe and the fp16 form's range guard (include/suo_hip.h: SUO_PIPE_*, suo_net_range_exceeded) -------------------------
    def pipe(self):
        """0 = fp32 MFMA, 1 = three bf16 terms, 2 = two fp16 terms (the default; range-guarded)."""
        return int(_lib.lib().suo_net_get_pipe(self._h))

    def set_pipe(self, pipe):
        _lib.check(_lib.lib().suo_net_set_pipe(self._h, int(pipe)), "suo_net_set_pipe")

    def range_exceeded(self):
        """True when a forward since the last call of this method left the fp16 range: every forward of this network that was enqueued and not yet
        checked is INVALID and must be re-issued (the network is on the bf16x3 form from now on).  The caller has synchronised on the outputs.
        forward(..., check=True) -- the default -- does all of this itself."""
        return bool(_lib.lib().suo_net_range_exceeded(self._h))

    def forward(self, images, boxes, prior_kp=None, want_prob=False, prior_uv=None, prior_mask=None, check=True, out_slot=None):
        """images: uint8 [H,W,3] (cv2 layout) or float32 [1,3,H,W] device/host tensor; boxes: list with
        one Tensor[L,4] (xyxy); prior_kp: list with one Tensor[L,41,256,256] or None.
        Returns the reference's dict: uv, cov, prob_logits, kp_mask_logits, kp_mask (+ prob if asked).
        check: on the fp16 form, wait for the call and re-issue it on bf16x3 if an activation left the range (the reference's callers read the
        outputs back right away, lib/object_slam.py:1100-1111, so the wait costs them nothing); check=False returns at once -- the caller then
        asks range_exceeded() after its own synchronisation and re-issues (what ObjectSLAM's device chains do).
        out_slot: any hashable -- the output tensors of (crop count, slot) are allocated once and handed out again by later calls naming the same slot (a caller that
        has consumed the previous call's outputs: five allocations per call saved on a path whose host time is on the critical path)."""
        assert self._h is not None, "load_state_dict first"
        assert isinstance(boxes, (list, tuple)) and len(boxes) == 1, "one image per call (lib/object_slam.py:1092-1099)"
        dev = self.device
        if isinstance(images, np.ndarray):
            images = torch.from_numpy(images)
        if images.dtype == torch.uint8:
            assert images.dim() == 3 and images.shape[2] == 3
            fmt, H, W = 0, int(images.shape[0]), int(images.shape[1])
        else:
            assert images.dim() == 4 and images.shape[0] == 1 and images.shape[1] == 3
            images = images.to(torch.float32)
            fmt, H, W = 1, int(images.shape[2]), int(images.shape[3])
        img = self._to_device(images)
        bx = torch.as_tensor(boxes[0], dtype=torch.float32).to(dev).contiguous()
        L = int(bx.shape[0])
        pr = None
        if prior_kp is not None:
            assert prior_uv is None, "give the dense prior heat-maps OR the prior keypoints, not both"
            pr = torch.cat([torch.as_tensor(p, dtype=torch.float32) for p in prior_kp]).to(dev).contiguous()
            assert tuple(pr.shape) == (L, NUM_KP, 256, 256)
        cached = None
        if out_slot is not None:
            cache = self.__dict__.setdefault("_out_cache", {})
            cached = cache.get((L, out_slot))
        if cached is None:
            uv = torch.empty((L, NUM_KP, 2), dtype=torch.float32, device=dev)
            cov = torch.empty((L, NUM_KP, 2, 2), dtype=torch.float32, device=dev)
            kpm = torch.empty((L, NUM_KP), dtype=torch.float32, device=dev)
            kpl = torch.empty((L, NUM_KP), dtype=torch.float32, device=dev)
            logits = torch.empty((L, NUM_KP, HEAT, HEAT), dtype=torch.float32, device=dev)
            if out_slot is not None:
                cache[(L, out_slot)] = (uv, cov, kpm, kpl, logits)
        else:
            uv, cov, kpm, kpl, logits = cached
        if prior_uv is not None:
            # the prior heat-maps are rendered on the device from the projected keypoints (suo_net_forward_prior_kp):
            # prior_uv [L,41,2] NDC, prior_mask [L,41] -- what make_prior_kp_input takes per object (utils.py:398-411)
            if isinstance(prior_uv, torch.Tensor) and prior_uv.is_cuda:      # already on the device (stage_block: one upload for every small array of the call)
                puv, pmk = prior_uv.reshape(L, NUM_KP, 2).contiguous(), prior_mask.reshape(L, NUM_KP).contiguous()
                assert puv.dtype == torch.float32 and pmk.dtype == torch.uint8
            else:
                puv = torch.as_tensor(np.asarray(prior_uv, dtype=np.float32)).reshape(L, NUM_KP, 2).to(dev).contiguous()
                pmk = torch.as_tensor(np.asarray(prior_mask, dtype=np.uint8)).reshape(L, NUM_KP).to(dev).contiguous()
            _lib.check(_lib.lib().suo_net_forward_prior_kp(self._h, _ptr(img), fmt, H, W, _ptr(bx), None, L, _ptr(puv), _ptr(pmk), _ptr(uv),
                                                           _ptr(cov), _ptr(kpm), _ptr(kpl), _ptr(logits), _stream()), "suo_net_forward_prior_kp")
        else:
            _lib.check(_lib.lib().suo_net_forward(self._h, _ptr(img), fmt, H, W, _ptr(bx), L, _ptr(pr), _ptr(uv), _ptr(cov),
                                                  _ptr(kpm), _ptr(kpl), _ptr(logits), _stream()), "suo_net_forward")
        if check and self.pipe() == 2:
            torch.cuda.current_stream().synchronize()
            if self.range_exceeded():
                return self.forward(images, boxes, prior_kp, want_prob, prior_uv, prior_mask, check=False, out_slot=out_slot)     # (now on bf16x3: fp32's range)
        ret = {"uv": uv, "cov": cov, "prob_logits": logits, "kp_mask_logits": kpl, "kp_mask": kpm}
        if want_prob:
            ret.update(decode_extras(logits))
        return ret

    __call__ = forward

    def _to_device(self, images):
        """Host frame -> device through a pinned staging buffer and the stream-ordered copy kernel (suo_upload) instead of a blocking
        pageable hipMemcpy; device tensors pass through.  The host-side copy into the staging buffer is a plain memmove ON PURPOSE: a
        torch CPU copy_ of a frame wakes the whole intra-op thread pool, and on a box whose cgroup grants fewer CPUs than it shows
        (the GPU boxes here: 256 visible, quota 16) the spinning pool eats the quota and every later host-side wait stalls for tens of
        milliseconds (measured: 36 ms instead of 5.7 ms per process_view)."""
        if images.is_cuda:
            return images.contiguous()
        images = images.contiguous()
        nbytes = images.numel() * images.element_size()
        st = getattr(self, "_stage", None)
        if st is None or st[0].numel() < nbytes:
            st = (torch.empty(nbytes, dtype=torch.uint8).pin_memory(), torch.empty(nbytes, dtype=torch.uint8, device=self.device), torch.cuda.Event())
            self._stage = st
        host, dev_buf, ev = st
        ev.synchronize()                                    # the previous upload has read the pinned buffer
        C.memmove(host.data_ptr(), images.data_ptr(), nbytes)
        _lib.check(_lib.lib().suo_upload(_ptr(dev_buf), C.c_void_p(host.data_ptr()), nbytes, _stream()), "suo_upload")
        ev.record()
        return dev_buf[:nbytes].view(images.dtype).view(images.shape)

    def stage_block(self, arrays):
        """Host arrays -> device tensors through ONE pinned block and ONE stream-ordered copy kernel (suo_upload), from a ring of two blocks: nothing here
        waits for the device except for the upload that used the same block two calls ago.  (A pageable .to(device) is ordered behind everything already
        enqueued on the stream AND blocks the host until it ran: with a second batch in flight that is a full device round trip per small array.)"""
        arrays = [np.ascontiguousarray(a) for a in arrays]
        offs, total = [], 0
        for a in arrays:
            offs.append(total)
            total += (a.nbytes + 15) & ~15
        ring = getattr(self, "_block_ring", None)
        if ring is None:
            ring = self._block_ring = {"slots": [None, None], "next": 0}
        k = ring["next"]
        ring["next"] = (k + 1) % 2
        st = ring["slots"][k]
        if st is None or st[0].numel() < total:
            if st is not None:
                st[2].synchronize()
            cap = max(total, 1 << 20)
            st = ring["slots"][k] = (torch.empty(cap, dtype=torch.uint8).pin_memory(), torch.empty(cap, dtype=torch.uint8, device=self.device), torch.cuda.Event())
        host, dev_buf, ev = st
        ev.synchronize()                                    # the upload that last read this pinned block has run
        base = host.data_ptr()
        for a, o in zip(arrays, offs):
            if a.nbytes:
                C.memmove(base + o, a.ctypes.data, a.nbytes)
        _lib.check(_lib.lib().suo_upload(_ptr(dev_buf), C.c_void_p(base), total, _stream()), "suo_upload")
        ev.record()
        tdt = {np.dtype(np.uint8): torch.uint8, np.dtype(np.float32): torch.float32, np.dtype(np.int32): torch.int32, np.dtype(np.bool_): torch.uint8,
               np.dtype(np.float64): torch.float64, np.dtype(np.int64): torch.int64}
        return [dev_buf[o:o + a.nbytes].view(tdt[a.dtype]).view(a.shape) for a, o in zip(arrays, offs)]

    def forward_frames(self, images, boxes_per_frame, check=True, extra=None):
        """Several independent frames in one call: images uint8 [B,H,W,3] (or a list of B frames), boxes_per_frame list of B arrays [L_b,4].
        Returns the same dict with the crops of all frames concatenated in frame order.  check: as forward().
        extra: host arrays to put on the device with the same upload (returned as ret["extra"], device tensors): what the caller's next kernels need."""
        assert self._h is not None, "load_state_dict first"
        dev = self.device
        if isinstance(images, torch.Tensor) and images.is_cuda:
            imgs = images.contiguous()
            bx = torch.cat([torch.as_tensor(b, dtype=torch.float32).reshape(-1, 4) for b in boxes_per_frame]).to(dev).contiguous()
            idx = torch.cat([torch.full((len(b),), i, dtype=torch.int32) for i, b in enumerate(boxes_per_frame)]).to(dev).contiguous()
            extra_dev = [torch.as_tensor(np.ascontiguousarray(a)).to(dev) for a in (extra or [])]
        else:
            frames = np.stack([np.asarray(f) for f in images]) if isinstance(images, (list, tuple)) else np.asarray(images)
            bxh = np.concatenate([np.asarray(b, np.float32).reshape(-1, 4) for b in boxes_per_frame])
            idxh = np.concatenate([np.full(len(b), i, np.int32) for i, b in enumerate(boxes_per_frame)])
            staged = self.stage_block([frames, bxh, idxh] + list(extra or []))
            imgs, bx, idx, extra_dev = staged[0], staged[1], staged[2], staged[3:]
        assert imgs.dtype == torch.uint8 and imgs.dim() == 4 and imgs.shape[3] == 3 and imgs.shape[0] == len(boxes_per_frame)
        L = int(bx.shape[0])
        uv = torch.empty((L, NUM_KP, 2), dtype=torch.float32, device=dev)
        cov = torch.empty((L, NUM_KP, 2, 2), dtype=torch.float32, device=dev)
        kpm = torch.empty((L, NUM_KP), dtype=torch.float32, device=dev)
        kpl = torch.empty((L, NUM_KP), dtype=torch.float32, device=dev)
        logits = torch.empty((L, NUM_KP, HEAT, HEAT), dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().suo_net_forward_frames(self._h, _ptr(imgs), 0, int(imgs.shape[1]), int(imgs.shape[2]), _ptr(bx), _ptr(idx), L,
                                                     None, _ptr(uv), _ptr(cov), _ptr(kpm), _ptr(kpl), _ptr(logits), _stream()),
                   "suo_net_forward_frames")
        if check and self.pipe() == 2:
            torch.cuda.current_stream().synchronize()
            if self.range_exceeded():
                return self.forward_frames(images, boxes_per_frame, check=False, extra=extra)
        return {"uv": uv, "cov": cov, "prob_logits": logits, "kp_mask_logits": kpl, "kp_mask": kpm, "extra": extra_dev}


def decode_extras(logits):
    """The decode kernel (suo_decode_heatmaps) on logits [L,41,64,64] given on the device -- "uv", "cov", "mean_logit" as the forward
    pass computes them -- with its two optional outputs: "prob", the
    soft-max the reference returns as ret["prob"] (pkpnet.py:111), and "argmax", the diagnostic hard arg-max index per
    heat-map (int32 [L,41], flat h*64+w, first maximum -- torch.argmax's convention)."""
    L = int(logits.shape[0])
    dev = logits.device
    logits = logits.contiguous()
    prob = torch.empty_like(logits)
    idx = torch.empty((L, NUM_KP), dtype=torch.int32, device=dev)
    uv = torch.empty((L, NUM_KP, 2), dtype=torch.float32, device=dev)
    cov = torch.empty((L, NUM_KP, 2, 2), dtype=torch.float32, device=dev)
    ml = torch.empty((L, NUM_KP), dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().suo_decode_heatmaps(_ptr(logits), L, _ptr(uv), _ptr(cov), _ptr(ml), _ptr(idx), _ptr(prob), _stream()),
               "suo_decode_heatmaps")
    return {"prob": prob, "argmax": idx, "uv": uv, "cov": cov, "mean_logit": ml}


def render_priors(prior_uv, prior_mask, device="cuda"):
    """make_prior_kp_input (lib/utils/utils.py:398-411) for L objects at once, on the device: prior_uv [L,41,2] NDC,
    prior_mask [L,41] -> float32 [L,41,256,256] (device tensor)."""
    puv = torch.as_tensor(np.asarray(prior_uv, dtype=np.float32)).reshape(-1, NUM_KP, 2).to(device).contiguous()
    L = int(puv.shape[0])
    pmk = torch.as_tensor(np.asarray(prior_mask, dtype=np.uint8)).reshape(L, NUM_KP).to(device).contiguous()
    out = torch.empty((L, NUM_KP, 256, 256), dtype=torch.float32, device=device)
    _lib.check(_lib.lib().suo_render_priors(_ptr(puv), _ptr(pmk), L, _ptr(out), _stream()), "suo_render_priors")
    return out


def keypoint_masks(uv, cov, kp_mask, model_kps_masks, bbox_thresh=0.9, kp_var_thresh=0.2):
    """Device version of the mask logic at lib/object_slam.py:1100-1115 -> uint8 [L,41]."""
    L = int(uv.shape[0])
    mm = None
    if isinstance(model_kps_masks, torch.Tensor) and model_kps_masks.is_cuda:
        mm = model_kps_masks.contiguous()                  # already on the device (uint8 [L,41]: PkpNet.stage_block)
    elif model_kps_masks is not None:
        mm = torch.as_tensor(np.asarray(model_kps_masks, dtype=np.uint8)).to(uv.device).contiguous()
    out = torch.empty((L, NUM_KP), dtype=torch.uint8, device=uv.device)
    _lib.check(_lib.lib().suo_keypoint_masks(_ptr(uv), _ptr(cov), _ptr(kp_mask), _ptr(mm), L, float(bbox_thresh),
                                             float(kp_var_thresh), _ptr(out), _stream()), "suo_keypoint_masks")
    return out
