#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the reference's own Python modules.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden.py

Imports, unmodified, /root/reference/lib/models/{hg.py,pkpnet.py,layers/Residual.py}
(with import-time shims for the absent ``torchvision`` and for ``np.int`` that
lib/labeling/kp_config.py:99 needs), loads the build's seeded state_dict
(suo_slam_amd/weights.py, strict=True => key/shape compatibility is itself
checked) and stores inputs' seeds + the reference's outputs as .npz fixtures.
The fixtures are data only; no reference source is copied.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

# ---- import shims ------------------------------------------------------------------
if not hasattr(np, "int"):
    np.int = int
if not hasattr(np, "bool"):
    np.bool = bool
tv = types.ModuleType("torchvision")
tv.ops = types.ModuleType("torchvision.ops")
sys.modules["torchvision"] = tv
sys.modules["torchvision.ops"] = tv.ops
import matplotlib  # noqa: E402
matplotlib.use("Agg")
sys.path.insert(0, REF)
from lib.models import pkpnet as ref_pkpnet  # noqa: E402
from lib.models.hg import Hourglass as RefHourglass  # noqa: E402
from lib.models.layers.Residual import Residual as RefResidual  # noqa: E402

from suo_slam_amd import weights as W  # noqa: E402

torch.set_num_threads(8)
torch.manual_seed(0)


def sub_state(sd, prefix):
    return {k[len(prefix) + 1:]: torch.from_numpy(v) for k, v in sd.items() if k.startswith(prefix + ".")}


def main():
    sd = W.make_random_state_dict(seed=0, logit_gain=8.0)
    out = {}

    # ---- whole PkpNet: strict load proves key compatibility (1274 entries incl. num_batches_tracked)
    net = ref_pkpnet.PkpNet(calc_cov=True)
    ref_sd = net.state_dict()
    float_keys = [k for k in ref_sd if not k.endswith("num_batches_tracked")]
    assert sorted(float_keys) == sorted(sd.keys()), "state_dict key mismatch"
    assert [k for k in ref_sd if not k.endswith("num_batches_tracked")] == list(sd.keys()), "key order mismatch"
    full = {k: torch.from_numpy(v) for k, v in sd.items()}
    for k in ref_sd:
        if k.endswith("num_batches_tracked"):
            full[k] = ref_sd[k]
    net.load_state_dict(full, strict=True)
    net.eval()
    out["n_state_entries"] = np.int64(len(ref_sd))
    out["n_params"] = np.int64(sum(p.numel() for p in net.parameters()))

    with torch.no_grad():
        # ---- Residual blocks on small inputs
        rng = np.random.Generator(np.random.PCG64(101))
        for name, cin, cout, hw in (("backbone.r1", 64, 128, 12), ("backbone.r4", 128, 128, 8),
                                    ("backbone.hourglass.0.up1_.0", 256, 256, 8)):
            m = RefResidual(cin, cout)
            m.load_state_dict(sub_state(sd, name), strict=False)
            m.eval()
            x = rng.standard_normal((2, cin, hw, hw)).astype(np.float32)
            out[f"res_in:{name}"] = x
            out[f"res_out:{name}"] = m(torch.from_numpy(x)).numpy()

        # ---- one depth-4 hourglass on a 16x16 map
        hg = RefHourglass(4, 2, 256)
        hg.load_state_dict(sub_state(sd, "backbone.hourglass.1"), strict=False)
        hg.eval()
        x = rng.standard_normal((1, 256, 16, 16)).astype(np.float32)
        out["hg_in"] = x
        out["hg_out"] = hg(torch.from_numpy(x)).numpy()

        # ---- full backbone on one 44x256x256 crop (input regenerated from the seed in tests)
        rng = np.random.Generator(np.random.PCG64(202))
        x = rng.uniform(0, 1, (1, 44, 256, 256)).astype(np.float32)
        raw = net.backbone(torch.from_numpy(x))
        out["backbone_in_seed"] = np.int64(202)
        out["backbone_logits"] = raw.numpy()

        # ---- decode on the backbone logits and on synthetic peaked heat-maps
        prob = ref_pkpnet.spatial_softmax(raw)
        r = ref_pkpnet.post_process_kp(prob, z=None, calc_sigma=True)
        out["backbone_uv"] = r["uv"].numpy()
        out["backbone_cov"] = r["cov"].numpy()
        logit = net.classifier(raw.mean(3).mean(2))
        out["backbone_kp_mask_logits"] = logit.numpy()
        out["backbone_kp_mask"] = torch.sigmoid(logit).numpy()
        # diagnostic hard arg-max per heat-map (the reference decodes with the soft arg-max only): torch.argmax of
        # the reference's own logits over the flattened map, and how far the runner-up is below the maximum
        flat = raw.flatten(2)
        out["backbone_argmax"] = torch.argmax(flat, -1).numpy().astype(np.int32)
        top2 = torch.topk(flat, 2, -1).values
        out["backbone_top2_gap"] = (top2[..., 0] - top2[..., 1]).numpy()
        out["backbone_prob_sample"] = prob[:, ::5, ::4, ::4].numpy()            # spatial_softmax of the reference (ret["prob"], pkpnet.py:111)

        rng = np.random.Generator(np.random.PCG64(303))
        L = 3
        cy = rng.uniform(4, 60, (L, 41, 1, 1))
        cx = rng.uniform(4, 60, (L, 41, 1, 1))
        s = rng.uniform(1.0, 6.0, (L, 41, 1, 1))
        amp = rng.uniform(2.0, 20.0, (L, 41, 1, 1))
        ii = np.arange(64)[None, None, :, None]
        jj = np.arange(64)[None, None, None, :]
        heat = amp * np.exp(-((ii - cy) ** 2 + (jj - cx) ** 2) / (2 * s * s)) + rng.normal(0, 0.3, (L, 41, 64, 64))
        heat = heat.astype(np.float32)
        out["decode_in"] = heat
        ht = torch.from_numpy(heat)
        prob = ref_pkpnet.spatial_softmax(ht)
        r = ref_pkpnet.post_process_kp(prob, z=None, calc_sigma=True)
        out["decode_uv"] = r["uv"].numpy()
        out["decode_cov"] = r["cov"].numpy()
        logit = net.classifier(ht.mean(3).mean(2))
        out["decode_kp_mask_logits"] = logit.numpy()
        out["decode_kp_mask"] = torch.sigmoid(logit).numpy()
        out["decode_argmax"] = torch.argmax(ht.flatten(2), -1).numpy().astype(np.int32)
        out["decode_prob_sample"] = prob[:, ::5, ::4, ::4].numpy()
        # heat-maps FULL of ties (values quantised to 8 levels): torch.argmax's first-maximum convention
        rng = np.random.Generator(np.random.PCG64(404))
        tie = np.floor(rng.uniform(0, 8, (2, 41, 64, 64))).astype(np.float32)
        out["tie_seed"] = np.int64(404)
        out["tie_argmax"] = torch.argmax(torch.from_numpy(tie).flatten(2), -1).numpy().astype(np.int32)
        xx, yy = ref_pkpnet.mesh_grid(64, 64)
        out["mesh_xx"] = xx.numpy()
        out["mesh_yy"] = yy.numpy()

    path = os.path.join(HERE, "cnn_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
