#!/usr/bin/env python3
"""Round-6 widening of the reference-generated CNN goldens.  Run in the build container only (needs /root/reference):
    python tests/golden/make_golden_wide.py
Imports the reference's lib/models/{hg.py,pkpnet.py,layers/Residual.py} exactly as tests/golden/make_golden.py does and records, into cnn_golden_wide.npz:
  * the reference backbone's logits (+ its decode: uv, cov, validity, hard arg-max and top-2 gap) on FIVE crops with different statistics
    (tests/golden/cnn_inputs.py: texture with zero priors, texture with stamped priors, heavy-tailed, dark under dense priors, saturated blocks);
  * the reference's Residual module on network-sized maps -- 256 -> 256 at 64x64 and 32x32, 128 -> 128 (r4) at 64x64 and 32x32, 128 -> 256 with conv4 (r5) at
    64x64 -- so that the Winograd / fused-tail / split-operand kernels are held to the reference PER BLOCK (rounds 1-5 held them to fp64 restatements in the tests).
Inputs are regenerated from seeds by the tests; outputs of the blocks are stored on the rows cnn_inputs.BLOCK_ROWS (all channels, all columns).  Data only."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
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
from lib.models.layers.Residual import Residual as RefResidual  # noqa: E402

from suo_slam_amd import weights as W  # noqa: E402
from tests.golden import cnn_inputs as I  # noqa: E402

torch.set_num_threads(8)


def sub_state(sd, prefix):
    return {k[len(prefix) + 1:]: torch.from_numpy(v) for k, v in sd.items() if k.startswith(prefix + ".")}


def main():
    sd = W.make_random_state_dict(seed=0, logit_gain=8.0)
    net = ref_pkpnet.PkpNet(calc_cov=True)
    ref_sd = net.state_dict()
    full = {k: torch.from_numpy(v) for k, v in sd.items()}
    for k in ref_sd:
        if k.endswith("num_batches_tracked"):
            full[k] = ref_sd[k]
    net.load_state_dict(full, strict=True)
    net.eval()
    out = {"kinds": np.array(I.CROP_KINDS)}
    with torch.no_grad():
        x = torch.from_numpy(np.stack([I.crop(k) for k in I.CROP_KINDS]))
        raw = net.backbone(x)
        out["logits"] = raw.numpy()
        prob = ref_pkpnet.spatial_softmax(raw)
        r = ref_pkpnet.post_process_kp(prob, z=None, calc_sigma=True)
        out["uv"], out["cov"] = r["uv"].numpy(), r["cov"].numpy()
        logit = net.classifier(raw.mean(3).mean(2))
        out["kp_mask_logits"], out["kp_mask"] = logit.numpy(), torch.sigmoid(logit).numpy()
        flat = raw.flatten(2)
        out["argmax"] = torch.argmax(flat, -1).numpy().astype(np.int32)
        top2 = torch.topk(flat, 2, -1).values
        out["top2_gap"] = (top2[..., 0] - top2[..., 1]).numpy()
        for i, (name, cin, cout, hw) in enumerate(I.BLOCKS):
            m = RefResidual(cin, cout)
            missing = m.load_state_dict(sub_state(sd, name), strict=False)
            assert not [k for k in missing.missing_keys if not k.endswith("num_batches_tracked")], missing
            m.eval()
            y = m(torch.from_numpy(I.block_input(i))).numpy()
            out["block%d_rows" % i] = y[:, :, I.block_rows(hw), :]
            out["block%d_absmax" % i] = np.float32(np.abs(y).max())
    path = os.path.join(HERE, "cnn_golden_wide.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")
    for k in ("logits",):
        print(k, out[k].shape, [float(np.abs(out[k][i]).max()) for i in range(len(I.CROP_KINDS))])


if __name__ == "__main__":
    main()
