"""Per-layer max |x| of every convolution INPUT of the keypoint network -- the operands the two-term fp16 form (suo_slam_amd/csrc/f16x2.h) has to hold:
activations enter that form times 16, so a 1x1 convolution needs max |x| < 4094 and a 3x3 one (Winograd: |B^T d B| <= 4 max |d|, the guard's bound) < 1023.
Runs the CPU oracle (tools/ may use it: this is a measurement, not the product) on one synthetic frame for
  (a) the seeded test weights (weights.make_random_state_dict: U(+-1/sqrt(fan_in)) convolutions, randomised BatchNorm statistics), and
  (b) the same with every convolution weight redrawn N(0, 1) / sqrt(fan_in) * gain (gain sqrt(2) = He scaling: the residual stream grows with depth).
python tools/measure_activation_range.py > profiles/r05_activation_range.txt"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cnn_oracle as O                      # noqa: E402
from suo_slam_amd import synthetic as S                 # noqa: E402
from suo_slam_amd import weights as W                   # noqa: E402


def normal_state_dict(seed, gain):
    sd = W.make_random_state_dict(seed=seed, logit_gain=1.0)
    rng = np.random.default_rng(seed + 1000)
    for k, v in sd.items():
        if k.endswith(".weight") and v.ndim == 4:
            fan_in = int(np.prod(v.shape[1:]))
            sd[k] = (rng.standard_normal(v.shape) * gain / np.sqrt(fan_in)).astype(np.float32)
    return sd


def measure(sd, crops):
    P = O.to_torch(sd)
    rec = []
    orig = O._conv

    def spy(x, P_, p, stride=1, padding=0):
        rec.append((p, int(P_[p + ".weight"].shape[-1]), float(x.abs().max())))
        return orig(x, P_, p, stride, padding)

    O._conv = spy
    try:
        with torch.no_grad():
            O.hourglass_net(crops, P)
    finally:
        O._conv = orig
    return rec


def main():
    rng = np.random.default_rng(5)
    fr = S.make_frame(rng, 2, noise=0.0)
    chw = O.image_to_chw(fr["image"])
    crops = torch.from_numpy(O.roi_align(chw, np.asarray(fr["boxes"], np.float32)[:2]))
    x = torch.cat([crops, torch.zeros(crops.shape[0], 41, 256, 256)], 1)
    for name, sd in (("seeded test weights (make_random_state_dict(seed=0, logit_gain=8))", W.make_random_state_dict(seed=0, logit_gain=8.0)),
                     ("N(0,1)/sqrt(fan_in) convolutions", normal_state_dict(0, 1.0)),
                     ("N(0,1)*sqrt(2)/sqrt(fan_in) convolutions (He)", normal_state_dict(0, float(np.sqrt(2.0))))):
        rec = measure(sd, x)
        m1 = max(r[2] for r in rec if r[1] == 1)
        m3 = max(r[2] for r in rec if r[1] == 3)
        print(f"# {name}: {len(rec)} convolutions; largest input magnitude of a 1x1 convolution {m1:.3f} (limit 4094), of a 3x3 {m3:.3f} (limit 1023), of the 7x7 stem {rec[0][2]:.3f} (not on the fp16 form)")
        worst = sorted(rec, key=lambda r: -r[2] * (4 if r[1] == 3 else 1))[:8]
        for p, ks, m in worst:
            print(f"    {p:48s} {ks}x{ks}   max |x| {m:10.3f}   headroom {((1023 if ks == 3 else 4094) / m):8.1f}x")
    return 0


if __name__ == "__main__":
    sys.exit(main())
