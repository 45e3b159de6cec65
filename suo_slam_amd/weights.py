"""Reference-checkpoint-compatible weight container for the keypoint network.

The reference stores ``checkpoint['model']`` = ``PkpNet.state_dict()``
(/root/reference/lib/object_slam.py:92-97, train.py:349-355).  The keys are
``backbone.*`` (HourglassNet, lib/models/hg.py:60-93) and ``classifier.2.*``
(lib/models/pkpnet.py:74-78).  No pretrained weights ship with the reference,
so the build generates seeded random weights *with the same keys and shapes*;
a real checkpoint's ``model`` dict drops in unchanged.

Everything here is numpy (PCG64) so the same seed gives bit-identical weights
on every machine, independent of the torch build.
"""
from __future__ import annotations

import numpy as np

NUM_KP = 41          # lib/labeling/kp_config.py:82-94
N_STACK = 2          # lib/models/hg.py:61
N_MODULES = 2
N_FEATS = 256
HG_DEPTH = 4         # lib/models/hg.py:77
BN_EPS = 1e-5        # torch.nn.BatchNorm2d default


def _conv_keys(prefix, cin, cout, k):
    return [(prefix + ".weight", (cout, cin, k, k)), (prefix + ".bias", (cout,))]


def _bn_keys(prefix, c):
    return [(prefix + ".weight", (c,)), (prefix + ".bias", (c,)),
            (prefix + ".running_mean", (c,)), (prefix + ".running_var", (c,))]


def _residual_keys(prefix, cin, cout):
    # registration order of lib/models/layers/Residual.py:4-18
    keys = []
    keys += _bn_keys(prefix + ".bn", cin)
    keys += _conv_keys(prefix + ".conv1", cin, cout // 2, 1)
    keys += _bn_keys(prefix + ".bn1", cout // 2)
    keys += _conv_keys(prefix + ".conv2", cout // 2, cout // 2, 3)
    keys += _bn_keys(prefix + ".bn2", cout // 2)
    keys += _conv_keys(prefix + ".conv3", cout // 2, cout, 1)
    if cin != cout:
        keys += _conv_keys(prefix + ".conv4", cin, cout, 1)
    return keys


def _hourglass_keys(prefix, n):
    # registration order of lib/models/hg.py:7-31: low2 / low2_ first, then up1_, low1_, low3_
    keys = []
    if n > 1:
        keys += _hourglass_keys(prefix + ".low2", n - 1)
    else:
        for j in range(N_MODULES):
            keys += _residual_keys(f"{prefix}.low2_.{j}", N_FEATS, N_FEATS)
    for name in ("up1_", "low1_", "low3_"):
        for j in range(N_MODULES):
            keys += _residual_keys(f"{prefix}.{name}.{j}", N_FEATS, N_FEATS)
    return keys


def state_dict_spec():
    """Ordered list of (key, shape) for every float tensor of PkpNet.state_dict()."""
    b = "backbone"
    keys = []
    keys += _conv_keys(f"{b}.conv1_", 3 + NUM_KP, 64, 7)
    keys += _bn_keys(f"{b}.bn1", 64)
    keys += _residual_keys(f"{b}.r1", 64, 128)
    keys += _residual_keys(f"{b}.r4", 128, 128)
    keys += _residual_keys(f"{b}.r5", 128, N_FEATS)
    for i in range(N_STACK):
        keys += _hourglass_keys(f"{b}.hourglass.{i}", HG_DEPTH)
    for i in range(N_STACK * N_MODULES):
        keys += _residual_keys(f"{b}.Residual.{i}", N_FEATS, N_FEATS)
    for i in range(N_STACK):
        keys += _conv_keys(f"{b}.lin_.{i}.0", N_FEATS, N_FEATS, 1)
        keys += _bn_keys(f"{b}.lin_.{i}.1", N_FEATS)
    for i in range(N_STACK):
        keys += _conv_keys(f"{b}.tmpOut.{i}", N_FEATS, NUM_KP, 1)
    for i in range(N_STACK - 1):
        keys += _conv_keys(f"{b}.ll_.{i}", N_FEATS, N_FEATS, 1)
    for i in range(N_STACK - 1):
        keys += _conv_keys(f"{b}.tmpOut_.{i}", NUM_KP, N_FEATS, 1)
    keys += [("classifier.2.weight", (NUM_KP, NUM_KP)), ("classifier.2.bias", (NUM_KP,))]
    return keys


def make_random_state_dict(seed: int = 0, logit_gain: float = 1.0):
    """Seeded random weights with non-trivial BatchNorm running statistics.

    Convs: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (PyTorch's default bound); BN:
    gamma~U(.5,1.5), beta~N(0,.1), running_mean~N(0,.1), running_var~U(.5,1.5)
    (SURVEY.md §8d).  ``logit_gain`` scales the final head so that heat-maps are
    peaked enough for the decode thresholds to be exercised.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for key, shape in state_dict_spec():
        leaf = key.rsplit(".", 1)[1]
        parent = key.rsplit(".", 1)[0]
        is_bn = (parent + ".running_mean", shape) in _BN_INDEX
        if is_bn:
            if leaf == "weight":
                v = rng.uniform(0.5, 1.5, shape)
            elif leaf == "bias":
                v = rng.normal(0.0, 0.1, shape)
            elif leaf == "running_mean":
                v = rng.normal(0.0, 0.1, shape)
            else:
                v = rng.uniform(0.5, 1.5, shape)
        else:
            if leaf == "weight":
                fan_in = int(np.prod(shape[1:]))
                bound = 1.0 / np.sqrt(fan_in)
                v = rng.uniform(-bound, bound, shape)
            else:
                wshape = dict(_SPEC)[parent + ".weight"]
                fan_in = int(np.prod(wshape[1:]))
                bound = 1.0 / np.sqrt(fan_in)
                v = rng.uniform(-bound, bound, shape)
        sd[key] = np.ascontiguousarray(v, dtype=np.float32)
    last = f"backbone.tmpOut.{N_STACK - 1}"
    sd[last + ".weight"] *= np.float32(logit_gain)
    sd[last + ".bias"] *= np.float32(logit_gain)
    return sd


_SPEC = state_dict_spec()
_BN_INDEX = {(k, s) for k, s in _SPEC if k.endswith(".running_mean")}


def num_params(sd=None):
    spec = _SPEC if sd is None else [(k, v.shape) for k, v in sd.items()]
    return sum(int(np.prod(s)) for k, s in spec
               if not (k.endswith("running_mean") or k.endswith("running_var")))
