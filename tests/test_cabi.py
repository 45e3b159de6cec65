"""The C-ABI library loads on a CPU-only box and exports every symbol include/suo_hip.h declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    from suo_slam_amd import build
    return build.build(verbose=False)


def test_header_symbols_exported(built):
    from suo_slam_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "suo_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(suo_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 15
    lib = _lib.lib()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/suo_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    assert lib.suo_version() >= 100


def test_no_cpu_fallback(built):
    """Without a GPU the product path must refuse to run rather than fall back."""
    import torch
    from suo_slam_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.SuoError):
        _lib.require_gpu()
    from suo_slam_amd.pkpnet import PkpNet
    from suo_slam_amd import weights
    with pytest.raises(_lib.SuoError):
        PkpNet(state_dict={k: v for k, v in list(weights.make_random_state_dict(0).items())[:2]})


def test_host_packers_roundtrip(built):
    import numpy as np
    from tests import hipops
    rng = np.random.default_rng(0)
    w = rng.standard_normal((70, 40)).astype(np.float32)
    p = hipops.pack_gemm(w, 96, 64)[:96 * 64].reshape(64 // 8, 96 // 32, 64, 4)      # first half: 32x32x2 form
    for (n, k) in ((0, 0), (69, 39), (33, 17), (5, 36)):
        assert p[k // 8, n // 32, (n % 32) + 32 * ((k % 8) // 4), k % 4] == w[n, k]
    assert p[7, 2, 40, 0] == 0  # padding
    wc = rng.standard_normal((64, 44, 7, 7)).astype(np.float32)
    pc = hipops.pack_conv(wc, 64, 48, 16)[:64 * 48 * 49].reshape(-1, 2, 64, 4)
    n, c, ky, kx = 37, 20, 3, 5
    kprime = ((c // 16) * 49 + ky * 7 + kx) * 16 + c % 16
    assert pc[kprime // 8, n // 32, (n % 32) + 32 * ((kprime % 8) // 4), kprime % 4] == wc[n, c, ky, kx]
