"""N2: the reference's checkpoint format drops in.  ``torch.save({'model': PkpNet.state_dict(), 'epoch', 'args'})``
(train.py:349-355) -> ``ObjectSLAM(chkpt_path, mesh_db)`` (object_slam.py:88-108) -> same network as building it
from the state_dict directly.  Key / shape compatibility with the reference's own module is proven separately by the
strict load in tests/golden/make_golden.py."""
import numpy as np
import pytest
import torch

from suo_slam_amd import synthetic as S
from suo_slam_amd import weights
from suo_slam_amd.object_slam import ObjectSLAM
from suo_slam_amd.pkpnet import PkpNet

pytestmark = pytest.mark.gpu


def _reference_style_checkpoint(path, sd, epoch=17):
    """float tensors under the reference's key names + the integer ``num_batches_tracked`` buffers every BatchNorm adds
    (1274 entries in total)."""
    model = {}
    for k, v in sd.items():
        model[k] = torch.from_numpy(np.array(v))
        if k.endswith("running_var"):
            model[k[:-len("running_var")] + "num_batches_tracked"] = torch.tensor(1234, dtype=torch.long)
    assert len(model) == 1274
    torch.save({"model": model, "epoch": epoch, "args": {"lr": 1e-3}}, path)
    return model


def test_checkpoint_file_round_trip(tmp_path):
    sd = weights.make_random_state_dict(seed=3, logit_gain=8.0)
    path = str(tmp_path / "pkpnet_ycbv.pt")
    _reference_style_checkpoint(path, sd)
    mesh_db = {i: {"is_symmetric": False, "diameter": 150.0} for i in range(1, 22)}
    slam = ObjectSLAM(path, mesh_db, single_view_mode=True)
    assert slam.model_epoch == 17
    direct = PkpNet(calc_cov=True, state_dict=sd, max_crops=16)
    rng = np.random.default_rng(0)
    fr = S.make_frame(rng, n_obj=3)
    boxes = [torch.from_numpy(fr["boxes"].astype(np.float32))]
    a = slam.model(fr["image"], boxes)
    b = direct(fr["image"], boxes)
    for k in ("uv", "cov", "kp_mask"):
        assert torch.equal(a[k].cpu(), b[k].cpu()), k
    direct.close()
    slam.model.close()


def test_checkpoint_with_missing_or_misshapen_tensor_fails_loudly(tmp_path):
    sd = weights.make_random_state_dict(seed=3)
    bad = dict(sd)
    bad.pop(next(k for k in bad if k.endswith("conv1.weight")))
    with pytest.raises(Exception) as e:
        PkpNet(calc_cov=True, state_dict=bad)
    assert "conv1.weight" in str(e.value)
    bad = dict(sd)
    k = next(k for k in bad if k.endswith("conv2.weight"))
    bad[k] = bad[k][:, :-1]
    with pytest.raises(Exception):
        PkpNet(calc_cov=True, state_dict=bad)
