"""RCCL where it can run: a ONE-rank "nccl" process group on the MI355X (VERDICT r3 #4a).  The multi-GPU global bundle adjustment
(suo_slam_amd/ba_dist.py) short-circuits its collectives at world size 1, and every N > 1 test is gloo on CPU -- so this
is the only place where the in-place dist.all_reduce on ph.lin / ph.sch / the ph.red[:3] view, the RCCL communicator bring-up and the
stream ordering between the phase kernels (csrc/lm_dist.hip) and the collectives execute before an 8-GPU node does.  Runs in a child
process (a process group outlives nothing else in the pytest session, and a hung bring-up cannot hang it)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n_cam,n_obj", [(12, 6), (32, 16)])
def test_one_rank_rccl_group_runs_every_collective_in_place_and_changes_nothing(n_cam, n_obj):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    port = 29500 + (os.getpid() % 400) + n_cam
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_one_rank.py"), str(n_cam), str(n_obj), str(port)], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RCCL_ONE_RANK ")][-1]
    out = json.loads(line[len("RCCL_ONE_RANK "):])
    assert out["backend"] == "nccl"
    # per LM trial two collectives, per LM iteration one, per classification one, + the result assembly
    assert out["all_reduce_calls"] >= 2 * out["trials"] + out["iterations"] + 2
    assert out["on_device"] == out["all_reduce_calls"]                      # every one on a device tensor: nothing staged through the host
    assert out["identical"], out                                            # a one-rank SUM is the identity: bit-identical to the short-circuit
