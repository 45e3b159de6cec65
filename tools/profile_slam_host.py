"""cProfile of the SLAM-mode geometry + host logic (the first half of tools/bench_slam.py): where does a view's time go?
python tools/profile_slam_host.py [n_views] [n_objs]"""
import cProfile
import os
import pstats
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from suo_slam_amd import bop  # noqa: E402
from suo_slam_amd.object_slam import ObjectSLAM  # noqa: E402
from tests import bop_tree  # noqa: E402

n_views = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n_objs = int(sys.argv[2]) if len(sys.argv) > 2 else 8

with tempfile.TemporaryDirectory() as root:
    desc = bop_tree.build_sequence(root, seed=3, n_views=n_views, n_objs=n_objs)
    ds = bop.BopDataset(desc["data_root"], desc["split"], bop_dset="ycbv", ignore_symmetry=True)
    mesh_db = bop.load_mesh_db(os.path.join(desc["data_root"], "models_bop-compat_eval"))
    scene = ds.scene_ids()[0]
    samples = [(v, ds.get_all_obj(scene, v), ds.obj_ids(scene, v)) for v in ds.view_ids(scene)]
    args = []
    for v, s, ids in samples:
        img = (255 * s["img"].numpy().transpose(1, 2, 0)).astype(np.uint8)
        args.append((v, img, s["K"].numpy(), np.array(ids), s["bboxes"].numpy(), s["model_kps"].numpy(), s["kp_model_masks"].numpy(),
                     s["kp_masks"].numpy(), s["kp_uvs"].numpy()))

    def run():
        slam = ObjectSLAM(None, mesh_db, debug_gt_kp=True, manual_kp_std=0.01)
        for a in args:
            slam.process_view(*a[:8], uv_gt=a[8])
        slam.collect_results(final=True)

    run()       # warm-up (library load, first launches)
    pr = cProfile.Profile()
    pr.enable()
    run()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(28)
    st.sort_stats("tottime").print_stats(18)
