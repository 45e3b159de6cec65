"""Import shims that let the REFERENCE's lib/object_slam.py run unmodified in the build container (used only by
tests/golden/make_slam_golden.py; never on the GPU box, never by the product).

The file's module-level imports that this image lacks are cv2, g2o, lambdatwist, torchvision and the BOP renderer
(lib/object_slam.py:2,9-10,18).  Stand-ins installed in sys.modules:

  cv2           setNumThreads + GaussianBlur (OpenCV's documented kernel for an impulse; only the prior stamp uses it)
  lambdatwist   pnp(xs, ys, threshold) -> oracle.geometry.pnp (the C restatement, pinned to the reference's own p4p.cpp), with
                the product's seeding convention so a replay draws the same samples: the j-th call inside one __run_kp_model
                uses seed base + j * 0x9E3779B97F4A7C15 (base advances by the number of calls), any other call seed 0
  g2o           a RECORDING stub of the object API optimize() uses (SURVEY.md 8b lists it): every vertex, edge, level,
                robust-kernel change and optimize(n) call is logged; optimize(n) runs ONE initializeOptimization(0)+optimize(n)
                of the oracle's LM (oracle/lm_oracle.c: orc_lm_round) on the graph exactly as the Python built it, poses kept as
                SE3Quat (quaternion + t) between calls like g2o does; chi2() uses the error stored by the last
                compute_error()/optimize(), as g2o does (after optimize(): at the accepted state)

What this pins: the reference's PYTHON -- graph construction order, ids, fixed flags, information matrices, levels, kernel
removal round, the optimize(n) sequence, culling -- not g2o's C++ (unbuildable here: Eigen / CHOLMOD absent)."""
import ctypes as C
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from oracle import geometry as G  # noqa: E402

SEED_STRIDE = 0x9E3779B97F4A7C15
_dp = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_u8 = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def _orc():
    L = G.lib()
    if not getattr(L, "_slam_stub_ready", False):
        L.orc_lm_round.restype = C.c_int
        L.orc_lm_round.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _u8, _dp, _u8, _ip, _ip, _dp, _dp, _dp, _dp, _u8, _u8, C.c_double,
                                   C.c_int, _dp, C.POINTER(C.c_int)]
        L.orc_pose_from_T.restype = None
        L.orc_pose_from_T.argtypes = [_dp, _dp]
        L.orc_pose_to_T.restype = None
        L.orc_pose_to_T.argtypes = [_dp, _dp]
        L.orc_edge_error_qt.restype = None
        L.orc_edge_error_qt.argtypes = [_dp] * 6
        L._slam_stub_ready = True
    return L


# ---- lambdatwist ------------------------------------------------------------------------------------------------------
class PnpStub:
    def __init__(self):
        self.base = 0            # the product's ObjectSLAM._pnp_seed
        self.in_kp_model = False
        self.j = 0
        self.log = []

    def begin_kp_model(self):
        self.in_kp_model, self.j = True, 0

    def end_kp_model(self):
        self.base += self.j
        self.in_kp_model = False

    def pnp(self, xs, ys, threshold=0.001):
        seed = (self.base + self.j * SEED_STRIDE) % 2 ** 64 if self.in_kp_model else 0
        if self.in_kp_model:
            self.j += 1
        T = G.pnp(np.asarray(xs, np.float64), np.asarray(ys, np.float64), threshold, seed=seed)[0]
        self.log.append({"n": len(xs), "seed": seed, "in_kp_model": self.in_kp_model})
        return T


# ---- g2o ----------------------------------------------------------------------------------------------------------------
class SE3Quat:
    def __init__(self, R=None, t=None):
        self.qt = np.zeros(7)
        T = np.zeros(12)
        T.reshape(3, 4)[:, :3] = np.eye(3) if R is None else np.asarray(R, np.float64)
        T.reshape(3, 4)[:, 3] = 0 if t is None else np.asarray(t, np.float64)
        self.T_in = T.reshape(3, 4).copy()          # what the Python handed over (g2o itself keeps only the quaternion)
        _orc().orc_pose_from_T(T, self.qt)

    def matrix(self):
        T = np.zeros(12)
        _orc().orc_pose_to_T(self.qt, T)
        M = np.eye(4)
        M[:3] = T.reshape(3, 4)
        return M


class VertexSE3Expmap:
    def __init__(self):
        self._id, self._fixed, self._est = None, False, None

    def set_id(self, i):
        self._id = int(i)

    def set_estimate(self, pose):
        self._est = pose

    def set_fixed(self, f):
        self._fixed = bool(f)

    def estimate(self):
        return self._est


class RobustKernelHuber:
    def __init__(self, delta=1.0):
        self.delta = float(delta)


class _Solver:
    def __init__(self, inner=None):
        self.inner = inner


class LinearSolverDenseSE3(_Solver):
    pass


class LinearSolverCholmodSE3(_Solver):
    pass


class BlockSolverSE3(_Solver):
    pass


class OptimizationAlgorithmLevenberg(_Solver):
    pass


class _Edge:
    fixed_object = False

    def __init__(self, cam_k, p, T_OtoG=None):
        self.cam_k = np.array(cam_k, np.float64).reshape(4)
        self.p = np.array(p, np.float64).reshape(3)
        self.T_OtoG = None if T_OtoG is None else np.array(T_OtoG, np.float64)[:3, :4]
        self.v = {}
        self.uv = self.info = self.kernel = None
        self.level = 0
        self.err = np.zeros(2)

    def set_vertex(self, i, v):
        self.v[i] = v

    def set_measurement(self, uv):
        self.uv = np.array(uv, np.float64).reshape(2)

    def set_information(self, I):
        self.info = np.array(I, np.float64).reshape(2, 2)

    def set_robust_kernel(self, k):
        self.kernel = k

    def set_level(self, level):
        self.level = int(level)

    def _poses(self):
        if self.fixed_object:
            return self.v[0].estimate().qt, SE3Quat(self.T_OtoG[:, :3], self.T_OtoG[:, 3]).qt
        return self.v[1].estimate().qt, self.v[0].estimate().qt

    def compute_error(self):
        cam, obj = self._poses()
        _orc().orc_edge_error_qt(cam, obj, self.cam_k, self.p, self.uv, self.err)

    def chi2(self):
        if self.info is None:                         # types_object_slam.cpp:46-49 exits the process here
            raise SystemExit("edge information not set")
        e, I = self.err, self.info
        return float(e[0] * (I[0, 0] * e[0] + I[0, 1] * e[1]) + e[1] * (I[1, 0] * e[0] + I[1, 1] * e[1]))      # error^T Omega error


class EdgeSE3ProjectFromObject(_Edge):
    pass


class EdgeSE3ProjectFromFixedObject(_Edge):
    fixed_object = True


class SparseOptimizer:
    last = None                  # the optimizer of the most recent ObjectSLAM.optimize() call (for the recorder)

    def __init__(self):
        self.algorithm = None
        self.vertices, self._edges, self.calls = [], [], []
        SparseOptimizer.last = self

    def set_algorithm(self, a):
        self.algorithm = a

    def add_vertex(self, v):
        self.vertices.append(v)

    def add_edge(self, e):
        self._edges.append(e)

    def edges(self):
        return self._edges

    def set_verbose(self, f):
        pass

    def initialize_optimization(self, level=0):
        self.calls.append(("init", int(level)))

    def optimize(self, n):
        E = self._edges
        cams, objs = [], []                                     # camera / object vertex lists in first-use order
        fixed_objs = []
        e_cam, e_obj = [], []
        for e in E:
            cv = e.v[0] if e.fixed_object else e.v[1]
            if cv not in cams:
                cams.append(cv)
            e_cam.append(cams.index(cv))
            if e.fixed_object:
                fixed_objs.append(SE3Quat(e.T_OtoG[:, :3], e.T_OtoG[:, 3]).qt)
                e_obj.append(-len(fixed_objs))
            else:
                if e.v[0] not in objs:
                    objs.append(e.v[0])
                e_obj.append(objs.index(e.v[0]))
        n_free_obj = len(objs)
        e_obj = np.array([o if o >= 0 else n_free_obj + (-o - 1) for o in e_obj], np.int32)
        cam_qt = np.ascontiguousarray(np.stack([v.estimate().qt for v in cams]))
        obj_qt = np.ascontiguousarray(np.stack([v.estimate().qt for v in objs] + fixed_objs))
        cam_fixed = np.array([v._fixed for v in cams], np.uint8)
        obj_fixed = np.array([v._fixed for v in objs] + [1] * len(fixed_objs), np.uint8)
        level = np.array([e.level for e in E], np.uint8)
        robust = np.array([e.kernel is not None for e in E], np.uint8)
        deltas = {e.kernel.delta for e in E if e.kernel is not None}
        assert len(deltas) <= 1
        # (the oracle keeps Omega as (xx, xy, yy); a float32 inverse is symmetric only to rounding: its symmetric part is what
        #  error^T Omega error and J^T Omega J's symmetric part see)
        info = np.array([[e.info[0, 0], 0.5 * (e.info[0, 1] + e.info[1, 0]), e.info[1, 1]] for e in E])
        err = np.zeros((len(E), 2))
        trials = C.c_int(0)
        its = _orc().orc_lm_round(len(cams), len(obj_qt), len(E), cam_qt, cam_fixed, obj_qt, obj_fixed, np.array(e_cam, np.int32), e_obj,
                                  np.array([e.cam_k for e in E]), np.array([e.p for e in E]), np.array([e.uv for e in E]),
                                  np.ascontiguousarray(info), level, robust, deltas.pop() if deltas else 0.0, int(n), err, C.byref(trials))
        self.calls.append(("optimize", int(n), level.copy(), robust.copy(), int(its), int(trials.value)))
        for v, qt in zip(cams, cam_qt):
            if not v._fixed:
                v.estimate().qt[:] = qt
        for v, qt in zip(objs, obj_qt[:n_free_obj]):
            if not v._fixed:
                v.estimate().qt[:] = qt
        for k, e in enumerate(E):                 # active edges keep the error of the accepted state (computeActiveErrors)
            both_fixed = (e.v[0]._fixed if e.fixed_object else (e.v[0]._fixed and e.v[1]._fixed))
            if e.level == 0 and not both_fixed and its >= 0:
                e.err[:] = err[k]
        return its


def install(pnp_stub):
    """Install every stand-in; returns the imported reference module lib.object_slam."""
    if not hasattr(np, "int"):
        np.int = int
    if not hasattr(np, "math"):
        import math
        np.math = math
    tv = types.ModuleType("torchvision")
    tv.ops = types.ModuleType("torchvision.ops")
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.ops"] = tv.ops

    def gaussian_blur(img, ksize, sigma):
        k = ksize[0]
        s = 0.3 * ((k - 1) * 0.5 - 1) + 0.8 if sigma <= 0 else sigma
        i = np.arange(k, dtype=np.float64) - (k - 1) / 2
        g = np.exp(-(i * i) / (2 * s * s))
        g /= g.sum()
        return np.outer(g, g).astype(np.float32)
    cv2 = types.ModuleType("cv2")
    cv2.setNumThreads = lambda n: None
    cv2.GaussianBlur = gaussian_blur
    sys.modules["cv2"] = cv2
    lt = types.ModuleType("lambdatwist")
    lt.pnp = pnp_stub.pnp
    sys.modules["lambdatwist"] = lt
    g2o = types.ModuleType("g2o")
    for cls in (SE3Quat, VertexSE3Expmap, RobustKernelHuber, LinearSolverDenseSE3, LinearSolverCholmodSE3, BlockSolverSE3,
                OptimizationAlgorithmLevenberg, EdgeSE3ProjectFromObject, EdgeSE3ProjectFromFixedObject, SparseOptimizer):
        setattr(g2o, cls.__name__, cls)
    sys.modules["g2o"] = g2o
    r = types.ModuleType("thirdparty.bop_toolkit.bop_toolkit_lib.renderer_py")
    r.RendererPython = object
    sys.modules["thirdparty.bop_toolkit.bop_toolkit_lib.renderer_py"] = r
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, "/root/reference")
    import lib.object_slam as ref_slam
    return ref_slam
