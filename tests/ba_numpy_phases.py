"""Test-only numpy implementation of the BA phase interface (suo_slam_amd/ba_dist.py: HipPhases) so the
multi-rank host schedule can be exercised on CPU with gloo.  Same algebra as csrc/lm_dist.hip, written
independently with dense numpy: poses as 4x4, SE3 exp-map left update, Huber-weighted normal equations."""
import numpy as np


def torch_from(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, np.float64))


def _skew(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])


def _exp(u):
    w, v = u[:3], u[3:]
    th = np.linalg.norm(w)
    Om = _skew(w)
    if th < 1e-5:
        R = np.eye(3) + Om + Om @ Om
        V = R
    else:
        R = np.eye(3) + np.sin(th) / th * Om + (1 - np.cos(th)) / th ** 2 * Om @ Om
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * Om + (th - np.sin(th)) / th ** 3 * Om @ Om
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = V @ v
    return T


def _huber(e2, delta):
    d2 = delta * delta
    if e2 <= d2:
        return e2, 1.0
    s = np.sqrt(e2)
    return 2 * s * delta - d2, delta / s


class NumpyPhases:
    """Buffer interface of ba_dist.HipPhases (lin / sch / red / good tensors, here on the CPU) over the array-returning
    numpy phases below."""

    def __init__(self, local, world=1):
        import torch
        self.P = local
        O = len(local.obj_T)
        nfo = sum(1 for o in range(O) if not local.obj_fixed[o])
        self.lin = torch.zeros(1 + 27 * O + world, dtype=torch.float64)
        self.sch = torch.zeros(36 * nfo * nfo + 6 * nfo + 1, dtype=torch.float64)
        self.red = torch.zeros(4, dtype=torch.float64)
        self.good = torch.zeros(1, dtype=torch.float64)
        self._init(local)

    def read(self, t):
        return t.numpy().copy()

    def classify(self, keep_all):
        self.good[0] = self._classify(keep_all)

    def linearize(self, robust_on, rank, world):
        out = self._linearize(robust_on)
        n = len(out) - 1
        self.lin.zero_()
        self.lin[:n] = torch_from(out[:n])
        self.lin[n + rank] = float(out[-1])

    def schur(self, lam):
        self.sch[:] = torch_from(self._schur(lam))

    def solve_update(self, lam, robust_on, world):
        sch, lin = self.sch.numpy(), self.lin.numpy()
        O = self.n_obj
        if int(round(sch[-1])) != world:
            self.red[:] = torch_from(np.array([self._chi(robust_on), 0.0, 0.0, 0.0]))
            return
        out = self._solve_update(lam, robust_on, np.concatenate([lin[1:1 + 27 * O], sch[:-1]]))
        self.red[:] = torch_from(np.array([out[0], out[1], out[3], out[2]]))

    def _init(self, local):
        self.cam = [np.vstack([t.reshape(3, 4), [0, 0, 0, 1]]) for t in local.cam_T]
        self.obj = [np.vstack([t.reshape(3, 4), [0, 0, 0, 1]]) for t in local.obj_T]
        self.E = len(local.edge_cam)
        self.level = np.zeros(self.E, np.uint8)
        self.n_obj = len(self.obj)
        self.slot = {}
        for o in range(self.n_obj):
            if not local.obj_fixed[o]:
                self.slot[o] = len(self.slot)
        self.ns = 6 * len(self.slot)

    def _err(self, e):
        P = self.P
        c, o = int(P.edge_cam[e]), int(P.edge_obj[e])
        pw = self.obj[o][:3, :3] @ P.edge_p[e] + self.obj[o][:3, 3]
        pc = self.cam[c][:3, :3] @ pw + self.cam[c][:3, 3]
        k = P.edge_camk[e]
        err = P.edge_uv[e] - np.array([k[0] * pc[0] / pc[2] + k[2], k[1] * pc[1] / pc[2] + k[3]])
        return err, pw, pc

    def _info(self, e):
        i = self.P.edge_info[e]
        return np.array([[i[0], i[1]], [i[1], i[2]]])

    def _active(self, e):
        P = self.P
        return self.level[e] == 0 and not (P.cam_fixed[int(P.edge_cam[e])] and P.obj_fixed[int(P.edge_obj[e])])

    def _classify(self, keep_all):
        good = 0
        for e in range(self.E):
            err, _, _ = self._err(e)
            c2 = float(err @ self._info(e) @ err)
            self.P.chi2[e] = c2
            if keep_all:
                good += 1
            elif c2 > self.P.chi2_thr:
                self.level[e] = 1
                self.P.inlier[e] = 0
            else:
                self.level[e] = 0
                self.P.inlier[e] = 1
                good += 1
        return float(good)

    def _chi(self, robust_on):
        chi = 0.0
        for e in range(self.E):
            if self._active(e):
                err, _, _ = self._err(e)
                c2 = float(err @ self._info(e) @ err)
                chi += _huber(c2, self.P.huber_delta)[0] if robust_on else c2
        return chi

    def _linearize(self, robust_on):
        P = self.P
        C, O = len(self.cam), self.n_obj
        self.Hcc = np.zeros((C, 6, 6)); self.bc = np.zeros((C, 6))
        Hoo = np.zeros((O, 6, 6)); bo = np.zeros((O, 6))
        self.Hco = {}
        chi = 0.0
        for e in range(self.E):
            if not self._active(e):
                continue
            c, o = int(P.edge_cam[e]), int(P.edge_obj[e])
            err, pw, pc = self._err(e)
            Om = self._info(e)
            c2 = float(err @ Om @ err)
            w = 1.0
            if robust_on:
                r, w = _huber(c2, P.huber_delta)
                chi += r
            else:
                chi += c2
            k = P.edge_camk[e]
            PJ = -np.array([[k[0] / pc[2], 0, -k[0] * pc[0] / pc[2] ** 2], [0, k[1] / pc[2], -k[1] * pc[1] / pc[2] ** 2]])
            Jc = PJ @ np.hstack([-_skew(pc), np.eye(3)])
            Jo = PJ @ self.cam[c][:3, :3] @ np.hstack([-_skew(pw), np.eye(3)])
            W = w * Om
            g = -W @ err
            if not P.cam_fixed[c]:
                self.Hcc[c] += Jc.T @ W @ Jc
                self.bc[c] += Jc.T @ g
            if not P.obj_fixed[o]:
                Hoo[o] += Jo.T @ W @ Jo
                bo[o] += Jo.T @ g
            if not P.cam_fixed[c] and not P.obj_fixed[o]:
                self.Hco[(c, o)] = self.Hco.get((c, o), np.zeros((6, 6))) + Jc.T @ W @ Jo
        out = np.zeros(2 + 27 * O)
        out[0] = chi
        iu = np.triu_indices(6)
        for o in range(O):
            out[1 + 27 * o:1 + 27 * o + 21] = Hoo[o][iu]
            out[1 + 27 * o + 21:1 + 27 * o + 27] = bo[o]
        free = [c for c in range(C) if not P.cam_fixed[c]]
        out[-1] = max([np.abs(np.diag(self.Hcc[c])).max() for c in free], default=0.0)
        return out

    def _schur(self, lam):
        P = self.P
        ns = self.ns
        self.bak = ([T.copy() for T in self.cam], [T.copy() for T in self.obj])
        S = np.zeros((ns, ns)); r = np.zeros(ns)
        self.Hinv, self.yc = {}, {}
        ok = 1.0
        for c in range(len(self.cam)):
            if P.cam_fixed[c]:
                continue
            A = self.Hcc[c] + lam * np.eye(6)
            try:
                np.linalg.cholesky(A)
                Ai = np.linalg.inv(A)
            except np.linalg.LinAlgError:
                ok = 0.0
                Ai = np.zeros((6, 6))
            self.Hinv[c] = Ai
            self.yc[c] = Ai @ self.bc[c]
            objs = [o for (cc, o) in self.Hco if cc == c]
            for o1 in objs:
                s1 = self.slot[o1]
                r[6 * s1:6 * s1 + 6] += self.Hco[(c, o1)].T @ self.yc[c]
                for o2 in objs:
                    s2 = self.slot[o2]
                    S[6 * s1:6 * s1 + 6, 6 * s2:6 * s2 + 6] += self.Hco[(c, o1)].T @ Ai @ self.Hco[(c, o2)]
        return np.concatenate([S.ravel(), r, [ok]])

    def _solve_update(self, lam, robust_on, totals):
        P = self.P
        ns, O = self.ns, self.n_obj
        HB = totals[:27 * O]
        St = totals[27 * O:27 * O + ns * ns].reshape(ns, ns)
        rt = totals[27 * O + ns * ns:]
        S = -St.copy()
        rhs = -rt.copy()
        iu = np.triu_indices(6)
        bo = {}
        for o, s in self.slot.items():
            H = np.zeros((6, 6))
            H[iu] = HB[27 * o:27 * o + 21]
            H = H + H.T - np.diag(np.diag(H))
            S[6 * s:6 * s + 6, 6 * s:6 * s + 6] += H + lam * np.eye(6)
            bo[o] = HB[27 * o + 21:27 * o + 27]
            rhs[6 * s:6 * s + 6] += bo[o]
        try:
            Lc = np.linalg.cholesky(S)
            x = np.linalg.solve(Lc.T, np.linalg.solve(Lc, rhs))
        except np.linalg.LinAlgError:
            return np.array([self._chi(robust_on), 0.0, 0.0, 0.0])
        sc_c = sc_o = 0.0
        for o, s in self.slot.items():
            xo = x[6 * s:6 * s + 6]
            sc_o += float(xo @ (lam * xo + bo[o]))
        xos = {o: x[6 * s:6 * s + 6] for o, s in self.slot.items()}
        for c in range(len(self.cam)):
            if P.cam_fixed[c]:
                continue
            xc = self.yc[c].copy()
            for (cc, o), H in self.Hco.items():
                if cc == c:
                    xc -= self.Hinv[c] @ H @ xos[o]
            sc_c += float(xc @ (lam * xc + self.bc[c]))
            self.cam[c] = _exp(xc) @ self.cam[c]
        for o, xo in xos.items():
            self.obj[o] = _exp(xo) @ self.obj[o]
        return np.array([self._chi(robust_on), sc_c, sc_o, 1.0])

    def restore(self):
        self.cam, self.obj = [T.copy() for T in self.bak[0]], [T.copy() for T in self.bak[1]]

    def download(self):
        for c, T in enumerate(self.cam):
            self.P.cam_T[c] = T[:3].ravel()
        for o, T in enumerate(self.obj):
            self.P.obj_T[o] = T[:3].ravel()
        return self.P
