/* ORACLE (test infrastructure only) -- CPU restatement, plain C double precision, of the reference's
 * uncertainty-weighted pose refinement / object-pose bundle adjustment.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the product never does.
 *
 * Restates (paths relative to /root/reference):
 *   ObjectSLAM.optimize robust rounds            lib/object_slam.py:842-896   (graph build :746-839 is the caller's SoA)
 *   EdgeSE3ProjectFromObject::computeError       thirdparty/g2opy/g2o/types/object_slam/types_object_slam.cpp:45-60
 *   EdgeSE3ProjectFromObject::linearizeOplus     same file :70-123
 *   EdgeSE3ProjectFromFixedObject (object fixed) same file :156-201, types_object_slam.h:73-79
 *   OptimizationAlgorithmLevenberg::solve        g2o/core/optimization_algorithm_levenberg.cpp:58-150
 *     computeLambdaInit :152-166, computeScale :168-175
 *   BaseBinaryEdge::constructQuadraticForm       g2o/core/base_binary_edge.hpp:64-129 (robustInformation base_edge.h:94-100)
 *   RobustKernelHuber::robustify                 g2o/core/robust_kernel_impl.cpp:65-78
 *   SparseOptimizer::initializeOptimization      g2o/core/sparse_optimizer.cpp:206-267 (level filter, active sets)
 *   SparseOptimizer::optimize / activeRobustChi2 g2o/core/sparse_optimizer.cpp:366-431, 102-116
 *   BlockSolver::setLambda / restoreDiagonal     g2o/core/block_solver.hpp:526-566 (lambda on every diagonal entry)
 *   VertexSE3Expmap::oplusImpl, SE3Quat::exp     g2o/types/sba/types_six_dof_expmap.h:100-103, slam3d/se3quat.h:220-254
 *
 * The linear system is the FULL dense 6*(#free cameras + #free objects) system, as the reference
 * solves it (no Schur: nothing is marginalised), factorised by dense Cholesky.
 * The HIP path eliminates cameras by Schur complement instead: an independent route to the same step.
 *
 * Pinning: g2o cannot be built in this image (needs Eigen3 / CHOLMOD, absent) => "parity unpinned";
 * pinned instead by finite-difference Jacobian tests, exact recovery on noise-free scenes and the
 * object_slam_demo.py scenario (tests/test_oracle_lm.py).
 * Documented deviation: chi2 used for re-classification is recomputed at the
 * accepted state; the reference reads a stale value when an LM call ends on a rejected trial.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

typedef struct { double q[4]; double t[3]; } pose_t;   /* q = (w,x,y,z) unit, x' = R(q) x + t */

static void q_to_R(const double* q, double* R) {       /* Eigen::Quaternion::toRotationMatrix */
    const double tx = 2 * q[1], ty = 2 * q[2], tz = 2 * q[3];
    const double twx = tx * q[0], twy = ty * q[0], twz = tz * q[0];
    const double txx = tx * q[1], txy = ty * q[1], txz = tz * q[1];
    const double tyy = ty * q[2], tyz = tz * q[2], tzz = tz * q[3];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

static void R_to_q(const double* R, double* q) {       /* Eigen quaternion-from-matrix + normalizeRotation */
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[0] = 0.5 * t;
        t = 0.5 / t;
        q[1] = (R[7] - R[5]) * t; q[2] = (R[2] - R[6]) * t; q[3] = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[i * 4]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
        q[1 + i] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R[k * 3 + j] - R[j * 3 + k]) * t;
        q[1 + j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
        q[1 + k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
    }
    if (q[0] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

static void q_mul(const double* a, const double* b, double* o) {
    o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    o[2] = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
    o[3] = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
}

static void pose_from_T(const double* T, pose_t* p) {  /* T row-major 3x4 */
    double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    R_to_q(R, p->q);
    p->t[0] = T[3]; p->t[1] = T[7]; p->t[2] = T[11];
}
static void pose_to_T(const pose_t* p, double* T) {
    double R[9];
    q_to_R(p->q, R);
    for (int r = 0; r < 3; ++r) { T[4 * r] = R[3 * r]; T[4 * r + 1] = R[3 * r + 1]; T[4 * r + 2] = R[3 * r + 2]; T[4 * r + 3] = p->t[r]; }
}
static void pose_map(const pose_t* p, const double* x, double* o) {
    double R[9];
    q_to_R(p->q, R);
    for (int r = 0; r < 3; ++r) o[r] = R[3 * r] * x[0] + R[3 * r + 1] * x[1] + R[3 * r + 2] * x[2] + p->t[r];
}

/* T <- exp(update) * T, update = [omega, upsilon] (se3quat.h:220-254, types_six_dof_expmap.h:100-103) */
static void pose_oplus(pose_t* p, const double* u) {
    const double* w = u;
    const double* ups = u + 3;
    double theta = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    double Om[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double Om2[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Om2[3 * r + c] = Om[3 * r] * Om[c] + Om[3 * r + 1] * Om[3 + c] + Om[3 * r + 2] * Om[6 + c];
    double R[9], V[9];
    if (theta < 0.00001) {
        for (int i = 0; i < 9; ++i) { R[i] = (i % 4 == 0 ? 1.0 : 0.0) + Om[i] + Om2[i]; V[i] = R[i]; }
    } else {
        double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta), c = (theta - sin(theta)) / pow(theta, 3);
        for (int i = 0; i < 9; ++i) {
            R[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * Om[i] + b * Om2[i];
            V[i] = (i % 4 == 0 ? 1.0 : 0.0) + b * Om[i] + c * Om2[i];
        }
    }
    pose_t e;
    R_to_q(R, e.q);
    for (int r = 0; r < 3; ++r) e.t[r] = V[3 * r] * ups[0] + V[3 * r + 1] * ups[1] + V[3 * r + 2] * ups[2];
    /* SE3Quat::operator* : t = t_e + R(q_e) t_old ; q = q_e * q_old ; normalizeRotation */
    double Re[9], nt[3], nq[4];
    q_to_R(e.q, Re);
    for (int r = 0; r < 3; ++r) nt[r] = e.t[r] + Re[3 * r] * p->t[0] + Re[3 * r + 1] * p->t[1] + Re[3 * r + 2] * p->t[2];
    q_mul(e.q, p->q, nq);
    if (nq[0] < 0) { nq[0] = -nq[0]; nq[1] = -nq[1]; nq[2] = -nq[2]; nq[3] = -nq[3]; }
    double n = sqrt(nq[0] * nq[0] + nq[1] * nq[1] + nq[2] * nq[2] + nq[3] * nq[3]);
    for (int k = 0; k < 4; ++k) p->q[k] = nq[k] / n;
    for (int k = 0; k < 3; ++k) p->t[k] = nt[k];
}

/* ------------------------------------------------------------------------------------------- */
typedef struct {
    int n_cam, n_obj, n_edge;
    pose_t *cam, *obj;
    const uint8_t *cam_fixed, *obj_fixed;
    const int *e_cam, *e_obj;
    const double *e_k, *e_p, *e_uv, *e_info;
    uint8_t* level;          /* 0 = in the optimisation, 1 = outlier */
    uint8_t* robust;         /* Huber kernel attached */
    double delta;
    double* err;             /* [n_edge][2] */
    /* active set of the current round */
    int* cam_idx; int* obj_idx; int nv;
    uint8_t* active;
} graph_t;

static void edge_error(const graph_t* g, int e, double* err) {
    double pw[3], pc[3];
    pose_map(&g->obj[g->e_obj[e]], g->e_p + 3 * e, pw);
    pose_map(&g->cam[g->e_cam[e]], pw, pc);
    const double* k = g->e_k + 4 * e;
    err[0] = g->e_uv[2 * e] - (k[0] * pc[0] / pc[2] + k[2]);
    err[1] = g->e_uv[2 * e + 1] - (k[1] * pc[1] / pc[2] + k[3]);
}
static double chi2_of(const graph_t* g, int e, const double* err) {
    const double* I = g->e_info + 3 * e;
    return err[0] * (I[0] * err[0] + I[1] * err[1]) + err[1] * (I[1] * err[0] + I[2] * err[1]);
}

ORC_API void orc_edge_jacobians(const double* camT, const double* objT, const double* k, const double* p, double* Jobj, double* Jcam) {
    /* exported for the finite-difference test: 2x6 row-major each (types_object_slam.cpp:70-123) */
    pose_t cam, obj;
    pose_from_T(camT, &cam);
    pose_from_T(objT, &obj);
    double pw[3], pc[3], Rc[9];
    pose_map(&obj, p, pw);
    pose_map(&cam, pw, pc);
    q_to_R(cam.q, Rc);
    double PJ[6] = {-(k[0] / pc[2]), 0, k[0] * pc[0] / (pc[2] * pc[2]), 0, -(k[1] / pc[2]), k[1] * pc[1] / (pc[2] * pc[2])};
    double PR[6];
    for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 3; ++c) PR[3 * r + c] = PJ[3 * r] * Rc[c] + PJ[3 * r + 1] * Rc[3 + c] + PJ[3 * r + 2] * Rc[6 + c];
    double Dw[18] = {0, pw[2], -pw[1], 1, 0, 0, -pw[2], 0, pw[0], 0, 1, 0, pw[1], -pw[0], 0, 0, 0, 1};
    double Dc[18] = {0, pc[2], -pc[1], 1, 0, 0, -pc[2], 0, pc[0], 0, 1, 0, pc[1], -pc[0], 0, 0, 0, 1};
    for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 6; ++c) {
            Jobj[6 * r + c] = PR[3 * r] * Dw[c] + PR[3 * r + 1] * Dw[6 + c] + PR[3 * r + 2] * Dw[12 + c];
            Jcam[6 * r + c] = PJ[3 * r] * Dc[c] + PJ[3 * r + 1] * Dc[6 + c] + PJ[3 * r + 2] * Dc[12 + c];
        }
}

static void compute_active_errors(graph_t* g) {
    for (int e = 0; e < g->n_edge; ++e) if (g->active[e]) edge_error(g, e, g->err + 2 * e);
}
static double huber_rho(double e2, double delta, double* rho1) {
    double dsqr = delta * delta;
    if (e2 <= dsqr) { *rho1 = 1.0; return e2; }
    double sq = sqrt(e2);
    *rho1 = delta / sq;
    return 2 * sq * delta - dsqr;
}
static double active_robust_chi2(const graph_t* g) {
    double chi = 0;
    for (int e = 0; e < g->n_edge; ++e) {
        if (!g->active[e]) continue;
        double c = chi2_of(g, e, g->err + 2 * e), r1;
        chi += g->robust[e] ? huber_rho(c, g->delta, &r1) : c;
    }
    return chi;
}

/* H (dense n x n, row-major, symmetric) and b for the active set */
static void build_system(const graph_t* g, double* H, double* b, int n) {
    memset(H, 0, sizeof(double) * (size_t)n * n);
    memset(b, 0, sizeof(double) * (size_t)n);
    for (int e = 0; e < g->n_edge; ++e) {
        if (!g->active[e]) continue;
        const pose_t* cam = &g->cam[g->e_cam[e]];
        const pose_t* obj = &g->obj[g->e_obj[e]];
        double Jo[12], Jc[12];
        /* Jacobians from the poses' quaternions directly (same arithmetic as orc_edge_jacobians) */
        {
            double pw[3], pc[3], Rc[9];
            pose_map(obj, g->e_p + 3 * e, pw);
            pose_map(cam, pw, pc);
            q_to_R(cam->q, Rc);
            const double* k = g->e_k + 4 * e;
            double PJ[6] = {-(k[0] / pc[2]), 0, k[0] * pc[0] / (pc[2] * pc[2]), 0, -(k[1] / pc[2]), k[1] * pc[1] / (pc[2] * pc[2])};
            double PR[6];
            for (int r = 0; r < 2; ++r)
                for (int c = 0; c < 3; ++c) PR[3 * r + c] = PJ[3 * r] * Rc[c] + PJ[3 * r + 1] * Rc[3 + c] + PJ[3 * r + 2] * Rc[6 + c];
            double Dw[18] = {0, pw[2], -pw[1], 1, 0, 0, -pw[2], 0, pw[0], 0, 1, 0, pw[1], -pw[0], 0, 0, 0, 1};
            double Dc[18] = {0, pc[2], -pc[1], 1, 0, 0, -pc[2], 0, pc[0], 0, 1, 0, pc[1], -pc[0], 0, 0, 0, 1};
            for (int r = 0; r < 2; ++r)
                for (int c = 0; c < 6; ++c) {
                    Jo[6 * r + c] = PR[3 * r] * Dw[c] + PR[3 * r + 1] * Dw[6 + c] + PR[3 * r + 2] * Dw[12 + c];
                    Jc[6 * r + c] = PJ[3 * r] * Dc[c] + PJ[3 * r + 1] * Dc[6 + c] + PJ[3 * r + 2] * Dc[12 + c];
                }
        }
        const double* I = g->e_info + 3 * e;
        const double* er = g->err + 2 * e;
        double w = 1.0;
        if (g->robust[e]) huber_rho(chi2_of(g, e, er), g->delta, &w);
        double O[4] = {w * I[0], w * I[1], w * I[1], w * I[2]};                  /* rho' * Omega */
        double orr[2] = {-(I[0] * er[0] + I[1] * er[1]) * w, -(I[1] * er[0] + I[2] * er[1]) * w};
        int io = g->obj_idx[g->e_obj[e]], ic = g->cam_idx[g->e_cam[e]];
        const double* Js[2] = {Jo, Jc};
        int idx[2] = {io, ic};
        for (int a = 0; a < 2; ++a) {
            if (idx[a] < 0) continue;
            for (int r = 0; r < 6; ++r) {
                double jo0 = Js[a][r] * O[0] + Js[a][6 + r] * O[2], jo1 = Js[a][r] * O[1] + Js[a][6 + r] * O[3];   /* (J^T Omega) row r */
                b[6 * idx[a] + r] += Js[a][r] * orr[0] + Js[a][6 + r] * orr[1];
                for (int bb = 0; bb < 2; ++bb) {
                    if (idx[bb] < 0) continue;
                    for (int c = 0; c < 6; ++c)
                        H[(size_t)(6 * idx[a] + r) * n + 6 * idx[bb] + c] += jo0 * Js[bb][c] + jo1 * Js[bb][6 + c];
                }
            }
        }
    }
}

/* dense Cholesky solve; returns 0 if not positive definite */
static int chol_solve(const double* A, const double* b, double* x, int n, double* L) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = A[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) s -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
            if (i == j) { if (!(s > 0) || !isfinite(s)) return 0; L[(size_t)i * n + i] = sqrt(s); }
            else L[(size_t)i * n + j] = s / L[(size_t)j * n + j];
        }
    for (int i = 0; i < n; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= L[(size_t)i * n + k] * x[k]; x[i] = s / L[(size_t)i * n + i]; }
    for (int i = n - 1; i >= 0; --i) { double s = x[i]; for (int k = i + 1; k < n; ++k) s -= L[(size_t)k * n + i] * x[k]; x[i] = s / L[(size_t)i * n + i]; }
    return 1;
}

static void apply_update(graph_t* g, const double* x) {
    for (int c = 0; c < g->n_cam; ++c) if (g->cam_idx[c] >= 0) pose_oplus(&g->cam[c], x + 6 * g->cam_idx[c]);
    for (int o = 0; o < g->n_obj; ++o) if (g->obj_idx[o] >= 0) pose_oplus(&g->obj[o], x + 6 * g->obj_idx[o]);
}

/* SparseOptimizer::initializeOptimization(0) + optimize(iterations) with Levenberg */
static int optimize_round(graph_t* g, int iterations, int* lm_trials_total) {
    /* active edges: level 0 and not all vertices fixed; active vertices: free with >= 1 active edge */
    int nv = 0;
    for (int c = 0; c < g->n_cam; ++c) g->cam_idx[c] = -1;
    for (int o = 0; o < g->n_obj; ++o) g->obj_idx[o] = -1;
    for (int e = 0; e < g->n_edge; ++e) {
        int c = g->e_cam[e], o = g->e_obj[e];
        g->active[e] = (g->level[e] == 0) && !(g->cam_fixed[c] && g->obj_fixed[o]);
        if (g->active[e]) {
            if (!g->cam_fixed[c] && g->cam_idx[c] < 0) g->cam_idx[c] = -2;
            if (!g->obj_fixed[o] && g->obj_idx[o] < 0) g->obj_idx[o] = -2;
        }
    }
    for (int c = 0; c < g->n_cam; ++c) if (g->cam_idx[c] == -2) g->cam_idx[c] = nv++;
    for (int o = 0; o < g->n_obj; ++o) if (g->obj_idx[o] == -2) g->obj_idx[o] = nv++;
    g->nv = nv;
    if (nv == 0) return -1;
    const int n = 6 * nv;
    double* H = (double*)malloc(sizeof(double) * (size_t)n * n);
    double* A = (double*)malloc(sizeof(double) * (size_t)n * n);
    double* L = (double*)calloc((size_t)n * n, sizeof(double));
    double* b = (double*)malloc(sizeof(double) * n);
    double* x = (double*)malloc(sizeof(double) * n);
    pose_t* cam_bak = (pose_t*)malloc(sizeof(pose_t) * (g->n_cam > 0 ? g->n_cam : 1));
    pose_t* obj_bak = (pose_t*)malloc(sizeof(pose_t) * (g->n_obj > 0 ? g->n_obj : 1));
    double lambda = -1, ni = 2;
    int done = 0;
    for (int it = 0; it < iterations; ++it) {
        compute_active_errors(g);
        double currentChi = active_robust_chi2(g), tempChi = currentChi;
        build_system(g, H, b, n);
        if (it == 0) {
            double maxd = 0;
            for (int i = 0; i < n; ++i) maxd = fmax(fabs(H[(size_t)i * n + i]), maxd);
            lambda = 1e-5 * maxd;
            ni = 2;
        }
        double rho = 0;
        int qmax = 0;
        do {
            memcpy(cam_bak, g->cam, sizeof(pose_t) * g->n_cam);
            memcpy(obj_bak, g->obj, sizeof(pose_t) * g->n_obj);
            memcpy(A, H, sizeof(double) * (size_t)n * n);
            for (int i = 0; i < n; ++i) A[(size_t)i * n + i] += lambda;
            int ok2 = chol_solve(A, b, x, n, L);
            if (!ok2) memset(x, 0, sizeof(double) * n);
            apply_update(g, x);
            compute_active_errors(g);
            tempChi = active_robust_chi2(g);
            if (!ok2) tempChi = DBL_MAX;
            rho = currentChi - tempChi;
            double scale = 0;
            for (int j = 0; j < n; ++j) scale += x[j] * (lambda * x[j] + b[j]);
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && isfinite(tempChi)) {
                double alpha = 1. - pow(2 * rho - 1, 3);
                alpha = fmin(alpha, 2. / 3.);
                double scaleFactor = fmax(1. / 3., alpha);
                lambda *= scaleFactor;
                ni = 2;
                currentChi = tempChi;
            } else {
                lambda *= ni;
                ni *= 2;
                memcpy(g->cam, cam_bak, sizeof(pose_t) * g->n_cam);
                memcpy(g->obj, obj_bak, sizeof(pose_t) * g->n_obj);
                if (!isfinite(lambda)) break;
            }
            qmax++;
            if (lm_trials_total) (*lm_trials_total)++;
        } while (rho < 0 && qmax < 10);
        done++;
        if (qmax == 10 || rho == 0 || !isfinite(lambda)) break;      /* Terminate */
    }
    free(H); free(A); free(L); free(b); free(x); free(cam_bak); free(obj_bak);
    return done;
}

/* ObjectSLAM.optimize rounds (object_slam.py:842-896) over a flat SoA graph.
 * cam_T/obj_T: row-major 3x4, updated in place.  edge_info = (xx, xy, yy) of Omega.
 * stats (optional, 4 ints): rounds executed, LM iterations, LM trials, final num_good. */
ORC_API int orc_optimize(int n_cam, int n_obj, int n_edge, double* cam_T, const uint8_t* cam_fixed, double* obj_T,
                         const uint8_t* obj_fixed, const int* edge_cam, const int* edge_obj, const double* edge_camk,
                         const double* edge_p, const double* edge_uv, const double* edge_info, uint8_t* edge_inlier,
                         double* edge_chi2, const int* its, int n_rounds, int init_with_outliers, double chi2_thr,
                         double huber_delta, int* stats) {
    graph_t g;
    memset(&g, 0, sizeof(g));
    g.n_cam = n_cam; g.n_obj = n_obj; g.n_edge = n_edge;
    g.cam = (pose_t*)malloc(sizeof(pose_t) * (n_cam > 0 ? n_cam : 1));
    g.obj = (pose_t*)malloc(sizeof(pose_t) * (n_obj > 0 ? n_obj : 1));
    for (int c = 0; c < n_cam; ++c) pose_from_T(cam_T + 12 * c, &g.cam[c]);
    for (int o = 0; o < n_obj; ++o) pose_from_T(obj_T + 12 * o, &g.obj[o]);
    g.cam_fixed = cam_fixed; g.obj_fixed = obj_fixed;
    g.e_cam = edge_cam; g.e_obj = edge_obj; g.e_k = edge_camk; g.e_p = edge_p; g.e_uv = edge_uv; g.e_info = edge_info;
    g.level = (uint8_t*)calloc(n_edge > 0 ? n_edge : 1, 1);
    g.robust = (uint8_t*)malloc(n_edge > 0 ? n_edge : 1);
    memset(g.robust, 1, n_edge > 0 ? n_edge : 1);
    g.active = (uint8_t*)calloc(n_edge > 0 ? n_edge : 1, 1);
    g.err = (double*)calloc((size_t)(n_edge > 0 ? n_edge : 1) * 2, sizeof(double));
    g.cam_idx = (int*)malloc(sizeof(int) * (n_cam > 0 ? n_cam : 1));
    g.obj_idx = (int*)malloc(sizeof(int) * (n_obj > 0 ? n_obj : 1));
    g.delta = huber_delta;
    int num_good = 0, rounds = 0, lm_its = 0, lm_trials = 0;
    if (init_with_outliers) {
        num_good = n_edge;
    } else {
        for (int e = 0; e < n_edge; ++e) {
            double er[2];
            edge_error(&g, e, er);
            double c = chi2_of(&g, e, er);
            if (edge_chi2) edge_chi2[e] = c;
            if (c > chi2_thr) { g.level[e] = 1; edge_inlier[e] = 0; }
            else { num_good++; g.level[e] = 0; edge_inlier[e] = 1; }
        }
    }
    const int drop = (n_rounds / 2) > 1 ? (n_rounds / 2) : 1;
    for (int it = 0; it < n_rounds; ++it) {
        if (n_edge < 4 || num_good < 4) break;
        int r = optimize_round(&g, its[it], &lm_trials);
        if (r > 0) lm_its += r;
        rounds++;
        num_good = 0;
        for (int e = 0; e < n_edge; ++e) {
            double er[2];
            edge_error(&g, e, er);
            double c = chi2_of(&g, e, er);
            if (edge_chi2) edge_chi2[e] = c;
            if (c > chi2_thr) { g.level[e] = 1; edge_inlier[e] = 0; }
            else { num_good++; g.level[e] = 0; edge_inlier[e] = 1; }
            if (it == drop) g.robust[e] = 0;
        }
    }
    for (int c = 0; c < n_cam; ++c) pose_to_T(&g.cam[c], cam_T + 12 * c);
    for (int o = 0; o < n_obj; ++o) pose_to_T(&g.obj[o], obj_T + 12 * o);
    if (stats) { stats[0] = rounds; stats[1] = lm_its; stats[2] = lm_trials; stats[3] = num_good; }
    free(g.cam); free(g.obj); free(g.level); free(g.robust); free(g.active); free(g.err); free(g.cam_idx); free(g.obj_idx);
    return 0;
}

/* ONE SparseOptimizer::initializeOptimization(0) + optimize(iterations) over caller-kept state: poses as g2o keeps them
 * (SE3Quat = unit quaternion (w,x,y,z) + t, 7 doubles each), per-edge level and robust-kernel flags.  This is what the
 * recording g2o stub of tests/golden/make_slam_golden.py calls when the REFERENCE's own ObjectSLAM.optimize
 * (lib/object_slam.py:703-930, imported unmodified) drives the rounds, so that its Python control flow is the thing pinned.
 * err_out [n_edge][2]: errors of the active edges at the accepted state (others untouched).  Returns LM iterations done. */
ORC_API int orc_lm_round(int n_cam, int n_obj, int n_edge, double* cam_qt, const uint8_t* cam_fixed, double* obj_qt,
                         const uint8_t* obj_fixed, const int* edge_cam, const int* edge_obj, const double* edge_camk,
                         const double* edge_p, const double* edge_uv, const double* edge_info, const uint8_t* level,
                         const uint8_t* robust, double huber_delta, int iterations, double* err_out, int* lm_trials) {
    graph_t g;
    memset(&g, 0, sizeof(g));
    g.n_cam = n_cam; g.n_obj = n_obj; g.n_edge = n_edge;
    g.cam = (pose_t*)malloc(sizeof(pose_t) * (n_cam > 0 ? n_cam : 1));
    g.obj = (pose_t*)malloc(sizeof(pose_t) * (n_obj > 0 ? n_obj : 1));
    for (int c = 0; c < n_cam; ++c) { memcpy(g.cam[c].q, cam_qt + 7 * c, 4 * sizeof(double)); memcpy(g.cam[c].t, cam_qt + 7 * c + 4, 3 * sizeof(double)); }
    for (int o = 0; o < n_obj; ++o) { memcpy(g.obj[o].q, obj_qt + 7 * o, 4 * sizeof(double)); memcpy(g.obj[o].t, obj_qt + 7 * o + 4, 3 * sizeof(double)); }
    g.cam_fixed = cam_fixed; g.obj_fixed = obj_fixed;
    g.e_cam = edge_cam; g.e_obj = edge_obj; g.e_k = edge_camk; g.e_p = edge_p; g.e_uv = edge_uv; g.e_info = edge_info;
    g.level = (uint8_t*)malloc(n_edge > 0 ? n_edge : 1);
    g.robust = (uint8_t*)malloc(n_edge > 0 ? n_edge : 1);
    memcpy(g.level, level, n_edge);
    memcpy(g.robust, robust, n_edge);
    g.active = (uint8_t*)calloc(n_edge > 0 ? n_edge : 1, 1);
    g.err = (double*)calloc((size_t)(n_edge > 0 ? n_edge : 1) * 2, sizeof(double));
    g.cam_idx = (int*)malloc(sizeof(int) * (n_cam > 0 ? n_cam : 1));
    g.obj_idx = (int*)malloc(sizeof(int) * (n_obj > 0 ? n_obj : 1));
    g.delta = huber_delta;
    int trials = 0;
    int r = optimize_round(&g, iterations, &trials);
    if (r >= 0) {
        compute_active_errors(&g);
        for (int e = 0; e < n_edge; ++e) if (g.active[e]) { err_out[2 * e] = g.err[2 * e]; err_out[2 * e + 1] = g.err[2 * e + 1]; }
    }
    for (int c = 0; c < n_cam; ++c) { memcpy(cam_qt + 7 * c, g.cam[c].q, 4 * sizeof(double)); memcpy(cam_qt + 7 * c + 4, g.cam[c].t, 3 * sizeof(double)); }
    for (int o = 0; o < n_obj; ++o) { memcpy(obj_qt + 7 * o, g.obj[o].q, 4 * sizeof(double)); memcpy(obj_qt + 7 * o + 4, g.obj[o].t, 3 * sizeof(double)); }
    if (lm_trials) *lm_trials = trials;
    free(g.cam); free(g.obj); free(g.level); free(g.robust); free(g.active); free(g.err); free(g.cam_idx); free(g.obj_idx);
    return r;
}
/* SE3Quat(R, t) construction and SE3Quat::matrix() / to_homogeneous_matrix (se3quat.h:55-60,104-113) for the same stub */
ORC_API void orc_pose_from_T(const double* T12, double* qt7) {
    pose_t p;
    pose_from_T(T12, &p);
    memcpy(qt7, p.q, 4 * sizeof(double)); memcpy(qt7 + 4, p.t, 3 * sizeof(double));
}
ORC_API void orc_pose_to_T(const double* qt7, double* T12) {
    pose_t p;
    memcpy(p.q, qt7, 4 * sizeof(double)); memcpy(p.t, qt7 + 4, 3 * sizeof(double));
    pose_to_T(&p, T12);
}
/* computeError of one edge from (q,t) poses (types_object_slam.cpp:45-60) */
ORC_API void orc_edge_error_qt(const double* cam_qt, const double* obj_qt, const double* k, const double* p, const double* uv, double* err) {
    graph_t g;
    memset(&g, 0, sizeof(g));
    pose_t cam, obj;
    memcpy(cam.q, cam_qt, 4 * sizeof(double)); memcpy(cam.t, cam_qt + 4, 3 * sizeof(double));
    memcpy(obj.q, obj_qt, 4 * sizeof(double)); memcpy(obj.t, obj_qt + 4, 3 * sizeof(double));
    int zero = 0;
    g.cam = &cam; g.obj = &obj; g.e_cam = &zero; g.e_obj = &zero; g.e_k = k; g.e_p = p; g.e_uv = uv;
    edge_error(&g, 0, err);
}

/* small exports for the unit tests */
ORC_API void orc_pose_oplus(double* T12, const double* u6) {
    pose_t p;
    pose_from_T(T12, &p);
    pose_oplus(&p, u6);
    pose_to_T(&p, T12);
}
ORC_API void orc_edge_error(const double* camT, const double* objT, const double* k, const double* p, const double* uv, double* err) {
    graph_t g;
    memset(&g, 0, sizeof(g));
    pose_t cam, obj;
    pose_from_T(camT, &cam);
    pose_from_T(objT, &obj);
    int zero = 0;
    g.cam = &cam; g.obj = &obj; g.e_cam = &zero; g.e_obj = &zero; g.e_k = k; g.e_p = p; g.e_uv = uv;
    edge_error(&g, 0, err);
}
