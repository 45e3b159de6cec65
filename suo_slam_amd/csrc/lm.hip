// Uncertainty-weighted pose refinement / object-pose bundle adjustment on gfx950 (fp64).
//
// Replaces the g2o machinery driven by ObjectSLAM.optimize (/root/reference/lib/object_slam.py:703-930):
//   robust rounds + chi2 re-classification                  lib/object_slam.py:842-896
//   EdgeSE3ProjectFromObject / ...FromFixedObject            thirdparty/g2opy/g2o/types/object_slam/types_object_slam.cpp:45-60,70-123,156-201
//   OptimizationAlgorithmLevenberg::solve (lambda schedule)  g2o/core/optimization_algorithm_levenberg.cpp:58-175
//   constructQuadraticForm with Huber weighting              g2o/core/base_binary_edge.hpp:64-129, robust_kernel_impl.cpp:65-78
//   SE3 exp-map left update                                  g2o/types/slam3d/se3quat.h:220-254
//
// MI355X-first structure.  The reference rebuilds a pointer graph through one pybind call per edge and
// solves the full 6*(#cams+#objs) system on one CPU thread.  Here a problem is a flat SoA resident in
// HBM and ONE workgroup runs every round / LM iteration / trial of it without returning to the host:
//   * edges are grouped by (camera, object) pair; one wavefront linearises a pair (lanes over its <=41
//     keypoint edges) and butterfly-reduces the 6x6 blocks H_cc, H_oo, H_co and the gradients;
//   * cameras are eliminated by Schur complement (block-diagonal H_cc, 6x6 inverses, one thread each),
//     the reduced object system lives in LDS and is factorised by a workgroup Cholesky; with no free
//     camera (single-view mode) the system is block diagonal and each object is solved by one thread;
//   * g2o's lambda schedule (tau = 1e-5, rho-gain update, nu doubling, <= 10 trials) runs on-device.
// Many problems (frames) run concurrently as independent workgroups.
#include <type_traits>

#include "suo_internal.h"

namespace suo {

#define DEV __device__ __forceinline__

constexpr int LM_THREADS = 256;
constexpr int LM_MAX_SCHUR_OBJ = 16;                 // reduced system <= 96 x 96 doubles in LDS
constexpr int LM_NS = 6 * LM_MAX_SCHUR_OBJ;

struct Pose { double q[4]; double t[3]; };

DEV void q_to_R(const double* q, double* R) {
    const double tx = 2 * q[1], ty = 2 * q[2], tz = 2 * q[3];
    const double twx = tx * q[0], twy = ty * q[0], twz = tz * q[0];
    const double txx = tx * q[1], txy = ty * q[1], txz = tz * q[1];
    const double tyy = ty * q[2], tyz = tz * q[2], tzz = tz * q[3];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

DEV void R_to_q(const double* R, double* q) {
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
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
        double v[3];
        v[i] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R[k * 3 + j] - R[j * 3 + k]) * t;
        v[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
        v[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
        q[1] = v[0]; q[2] = v[1]; q[3] = v[2];
    }
    if (q[0] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

DEV void q_mul(const double* a, const double* b, double* o) {
    o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    o[2] = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
    o[3] = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
}

DEV void pose_from_T(const double* T, Pose& p) {
    const double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    R_to_q(R, p.q);
    p.t[0] = T[3]; p.t[1] = T[7]; p.t[2] = T[11];
}
DEV void pose_to_T(const Pose& p, double* T) {
    double R[9];
    q_to_R(p.q, R);
    for (int r = 0; r < 3; ++r) { T[4 * r] = R[3 * r]; T[4 * r + 1] = R[3 * r + 1]; T[4 * r + 2] = R[3 * r + 2]; T[4 * r + 3] = p.t[r]; }
}

// T <- exp([omega, upsilon]) * T
DEV void pose_oplus(Pose& p, const double* u) {
    const double* w = u;
    const double* ups = u + 3;
    const double theta = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    const double Om[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double Om2[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Om2[3 * r + c] = Om[3 * r] * Om[c] + Om[3 * r + 1] * Om[3 + c] + Om[3 * r + 2] * Om[6 + c];
    double R[9], V[9];
    if (theta < 0.00001) {
        for (int i = 0; i < 9; ++i) { R[i] = (i % 4 == 0 ? 1.0 : 0.0) + Om[i] + Om2[i]; V[i] = R[i]; }
    } else {
        const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta), c = (theta - sin(theta)) / pow(theta, 3.0);
        for (int i = 0; i < 9; ++i) {
            R[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * Om[i] + b * Om2[i];
            V[i] = (i % 4 == 0 ? 1.0 : 0.0) + b * Om[i] + c * Om2[i];
        }
    }
    double eq[4], et[3];
    R_to_q(R, eq);
    for (int r = 0; r < 3; ++r) et[r] = V[3 * r] * ups[0] + V[3 * r + 1] * ups[1] + V[3 * r + 2] * ups[2];
    double Re[9], nt[3], nq[4];
    q_to_R(eq, Re);
    for (int r = 0; r < 3; ++r) nt[r] = et[r] + Re[3 * r] * p.t[0] + Re[3 * r + 1] * p.t[1] + Re[3 * r + 2] * p.t[2];
    q_mul(eq, p.q, nq);
    if (nq[0] < 0) { nq[0] = -nq[0]; nq[1] = -nq[1]; nq[2] = -nq[2]; nq[3] = -nq[3]; }
    const double n = sqrt(nq[0] * nq[0] + nq[1] * nq[1] + nq[2] * nq[2] + nq[3] * nq[3]);
    for (int k = 0; k < 4; ++k) p.q[k] = nq[k] / n;
    for (int k = 0; k < 3; ++k) p.t[k] = nt[k];
}

DEV double huber_rho(double e2, double delta, double& rho1) {
    const double dsqr = delta * delta;
    if (e2 <= dsqr) { rho1 = 1.0; return e2; }
    const double sq = sqrt(e2);
    rho1 = delta / sq;
    return 2 * sq * delta - dsqr;
}

DEV double wsum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// deterministic workgroup sum / max (fixed order); `red` = LM_THREADS/64 doubles of LDS
DEV double block_sum(double v, double* red) {
    v = wsum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0;
    for (int i = 0; i < LM_THREADS / 64; ++i) s += red[i];
    return s;
}
DEV double block_max(double v, double* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = red[0];
    for (int i = 1; i < LM_THREADS / 64; ++i) s = fmax(s, red[i]);
    return s;
}

// in-place inverse of a symmetric positive definite 6x6 by Cholesky; false if not PD
__device__ bool spd_inverse6(const double* A, double* Ainv) {
    double L[36];
    for (int i = 0; i < 36; ++i) L[i] = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = A[i * 6 + j];
            for (int k = 0; k < j; ++k) s -= L[i * 6 + k] * L[j * 6 + k];
            if (i == j) { if (!(s > 0) || !isfinite(s)) return false; L[i * 6 + i] = sqrt(s); }
            else L[i * 6 + j] = s / L[j * 6 + j];
        }
    for (int c = 0; c < 6; ++c) {          // solve A x = e_c
        double y[6], x[6];
        for (int i = 0; i < 6; ++i) { double s = (i == c) ? 1.0 : 0.0; for (int k = 0; k < i; ++k) s -= L[i * 6 + k] * y[k]; y[i] = s / L[i * 6 + i]; }
        for (int i = 5; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < 6; ++k) s -= L[k * 6 + i] * x[k]; x[i] = s / L[i * 6 + i]; }
        for (int i = 0; i < 6; ++i) Ainv[i * 6 + c] = x[i];
    }
    return true;
}

struct LmProblem {
    // sizes
    int n_cam, n_obj, n_edge, n_pair;
    // poses (row-major 3x4), in/out
    double* cam_T; double* obj_T;
    const uint8_t* cam_fixed; const uint8_t* obj_fixed;
    // edges, sorted by pair
    const int* edge_pair;                    // [n_edge]
    const double* edge_k; const double* edge_p; const double* edge_uv; const double* edge_info;
    uint8_t* edge_inlier; double* edge_chi2; // in/out, out
    // pairs
    const int* pair_cam; const int* pair_obj; const int* pair_start;   // pair_start [n_pair+1]
    const int* cam_pair_ptr; const int* cam_pair_idx;                  // CSR camera -> pairs
    const int* obj_pair_ptr; const int* obj_pair_idx;                  // CSR object -> pairs
    // parameters
    int its[8]; int n_rounds; int init_with_outliers; double chi2_thr; double huber_delta;
    // scratch (device)
    Pose* cam; Pose* obj; Pose* cam_bak; Pose* obj_bak;
    double* err;                 // [n_edge][2]
    double* jac;                 // [n_edge][29]: Jc(12) Jo(12) rho'*Omega (xx,xy,yy) omega_r(2)
    uint8_t* level;              // [n_edge]
    double* pair_part;           // [n_pair][90]: Hcc(21) Hoo(21) Hco(36) bc(6) bo(6)
    double* Hcc; double* bc;     // [n_cam][36], [n_cam][6]
    double* Hoo; double* bo;     // [n_obj][36], [n_obj][6]
    double* Hcc_inv;             // [n_cam][36]
    double* Y;                   // [n_pair][36]  = Hcc_inv * Hco
    double* yc;                  // [n_cam][6]    = Hcc_inv * bc
    double* xc; double* xo;      // [n_cam][6], [n_obj][6]
    int* obj_slot;               // [n_obj] position in the reduced system (free objects only) or -1
    int* stats;                  // [4] rounds, LM iterations, LM trials, num_good
};

DEV bool edge_active(const LmProblem& P, int e) {
    const int p = P.edge_pair[e];
    return P.level[e] == 0 && !(P.cam_fixed[P.pair_cam[p]] && P.obj_fixed[P.pair_obj[p]]);
}

DEV void edge_error(const LmProblem& P, int e, double* err, double* pw_out, double* pc_out) {
    const int p = P.edge_pair[e];
    const Pose& cam = P.cam[P.pair_cam[p]];
    const Pose& obj = P.obj[P.pair_obj[p]];
    double Ro[9], Rc[9], pw[3], pc[3];
    q_to_R(obj.q, Ro);
    q_to_R(cam.q, Rc);
    const double* x = P.edge_p + 3 * e;
    for (int r = 0; r < 3; ++r) pw[r] = Ro[3 * r] * x[0] + Ro[3 * r + 1] * x[1] + Ro[3 * r + 2] * x[2] + obj.t[r];
    for (int r = 0; r < 3; ++r) pc[r] = Rc[3 * r] * pw[0] + Rc[3 * r + 1] * pw[1] + Rc[3 * r + 2] * pw[2] + cam.t[r];
    const double* k = P.edge_k + 4 * e;
    err[0] = P.edge_uv[2 * e] - (k[0] * pc[0] / pc[2] + k[2]);
    err[1] = P.edge_uv[2 * e + 1] - (k[1] * pc[1] / pc[2] + k[3]);
    if (pw_out) { for (int r = 0; r < 3; ++r) { pw_out[r] = pw[r]; pc_out[r] = pc[r]; } }
}
DEV double edge_chi2(const LmProblem& P, int e, const double* err) {
    const double* I = P.edge_info + 3 * e;
    return err[0] * (I[0] * err[0] + I[1] * err[1]) + err[1] * (I[1] * err[0] + I[2] * err[1]);
}

// computeActiveErrors + activeRobustChi2; with_jac also stores the edge Jacobians (linearizeOplus) and the
// Huber-weighted information / gradient factors used by constructQuadraticForm.  Threads over edges.
DEV double active_errors_and_chi2(const LmProblem& P, bool robust_on, bool with_jac, double* red) {
    double c = 0;
    for (int e = threadIdx.x; e < P.n_edge; e += LM_THREADS) {
        if (!edge_active(P, e)) continue;
        double er[2], pw[3], pc[3];
        edge_error(P, e, er, pw, pc);
        P.err[2 * e] = er[0];
        P.err[2 * e + 1] = er[1];
        const double c2 = edge_chi2(P, e, er);
        double w = 1.0;
        c += robust_on ? huber_rho(c2, P.huber_delta, w) : c2;
        if (with_jac) {
            double Rc[9];
            q_to_R(P.cam[P.pair_cam[P.edge_pair[e]]].q, Rc);
            const double* k = P.edge_k + 4 * e;
            const double PJ[6] = {-(k[0] / pc[2]), 0, k[0] * pc[0] / (pc[2] * pc[2]), 0, -(k[1] / pc[2]), k[1] * pc[1] / (pc[2] * pc[2])};
            double PR[6];
            for (int r = 0; r < 2; ++r)
                for (int cc = 0; cc < 3; ++cc) PR[3 * r + cc] = PJ[3 * r] * Rc[cc] + PJ[3 * r + 1] * Rc[3 + cc] + PJ[3 * r + 2] * Rc[6 + cc];
            const double Dw[18] = {0, pw[2], -pw[1], 1, 0, 0, -pw[2], 0, pw[0], 0, 1, 0, pw[1], -pw[0], 0, 0, 0, 1};
            const double Dc[18] = {0, pc[2], -pc[1], 1, 0, 0, -pc[2], 0, pc[0], 0, 1, 0, pc[1], -pc[0], 0, 0, 0, 1};
            double* J = P.jac + 29 * (size_t)e;
            for (int r = 0; r < 2; ++r)
                for (int cc = 0; cc < 6; ++cc) {
                    J[6 * r + cc] = PJ[3 * r] * Dc[cc] + PJ[3 * r + 1] * Dc[6 + cc] + PJ[3 * r + 2] * Dc[12 + cc];            // Jc
                    J[12 + 6 * r + cc] = PR[3 * r] * Dw[cc] + PR[3 * r + 1] * Dw[6 + cc] + PR[3 * r + 2] * Dw[12 + cc];       // Jo
                }
            const double* I = P.edge_info + 3 * e;
            J[24] = w * I[0]; J[25] = w * I[1]; J[26] = w * I[2];
            J[27] = -(I[0] * er[0] + I[1] * er[1]) * w;
            J[28] = -(I[1] * er[0] + I[2] * er[1]) * w;
        }
    }
    return block_sum(c, red);
}

// One thread per (pair, entry): entry k of [Hcc(21) | Hoo(21) | Hco(36) | bc(6) | bo(6)] summed over the pair's
// active edges in edge order (deterministic, no cross-lane reduction).
DEV void accumulate_pairs(const LmProblem& P) {
    for (int idx = threadIdx.x; idx < P.n_pair * 90; idx += LM_THREADS) {
        const int p = idx / 90, k = idx - p * 90;
        const bool cfree = !P.cam_fixed[P.pair_cam[p]], ofree = !P.obj_fixed[P.pair_obj[p]];
        int a_off, b_off, r, c, kind;      // kind 0: A^T O B block entry (r,c); 1: gradient entry r
        if (k < 21) { if (!cfree) continue; kind = 0; a_off = 0; b_off = 0; int u = k; r = 0; while (u >= 6 - r) { u -= 6 - r; ++r; } c = r + u; }
        else if (k < 42) { if (!ofree) continue; kind = 0; a_off = 12; b_off = 12; int u = k - 21; r = 0; while (u >= 6 - r) { u -= 6 - r; ++r; } c = r + u; }
        else if (k < 78) { if (!(cfree && ofree)) continue; kind = 0; a_off = 0; b_off = 12; r = (k - 42) / 6; c = (k - 42) - r * 6; }
        else if (k < 84) { if (!cfree) continue; kind = 1; a_off = 0; b_off = 0; r = k - 78; c = 0; }
        else { if (!ofree) continue; kind = 1; a_off = 12; b_off = 0; r = k - 84; c = 0; }
        double s = 0;
        for (int e = P.pair_start[p]; e < P.pair_start[p + 1]; ++e) {
            if (P.level[e] != 0) continue;
            const double* J = P.jac + 29 * (size_t)e;
            const double a0 = J[a_off + r], a1 = J[a_off + 6 + r];
            if (kind == 0) s += (a0 * J[24] + a1 * J[25]) * J[b_off + c] + (a0 * J[25] + a1 * J[26]) * J[b_off + 6 + c];
            else s += a0 * J[27] + a1 * J[28];
        }
        P.pair_part[idx] = s;
    }
}

// solve the symmetric positive definite 6x6 system A x = b by Cholesky; false if not PD
__device__ bool spd_solve6(const double* A, const double* b, double* x) {
    double L[36];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = A[i * 6 + j];
            for (int k = 0; k < j; ++k) s -= L[i * 6 + k] * L[j * 6 + k];
            if (i == j) { if (!(s > 0) || !isfinite(s)) return false; L[i * 6 + i] = sqrt(s); }
            else L[i * 6 + j] = s / L[j * 6 + j];
        }
    double y[6];
    for (int i = 0; i < 6; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= L[i * 6 + k] * y[k]; y[i] = s / L[i * 6 + i]; }
    for (int i = 5; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < 6; ++k) s -= L[k * 6 + i] * x[k]; x[i] = s / L[i * 6 + i]; }
    return true;
}

DEV void unpack_sym21(const double* s, double* A) {
    int u = 0;
    for (int r = 0; r < 6; ++r)
        for (int c = r; c < 6; ++c) { A[r * 6 + c] = s[u]; A[c * 6 + r] = s[u]; ++u; }
}

constexpr int LM_LDS_BYTES = 150 * 1024;     // dynamic LDS per workgroup (160 KiB per CU on gfx950)

// Move one array of the problem into LDS when it still fits (flat pointers address LDS transparently): the
// working set of a single-view frame (poses, 6x6 blocks, ~100 edges and their Jacobians) then never leaves
// the CU during the ~80 LM trials.  Larger problems keep whatever does not fit in HBM/L2.
template <typename T>
DEV void lds_relocate(T*& ptr, size_t count, unsigned char* lds, size_t& off, size_t cap, bool copy_in) {
    const size_t bytes = (count * sizeof(T) + 15) & ~(size_t)15;
    if (count == 0 || off + bytes > cap) return;
    typedef typename std::remove_const<T>::type U;
    U* dst = (U*)(lds + off);
    if (copy_in)
        for (size_t i = threadIdx.x; i < count; i += LM_THREADS) dst[i] = ptr[i];
    ptr = dst;
    off += bytes;
}

__global__ __launch_bounds__(LM_THREADS) void lm_kernel(const LmProblem* __restrict__ problems, int lds_bytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lm_lds[];
    const LmProblem& G = problems[blockIdx.x];      // the problem as laid out in HBM
    LmProblem P = G;                                // working copy whose pointers may be redirected to LDS
    __shared__ double red[LM_THREADS / 64];
    __shared__ int sh_flag, sh_ok, sh_good;
    const int tid = threadIdx.x;

    if (tid == 0) {
        int ns = 0, nfc = 0;
        for (int o = 0; o < G.n_obj; ++o) G.obj_slot[o] = G.obj_fixed[o] ? -1 : ns++;
        for (int c = 0; c < G.n_cam; ++c) nfc += G.cam_fixed[c] ? 0 : 1;
        sh_flag = ns | (nfc << 16);
    }
    __syncthreads();
    const int n_free_obj = sh_flag & 0xffff, n_free_cam = sh_flag >> 16;
    const bool schur = n_free_cam > 0 && n_free_obj > 0;
    const int ns = 6 * n_free_obj;
    __syncthreads();
    if (schur && n_free_obj > LM_MAX_SCHUR_OBJ) {       // unsupported size: report and leave poses untouched
        if (tid == 0) { G.stats[0] = -1; G.stats[1] = G.stats[2] = G.stats[3] = 0; }
        return;
    }
    // ---- LDS carve-up: reduced system first (Schur only), then the problem's arrays by access frequency ---
    size_t off = 0;
    double* S = (double*)lm_lds;            // reduced (object) system / its Cholesky factor, ns x ns
    double* rhs = S;
    double* colbuf = S;
    if (schur) {
        off = ((size_t)ns * ns * sizeof(double) + 15) & ~(size_t)15;
        rhs = (double*)(lm_lds + off); off += ((size_t)ns * sizeof(double) + 15) & ~(size_t)15;
        colbuf = (double*)(lm_lds + off); off += ((size_t)ns * sizeof(double) + 15) & ~(size_t)15;
    }
    {
        const size_t C = P.n_cam, O = P.n_obj, E = P.n_edge, NP = P.n_pair, cap = (size_t)lds_bytes;
        lds_relocate(P.cam, C, lm_lds, off, cap, false);
        lds_relocate(P.obj, O, lm_lds, off, cap, false);
        lds_relocate(P.cam_bak, C, lm_lds, off, cap, false);
        lds_relocate(P.obj_bak, O, lm_lds, off, cap, false);
        lds_relocate(P.cam_fixed, C, lm_lds, off, cap, true);
        lds_relocate(P.obj_fixed, O, lm_lds, off, cap, true);
        lds_relocate(P.obj_slot, O, lm_lds, off, cap, true);
        lds_relocate(P.Hoo, 36 * O, lm_lds, off, cap, false);
        lds_relocate(P.bo, 6 * O, lm_lds, off, cap, false);
        lds_relocate(P.xo, 6 * O, lm_lds, off, cap, false);
        lds_relocate(P.Hcc, 36 * C, lm_lds, off, cap, false);
        lds_relocate(P.bc, 6 * C, lm_lds, off, cap, false);
        lds_relocate(P.xc, 6 * C, lm_lds, off, cap, false);
        lds_relocate(P.yc, 6 * C, lm_lds, off, cap, false);
        lds_relocate(P.pair_cam, NP, lm_lds, off, cap, true);
        lds_relocate(P.pair_obj, NP, lm_lds, off, cap, true);
        lds_relocate(P.pair_start, NP + 1, lm_lds, off, cap, true);
        lds_relocate(P.cam_pair_ptr, C + 1, lm_lds, off, cap, true);
        lds_relocate(P.cam_pair_idx, NP, lm_lds, off, cap, true);
        lds_relocate(P.obj_pair_ptr, O + 1, lm_lds, off, cap, true);
        lds_relocate(P.obj_pair_idx, NP, lm_lds, off, cap, true);
        lds_relocate(P.edge_pair, E, lm_lds, off, cap, true);
        lds_relocate(P.level, E, lm_lds, off, cap, false);
        lds_relocate(P.edge_k, 4 * E, lm_lds, off, cap, true);
        lds_relocate(P.edge_p, 3 * E, lm_lds, off, cap, true);
        lds_relocate(P.edge_uv, 2 * E, lm_lds, off, cap, true);
        lds_relocate(P.edge_info, 3 * E, lm_lds, off, cap, true);
        lds_relocate(P.err, 2 * E, lm_lds, off, cap, false);
        lds_relocate(P.pair_part, 90 * NP, lm_lds, off, cap, false);
        lds_relocate(P.jac, 29 * E, lm_lds, off, cap, false);
        lds_relocate(P.Hcc_inv, 36 * C, lm_lds, off, cap, false);
        lds_relocate(P.Y, 36 * NP, lm_lds, off, cap, false);
    }
    __syncthreads();

    // ---- load poses ---------------------------------------------------------------------------
    for (int c = tid; c < P.n_cam; c += LM_THREADS) pose_from_T(P.cam_T + 12 * c, P.cam[c]);
    for (int o = tid; o < P.n_obj; o += LM_THREADS) pose_from_T(P.obj_T + 12 * o, P.obj[o]);
    __syncthreads();

    // ---- initial classification (object_slam.py:848-866) --------------------------------------
    for (int e = tid; e < P.n_edge; e += LM_THREADS) P.level[e] = 0;
    __syncthreads();
    int my_good = 0;
    if (P.init_with_outliers) {
        my_good = 0;
        if (tid == 0) sh_good = P.n_edge;
    } else {
        for (int e = tid; e < P.n_edge; e += LM_THREADS) {
            double er[2];
            edge_error(P, e, er, nullptr, nullptr);
            const double c2 = edge_chi2(P, e, er);
            P.edge_chi2[e] = c2;
            if (c2 > P.chi2_thr) { P.level[e] = 1; P.edge_inlier[e] = 0; }
            else { P.level[e] = 0; P.edge_inlier[e] = 1; ++my_good; }
        }
        const double g = block_sum((double)my_good, red);
        if (tid == 0) sh_good = (int)g;
    }
    __syncthreads();
    int num_good = sh_good;
    bool robust_on = true;
    int rounds = 0, lm_its = 0, lm_trials = 0;
    const int drop = (P.n_rounds / 2) > 1 ? (P.n_rounds / 2) : 1;

    for (int round = 0; round < P.n_rounds; ++round) {
        if (P.n_edge < 4 || num_good < 4) break;
        ++rounds;
        // any active edge at all?  (g2o: 0 vertices to optimise -> optimize() returns without iterating)
        double nact = 0;
        for (int e = tid; e < P.n_edge; e += LM_THREADS) nact += edge_active(P, e) ? 1.0 : 0.0;
        nact = block_sum(nact, red);
        const int iterations = nact > 0 ? P.its[round] : 0;
        double lambda = -1, ni = 2;
        for (int it = 0; it < iterations; ++it) {
            // ---- errors, chi2, linearisation ---------------------------------------------------
            double currentChi = active_errors_and_chi2(P, robust_on, true, red);
            accumulate_pairs(P);
            __syncthreads();
            // ---- gather the diagonal blocks (fixed summation order) ----------------------------
            for (int idx = tid; idx < P.n_cam * 27; idx += LM_THREADS) {
                const int c = idx / 27, k = idx - c * 27;
                if (P.cam_fixed[c]) continue;
                double s = 0;
                for (int j = P.cam_pair_ptr[c]; j < P.cam_pair_ptr[c + 1]; ++j)
                    s += P.pair_part[90 * (size_t)P.cam_pair_idx[j] + (k < 21 ? k : 78 + (k - 21))];
                if (k < 21) P.Hcc[36 * c + k] = s; else P.bc[6 * c + (k - 21)] = s;     // Hcc packed upper (21) for now
            }
            for (int idx = tid; idx < P.n_obj * 27; idx += LM_THREADS) {
                const int o = idx / 27, k = idx - o * 27;
                if (P.obj_fixed[o]) continue;
                double s = 0;
                for (int j = P.obj_pair_ptr[o]; j < P.obj_pair_ptr[o + 1]; ++j)
                    s += P.pair_part[90 * (size_t)P.obj_pair_idx[j] + (k < 21 ? 21 + k : 84 + (k - 21))];
                if (k < 21) P.Hoo[36 * o + k] = s; else P.bo[6 * o + (k - 21)] = s;
            }
            __syncthreads();
            if (it == 0) {      // computeLambdaInit: tau * max |diag|
                double md = 0;
                const int diag21[6] = {0, 6, 11, 15, 18, 20};
                for (int idx = tid; idx < (P.n_cam + P.n_obj) * 6; idx += LM_THREADS) {
                    const int v = idx / 6, d = idx - v * 6;
                    if (v < P.n_cam) { if (!P.cam_fixed[v]) md = fmax(md, fabs(P.Hcc[36 * v + diag21[d]])); }
                    else { const int o = v - P.n_cam; if (!P.obj_fixed[o]) md = fmax(md, fabs(P.Hoo[36 * o + diag21[d]])); }
                }
                md = block_max(md, red);
                lambda = 1e-5 * md;
                ni = 2;
            }
            // ---- trials ----------------------------------------------------------------------
            double rho = 0;
            int qmax = 0;
            bool lam_finite = true;
            do {
                // push()
                for (int c = tid; c < P.n_cam; c += LM_THREADS) P.cam_bak[c] = P.cam[c];
                for (int o = tid; o < P.n_obj; o += LM_THREADS) P.obj_bak[o] = P.obj[o];
                if (tid == 0) sh_ok = 1;
                __syncthreads();
                // cameras: (Hcc + lambda I)^-1 and y_c = Hcc^-1 b_c (the inverse is only needed for the Schur complement)
                for (int c = tid; c < P.n_cam; c += LM_THREADS) {
                    if (P.cam_fixed[c]) continue;
                    double A[36];
                    unpack_sym21(P.Hcc + 36 * c, A);
                    for (int d = 0; d < 6; ++d) A[d * 7] += lambda;
                    if (schur) {
                        double Ai[36];
                        if (!spd_inverse6(A, Ai)) { sh_ok = 0; for (int i = 0; i < 36; ++i) Ai[i] = 0; }
                        for (int i = 0; i < 36; ++i) P.Hcc_inv[36 * c + i] = Ai[i];
                        for (int r = 0; r < 6; ++r) {
                            double sacc = 0;
                            for (int k = 0; k < 6; ++k) sacc += Ai[r * 6 + k] * P.bc[6 * c + k];
                            P.yc[6 * c + r] = sacc;
                        }
                    } else {
                        double x[6] = {0, 0, 0, 0, 0, 0};
                        if (!spd_solve6(A, P.bc + 6 * c, x)) sh_ok = 0;
                        for (int r = 0; r < 6; ++r) P.yc[6 * c + r] = x[r];
                    }
                }
                __syncthreads();
                if (!schur) {
                    // block-diagonal: one thread per free object / camera
                    for (int o = tid; o < P.n_obj; o += LM_THREADS) {
                        if (P.obj_fixed[o]) continue;
                        double A[36], x[6] = {0, 0, 0, 0, 0, 0};
                        unpack_sym21(P.Hoo + 36 * o, A);
                        for (int d = 0; d < 6; ++d) A[d * 7] += lambda;
                        if (!spd_solve6(A, P.bo + 6 * o, x)) sh_ok = 0;
                        for (int r = 0; r < 6; ++r) P.xo[6 * o + r] = x[r];
                    }
                    for (int idx = tid; idx < P.n_cam * 6; idx += LM_THREADS) P.xc[idx] = P.cam_fixed[idx / 6] ? 0.0 : P.yc[idx];
                    __syncthreads();
                } else {
                    // Y[p] = Hcc^-1 Hco[p]
                    for (int idx = tid; idx < P.n_pair * 36; idx += LM_THREADS) {
                        const int p = idx / 36, rc = idx - p * 36, r = rc / 6, cc = rc - r * 6;
                        const int c = P.pair_cam[p];
                        double s = 0;
                        if (!P.cam_fixed[c] && !P.obj_fixed[P.pair_obj[p]]) {
                            const double* Hco = P.pair_part + 90 * (size_t)p + 42;
                            for (int k = 0; k < 6; ++k) s += P.Hcc_inv[36 * c + r * 6 + k] * Hco[k * 6 + cc];
                        }
                        P.Y[idx] = s;
                    }
                    // S = blockdiag(Hoo + lambda I);  rhs = b_o
                    for (int idx = tid; idx < ns * ns; idx += LM_THREADS) S[idx] = 0;
                    __syncthreads();
                    for (int idx = tid; idx < P.n_obj * 36; idx += LM_THREADS) {
                        const int o = idx / 36, rc = idx - o * 36, r = rc / 6, cc = rc - r * 6;
                        const int so = P.obj_slot[o];
                        if (so < 0) continue;
                        const int rr = r < cc ? r : cc, c2 = r < cc ? cc : r;
                        const int packed = rr * 6 - rr * (rr - 1) / 2 + (c2 - rr);
                        S[(6 * so + r) * ns + 6 * so + cc] = P.Hoo[36 * o + packed] + (r == cc ? lambda : 0.0);
                    }
                    for (int idx = tid; idx < P.n_obj * 6; idx += LM_THREADS) {
                        const int o = idx / 6;
                        if (P.obj_slot[o] >= 0) rhs[6 * P.obj_slot[o] + (idx - o * 6)] = P.bo[idx];
                    }
                    __syncthreads();
                    // S -= sum_c Hco(c,o1)^T Y(c,o2);  rhs -= sum_c Hco(c,o)^T y_c    (fixed camera order)
                    for (int idx = tid; idx < ns * ns; idx += LM_THREADS) {
                        const int row = idx / ns, col = idx - row * ns;
                        const int s1 = row / 6, i = row - s1 * 6, s2 = col / 6, j = col - s2 * 6;
                        double acc = 0;
                        // walk the pairs of object slot s1 (CSR by object), find the same camera's pair with slot s2
                        int o1 = -1, o2 = -1;
                        for (int o = 0; o < P.n_obj; ++o) { if (P.obj_slot[o] == s1) o1 = o; if (P.obj_slot[o] == s2) o2 = o; }
                        for (int a = P.obj_pair_ptr[o1]; a < P.obj_pair_ptr[o1 + 1]; ++a) {
                            const int p1 = P.obj_pair_idx[a], c = P.pair_cam[p1];
                            if (P.cam_fixed[c]) continue;
                            int p2 = -1;
                            for (int b = P.cam_pair_ptr[c]; b < P.cam_pair_ptr[c + 1]; ++b)
                                if (P.pair_obj[P.cam_pair_idx[b]] == o2) { p2 = P.cam_pair_idx[b]; break; }
                            if (p2 < 0) continue;
                            const double* H1 = P.pair_part + 90 * (size_t)p1 + 42;   // Hco(c,o1) [6x6], row = cam dof
                            const double* Y2 = P.Y + 36 * (size_t)p2;
                            for (int k = 0; k < 6; ++k) acc += H1[k * 6 + i] * Y2[k * 6 + j];
                        }
                        S[idx] -= acc;
                    }
                    for (int row = tid; row < ns; row += LM_THREADS) {
                        const int s1 = row / 6, i = row - s1 * 6;
                        int o1 = -1;
                        for (int o = 0; o < P.n_obj; ++o) if (P.obj_slot[o] == s1) o1 = o;
                        double acc = 0;
                        for (int a = P.obj_pair_ptr[o1]; a < P.obj_pair_ptr[o1 + 1]; ++a) {
                            const int p1 = P.obj_pair_idx[a], c = P.pair_cam[p1];
                            if (P.cam_fixed[c]) continue;
                            const double* H1 = P.pair_part + 90 * (size_t)p1 + 42;
                            for (int k = 0; k < 6; ++k) acc += H1[k * 6 + i] * P.yc[6 * c + k];
                        }
                        rhs[row] -= acc;
                    }
                    __syncthreads();
                    // workgroup Cholesky S = L L^T (lower, in place), then forward / backward substitution
                    for (int j = 0; j < ns; ++j) {
                        for (int i = j + tid; i < ns; i += LM_THREADS) {
                            double s = S[i * ns + j];
                            for (int k = 0; k < j; ++k) s -= S[i * ns + k] * S[j * ns + k];
                            colbuf[i] = s;
                        }
                        __syncthreads();
                        const double piv = colbuf[j];
                        if (!(piv > 0) || !isfinite(piv)) { if (tid == 0) sh_ok = 0; }
                        const double d = sqrt(piv > 0 ? piv : 1.0);
                        for (int i = j + tid; i < ns; i += LM_THREADS) S[i * ns + j] = (i == j) ? d : colbuf[i] / d;
                        __syncthreads();
                    }
                    for (int j = 0; j < ns; ++j) {          // L y = rhs
                        if (tid == 0) rhs[j] = rhs[j] / S[j * ns + j];
                        __syncthreads();
                        const double yj = rhs[j];
                        for (int i = j + 1 + tid; i < ns; i += LM_THREADS) rhs[i] -= S[i * ns + j] * yj;
                        __syncthreads();
                    }
                    for (int j = ns - 1; j >= 0; --j) {     // L^T x = y
                        if (tid == 0) rhs[j] = rhs[j] / S[j * ns + j];
                        __syncthreads();
                        const double xj = rhs[j];
                        for (int i = tid; i < j; i += LM_THREADS) rhs[i] -= S[j * ns + i] * xj;
                        __syncthreads();
                    }
                    for (int idx = tid; idx < P.n_obj * 6; idx += LM_THREADS) {
                        const int o = idx / 6;
                        P.xo[idx] = P.obj_slot[o] >= 0 ? rhs[6 * P.obj_slot[o] + (idx - o * 6)] : 0.0;
                    }
                    __syncthreads();
                    // x_c = y_c - sum_o Y(c,o) x_o
                    for (int idx = tid; idx < P.n_cam * 6; idx += LM_THREADS) {
                        const int c = idx / 6, r = idx - c * 6;
                        double s = 0;
                        if (!P.cam_fixed[c]) {
                            s = P.yc[idx];
                            for (int b = P.cam_pair_ptr[c]; b < P.cam_pair_ptr[c + 1]; ++b) {
                                const int p = P.cam_pair_idx[b], o = P.pair_obj[p];
                                if (P.obj_fixed[o]) continue;
                                for (int k = 0; k < 6; ++k) s -= P.Y[36 * (size_t)p + r * 6 + k] * P.xo[6 * o + k];
                            }
                        }
                        P.xc[idx] = s;
                    }
                    __syncthreads();
                }
                const bool ok2 = sh_ok != 0;
                // update(x)
                if (ok2) {
                    for (int c = tid; c < P.n_cam; c += LM_THREADS) if (!P.cam_fixed[c]) pose_oplus(P.cam[c], P.xc + 6 * c);
                    for (int o = tid; o < P.n_obj; o += LM_THREADS) if (!P.obj_fixed[o]) pose_oplus(P.obj[o], P.xo + 6 * o);
                }
                __syncthreads();
                double tempChi = active_errors_and_chi2(P, robust_on, false, red);
                if (!ok2) tempChi = 1.7976931348623157e308;
                // computeScale: sum x (lambda x + b)
                double sc = 0;
                if (ok2) {
                    for (int idx = tid; idx < P.n_cam * 6; idx += LM_THREADS)
                        if (!P.cam_fixed[idx / 6]) sc += P.xc[idx] * (lambda * P.xc[idx] + P.bc[idx]);
                    for (int idx = tid; idx < P.n_obj * 6; idx += LM_THREADS)
                        if (!P.obj_fixed[idx / 6]) sc += P.xo[idx] * (lambda * P.xo[idx] + P.bo[idx]);
                }
                sc = block_sum(sc, red);
                rho = (currentChi - tempChi) / (sc + 1e-3);
                if (rho > 0 && isfinite(tempChi)) {
                    double alpha = 1. - pow(2 * rho - 1, 3.0);
                    alpha = fmin(alpha, 2. / 3.);
                    lambda *= fmax(1. / 3., alpha);
                    ni = 2;
                    currentChi = tempChi;
                } else {
                    lambda *= ni;
                    ni *= 2;
                    __syncthreads();
                    for (int c = tid; c < P.n_cam; c += LM_THREADS) P.cam[c] = P.cam_bak[c];     // pop()
                    for (int o = tid; o < P.n_obj; o += LM_THREADS) P.obj[o] = P.obj_bak[o];
                    __syncthreads();
                    if (!isfinite(lambda)) { lam_finite = false; break; }
                }
                ++qmax;
                ++lm_trials;
            } while (rho < 0 && qmax < 10);
            ++lm_its;
            if (qmax == 10 || rho == 0 || !lam_finite) break;      // Terminate
        }
        // ---- re-classification (object_slam.py:877-896), chi2 at the accepted state -----------
        __syncthreads();
        my_good = 0;
        for (int e = tid; e < P.n_edge; e += LM_THREADS) {
            double er[2];
            edge_error(P, e, er, nullptr, nullptr);
            const double c2 = edge_chi2(P, e, er);
            P.edge_chi2[e] = c2;
            if (c2 > P.chi2_thr) { P.level[e] = 1; P.edge_inlier[e] = 0; }
            else { P.level[e] = 0; P.edge_inlier[e] = 1; ++my_good; }
        }
        num_good = (int)block_sum((double)my_good, red);
        if (round == drop) robust_on = false;
        __syncthreads();
    }
    for (int c = tid; c < P.n_cam; c += LM_THREADS) pose_to_T(P.cam[c], P.cam_T + 12 * c);
    for (int o = tid; o < P.n_obj; o += LM_THREADS) pose_to_T(P.obj[o], P.obj_T + 12 * o);
    if (tid == 0) { P.stats[0] = rounds; P.stats[1] = lm_its; P.stats[2] = lm_trials; P.stats[3] = num_good; }
}

// Dynamic LDS a problem of this size wants (everything resident), capped at LM_LDS_BYTES.  Small problems ask
// for little, so their workgroup can share a CU with the CNN's workgroups instead of waiting for an empty one.
int lm_lds_bytes(int C, int O, int E, int NP, int n_free_obj_schur) {
    size_t b = 0;
    const size_t ns = 6 * (size_t)n_free_obj_schur;
    b += 8 * (ns * ns + 2 * ns) + 48;
    b += 4 * 56 * (size_t)(C + O) / 2 * 2 + 2 * (size_t)(C + O) + 4 * (size_t)O;
    b += 8 * 48 * (size_t)O + 8 * 54 * (size_t)C;
    b += 4 * (3 * (size_t)NP + 1) + 4 * ((size_t)C + 1 + NP) + 4 * ((size_t)O + 1 + NP);
    b += (4 + 1 + 8 * 12 + 16) * (size_t)E + 720 * (size_t)NP + 232 * (size_t)E + 288 * (size_t)C + 288 * (size_t)NP;
    b += 16 * 40;                                    // per-array alignment slack
    b = (b + 1023) & ~(size_t)1023;
    return (int)(b > (size_t)LM_LDS_BYTES ? (size_t)LM_LDS_BYTES : b);
}

int launch_lm(const void* problems_dev, int n_problems, int lds_bytes, hipStream_t s) {
    if (n_problems <= 0) return SUO_OK;
    if (lds_bytes <= 0 || lds_bytes > LM_LDS_BYTES) lds_bytes = LM_LDS_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        SUO_HIP_CHECK(hipFuncSetAttribute((const void*)lm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LM_LDS_BYTES));
        attr_set = true;
    }
    hipLaunchKernelGGL(lm_kernel, dim3(n_problems), dim3(LM_THREADS), lds_bytes, s, (const LmProblem*)problems_dev, lds_bytes);
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}

size_t lm_problem_struct_size() { return sizeof(LmProblem); }

}  // namespace suo
