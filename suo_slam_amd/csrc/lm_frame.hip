// Single-view pose refinement as ONE WAVE PER OBJECT: the LM algorithm of csrc/lm.hip specialised to "no free camera".
//
// evaluate.py's single-view mode (BASELINE configs[1], [3], [4]) calls ObjectSLAM.optimize() once per frame with the camera
// fixed at identity and every object free (/root/reference/lib/object_slam.py:746-778, its = [10,10,40,40]).  The Hessian
// is then block diagonal -- one 6x6 block per object -- and the objects are coupled only through g2o's scalars:
// ONE lambda, ONE gain ratio, ONE chi2 for the whole graph (optimization_algorithm_levenberg.cpp:58-150).  The general kernel
// runs such a frame as ~60 trials x ~28 us of workgroup-wide phases (pair blocks, gathers, block solves) separated by
// barriers, with the working set in LDS: 1.8 ms per frame, longer than the network's share of a frame at 8 crops.  Here
//   * wave w of the workgroup owns object w: its pose lives in registers (identical in all 64 lanes), every lane owns one
//     edge (keypoint) of the object -- constants in registers -- and evaluates error, Jacobian and Huber weight itself;
//   * J^T W J (21) and J^T W r (6) are summed by a shuffle tree, every lane solves the 6x6 system identically and applies
//     the update to its own copy of the pose: no broadcast, no LDS, no barrier inside an object;
//   * per trial the waves exchange three numbers each (chi2, step scale, ok) through LDS slots with ONE workgroup barrier,
//     sum them in object order and take the same accept / reject / lambda decision.
// Same rounds / robust-kernel schedule / lambda schedule / re-classification as csrc/lm.hip; the summation order of H, b
// and chi2 differs (rounding level).  Frames run as independent workgroups.
#include <type_traits>

#include "lm_device.h"

namespace suo {

constexpr int LF_MAX_OBJ = 16;                  // waves per workgroup (1024 threads)

struct LfEdge {                                 // one edge's constants, camera pose folded in
    double k[4], p[3], uv[2], info[3];
    double Rc[9], tc[3];
};

DEV void lf_load_edge(const LmProblem& P, int e, LfEdge& E) {
    for (int i = 0; i < 4; ++i) E.k[i] = P.edge_k[4 * e + i];
    for (int i = 0; i < 3; ++i) { E.p[i] = P.edge_p[3 * e + i]; E.info[i] = P.edge_info[3 * e + i]; }
    E.uv[0] = P.edge_uv[2 * e]; E.uv[1] = P.edge_uv[2 * e + 1];
    const int c = P.pair_cam[P.edge_pair[e]];
    Pose cam;
    pose_from_T(P.cam_T + 12 * c, cam);        // the same quaternion round trip as the general kernel's pose table
    q_to_R(cam.q, E.Rc);
    for (int i = 0; i < 3; ++i) E.tc[i] = cam.t[i];
}

// error (EdgeSE3ProjectFromObject::computeError, types_object_slam.cpp:45-60) and, if asked, the 2x6 object Jacobian
// (linearizeOplus, :70-123) of one edge at object pose (Ro, to); same formulas as edge_pass_partial (csrc/lm_device.h)
DEV void lf_edge(const LfEdge& E, const double* Ro, const double* to, double* er, double* Jo) {
    double pw[3], pc[3];
    for (int r = 0; r < 3; ++r) pw[r] = Ro[3 * r] * E.p[0] + Ro[3 * r + 1] * E.p[1] + Ro[3 * r + 2] * E.p[2] + to[r];
    for (int r = 0; r < 3; ++r) pc[r] = E.Rc[3 * r] * pw[0] + E.Rc[3 * r + 1] * pw[1] + E.Rc[3 * r + 2] * pw[2] + E.tc[r];
    er[0] = E.uv[0] - (E.k[0] * pc[0] / pc[2] + E.k[2]);
    er[1] = E.uv[1] - (E.k[1] * pc[1] / pc[2] + E.k[3]);
    if (!Jo) return;
    const double PJ[6] = {-(E.k[0] / pc[2]), 0, E.k[0] * pc[0] / (pc[2] * pc[2]), 0, -(E.k[1] / pc[2]), E.k[1] * pc[1] / (pc[2] * pc[2])};
    double PR[6];
    for (int r = 0; r < 2; ++r)
        for (int cc = 0; cc < 3; ++cc) PR[3 * r + cc] = PJ[3 * r] * E.Rc[cc] + PJ[3 * r + 1] * E.Rc[3 + cc] + PJ[3 * r + 2] * E.Rc[6 + cc];
    const double Dw[18] = {0, pw[2], -pw[1], 1, 0, 0, -pw[2], 0, pw[0], 0, 1, 0, pw[1], -pw[0], 0, 0, 0, 1};
    for (int r = 0; r < 2; ++r)
        for (int cc = 0; cc < 6; ++cc) Jo[6 * r + cc] = PR[3 * r] * Dw[cc] + PR[3 * r + 1] * Dw[6 + cc] + PR[3 * r + 2] * Dw[12 + cc];
}
DEV double lf_chi2(const LfEdge& E, const double* er) {
    return er[0] * (E.info[0] * er[0] + E.info[1] * er[1]) + er[1] * (E.info[1] * er[0] + E.info[2] * er[1]);
}

// all-reduce of N doubles per lane (DPP + readlane sums of csrc/lm_device.h: no LDS crossbar round trips)
template <int N>
DEV void wsum_many(double* v) {
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = wsum(v[k]);
}

template <int MAXW>
__global__ __launch_bounds__(64 * MAXW) void lm_frame_kernel(const LmProblem* __restrict__ problems) {
    const LmProblem& P = problems[blockIdx.x];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_waves = blockDim.x >> 6;
    __shared__ double xchg[2][LF_MAX_OBJ][4];   // per trial parity: [object] = {chi2, scale, ok, max diag}
    const bool have = w < P.n_obj;              // waves beyond the frame's objects only keep the barriers company
    const bool free_obj = have && !P.obj_fixed[w];
    // this object's edges: the pairs of object w (one per camera that sees it; one in single-view mode), contiguous edges each.
    // Lane l owns edge number l of the object (registers); edges beyond 64 per object are re-read from memory.
    int n_own = 0, e_first = -1;
    if (have)
        for (int a = P.obj_pair_ptr[w]; a < P.obj_pair_ptr[w + 1]; ++a) {
            const int p = P.obj_pair_idx[a];
            for (int e = P.pair_start[p]; e < pair_hi(P, p); ++e) { if (n_own == lane) e_first = e; ++n_own; }
        }
    LfEdge E0;
    int lvl0 = 0;
    if (e_first >= 0) lf_load_edge(P, e_first, E0);
    // edge number j >= 64 of this object (rare: more than 64 measurements of one object)
    auto nth_edge = [&](int j) -> int {
        int n = 0;
        for (int a = P.obj_pair_ptr[w]; a < P.obj_pair_ptr[w + 1]; ++a) {
            const int p = P.obj_pair_idx[a], cnt = pair_hi(P, p) - P.pair_start[p];
            if (j < n + cnt) return P.pair_start[p] + (j - n);
            n += cnt;
        }
        return -1;
    };
    for (int j = 64 + lane; j < n_own; j += 64) P.level[nth_edge(j)] = 0;

    Pose pose;
    if (have) pose_from_T(P.obj_T + 12 * w, pose); else { pose.q[0] = 1; pose.q[1] = pose.q[2] = pose.q[3] = 0; pose.t[0] = pose.t[1] = pose.t[2] = 0; }
    int xp = 0;                                 // exchange-slot parity

    // workgroup sum (object order) of one value per wave + the same for up to three more values, one barrier
    auto exchange = [&](double a, double b, double c, double d, double* out) {
        if (lane == 0) { xchg[xp][w][0] = a; xchg[xp][w][1] = b; xchg[xp][w][2] = c; xchg[xp][w][3] = d; }
        __syncthreads();
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        for (int o = 0; o < n_waves; ++o) { s0 += xchg[xp][o][0]; s1 += xchg[xp][o][1]; s2 += xchg[xp][o][2]; s3 = fmax(s3, xchg[xp][o][3]); }
        out[0] = s0; out[1] = s1; out[2] = s2; out[3] = s3;
        xp ^= 1;                                // the next exchange writes the other slot set: no second barrier needed
    };

    // chi2 (re-)classification of the own edges (object_slam.py:855-866, 877-893); returns the wave's inlier count
    auto classify = [&](bool keep_all) -> double {
        double Ro[9], good = 0;
        q_to_R(pose.q, Ro);
        if (e_first >= 0) {
            double er[2];
            lf_edge(E0, Ro, pose.t, er, nullptr);
            const double c2 = lf_chi2(E0, er);
            P.edge_chi2[e_first] = c2;
            if (keep_all) { lvl0 = 0; good += 1; }
            else if (c2 > P.chi2_thr) { lvl0 = 1; P.edge_inlier[e_first] = 0; }
            else { lvl0 = 0; P.edge_inlier[e_first] = 1; good += 1; }
        }
        for (int j = 64 + lane; j < n_own; j += 64) {
            const int e = nth_edge(j);
            LfEdge E;
            lf_load_edge(P, e, E);
            double er[2];
            lf_edge(E, Ro, pose.t, er, nullptr);
            const double c2 = lf_chi2(E, er);
            P.edge_chi2[e] = c2;
            if (keep_all) { P.level[e] = 0; good += 1; }
            else if (c2 > P.chi2_thr) { P.level[e] = 1; P.edge_inlier[e] = 0; }
            else { P.level[e] = 0; P.edge_inlier[e] = 1; good += 1; }
        }
        return wsum(good);
    };
    // robustified chi2 of the own ACTIVE edges at pose (Ro, to); with h != nullptr also J^T W J (21, packed upper) and J^T W r (6)
    // (WITH_H as a type, h as an array reference: with a nullable pointer the 27 sums lived in scratch memory -- the lambda served both
    //  call shapes and the array escaped through it -- and every update of them was a memory round trip)
    auto edge_pass = [&](const double* Ro, const double* to, bool robust_on, auto with_h, double (&h)[27]) -> double {
        constexpr bool WITH_H = decltype(with_h)::value;
        double c = 0;
        auto one = [&](const LfEdge& E) {
            double er[2], Jo[12];
            lf_edge(E, Ro, to, er, WITH_H ? Jo : nullptr);
            const double c2 = lf_chi2(E, er);
            double wgt = 1.0;
            c += robust_on ? huber_rho(c2, P.huber_delta, wgt) : c2;
            if constexpr (WITH_H) {
                const double i0 = wgt * E.info[0], i1 = wgt * E.info[1], i2 = wgt * E.info[2];
                const double g0 = -(E.info[0] * er[0] + E.info[1] * er[1]) * wgt, g1 = -(E.info[1] * er[0] + E.info[2] * er[1]) * wgt;
                double wj0[6], wj1[6];
#pragma unroll
                for (int cc = 0; cc < 6; ++cc) { wj0[cc] = i0 * Jo[cc] + i1 * Jo[6 + cc]; wj1[cc] = i1 * Jo[cc] + i2 * Jo[6 + cc]; }
                int u = 0;
#pragma unroll
                for (int r = 0; r < 6; ++r)
#pragma unroll
                    for (int cc = r; cc < 6; ++cc) h[u++] += Jo[r] * wj0[cc] + Jo[6 + r] * wj1[cc];
#pragma unroll
                for (int r = 0; r < 6; ++r) h[21 + r] += Jo[r] * g0 + Jo[6 + r] * g1;
            }
        };
        if (free_obj) {
            if (e_first >= 0 && lvl0 == 0) one(E0);
            for (int j = 64 + lane; j < n_own; j += 64) {
                const int e = nth_edge(j);
                if (P.level[e] != 0) continue;
                LfEdge E;
                lf_load_edge(P, e, E);
                one(E);
            }
        }
        return wsum(c);
    };

    double ex[4];
    int num_good;
    {
        const double g = P.init_with_outliers ? classify(true) : classify(false);
        exchange(g, 0, 0, 0, ex);
        num_good = P.init_with_outliers ? P.n_edge : (int)ex[0];
    }
    bool robust_on = true;
    int rounds = 0, lm_its = 0, lm_trials = 0;
    const int drop = (P.n_rounds / 2) > 1 ? (P.n_rounds / 2) : 1;

    for (int round = 0; round < P.n_rounds; ++round) {
        if (P.n_edge < 4 || num_good < 4) break;
        ++rounds;
        double nact = 0;                                         // any active edge at all? (g2o: nothing to optimise -> no iterations)
        if (free_obj) {
            if (e_first >= 0 && lvl0 == 0) nact += 1;
            for (int j = 64 + lane; j < n_own; j += 64) nact += P.level[nth_edge(j)] == 0 ? 1.0 : 0.0;
        }
        exchange(wsum(nact), 0, 0, 0, ex);
        const int iterations = ex[0] > 0 ? P.its[round] : 0;
        double lambda = -1, ni = 2;
        for (int it = 0; it < iterations; ++it) {
            // ---- errors, chi2, the object's 6x6 system ------------------------------------------------------
            double Ro[9], h[27];
            q_to_R(pose.q, Ro);
#pragma unroll
            for (int k = 0; k < 27; ++k) h[k] = 0;
            const double chi_o = edge_pass(Ro, pose.t, robust_on, std::true_type{}, h);
            wsum_many<27>(h);
            double md = 0;
            if (free_obj) {
                const int diag21[6] = {0, 6, 11, 15, 18, 20};
#pragma unroll
                for (int d = 0; d < 6; ++d) md = fmax(md, fabs(h[diag21[d]]));
            }
            exchange(chi_o, 0, 0, md, ex);
            double currentChi = ex[0];
            if (it == 0) { lambda = 1e-5 * ex[3]; ni = 2; }      // computeLambdaInit: tau * max |diag H| over all free vertices
            // ---- trials ---------------------------------------------------------------------------------------
            double rho = 0;
            int qmax = 0;
            bool lam_finite = true;
            do {
                double A[36], b6[6], x[6] = {0, 0, 0, 0, 0, 0};
                {
                    int u = 0;
#pragma unroll
                    for (int r = 0; r < 6; ++r)
#pragma unroll
                        for (int c = r; c < 6; ++c) { A[r * 6 + c] = h[u]; A[c * 6 + r] = h[u]; ++u; }
#pragma unroll
                    for (int d = 0; d < 6; ++d) { A[d * 7] += lambda; b6[d] = h[21 + d]; }
                }
                bool ok_o = true;
                Pose trial = pose;
                double sc_o = 0;
                if (free_obj) {
                    ok_o = spd_solve6(A, b6, x);                  // every lane, identically
                    if (ok_o) {
                        pose_oplus(trial, x);
                        for (int d = 0; d < 6; ++d) sc_o += x[d] * (lambda * x[d] + h[21 + d]);      // computeScale: sum x (lambda x + b)
                    }
                }
                double Rt[9];
                q_to_R(trial.q, Rt);
                // (a failed block anywhere rejects the whole trial: the chi2 evaluated here is then discarded)
                const double temp_o = edge_pass(Rt, trial.t, robust_on, std::false_type{}, h);
                exchange(temp_o, sc_o, ok_o ? 0.0 : 1.0, 0, ex);
                const bool ok2 = ex[2] == 0.0;
                const double tempChi = ok2 ? ex[0] : 1.7976931348623157e308;
                const double sc = ok2 ? ex[1] : 0.0;
                rho = (currentChi - tempChi) / (sc + 1e-3);
                if (rho > 0 && isfinite(tempChi)) {
                    double alpha = 1. - pow(2 * rho - 1, 3.0);
                    alpha = fmin(alpha, 2. / 3.);
                    lambda *= fmax(1. / 3., alpha);
                    ni = 2;
                    currentChi = tempChi;
                    pose = trial;                                 // update(x) is kept
                } else {
                    lambda *= ni;
                    ni *= 2;                                      // pop(): the trial pose is simply dropped
                    if (!isfinite(lambda)) { lam_finite = false; break; }
                }
                ++qmax;
                ++lm_trials;
            } while (rho < 0 && qmax < 10);
            ++lm_its;
            if (qmax == 10 || rho == 0 || !lam_finite) break;    // Terminate
        }
        // ---- re-classification (object_slam.py:877-896), chi2 at the accepted state ---------------------------
        exchange(classify(false), 0, 0, 0, ex);
        num_good = (int)ex[0];
        if (round == drop) robust_on = false;
    }
    if (free_obj && lane == 0) pose_to_T(pose, P.obj_T + 12 * w);      // (a fixed object keeps the bits it came with)
    if (e_first >= 0) P.level[e_first] = (uint8_t)lvl0;
    for (int c = threadIdx.x; c < P.n_cam; c += blockDim.x) {    // cameras are fixed: the same quaternion round trip as csrc/lm.hip
        Pose cam;
        pose_from_T(P.cam_T + 12 * c, cam);
        pose_to_T(cam, P.cam_T + 12 * c);
    }
    if (threadIdx.x == 0) { P.stats[0] = rounds; P.stats[1] = lm_its; P.stats[2] = lm_trials; P.stats[3] = num_good; }
}

// problems without a free camera and with at most LF_MAX_OBJ objects (the caller checks); one workgroup each, a wave per object
int launch_lm_frame(const void* problems_dev, int n_problems, int max_obj, hipStream_t s) {
    if (n_problems <= 0) return SUO_OK;
    if (max_obj < 1 || max_obj > LF_MAX_OBJ) { suo_set_error("lm_frame: %d objects per problem", max_obj); return SUO_ERR_ARG; }
    // (two builds: up to 8 waves may use 256 registers each, 9-16 waves are held to 128)
    if (max_obj <= 8) hipLaunchKernelGGL(lm_frame_kernel<8>, dim3(n_problems), dim3(64 * max_obj), 0, s, (const LmProblem*)problems_dev);
    else hipLaunchKernelGGL(lm_frame_kernel<16>, dim3(n_problems), dim3(64 * max_obj), 0, s, (const LmProblem*)problems_dev);
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}

}  // namespace suo
