// Single-view pose refinement, ONE WAVE PER FRAME: the LM algorithm of csrc/lm_frame.hip (itself csrc/lm.hip specialised to "no free
// camera") with the frame's objects side by side in one wavefront.
//
// csrc/lm_frame.hip gives every object its own wave.  Measured (rocprofv3, 8 objects, ~100 LM trials): 0.64-1.4 ms per frame, i.e.
// ~23 k cycles per trial -- not latency: VALU issue.  Per trial every one of the 8 waves runs the same ~2 k instructions of uniform
// work on one object's 6x6 system (Cholesky, exp map with fp64 sin / cos / pow, quaternion round trips, the exchange through LDS and a
// workgroup barrier) plus 28 wave-wide fp64 sums of 23 instructions each for at most 41 edges, two waves to a SIMD.  Here
//   * G = 8 lanes own an object (G = 4 for frames of 9-16 objects): lane (object, sub) walks the object's edges sub, sub + G, ...;
//     edge constants sit in LDS (12 doubles per edge), the object's pose is replicated in its G lanes;
//   * J^T W J (21) + J^T W r (6) + chi2 are summed inside the group by log2(G) DPP steps -- no cross-row traffic, no v_readlane;
//   * the 6x6 solve, the exponential-map update and the quaternion round trips run ONCE per trial for all objects at the same time
//     (every group on its own system, in lock-step);
//   * g2o's per-trial scalars (chi2, step scale, solver status: one lambda / rho for the whole graph,
//     optimization_algorithm_levenberg.cpp:58-150) are summed over the objects with v_readlane in object order: no LDS exchange, no
//     barrier -- the kernel has no barrier at all.
// Same rounds / robust-kernel schedule / lambda schedule / re-classification as csrc/lm_frame.hip; sums run in another order (rounding
// level).  Takes problems with exactly one (fixed) camera, every object seen by it through at most one pair, at most 64 / G objects and
// as many edges as fit the LDS allotment: the single-view frame of evaluate.py.  Everything else stays on csrc/lm_frame.hip / lm.hip.
#include <algorithm>
#include <type_traits>

#include "lm_device.h"

namespace suo {

constexpr int LF2_EDGE_DOUBLES = 12;            // k[4], p[3], uv[2], info[3]
constexpr int LF2_MAX_EDGES = 656;              // 16 objects x 41 keypoints = 62 976 bytes of LDS

template <int G>
DEV double gsum(double v) {                     // sum over the G lanes of a group, the same value in each of them
    v += dpp_get<0xB1>(v);                      // lane ^ 1
    v += dpp_get<0x4E>(v);                      // lane ^ 2
    if (G >= 8) v += dpp_get<0x141>(v);         // row_half_mirror: the other quad of the 8
    if (G >= 16) v += dpp_get<0x140>(v);        // row_mirror: the other half row
    return v;
}
DEV double lane_value(double v, int src) {      // src wave-uniform
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}

#ifdef SUO_LF2_PROFILE
#define LF2_T(i) do { const long long _t = clock64(); pt[i] += _t - p0; p0 = _t; } while (0)
#else
#define LF2_T(i) do { } while (0)
#endif

template <int G>
DEV void lm_frame2_body(const LmProblem& P, double* Es) {
    const int lane = threadIdx.x;
#ifdef SUO_LF2_PROFILE
    long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, p0 = clock64();
#endif
    const int og = lane / G, sub = lane % G;
    const int nobj = P.n_obj;
    const bool have = og < nobj;
    const bool free_obj = have && !P.obj_fixed[og];
    // the object's edges: one pair (camera 0, object og), contiguous
    int e0 = 0, n_own = 0, lbase = 0;
    for (int o = 0; o < nobj; ++o) {
        int cnt = 0, start = 0;
        if (P.obj_pair_ptr[o + 1] > P.obj_pair_ptr[o]) {
            const int p = P.obj_pair_idx[P.obj_pair_ptr[o]];
            start = P.pair_start[p];
            cnt = pair_hi(P, p) - start;
        }
        if (o < og) lbase += cnt;
        if (o == og) { e0 = start; n_own = cnt; }
    }
    if (!have) n_own = 0;
    int epl = 0;                                 // edges per lane: the longest object decides the (uniform) trip count
    {
        int m = (n_own + G - 1) / G;
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) m = max(m, __shfl_xor(m, s, 64));
        epl = m;
    }
    for (int j = sub; j < n_own; j += G) {
        const int e = e0 + j;
        double* E = Es + (size_t)(lbase + j) * LF2_EDGE_DOUBLES;
        for (int i = 0; i < 4; ++i) E[i] = P.edge_k[4 * e + i];
        for (int i = 0; i < 3; ++i) { E[4 + i] = P.edge_p[3 * e + i]; E[9 + i] = P.edge_info[3 * e + i]; }
        E[7] = P.edge_uv[2 * e]; E[8] = P.edge_uv[2 * e + 1];
    }
    // the one fixed camera: the same quaternion round trip as the other kernels' pose tables
    double Rc[9], tc[3];
    {
        Pose cam;
        pose_from_T(P.cam_T, cam);
        q_to_R(cam.q, Rc);
        for (int i = 0; i < 3; ++i) tc[i] = cam.t[i];
    }
    Pose pose;
    if (have) pose_from_T(P.obj_T + 12 * og, pose); else { pose.q[0] = 1; pose.q[1] = pose.q[2] = pose.q[3] = 0; pose.t[0] = pose.t[1] = pose.t[2] = 0; }
    unsigned lvl = 0;                            // bit j: own edge number j (global number sub + j G) is an outlier (level 1)
    __builtin_amdgcn_wave_barrier();

    // frame-wide sum / max over the objects of a value that is uniform inside every group (0 in groups without an object): the groups
    // of a 16-lane row meet by DPP, the four row sums by v_readlane, added in row order
    auto osum = [&](double v) -> double {
        if (!have) v = 0;
        if (G <= 8) v += dpp_get<0x140>(v);                  // row_mirror: lane i <-> 15 - i, the other 8-lane group (G = 8)
        if (G <= 4) { v += dpp_get<0x141>(v); }              // row_half_mirror: the other 4-lane group of the half row (G = 4)
        return ((lane_value(v, 0) + lane_value(v, 16)) + lane_value(v, 32)) + lane_value(v, 48);
    };
    auto omax = [&](double v) -> double {
        if (!have) v = 0;
        if (G <= 8) v = fmax(v, dpp_get<0x140>(v));
        if (G <= 4) v = fmax(v, dpp_get<0x141>(v));
        return fmax(fmax(lane_value(v, 0), lane_value(v, 16)), fmax(lane_value(v, 32), lane_value(v, 48)));
    };
    // error (EdgeSE3ProjectFromObject::computeError, types_object_slam.cpp:45-60) and, if asked, the 2x6 object Jacobian (:70-123)
    auto edge = [&](const double* E, const double* Ro, const double* to, double* er, double* Jo) {
        double pw[3], pc[3];
        for (int r = 0; r < 3; ++r) pw[r] = Ro[3 * r] * E[4] + Ro[3 * r + 1] * E[5] + Ro[3 * r + 2] * E[6] + to[r];
        for (int r = 0; r < 3; ++r) pc[r] = Rc[3 * r] * pw[0] + Rc[3 * r + 1] * pw[1] + Rc[3 * r + 2] * pw[2] + tc[r];
        // (one reciprocal per edge instead of the five divisions of the formulas as g2o writes them: fp64 division is ~30 dependent
        //  instructions, and this wave has nobody to hide them behind)
        const double iz = 1.0 / pc[2];
        er[0] = E[7] - (E[0] * pc[0] * iz + E[2]);
        er[1] = E[8] - (E[1] * pc[1] * iz + E[3]);
        if (!Jo) return;
        const double PJ[6] = {-(E[0] * iz), 0, E[0] * pc[0] * iz * iz, 0, -(E[1] * iz), E[1] * pc[1] * iz * iz};
        // PJ has two structural zeros (PJ[1], PJ[3]) and D_w = [-[p_w]x | I] six plus the identity: the products by 0 and 1 of the
        // formulas as csrc/lm_frame.hip writes them are left out (the same values; only the sign of a zero can differ)
        double PR[6];
        for (int cc = 0; cc < 3; ++cc) {
            PR[cc] = PJ[0] * Rc[cc] + PJ[2] * Rc[6 + cc];
            PR[3 + cc] = PJ[4] * Rc[3 + cc] + PJ[5] * Rc[6 + cc];
        }
        for (int r = 0; r < 2; ++r) {
            const double* a = PR + 3 * r;
            Jo[6 * r + 0] = a[2] * pw[1] - a[1] * pw[2];
            Jo[6 * r + 1] = a[0] * pw[2] - a[2] * pw[0];
            Jo[6 * r + 2] = a[1] * pw[0] - a[0] * pw[1];
            Jo[6 * r + 3] = a[0]; Jo[6 * r + 4] = a[1]; Jo[6 * r + 5] = a[2];
        }
    };
    auto chi2_of = [&](const double* E, const double* er) -> double {
        return er[0] * (E[9] * er[0] + E[10] * er[1]) + er[1] * (E[10] * er[0] + E[11] * er[1]);
    };
    // chi2 (re-)classification of the own edges (object_slam.py:855-866, 877-893); returns the frame's inlier count
    auto classify = [&](bool keep_all) -> double {
        double Ro[9], good = 0;
        q_to_R(pose.q, Ro);
        for (int j = 0; j < epl; ++j) {
            const int k = sub + j * G;
            if (k < n_own) {
                const double* E = Es + (size_t)(lbase + k) * LF2_EDGE_DOUBLES;
                double er[2];
                edge(E, Ro, pose.t, er, nullptr);
                const double c2 = chi2_of(E, er);
                P.edge_chi2[e0 + k] = c2;
                if (keep_all) { lvl &= ~(1u << j); good += 1; }
                else if (c2 > P.chi2_thr) { lvl |= 1u << j; P.edge_inlier[e0 + k] = 0; }
                else { lvl &= ~(1u << j); P.edge_inlier[e0 + k] = 1; good += 1; }
            }
        }
        return osum(gsum<G>(good));
    };
    // robustified chi2 of the own ACTIVE edges at pose (Ro, to); WITH_H: also J^T W J (21, packed upper) and J^T W r (6)
    auto edge_pass = [&](const double* Ro, const double* to, bool robust_on, auto with_h, double (&h)[27]) -> double {
        constexpr bool WITH_H = decltype(with_h)::value;
        double c = 0;
        for (int j = 0; j < epl; ++j) {
            const int k = sub + j * G;
            if (free_obj && k < n_own && !((lvl >> j) & 1u)) {
                const double* E = Es + (size_t)(lbase + k) * LF2_EDGE_DOUBLES;
                double er[2], Jo[12];
                edge(E, Ro, to, er, WITH_H ? Jo : nullptr);
                const double c2 = chi2_of(E, er);
                double wgt = 1.0;
                c += robust_on ? huber_rho(c2, P.huber_delta, wgt) : c2;
                if constexpr (WITH_H) {
                    const double i0 = wgt * E[9], i1 = wgt * E[10], i2 = wgt * E[11];
                    const double g0 = -(E[9] * er[0] + E[10] * er[1]) * wgt, g1 = -(E[10] * er[0] + E[11] * er[1]) * wgt;
                    double wj0[6], wj1[6];
#pragma unroll
                    for (int cc = 0; cc < 6; ++cc) { wj0[cc] = i0 * Jo[cc] + i1 * Jo[6 + cc]; wj1[cc] = i1 * Jo[cc] + i2 * Jo[6 + cc]; }
                    int u = 0;
#pragma unroll
                    for (int r = 0; r < 6; ++r)
#pragma unroll
                        for (int cc = r; cc < 6; ++cc) { h[u] = fma(Jo[6 + r], wj1[cc], fma(Jo[r], wj0[cc], h[u])); ++u; }     // (explicit fused accumulation: half the instructions)
#pragma unroll
                    for (int r = 0; r < 6; ++r) h[21 + r] = fma(Jo[6 + r], g1, fma(Jo[r], g0, h[21 + r]));
                }
            }
        }
        return gsum<G>(c);
    };

    int num_good;
    {
        const double g = P.init_with_outliers ? classify(true) : classify(false);
        num_good = P.init_with_outliers ? P.n_edge : (int)g;
    }
    bool robust_on = true;
    int rounds = 0, lm_its = 0, lm_trials = 0;
    const int drop = (P.n_rounds / 2) > 1 ? (P.n_rounds / 2) : 1;

    for (int round = 0; round < P.n_rounds; ++round) {
        if (P.n_edge < 4 || num_good < 4) break;
        ++rounds;
        double nact = 0;                                         // any active edge at all? (g2o: nothing to optimise -> no iterations)
        for (int j = 0; j < epl; ++j) {
            const int k = sub + j * G;
            if (free_obj && k < n_own && !((lvl >> j) & 1u)) nact += 1;
        }
        const int iterations = osum(gsum<G>(nact)) > 0 ? P.its[round] : 0;
        double lambda = -1, ni = 2;
        // The linearisation of iteration it + 1 is taken at the pose iteration it accepted -- the very pose whose chi2 the accepting trial
        // has just evaluated edge by edge.  So every trial pass also accumulates J^T W J / J^T W r at ITS pose (h2): accepted, they ARE the
        // next iteration's system (same inputs, same instructions: bit-identical to linearising again) and the separate pass per
        // iteration is gone; rejected (rare: 104 trials for 100 iterations on the bench frame), they are dropped.
        double h[27], chi_o = 0;
        for (int it = 0; it < iterations; ++it) {
            // ---- errors, chi2, every object's 6x6 system --------------------------------------------------------
            if (it == 0) {
                double Ro[9];
                q_to_R(pose.q, Ro);
#pragma unroll
                for (int k = 0; k < 27; ++k) h[k] = 0;
                LF2_T(0);
                chi_o = edge_pass(Ro, pose.t, robust_on, std::true_type{}, h);
                LF2_T(1);
#pragma unroll
                for (int k = 0; k < 27; ++k) h[k] = gsum<G>(h[k]);
            }
            double currentChi = osum(chi_o);
            LF2_T(2);
            if (it == 0) {                                       // computeLambdaInit: tau * max |diag H| over all free vertices
                double md = 0;
                if (free_obj) {
                    const int diag21[6] = {0, 6, 11, 15, 18, 20};
#pragma unroll
                    for (int d = 0; d < 6; ++d) md = fmax(md, fabs(h[diag21[d]]));
                }
                lambda = 1e-5 * omax(md);
                ni = 2;
            }
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
                LF2_T(0);
                if (free_obj) {
                    ok_o = spd_solve6(A, b6, x);                  // every group its own system, in lock-step
                    LF2_T(3);
                    if (ok_o) {
                        pose_oplus(trial, x);
                        for (int d = 0; d < 6; ++d) sc_o += x[d] * (lambda * x[d] + h[21 + d]);      // computeScale: sum x (lambda x + b)
                    }
                    LF2_T(4);
                }
                double Rt[9], h2[27];
                q_to_R(trial.q, Rt);
#pragma unroll
                for (int k = 0; k < 27; ++k) h2[k] = 0;
                // (a failed block anywhere rejects the whole trial: the chi2 evaluated here is then discarded)
                const double temp_o = edge_pass(Rt, trial.t, robust_on, std::true_type{}, h2);
                LF2_T(5);
                const double s_bad = osum(ok_o ? 0.0 : 1.0), s_chi = osum(temp_o), s_sc = osum(sc_o);      // (independent: they overlap)
                const bool ok2 = s_bad == 0.0;
                const double tempChi = ok2 ? s_chi : 1.7976931348623157e308;
                const double sc = ok2 ? s_sc : 0.0;
                LF2_T(6);
                rho = (currentChi - tempChi) / (sc + 1e-3);
                if (rho > 0 && isfinite(tempChi)) {
                    const double r21 = 2 * rho - 1;
                    double alpha = 1. - r21 * r21 * r21;
                    alpha = fmin(alpha, 2. / 3.);
                    lambda *= fmax(1. / 3., alpha);
                    ni = 2;
                    currentChi = tempChi;
                    pose = trial;                                 // update(x) is kept
#pragma unroll
                    for (int k = 0; k < 27; ++k) h[k] = gsum<G>(h2[k]);      // ... and with it the system at the new pose
                    chi_o = temp_o;
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
        num_good = (int)classify(false);
        if (round == drop) robust_on = false;
    }
    if (free_obj && sub == 0) pose_to_T(pose, P.obj_T + 12 * og);      // (a fixed object keeps the bits it came with)
    for (int j = 0; j < epl; ++j) {
        const int k = sub + j * G;
        if (k < n_own) P.level[e0 + k] = (uint8_t)((lvl >> j) & 1u);
    }
    if (lane == 0) {                                                     // the fixed camera: the same quaternion round trip as csrc/lm.hip
        Pose cam;
        pose_from_T(P.cam_T, cam);
        pose_to_T(cam, P.cam_T);
        P.stats[0] = rounds; P.stats[1] = lm_its; P.stats[2] = lm_trials; P.stats[3] = num_good;
    }
#ifdef SUO_LF2_PROFILE
    LF2_T(7);
    if (blockIdx.x == 0 && lane == 0)
        printf("lm_frame2 G=%d epl=%d its=%d trials=%d cycles: misc %lld  edges+H %lld  gsum+osum %lld  solve %lld  oplus %lld  trial edges %lld  osum x3 %lld  rest %lld\n",
               G, epl, lm_its, lm_trials, pt[0], pt[1], pt[2], pt[3], pt[4], pt[5], pt[6], pt[7]);
#endif
}

// frames of <= 8 objects: 8 lanes per object; 9-16: 4.  Chosen per FRAME, so that a frame's result does not depend on what else is in
// the launch.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void lm_frame2_kernel(const LmProblem* __restrict__ problems) {
    extern __shared__ __attribute__((aligned(16))) double Es[];          // [edge][12]
    const LmProblem& P = problems[blockIdx.x];
    if (P.n_obj <= 8) lm_frame2_body<8>(P, Es);
    else lm_frame2_body<4>(P, Es);
}

int lm_frame2_max_edges() { return LF2_MAX_EDGES; }

// problems with ONE fixed camera, at most 16 objects (each seen through at most one pair) and at most LF2_MAX_EDGES edges (the caller
// checks); one wave each
int launch_lm_frame2(const void* problems_dev, int n_problems, int max_obj, int max_edges, hipStream_t s) {
    if (n_problems <= 0) return SUO_OK;
    if (max_obj < 1 || max_obj > 16 || max_edges > LF2_MAX_EDGES) { suo_set_error("lm_frame2: %d objects / %d edges per problem", max_obj, max_edges); return SUO_ERR_ARG; }
    const size_t lds = (size_t)std::max(max_edges, 1) * LF2_EDGE_DOUBLES * sizeof(double);
    hipLaunchKernelGGL(lm_frame2_kernel, dim3(n_problems), dim3(64), lds, s, (const LmProblem*)problems_dev);
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}

}  // namespace suo
