// Camera tracking (ObjectSLAM.optimize(curr_only=True), /root/reference/lib/object_slam.py:444,703-930: the current camera against the mapped
// objects, its = [10,10,10,10]) as ONE wave with everything in registers / LDS -- csrc/lm_cam.hip restructured the way csrc/lm_frame2.hip
// restructured the frame kernel.
//
// csrc/lm_cam.hip walks the general problem layout: every edge pass re-derives both rotation matrices per edge from the poses in GLOBAL memory,
// writes 29 doubles of Jacobian per edge back to global memory and reads them again for J^T W J, lane 0 alone applies the update through
// global memory between two fences, and the gain ratio goes through pow().  Measured 570 us per view (profiles/r03_slam_kernel_stats.txt),
// ~12 us per LM trial for ~120 edges.  Here:
//   * the objects are fixed, so p_w = R_o p + t_o is formed ONCE per edge (EdgeSE3ProjectFromFixedObject, types_object_slam.h:73-79) and sits
//     in LDS with the edge's intrinsics / measurement / information (12 doubles);
//   * the camera pose lives in registers, replicated in every lane; lane l owns edges l, l + 64, ...; J^T W J (21) + J^T W r (6) + chi2 are
//     wave sums (DPP + v_readlane, csrc/lm_device.h: wsum); the 6x6 solve and the exponential-map update run in every lane identically;
//   * the accepting trial's edge pass also accumulates the NEXT iteration's system (as csrc/lm_frame2.hip): one edge pass per iteration.
// Same rounds / robust-kernel schedule / lambda schedule / re-classification as csrc/lm_cam.hip; sums in another order (rounding level).
// Takes problems with exactly one camera (free), every object fixed, at most 64 * 32 edges that fit the LDS allotment; everything else stays
// on csrc/lm_cam.hip.
#include <algorithm>
#include <type_traits>

#include "lm_device.h"

namespace suo {

constexpr int LC2_EDGE_DOUBLES = 12;            // k[4], p_w[3], uv[2], info[3]
constexpr int LC2_MAX_EDGES = 1024;             // 96 KB of LDS

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void lm_cam2_kernel(const LmProblem* __restrict__ problems) {
    extern __shared__ __attribute__((aligned(16))) double Es[];          // [edge][12]
    const LmProblem& P = problems[blockIdx.x];
    const int lane = threadIdx.x;
    const int ne = P.n_edge;
    const int epl = (ne + 63) / 64;                                      // edges per lane (uniform trip count)
    // edge constants: p_w through the same quaternion round trip of the object's pose as the other kernels' pose tables
    for (int e = lane; e < ne; e += 64) {
        const int o = P.pair_obj[P.edge_pair[e]];
        Pose ob;
        pose_from_T(P.obj_T + 12 * o, ob);
        double Ro[9];
        q_to_R(ob.q, Ro);
        const double* x = P.edge_p + 3 * e;
        double* E = Es + (size_t)e * LC2_EDGE_DOUBLES;
        for (int i = 0; i < 4; ++i) E[i] = P.edge_k[4 * e + i];
        for (int r = 0; r < 3; ++r) E[4 + r] = Ro[3 * r] * x[0] + Ro[3 * r + 1] * x[1] + Ro[3 * r + 2] * x[2] + ob.t[r];
        E[7] = P.edge_uv[2 * e]; E[8] = P.edge_uv[2 * e + 1];
        for (int i = 0; i < 3; ++i) E[9 + i] = P.edge_info[3 * e + i];
    }
    Pose pose;
    pose_from_T(P.cam_T, pose);
    unsigned lvl = 0;                            // bit j: own edge lane + 64 j is an outlier (level 1)
    __builtin_amdgcn_wave_barrier();

    // error (computeError, types_object_slam.cpp:156-169) and, if asked, the 2x6 camera Jacobian (:177-201): J = -Jpi [-[p_c]x | I]
    auto edge = [&](const double* E, const double* Rc, const double* tc, double* er, double* Jc) {
        double pc[3];
        for (int r = 0; r < 3; ++r) pc[r] = Rc[3 * r] * E[4] + Rc[3 * r + 1] * E[5] + Rc[3 * r + 2] * E[6] + tc[r];
        const double iz = 1.0 / pc[2];                                   // (one reciprocal per edge: csrc/lm_frame2.hip)
        er[0] = E[7] - (E[0] * pc[0] * iz + E[2]);
        er[1] = E[8] - (E[1] * pc[1] * iz + E[3]);
        if (!Jc) return;
        const double a = -(E[0] * iz), c = E[0] * pc[0] * iz * iz, b = -(E[1] * iz), d = E[1] * pc[1] * iz * iz;
        Jc[0] = c * pc[1];              Jc[1] = a * pc[2] - c * pc[0]; Jc[2] = -(a * pc[1]); Jc[3] = a; Jc[4] = 0; Jc[5] = c;
        Jc[6] = d * pc[1] - b * pc[2];  Jc[7] = -(d * pc[0]);          Jc[8] = b * pc[0];    Jc[9] = 0; Jc[10] = b; Jc[11] = d;
    };
    auto chi2_of = [&](const double* E, const double* er) -> double {
        return er[0] * (E[9] * er[0] + E[10] * er[1]) + er[1] * (E[10] * er[0] + E[11] * er[1]);
    };
    auto classify = [&]() -> double {
        double Rc[9], good = 0;
        q_to_R(pose.q, Rc);
        for (int j = 0; j < epl; ++j) {
            const int e = lane + 64 * j;
            if (e < ne) {
                const double* E = Es + (size_t)e * LC2_EDGE_DOUBLES;
                double er[2];
                edge(E, Rc, pose.t, er, nullptr);
                const double c2 = chi2_of(E, er);
                P.edge_chi2[e] = c2;
                if (c2 > P.chi2_thr) { lvl |= 1u << j; P.edge_inlier[e] = 0; }
                else { lvl &= ~(1u << j); P.edge_inlier[e] = 1; good += 1; }
            }
        }
        return wsum(good);
    };
    // robustified chi2 of the own ACTIVE edges at pose (Rc, tc) + J^T W J (21, packed upper) and J^T W r (6); returns this lane's share
    auto edge_pass = [&](const double* Rc, const double* tc, bool robust_on, double (&h)[27]) -> double {
        double cs = 0;
        for (int j = 0; j < epl; ++j) {
            const int e = lane + 64 * j;
            if (e < ne && !((lvl >> j) & 1u)) {
                const double* E = Es + (size_t)e * LC2_EDGE_DOUBLES;
                double er[2], J[12];
                edge(E, Rc, tc, er, J);
                const double c2 = chi2_of(E, er);
                double wgt = 1.0;
                cs += robust_on ? huber_rho(c2, P.huber_delta, wgt) : c2;
                const double i0 = wgt * E[9], i1 = wgt * E[10], i2 = wgt * E[11];
                const double g0 = -(E[9] * er[0] + E[10] * er[1]) * wgt, g1 = -(E[10] * er[0] + E[11] * er[1]) * wgt;
                double wj0[6], wj1[6];
#pragma unroll
                for (int cc = 0; cc < 6; ++cc) { wj0[cc] = i0 * J[cc] + i1 * J[6 + cc]; wj1[cc] = i1 * J[cc] + i2 * J[6 + cc]; }
                int u = 0;
#pragma unroll
                for (int r = 0; r < 6; ++r)
#pragma unroll
                    for (int cc = r; cc < 6; ++cc) { h[u] = fma(J[6 + r], wj1[cc], fma(J[r], wj0[cc], h[u])); ++u; }
#pragma unroll
                for (int r = 0; r < 6; ++r) h[21 + r] = fma(J[6 + r], g1, fma(J[r], g0, h[21 + r]));
            }
        }
        return cs;
    };

    int num_good = P.init_with_outliers ? ne : (int)classify();
    bool robust_on = true;
    int rounds = 0, lm_its = 0, lm_trials = 0;
    const int drop = (P.n_rounds / 2) > 1 ? (P.n_rounds / 2) : 1;

    for (int round = 0; round < P.n_rounds; ++round) {
        if (ne < 4 || num_good < 4) break;
        ++rounds;
        double nact = 0;
        for (int j = 0; j < epl; ++j) if (lane + 64 * j < ne && !((lvl >> j) & 1u)) nact += 1;
        const int iterations = wsum(nact) > 0 ? P.its[round] : 0;
        double lambda = -1, ni = 2;
        double h[27], currentChi = 0;
        for (int it = 0; it < iterations; ++it) {
            if (it == 0) {
                double Rc[9];
                q_to_R(pose.q, Rc);
#pragma unroll
                for (int k = 0; k < 27; ++k) h[k] = 0;
                const double c = edge_pass(Rc, pose.t, robust_on, h);
#pragma unroll
                for (int k = 0; k < 27; ++k) h[k] = wsum(h[k]);
                currentChi = wsum(c);
                const int diag21[6] = {0, 6, 11, 15, 18, 20};           // computeLambdaInit: tau * max |diag H|
                double md = 0;
#pragma unroll
                for (int d = 0; d < 6; ++d) md = fmax(md, fabs(h[diag21[d]]));
                lambda = 1e-5 * md;
                ni = 2;
            }
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
                const bool ok2 = spd_solve6(A, b6, x);                   // every lane, identically
                Pose trial = pose;
                double sc = 0;
                if (ok2) {
                    pose_oplus(trial, x);
                    for (int d = 0; d < 6; ++d) sc += x[d] * (lambda * x[d] + h[21 + d]);      // computeScale: sum x (lambda x + b)
                }
                double Rt[9], h2[27];
                q_to_R(trial.q, Rt);
#pragma unroll
                for (int k = 0; k < 27; ++k) h2[k] = 0;
                double tempChi = wsum(edge_pass(Rt, trial.t, robust_on, h2));
                if (!ok2) tempChi = 1.7976931348623157e308;
                rho = (currentChi - tempChi) / (sc + 1e-3);
                if (rho > 0 && isfinite(tempChi)) {
                    const double r21 = 2 * rho - 1;
                    double alpha = 1. - r21 * r21 * r21;
                    alpha = fmin(alpha, 2. / 3.);
                    lambda *= fmax(1. / 3., alpha);
                    ni = 2;
                    currentChi = tempChi;
                    pose = trial;                                        // update(x) is kept ...
#pragma unroll
                    for (int k = 0; k < 27; ++k) h[k] = wsum(h2[k]);      // ... and with it the system at the new pose
                } else {
                    lambda *= ni;
                    ni *= 2;                                             // pop(): the trial pose is simply dropped
                    if (!isfinite(lambda)) { lam_finite = false; break; }
                }
                ++qmax;
                ++lm_trials;
            } while (rho < 0 && qmax < 10);
            ++lm_its;
            if (qmax == 10 || rho == 0 || !lam_finite) break;            // Terminate
        }
        num_good = (int)classify();
        if (round == drop) robust_on = false;
    }
    if (lane == 0) pose_to_T(pose, P.cam_T);
    for (int o = lane; o < P.n_obj; o += 64) {                           // fixed objects: the same quaternion round trip as csrc/lm_cam.hip
        Pose ob;
        pose_from_T(P.obj_T + 12 * o, ob);
        pose_to_T(ob, P.obj_T + 12 * o);
    }
    for (int j = 0; j < epl; ++j) {
        const int e = lane + 64 * j;
        if (e < ne) P.level[e] = (uint8_t)((lvl >> j) & 1u);
    }
    if (lane == 0) { P.stats[0] = rounds; P.stats[1] = lm_its; P.stats[2] = lm_trials; P.stats[3] = num_good; }
}

int lm_cam2_max_edges() { return LC2_MAX_EDGES; }

// problems with ONE camera (free), every object fixed and at most LC2_MAX_EDGES edges (the caller checks); one wave each
int launch_lm_cam2(const void* problems_dev, int n_problems, int max_edges, hipStream_t s) {
    if (n_problems <= 0) return SUO_OK;
    if (max_edges > LC2_MAX_EDGES) { suo_set_error("lm_cam2: %d edges per problem", max_edges); return SUO_ERR_ARG; }
    const size_t lds = (size_t)std::max(max_edges, 1) * LC2_EDGE_DOUBLES * sizeof(double);
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024) {
        (void)hipFuncSetAttribute((const void*)lm_cam2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LC2_MAX_EDGES * LC2_EDGE_DOUBLES * (int)sizeof(double));
        attr_set = true;
    }
    hipLaunchKernelGGL(lm_cam2_kernel, dim3(n_problems), dim3(64), lds, s, (const LmProblem*)problems_dev);
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}

}  // namespace suo
