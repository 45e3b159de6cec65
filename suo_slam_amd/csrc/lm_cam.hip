// Camera tracking as ONE wave: the LM algorithm of csrc/lm.hip specialised to "one free camera, every object fixed".
//
// ObjectSLAM runs this problem once per view (optimize(curr_only=True), /root/reference/lib/object_slam.py:444,703-930:
// the current camera against the mapped objects, its = [10,10,10,10]).  The unknown is a single 6-vector, so there is
// nothing to eliminate and nothing to share between workgroup waves -- yet the general kernel spends 1.6 ms on it: ~45
// trials of ~35 us, each a sequence of workgroup-wide phases (pair blocks, gathers, block solves) with barriers and LDS
// reductions between them and most of its 256 threads idle.  Here a 64-lane wave owns the problem:
//   * every lane linearises its edges (same edge_pass_partial as the general kernel) and adds their J^T W J / J^T W r
//     contributions to 27 registers; a shuffle tree leaves the 6x6 system in every lane,
//   * every lane solves it (identical instruction stream -> identical result) -- no broadcast,
//   * lane 0 applies / restores the pose; the only synchronisation is the memory fence of a one-wave workgroup barrier.
// Same rounds / robust-kernel schedule / lambda schedule / re-classification as csrc/lm.hip; only the summation order of
// H, b and chi2 differs (rounding level).  Many problems run as independent one-wave workgroups.
#include "lm_device.h"

namespace suo {

DEV double wave_sum_all(double v) { return wsum(v); }        // DPP + readlane form (csrc/lm_device.h)

__global__ __launch_bounds__(64) void lm_cam_kernel(const LmProblem* __restrict__ problems) {
    const LmProblem& P = problems[blockIdx.x];
    const int lane = threadIdx.x;
    int c0 = 0;                                                  // the free camera (the launcher checked there is exactly one)
    for (int c = 0; c < P.n_cam; ++c) if (!P.cam_fixed[c]) c0 = c;
    for (int c = lane; c < P.n_cam; c += 64) pose_from_T(P.cam_T + 12 * c, P.cam[c]);
    for (int o = lane; o < P.n_obj; o += 64) pose_from_T(P.obj_T + 12 * o, P.obj[o]);
    for (int e = lane; e < P.n_edge; e += 64) P.level[e] = 0;
    __syncthreads();                                             // one wave: no waiting, just the fence

    auto classify = [&]() -> int {                               // object_slam.py:848-866 / 877-896
        double good = 0;
        for (int e = lane; e < P.n_edge; e += 64) {
            double er[2];
            edge_error(P, e, er, nullptr, nullptr);
            const double c2 = edge_chi2(P, e, er);
            P.edge_chi2[e] = c2;
            if (c2 > P.chi2_thr) { P.level[e] = 1; P.edge_inlier[e] = 0; }
            else { P.level[e] = 0; P.edge_inlier[e] = 1; good += 1; }
        }
        return (int)wave_sum_all(good);
    };
    int num_good = P.init_with_outliers ? P.n_edge : classify();
    bool robust_on = true;
    int rounds = 0, lm_its = 0, lm_trials = 0;
    const int drop = (P.n_rounds / 2) > 1 ? (P.n_rounds / 2) : 1;

    for (int round = 0; round < P.n_rounds; ++round) {
        if (P.n_edge < 4 || num_good < 4) break;
        ++rounds;
        double nact = 0;
        for (int e = lane; e < P.n_edge; e += 64) nact += edge_active(P, e) ? 1.0 : 0.0;
        nact = wave_sum_all(nact);
        const int iterations = nact > 0 ? P.its[round] : 0;
        double lambda = -1, ni = 2;
        for (int it = 0; it < iterations; ++it) {
            // ---- errors, chi2, Jacobians; H (21, packed upper) and b (6) of the camera --------------------------
            double currentChi = wave_sum_all(edge_pass_partial(P, 0, P.n_edge, robust_on, true, lane, 64));
            double h[27];
#pragma unroll
            for (int k = 0; k < 27; ++k) h[k] = 0;
            for (int e = lane; e < P.n_edge; e += 64) {          // this lane's own Jacobians (written just above)
                if (!edge_active(P, e)) continue;
                const double* J = P.jac + 29 * (size_t)e;
                double wj0[6], wj1[6];                            // W J_c rows
#pragma unroll
                for (int c = 0; c < 6; ++c) { wj0[c] = J[24] * J[c] + J[25] * J[6 + c]; wj1[c] = J[25] * J[c] + J[26] * J[6 + c]; }
                int u = 0;
#pragma unroll
                for (int r = 0; r < 6; ++r)
#pragma unroll
                    for (int c = r; c < 6; ++c) h[u++] += J[r] * wj0[c] + J[6 + r] * wj1[c];
#pragma unroll
                for (int r = 0; r < 6; ++r) h[21 + r] += J[r] * J[27] + J[6 + r] * J[28];
            }
#pragma unroll
            for (int k = 0; k < 27; ++k) h[k] = wave_sum_all(h[k]);
            if (it == 0) {                                       // computeLambdaInit: tau * max |diag|
                const int diag21[6] = {0, 6, 11, 15, 18, 20};
                double md = 0;
#pragma unroll
                for (int d = 0; d < 6; ++d) md = fmax(md, fabs(h[diag21[d]]));
                lambda = 1e-5 * md;
                ni = 2;
            }
            // ---- trials ------------------------------------------------------------------------------------
            double rho = 0;
            int qmax = 0;
            bool lam_finite = true;
            do {
                const Pose bak = P.cam[c0];                      // push()
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
                const bool ok2 = spd_solve6(A, b6, x);           // every lane, identically
                if (ok2 && lane == 0) pose_oplus(P.cam[c0], x);
                __syncthreads();
                double tempChi = wave_sum_all(edge_pass_partial(P, 0, P.n_edge, robust_on, false, lane, 64));
                if (!ok2) tempChi = 1.7976931348623157e308;
                double sc = 0;                                    // computeScale: sum x (lambda x + b)
                if (ok2)
                    for (int d = 0; d < 6; ++d) sc += x[d] * (lambda * x[d] + h[21 + d]);
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
                    __syncthreads();                              // every lane has read the trial pose
                    if (lane == 0) P.cam[c0] = bak;               // pop()
                    __syncthreads();
                    if (!isfinite(lambda)) { lam_finite = false; break; }
                }
                ++qmax;
                ++lm_trials;
            } while (rho < 0 && qmax < 10);
            ++lm_its;
            if (qmax == 10 || rho == 0 || !lam_finite) break;    // Terminate
        }
        __syncthreads();
        num_good = classify();
        if (round == drop) robust_on = false;
        __syncthreads();
    }
    for (int c = lane; c < P.n_cam; c += 64) pose_to_T(P.cam[c], P.cam_T + 12 * c);
    for (int o = lane; o < P.n_obj; o += 64) pose_to_T(P.obj[o], P.obj_T + 12 * o);
    if (lane == 0) { P.stats[0] = rounds; P.stats[1] = lm_its; P.stats[2] = lm_trials; P.stats[3] = num_good; }
}

// problems with exactly one free camera and no free object (the caller checks), one wave each
int launch_lm_cam(const void* problems_dev, int n_problems, hipStream_t s) {
    if (n_problems <= 0) return SUO_OK;
    hipLaunchKernelGGL(lm_cam_kernel, dim3(n_problems), dim3(64), 0, s, (const LmProblem*)problems_dev);
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}

}  // namespace suo
