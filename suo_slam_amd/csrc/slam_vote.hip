// SLAM tracking, between the two network passes of a view (round 6): the camera-hypothesis vote of __estimate_camera_pose
// (/root/reference/lib/object_slam.py:975-1072) and the projection of the symmetric objects' prior keypoints (:486-514) ON THE DEVICE, behind pass A's PnP
// (csrc/frame_geom.hip, do_lm = 0) and in front of pass B's network call -- so that pass B is enqueued without a host round trip.  One workgroup:
//   1. pass A's detections as scoring rows in LDS: the valid keypoints of every crop in mask order (the reference's boolean indexing), widened to double
//   2. hypotheses  T_GtoC[i] = T_pnp[i] @ inv(T_OtoG[i])  for every crop i whose PnP pose was accepted and whose object is in the map (:994-997)
//   3. counts[i] = sum over those crops j of  #{k: z > 0, chi2_k <= chi2_max}  under  T_GtoC[i] @ float32(T_OtoG[j])  (:1000-1066; the float32 container of :1004;
//      the per-keypoint expression is slam_score_kernel's, csrc/slam_score.hip -- every keypoint of a detection counts: a fresh detection's inlier flags are all set)
//   4. the first hypothesis with the most inliers, if it has at least min_inliers (:1067-1071)
//   5. for every crop s of pass B whose object is in the map: uv = project(K_bbox[s], T_GtoC @ T_OtoG[s], model keypoints) where ALL depths are positive (:497-512)
//      -> prior_uv [n_b][41][2] float32 / prior_mask [n_b][41], what suo_net_forward_prior_kp renders the prior heat-maps from
// The pose algebra the host route leaves to numpy -- the 4 x 4 products, (-R^T) t, points @ R^T, points @ K^T -- is formed as numpy's BLAS forms it: an FMA chain
// in ascending k whose first term is a plain product, fma(a3, b3, fma(a2, b2, fma(a1, b1, a0 b0))) (checked against numpy 2.2 / OpenBLAS on 300 random 4 x 4 pairs,
// stacked products, 400 transposed matrix-vector products and n x 3 @ 3 x 3 products: bit for bit); the per-keypoint scoring expression is slam_score_kernel's
// plain-sum one.  So the chain's votes, camera poses and priors equal the host route's (tests/test_gpu_slam_chain.py).
#include <math.h>

#include "../../include/suo_hip.h"
#include "suo_internal.h"

namespace suo {

constexpr int SV_KP = 41, SV_MAX = 16;
constexpr int SV_ROW = SV_KP * 9 + 9;                            // pts [41][3] | uv [41][2] | cov [41][4] | K [9]
// host block (doubles), staged by the caller: a_in_map [16] | a_T [16][12] | a_K [16][9] | b_in_map [16] | b_T [16][12] | b_K [16][9]
constexpr int HB_A_IN = 0, HB_A_T = 16, HB_A_K = HB_A_T + 16 * 12, HB_B_IN = HB_A_K + 16 * 9, HB_B_T = HB_B_IN + 16, HB_B_K = HB_B_T + 16 * 12, HB_SIZE = HB_B_K + 16 * 9;
static_assert(HB_SIZE == SUO_SLAM_VOTE_BLOCK, "include/suo_hip.h");

struct SlamVoteArgs {
    int n_a, n_b;
    const double* T_pnp; const uint8_t* accepted; const int* n_kp;                       // pass A's chain results (device)
    const float* uv; const float* cov; const uint8_t* mask; const float* kps_a;          // pass A's network outputs / masks / model keypoints
    const double* blk;                                                                   // the staged host block
    const float* kps_b; const uint8_t* kmask_b;
    float* prior_uv; uint8_t* prior_mask; double* out;
    int has_cov, min_inliers; double kp_std2, chi2_max;
};

__device__ __forceinline__ double sv_chain3(double a0, double b0, double a1, double b1, double a2, double b2) { return fma(a2, b2, fma(a1, b1, a0 * b0)); }
// C[3][4] = A[4][4-implied] @ B: rows 0-2 of the product of two rigid transforms given as [3][4] (their fourth rows are 0 0 0 1)
__device__ __forceinline__ void sv_mul34(const double* A, const double* B, double* C) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) {
            // sum_k A[i][k] B[k][j], k = 0 .. 3 with B[3] = (0, 0, 0, 1)
            const double b3 = j == 3 ? 1.0 : 0.0;
            C[i * 4 + j] = fma(A[i * 4 + 3], b3, sv_chain3(A[i * 4], B[j], A[i * 4 + 1], B[4 + j], A[i * 4 + 2], B[8 + j]));
        }
}

__global__ __launch_bounds__(1024) void slam_vote_kernel(const SlamVoteArgs a) {
    __shared__ double rows[SV_MAX][SV_ROW];
    __shared__ double H[SV_MAX][12], T32[SV_MAX][12], cam[12];
    __shared__ int nrow[SV_MAX], valid[SV_MAX], counts[SV_MAX], best_s, nan_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) { best_s = -1; nan_s = 0; }
    // 1. rows: wave j compacts crop j's valid keypoints in mask order
    if (w < a.n_a) {
        const int j = w;
        const bool m = lane < SV_KP && a.mask[j * SV_KP + lane] != 0;
        const unsigned long long b = __ballot(m);
        const int pos = __popcll(b & ((1ull << lane) - 1ull));
        if (m) {
            for (int c = 0; c < 3; ++c) rows[j][pos * 3 + c] = (double)a.kps_a[(j * SV_KP + lane) * 3 + c];
            for (int c = 0; c < 2; ++c) rows[j][SV_KP * 3 + pos * 2 + c] = (double)a.uv[(j * SV_KP + lane) * 2 + c];
            for (int c = 0; c < 4; ++c) rows[j][SV_KP * 5 + pos * 4 + c] = (double)a.cov[(j * SV_KP + lane) * 4 + c];
        }
        if (lane < 9) rows[j][SV_KP * 9 + lane] = a.blk[HB_A_K + j * 9 + lane];
        if (lane == 0) {
            nrow[j] = __popcll(b);
            valid[j] = (a.accepted[j] != 0 && a.blk[HB_A_IN + j] != 0.0) ? 1 : 0;
            counts[j] = 0;
        }
    }
    __syncthreads();
    // 2. hypotheses and the float32 containers of the map poses (one thread per crop)
    if (tid < a.n_a && valid[tid]) {
        const double* To = a.blk + HB_A_T + tid * 12;             // T_OtoG [3][4]
        double inv[12];
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) inv[i * 4 + j] = To[j * 4 + i];
            inv[i * 4 + 3] = sv_chain3(-To[i], To[3], -To[4 + i], To[7], -To[8 + i], To[11]);      // (-R^T) @ t
        }
        double P[12];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 4; ++j) P[i * 4 + j] = a.T_pnp[tid * 16 + i * 4 + j];
        sv_mul34(P, inv, H[tid]);
        for (int k = 0; k < 12; ++k) T32[tid][k] = (double)(float)To[k];
    }
    __syncthreads();
    // 3. wave i scores hypothesis i against every scored crop j
    if (w < a.n_a && valid[w]) {
        int total = 0;
        bool bad = false;
        for (int j = 0; j < a.n_a; ++j) {
            if (!valid[j] || nrow[j] <= 0) continue;
            double T[12];
            sv_mul34(H[w], T32[j], T);
            bool in = false;
            if (lane < nrow[j]) {
                const double* row = rows[j];
                const double x = row[lane * 3], y = row[lane * 3 + 1], z = row[lane * 3 + 2];
                double p[3], q[3];
                for (int r = 0; r < 3; ++r) p[r] = ((x * T[r * 4] + y * T[r * 4 + 1]) + z * T[r * 4 + 2]) + T[r * 4 + 3];
                for (int r = 0; r < 3; ++r) q[r] = (p[0] * row[SV_KP * 9 + r * 3] + p[1] * row[SV_KP * 9 + r * 3 + 1]) + p[2] * row[SV_KP * 9 + r * 3 + 2];
                if (q[2] > 0.0) {
                    const double rx = row[SV_KP * 3 + lane * 2] - q[0] / q[2], ry = row[SV_KP * 3 + lane * 2 + 1] - q[1] / q[2];
                    double chi2;
                    if (a.has_cov) {
                        const double* c = row + SV_KP * 5 + lane * 4;
                        const double aa = fmax(c[0], 1e-4), d = fmax(c[3], 1e-4), bb = c[1], cc = c[2];
                        chi2 = ((d * rx * rx - (bb + cc) * rx * ry) + aa * ry * ry) / (aa * d - bb * cc);
                    } else {
                        chi2 = (rx * rx + ry * ry) / a.kp_std2;
                    }
                    bad = bad || chi2 != chi2;
                    in = chi2 <= a.chi2_max;
                }
            }
            total += __popcll(__ballot(in));
        }
        if (__ballot(bad) && lane == 0) nan_s = 1;
        if (lane == 0) counts[w] = total;
    }
    __syncthreads();
    // 4. the first hypothesis with the most inliers
    if (tid == 0) {
        int best = -1, best_n = -1, nh = 0;
        for (int i = 0; i < a.n_a; ++i) {
            if (!valid[i]) continue;
            ++nh;
            if (counts[i] >= a.min_inliers && counts[i] > best_n) { best = i; best_n = counts[i]; }
        }
        best_s = best;
        for (int k = 0; k < 12; ++k) { cam[k] = best >= 0 ? H[best][k] : 0.0; a.out[k] = cam[k]; }
        a.out[12] = (double)best; a.out[13] = (double)nh; a.out[14] = (double)best_n;
        for (int i = 0; i < SV_MAX; ++i) a.out[15 + i] = (i < a.n_a && valid[i]) ? (double)counts[i] : -1.0;
        a.out[15 + SV_MAX] = (double)nan_s;
    }
    __syncthreads();
    // 5. priors of pass B's crops
    if (w < a.n_b) {
        const int s = w;
        float u0 = 0.f, u1 = 0.f;
        bool m = false, ok = false;
        if (best_s >= 0 && a.blk[HB_B_IN + s] != 0.0) {
            double T[12];
            sv_mul34(cam, a.blk + HB_B_T + s * 12, T);
            m = lane < SV_KP && a.kmask_b[s * SV_KP + lane] != 0;
            bool front = true;
            if (m) {
                const double x = (double)a.kps_b[(s * SV_KP + lane) * 3], y = (double)a.kps_b[(s * SV_KP + lane) * 3 + 1], z = (double)a.kps_b[(s * SV_KP + lane) * 3 + 2];
                const double* K = a.blk + HB_B_K + s * 9;
                double p[3], q[3];
                for (int r = 0; r < 3; ++r) p[r] = sv_chain3(x, T[r * 4], y, T[r * 4 + 1], z, T[r * 4 + 2]) + T[r * 4 + 3];      // kps @ R^T, then + t
                for (int r = 0; r < 3; ++r) q[r] = sv_chain3(p[0], K[r * 3], p[1], K[r * 3 + 1], p[2], K[r * 3 + 2]);              // kps_in_C @ K^T
                front = q[2] > 0.0;
                u0 = (float)(q[0] / q[2]); u1 = (float)(q[1] / q[2]);
            }
            ok = __ballot(m && !front) == 0ull;               // np.all(uvd[:, 2] > 0) over the object's keypoints
        }
        if (lane < SV_KP) {
            const bool on = ok && m;
            a.prior_uv[(s * SV_KP + lane) * 2] = on ? u0 : 0.f;
            a.prior_uv[(s * SV_KP + lane) * 2 + 1] = on ? u1 : 0.f;
            a.prior_mask[s * SV_KP + lane] = on ? 1 : 0;
        }
    }
}

}  // namespace suo

extern "C" int suo_slam_vote(int n_a, const double* T_pnp_dev, const uint8_t* accepted_dev, const int* n_kp_dev, const float* uv_dev, const float* cov_dev,
                             const uint8_t* mask_dev, const float* model_kps_a_dev, const double* block_dev, int n_b, const float* model_kps_b_dev,
                             const uint8_t* model_mask_b_dev, int has_cov, double kp_std2, double chi2_max, int min_inliers, float* prior_uv_dev,
                             uint8_t* prior_mask_dev, double* out_dev, void* stream) {
    if (n_a <= 0 || n_a > suo::SV_MAX || n_b <= 0 || n_b > suo::SV_MAX || !T_pnp_dev || !accepted_dev || !uv_dev || !cov_dev || !mask_dev || !model_kps_a_dev ||
        !block_dev || !model_kps_b_dev || !model_mask_b_dev || !prior_uv_dev || !prior_mask_dev || !out_dev) {
        suo_set_error("suo_slam_vote: bad arguments (1 <= n_a, n_b <= %d, no null pointers)", suo::SV_MAX);
        return SUO_ERR_ARG;
    }
    suo::SlamVoteArgs a = {n_a, n_b, T_pnp_dev, accepted_dev, n_kp_dev, uv_dev, cov_dev, mask_dev, model_kps_a_dev, block_dev, model_kps_b_dev, model_mask_b_dev,
                           prior_uv_dev, prior_mask_dev, out_dev, has_cov, min_inliers, kp_std2, chi2_max};
    hipLaunchKernelGGL(suo::slam_vote_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}
